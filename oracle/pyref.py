"""CPU oracle (pure Python integers) for the Paillier-in-Halo2 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (paillier_halo2_amd/, include/)
may import this module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may.  It restates, with Python `int`/`pow`, the arithmetic that the reference computes
with num-bigint and with its (un-vendored) halo2curves / halo2-axiom / biguint-halo2
dependencies.

PARITY STATUS: "parity unpinned" for proof / commitment / cell-layout bytes -- the reference
(/root/reference, 548 lines of Rust) holds no golden vectors, KATs or fixtures
 and cannot be built here (no cargo/rustc, un-vendored git
deps).  What IS pinned, by mathematics: every function below has a unique correct output
(modular products, the MSM group element in affine form, the DFT over Fr), so the oracle is
cross-checked three independent ways in tests/test_oracle.py (Python int vs the C restatement
in oracle/pz_oracle.c vs closed forms such as the discrete-log identity for walk bases).

Reference anchors:
  paillier_enc_native / paillier_add_native   -> /root/reference/src/paillier.rs:87-97
  get_biguint (limb fold, little-endian limbs) -> /root/reference/src/paillier.rs:22-30
  encrypt / add operation order                -> /root/reference/src/paillier.rs:32-60, 62-85
  harness order (assign n,g,m,r -> encrypt ..) -> /root/reference/src/bench.rs:33-117
  pow_mod_fixed_exp schedule (LSB->MSB, square every bit, multiply on set bits, acc=1)
      -> dependency biguint-halo2 (git aerius-labs/biguint-halo2, unpinned; Cargo.toml:11),
         call sites paillier.rs:51,55; restated from the halo2-rsa BigUintChip lineage.
  best_multiexp / best_fft semantics -> dependency halo2curves (unpinned, via Cargo.toml:9-10).
"""
from __future__ import annotations

import random
from typing import Iterable, List, Sequence, Tuple

# ----------------------------------------------------------------------------------------
# BN254 constants (verified numerically there and again in tests/test_oracle.py)
# ----------------------------------------------------------------------------------------
FQ_P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
FR_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FR_S = 28  # two-adicity of r-1
FR_GENERATOR = 7
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (FR_R - 1) >> FR_S, FR_R)
MONT_R = 1 << 256
G1_B = 3
G1_GEN = (1, 2)
MASK64 = (1 << 64) - 1


# ----------------------------------------------------------------------------------------
# limb / Montgomery helpers: the in-memory layout of halo2curves Fr/Fq is 4 x u64 little-endian
# limbs of (a * 2^256 mod p)  
# ----------------------------------------------------------------------------------------
def to_limbs(x: int, n: int) -> List[int]:
    assert 0 <= x < (1 << (64 * n)), "value does not fit"
    return [(x >> (64 * i)) & MASK64 for i in range(n)]


def from_limbs(limbs: Iterable[int]) -> int:
    acc = 0
    for i, l in enumerate(limbs):
        acc |= int(l) << (64 * i)
    return acc


def to_mont(x: int, mod: int) -> int:
    return (x * MONT_R) % mod


def from_mont(x: int, mod: int) -> int:
    return (x * pow(MONT_R, -1, mod)) % mod


# ----------------------------------------------------------------------------------------
# Paillier native formulas  (paillier.rs:87-97)
# ----------------------------------------------------------------------------------------
def paillier_enc_native(n: int, g: int, m: int, r: int) -> int:
    """paillier.rs:87-92: n2 = n*n; (g.modpow(m,n2) * r.modpow(n,n2)) % n2"""
    n2 = n * n
    gm = pow(g, m, n2)
    rn = pow(r, n, n2)
    return (gm * rn) % n2


def paillier_add_native(n: int, c1: int, c2: int) -> int:
    """paillier.rs:94-97"""
    n2 = n * n
    return (c1 * c2) % n2


def decompose_biguint(x: int, num_limbs: int, limb_bits: int) -> List[int]:
    """assign_integer's limb split: little-endian limbs of width limb_bits."""
    mask = (1 << limb_bits) - 1
    out = [(x >> (limb_bits * i)) & mask for i in range(num_limbs)]
    assert x >> (limb_bits * num_limbs) == 0, "integer does not fit in num_limbs"
    return out


def get_biguint(limbs: Sequence[int], max_limb_bits: int) -> int:
    """paillier.rs:22-30: fold limbs MSB->LSB with shift max_limb_bits."""
    acc = 0
    for limb in reversed(list(limbs)):
        acc = (acc << max_limb_bits) + int(limb)
    return acc


def exp_bits_lsb_first(e: int) -> List[int]:
    """pow_mod_fixed_exp decomposes e into bits_size(e) bits, LSB first (e == 0 -> no bits)."""
    return [(e >> i) & 1 for i in range(e.bit_length())]


Step = Tuple[int, int, int, int]  # (a, b, q, r) with a*b == q*modulus + r, 0 <= r < modulus


def mul_mod_step(a: int, b: int, modulus: int) -> Step:
    """BigUintChip::mul_mod witness part: full = a*b; (q, r) = full.div_rem(modulus)."""
    q, r = divmod(a * b, modulus)
    return (a, b, q, r)


def pow_mod_fixed_exp_trace(a: int, e: int, modulus: int) -> Tuple[int, List[Step]]:
    """Step trace of BigUintChip::pow_mod_fixed_exp:

        acc = 1; squared = a
        for bit in bits(e) LSB->MSB:
            cur = squared
            squared = square_mod(cur)              # emitted for EVERY bit (the last one is unused)
            if bit: acc = mul_mod(acc, cur)
        return acc

    Returns (acc, steps) with steps in emission order.
    """
    steps: List[Step] = []
    acc = 1
    squared = a
    for bit in exp_bits_lsb_first(e):
        cur = squared
        st = mul_mod_step(cur, cur, modulus)
        steps.append(st)
        squared = st[3]
        if bit:
            st = mul_mod_step(acc, cur, modulus)
            steps.append(st)
            acc = st[3]
    return acc, steps


def encrypt_trace(n: int, g: int, m: int, r: int) -> Tuple[int, List[Step], List[Step], Step]:
    """Operation order of PaillierChip::encrypt (paillier.rs:32-60):
    n2 = n^2 (square + refresh), gm = pow_mod_fixed_exp(g_ext, m, n2),
    rn = pow_mod_fixed_exp(r_ext, n, n2), c = mul_mod(gm, rn, n2)."""
    n2 = n * n
    gm, steps_g = pow_mod_fixed_exp_trace(g, m, n2)
    rn, steps_r = pow_mod_fixed_exp_trace(r, n, n2)
    final = mul_mod_step(gm, rn, n2)
    return final[3], steps_g, steps_r, final


def add_trace(n: int, c1: int, c2: int) -> Tuple[int, Step]:
    """PaillierChip::add (paillier.rs:62-85): one mul_mod(c1_ext, c2_ext, n2)."""
    st = mul_mod_step(c1, c2, n * n)
    return st[3], st


# ----------------------------------------------------------------------------------------
# BN254 G1 (y^2 = x^3 + 3), affine identity encoded as (0, 0) like halo2curves G1Affine
# ----------------------------------------------------------------------------------------
Affine = Tuple[int, int]
Jac = Tuple[int, int, int]
AFF_INF: Affine = (0, 0)
JAC_INF: Jac = (0, 1, 0)


def g1_is_on_curve(pt: Affine) -> bool:
    x, y = pt
    if pt == AFF_INF:
        return True
    return (y * y - x * x * x - G1_B) % FQ_P == 0


def jac_double(pt: Jac) -> Jac:
    X, Y, Z = pt
    if Z == 0 or Y == 0:
        return JAC_INF
    p = FQ_P
    A = X * X % p
    B = Y * Y % p
    C = B * B % p
    D = 2 * ((X + B) * (X + B) - A - C) % p
    E = 3 * A % p
    F = E * E % p
    X3 = (F - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def jac_add(a: Jac, b: Jac) -> Jac:
    if a[2] == 0:
        return b
    if b[2] == 0:
        return a
    p = FQ_P
    X1, Y1, Z1 = a
    X2, Y2, Z2 = b
    Z1Z1 = Z1 * Z1 % p
    Z2Z2 = Z2 * Z2 % p
    U1 = X1 * Z2Z2 % p
    U2 = X2 * Z1Z1 % p
    S1 = Y1 * Z2 * Z2Z2 % p
    S2 = Y2 * Z1 * Z1Z1 % p
    if U1 == U2:
        if S1 == S2:
            return jac_double(a)
        return JAC_INF
    H = (U2 - U1) % p
    R = (S2 - S1) % p
    HH = H * H % p
    HHH = H * HH % p
    V = U1 * HH % p
    X3 = (R * R - HHH - 2 * V) % p
    Y3 = (R * (V - X3) - S1 * HHH) % p
    Z3 = Z1 * Z2 * H % p
    return (X3, Y3, Z3)


def aff_to_jac(pt: Affine) -> Jac:
    if pt == AFF_INF:
        return JAC_INF
    return (pt[0], pt[1], 1)


def jac_to_aff(pt: Jac) -> Affine:
    X, Y, Z = pt
    if Z == 0:
        return AFF_INF
    zi = pow(Z, -1, FQ_P)
    zi2 = zi * zi % FQ_P
    return (X * zi2 % FQ_P, Y * zi2 * zi % FQ_P)


def aff_neg(pt: Affine) -> Affine:
    if pt == AFF_INF:
        return pt
    return (pt[0], (-pt[1]) % FQ_P)


def g1_mul(pt: Affine, k: int) -> Affine:
    k %= FR_R
    acc = JAC_INF
    base = aff_to_jac(pt)
    for i in reversed(range(k.bit_length())):
        acc = jac_double(acc)
        if (k >> i) & 1:
            acc = jac_add(acc, base)
    return jac_to_aff(acc)


def g1_add_aff(a: Affine, b: Affine) -> Affine:
    return jac_to_aff(jac_add(aff_to_jac(a), aff_to_jac(b)))


def msm_naive(scalars: Sequence[int], bases: Sequence[Affine]) -> Affine:
    """Definition of best_multiexp's value: sum_i scalars[i] * bases[i] (double-and-add)."""
    assert len(scalars) == len(bases)
    acc = JAC_INF
    for s, b in zip(scalars, bases):
        acc = jac_add(acc, aff_to_jac(g1_mul(b, s)))
    return jac_to_aff(acc)


def msm_pippenger(scalars: Sequence[int], bases: Sequence[Affine], c: int | None = None) -> Affine:
    """Bucket method, same value as msm_naive; used for the larger golden cases."""
    n = len(scalars)
    assert n == len(bases)
    if n == 0:
        return AFF_INF
    if c is None:
        c = 3 if n < 32 else max(3, n.bit_length() - 2)
    nwin = (254 + c - 1) // c
    total = JAC_INF
    for w in reversed(range(nwin)):
        for _ in range(c):
            total = jac_double(total)
        buckets = [JAC_INF] * ((1 << c) - 1)
        for s, b in zip(scalars, bases):
            d = ((s % FR_R) >> (w * c)) & ((1 << c) - 1)
            if d:
                buckets[d - 1] = jac_add(buckets[d - 1], aff_to_jac(b))
        run = JAC_INF
        acc = JAC_INF
        for bkt in reversed(buckets):
            run = jac_add(run, bkt)
            acc = jac_add(acc, run)
        total = jac_add(total, acc)
    return jac_to_aff(total)


def walk_bases(n: int, s: int, t: int) -> List[Affine]:
    """P_i = [s + i*t] G for i < n  (the seeded walk of SURVEY.md section 8d, config c4).
    Known discrete logs give the size-independent MSM check  sum k_i P_i = [sum k_i (s+i t)] G."""
    out: List[Affine] = []
    step = aff_to_jac(g1_mul(G1_GEN, t))
    cur = aff_to_jac(g1_mul(G1_GEN, s))
    for _ in range(n):
        out.append(jac_to_aff(cur))
        cur = jac_add(cur, step)
    return out


def msm_walk_expected(scalars: Sequence[int], s: int, t: int) -> Affine:
    k = 0
    for i, sc in enumerate(scalars):
        k = (k + sc * (s + i * t)) % FR_R
    return g1_mul(G1_GEN, k)


# ----------------------------------------------------------------------------------------
# NTT over Fr: value semantics of halo2curves best_fft(a, omega, log_n):
#   out[k] = sum_j a[j] * omega^(j*k)   (in-place, natural order in and out)
# ----------------------------------------------------------------------------------------
def fr_omega(log_n: int) -> int:
    """omega for a 2^log_n domain = ROOT_OF_UNITY^(2^(28-log_n))  (SURVEY.md section 8a row a8)."""
    assert 0 <= log_n <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - log_n), FR_R)


def ntt_naive(a: Sequence[int], omega: int) -> List[int]:
    n = len(a)
    return [sum(a[j] * pow(omega, j * k, FR_R) for j in range(n)) % FR_R for k in range(n)]


def ntt(a: Sequence[int], omega: int) -> List[int]:
    """Iterative radix-2 DIT following best_fft's structure: bit-reverse, then log_n butterfly
    layers with twiddle omega^(n/(2m))."""
    n = len(a)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    a = list(a)
    for k in range(n):
        rk = int(format(k, "0%db" % log_n)[::-1], 2) if log_n else 0
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), FR_R)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % FR_R
                a[k + j + m] = (a[k + j] - t) % FR_R
                a[k + j] = (a[k + j] + t) % FR_R
                w = w * w_m % FR_R
        m *= 2
    return a


def intt(a: Sequence[int], omega: int) -> List[int]:
    """EvaluationDomain::ifft = best_fft with omega^-1 then scale by n^-1."""
    n = len(a)
    out = ntt(a, pow(omega, -1, FR_R))
    ninv = pow(n, -1, FR_R)
    return [x * ninv % FR_R for x in out]


def coset_scale(a: Sequence[int], g: int) -> List[int]:
    """distribute_powers: a[i] *= g^i (used by coeff_to_extended before the extended NTT)."""
    out = []
    cur = 1
    for x in a:
        out.append(x * cur % FR_R)
        cur = cur * g % FR_R
    return out


# ----------------------------------------------------------------------------------------
# "next" rows: the prover steps between commitments and NTTs.  The reference
# reaches them only through create_proof (/root/reference/src/bench.rs:161-171); the formulas restate the published
# halo2 protocol as halo2-axiom implements it (dependency, tag [D]) -- pinned by their defining identities, which
# tests/test_oracle.py checks (z recurrence, divisibility by X^n - 1, p(X) - p(x) = (X - x) q(X)).
# ----------------------------------------------------------------------------------------
def batch_invert(a: Sequence[int]) -> List[int]:
    """halo2 BatchInvert: zeros stay zero."""
    return [pow(x, -1, FR_R) if x % FR_R else 0 for x in a]


def prefix_product(a: Sequence[int], z0: int = 1) -> List[int]:
    """z[0] = z0, z[i+1] = z[i] * a[i]  (n outputs; a[n-1] is not consumed)."""
    z = [z0 % FR_R]
    for x in a[:-1]:
        z.append(z[-1] * x % FR_R)
    return z


def permutation_product(cols: Sequence[Sequence[int]], sigmas: Sequence[Sequence[int]], omega: int, beta: int, gamma: int,
                        delta_start: int, delta: int, z0: int = 1) -> List[int]:
    """permutation::Argument::commit for one chunk of columns (no blinding rows):
    z[i+1] = z[i] * prod_j (v_j[i] + beta*delta_start*delta^j*omega^i + gamma) / (v_j[i] + beta*sigma_j[i] + gamma)."""
    n = len(cols[0]) if cols else 0
    mv = []
    w = 1
    for i in range(n):
        num = den = 1
        d = beta * delta_start % FR_R * w % FR_R
        for v, sg in zip(cols, sigmas):
            num = num * ((v[i] + d + gamma) % FR_R) % FR_R
            den = den * ((v[i] + beta * sg[i] + gamma) % FR_R) % FR_R
            d = d * delta % FR_R
        mv.append(num * (pow(den, -1, FR_R) if den else 0) % FR_R)
        w = w * omega % FR_R
    return prefix_product(mv, z0) if n else []


def permute_expression_pair(inp: Sequence[int], tab: Sequence[int]) -> Tuple[List[int], List[int]]:
    """halo2 lookup::prover::permute_expression_pair on the usable rows: input sorted ascending; the table cell
    equals the input cell where a run of equal inputs starts, the other rows take the left-over table values in
    ascending (BTreeMap) order.  Raises if an input value is missing from the table (halo2 returns an error)."""
    A = sorted(x % FR_R for x in inp)
    left = {}
    for t in tab:
        left[t % FR_R] = left.get(t % FR_R, 0) + 1
    S: List[int] = [0] * len(A)
    repeated = []
    for i, v in enumerate(A):
        if i == 0 or v != A[i - 1]:
            if left.get(v, 0) == 0:
                raise ValueError("lookup input %d not in the table" % v)
            left[v] -= 1
            S[i] = v
        else:
            repeated.append(i)
    rest = [v for v in sorted(left) for _ in range(left[v])]
    assert len(rest) == len(repeated)
    for i, v in zip(repeated, rest):
        S[i] = v
    return A, S


def lookup_product(A: Sequence[int], S: Sequence[int], Ap: Sequence[int], Sp: Sequence[int], beta: int, gamma: int,
                   z0: int = 1) -> List[int]:
    """z[i+1] = z[i] (A[i]+beta)(S[i]+gamma) / ((A'[i]+beta)(S'[i]+gamma))"""
    mv = []
    for a, s_, ap, sp in zip(A, S, Ap, Sp):
        den = (ap + beta) * (sp + gamma) % FR_R
        mv.append((a + beta) * (s_ + gamma) % FR_R * (pow(den, -1, FR_R) if den else 0) % FR_R)
    return prefix_product(mv, z0) if mv else []


def quotient_gate(adv_ext: Sequence[Sequence[int]], sel_ext: Sequence[Sequence[int]], step: int, y: int,
                  h: Sequence[int], rows=None, N=None):
    """evaluate_h, custom-gate part: per column (in order) h = h*y + sel*(a0 + a1*a2 - a3), rotations = step indices
    of the extended domain (halo2-lib's vertical gate q*(a + b*c - d), rows i..i+3).
    rows / N: only these rows of a domain of N points (the arrays may then be sparse {index: value} maps holding just the rows
    touched); returns {row: value}.  Same for the two functions below."""
    N = len(h) if N is None else N
    out = list(h) if rows is None else {i: h[i] for i in rows}
    for a, q in zip(adv_ext, sel_ext):
        for i in (range(N) if rows is None else rows):
            e = (a[i] + a[(i + step) % N] * a[(i + 2 * step) % N] - a[(i + 3 * step) % N]) % FR_R
            out[i] = (out[i] * y + q[i] * e) % FR_R
    return out


def quotient_permutation(cols_ext, sigma_ext, z_ext, chunk_len: int, step: int, last_rot: int, l0, l_last, l_active,
                         beta: int, gamma: int, delta: int, x0: int, w_ext: int, y: int, h, rows=None, N=None):
    """evaluate_h "Permutations" block (halo2 plonk/evaluation.rs), extended domain, X_i = x0 * w_ext^i."""
    N = len(h) if N is None else N
    out = list(h) if rows is None else {i: h[i] for i in rows}
    ns = len(z_ext)
    for i in (range(N) if rows is None else rows):
        xi = x0 * pow(w_ext, i, FR_R) % FR_R
        inx, ila = (i + step) % N, (i - last_rot * step) % N
        v = out[i]
        v = (v * y + (1 - z_ext[0][i]) * l0[i]) % FR_R
        zl = z_ext[ns - 1][i]
        v = (v * y + (zl * zl - zl) * l_last[i]) % FR_R
        for j in range(1, ns):
            v = (v * y + (z_ext[j][i] - z_ext[j - 1][ila]) * l0[i]) % FR_R
        cur = beta * xi % FR_R
        c = 0
        for j in range(ns):
            left, right = z_ext[j][inx], z_ext[j][i]
            for _ in range(chunk_len):
                if c >= len(cols_ext):
                    break
                left = left * ((cols_ext[c][i] + beta * sigma_ext[c][i] + gamma) % FR_R) % FR_R
                right = right * ((cols_ext[c][i] + cur + gamma) % FR_R) % FR_R
                cur = cur * delta % FR_R
                c += 1
            v = (v * y + (left - right) * l_active[i]) % FR_R
        out[i] = v
    return out


def quotient_lookup(a_ext, s_ext, ap_ext, sp_ext, z_ext, step: int, l0, l_last, l_active, beta: int, gamma: int, y: int,
                    h, rows=None, N=None):
    """evaluate_h "Lookups" block for single-expression lookups sharing the table s."""
    N = len(h) if N is None else N
    out = list(h) if rows is None else {i: h[i] for i in rows}
    for i in (range(N) if rows is None else rows):
        inx, ipr = (i + step) % N, (i - step) % N
        v = out[i]
        for a, ap, sp, z in zip(a_ext, ap_ext, sp_ext, z_ext):
            v = (v * y + (1 - z[i]) * l0[i]) % FR_R
            v = (v * y + (z[i] * z[i] - z[i]) * l_last[i]) % FR_R
            lhs = z[inx] * (ap[i] + beta) % FR_R * (sp[i] + gamma) % FR_R
            rhs = z[i] * (a[i] + beta) % FR_R * (s_ext[i] + gamma) % FR_R
            v = (v * y + (lhs - rhs) * l_active[i]) % FR_R
            ams = (ap[i] - sp[i]) % FR_R
            v = (v * y + ams * l0[i]) % FR_R
            v = (v * y + ams * (ap[i] - ap[ipr]) % FR_R * l_active[i]) % FR_R
        out[i] = v
    return out


def quotient_finish(h: Sequence[int], log_n: int, log_e: int, coset_g: int, omega_ext: int) -> List[int]:
    """divide by the vanishing polynomial X^n - 1 at the points coset_g * omega_ext^i."""
    E = 1 << log_e
    tinv = [pow((pow(coset_g * pow(omega_ext, r, FR_R) % FR_R, 1 << log_n, FR_R) - 1) % FR_R, -1, FR_R) for r in range(E)]
    return [x * tinv[i % E] % FR_R for i, x in enumerate(h)]


def distribute_powers(a: Sequence[int], g: int, c: int = 1) -> List[int]:
    out, cur = [], c % FR_R
    for x in a:
        out.append(x * cur % FR_R)
        cur = cur * g % FR_R
    return out


def kate_division(a: Sequence[int], x: int) -> List[int]:
    """halo2 kate_division: q = (p - p(x)) / (X - x); n-1 coefficients, returned zero-padded to n."""
    n = len(a)
    q = [0] * n
    cur = 0
    for i in range(n - 1, 0, -1):
        cur = (a[i] + x * cur) % FR_R
        q[i - 1] = cur
    return q


def poly_eval(a: Sequence[int], x: int) -> int:
    acc = 0
    for c in reversed(a):
        acc = (acc * x + c) % FR_R
    return acc


# ----------------------------------------------------------------------------------------
# seeded synthetic inputs
# ----------------------------------------------------------------------------------------
def synth_paillier_inputs(enc_bits: int, seed: int, standard_g: bool = True):
    """n = p*q forced to exactly enc_bits bits (p, q random odd enc_bits/2-bit with top bit set;
    primality NOT required -- the reference feeds a raw random n, paillier.rs:173), g = n+1 or
    random < n, m uniform in [0,n), r uniform in [1,n)."""
    rng = random.Random(seed)
    half = enc_bits // 2
    while True:
        p = rng.getrandbits(half) | (1 << (half - 1)) | 1
        q = rng.getrandbits(half) | (1 << (half - 1)) | 1
        n = p * q
        if n.bit_length() == enc_bits:
            break
    g = n + 1 if standard_g else rng.randrange(2, n)
    m = rng.randrange(0, n)
    r = rng.randrange(1, n)
    return n, g, m, r


def witness_like_scalars(n: int, seed: int) -> List[int]:
    """c4 scalar mix (ii): 60 % < 2^16, 30 % < 2^64, 10 % < 2^135  (SURVEY.md section 8d)."""
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        u = rng.random()
        if u < 0.6:
            out.append(rng.getrandbits(16))
        elif u < 0.9:
            out.append(rng.getrandbits(64))
        else:
            out.append(rng.getrandbits(135))
    return out


# ----------------------------------------------------------------------------------------
# K4 oracle: the advice / lookup cell stream of one BigUintChip::mul_mod step, as canonical
# field integers, in the layout of paillier_halo2_amd/layout.py.  The cell
# patterns restate halo2-lib v0.4 gate primitives [D]; layout parity with the reference's floating
# dependency versions is unpinned, the VALUES follow from (a, b, q, r, n).
# ----------------------------------------------------------------------------------------
def _range_check_cells(x: int, bits: int, lb: int):
    """RangeChip::range_check(x, bits) -> (advice cells, lookup cells)"""
    k = -(-bits // lb)
    rem = bits % lb
    mask = (1 << lb) - 1
    digs = [(x >> (lb * i)) & mask for i in range(k)]
    adv = []
    if k > 1:
        adv.append(digs[0])
        acc = digs[0]
        for g in range(1, k):
            acc += digs[g] << (lb * g)
            adv += [digs[g], 1 << (lb * g), acc]
    lk = list(digs)
    last = digs[-1]
    if rem == 1:
        adv += [0, last, last, last]
    elif rem > 1:
        chk = last << (lb - rem)
        adv += [0, last, 1 << (lb - rem), chk]
        lk.append(chk)
    return adv, lk


def _is_zero_cells(d: int):
    d %= FR_R
    if d == 0:
        return [1, 0, 1, 1, 0, 0, 1, 0], 1
    return [0, d, pow(d, -1, FR_R), 1, 0, d, 0, 0], 0


def _is_equal_cells(x: int, y: int):
    d = (x - y) % FR_R
    z, bit = _is_zero_cells(d)
    return [d, y % FR_R, 1, x % FR_R] + z, bit


def _div_mod_cells(v: int, limb_bits: int):
    """BigUintChip::div_mod_unsafe(v, 2^limb_bits): 22 cells, returns (cells, quotient, remainder)"""
    base = 1 << limb_bits
    qd, rd = v >> limb_bits, v & (base - 1)
    prod = qd * base
    eq, _ = _is_equal_cells(rd, v - prod)
    return [qd, rd, 0, qd, base, prod, v - prod, prod, 1, v] + eq, qd, rd


def _mul_cells(x_limbs, y_limbs, D):
    """load_zero + truncated mul_no_carry over D limbs; returns (cells, product limbs)"""
    L = len(x_limbs)
    xe = list(x_limbs) + [0] * (D - L)
    ye = list(y_limbs) + [0] * (D - L)
    cells = [0]
    prod = []
    for i in range(D):
        cells.append(0)
        s = 0
        for j in range(i + 1):
            s += xe[j] * ye[i - j]
            cells += [xe[j], ye[i - j], s]
        prod.append(s)
    return cells, prod


def expand_mul_mod_cells(a: int, b: int, q: int, r: int, n: int, L: int, lookup_bits: int, limb_bits: int = 64):
    """-> (advice cells, lookup cells) as canonical integers mod FR_R"""
    lb = lookup_bits
    D = 2 * L - 1
    base = 1 << limb_bits
    lim = lambda x: decompose_biguint(x, L, limb_bits)
    al, bl, ql, rl, nl = lim(a), lim(b), lim(q), lim(r), lim(n)
    adv, lk = [], []
    # 1. assign_integer q, n, r
    for X in (ql, nl, rl):
        adv += X
        for x in X:
            c, l = _range_check_cells(x, limb_bits, lb)
            adv += c
            lk += l
    # 2. the two limb convolutions
    c_ab, p_ab = _mul_cells(al, bl, D)
    c_qn, p_qn = _mul_cells(ql, nl, D)
    adv += c_ab + c_qn
    # 3. qn + r
    qnr = list(p_qn)
    for i in range(L):
        adv += [p_qn[i], 1, rl[i], p_qn[i] + rl[i]]
        qnr[i] = p_qn[i] + rl[i]
    # 4. is_equal_muled(ab, qn + r)
    m = base - 1
    MAX = L * m * m + m
    cb = (2 * MAX).bit_length() - limb_bits
    adv += [0, 1]
    carry, accx, eq_bit = 0, 0, 1
    for i in range(D):
        diff = p_ab[i] - qnr[i]
        adv += [diff, qnr[i], 1, p_ab[i]]
        s = diff + carry + MAX
        adv += [diff, carry, 1, diff + carry, MAX, 1, s]
        assert s >= 0
        dm, new_carry, c = _div_mod_cells(s, limb_bits)
        adv += dm
        t = accx + MAX
        adv += [accx, 1, MAX, t]
        dm2, q_acc, mod_acc = _div_mod_cells(t, limb_bits)
        adv += dm2
        eqc, e = _is_equal_cells(c, mod_acc)
        adv += eqc
        adv += [0, eq_bit, e, eq_bit & e]
        eq_bit &= e
        accx = q_acc
        if i < D - 1:
            cc, ll = _range_check_cells(new_carry, cb, lb)
            adv += cc
            lk += ll
        else:
            eqc, e = _is_equal_cells(new_carry, accx)
            adv += eqc
            adv += [0, eq_bit, e, eq_bit & e]
            eq_bit &= e
        carry = new_carry
    # 5. r < n
    borrow = 0
    for i in range(L):
        nb = nl[i] + borrow
        adv += [nl[i], 1, borrow, nb]
        shift = rl[i] - nb + base
        lt = 1 if rl[i] < nb else 0
        out = shift & (base - 1)
        adv += [shift, lt, out]
        adv += [rl[i], lt, base, rl[i] + lt * base]
        cc, ll = _range_check_cells(out, limb_bits, lb)
        adv += cc
        lk += ll
        borrow = lt
    adv.append(borrow)
    return [x % FR_R for x in adv], [x % FR_R for x in lk]


# ----------------------------------------------------------------------------------------
# MockProver analogue for the K4 cell stream: halo2-lib has ONE gate, q * (a + b*c - d) = 0 on four
# vertically consecutive cells [a, b, c, d].  gate_offsets_mul_mod lists where BigUintChip::mul_mod
# enables the selector in one step's block (same walk as expand_mul_mod_cells); check_gates verifies the
# identity on every enabled window.  This is the "circuit satisfied" half of the reference's tests
# (base_test().expect_satisfied(true), paillier.rs:167-171) restricted to the gate constraints of the
# hot-path cells; copy constraints and lookups are dependency-defined wiring, not values.
# ----------------------------------------------------------------------------------------
def _rc_gates(off, bits, lb):
    k = -(-bits // lb)
    rem = bits % lb
    g = []
    n = 0
    if k > 1:
        g += [off + 3 * i for i in range(k - 1)]  # inner product: windows [acc_{i}, d_{i+1}, base_{i+1}, acc_{i+1}]
        n = 1 + 3 * (k - 1)
    if rem >= 1:
        g.append(off + n)
        n += 4
    return g, n


def gate_offsets_mul_mod(L: int, lookup_bits: int, limb_bits: int = 64):
    lb = lookup_bits
    D = 2 * L - 1
    gates = []
    off = 0
    for _ in range(3):  # assign q, n, r
        off += L
        for _ in range(L):
            g, n = _rc_gates(off, limb_bits, lb)
            gates += g
            off += n
    for _ in range(2):  # two convolutions: rows [0, (x, y, s)...]: windows start at every sum cell / the leading 0
        off += 1
        for i in range(D):
            gates += [off + 3 * j for j in range(i + 1)]
            off += 1 + 3 * (i + 1)
    for _ in range(L):  # qn + r : [a, 1, b, out]
        gates.append(off)
        off += 4
    m = (1 << limb_bits) - 1
    cb = (2 * (L * m * m + m)).bit_length() - limb_bits
    off += 2
    def is_equal_gates(o):  # sub gate + is_zero's two gates (rows 0 and 4 of its 8 cells)
        return [o, o + 4, o + 8]
    def div_mod_gates(o):  # [qd, rd] witnesses, mul gate, sub gate, is_equal
        return [o + 2, o + 6] + is_equal_gates(o + 10)
    for i in range(D):
        gates.append(off)                    # sub
        gates += [off + 4, off + 7]          # sum of 3: [x0, x1, 1, s1, x2, 1, s2]
        gates += div_mod_gates(off + 11)
        gates.append(off + 33)               # add
        gates += div_mod_gates(off + 37)
        gates += is_equal_gates(off + 59)
        gates.append(off + 71)               # and
        off += 75
        if i < D - 1:
            g, n = _rc_gates(off, cb, lb)
            gates += g
            off += n
        else:
            gates += is_equal_gates(off)
            gates.append(off + 12)
            off += 16
    for _ in range(L):  # r < n
        gates.append(off)          # add [n_i, 1, borrow, nb]
        gates.append(off + 7)      # [r_i, lt, 2^64, r_i + lt*2^64]
        off += 11                  # (cells 4..6 [shift, lt, out] are witnesses tied by copy constraints)
        g, n = _rc_gates(off, limb_bits, lb)
        gates += g
        off += n
    off += 1
    return gates, off


def check_gates(cells, gates):
    """returns the list of gate offsets whose identity a + b*c == d (mod r) fails"""
    bad = []
    for o in gates:
        a, b, c, d = cells[o], cells[o + 1], cells[o + 2], cells[o + 3]
        if (a + b * c - d) % FR_R != 0:
            bad.append(o)
    return bad


# ----------------------------------------------------------------------------------------
# The WHOLE circuit's cell stream: every BigUintChip / Context call of the drivers
# bench.rs:33-75 (paillier_enc_test) and bench.rs:77-117 (paillier_enc_add_test) with PaillierChip::encrypt / add
# (paillier.rs:32-60, 62-85) in between, in call order.  The per-operation cell patterns restate the biguint-halo2 /
# halo2-lib dependency code [D] (halo2-rsa lineage) like expand_mul_mod_cells above: layout parity with the reference's
# floating dependency versions is unpinned, the VALUES follow from the inputs.  Conventions of this restatement:
#   assign_integer(x, bits)   : the limbs as witness cells, then range_check(limb, limb_bits) each      (bench.rs:44-66)
#   square(n) = mul(n, n)     : load_zero cell + truncated mul_no_carry over 2 Ln - 1 limbs               (paillier.rs:39)
#   refresh(n2, aux)          : load_zero; per limb i: (inc[i] + 1) x div_mod_unsafe(limb, 2^W) (22 cells), the j-th
#                               remainder (j > 0) added into limb i + j (gate.add, 4 cells); the last quotient is
#                               constrained to the constant 0 (no advice cell); then range_check of every fresh limb
#                               (paillier.rs:40-45; inc = RefreshAux::new(limb_bits, Ln, Ln).increased_limbs_vec)
#   ctx.load_zero()           : one cell per call                                                        (paillier.rs:47)
#   extend_limbs              : no cells (copies of the zero cell)                                      (paillier.rs:49,53)
#   pow_mod_fixed_exp         : assign_constant(1) (one cell), load_zero (one cell), then its mul_mod steps  (:51,55)
#   assert_equal_fresh(a, b)  : load_zero, load_constant(1), per limb is_equal (12 cells) + and (4 cells); the result
#                               is constrained to the constant 1                                        (bench.rs:74)
# ----------------------------------------------------------------------------------------
def refresh_aux(limb_bits: int, num_limbs_l: int, num_limbs_r: int) -> List[int]:
    """RefreshAux::new(..).increased_limbs_vec: how many extra limbs each product limb's maximal value spills into"""
    mx = (1 << limb_bits) - 1
    d = num_limbs_l + num_limbs_r - 1
    muled = []
    for i in range(d):
        cnt = sum(1 for j in range(num_limbs_l) if 0 <= i - j < num_limbs_r)
        muled.append(cnt * mx * mx)
    inc = []
    cur = 0
    while cur < len(muled):
        bits = muled[cur].bit_length()
        chunks = max(1, -(-bits // limb_bits))
        inc.append(chunks - 1)
        val = muled[cur]
        for i in range(chunks):
            piece = val & mx
            val >>= limb_bits
            if cur + i < len(muled):
                muled[cur + i] = piece if i == 0 else muled[cur + i] + piece
            else:
                muled.append(piece)
        cur += 1
    return inc


def expand_assign_cells(x: int, num_limbs: int, limb_bits: int, lb: int):
    limbs = decompose_biguint(x, num_limbs, limb_bits)
    adv, lk = list(limbs), []
    for v in limbs:
        c, l = _range_check_cells(v, limb_bits, lb)
        adv += c
        lk += l
    return adv, lk


def expand_refresh_cells(prod: Sequence[int], inc: Sequence[int], limb_bits: int, lb: int):
    """-> (advice, lookup, fresh limbs)"""
    nf = len(inc)
    cur = list(prod) + [0] * (nf - len(prod))
    adv, lk = [0], []
    for i in range(nf):
        limb = cur[i]
        for j in range(inc[i] + 1):
            dm, q, n = _div_mod_cells(limb, limb_bits)
            adv += dm
            if j == 0:
                cur[i] = n
            else:
                adv += [cur[i + j], 1, n, cur[i + j] + n]
                cur[i + j] += n
            limb = q
        assert limb == 0, "refresh: the limb does not decompose in inc[i] + 1 pieces"
    for v in cur:
        c, l = _range_check_cells(v, limb_bits, lb)
        adv += c
        lk += l
    return adv, lk, cur


def expand_assert_equal_fresh_cells(a_limbs: Sequence[int], b_limbs: Sequence[int]):
    adv = [0, 1]
    eq = 1
    for x, y in zip(a_limbs, b_limbs):
        c, e = _is_equal_cells(x, y)
        adv += c
        adv += [0, eq, e, eq & e]
        eq &= e
    return adv, eq


def expand_circuit_cells(kind: str, n: int, g: int, x: int, y: int, res: int, enc_bits: int, limb_bits: int, lb: int):
    """kind 'encrypt': (x, y) = (m, r), the cells of paillier_enc_test; kind 'add': (x, y) = (c1, c2), paillier_enc_add_test.
    -> (advice cells, lookup cells, segment offsets {name: (advice offset, lookup offset)}) as canonical integers"""
    Ln = enc_bits // limb_bits
    L = 2 * Ln
    adv, lk, seg = [], [], {}

    def put(name, a, l=()):
        seg[name] = (len(adv), len(lk))
        adv.extend(a)
        lk.extend(l)

    for name, v in (("assign_n", n), ("assign_g", g), ("assign_x", x), ("assign_y", y)):
        put(name, *expand_assign_cells(v, Ln, limb_bits, lb))
    nl = decompose_biguint(n, Ln, limb_bits)
    sq_cells, prod = _mul_cells(nl, nl, 2 * Ln - 1)
    put("square", sq_cells)
    inc = refresh_aux(limb_bits, Ln, Ln)
    assert len(inc) == L
    r_adv, r_lk, fresh = expand_refresh_cells(prod, inc, limb_bits, lb)
    assert get_biguint(fresh, limb_bits) == n * n
    put("refresh", r_adv, r_lk)
    put("load_zero", [0])
    n2 = n * n
    if kind == "encrypt":
        c, sg, sr, fin = encrypt_trace(n, g, x, y)
        for name, steps in (("pow_g", sg), ("pow_r", sr)):
            put(name, [1, 0])
            for st in steps:
                a_, l_ = expand_mul_mod_cells(*st, n2, L, lb, limb_bits)
                adv.extend(a_)
                lk.extend(l_)
    else:
        c, fin = add_trace(n, x, y)
    put("final", *expand_mul_mod_cells(*fin, n2, L, lb, limb_bits))
    put("assign_res", *expand_assign_cells(res, L, limb_bits, lb))
    ae, bit = expand_assert_equal_fresh_cells(decompose_biguint(c, L, limb_bits), decompose_biguint(res, L, limb_bits))
    put("assert_equal", ae)
    seg["end"] = (len(adv), len(lk))
    seg["satisfied"] = bool(bit)
    return [v % FR_R for v in adv], [v % FR_R for v in lk], seg


def _cut_windows(pieces, adv_windows, lk_windows):
    """pieces: [(advice cells, lookup cells, maker of both lists)] in stream order -> (total advice cells, total lookup cells,
    [cells of each advice window], [cells of each lookup window]); only the pieces a window touches are expanded"""
    tot_a, tot_l = sum(p[0] for p in pieces), sum(p[1] for p in pieces)

    def cut(windows, which):
        outs = []
        for lo, hi in windows:
            out, off = [], 0
            for p_ in pieces:
                ln = p_[which]
                if ln and off < hi and off + ln > lo:
                    cells = p_[2]()[which]
                    assert len(cells) == ln
                    out += cells[max(lo - off, 0): hi - off]
                off += ln
                if off >= hi:
                    break
            outs.append([v % FR_R for v in out])
        return outs

    return tot_a, tot_l, cut(adv_windows, 0), cut(lk_windows, 1)


def encrypt_circuit_cells_windows(n: int, g: int, m: int, r: int, res: int, enc_bits: int, limb_bits: int, lb: int, adv_windows,
                                   lk_windows=()):
    """The cells [lo, hi) of expand_circuit_cells('encrypt', ...)'s advice / lookup streams WITHOUT building the streams (4e8
    cells at 2048 bits): every mul_mod has the same cell count, so only the steps a window touches are expanded.
    -> (total advice cells, total lookup cells, [cells of each advice window], [cells of each lookup window])"""
    Ln = enc_bits // limb_bits
    L = 2 * Ln
    n2 = n * n
    pre_a, pre_l = [], []
    for v in (n, g, m, r):
        a_, l_ = expand_assign_cells(v, Ln, limb_bits, lb)
        pre_a += a_
        pre_l += l_
    nl = decompose_biguint(n, Ln, limb_bits)
    sq_cells, prod = _mul_cells(nl, nl, 2 * Ln - 1)
    pre_a += sq_cells
    r_adv, r_lk, _ = expand_refresh_cells(prod, refresh_aux(limb_bits, Ln, Ln), limb_bits, lb)
    pre_a += r_adv + [0]          # refresh, load_zero
    pre_l += r_lk
    c, sg, sr, fin = encrypt_trace(n, g, m, r)
    fa, fl = expand_mul_mod_cells(*fin, n2, L, lb, limb_bits)
    cps, lps = len(fa), len(fl)
    ta, tl = expand_assign_cells(res, L, limb_bits, lb)
    ae, _ = expand_assert_equal_fresh_cells(decompose_biguint(c, L, limb_bits), decompose_biguint(res, L, limb_bits))
    # pieces in stream order: (advice cells, lookup cells, maker of both lists)
    pieces = [(len(pre_a), len(pre_l), lambda: (pre_a, pre_l))]
    for steps in (sg, sr):
        pieces.append((2, 0, lambda: ([1, 0], [])))
        pieces += [(cps, lps, (lambda st=st: expand_mul_mod_cells(*st, n2, L, lb, limb_bits))) for st in steps]
    pieces.append((cps, lps, lambda: (fa, fl)))
    pieces.append((len(ta), len(tl), lambda: (ta, tl)))
    pieces.append((len(ae), 0, lambda: (ae, [])))
    return _cut_windows(pieces, adv_windows, lk_windows)


# gate positions (MockProver analogue) of the operations above, relative to the operation's first cell
def gate_offsets_assign(num_limbs: int, limb_bits: int, lb: int):
    gates, off = [], num_limbs
    for _ in range(num_limbs):
        g, n = _rc_gates(off, limb_bits, lb)
        gates += g
        off += n
    return gates, off


def gate_offsets_square(Ln: int):
    gates, off = [], 1
    for i in range(2 * Ln - 1):
        gates += [off + 3 * j for j in range(i + 1)]
        off += 1 + 3 * (i + 1)
    return gates, off


def gate_offsets_refresh(inc: Sequence[int], limb_bits: int, lb: int):
    gates, off = [], 1
    for i in range(len(inc)):
        for j in range(inc[i] + 1):
            gates += [off + 2, off + 6, off + 10, off + 14, off + 18]   # div_mod_unsafe: mul, sub, is_equal (3 gates)
            off += 22
            if j:
                gates.append(off)
                off += 4
    for _ in range(len(inc)):
        g, n = _rc_gates(off, limb_bits, lb)
        gates += g
        off += n
    return gates, off


def gate_offsets_assert_equal(L: int):
    gates, off = [], 2
    for _ in range(L):
        gates += [off, off + 4, off + 8, off + 12]
        off += 16
    return gates, off


def gate_offsets_circuit(kind: str, enc_bits: int, limb_bits: int, lb: int, n_steps_g: int = 0, n_steps_r: int = 0):
    """every enabled gate window of the whole circuit's advice stream, and the stream's length"""
    Ln = enc_bits // limb_bits
    L = 2 * Ln
    gates, off = [], 0

    def add(part):
        nonlocal off
        g, n = part
        gates.extend(off + o for o in g)
        off += n

    for _ in range(4):
        add(gate_offsets_assign(Ln, limb_bits, lb))
    add(gate_offsets_square(Ln))
    add(gate_offsets_refresh(refresh_aux(limb_bits, Ln, Ln), limb_bits, lb))
    off += 1
    mm = gate_offsets_mul_mod(L, lb, limb_bits)
    if kind == "encrypt":
        for ns in (n_steps_g, n_steps_r):
            off += 2
            for _ in range(ns):
                add(mm)
    add(mm)
    add(gate_offsets_assign(L, limb_bits, lb))
    add(gate_offsets_assert_equal(L))
    return gates, off


def gate_mask_circuit(kind: str, enc_bits: int, limb_bits: int, lb: int, n_steps_g: int = 0, n_steps_r: int = 0):
    """gate_offsets_circuit as a numpy uint8 selector over the whole advice stream (1 where a gate window starts), built
    from per-operation masks -- the mul_mod block's mask is tiled, so the 4e8-cell stream of a 2048-bit key costs no Python
    loop.  -> (mask, stream length)"""
    import numpy as np

    Ln = enc_bits // limb_bits
    L = 2 * Ln

    def mask(part):
        g, n = part
        m = np.zeros(n, dtype=np.uint8)
        m[np.asarray(g, dtype=np.int64)] = 1
        return m

    parts = [mask(gate_offsets_assign(Ln, limb_bits, lb))] * 4
    parts.append(mask(gate_offsets_square(Ln)))
    parts.append(mask(gate_offsets_refresh(refresh_aux(limb_bits, Ln, Ln), limb_bits, lb)))
    parts.append(np.zeros(1, dtype=np.uint8))
    mm = mask(gate_offsets_mul_mod(L, lb, limb_bits))
    if kind == "encrypt":
        for ns in (n_steps_g, n_steps_r):
            parts.append(np.zeros(2, dtype=np.uint8))
            parts.append(np.tile(mm, ns))
    parts.append(mm)
    parts.append(mask(gate_offsets_assign(L, limb_bits, lb)))
    parts.append(mask(gate_offsets_assert_equal(L)))
    out = np.concatenate(parts)
    return out, out.shape[0]


# ----------------------------------------------------------------------------------------
# Copy constraints (the other half of MockProver's check): which cells of the stream must hold the SAME value because the
# chip passes one assigned cell to several places.  expand_circuit_cells_wired walks the drivers exactly like
# expand_circuit_cells, but every cell that is a copy names the cell it copies; the result is the value stream (must equal
# expand_circuit_cells') and the list of (source cell, copy cell) pairs:
#   * range_check: the recomposed accumulator equals the checked cell; the `rem` row re-uses the last digit
#   * square / mul_no_carry: every (x_j, y_{i-j}) operand cell copies the operand's limb (or the zero cell for extended limbs)
#   * refresh: div_mod_unsafe's dividend copies the limb it splits; carried remainders copy theirs
#   * mul_mod: operands a / b copy the limbs of the value they are (previous remainders, g / r limbs extended with the zero
#     cell: extend_limbs, paillier.rs:49,53; the constant 1); the re-assigned n copies the refreshed n^2 limbs (paillier.rs:45);
#     qn + r and r < n re-use the q, n, r cells
#   * assert_equal_fresh (bench.rs:74): x / y of every is_equal copy the circuit's result limbs / the assigned res limbs
# Dependency-defined wiring [D] like the cell patterns themselves: what is checked is that the GPU-written stream has equal
# values wherever this restatement's circuit would need them equal.
# ----------------------------------------------------------------------------------------
def expand_circuit_cells_wired(kind: str, n: int, g: int, x: int, y: int, res: int, enc_bits: int, limb_bits: int, lb: int,
                               full: bool = False):
    """-> (advice values, [(src index, copy index)], satisfied bit); full=True: additionally the CONSTANT cells [(index, value)]
    (halo2-lib assigns a constant as an advice cell tied to a cell of a fixed column: assign_raw_constants) and, for every cell of
    the lookup stream in order, the index of the advice cell it copies (RangeChip pushes the cell to cells_to_lookup; the
    lookup-advice column gets a copy tied by an equality constraint)"""
    Ln = enc_bits // limb_bits
    L = 2 * Ln
    base = 1 << limb_bits
    adv: List[int] = []
    pairs: List[Tuple[int, int]] = []
    consts: List[Tuple[int, int]] = []
    lk_src: List[int] = []

    def putc(v):
        """a constant cell"""
        adv.append(v % FR_R)
        consts.append((len(adv) - 1, v % FR_R))
        return len(adv) - 1

    def put(v, src=None):
        adv.append(v % FR_R)
        if src is not None:
            pairs.append((src, len(adv) - 1))
        return len(adv) - 1

    def range_check(xv, xcell, bits):
        """-> the cell that holds the checked value afterwards (the recomposed accumulator; xcell itself for one digit)"""
        k = -(-bits // lb)
        rem = bits % lb
        mask = (1 << lb) - 1
        digs = [(xv >> (lb * i)) & mask for i in range(k)]
        last_cell = xcell          # k == 1: the checked cell itself is looked up
        holder = xcell
        if k > 1:
            c0 = put(digs[0])
            lk_src.append(c0)
            acc = digs[0]
            acc_cell = c0
            for gi in range(1, k):
                acc += digs[gi] << (lb * gi)
                last_cell = put(digs[gi])
                lk_src.append(last_cell)
                putc(1 << (lb * gi))
                acc_cell = put(acc)
            if xcell is not None:
                pairs.append((xcell, acc_cell))
            holder = acc_cell
        else:
            assert xcell is not None
            lk_src.append(xcell)
        last = digs[-1]
        if rem == 1:
            putc(0); put(last, last_cell); put(last, last_cell); put(last, last_cell)
        elif rem > 1:
            putc(0); put(last, last_cell); putc(1 << (lb - rem)); lk_src.append(put(last << (lb - rem)))
        return holder

    def assign(v, nl):
        limbs = decompose_biguint(v, nl, limb_bits)
        cells = [put(l) for l in limbs]
        for l, c in zip(limbs, cells):
            range_check(l, c, limb_bits)
        return list(zip(limbs, cells))

    def mul_cells(xs, ys, D):
        """xs / ys: [(value, cell)] ; shorter operands are extended with the load_zero cell this call pushes first"""
        zc = putc(0)
        xe = list(xs) + [(0, zc)] * (D - len(xs))
        ye = list(ys) + [(0, zc)] * (D - len(ys))
        prod = []
        for i in range(D):
            putc(0)
            s = 0
            cell = None
            for j in range(i + 1):
                s += xe[j][0] * ye[i - j][0]
                put(xe[j][0], xe[j][1])
                put(ye[i - j][0], ye[i - j][1])
                cell = put(s)
            prod.append((s, cell))
        return prod

    def is_equal(xv, xc, yv, yc):
        d = (xv - yv) % FR_R
        c_d = put(d); put(yv, yc); putc(1); put(xv, xc)
        z, bit = _is_zero_cells(d)
        # is_zero(a): [is_zero, a, inv, 1 | 0, a, is_zero, 0]: is_zero + a * inv = 1 and 0 + a * is_zero = 0
        c_z = put(z[0]); c_a = put(z[1], c_d); put(z[2]); putc(z[3]); putc(z[4]); put(z[5], c_a); put(z[6], c_z); putc(z[7])
        return bit

    def div_mod(v, vcell):
        qd, rd = v >> limb_bits, v & (base - 1)
        prod = qd * base
        c_qd = put(qd); c_rd = put(rd); putc(0); put(qd, c_qd); putc(base); c_pr = put(prod)
        put(v - prod); put(prod, c_pr); putc(1); put(v, vcell)
        is_equal(rd, c_rd, v - prod, None)
        return (qd, c_qd), (rd, c_rd)

    def mul_mod(a, b, nfresh):
        """a, b: [(limb, cell)] x L; nfresh: the refreshed n^2 limbs [(limb, cell)] -> r as [(limb, cell)]"""
        av, bv = get_biguint([t[0] for t in a], limb_bits), get_biguint([t[0] for t in b], limb_bits)
        n2 = get_biguint([t[0] for t in nfresh], limb_bits)
        q, r = divmod(av * bv, n2)
        ql, nl_, rl = assign(q, L), assign(n2, L), assign(r, L)
        for (v, c), (_, src) in zip(nl_, nfresh):
            pairs.append((src, c))
        D = 2 * L - 1
        p_ab = mul_cells(a, b, D)
        p_qn = mul_cells(ql, nl_, D)
        qnr = list(p_qn)
        for i in range(L):
            put(p_qn[i][0], p_qn[i][1]); putc(1); put(rl[i][0], rl[i][1])
            qnr[i] = (p_qn[i][0] + rl[i][0], put(p_qn[i][0] + rl[i][0]))
        m = base - 1
        MAX = L * m * m + m
        cb = (2 * MAX).bit_length() - limb_bits
        c_zero = putc(0); c_one = putc(1)
        carry, accx, eq_bit = (0, c_zero), (0, c_zero), 1
        eq_cell = c_one
        for i in range(D):
            diff = p_ab[i][0] - qnr[i][0]
            c_diff = put(diff); put(qnr[i][0], qnr[i][1]); putc(1); put(p_ab[i][0], p_ab[i][1])
            s = diff + carry[0] + MAX
            put(diff, c_diff); put(carry[0], carry[1]); putc(1); put(diff + carry[0]); putc(MAX); putc(1); c_s = put(s)
            new_carry, cmod = div_mod(s, c_s)
            t = accx[0] + MAX
            put(accx[0], accx[1]); putc(1); putc(MAX); c_t = put(t)
            q_acc, mod_acc = div_mod(t, c_t)
            e = is_equal(cmod[0], cmod[1], mod_acc[0], mod_acc[1])
            putc(0); put(eq_bit, eq_cell); put(e); eq_cell = put(eq_bit & e)
            eq_bit &= e
            accx = q_acc
            if i < D - 1:
                range_check(new_carry[0], new_carry[1], cb)
            else:
                e = is_equal(new_carry[0], new_carry[1], accx[0], accx[1])
                putc(0); put(eq_bit, eq_cell); put(e); eq_cell = put(eq_bit & e)
                eq_bit &= e
            carry = new_carry
        borrow = (0, c_zero)
        for i in range(L):
            nb = nl_[i][0] + borrow[0]
            put(nl_[i][0], nl_[i][1]); putc(1); put(borrow[0], borrow[1]); put(nb)
            shift = rl[i][0] - nb + base
            lt = 1 if rl[i][0] < nb else 0
            out = shift & (base - 1)
            put(shift); c_lt = put(lt); c_out = put(out)
            put(rl[i][0], rl[i][1]); put(lt, c_lt); putc(base); put(rl[i][0] + lt * base)
            range_check(out, c_out, limb_bits)
            borrow = (lt, c_lt)
        put(borrow[0], borrow[1])
        return rl

    n_c, g_c, x_c, y_c = assign(n, Ln), assign(g, Ln), assign(x, Ln), assign(y, Ln)
    prod = mul_cells(n_c, n_c, 2 * Ln - 1)
    # refresh
    inc = refresh_aux(limb_bits, Ln, Ln)
    putc(0)
    cur = list(prod) + [(0, None)] * (len(inc) - len(prod))
    for i in range(len(inc)):
        limb = cur[i]
        for j in range(inc[i] + 1):
            qd, rd = div_mod(limb[0], limb[1])
            if j == 0:
                cur[i] = rd
            else:
                tgt = cur[i + j]
                put(tgt[0], tgt[1]); putc(1); put(rd[0], rd[1])
                cur[i + j] = (tgt[0] + rd[0], put(tgt[0] + rd[0]))
            limb = qd
    # the refreshed limbs: a limb no cell was written for is the constant zero, its range check stands alone
    fresh = []
    for v, c in cur:
        holder = range_check(v, c, limb_bits)
        fresh.append((v, c if c is not None else holder))
    zero = putc(0)                     # ctx.load_zero() (paillier.rs:47 / 77)
    ext = lambda limbs: list(limbs) + [(0, zero)] * (L - len(limbs))
    if kind == "encrypt":
        # pow_mod_fixed_exp: assign_constant(1), load_zero, then the steps in pow_mod_fixed_exp_trace's order; a step's
        # operands are the limbs of earlier remainders (or of the base / the constant 1)
        def pow_mod_traced(base_limbs, e):
            one = putc(1)
            z2 = putc(0)
            acc = [(1, one)] + [(0, z2)] * (L - 1)
            _, steps = pow_mod_fixed_exp_trace(get_biguint([t[0] for t in base_limbs], limb_bits), e, n * n)
            vals = {get_biguint([t[0] for t in base_limbs], limb_bits): base_limbs, 1: acc}
            last = acc
            for (a_, b_, q_, r_) in steps:
                ra = mul_mod(vals[a_], vals[b_], fresh)
                vals[r_] = ra
                last = ra
            return vals, steps
        vals_g, steps_g = pow_mod_traced(ext(g_c), x)
        gm_val = pow(g, x, n * n)
        gm = vals_g[gm_val] if steps_g else vals_g[1]
        vals_r, steps_r = pow_mod_traced(ext(y_c), n)
        rn_val = pow(y, n, n * n)
        rn = vals_r[rn_val]
        c_limbs = mul_mod(gm, rn, fresh)
    elif kind == "encrypt_uniform":
        # the uniform-shape circuit (expand_uniform_circuit_cells): g^m through pow_mod over the message's BITS in circuit --
        # num_to_bits of every limb of m (the recomposition copies the limb, every bit is asserted boolean), then per bit
        # mul_mod(acc, sq), a limb-wise select(bit, product, acc) and square_mod(sq); r^n and the rest as in 'encrypt'
        W = limb_bits
        one = putc(1)
        z2 = putc(0)
        acc = [(1, one)] + [(0, z2)] * (L - 1)
        sq = ext(g_c)
        for li in range(Ln):
            xv, xc = x_c[li]
            bits = [(xv >> i) & 1 for i in range(W)]
            bit_cells = [put(bits[0])]
            accv = bits[0]
            acc_cell = bit_cells[0]
            for i in range(1, W):
                accv += bits[i] << i
                bit_cells.append(put(bits[i]))
                putc(1 << i)
                acc_cell = put(accv)
            pairs.append((xc, acc_cell))
            for b, bc in zip(bits, bit_cells):
                putc(0); put(b, bc); put(b, bc); put(b, bc)
            for bi in range(W):
                mul = mul_mod(acc, sq, fresh)
                new_acc = []
                for t in range(L):
                    (av, ac), (bv, bc_) = mul[t], acc[t]
                    d = av - bv
                    c_d = put(d); putc(1); put(bv, bc_); put(av, ac); put(bv, bc_); put(bits[bi], bit_cells[bi]); put(d, c_d)
                    new_acc.append((d * bits[bi] + bv, put(d * bits[bi] + bv)))
                acc = new_acc
                sq = mul_mod(sq, sq, fresh)
        gm = acc

        def pow_mod_traced_r(base_limbs, e):
            one_ = putc(1)
            z2_ = putc(0)
            acc_ = [(1, one_)] + [(0, z2_)] * (L - 1)
            sq_ = base_limbs
            for bit in exp_bits_lsb_first(e):
                cur = sq_
                sq_ = mul_mod(cur, cur, fresh)
                if bit:
                    acc_ = mul_mod(acc_, cur, fresh)
            return acc_
        rn = pow_mod_traced_r(ext(y_c), n)
        c_limbs = mul_mod(gm, rn, fresh)
    else:
        c_limbs = mul_mod(ext(x_c), ext(y_c), fresh)
    res_c = assign(res, L)
    putc(0); eq_cell = putc(1)
    eq = 1
    for (cv, cc), (rv, rc) in zip(c_limbs, res_c):
        e = is_equal(cv, cc, rv, rc)
        putc(0); put(eq, eq_cell); put(e); eq_cell = put(eq & e)
        eq &= e
    if full:
        return {"advice": adv, "pairs": pairs, "consts": consts, "lookup_src": lk_src, "satisfied": eq, "result_cell": eq_cell}
    return adv, pairs, eq


# ----------------------------------------------------------------------------------------
# SHPLONK multi-point opening (halo2 multiopen::shplonk::prover).  Restated from the published
# protocol as halo2 implements it (dependency, tag [D]); pinned by the opening identity the two output polynomials satisfy
# (tests/test_oracle.py::test_shplonk_identity).  sets: list of (polys (coefficient lists), point indices); points: the union.
# ----------------------------------------------------------------------------------------
def _poly_mul_linear(p, x):   # p(X) * (X - x)
    out = [0] * (len(p) + 1)
    for i, c in enumerate(p):
        out[i + 1] = (out[i + 1] + c) % FR_R
        out[i] = (out[i] - c * x) % FR_R
    return out


def interpolate(xs: Sequence[int], ys: Sequence[int]) -> List[int]:
    r = [0] * len(xs)
    for t, (xt, yt) in enumerate(zip(xs, ys)):
        b, den = [1], 1
        for s_, xs_ in enumerate(xs):
            if s_ != t:
                b = _poly_mul_linear(b, xs_)
                den = den * (xt - xs_) % FR_R
        c = yt * pow(den, -1, FR_R) % FR_R
        for k, bk in enumerate(b):
            r[k] = (r[k] + bk * c) % FR_R
    return r


def shplonk_h(sets, points: Sequence[int], y: int, v: int, n: int):
    """-> (h coefficients (n), folded polynomials C_k, interpolations R_k)"""
    h = [0] * n
    Cs, Rs = [], []
    for k, (polys, idx) in enumerate(sets):
        C = [sum(pow(y, j, FR_R) * p[i] for j, p in enumerate(polys)) % FR_R for i in range(n)]
        xs = [points[i] for i in idx]
        R = interpolate(xs, [poly_eval(C, x) for x in xs])
        N = list(C)
        for i, c in enumerate(R):
            N[i] = (N[i] - c) % FR_R
        for x in xs:
            assert poly_eval(N, x) == 0
            N = kate_division(N, x)
        vk = pow(v, k, FR_R)
        h = [(a + vk * b) % FR_R for a, b in zip(h, N)]
        Cs.append(C)
        Rs.append(R)
    return h, Cs, Rs


def shplonk_h2(sets, points: Sequence[int], y: int, v: int, u: int, n: int):
    h, Cs, Rs = shplonk_h(sets, points, y, v, n)
    zt = 1
    for t in points:
        zt = zt * (u - t) % FR_R
    L = [(-zt * c) % FR_R for c in h]
    z0 = None
    for k, ((polys, idx), C, R) in enumerate(zip(sets, Cs, Rs)):
        zk = 1
        for t, x in enumerate(points):
            if t not in idx:
                zk = zk * (u - x) % FR_R
        if k == 0:
            z0 = zk
        ck = pow(v, k, FR_R) * zk % FR_R
        L = [(a + ck * b) % FR_R for a, b in zip(L, C)]
        L[0] = (L[0] - ck * poly_eval(R, u)) % FR_R
    assert poly_eval(L, u) == 0
    q = kate_division(L, u)
    z0i = pow(z0, -1, FR_R)
    return h, [c * z0i % FR_R for c in q], z0


# ----------------------------------------------------------------------------------------
# the uniform-shape encrypt circuit.  g^m through BigUintChip::pow_mod (exponent bits IN the
# circuit, halo2-rsa lineage [D]): for each of exactly m_bits bits, LSB first:  muled = mul_mod(acc, squared);
# acc = select(bit, muled, acc); squared = square_mod(squared).  The step count no longer depends on the message.
# ----------------------------------------------------------------------------------------
def pow_mod_uniform_trace(a: int, e: int, m_bits: int, modulus: int) -> Tuple[int, List[Step]]:
    assert e >> m_bits == 0, "exponent does not fit m_bits"
    steps: List[Step] = []
    acc, squared = 1, a
    for i in range(m_bits):
        st = mul_mod_step(acc, squared, modulus)
        steps.append(st)
        if (e >> i) & 1:
            acc = st[3]
        st = mul_mod_step(squared, squared, modulus)
        steps.append(st)
        squared = st[3]
    return acc, steps


def encrypt_uniform_trace(n: int, g: int, m: int, r: int, m_bits: int):
    n2 = n * n
    gm, steps_g = pow_mod_uniform_trace(g, m, m_bits, n2)
    rn, steps_r = pow_mod_fixed_exp_trace(r, n, n2)
    final = mul_mod_step(gm, rn, n2)
    return final[3], steps_g, steps_r, final


def _num_to_bits_cells(x: int, nbits: int):
    """FlexGate::num_to_bits(x, nbits): the inner product of the bits with the powers of two (1 + 3 (nbits - 1) cells, the
    bits are its a-operands), then assert_bit of every bit ([0, b, b, b])"""
    bits = [(x >> i) & 1 for i in range(nbits)]
    adv = [bits[0]]
    acc = bits[0]
    for i in range(1, nbits):
        acc += bits[i] << i
        adv += [bits[i], 1 << i, acc]
    for b in bits:
        adv += [0, b, b, b]
    return adv, bits


def _select_cells(a: int, b: int, sel: int):
    """FlexGate::select(a, b, sel) = sel ? a : b : | a - b | 1 | b | a | b | sel | a - b | out |"""
    d = a - b
    return [d, 1, b, a, b, sel, d, d * sel + b]


def expand_uniform_circuit_cells(n: int, g: int, m: int, r: int, res: int, enc_bits: int, limb_bits: int, lb: int):
    """the uniform-shape encrypt circuit: as expand_circuit_cells('encrypt'), with g^m through pow_mod:
    assign_constant(1), load_zero, then per limb of m: num_to_bits, and per bit: mul_mod(acc, sq), select limb by limb,
    square_mod(sq)"""
    Ln = enc_bits // limb_bits
    L = 2 * Ln
    adv, lk, seg = [], [], {}

    def put(name, a, l=()):
        seg[name] = (len(adv), len(lk))
        adv.extend(a)
        lk.extend(l)

    for name, v in (("assign_n", n), ("assign_g", g), ("assign_x", m), ("assign_y", r)):
        put(name, *expand_assign_cells(v, Ln, limb_bits, lb))
    nl = decompose_biguint(n, Ln, limb_bits)
    sq_cells, prod = _mul_cells(nl, nl, 2 * Ln - 1)
    put("square", sq_cells)
    r_adv, r_lk, fresh = expand_refresh_cells(prod, refresh_aux(limb_bits, Ln, Ln), limb_bits, lb)
    put("refresh", r_adv, r_lk)
    put("load_zero", [0])
    n2 = n * n
    c, sg, sr, fin = encrypt_uniform_trace(n, g, m, r, enc_bits)
    put("pow_g", [1, 0])
    ml = decompose_biguint(m, Ln, limb_bits)
    for li in range(Ln):
        nb_cells, bits = _num_to_bits_cells(ml[li], limb_bits)
        adv.extend(nb_cells)
        for bi in range(limb_bits):
            i = li * limb_bits + bi
            st_mul, st_sq = sg[2 * i], sg[2 * i + 1]
            a_, l_ = expand_mul_mod_cells(*st_mul, n2, L, lb, limb_bits)
            adv.extend(a_)
            lk.extend(l_)
            acc_l, mul_l = decompose_biguint(st_mul[0], L, limb_bits), decompose_biguint(st_mul[3], L, limb_bits)
            for t in range(L):
                adv.extend(_select_cells(mul_l[t], acc_l[t], bits[bi]))
            a_, l_ = expand_mul_mod_cells(*st_sq, n2, L, lb, limb_bits)
            adv.extend(a_)
            lk.extend(l_)
    put("pow_r", [1, 0])
    for st in sr:
        a_, l_ = expand_mul_mod_cells(*st, n2, L, lb, limb_bits)
        adv.extend(a_)
        lk.extend(l_)
    put("final", *expand_mul_mod_cells(*fin, n2, L, lb, limb_bits))
    put("assign_res", *expand_assign_cells(res, L, limb_bits, lb))
    ae, bit = expand_assert_equal_fresh_cells(decompose_biguint(c, L, limb_bits), decompose_biguint(res, L, limb_bits))
    put("assert_equal", ae)
    seg["end"] = (len(adv), len(lk))
    seg["satisfied"] = bool(bit)
    return [v % FR_R for v in adv], [v % FR_R for v in lk], seg


def uniform_circuit_cells_windows(n: int, g: int, m: int, r: int, res: int, enc_bits: int, limb_bits: int, lb: int, adv_windows,
                                  lk_windows=()):
    """encrypt_circuit_cells_windows for the uniform-shape circuit (expand_uniform_circuit_cells' streams)"""
    Ln = enc_bits // limb_bits
    L = 2 * Ln
    n2 = n * n
    pre_a, pre_l = [], []
    for v in (n, g, m, r):
        a_, l_ = expand_assign_cells(v, Ln, limb_bits, lb)
        pre_a += a_
        pre_l += l_
    nl = decompose_biguint(n, Ln, limb_bits)
    sq_cells, prod = _mul_cells(nl, nl, 2 * Ln - 1)
    pre_a += sq_cells
    r_adv, r_lk, _ = expand_refresh_cells(prod, refresh_aux(limb_bits, Ln, Ln), limb_bits, lb)
    pre_a += r_adv + [0]
    pre_l += r_lk
    c, sg, sr, fin = encrypt_uniform_trace(n, g, m, r, enc_bits)
    fa, fl = expand_mul_mod_cells(*fin, n2, L, lb, limb_bits)
    cps, lps = len(fa), len(fl)
    ta, tl = expand_assign_cells(res, L, limb_bits, lb)
    ae, _ = expand_assert_equal_fresh_cells(decompose_biguint(c, L, limb_bits), decompose_biguint(res, L, limb_bits))
    step = lambda st: (cps, lps, (lambda st=st: expand_mul_mod_cells(*st, n2, L, lb, limb_bits)))
    pieces = [(len(pre_a), len(pre_l), lambda: (pre_a, pre_l)), (2, 0, lambda: ([1, 0], []))]
    ml = decompose_biguint(m, Ln, limb_bits)
    for li in range(Ln):
        nb_cells, bits = _num_to_bits_cells(ml[li], limb_bits)
        pieces.append((len(nb_cells), 0, (lambda c_=nb_cells: (c_, []))))
        for bi in range(limb_bits):
            i = li * limb_bits + bi
            st_mul, st_sq = sg[2 * i], sg[2 * i + 1]
            pieces.append(step(st_mul))

            def sel(st_mul=st_mul, b=bits[bi]):
                acc_l, mul_l = decompose_biguint(st_mul[0], L, limb_bits), decompose_biguint(st_mul[3], L, limb_bits)
                out = []
                for t in range(L):
                    out.extend(_select_cells(mul_l[t], acc_l[t], b))
                return out, []

            pieces.append((8 * L, 0, sel))
            pieces.append(step(st_sq))
    pieces.append((2, 0, lambda: ([1, 0], [])))
    pieces += [step(st) for st in sr]
    pieces.append((cps, lps, lambda: (fa, fl)))
    pieces.append((len(ta), len(tl), lambda: (ta, tl)))
    pieces.append((len(ae), 0, lambda: (ae, [])))
    return _cut_windows(pieces, adv_windows, lk_windows)


def gate_offsets_uniform_circuit(enc_bits: int, limb_bits: int, lb: int, n_steps_r: int):
    Ln = enc_bits // limb_bits
    L = 2 * Ln
    gates, off = [], 0

    def add(part):
        nonlocal off
        g_, n_ = part
        gates.extend(off + o for o in g_)
        off += n_

    for _ in range(4):
        add(gate_offsets_assign(Ln, limb_bits, lb))
    add(gate_offsets_square(Ln))
    add(gate_offsets_refresh(refresh_aux(limb_bits, Ln, Ln), limb_bits, lb))
    off += 1
    mm = gate_offsets_mul_mod(L, lb, limb_bits)
    off += 2
    W = limb_bits
    for _ in range(Ln):
        add(([3 * i for i in range(W - 1)] + [1 + 3 * (W - 1) + 4 * i for i in range(W)], 7 * W - 2))
        for _ in range(W):
            add(mm)
            add(([8 * t + o for t in range(L) for o in (0, 4)], 8 * L))
            add(mm)
    off += 2
    for _ in range(n_steps_r):
        add(mm)
    add(mm)
    add(gate_offsets_assign(L, limb_bits, lb))
    add(gate_offsets_assert_equal(L))
    return gates, off
