"""Pins the oracle: constants, Python-int restatement vs C restatement vs closed forms.
The reference holds no golden vectors, so this three-way agreement plus the
committed fixtures in tests/golden/ is what the oracle is anchored to ("parity unpinned" at
proof-byte level, mathematically pinned at kernel level)."""
import random

import numpy as np
import pytest

from oracle import pyref as P


def test_constants():
    assert P.FQ_P.bit_length() == 254 and P.FR_R.bit_length() == 254
    assert P.FQ_P % 4 == 3
    assert (P.FR_R - 1) % (1 << 28) == 0 and ((P.FR_R - 1) >> 28) % 2 == 1
    assert P.FR_ROOT_OF_UNITY == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    assert pow(P.FR_ROOT_OF_UNITY, 1 << 28, P.FR_R) == 1
    assert pow(P.FR_ROOT_OF_UNITY, 1 << 27, P.FR_R) != 1
    # Montgomery constants quoted in SURVEY.md section 8c
    assert (-pow(P.FQ_P, -1, 1 << 64)) % (1 << 64) == 0x87D20782E4866389
    assert (-pow(P.FR_R, -1, 1 << 64)) % (1 << 64) == 0xC2E1F593EFFFFFFF
    assert (1 << 256) % P.FQ_P == 0x0E0A77C19A07DF2F666EA36F7879462C0A78EB28F5C70B3DD35D438DC58F0D9D
    assert (1 << 256) % P.FR_R == 0x0E0A77C19A07DF2F666EA36F7879462E36FC76959F60CD29AC96341C4FFFFFFB
    assert P.g1_is_on_curve(P.G1_GEN)
    assert P.g1_mul(P.G1_GEN, P.FR_R) == P.AFF_INF


def test_paillier_native_small():
    # the reference's own test shapes: 128-bit / 64-bit limbs and 264-bit / 88-bit limbs
    rng = random.Random(1)
    for bits in (128, 264):
        n, g, m, r = (rng.getrandbits(bits) | 1 for _ in range(4))
        c = P.paillier_enc_native(n, g, m, r)
        res, sg, sr, fin = P.encrypt_trace(n, g, m, r)
        assert res == c
        for (a, b, q, rr) in sg + sr + [fin]:
            assert a * b == q * n * n + rr and 0 <= rr < n * n
        assert len(sg) == m.bit_length() + bin(m).count("1")
        assert len(sr) == n.bit_length() + bin(n).count("1")
        c1, c2 = rng.getrandbits(bits), rng.getrandbits(bits)
        assert P.add_trace(n, c1, c2)[0] == P.paillier_add_native(n, c1, c2)


def test_get_biguint_roundtrip():
    rng = random.Random(2)
    for bits, lb in ((128, 64), (264, 88), (2048, 64)):
        x = rng.getrandbits(bits)
        limbs = P.decompose_biguint(x, bits // lb, lb)
        assert P.get_biguint(limbs, lb) == x


def test_pow_mod_edge_cases():
    assert P.pow_mod_fixed_exp_trace(5, 0, 77) == (1, [])
    acc, steps = P.pow_mod_fixed_exp_trace(5, 1, 77)
    assert acc == 5 and len(steps) == 2  # one (unused) squaring + one multiply
    acc, steps = P.pow_mod_fixed_exp_trace(0, 6, 77)
    assert acc == 0


def test_c_bigint_vs_python(cref):
    rng = random.Random(3)
    for L, bits in ((4, 256), (6, 352), (64, 4096), (96, 6144)):
        for _ in range(6):
            mod = rng.getrandbits(bits) | (1 << (bits - 1))
            a, b = rng.randrange(mod), rng.randrange(mod)
            rc, q, r = cref.mul_mod_step(L, a, b, mod)
            assert rc == 0 and (q, r) == divmod(a * b, mod)
        # short / even / tiny moduli
        for mod in (1, 2, 3, (1 << 64) - 1, 1 << 64, rng.getrandbits(bits // 2) | 1, rng.getrandbits(bits) & ~1 | 2):
            a = rng.getrandbits(min(bits // 2, max(1, mod.bit_length())))
            b = rng.getrandbits(min(bits // 2, max(1, mod.bit_length())))
            rc, q, r = cref.mul_mod_step(L, a, b, mod)
            if (a * b) // mod >= 1 << (64 * L):
                assert rc == -2
            else:
                assert rc == 0 and (q, r) == divmod(a * b, mod), (L, mod)
        assert cref.mul_mod_step(L, 1, 1, 0)[0] == -1


def test_c_pow_trace_vs_python(cref):
    rng = random.Random(4)
    for L, bits in ((4, 128), (6, 176), (64, 2048)):
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        n2 = n * n
        base = rng.randrange(n)
        e = rng.getrandbits(37 if L == 64 else bits)
        rc, res, steps = cref.pow_mod_trace(L, n2, base, e, L // 2)
        acc, psteps = P.pow_mod_fixed_exp_trace(base, e, n2)
        assert rc == 0 and res == acc and len(steps) == len(psteps)
        for st, (a, b, q, r) in zip(steps, psteps):
            assert [cref.limbs_to_int(st[i]) for i in range(4)] == [a, b, q, r]
    for bits, Ln in ((128, 2), (256, 4)):
        n, g, m, r = P.synth_paillier_inputs(bits, 99)
        assert cref.paillier_enc(Ln, n, g, m, r) == P.paillier_enc_native(n, g, m, r)


def test_c_field_vs_python(cref):
    rng = random.Random(5)
    for which, mod in (("fr", P.FR_R), ("fq", P.FQ_P)):
        xs = [0, 1, mod - 1] + [rng.randrange(mod) for _ in range(20)]
        mont = cref.to_mont(cref.ints_to_array(xs, 4), which)
        assert [cref.limbs_to_int(v) for v in mont] == [P.to_mont(x, mod) for x in xs]
        assert [cref.limbs_to_int(v) for v in cref.from_mont(mont, which)] == xs


def test_c_msm_vs_python(cref):
    rng = random.Random(6)
    for n in (0, 1, 2, 3, 5, 31, 33, 64, 200):
        bases = P.walk_bases(n, 12345, 67891) if n else []
        if n >= 3:
            bases[2] = P.AFF_INF  # identity base
        scalars = [rng.randrange(P.FR_R) for _ in range(n)]
        for i, v in enumerate([0, 1, P.FR_R - 1, rng.getrandbits(64), rng.getrandbits(140)][:n]):
            scalars[i] = v
        want = P.msm_pippenger(scalars, bases) if n > 40 else P.msm_naive(scalars, bases)
        got = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont(scalars).reshape(-1, 4),
                                            cref.affine_ints_to_mont(bases).reshape(-1, 8), threads=3))
        assert cref.affine_mont_to_ints(got)[0] == want, n


def test_walk_bases_and_dlog_identity(cref):
    s, t = 0x1234567, 0x89ABCDEF1
    n = 300
    wb = cref.walk_bases(n, s, t)
    py = P.walk_bases(8, s, t)
    assert cref.affine_mont_to_ints(wb[:8]) == py
    assert all(cref.g1_on_curve(wb[i]) for i in (0, 1, 17, n - 1))
    scalars = P.witness_like_scalars(n, 7)
    got = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont(scalars), wb))
    assert cref.affine_mont_to_ints(got)[0] == P.msm_walk_expected(scalars, s, t)
    # adversarial walk: G, 2G, 3G ... forces doubling / cancellation cases inside buckets
    wb = cref.walk_bases(64, 1, 1)
    scalars = [1] * 32 + [P.FR_R - 1] * 32
    got = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont(scalars), wb))
    assert cref.affine_mont_to_ints(got)[0] == P.msm_walk_expected(scalars, 1, 1)


def test_c_ntt_vs_python(cref):
    rng = random.Random(8)
    for log_n in range(0, 9):
        n = 1 << log_n
        a = [rng.randrange(P.FR_R) for _ in range(n)]
        omega = P.fr_omega(log_n)
        want = P.ntt(a, omega)
        if log_n <= 5:
            assert want == P.ntt_naive(a, omega)
        got = cref.ntt_fr(cref.fr_ints_to_mont(a), cref.fr_ints_to_mont([omega])[0], log_n, threads=2)
        assert cref.fr_mont_to_ints(got) == want
        assert P.intt(want, omega) == a
    # coset path: distribute_powers then NTT
    a = [rng.randrange(P.FR_R) for _ in range(16)]
    g = 7
    got = cref.fr_distribute_powers(cref.fr_ints_to_mont(a), cref.fr_ints_to_mont([g])[0])
    assert cref.fr_mont_to_ints(got) == P.coset_scale(a, g)


def test_cell_stream_satisfies_the_gate():
    """MockProver analogue (reference: base_test().expect_satisfied(true), paillier.rs:167-171): every enabled
    4-cell window of the mul_mod cell stream satisfies halo2-lib's single gate a + b*c = d; a wrong quotient
    or remainder is caught (the reference never tests expect_satisfied(false))."""
    rng = random.Random(9)
    for L, lb in ((2, 15), (4, 16), (4, 14), (4, 8), (64, 16)):
        bits = 64 * L
        n = rng.getrandbits(bits) | (1 << (bits - 1))
        a, b = rng.randrange(n), rng.randrange(n)
        q, r = divmod(a * b, n)
        adv, lk = P.expand_mul_mod_cells(a, b, q, r, n, L, lb)
        gates, end = P.gate_offsets_mul_mod(L, lb)
        assert end == len(adv)
        assert P.check_gates(adv, gates) == []
        assert max(lk) < (1 << lb)
        assert adv[-1] == 1  # r < n
        # eq bit chain ends at one: the cell before the lt segment's first gate is the final `and` output
        # negative: corrupt one cell inside a gate window
        g0 = gates[len(gates) // 2]
        bad = list(adv)
        bad[g0 + 3] = (bad[g0 + 3] + 1) % P.FR_R
        assert g0 in P.check_gates(bad, gates)
    # an inconsistent step (r off by one) breaks the carry chain: the final equality bit is 0
    L, lb = 4, 16
    n = rng.getrandbits(256) | (1 << 255)
    a, b = rng.randrange(n), rng.randrange(n)
    q, r = divmod(a * b, n)
    try:
        adv, _ = P.expand_mul_mod_cells(a, b, q, (r + 1) % n, n, L, lb)
        sc_end = P.gate_offsets_mul_mod(L, lb)[1]
        # locate the last `and` output of the eq segment: 1 cell + L*(11+rc) before the end
        from paillier_halo2_amd import layout
        off_lt = layout.mul_mod_cells(L, 64, lb).seg["lt"]
        assert adv[off_lt - 1] == 0
    except AssertionError:
        raise


# ------------------------------------------------------------------------------------------ next rows
def test_next_row_oracles_satisfy_their_defining_identities():
    """the oracle's grand-product / quotient / opening helpers are pinned by their identities (no reference fixture
    exists for them: dependency behaviour, SURVEY tag [D])"""
    rng = random.Random(77)
    R = P.FR_R
    a = [rng.randrange(R) for _ in range(33)]
    a[5] = 0
    inv = P.batch_invert(a)
    assert all((x * y % R == 1) if x else y == 0 for x, y in zip(a, inv))
    z = P.prefix_product(a, 9)
    assert z[0] == 9 and all(z[i + 1] == z[i] * a[i] % R for i in range(len(a) - 1)) and len(z) == len(a)
    # kate_division: p(t) - p(x) == (t - x) q(t)
    x, t = rng.randrange(R), rng.randrange(R)
    q = P.kate_division(a, x)
    assert q[-1] == 0
    assert (P.poly_eval(a, t) - P.poly_eval(a, x)) % R == (t - x) * P.poly_eval(q, t) % R
    # quotient of a gate-satisfying column: h (X^n - 1) == q (a0 + a1 a2 - a3)
    log_n, log_e = 3, 2
    n, N = 1 << log_n, 1 << (log_n + log_e)
    w_n, w_ext, g = P.fr_omega(log_n), P.fr_omega(log_n + log_e), P.FR_GENERATOR
    col = [rng.randrange(R) for _ in range(n)]
    col[3] = (col[0] + col[1] * col[2]) % R      # gate enabled at row 0
    sel = [1] + [0] * (n - 1)
    def extend(v):
        c = P.intt(v, w_n) + [0] * (N - n)
        return P.ntt(P.coset_scale(c, g), w_ext)
    h = P.quotient_finish(P.quotient_gate([extend(col)], [extend(sel)], 1 << log_e, 1, [0] * N), log_n, log_e, g, w_ext)
    hc = P.distribute_powers(P.intt(h, w_ext), pow(g, -1, R))
    assert not any(hc[2 * n - 2:]) and any(hc)
    cc, sc = P.intt(col, w_n), P.intt(sel, w_n)
    ev = lambda c, pt: P.poly_eval(c, pt)
    lhs = ev(sc, x) * (ev(cc, x) + ev(cc, x * w_n % R) * ev(cc, x * w_n * w_n % R) - ev(cc, x * pow(w_n, 3, R) % R)) % R
    assert lhs == ev(hc, x) * (pow(x, n, R) - 1) % R


def test_lookup_oracle_row_rule_and_telescoping():
    """permute_expression_pair: sorted input, table a permutation of the table, each row starts a run or repeats the
    previous input; the lookup product returns to z0 after the last row (what halo2's verifier enforces)"""
    rng = random.Random(78)
    R = P.FR_R
    rows, M = 300, 16
    table = [i if i < M else 0 for i in range(rows)]
    A = [rng.choice((0, 1, rng.randrange(M))) for _ in range(rows)]
    Ap, Sp = P.permute_expression_pair(A, table)
    assert Ap == sorted(A) and sorted(Sp) == sorted(table)
    assert all(Ap[i] == Sp[i] or Ap[i] == Ap[i - 1] for i in range(rows)) and Ap[0] == Sp[0]
    beta, gamma = rng.randrange(R), rng.randrange(R)
    z = P.lookup_product(A, table, Ap, Sp, beta, gamma, 1)
    i = rows - 1
    assert z[-1] * (A[i] + beta) * (table[i] + gamma) % R == (Ap[i] + beta) * (Sp[i] + gamma) % R
    with pytest.raises(ValueError):
        P.permute_expression_pair([17] + A[1:], table)


def test_c_oracle_under_address_and_ub_sanitizers():
    """the C restatement driven through its exported entry points under ASan + UBSan (CPU only: GPU sanitizers are
    not available on the pool) -- memory errors in the checker would silently weaken every parity claim"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(root, "oracle", "_asan", "selftest")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "ALL OK" in out.stdout


# ------------------------------------------------------------------------------------------------------------
# known-answer vectors from OUTSIDE this repository's own derivations (tests/golden/external_kats.json names the
# source of each): EIP-196 alt_bn128 ecAdd / ecMul precompile vectors and halo2curves' Fr constants.  Both
# restatements must reproduce them; tests/test_gpu_kernels.py::test_external_kats runs the HIP path on the same.
# ------------------------------------------------------------------------------------------------------------
def _kat_limbs(l):
    return sum(int(v, 16) << (64 * i) for i, v in enumerate(l))


def test_external_kats_ec(cref):
    from tests.util import H, load_golden

    k = load_golden("external_kats.json")
    for v in k["ecadd"]:
        a, b, c = (H(v["x1"]), H(v["y1"])), (H(v["x2"]), H(v["y2"])), (H(v["x3"]), H(v["y3"]))
        assert P.g1_is_on_curve(a) and P.g1_is_on_curve(b) and P.g1_is_on_curve(c), v["source"]
        assert P.g1_add_aff(a, b) == c, v["source"]
        am, bm = cref.affine_ints_to_mont([a])[0], cref.affine_ints_to_mont([b])[0]
        one = cref.int_to_limbs(1, 4)
        ja, jb = cref.g1_mul(am, 1), cref.g1_mul(bm, 1)
        assert cref.affine_mont_to_ints(cref.g1_normalize(cref.g1_add(ja, jb)))[0] == c, v["source"]
    for v in k["ecmul"]:
        pt, kk, c = (H(v["x"]), H(v["y"])), H(v["k"]), (H(v["x3"]), H(v["y3"]))
        assert P.g1_is_on_curve(pt) and P.g1_is_on_curve(c), v["source"]
        assert P.g1_mul(pt, kk) == c, v["source"]          # g1_mul reduces k mod r (the group order)
        got = cref.g1_normalize(cref.g1_mul(cref.affine_ints_to_mont([pt])[0], kk % P.FR_R))
        assert cref.affine_mont_to_ints(got)[0] == c, v["source"]


def test_external_kats_halo2curves_fr(cref):
    from tests.util import load_golden

    c = load_golden("external_kats.json")["halo2curves_fr"]
    root, root_inv, two_inv, delta, zeta = (_kat_limbs(c[n]) for n in ("ROOT_OF_UNITY", "ROOT_OF_UNITY_INV", "TWO_INV", "DELTA", "ZETA"))
    g = int(c["GENERATOR"])
    assert c["S"] == P.FR_S and g == P.FR_GENERATOR
    assert root == P.FR_ROOT_OF_UNITY == pow(g, (P.FR_R - 1) >> c["S"], P.FR_R)
    assert root * root_inv % P.FR_R == 1 and 2 * two_inv % P.FR_R == 1
    assert delta == pow(g, 1 << c["S"], P.FR_R) and pow(delta, (P.FR_R - 1) >> c["S"], P.FR_R) == 1
    assert zeta != 1 and pow(zeta, 3, P.FR_R) == 1 and zeta == pow(g, 2 * (P.FR_R - 1) // 3, P.FR_R)
    # the C restatement's Montgomery arithmetic on the same constants
    m = cref.fr_ints_to_mont([root, root_inv, two_inv, delta, zeta])
    mul = lambda x, y: cref.fr_scale(x.reshape(1, 4), y)[0]
    one = cref.fr_ints_to_mont([1])[0]
    assert np.array_equal(mul(m[0], m[1]), one)
    assert np.array_equal(mul(mul(m[4], m[4]), m[4]), one)
    # best_fft with omega = ROOT_OF_UNITY^(2^(S-k)): the transform of e_1 is the domain itself
    k_ = 6
    e1 = cref.fr_ints_to_mont([0, 1] + [0] * ((1 << k_) - 2))
    w = pow(root, 1 << (c["S"] - k_), P.FR_R)
    dom = cref.fr_mont_to_ints(cref.ntt_fr(e1, cref.fr_ints_to_mont([w])[0], k_))
    assert dom == [pow(w, i, P.FR_R) for i in range(1 << k_)] and dom[1 << (k_ - 1)] == P.FR_R - 1


def test_shplonk_identity():
    """the SHPLONK restatement satisfies the opening identity at a 'toxic' point s:
    sum_k v^k z_k (sum_j y^j p_kj(s) - R_k(u)) - Z_T(u) h(s) == z_0 (s - u) h'(s)"""
    rng = random.Random(88)
    n = 32
    w = P.fr_omega(5)
    x = rng.randrange(P.FR_R)
    points = [x, x * w % P.FR_R, x * pow(w, -1, P.FR_R) % P.FR_R]
    polys = [[rng.randrange(P.FR_R) for _ in range(n)] for _ in range(5)]
    sets = [(polys[0:2], [0]), (polys[2:4], [0, 1]), (polys[4:5], [0, 1, 2])]
    y, v, u, s = (rng.randrange(P.FR_R) for _ in range(4))
    h, h2, z0 = P.shplonk_h2(sets, points, y, v, u, n)
    zt = 1
    for t in points:
        zt = zt * (u - t) % P.FR_R
    lhs = (-zt * P.poly_eval(h, s)) % P.FR_R
    for k, (ps, idx) in enumerate(sets):
        zk = 1
        for t, pt in enumerate(points):
            if t not in idx:
                zk = zk * (u - pt) % P.FR_R
        C_s = sum(pow(y, j, P.FR_R) * P.poly_eval(p, s) for j, p in enumerate(ps)) % P.FR_R
        xs = [points[i] for i in idx]
        R = P.interpolate(xs, [sum(pow(y, j, P.FR_R) * P.poly_eval(p, xx) for j, p in enumerate(ps)) % P.FR_R for xx in xs])
        lhs = (lhs + pow(v, k, P.FR_R) * zk * (C_s - P.poly_eval(R, u))) % P.FR_R
    assert lhs == z0 * (s - u) * P.poly_eval(h2, s) % P.FR_R


def test_circuit_cell_windows_match_the_full_expansion():
    """encrypt_circuit_cells_windows (what the at-size GPU test of config c2 samples columns with) returns exactly the
    slices of expand_circuit_cells' streams"""
    import random

    bits, W, lb = 128, 64, 8
    rng = random.Random(5)
    n, g, m, r = rng.getrandbits(bits) | 1, rng.getrandbits(bits), rng.getrandbits(20), rng.getrandbits(bits)
    res = P.paillier_enc_native(n, g, m, r)
    adv, lk, _ = P.expand_circuit_cells("encrypt", n, g, m, r, res, bits, W, lb)
    wins = [(0, 1000), (777, 5555), (len(adv) // 2, len(adv) // 2 + 3000), (len(adv) - 900, len(adv))]
    lwins = [(0, 100), (1000, 2500), (len(lk) - 50, len(lk))]
    ta, tl, a, l = P.encrypt_circuit_cells_windows(n, g, m, r, res, bits, W, lb, wins, lwins)
    assert (ta, tl) == (len(adv), len(lk))
    assert all(c == adv[lo:hi] for (lo, hi), c in zip(wins, a))
    assert all(c == lk[lo:hi] for (lo, hi), c in zip(lwins, l))


def test_uniform_circuit_cell_windows_match_the_full_expansion():
    """uniform_circuit_cells_windows (the at-size GPU test of the uniform-shape circuit samples columns with it) returns
    exactly the slices of expand_uniform_circuit_cells' streams"""
    import random

    bits, W, lb = 128, 64, 8
    rng = random.Random(6)
    n, g, m, r = rng.getrandbits(bits) | 1, rng.getrandbits(bits), rng.getrandbits(bits), rng.getrandbits(bits)
    res = P.paillier_enc_native(n, g, m, r)
    adv, lk, _ = P.expand_uniform_circuit_cells(n, g, m, r, res, bits, W, lb)
    wins = [(0, 1000), (777, 9555), (len(adv) // 2, len(adv) // 2 + 30000), (len(adv) - 900, len(adv))]
    lwins = [(0, 100), (1000, 2500), (len(lk) - 50, len(lk))]
    ta, tl, a, l = P.uniform_circuit_cells_windows(n, g, m, r, res, bits, W, lb, wins, lwins)
    assert (ta, tl) == (len(adv), len(lk))
    assert all(c == adv[lo:hi] for (lo, hi), c in zip(wins, a))
    assert all(c == lk[lo:hi] for (lo, hi), c in zip(lwins, l))
