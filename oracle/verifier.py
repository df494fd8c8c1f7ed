"""TEST INFRASTRUCTURE (oracle): what halo2's VERIFIER computes from a proof's evaluations and commitments, in plain Python
integers -- the check the connected-proof tests apply to the device prover's output (paillier_halo2_amd/prover.py).

Restated from the published PLONK / halo2 protocol as halo2-axiom implements it (plonk/verifier.rs, permutation/verifier.rs,
lookup/verifier.rs, vanishing/verifier.rs, multiopen/shplonk/verifier.rs; dependency of the reference, SURVEY tag [D], reached from
/root/reference/src/bench.rs:164-171 `check_proof`).  Nothing here is shared with the prover's code path: the prover evaluates the
constraint lines on the extended coset, this file evaluates them at ONE point from the opened values only.

  expected_h(..)   : the verifier's expected_h_eval = (sum of all constraint lines folded by y) / (x^n - 1)
  replay_challenges(..): the verifier's side of Fiat-Shamir -- every challenge re-derived from the proof's own commitments and evaluations
                     (halo2's Blake2b transcript in its primitives; the drivers' HashTranscript / host/transcript.hpp are the prover's side)
  shplonk_check(..): the multi-point opening's final identity, in the exponent, against the proof's ACTUAL commitments (the tests
                     know the toxic scalar s of their SRS, so e(., [s - u]_2) becomes a scalar multiplication)
"""
from __future__ import annotations

from typing import Dict, List, Sequence

from . import pyref as P

R = P.FR_R


def lagrange_at(k: int, bf: int, x: int):
    """l_0(x), l_last(x), l_blind(x) = sum of the blinding rows' Lagrange polynomials (EvaluationDomain::l_i_range)"""
    n = 1 << k
    w = P.fr_omega(k)
    xn = pow(x, n, R)
    li = lambda i: (xn - 1) * pow(w, i % n, R) % R * pow(n * (x - pow(w, i % n, R)) % R, -1, R) % R
    u = n - (bf + 1)
    return li(0), li(u), sum(li(i) for i in range(u + 1, n)) % R


def expected_h(k: int, bf: int, A: int, Lk: int, chunk: int, ev: Dict[str, List[List[int]]], beta: int, gamma: int, y: int, x: int,
               delta: int) -> int:
    """ev (canonical integers): advice [A][4] at x, wx, w^2x, w^3x; lookup_advice [Lk][1]; fixed [A + 2][1] = selectors, constants,
    table; sigma [m][1]; perm_z [S][3] at x, wx, w^-(bf+1) x; lookup_z [Lk][2] at x, wx; perm_inputs [Lk][2] at x, w^-1 x;
    perm_tables [Lk][1]"""
    n = 1 << k
    l0, llast, lblind = lagrange_at(k, bf, x)
    lact = (1 - llast - lblind) % R
    acc = 0

    def line(v):
        nonlocal acc
        acc = (acc * y + v) % R

    # custom gates, one per basic-gate column: q (a + b c - d) over four consecutive rows
    for j in range(A):
        a0, a1, a2, a3 = ev["advice"][j]
        line(ev["fixed"][j][0] * (a0 + a1 * a2 - a3))
    # permutation argument over [advice | lookup advice | constants]
    vals = [ev["advice"][j][0] for j in range(A)] + [ev["lookup_advice"][j][0] for j in range(Lk)] + [ev["fixed"][A][0]]
    m = len(vals)
    S = -(-m // chunk)
    z = ev["perm_z"]
    line(l0 * (1 - z[0][0]))
    line(llast * (z[S - 1][0] * z[S - 1][0] - z[S - 1][0]))
    for j in range(1, S):
        line(l0 * (z[j][0] - z[j - 1][2]))
    cur = beta * x % R                                   # beta * delta^c * x
    for j in range(S):
        left, right = z[j][1], z[j][0]
        for c in range(j * chunk, min(m, (j + 1) * chunk)):
            left = left * (vals[c] + beta * ev["sigma"][c][0] + gamma) % R
            right = right * (vals[c] + cur + gamma) % R
            cur = cur * delta % R
        line(lact * (left - right))
    # lookups: input = the lookup-advice column, table = the table column
    tab = ev["fixed"][A + 1][0]
    for j in range(Lk):
        a = ev["lookup_advice"][j][0]
        zx, zwx = ev["lookup_z"][j]
        ap, ap_prev = ev["perm_inputs"][j]
        sp = ev["perm_tables"][j][0]
        line(l0 * (1 - zx))
        line(llast * (zx * zx - zx))
        line(lact * (zwx * (ap + beta) % R * (sp + gamma) - zx * (a + beta) % R * (tab + gamma)))
        line(l0 * (ap - sp))
        line(lact * (ap - sp) % R * (ap - ap_prev))
    xn = pow(x, n, R)
    return acc * pow(xn - 1, -1, R) % R


def shplonk_check(cref, layout, points: Sequence[int], commitments: Dict[str, Sequence], ev: Dict[str, List[List[int]]], y: int, v: int,
                  u: int, w1, w2, s_tox: int) -> bool:
    """layout: [(point indices, [(family, index)])] (the prover's query order); commitments[family][index] / w1 / w2: affine points as
    the C ABI stores them (8 Montgomery words); ev as above (+ "h", "random").  Checks, with [.] = . G,
        sum_k v^k z_k ( sum_j y^j C_kj - [R_k(u)] ) - Z_T(u) W1  ==  z_0 (s - u) W2,       z_k = Z_{T \\ S_k}(u)
    by ONE multi-scalar multiplication over the commitments (oracle/pz_oracle.c's best_multiexp restatement) -- the pairing check
    e(lhs, [1]_2) = e(W2, [z_0 (s - u)]_2) of halo2's verifier with the G2 side collapsed by the known scalar."""
    import numpy as np

    zt = 1
    for t in points:
        zt = zt * (u - t) % R
    scalars: List[int] = []
    bases = []
    const = 0                                              # the coefficient of G: - sum_k v^k z_k R_k(u)
    z0 = None
    for kk, (idx, members) in enumerate(layout):
        zk = 1
        for t, pt in enumerate(points):
            if t not in idx:
                zk = zk * (u - pt) % R
        if kk == 0:
            z0 = zk
        xs = [points[i] for i in idx]
        folded = [sum(pow(y, j, R) * ev[f][i][q] for j, (f, i) in enumerate(members)) % R for q in range(len(xs))]
        Rk_u = P.poly_eval(P.interpolate(xs, folded), u)
        vk = pow(v, kk, R) * zk % R
        const = (const - vk * Rk_u) % R
        for j, (f, i) in enumerate(members):
            scalars.append(vk * pow(y, j, R) % R)
            bases.append(np.asarray(commitments[f][i], dtype=np.uint64))
    scalars.append((-zt) % R)
    bases.append(np.asarray(w1, dtype=np.uint64).reshape(8))
    scalars.append((-(z0 * (s_tox - u))) % R)
    bases.append(np.asarray(w2, dtype=np.uint64).reshape(8))
    scalars.append(const)
    bases.append(cref.affine_ints_to_mont([P.G1_GEN])[0])
    acc = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont(scalars), np.stack(bases)))
    return not np.asarray(acc).any()                       # the identity: (0, 0) in the ABI's affine form


COMMITMENT_ROUNDS = ((("advice", "lookup_advice"), ("theta",)), (("perm_inputs", "perm_tables"), ("beta", "gamma")),
                     (("perm_z", "lookup_z", "random"), ("y",)), (("h",), ("x",)))
EVAL_FAMILIES = ("advice", "lookup_advice", "fixed", "sigma", "perm_z", "lookup_z", "perm_inputs", "perm_tables", "random")


def replay_challenges(seed: bytes, commitments: Dict[str, Sequence], evals: Dict[str, Sequence]) -> Dict[str, int]:
    """what a verifier does before any arithmetic: read the proof in the order the prover wrote it and draw each challenge from
    everything read so far.  Transcript = halo2's `Blake2bRead` [D] (halo2_proofs transcript/blake2b.rs) in its primitives: BLAKE2b-512,
    personalisation "Halo2-Transcript", a domain byte per item (1 point, 2 scalar, 0 challenge), challenge = the digest of a clone of the
    state as a little-endian 512-bit integer mod r (Fr::from_uniform_bytes).  Field elements enter as their 4 Montgomery words (the
    form they have in a proof of this repo), a point as x then y; `seed` stands where halo2 absorbs the verifying key's digest.

    commitments: family -> rows of 8 words (affine, Montgomery); evals: family -> rows of 4 * points words (Montgomery; the
    "lookup_advice" family is followed by the "constants" row if that is held separately).  -> {challenge name: integer}"""
    import hashlib
    import struct

    h = hashlib.blake2b(bytes(seed), digest_size=64, person=b"Halo2-Transcript")
    out: Dict[str, int] = {}

    def words(row):
        return [int(w) for w in (row.reshape(-1) if hasattr(row, "reshape") else row)]

    def point(row):
        w = words(row)
        assert len(w) == 8
        h.update(b"\x01" + struct.pack("<8Q", *w))

    def scalars(row):
        w = words(row)
        assert len(w) % 4 == 0
        for i in range(0, len(w), 4):
            h.update(b"\x02" + struct.pack("<4Q", *w[i:i + 4]))

    def draw(name):
        h.update(b"\x00")
        out[name] = int.from_bytes(h.copy().digest(), "little") % R

    for fams, names in COMMITMENT_ROUNDS:
        for f in fams:
            for row in commitments[f]:
                point(row)
        for nm in names:
            draw(nm)
    for f in EVAL_FAMILIES:
        for row in evals[f]:
            scalars(row)
        if f == "lookup_advice" and "constants" in evals:
            for row in evals["constants"]:
                scalars(row)
    draw("sh_y")
    draw("sh_v")
    for row in commitments["w1"]:
        point(row)
    draw("sh_u")
    return out
