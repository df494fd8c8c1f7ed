"""CPU: both oracle restatements (Python ints, C) against the committed golden fixtures."""
import hashlib

import numpy as np

from oracle import pyref as P
from tests.util import H, load_golden, steps_digest_arr, steps_digest_ints


def test_paillier_golden_python():
    g = load_golden("paillier.json")
    for case in g["encrypt"]:
        n, gg, m, r = (H(case[k]) for k in ("n", "g", "m", "r"))
        assert P.paillier_enc_native(n, gg, m, r) == H(case["c"])
        c, sg, sr, fin = P.encrypt_trace(n, gg, m, r)
        L = 2 * (case["enc_bits"] // 64)
        assert (len(sg), len(sr)) == (case["n_steps_g"], case["n_steps_r"])
        assert steps_digest_ints(sg + sr + [fin], L) == case["steps_sha256"]
        if "steps" in case:
            assert [[format(v, "x") for v in st] for st in sg + sr + [fin]] == case["steps"]
    for case in g["add"]:
        n, c1, c2 = (H(case[k]) for k in ("n", "c1", "c2"))
        assert P.paillier_add_native(n, c1, c2) == H(case["res"])
        if "advice_sha256" in case:   # the reference's add-test shape (264-bit key, 88-bit limbs): cell streams pinned
            res, st = P.add_trace(n, c1, c2)
            W, lb = case["limb_bits"], case["lookup_bits"]
            adv, lk = P.expand_mul_mod_cells(st[0], st[1], st[2], st[3], n * n, 2 * case["enc_bits"] // W, lb, W)
            assert (len(adv), len(lk)) == (case["advice_cells"], case["lookup_cells"])
            assert cells_digest(adv) == case["advice_sha256"] and cells_digest(lk) == case["lookup_sha256"]


def cells_digest(cells):
    import hashlib

    h = hashlib.sha256()
    for v in cells:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


def test_paillier_golden_c(cref):
    g = load_golden("paillier.json")
    for case in g["encrypt"]:
        n, gg, m, r = (H(case[k]) for k in ("n", "g", "m", "r"))
        Ln = case["enc_bits"] // 64
        L = 2 * Ln
        assert cref.paillier_enc(Ln, n, gg, m, r) == H(case["c"])
        rc, res_g, st_g = cref.pow_mod_trace(L, n * n, gg, m, Ln)
        rc2, res_r, st_r = cref.pow_mod_trace(L, n * n, r, n, Ln)
        assert rc == 0 and rc2 == 0
        rc3, q, rr = cref.mul_mod_step(L, res_g, res_r, n * n)
        fin = np.stack([cref.int_to_limbs(v, L) for v in (res_g, res_r, q, rr)])[None]
        allsteps = np.concatenate([st_g, st_r, fin])
        assert steps_digest_arr(allsteps, L) == case["steps_sha256"]


def test_msm_golden(cref):
    g = load_golden("msm.json")
    for case in g["cases"]:
        n = case["n"]
        bases = cref.walk_bases(n, H(case["walk_s"]), H(case["walk_t"]))
        for i in case["identity_at"]:
            bases[i] = 0
        scalars = [H(x) for x in case["scalars"]]
        got = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont(scalars), bases, threads=2))
        assert cref.affine_mont_to_ints(got)[0] == (H(case["result"][0]), H(case["result"][1]))


def test_ntt_golden(cref):
    g = load_golden("ntt.json")
    for case in g["cases"]:
        a = [H(x) for x in case["a"]]
        got = cref.ntt_fr(cref.fr_ints_to_mont(a), cref.fr_ints_to_mont([H(case["omega"])])[0], case["log_n"])
        assert cref.fr_mont_to_ints(got) == [H(x) for x in case["out"]]
    for case in g["digests"]:
        a = [H(x) for x in case["a"]]
        omega = P.fr_omega(case["log_n"])
        got = cref.fr_mont_to_ints(cref.ntt_fr(cref.fr_ints_to_mont(a), cref.fr_ints_to_mont([omega])[0], case["log_n"]))
        h = hashlib.sha256(b"".join(x.to_bytes(32, "little") for x in got)).hexdigest()
        assert h == case["out_sha256"]
