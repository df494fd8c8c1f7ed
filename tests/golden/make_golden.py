"""Generates tests/golden/*.json with the repo's own Python-int oracle (oracle/pyref.py).

The reference ships no golden vectors and is not runnable here, so these fixtures
pin the ORACLE's outputs on seeded inputs; tests/test_golden.py checks both CPU restatements
against them and tests/test_gpu_*.py check the HIP kernels against them.  Values are canonical
integers in hex (no Montgomery form) so the files are representation independent.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as P  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
hx = lambda x: format(x, "x")


def steps_digest(steps, L):
    h = hashlib.sha256()
    for st in steps:
        for v in st:
            h.update(v.to_bytes(8 * L, "little"))
    return h.hexdigest()


def cells_digest(cells):
    h = hashlib.sha256()
    for v in cells:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


def gen_paillier():
    cases = []
    # (enc_bits, limb_bits) -- the reference's two test shapes (paillier.rs:115-116, 186-187; bench.rs:139-140)
    # plus the BASELINE configs' 2048 / 3072-bit keys
    for enc_bits, seed, std_g in ((128, 0x5042, False), (128, 0x5142, True), (192, 0x5242, False),
                                   (2048, 0x5043, True), (3072, 0x5046, True)):
        n, g, m, r = P.synth_paillier_inputs(enc_bits, seed, standard_g=std_g)
        if enc_bits >= 2048:
            # keep the fixture small: short message, full-size key (the r^n chain is checked by digest)
            m = m & ((1 << 40) - 1)
        c, sg, sr, fin = P.encrypt_trace(n, g, m, r)
        L = 2 * (enc_bits // 64)
        case = dict(enc_bits=enc_bits, limb_bits=64, n=hx(n), g=hx(g), m=hx(m), r=hx(r), c=hx(c),
                    n_steps_g=len(sg), n_steps_r=len(sr), steps_sha256=steps_digest(sg + sr + [fin], L))
        if enc_bits <= 192:
            case["steps"] = [[hx(v) for v in st] for st in (sg + sr + [fin])]
        cases.append(case)
    adds = []
    rng = random.Random(0x5044)
    for enc_bits in (128, 2048):
        n, _, _, _ = P.synth_paillier_inputs(enc_bits, 0x5044)
        c1, c2 = rng.getrandbits(enc_bits), rng.getrandbits(enc_bits)
        res, st = P.add_trace(n, c1, c2)
        adds.append(dict(enc_bits=enc_bits, n=hx(n), c1=hx(c1), c2=hx(c2), res=hx(res), q=hx(st[2])))
    # true 4096-bit ciphertext operands (extension noted in SURVEY.md section 8d c3)
    n, g, m, r = P.synth_paillier_inputs(2048, 0x5045)
    c1 = pow(n + 1, 12345, n * n) * pow(3, n, n * n) % (n * n)
    c2 = pow(n + 1, 54321, n * n) * pow(5, n, n * n) % (n * n)
    res, st = P.add_trace(n, c1, c2)
    adds.append(dict(enc_bits=2048, full_width=True, n=hx(n), c1=hx(c1), c2=hx(c2), res=hx(res), q=hx(st[2])))
    # the reference's own add-test shape: 264-bit key on 88-bit limbs, lookup_bits 15 (paillier.rs:186-187, 247);
    # the cell streams of its single mul_mod step are pinned by digest
    rng88 = random.Random(0x5847)
    n, _, _, _ = P.synth_paillier_inputs(264, 0x5847)
    c1, c2 = rng88.getrandbits(264), rng88.getrandbits(264)
    res, st = P.add_trace(n, c1, c2)
    adv, lk = P.expand_mul_mod_cells(st[0], st[1], st[2], st[3], n * n, 6, 15, 88)
    adds.append(dict(enc_bits=264, limb_bits=88, lookup_bits=15, n=hx(n), c1=hx(c1), c2=hx(c2), res=hx(res), q=hx(st[2]),
                     advice_cells=len(adv), lookup_cells=len(lk), advice_sha256=cells_digest(adv), lookup_sha256=cells_digest(lk)))
    json.dump(dict(encrypt=cases, add=adds), open(os.path.join(OUT, "paillier.json"), "w"), indent=0)


def gen_msm():
    rng = random.Random(0x4D534D)
    cases = []
    for n in (1, 2, 63, 64, 65, 200):
        s, t = rng.randrange(1, P.FR_R), rng.randrange(1, P.FR_R)
        bases = P.walk_bases(n, s, t)
        scalars = [rng.randrange(P.FR_R) for _ in range(n)]
        edge = [0, 1, P.FR_R - 1, rng.getrandbits(64), rng.getrandbits(140), (P.FR_R - 1) // 2, (P.FR_R + 1) // 2]
        for i, v in enumerate(edge[:n]):
            scalars[i] = v
        if n >= 4:
            bases[3] = P.AFF_INF
        res = P.msm_pippenger(scalars, bases, c=4) if n > 8 else P.msm_naive(scalars, bases)
        cases.append(dict(n=n, walk_s=hx(s), walk_t=hx(t), identity_at=[3] if n >= 4 else [],
                          scalars=[hx(x) for x in scalars], result=[hx(res[0]), hx(res[1])]))
    # adversarial: G, 2G, 3G, ... with repeated / cancelling scalars
    n = 64
    scalars = [1] * 16 + [P.FR_R - 1] * 16 + [2] * 16 + [7] * 16
    res = P.msm_walk_expected(scalars, 1, 1)
    cases.append(dict(n=n, walk_s="1", walk_t="1", identity_at=[], scalars=[hx(x) for x in scalars],
                      result=[hx(res[0]), hx(res[1])]))
    json.dump(dict(cases=cases), open(os.path.join(OUT, "msm.json"), "w"), indent=0)


def gen_ntt():
    rng = random.Random(0x4E5454)
    cases = []
    for log_n in range(0, 9):
        a = [rng.randrange(P.FR_R) for _ in range(1 << log_n)]
        omega = P.fr_omega(log_n)
        cases.append(dict(log_n=log_n, omega=hx(omega), a=[hx(x) for x in a], out=[hx(x) for x in P.ntt(a, omega)]))
    # larger sizes by digest only
    dig = []
    for log_n in (10, 12):
        a = [rng.randrange(P.FR_R) for _ in range(1 << log_n)]
        omega = P.fr_omega(log_n)
        out = P.ntt(a, omega)
        h = hashlib.sha256(b"".join(x.to_bytes(32, "little") for x in out)).hexdigest()
        dig.append(dict(log_n=log_n, seed_hex=[hx(a[0]), hx(a[-1])], a=[hx(x) for x in a], out_sha256=h))
    json.dump(dict(cases=cases, digests=dig), open(os.path.join(OUT, "ntt.json"), "w"), indent=0)


if __name__ == "__main__":
    gen_paillier()
    gen_msm()
    gen_ntt()
    for f in ("paillier.json", "msm.json", "ntt.json"):
        print(f, os.path.getsize(os.path.join(OUT, f)))
