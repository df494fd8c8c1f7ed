"""GPU: the C++ host mirror of the reference's chip interface (paillier_halo2_amd/host/) driven by
tests/cpp/test_paillier.cpp, which restates the reference's four #[test] functions with seeded inputs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_mirror_of_reference_tests():
    exe = os.path.join(ROOT, "tests", "cpp", "test_paillier")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(p.stdout[-3000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "ALL OK" in p.stdout


def test_cpp_mirror_builds():
    """CPU: the header-only host mirror compiles and links against the C ABI"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-B"], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "test_paillier"))


def _splitmix64(x):
    import numpy as np

    M = np.uint64
    with np.errstate(over="ignore"):
        x = x + M(0x9E3779B97F4A7C15)
        x = (x ^ (x >> M(30))) * M(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> M(27))) * M(0x94D049BB133111EB)
        return x ^ (x >> M(31))


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,world,split", [(16, 2, "windows"), (16, 8, "windows"), (16, 2, "points"), (16, 8, "points"), (20, 8, "points"),
                                               (20, 8, "windows"), (14, 3, "windows")])
def test_cpp_sharded_msm_contexts(log_n, world, split):
    """config c4 from plain C++ (paillier_halo2_amd/host/msm_sharded.cpp): N contexts on this one device, a host thread each, its
    window range or point range through pz_msm_g1_dev, 96-byte downloads, pz_g1_sum in rank order == the whole MSM on one context
    == the closed form of the bases' discrete logarithms (recomputed here from the program's counter-mode generator)."""
    import json

    import numpy as np

    from oracle import cref, pyref as P
    from tests.util import walk_dlog_sum

    exe = os.path.join(ROOT, "tests", "cpp", "msm_sharded")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    seed = 0x5045 + log_n
    p = subprocess.run([exe, str(log_n), str(world), split, str(seed), "2"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["equal"] is True and out["multi_call_equal"] is True and out["contexts"] == world and out["split"] == split
    assert out["whole_affine_mont"] == out["sharded_affine_mont"]
    # the closed form: sum_i c_i (s + i t) with the program's inputs
    n = 1 << log_n
    idx = np.arange(4 * n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        sc = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + idx).reshape(n, 4)
    sc[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    s, t = int(out["s_lo"]) + (1 << 200), int(out["t_lo"])
    assert s == (int(_splitmix64(np.uint64(seed))) | 1) + (1 << 200)
    want = P.g1_mul(P.G1_GEN, walk_dlog_sum(sc, s, t))
    got = np.array([[int(out["whole_affine_mont"][c][64 - 16 * (k + 1): 64 - 16 * k], 16) for c in (0, 1) for k in range(4)]], dtype=np.uint64)
    assert tuple(cref.affine_mont_to_ints(got)[0]) == tuple(want)


@pytest.mark.gpu
def test_cpp_prover_replicas_small_job():
    """config c5's shape of work from plain C++: N independent provers (threads, three contexts each) on one device, timed as one
    job; every replica verifies its own pipeline and all replicas' commitment hashes agree"""
    import argparse
    import json

    import torch

    import bench
    import paillier_halo2_amd as pz

    eng = pz.Engine(0)
    eng.bind_torch_stream()
    try:
        wl = bench.ProofWorkload(eng, torch, 256, 12, seed=0x78, scale=1.0, pool=32)
        out = bench.dropin_device_resident(wl, argparse.Namespace(steps=2, warmup=1, seed=0x78), lambda s: None, replicas=3)
        assert "error" not in out, out
        assert out["replicas"] == 3 and out["verified"] is True and len(out["per_replica"]) == 3
        hs = [json.dumps(r["commitment_hash_by_message"], sort_keys=True) for r in out["per_replica"]]
        assert len(set(hs)) == 1 and all(r["verified"] for r in out["per_replica"])
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("circuit", ["encrypt", "encrypt_uniform"])
def test_cpp_connected_proof_verifies(cref, tmp_path, circuit):
    """the compiled prover (host/create_proof.hpp over the C ABI, no torch): keygen + two connected proofs of the reference's bench shape
    (128-bit n, k = 14), each checked the way halo2's verifier would (oracle/verifier.py) -- degree, the identity at x from the
    evaluations alone, SHPLONK against the proof's commitments and the key's -- with the challenges the binary's own transcript drew"""
    import random

    from oracle import pyref as P
    from oracle import verifier as V
    from paillier_halo2_amd import circuit_structure as CS
    from paillier_halo2_amd import consts, prover, prover_job

    BITS, K, LB = 128, 14, 13
    R = P.FR_R
    nn, g, m, r = P.synth_paillier_inputs(BITS, 0x5043, standard_g=False)
    uniform = circuit == "encrypt_uniform"
    sa = CS.stream_structure(circuit, BITS, 64, LB, m, nn)
    ng, nr = sa.n_steps_g, sa.n_steps_r
    st, starts = CS.columns(sa, K, LB, device="cpu")
    rng = random.Random(0x6a6f62)
    msgs = [(m, r), ((rng.randrange(0, nn) if uniform else m), rng.randrange(1, nn))]
    s_tox = rng.randrange(2, R)
    job, proof = str(tmp_path / "job.bin"), str(tmp_path / "proof.bin")
    prover_job.write_job(job, st, starts, BITS, 2 if uniform else 0, ng, nr, nn, g, msgs, s_tox, seed=7, proofs=2, tile=64)
    line = prover_job.run(job, proof)
    assert line["quotient_degree_ok"] is True and line["proofs"] == 2
    rec = prover_job.read_proofs(proof)
    A, Lk, mcols, S, n = st.n_adv, st.n_lk, st.m, -(-st.m // prover.CHUNK), 1 << K
    dom = prover.Domain(K, st.blinding_factors)

    def ints(a):
        flat = cref.fr_mont_to_ints(np.ascontiguousarray(a).reshape(-1, 4))
        p = a.shape[1] // 4
        return [flat[i * p:(i + 1) * p] for i in range(a.shape[0])]

    L = lambda x: sum(int(v) << (64 * i) for i, v in enumerate(x))
    for pi in range(2):
        pre = "p%d/" % pi
        mm, rr = msgs[pi]
        assert L(rec[pre + "ciphertext"][0]) == P.paillier_enc_native(nn, g, mm, rr)
        assert rec[pre + "flags"][0].tolist() == [1, pi]
        ch = {nm: L(rec[pre + "ch/" + nm][0]) for nm in ("theta", "beta", "gamma", "y", "x", "sh_y", "sh_v", "sh_u")}
        ev = {k_[len(pre) + 2:]: ints(v_) for k_, v_ in rec.items() if k_.startswith(pre + "e/")}
        ev["constants"] = ev["lookup_advice"][Lk:]
        ev["lookup_advice"] = ev["lookup_advice"][:Lk]
        want = V.expected_h(K, st.blinding_factors, A, Lk, prover.CHUNK, ev, ch["beta"], ch["gamma"], ch["y"], ch["x"], prover.DELTA)
        assert want == ev["h"][0][0], "h(x)(x^n - 1) != the expression of the evaluations (proof %d)" % pi
        com = {k_[len(pre) + 2:]: v_ for k_, v_ in rec.items() if k_.startswith(pre + "c/")}
        assert com["advice"].shape == (A, 8) and com["perm_z"].shape == (S, 8) and com["h"].shape == (3, 8)
        xn = pow(ch["x"], n, R)
        hc = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont([pow(xn, i, R) for i in range(3)]), com["h"]))
        com.update(fixed=rec["vk/fixed"], sigma=rec["vk/sigma"], h=[hc])
        lay = prover.query_layout(A, Lk, mcols, S)
        pts = prover.rotation_points(dom, ch["x"])
        assert V.shplonk_check(cref, lay, pts, com, ev, ch["sh_y"], ch["sh_v"], ch["sh_u"], com["w1"][0], com["w2"][0], s_tox), pi
    # the two proofs are of different randomness (and, uniform circuit, different messages) under ONE key
    assert not np.array_equal(rec["p0/c/advice"], rec["p1/c/advice"])
    # negative control: one bit of one witness cell flipped after K4 (a gated cell of the first column) -> the quotient is not a
    # polynomial of degree <= 3n - 4, and the identity at x fails
    bad = str(tmp_path / "bad.bin")
    env = dict(os.environ, PZ_PROVE_TAMPER_WORD=str(4 * 5))
    line = prover_job.run(job, bad, env=env, allow_unsatisfied=True)
    assert line["quotient_degree_ok"] is False
    rec = prover_job.read_proofs(bad)
    ch = {nm: L(rec["p0/ch/" + nm][0]) for nm in ("beta", "gamma", "y", "x")}
    ev = {k_[5:]: ints(v_) for k_, v_ in rec.items() if k_.startswith("p0/e/")}
    ev["constants"] = ev["lookup_advice"][Lk:]
    ev["lookup_advice"] = ev["lookup_advice"][:Lk]
    assert V.expected_h(K, st.blinding_factors, A, Lk, prover.CHUNK, ev, ch["beta"], ch["gamma"], ch["y"], ch["x"], prover.DELTA) != ev["h"][0][0]
    print(line)
