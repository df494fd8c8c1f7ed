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

    import bench_connected
    from oracle import pyref as P
    from paillier_halo2_amd import circuit_structure as CS
    from paillier_halo2_amd import prover_job

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
    for pi in range(2):
        mm, rr = msgs[pi]
        out = bench_connected.verify_file_proof(cref, rec, "p%d/" % pi, st, K, s_tox)
        assert out["verified"] is True, (pi, out)
        assert out["ciphertext"] == P.paillier_enc_native(nn, g, mm, rr)
        assert rec["p%d/flags" % pi][0].tolist() == [1, pi]
    # the two proofs are of different randomness (and, uniform circuit, different messages) under ONE key
    assert not np.array_equal(rec["p0/c/advice"], rec["p1/c/advice"])
    # the same job through the LIBRARY'S stepper (PZ_PROVE_VIA_STEPPER=1: pz_pk_create on the job's host arrays + pz_proof_*, the next
    # proof's witness written by a second host thread on a second context): the same key, proofs that verify
    via = str(tmp_path / "via.bin")
    line2 = prover_job.run(job, via, env=dict(os.environ, PZ_PROVE_VIA_STEPPER="1"))
    assert line2["quotient_degree_ok"] is True and "stepper" in line2["via"]
    rec2 = prover_job.read_proofs(via)
    assert np.array_equal(rec2["vk/fixed"], rec["vk/fixed"]) and np.array_equal(rec2["vk/sigma"], rec["vk/sigma"])
    for pi in range(2):
        mm, rr = msgs[pi]
        out2 = bench_connected.verify_file_proof(cref, rec2, "p%d/" % pi, st, K, s_tox)
        assert out2["verified"] is True, (pi, out2)
        assert out2["ciphertext"] == P.paillier_enc_native(nn, g, mm, rr)
    # negative control: one bit of one witness cell flipped after K4 (a gated cell of the first column) -> the quotient is not a
    # polynomial of degree <= 3n - 4, and the identity at x fails
    bad = str(tmp_path / "bad.bin")
    env = dict(os.environ, PZ_PROVE_TAMPER_WORD=str(4 * 5))
    line = prover_job.run(job, bad, env=env, allow_unsatisfied=True)
    assert line["quotient_degree_ok"] is False
    rec = prover_job.read_proofs(bad)
    out = bench_connected.verify_file_proof(cref, rec, "p0/", st, K, s_tox)
    assert out["verified"] is False and out["quotient_degree_le_3n_minus_4"] is False
    assert out["h_x_times_xn_minus_1_equals_expression_of_evaluations"] is False
    assert out["shplonk_identity_on_the_proofs_commitments"] is True      # the openings of a wrong proof are still honest openings
    print(line)


@pytest.mark.gpu
def test_cpp_fresh_message_mode_verifies(cref):
    """`prove_connected --fresh`: a NEW key pair and message per proof from compiled code alone -- the circuit structure generated by the
    library on the device per message (pz_circuit_structure_dev), keygen on its device arrays, K3 + K4, create_proof, device blocks
    recycled from key to key -- at the reference's bench shape (128-bit n, k = 14; /root/reference/src/bench.rs:139-140,161-171): three
    steps, the last proof checked as the verifier would with ITS OWN key's commitments, its ciphertext against g^m r^n mod n^2"""
    import bench_connected

    out = bench_connected.cpp_fresh_message(128, 14, 13, 0x5043, steps=3, verify_with=cref)
    assert out["quotient_degree_ok"] is True and out["verified"] is True, out
    assert out["verification"]["ciphertext_is_g_m_r_n"] is True and out["steps"] == 3
