"""GPU: the C++ host mirror of the reference's chip interface (paillier_halo2_amd/host/) driven by
tests/cpp/test_paillier.cpp, which restates the reference's four #[test] functions with seeded inputs."""
import os
import subprocess

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
