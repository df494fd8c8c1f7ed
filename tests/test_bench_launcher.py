"""`python bench.py --gpus N` typed as it stands must start its own ranks: the parent makes no GPU
call, spawns torch.distributed.run as a child, forwards rank 0's JSON line and returns non-zero when a rank fails.
Rehearsed here on the CPU with the gloo backend and the stub workload (no library, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_world2_gloo():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "stub", "--backend", "gloo"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout          # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo"
    assert out["steps"] == 3 and out["warmup"] == 1 and out["value"] > 0
    assert out["comparable"] is False


def test_self_launch_propagates_failure():
    # a rank that dies (unknown backend) must make the parent exit non-zero
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "stub", "--backend", "no_such_backend"])
    assert p.returncode != 0


def test_parent_makes_no_gpu_call():
    """the launcher branch runs before torch / the library are imported: with both made un-importable in the PARENT only
    (a sitecustomize that poisons them unless RANK is set) the launch still succeeds"""
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "sitecustomize.py"), "w") as f:
            f.write("import os, sys\n"
                    "if 'RANK' not in os.environ and any(a.endswith('bench.py') for a in sys.argv[:1]):\n"
                    "    sys.modules['torch'] = None\n"
                    "    sys.modules['paillier_halo2_amd'] = None\n")
        pp = d + os.pathsep + os.environ.get("PYTHONPATH", "")
        p = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--workload", "stub", "--backend", "gloo"], {"PYTHONPATH": pp})
        assert p.returncode == 0, p.stderr[-2000:]
        assert json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])["rccl_ranks"] == 2


def test_single_rank_without_launcher():
    p = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--workload", "stub"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1


def test_force_dist_single_rank_creates_the_group():
    """--gpus 1 --force-dist: one rank, still through the launcher and a process group (what the GPU test runs with backend nccl)"""
    p = _run(["--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "0", "--workload", "stub", "--backend", "gloo"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["backend"] == "gloo"
