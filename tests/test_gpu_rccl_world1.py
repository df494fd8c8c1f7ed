"""The RCCL path on the hardware that exists: `bench.py --gpus 1 --force-dist` starts through
torch.distributed.run, creates the NCCL (= RCCL) process group on the one GPU and runs every collective the N > 1 path makes --
all_gather_into_tensor of the partial points on device tensors + pz_g1_sum_dev (msm22, both splits), the commitment all-gather of
--parallel columns (c2), the barriers and the MAX all-reduce of the timing -- so the first RCCL call this code makes is not on the
driver's 8-GPU node.  Results must equal the runs without a process group."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rccl_world1.err"), "a") as f:
            f.write(" ".join(extra) + "\n" + p.stdout[-4000:] + "\n----\n" + p.stderr[-8000:] + "\n")
    assert p.returncode == 0, (extra, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("split", ["windows", "points"])
def test_msm_sharded_world1_over_rccl(split):
    common = ["--gpus", "1", "--workload", "msm22", "--log-n", "18", "--steps", "3", "--warmup", "1", "--msm-split", split]
    plain = _bench(common)
    assert plain["backend"] is None and plain["rccl_ranks"] == 1
    forced = _bench(common + ["--force-dist"])
    assert forced["backend"] == "nccl" and forced["rccl_ranks"] == 1 and forced["n_gpus"] == 1
    assert forced["result_affine_x_limb0"] == plain["result_affine_x_limb0"]     # same point through all-gather + device fold
    assert forced["config"]["split"] == split and forced["value"] > 0
    out = os.path.join(ROOT, "gpurun_out", "r04_msm18_world1_nccl_%s.json" % split)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(forced, f)


def test_c2_column_parallel_world1_over_rccl():
    """ONE proof's columns split over the ranks of a one-rank group: the commitment all-gathers run on device tensors"""
    common = ["--gpus", "1", "--enc-bits", "256", "--k", "13", "--steps", "2", "--warmup", "1", "--parallel", "columns",
              "--no-cpu-baseline", "--no-dropin", "--no-tail", "--no-verify"]
    forced = _bench(common + ["--force-dist"])
    assert forced["backend"] == "nccl" and forced["rccl_ranks"] == 1
    assert forced["scaling"] == "strong" and forced["value"] > 0


@pytest.mark.parametrize("split", ["windows", "points"])
def test_every_ranks_code_path_through_the_rccl_group_of_one(split):
    """rank-dependent code (the `lo * 64` / `lo * 32` offsets of the point split, window ranges, padded gather
    slices of the column split) has only ever seen rank 0 on a GPU.  `--emulate-ranks 8 --force-dist`: one process plays ranks 0..7 in
    turn, each collective through the NCCL group of one; every rank's folded point equals the whole 2^18-point MSM, every rank's
    gathered commitment matrix equals all columns committed at once"""
    out = _bench(["--gpus", "1", "--workload", "msm22", "--log-n", "18", "--msm-split", split, "--emulate-ranks", "8", "--force-dist",
                  "--parallel", "columns"])
    assert out["backend"] == "nccl" and out["rccl_ranks"] == 1
    em = out["emulate_ranks"]
    assert em["through_process_group"] and em["world"] == 8 and em["split"] == split
    assert em["ranks_equal_whole_msm"] == [True] * 8
    ec = out["emulate_ranks_columns"]
    assert ec["through_process_group"] and ec["ranks_equal_single_call"] == [True] * 8
