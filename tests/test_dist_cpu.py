"""CPU, world_size 2 over gloo: the window-sharded MSM's partition + all-gather + fixed-order fold
(paillier_halo2_amd/dist.py).  The per-rank partial and the fold are injected from the oracle here
(tests may use it); on a GPU the defaults call libpz_hip.so (tests/test_gpu_kernels.py covers the
window-range entry point itself)."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyref as P
from paillier_halo2_amd.dist import column_range, gather_commitments, point_range, sharded_msm, window_range


def test_window_range_partition():
    for W in (1, 7, 16, 20, 24):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = window_range(W, r, world)
                assert 0 <= lo <= hi <= W
                cover += list(range(lo, hi))
            assert cover == list(range(W))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, c, seed, q, split="windows"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cref

    rng = random.Random(seed)
    s, t = rng.randrange(1, P.FR_R), rng.randrange(1, P.FR_R)
    bases = P.walk_bases(n, s, t)
    scalars = [rng.randrange(P.FR_R) for _ in range(n)]
    nwin = 253 // c + 1

    def partial(lo, hi):
        # this rank's windows only: sum_i (digits of scalar_i in [lo,hi)) * P_i, unsigned c-bit digits
        mask = (1 << c) - 1
        sc = [sum(((k >> (w * c)) & mask) << (w * c) for w in range(lo, hi)) for k in scalars]
        jac = cref.msm_g1(cref.fr_ints_to_mont(sc), cref.affine_ints_to_mont(bases))
        return torch.from_numpy(jac.astype(np.int64))

    def fold(parts_t):   # (world, 12) tensor, rank order
        parts = parts_t.numpy().astype(np.uint64)
        acc = parts[0]
        for p in parts[1:]:
            acc = cref.g1_add(acc, p)
        return torch.from_numpy(acc.astype(np.int64))

    def partial_points(lo, hi):
        assert (lo, hi) == point_range(n, rank, world)
        if hi == lo:
            return torch.zeros(12, dtype=torch.int64)
        jac = cref.msm_g1(cref.fr_ints_to_mont(scalars[lo:hi]), cref.affine_ints_to_mont(bases[lo:hi]))
        return torch.from_numpy(jac.astype(np.int64))

    if split == "points":
        res = sharded_msm(torch, dist, rank, world, n, partial_points, fold)
    else:
        res = sharded_msm(torch, dist, rank, world, nwin, partial, fold)
    got = cref.affine_mont_to_ints(cref.g1_normalize(res.numpy().astype(np.uint64)))[0]
    q.put((rank, got, P.msm_walk_expected(scalars, s, t)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,split", [(2, "windows"), (2, "points")])
def test_sharded_msm_gloo(world, split, cref):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 96, 13, 0x5045, q, split)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, want in out:
        assert got == want, rank
    assert out[0][1] == out[1][1]


def _gather_worker(rank, world, port, n_cols, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = column_range(n_cols, rank, world)
    share = torch.arange(lo * 12, hi * 12, dtype=torch.int64).reshape(hi - lo, 12)   # column c holds 12c .. 12c+11
    out = gather_commitments(torch, dist, share, n_cols, rank, world)
    q.put((rank, out.reshape(-1).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_cols", [7, 8, 1])
def test_column_parallel_gather_gloo(n_cols):
    """column-parallel proving: every rank ends up with all columns' commitments in column order, ragged split included"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, n_cols, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, vals in out:
        assert vals == list(range(12 * n_cols)), rank


def test_emulated_rank_reassembles_the_group():
    """EmulatedRank (one process playing every rank in turn): after a full pass every rank's gather holds all contributions in rank
    order, and gather_commitments re-cuts ragged column shares exactly like a real group of `world` ranks would"""
    from paillier_halo2_amd.dist import EmulatedRank

    world, n_cols = 5, 13
    cols = torch.arange(n_cols * 12, dtype=torch.int64).view(n_cols, 12)
    store = {}
    for pass_ in range(2):
        for r in range(world):
            em = EmulatedRank(None, r, world, store)
            lo, hi = column_range(n_cols, r, world)
            got = gather_commitments(torch, em, cols[lo:hi].clone(), n_cols, r, world)
            part = torch.full((12,), r, dtype=torch.int64)
            parts = sharded_msm(torch, em, r, world, 16, lambda a, b: part, lambda p: p.clone())
            if pass_ == 1:
                assert torch.equal(got, cols), r
                assert torch.equal(parts, torch.arange(world, dtype=torch.int64).view(world, 1).expand(world, 12)), r
