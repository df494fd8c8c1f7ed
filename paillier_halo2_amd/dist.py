"""Multi-GPU split of ONE large MSM (config c4): disjoint Pippenger windows per rank (north_star's split) or
disjoint point ranges per rank (SURVEY.md section 8e's alternative: N/world bases and scalars per GPU, 1/world of the
table memory and scalar reads), one exchange either way.

Each rank holds the whole base table and all scalars (402 MB at 2^22 -- nothing next to 288 GB),
accumulates only its window range [lo, hi) (pz_msm_g1_dev's win_lo / win_hi) and produces one
Jacobian point; the exchange is an all-gather of world x 96 bytes over RCCL (xGMI) followed by the
same fixed-order elliptic-curve fold on every rank, on the device (pz_g1_sum_dev) -- an all-reduce in effect (EC
addition is not an ncclRedOp).  Latency-bound: per-link bandwidth is irrelevant at 96 B.

The reference has no distributed path at all; this module is new.
`partial_fn` / `fold_fn` are injection points so the sharding + collective logic can be tested on
CPU with gloo (tests/test_dist_cpu.py); the defaults call the HIP library and nothing else.
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Tuple

import numpy as np


def window_range(n_windows: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous, disjoint, exhaustive split of [0, n_windows) -- rank r gets [lo, hi)"""
    return (rank * n_windows) // world, ((rank + 1) * n_windows) // world


def point_range(n_points: int, rank: int, world: int) -> Tuple[int, int]:
    """the same contiguous split applied to the points [0, n) instead of the windows"""
    return window_range(n_points, rank, world)


def column_range(n_cols: int, rank: int, world: int) -> Tuple[int, int]:
    """column-parallel proving (SURVEY.md section 8e, c2 at > 1 GPU): rank r commits / transforms columns [lo, hi)"""
    return window_range(n_cols, rank, world)


def gather_commitments(torch, dist, share, n_cols: int, rank: int, world: int):
    """share: (hi - lo, 12) int64 tensor with this rank's Jacobian commitments for its column_range(n_cols); returns the
    (n_cols, 12) tensor of all columns on every rank.  One all-gather of equal-sized (padded) shares: n_cols * 96 B in
    total (474 KB for a c2 proof) -- latency-bound on xGMI, like the 96-byte exchange of the sharded MSM."""
    lo, hi = column_range(n_cols, rank, world)
    assert share.shape[0] == hi - lo
    if dist is None:     # no process group; a group of ONE rank still runs its collective (bench.py --force-dist: the RCCL path on one GPU)
        return share
    per = -(-n_cols // world)
    pad = torch.zeros((per, share.shape[1]), dtype=share.dtype, device=share.device)
    pad[: hi - lo] = share
    out = torch.zeros((world * per, share.shape[1]), dtype=share.dtype, device=share.device)
    dist.all_gather_into_tensor(out, pad)
    parts = []
    for r in range(world):
        a, b = column_range(n_cols, r, world)
        parts.append(out[r * per: r * per + (b - a)])
    return torch.cat(parts)


def sharded_msm(torch, dist, rank: int, world: int, n_units: int,
                partial_fn: Callable[[int, int], "torch.Tensor"],
                fold_fn: Callable[["torch.Tensor"], "torch.Tensor"]):
    """n_units = number of windows (window split) or of points (point split); partial_fn(lo, hi) -> int64 tensor
    (12,) holding this rank's partial point for units [lo, hi) (on the device the process group communicates from);
    fold_fn(parts: (world, 12) int64 tensor on that same device, rank order == the fixed fold order) -> (12,) tensor.
    Returns the full MSM as a Jacobian point, identical on every rank, still on the device: nothing on this path
    touches the host (the all-gather lands in a device tensor, the fold is one small kernel)."""
    lo, hi = window_range(n_units, rank, world)
    part = partial_fn(lo, hi).reshape(12).contiguous()
    if dist is None:     # no process group (a group of one rank still all-gathers: the same calls an 8-rank run makes)
        return fold_fn(part.reshape(1, 12))
    parts = torch.empty(world * 12, dtype=part.dtype, device=part.device)
    dist.all_gather_into_tensor(parts, part)
    return fold_fn(parts.view(world, 12))


def hip_fold_fn(eng, torch):
    """the fixed-order EC fold on the device (pz_g1_sum_dev).  The kernel writes a resident 96-byte buffer; what is returned is a
    copy of it (a device-to-device copy on the same stream), so a caller may keep one MSM's result across the next call"""
    d_res = torch.zeros(12, dtype=torch.int64, device="cuda")

    def fn(parts):
        eng.g1_sum_dev(parts.data_ptr(), parts.shape[0], d_res.data_ptr())
        return d_res.clone()

    return fn


def hip_partial_fn(eng, torch, bases, d_scalars, n: int):
    d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")

    def fn(lo, hi):
        eng.msm_dev(bases, d_scalars.data_ptr(), 1, n, 4 * n, d_out.data_ptr(), lo, hi)
        return d_out

    return fn


def hip_partial_points_fn(eng, torch, d_bases, d_scalars, n: int, rank: int, world: int):
    """point split: this rank's table holds only its own N/world bases (built once, outside the timed loop)"""
    lo, hi = point_range(n, rank, world)
    d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    bases = eng.load_bases_dev(d_bases.data_ptr() + lo * 64, hi - lo) if hi > lo else None

    def fn(lo_, hi_):
        assert (lo_, hi_) == (lo, hi)
        if bases is None:
            d_out.zero_()  # Jacobian identity: z = 0
            return d_out
        eng.msm_dev(bases, d_scalars.data_ptr() + lo * 32, 1, hi - lo, 4 * (hi - lo), d_out.data_ptr(), 0, bases.n_windows)
        return d_out

    return fn, bases


class EmulatedRank:
    """Stands in for `torch.distributed` when ONE process plays rank `rank` of `world` over a real process group of one rank (RCCL on
    the one GPU there is): every collective the caller makes still goes through the real communicator -- with this process's data as
    the group's only contribution -- and the result is placed in slot `rank` of the gathered tensor, the other slots being what the
    other emulated ranks contributed when THEIR turn ran (kept in `store`, keyed by call order).  Rank-dependent code -- offsets,
    ranges, padding -- thus executes on the GPU for every rank, not only for rank 0."""

    class ReduceOp:
        MAX = "max"

    def __init__(self, real_dist, rank: int, world: int, store: dict):
        self.real, self.rank, self.world, self.store = real_dist, rank, world, store
        self.calls = 0

    def get_world_size(self):
        return self.world

    def all_gather_into_tensor(self, out, inp):
        import torch

        per = inp.numel()
        assert out.numel() == self.world * per
        mine = torch.empty_like(inp)
        if self.real is not None:
            self.real.all_gather_into_tensor(mine, inp.contiguous())     # the real collective: a group of one rank
        else:
            mine.copy_(inp)
        slot = self.store.setdefault(self.calls, {})
        slot[self.rank] = mine.reshape(-1).clone()
        flat = out.view(-1)
        flat.zero_()
        for r, t in slot.items():
            flat[r * per:(r + 1) * per] = t
        self.calls += 1


def emulate_ranks_msm(eng, torch, real_dist, world: int, log_n: int, split: str, log=lambda s: None):
    """config c4 with EVERY rank's exact code path run in turn on this GPU (sharded_msm with rank = r of `world`: the `lo * 64` /
    `lo * 32` offsets of the point split, the window ranges of the window split), each through the real one-rank process group.  Two
    passes: the first fills the emulated group's slots, in the second every rank sees all partials and folds them.  -> dict with the
    equality of every rank's folded point with the whole MSM computed in one call."""
    n = 1 << log_n
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0x5045)

    def rand_fr(count):
        x = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
        x[:, 3] &= 0x0FFFFFFFFFFFFFFF
        return x

    ks = rand_fr(n)
    d_b = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.g1_fixed_base_mul_dev(ks.data_ptr(), n, d_b.data_ptr())
    eng.sync()
    d_s = rand_fr(n)
    whole = eng.load_bases_dev(d_b.data_ptr(), n)
    d_ref = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    eng.msm_dev(whole, d_s.data_ptr(), 1, n, 4 * n, d_ref.data_ptr())
    eng.sync()
    ref = eng.g1_normalize(d_ref.cpu().numpy().astype(np.uint64))[0]
    fold = hip_fold_fn(eng, torch)
    store: dict = {}
    equal = []
    for pass_ in range(2):
        for r in range(world):
            em = EmulatedRank(real_dist, r, world, store)
            if split == "points":
                pfn, bases = hip_partial_points_fn(eng, torch, d_b, d_s, n, r, world)
                units = n
            else:
                bases, pfn, units = whole, hip_partial_fn(eng, torch, whole, d_s, n), whole.n_windows
            res = sharded_msm(torch, em, r, world, units, pfn, fold)
            eng.sync()
            if pass_ == 1:
                equal.append(bool(np.array_equal(eng.g1_normalize(res.cpu().numpy().astype(np.uint64))[0], ref)))
            if split == "points" and bases is not None:
                bases.free()
    whole.free()
    log("msm emulate-ranks %d (%s split, 2^%d): every rank's fold == whole MSM: %s" % (world, split, log_n, all(equal)))
    return {"world": world, "split": split, "log_n": log_n, "ranks_equal_whole_msm": equal, "all_equal": all(equal),
            "through_process_group": real_dist is not None}


def emulate_ranks_columns(eng, torch, real_dist, world: int, bases, d_cols, n_cols: int, n: int, log=lambda s: None):
    """column-parallel proving (--parallel columns) with every rank's share computed in turn: rank r commits column_range(n_cols, r,
    world) and gather_commitments pads / gathers / re-cuts; after the second pass every rank's gathered matrix must equal the
    commitments of all columns computed in one call."""
    d_all = torch.zeros((n_cols, 12), dtype=torch.int64, device="cuda")
    eng.msm_dev(bases, d_cols.data_ptr(), n_cols, n, 4 * n, d_all.data_ptr())
    eng.sync()
    ref = eng.g1_normalize(d_all.cpu().numpy().astype(np.uint64))
    store: dict = {}
    equal = []
    for pass_ in range(2):
        for r in range(world):
            em = EmulatedRank(real_dist, r, world, store)
            lo, hi = column_range(n_cols, r, world)
            share = torch.zeros((hi - lo, 12), dtype=torch.int64, device="cuda")
            if hi > lo:
                eng.msm_dev(bases, d_cols.data_ptr() + lo * n * 32, hi - lo, n, 4 * n, share.data_ptr())
            got = gather_commitments(torch, em, share, n_cols, r, world)
            eng.sync()
            if pass_ == 1:
                equal.append(bool(np.array_equal(eng.g1_normalize(got.cpu().numpy().astype(np.uint64)), ref)))
    log("columns emulate-ranks %d (%d columns): every rank's gathered commitments == all columns at once: %s" % (world, n_cols, all(equal)))
    return {"world": world, "n_cols": n_cols, "ranks_equal_single_call": equal, "all_equal": all(equal), "through_process_group": real_dist is not None}


def bench_sharded_msm(eng, torch, dist, rank, world, log_n, steps, warmup, barrier, log, split="windows", scalars="uniform"):
    """config c4: one 2^log_n-point MSM, windows (or point ranges) sharded over `world` ranks."""
    n = 1 << log_n
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0x5045)  # same on every rank: identical bases and scalars

    def rand_fr(count):
        x = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
        x[:, 3] &= 0x0FFFFFFFFFFFFFFF
        return x

    t0 = time.time()
    ks = rand_fr(n)
    d_b = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.g1_fixed_base_mul_dev(ks.data_ptr(), n, d_b.data_ptr())
    eng.sync()
    d_s = rand_fr(n)
    if scalars == "witness":
        # 60 % of the scalars below 2^16, 30 % below 2^64, 10 % below 2^135 (canonical values)
        cls = torch.rand(n, device="cuda", generator=gen)
        d_s[:, 3] = 0
        d_s[:, 2] = torch.where(cls >= 0.9, d_s[:, 2] & 0x7F, torch.zeros_like(d_s[:, 2]))
        d_s[:, 1] = torch.where(cls >= 0.9, d_s[:, 1], torch.zeros_like(d_s[:, 1]))
        d_s[:, 0] = torch.where(cls < 0.6, d_s[:, 0] & 0xFFFF, d_s[:, 0])
        eng.fr_convert_dev(d_s.data_ptr(), n, True)   # the ABI takes Montgomery form
    if split == "points":
        pfn, bases = hip_partial_points_fn(eng, torch, d_b, d_s, n, rank, world)
        units = n
    else:
        bases = eng.load_bases_dev(d_b.data_ptr(), n)
        pfn = hip_partial_fn(eng, torch, bases, d_s, n)
        units = bases.n_windows
    del d_b, ks
    log("msm22 setup %.1fs (n=2^%d, c=%d, %d windows, split by %s)" % (time.time() - t0, log_n, bases.window_bits, bases.n_windows, split))
    fold = hip_fold_fn(eng, torch)
    res = None
    for _ in range(warmup):
        res = sharded_msm(torch, dist, rank, world, units, pfn, fold)
    barrier()
    eng.timing_enable(True)
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = sharded_msm(torch, dist, rank, world, units, pfn, fold)
    barrier()
    dt = time.perf_counter() - t0
    acc_ms, acc_n = eng.timing_get(0)
    eng.timing_enable(False)
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    aff = eng.g1_normalize(res.cpu().numpy().astype(np.uint64))[0]
    alg = n * 96.0
    ach = alg * steps / dt / 1e9
    return {
        "metric": "MSM/s (one 2^%d-point BN254 G1 MSM, Pippenger windows sharded across GPUs)" % log_n,
        "value": steps / dt, "unit": "MSM/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32 limbs (254-bit modular integers)", "data": "synthetic",
        "config": {"workload": "c4: single 2^%d-point MSM, %s scalars" % (log_n, "witness-like" if scalars == "witness" else "uniform"), "window_bits": bases.window_bits,
                   "n_windows": bases.n_windows, "split": split,
                   "parallelism": "%s/%d + all_gather(96 B) + fold" % (split, world)},
        "roofline": {"bound": "hbm", "kernel": "whole MSM", "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                     "frac": ach / 8000.0, "traffic": None, "accumulate_ms_per_msm": acc_ms / max(1, steps)},
        "result_affine_x_limb0": int(aff[0]),
    }


def emulate_sharded_msm(eng, torch, world: int, log_n: int, steps: int, warmup: int, log, scalars="uniform", share_window_bits=0,
                        window_split_bits=0):
    """config c4 on ONE GPU: each of `world` ranks' shares of the 2^log_n-point MSM run in turn (window split and point
    split), with per-share stage times from HIP events (digit sort, bucket accumulation, bucket folds + reduction tree) and
    the device fold of the `world` partial points.  predicted_efficiency = T(one GPU, whole MSM) / (world * T(slowest
    share + fold)): what an N-GPU run can reach at best (the 96-byte all-gather, tens of microseconds over xGMI, comes on
    top and is not emulated).  The shares' results are folded and compared with the whole MSM."""
    from . import engine as E

    n = 1 << log_n
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0x5045)

    def rand_fr(count):
        x = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
        x[:, 3] &= 0x0FFFFFFFFFFFFFFF
        return x

    ks = rand_fr(n)
    d_b = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.g1_fixed_base_mul_dev(ks.data_ptr(), n, d_b.data_ptr())
    eng.sync()
    d_s = rand_fr(n)
    if scalars == "witness":
        cls = torch.rand(n, device="cuda", generator=gen)
        d_s[:, 3] = 0
        d_s[:, 2] = torch.where(cls >= 0.9, d_s[:, 2] & 0x7F, torch.zeros_like(d_s[:, 2]))
        d_s[:, 1] = torch.where(cls >= 0.9, d_s[:, 1], torch.zeros_like(d_s[:, 1]))
        d_s[:, 0] = torch.where(cls < 0.6, d_s[:, 0] & 0xFFFF, d_s[:, 0])
        eng.fr_convert_dev(d_s.data_ptr(), n, True)
    del ks
    d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    d_parts = torch.zeros((world, 12), dtype=torch.int64, device="cuda")
    eng.timing_enable(True)

    def timed(fn):
        """-> dict of ms per run: wall (device time between two events on the stream) + the library's stage classes"""
        for _ in range(warmup):
            fn()
        eng.sync()
        eng.timing_reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        e1.synchronize()
        out = {"total": e0.elapsed_time(e1) / steps}
        for name, cls_ in (("sort", E.T_MSM_SORT), ("accumulate", E.T_MSM_ACC), ("tree", E.T_MSM_TREE)):
            out[name] = eng.timing_get(cls_)[0] / steps
        return out

    tb = eng.load_bases_dev(d_b.data_ptr(), n)
    full = timed(lambda: eng.msm_dev(tb, d_s.data_ptr(), 1, n, 4 * n, d_out.data_ptr()))
    eng.sync()
    ref = eng.g1_normalize(d_out.cpu().numpy().astype(np.uint64))[0]
    res = {"full": full, "splits": {}}
    t_fold = timed(lambda: eng.g1_sum_dev(d_parts.data_ptr(), world, d_out.data_ptr()))["total"]
    # ---- window split (window_split_bits: the shares' table with another window width than the one-GPU run's -- more, narrower windows
    # divide more evenly over the ranks and shrink every share's bucket set)
    tbw = eng.load_bases_dev(d_b.data_ptr(), n, window_split_bits) if window_split_bits else tb
    shares = []
    for r in range(world):
        lo, hi = window_range(tbw.n_windows, r, world)
        shares.append(timed(lambda: eng.msm_dev(tbw, d_s.data_ptr(), 1, n, 4 * n, d_parts[r].data_ptr(), lo, hi)))
    eng.g1_sum_dev(d_parts.data_ptr(), world, d_out.data_ptr())
    eng.sync()
    ok_w = bool(np.array_equal(eng.g1_normalize(d_out.cpu().numpy().astype(np.uint64))[0], ref))
    worst = max(sh["total"] for sh in shares)
    res["splits"]["windows"] = {"shares": shares, "slowest_share_ms": worst, "fold_ms": t_fold, "window_bits": window_split_bits or 16,
                                "windows": tbw.n_windows,
                                "predicted_efficiency": full["total"] / (world * (worst + t_fold)), "equals_full_msm": ok_w}
    if tbw is not tb:
        tbw.free()
    tb.free()
    # ---- point split: every rank's table holds its own n / world bases
    shares = []
    for r in range(world):
        lo, hi = point_range(n, r, world)
        tr = eng.load_bases_dev(d_b.data_ptr() + lo * 64, hi - lo, share_window_bits)
        shares.append(timed(lambda: eng.msm_dev(tr, d_s.data_ptr() + lo * 32, 1, hi - lo, 4 * (hi - lo), d_parts[r].data_ptr())))
        eng.sync()
        tr.free()
    eng.g1_sum_dev(d_parts.data_ptr(), world, d_out.data_ptr())
    eng.sync()
    ok_p = bool(np.array_equal(eng.g1_normalize(d_out.cpu().numpy().astype(np.uint64))[0], ref))
    worst = max(sh["total"] for sh in shares)
    res["splits"]["points"] = {"shares": shares, "slowest_share_ms": worst, "fold_ms": t_fold, "window_bits": share_window_bits or 16,
                               "predicted_efficiency": full["total"] / (world * (worst + t_fold)), "equals_full_msm": ok_p}
    eng.timing_enable(False)
    best = max(res["splits"], key=lambda k: res["splits"][k]["predicted_efficiency"])
    log("msm22 emulate-world %d: full %.2f ms; windows %.3f, points %.3f predicted efficiency" % (
        world, full["total"], res["splits"]["windows"]["predicted_efficiency"], res["splits"]["points"]["predicted_efficiency"]))
    return {
        "metric": "predicted %d-GPU efficiency of one 2^%d-point MSM (each rank's share emulated on one GPU)" % (world, log_n),
        "value": res["splits"][best]["predicted_efficiency"], "unit": "fraction of linear speed-up", "n_gpus": 1, "steps": steps,
        "warmup": warmup, "ms_per_step": full["total"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32 limbs (29-bit reduced radix, 254-bit modular integers)", "data": "synthetic",
        "config": {"workload": "c4 emulated: single 2^%d-point MSM, %s scalars, world %d" % (log_n, "witness-like" if scalars == "witness" else "uniform", world),
                   "best_split": best, "point_share_window_bits": share_window_bits or 16},
        "emulation": res,
    }
