"""probe: accumulate / whole-MSM time per column, witness columns vs full-width columns (env PZ_MSM_CHUNK)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import paillier_halo2_amd as pz
import bench

eng = pz.Engine(0)
os.environ["PZ_BENCH_PIPELINE"] = "0"
wl = bench.ProofWorkload(eng, torch, 2048, 17, seed=0x5043, scale=1.0)
wl.produce(0)
torch.cuda.synchronize()
eng.timing_enable(True)
def run(name, fn, ncols):
    fn(); torch.cuda.synchronize()
    eng.timing_reset()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    acc, _ = eng.timing_get(0)
    print("%s chunk=%s cols=%d total %.1f ms (%.1f us/col) accumulate %.1f ms (%.1f us/col)" % (
        name, os.environ.get("PZ_MSM_CHUNK", "auto"), ncols, dt * 1e3, dt * 1e6 / ncols, acc, acc * 1e3 / ncols), flush=True)
nw = 512
run("witness", lambda: eng.msm_dev(wl.bases, wl.d_adv[0].data_ptr(), nw, wl.rows, 4 * wl.rows, wl.d_out_adv.data_ptr()), nw)
run("full   ", lambda: eng.msm_dev(wl.bases, wl.col_f.data_ptr(), wl.pool, wl.n, 4 * wl.n, wl.d_out.data_ptr()), wl.pool)
