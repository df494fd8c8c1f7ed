"""The prover steps AFTER the hot path alone on the GPU, as bench.py's with_next_rows runs them (ProofWorkload.tail_products /
tail_quotient / tail_evals on a c2 proof's own columns): the workload rocprofv3 is pointed at for the tail kernels' stats and
counters.  Usage: python profiles/probes/tail_only.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import paillier_halo2_amd as pz

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = pz.Engine(0)
eng.bind_torch_stream()
wl = bench.ProofWorkload(eng, torch, 2048, 17, seed=0x5043, scale=1.0)
wl.run(1)                 # slot 0 holds a proof's columns, d_ext the last extended tile
torch.cuda.synchronize()
wl.tail_setup()
for name, fn in (("products", lambda: wl.tail_products(0)), ("quotient", wl.tail_quotient), ("evaluations_and_openings", wl.tail_evals)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print("%-28s %8.2f ms per proof" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
