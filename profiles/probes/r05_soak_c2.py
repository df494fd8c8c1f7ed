"""Round-5 soak at FULL size (evidence, not a unit test): connected proofs of config c2 (2048-bit n, k = 17; 174 GB resident) back to
back for a wall-clock budget, EVERY proof checked as the verifier would (quotient degree, h(x)(x^n - 1) == the expression of the
evaluations, SHPLONK identity on the proof's commitments), with a NEW key and message (structure + keygen) every `rekey` proofs.
Usage: python profiles/probes/r05_soak_c2.py [seconds, default 600] [rekey, default 25]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import paillier_halo2_amd as pz
import bench_connected
from oracle import cref

cref.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
rekey = int(sys.argv[2]) if len(sys.argv) > 2 else 25
eng = pz.Engine(0)
eng.bind_torch_stream()
log = lambda s: None
t_start = time.time()
wl = bench_connected.ConnectedWorkload(eng, torch, 2048, 17, 0x50AC, log=log, trim=False)
srs = (wl.bl, wl.bm, wl.s_tox)
first = wl
proofs = bad = keys = 0
lat = []
while time.time() - t_start < budget:
    t0 = time.perf_counter()
    wl.step(timed=False)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
    v = wl.verify(cref)
    proofs += 1
    if not v["verified"]:
        bad += 1
        print("[soak] proof %d NOT verified: %s" % (proofs, v), flush=True)
    if proofs % 25 == 0:
        free, tot = torch.cuda.mem_get_info()
        print("[soak] %6.0f s: %d proofs (%d keys), %d failed; last 25: %.1f ms median; %.1f GB free" % (
            time.time() - t_start, proofs, keys + 1, bad, sorted(lat[-25:])[12] * 1e3, free / 1e9), flush=True)
    if proofs % rekey == 0:
        if wl is first:
            for nm in ("pk", "ws", "slots", "cols", "d_steps"):
                setattr(wl, nm, None)
        else:
            wl.release(trim=False)
        gc.collect()
        keys += 1
        wl = bench_connected.ConnectedWorkload(eng, torch, 2048, 17, 0x50AC + keys, log=log, srs=srs, trim=False)
print("[soak] c2 connected proofs: %d of %d verified over %.0f s, %d keys; median %.1f ms, max %.1f ms per proof" % (
    proofs - bad, proofs, time.time() - t_start, keys + 1, sorted(lat)[len(lat) // 2] * 1e3, max(lat) * 1e3), flush=True)
sys.exit(1 if bad else 0)
