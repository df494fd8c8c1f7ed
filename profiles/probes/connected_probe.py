"""The connected-proof workload of bench.py (bench_connected.ConnectedWorkload) alone: setup times, memory, per-phase times, the
checker's verdict.  Usage: python profiles/probes/connected_probe.py [enc_bits k steps [circuit]]   (default 2048 17 3 encrypt)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import paillier_halo2_amd as pz
import bench_connected

bits, k, steps = (int(x) for x in (sys.argv[1:4] + ["2048", "17", "3"][len(sys.argv) - 1:]))
circuit = sys.argv[4] if len(sys.argv) > 4 else "encrypt"
eng = pz.Engine(0)
eng.bind_torch_stream()
log = lambda s: print("[probe] " + s, flush=True)
t0 = time.time()
wl = bench_connected.ConnectedWorkload(eng, torch, bits, k, 0x5043, log=log, circuit=circuit)
log("setup %.1f s: structure %s keygen %.0f ms memory %s counts %s" % (time.time() - t0, wl.structure_ms, wl.keygen_ms, wl.memory_gb, wl.counts()))
wl.run(1, timed=False)
torch.cuda.synchronize()
log("warm-up proof done; torch max allocated %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
t1 = time.perf_counter()
wl.run(steps, timed=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t1
log("%d proofs: %.1f ms per proof (%.3f proofs/s)" % (steps, dt / steps * 1e3, steps / dt))
if os.environ.get("PZ_PROBE_IN_FLIGHT", "2") != "1":
    nl = int(os.environ.get("PZ_PROBE_IN_FLIGHT", "2"))
    free, tot = torch.cuda.mem_get_info()
    log("before the second lane: %.1f GB free of %.1f" % (free / 1e9, tot / 1e9))
    wl.run_in_flight(2, nl)            # warm-up: the second lane's workspaces
    free, tot = torch.cuda.mem_get_info()
    log("with %d lanes: %.1f GB free of %.1f" % (nl, free / 1e9, tot / 1e9))
    for rep in range(2):
        dtf = wl.run_in_flight(2 * steps, nl)
        log("%d proofs, %d in flight: %.1f ms per proof (%.3f proofs/s); latency per proof %s ms" % (
            2 * steps, nl, dtf / (2 * steps) * 1e3, 2 * steps / dtf, [round(x) for x in wl.in_flight_latency_ms]))
    from oracle import cref
    cref.build()
    log("verify (last proof of the in-flight run): %s" % wl.verify(cref))
wl.run(1, timed=True)
log("phases (one more proof, synchronised per phase): %s" % {k_: round(v_, 1) for k_, v_ in wl.phase_ms(1).items()})
from oracle import cref
cref.build()
log("verify: %s" % wl.verify(cref))
print(json.dumps({"ms_per_proof": dt / steps * 1e3, "phases": wl.phase_ms(1), "memory_gb": wl.memory_gb, "structure_ms": wl.structure_ms, "keygen_ms": wl.keygen_ms}))
