"""The library's keygen / create_proof stepper (pz_pk_* / pz_proof_*) at a given shape, driven through ctypes with a hashing transcript;
the proof checked as the verifier would.  Usage: python profiles/probes/native_stepper_probe.py [enc_bits k proofs]  (default 2048 17 4)"""
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import paillier_halo2_amd as pz
import bench_connected
from paillier_halo2_amd import prover, prover_native

bits, k, proofs = (int(x) for x in (sys.argv[1:4] + ["2048", "17", "4"][len(sys.argv) - 1:]))
eng = pz.Engine(0)
eng.bind_torch_stream()
log = lambda s: print("[probe] " + s, flush=True)
wl = bench_connected.ConnectedWorkload(eng, torch, bits, k, 0x5043, log=lambda s: None, pipeline=False)
# the Python key leaves the device: the library builds its own from the same structure
for nm in ("pk", "ws"):
    setattr(wl, nm, None)
gc.collect(); torch.cuda.empty_cache()
t0 = time.perf_counter()
key = prover_native.NativeKey(eng, wl.cs, wl.bl, wl.bm)
log("pz_pk_create %.0f ms; sets %d, blinding words %d, evaluation words %d" % ((time.perf_counter() - t0) * 1e3, key.n_sets, key.blinding_words, key.evals_words))
times = []
for i in range(proofs):
    eng.sync()
    t1 = time.perf_counter()
    wl.produce(eng)                                   # K3 + K4 into the witness slot
    tr = prover.HashTranscript(b"native-%d" % i)
    pr = prover_native.create_proof(key, wl.cols.data_ptr(), tr, seed=100 + i)
    eng.sync()
    times.append((time.perf_counter() - t1) * 1e3)
log("%d proofs: %s ms (incl. K3 + K4, one context, no overlap)" % (proofs, [round(t, 1) for t in times]))
from oracle import cref
cref.build()

class _K:     # the checker reads the key's commitments through the same interface as the Python key's
    pass
wl.last = (pr, tr.challenges(), (wl.produced - 1) % len(wl.variants))
wl.pk = _K(); wl.pk.vk_commitments = key.vk_commitments; wl.pk.n_sets = key.n_sets; wl.pk.dom = prover.Domain(k, wl.cs.blinding_factors)
ver = wl.verify(cref)
log("verify: %s" % ver)
print(json.dumps({"pz_proof_ms": times, "verified": ver["verified"]}))
key.free()
