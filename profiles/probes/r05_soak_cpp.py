"""Round-5 soak of the COMPILED connected prover at full size (evidence, not a unit test): tests/cpp/prove_connected proves N times in a
row at config c2 (2048-bit n, k = 17; three messages' randomness cycled, next witness pipelined on a second context) and EVERY proof in
its output file is checked as the verifier would.  Usage: python profiles/probes/r05_soak_cpp.py [proofs, default 300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import random
import bench
import bench_connected
from oracle import cref
from paillier_halo2_amd import circuit_structure as CS
from paillier_halo2_amd import consts, prover_job

cref.build()
proofs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bits, k = 2048, 17
lb = k - 1
nn, g, m, r = bench.synth_inputs(bits, 0x50AD)
sa = CS.stream_structure("encrypt", bits, 64, lb, m, nn, device="cuda")
st, starts = CS.columns(sa, k, lb, device="cuda", keep_on_device=True)
ng, nr = sa.n_steps_g, sa.n_steps_r
del sa
rng = random.Random(0x50AD)
msgs = [(m, r)] + [(m, rng.randrange(1, nn)) for _ in range(4)]
s_tox = rng.randrange(2, consts.FR_R)
job, proof = "/tmp/pz_soak_job.bin", "/tmp/pz_soak_proof.bin"
prover_job.write_job(job, st, starts, bits, 0, ng, nr, nn, g, msgs, s_tox, seed=0x50AD, proofs=proofs)
import torch
st.selectors, st.map_col, st.map_row = (x.cpu() for x in (st.selectors, st.map_col, st.map_row))     # the device is the binary's now
torch.cuda.empty_cache()
print("[soak] job written (%.1f GB); running %d proofs" % (os.path.getsize(job) / 1e9, proofs), flush=True)
t0 = time.time()
line = prover_job.run(job, proof, timeout=1000)
print("[soak] binary: %s (%.0f s wall)" % (line, time.time() - t0), flush=True)
rec = prover_job.read_proofs(proof)
bad = 0
L = lambda x: sum(int(v) << (64 * i) for i, v in enumerate(x))
for pi in range(proofs):
    out = bench_connected.verify_file_proof(cref, rec, "p%d/" % pi, st, k, s_tox)
    mm, rr = msgs[pi % len(msgs)]
    ok = out["verified"] and out["ciphertext"] == pow(g, mm, nn * nn) * pow(rr, nn, nn * nn) % (nn * nn)
    bad += not ok
    if not ok:
        print("[soak] proof %d NOT verified: %s" % (pi, out), flush=True)
    if (pi + 1) % 50 == 0:
        print("[soak] checked %d proofs, %d failed" % (pi + 1, bad), flush=True)
print("[soak] compiled prover at c2: %d of %d proofs verified; mean %.1f ms per proof, best %.1f" % (proofs - bad, proofs, line["mean_proof_ms"], line["best_proof_ms"]), flush=True)
os.remove(job); os.remove(proof)
sys.exit(1 if bad else 0)
