"""times the compiled connected prover (tests/cpp/prove_connected) at a given shape: python cpp_connected_probe.py BITS K [PROOFS] [CIRCUIT]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paillier_halo2_amd import circuit_structure as CS
from paillier_halo2_amd import prover_job

bits, k = int(sys.argv[1]), int(sys.argv[2])
proofs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
circuit = sys.argv[4] if len(sys.argv) > 4 else "encrypt"
lb = k - 1
nn, g, m, r = bench.synth_inputs(bits, 0x70)
t0 = time.perf_counter()
sa = CS.stream_structure(circuit, bits, 64, lb, m, nn)
st, starts = CS.columns(sa, k, lb)
t1 = time.perf_counter()
job, proof = "/tmp/pz_job.bin", "/tmp/pz_proof.bin"
prover_job.write_job(job, st, starts, bits, 2 if circuit == "encrypt_uniform" else 0, sa.n_steps_g, sa.n_steps_r, nn, g, [(m, r)], 0x1234567, seed=3,
                     proofs=proofs)
t2 = time.perf_counter()
line = prover_job.run(job, proof, timeout=900)
t3 = time.perf_counter()
line.update(structure_s=t1 - t0, job_write_s=t2 - t1, job_bytes=os.path.getsize(job), binary_wall_s=t3 - t2)
print(json.dumps(line))
os.remove(job)
os.remove(proof)
