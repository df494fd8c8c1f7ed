"""K3 alone on the GPU: wall time of one 2048-bit (and 3072-bit) Paillier encrypt trace, per dependent mul_mod step.
Usage: python profiles/probes/k3_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import paillier_halo2_amd as pz
from paillier_halo2_amd import consts
import bench

eng = pz.Engine(0)
eng.bind_torch_stream()
for bits in (2048, 3072):
    nn, g, m, r = bench.synth_inputs(bits, 0x5043)
    Ln = bits // 64
    arr = lambda x: consts.int_to_limbs(x, Ln)
    steps = m.bit_length() + bin(m).count("1") + nn.bit_length() + bin(nn).count("1") + 1
    d_steps = torch.zeros((steps, 4, 2 * Ln), dtype=torch.int64, device="cuda")
    best = None
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.paillier_encrypt_dev(Ln, arr(nn), arr(g), arr(m), arr(r), d_steps.data_ptr(), steps)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    # the two chains run side by side: the longer one has bits(n) + popcount(n) steps, of which bits(n) are dependent squarings
    dep = max(nn.bit_length(), m.bit_length())
    print("encrypt %d-bit: %.2f ms for %d steps; %.2f us per dependent step (%d exponent bits + setup)" % (bits, best * 1e3, steps, best * 1e6 / dep, dep), flush=True)
