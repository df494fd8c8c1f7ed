"""Per-phase wave cycles of k_ntt_strided29<true> (the coset pre-scale pass) from a PROBE build of pz_ntt.hip with s_memtime
stamps (scratch/stamp: built by the recipe in DESIGN.md section 6.1); PZ_LIB names that library."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import paillier_halo2_amd._lib as L

L.SO_PATH = os.path.abspath(os.environ["PZ_LIB"])
import paillier_halo2_amd as pz
from paillier_halo2_amd import consts

eng = pz.Engine(0)
eng.bind_torch_stream()
lib = L.lib()
lib.pz_probe_stamps.restype = C.c_int
lib.pz_probe_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
k, n, nb = 17, 1 << 17, 64
gen = torch.Generator(device="cuda")
gen.manual_seed(1)
x = torch.randint(-(1 << 63), (1 << 63) - 1, (nb * n, 4), dtype=torch.int64, device="cuda", generator=gen)
x[:, 3] &= 0x0FFFFFFFFFFFFFFF
cols = x.view(nb, n, 4)
ext = torch.zeros((nb, 4 * n, 4), dtype=torch.int64, device="cuda")
n_inv = consts.fr_mont_limbs(pow(n, -1, consts.FR_R))
w_n = consts.fr_mont_limbs(consts.fr_omega(k))
w_ext = consts.fr_omega(k + 2)
gens = np.stack([consts.fr_mont_limbs(consts.FR_GENERATOR * pow(w_ext, r, consts.FR_R) % consts.FR_R) for r in range(4)])
for _ in range(2):
    eng.ntt_extend_dev(cols.data_ptr(), nb, 4 * n, ext.data_ptr(), 16 * n, k, 2, w_n, gens, n_inv)
torch.cuda.synchronize()
lib.pz_probe_stamps(None, 1)
R = 4
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(R):
    eng.ntt_extend_dev(cols.data_ptr(), nb, 4 * n, ext.data_ptr(), 16 * n, k, 2, w_n, gens, n_inv)
e1.record()
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
lib.pz_probe_stamps(out, 0)
v = [int(a) for a in out]
waves = v[15]
names = {0: "convert (unpack, pre-scale product, LDS put)", 1: "prefetch issue", 5: "barrier waits", 2: "stage LDS reads (issue + wait)",
         3: "stage compute (carries, 3-4 products, butterflies)", 4: "stage LDS writes (issue + wait)", 6: "radix-2 tail stage",
         7: "epilogue: twiddle loads issued, vmcnt(0)", 8: "epilogue: products + stores", 9: "final barrier", 11: "loop overhead"}
tot = sum(v[:12])
print("ext transform %.1f us per polynomial; waves flushed %d; cycles per wave %.0f" % (e0.elapsed_time(e1) * 1000 / (R * nb), waves, tot / max(1, waves)))
for i in sorted(names):
    print("  %-55s %6.2f %%   %10.0f cycles per wave" % (names[i], 100.0 * v[i] / tot, v[i] / max(1, waves)))
