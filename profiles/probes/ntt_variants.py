"""Timing experiments on the K2 kernels: loads the library named by PZ_LIB (a build of csrc with -DPZ_NTT_EXP=n, see
pz_ntt.hip; results are WRONG under any experiment, only the time means something) and prints us per polynomial for
64 polynomials x (iNTT 2^17 + coset extension to 2^19)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import paillier_halo2_amd._lib as L

if os.environ.get("PZ_LIB"):
    L.SO_PATH = os.path.abspath(os.environ["PZ_LIB"])
import paillier_halo2_amd as pz
from paillier_halo2_amd import consts

eng = pz.Engine(0)
eng.bind_torch_stream()
k, n, nb = 17, 1 << 17, 64
gen = torch.Generator(device="cuda")
gen.manual_seed(1)
x = torch.randint(-(1 << 63), (1 << 63) - 1, (nb * n, 4), dtype=torch.int64, device="cuda", generator=gen)
x[:, 3] &= 0x0FFFFFFFFFFFFFFF
cols = x.view(nb, n, 4)
ext = torch.zeros((nb, 4 * n, 4), dtype=torch.int64, device="cuda")
w_inv = consts.fr_mont_limbs(pow(consts.fr_omega(k), -1, consts.FR_R))
n_inv = consts.fr_mont_limbs(pow(n, -1, consts.FR_R))
w_n = consts.fr_mont_limbs(consts.fr_omega(k))
w_ext = consts.fr_omega(k + 2)
gens = np.stack([consts.fr_mont_limbs(consts.FR_GENERATOR * pow(w_ext, r, consts.FR_R) % consts.FR_R) for r in range(4)])


def run(rounds):
    for _ in range(rounds):
        eng.ntt_dev(cols.data_ptr(), nb, 4 * n, w_inv, k, None, None)
        eng.ntt_extend_dev(cols.data_ptr(), nb, 4 * n, ext.data_ptr(), 16 * n, k, 2, w_n, gens, n_inv)


run(2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 8
e0.record()
run(R)
e1.record()
torch.cuda.synchronize()
print("%-28s %.1f us per polynomial" % (os.environ.get("PZ_LIB", "default"), e0.elapsed_time(e1) * 1000 / (R * nb)))
