"""Device-resident pz_msm_g1_dev: ms per column against the number of columns per call (2^17 points, uniform scalars), with
the kernel-class split of the library timers."""
import os
import sys

import torch

sys.path.insert(0, ".")
import paillier_halo2_amd as pz
from paillier_halo2_amd import consts

eng = pz.Engine(0)
eng.bind_torch_stream()
k, n = 17, 1 << 17
gen = torch.Generator(device="cuda")
gen.manual_seed(1)
x = torch.randint(-(1 << 63), (1 << 63) - 1, (256 * n, 4), dtype=torch.int64, device="cuda", generator=gen)
x[:, 3] &= 0x0FFFFFFFFFFFFFFF
d_b = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
eng.srs_setup_g1_dev(k, consts.fr_mont_limbs(123456789), consts.fr_mont_limbs(consts.fr_omega(k)), 0, d_b.data_ptr())
eng.sync()
bases = eng.load_bases_dev(d_b.data_ptr(), n)
out = torch.zeros((256, 12), dtype=torch.int64, device="cuda")
for nc in (1, 4, 8, 9, 16, 32, 64, 128, 256):
    eng.msm_dev(bases, x.data_ptr(), nc, n, 4 * n, out.data_ptr())
    eng.sync()
    eng.timing_enable(True)
    eng.timing_reset()
    R = 3
    for _ in range(R):
        eng.msm_dev(bases, x.data_ptr(), nc, n, 4 * n, out.data_ptr())
    eng.sync()
    t = {c: eng.timing_get(getattr(pz.engine, c))[0] / R for c in ("T_MSM_SORT", "T_MSM_ACC", "T_MSM_TREE", "T_MSM_ALL")}
    eng.timing_enable(False)
    print("%4d columns: %.3f ms per column   (sort %.2f  accumulate %.2f  tree %.2f  all %.2f ms)" % (
        nc, t["T_MSM_ALL"] / nc, t["T_MSM_SORT"], t["T_MSM_ACC"], t["T_MSM_TREE"], t["T_MSM_ALL"]))
