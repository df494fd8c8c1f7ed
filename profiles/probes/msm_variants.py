"""Timing of the K1 kernel classes for a library variant (PZ_LIB): 256 full-width 2^17-point columns and 512 witness-like
ones, per launch sequence: sort / accumulate / tree / all, in ms."""
import os
import sys

import torch

sys.path.insert(0, ".")
import paillier_halo2_amd._lib as L

if os.environ.get("PZ_LIB"):
    L.SO_PATH = os.path.abspath(os.environ["PZ_LIB"])
import paillier_halo2_amd as pz
from paillier_halo2_amd import consts

eng = pz.Engine(0)
eng.bind_torch_stream()
k, n = 17, 1 << 17
gen = torch.Generator(device="cuda")
gen.manual_seed(1)


def rand_fr(count):
    x = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
    x[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return x


d_b = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
eng.srs_setup_g1_dev(k, consts.fr_mont_limbs(123456789), consts.fr_mont_limbs(consts.fr_omega(k)), 0, d_b.data_ptr())
eng.sync()
bases = eng.load_bases_dev(d_b.data_ptr(), n)
nc = 256
full = rand_fr(nc * n).view(nc, n, 4)
out = torch.zeros((2 * nc, 12), dtype=torch.int64, device="cuda")
wit = rand_fr(2 * nc * n).view(2 * nc, n, 4)
u = torch.rand(2 * nc * n, device="cuda", generator=gen).view(2 * nc, n)
wit[:, :, 3] = 0
wit[:, :, 2] = torch.where(u >= 0.9, wit[:, :, 2] & 0x7F, torch.zeros_like(wit[:, :, 2]))
wit[:, :, 1] = torch.where(u >= 0.9, wit[:, :, 1], torch.zeros_like(wit[:, :, 1]))
wit[:, :, 0] = torch.where(u < 0.6, wit[:, :, 0] & 0xFFFF, wit[:, :, 0])
eng.fr_convert_dev(wit.data_ptr(), 2 * nc * n, True)
res = {}
for name, buf, cols in (("full", full, nc), ("witness", wit, 2 * nc)):
    eng.msm_dev(bases, buf.data_ptr(), cols, n, 4 * n, out.data_ptr())
    eng.sync()
    eng.timing_enable(True)
    eng.timing_reset()
    R = 3
    for _ in range(R):
        eng.msm_dev(bases, buf.data_ptr(), cols, n, 4 * n, out.data_ptr())
    eng.sync()
    t = {c: eng.timing_get(getattr(pz.engine, c))[0] / R for c in ("T_MSM_SORT", "T_MSM_ACC", "T_MSM_TREE", "T_MSM_ALL")}
    eng.timing_enable(False)
    res[name] = t
    chk = int(out[:cols].sum().item()) & 0xFFFFFFFF
    print("%-34s %-8s sort %.2f  accumulate %.2f  tree %.2f  all %.2f ms   (checksum %08x)" % (
        os.environ.get("PZ_LIB", "default")[-34:], name, t["T_MSM_SORT"], t["T_MSM_ACC"], t["T_MSM_TREE"], t["T_MSM_ALL"], chk))
