"""256 full-width + 512 witness-like 2^17 column MSMs, twice: the workload rocprofv3 is pointed at"""
import sys
import torch
sys.path.insert(0, ".")
import paillier_halo2_amd as pz
from paillier_halo2_amd import consts
which = sys.argv[1] if len(sys.argv) > 1 else "both"
eng = pz.Engine(0)
eng.bind_torch_stream()
k, n = 17, 1 << 17
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
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
for _ in range(2):
    if which in ("both", "full"):
        eng.msm_dev(bases, full.data_ptr(), nc, n, 4 * n, out.data_ptr())
    if which in ("both", "wit"):
        eng.msm_dev(bases, wit.data_ptr(), 2 * nc, n, 4 * n, out.data_ptr())
torch.cuda.synchronize()
