"""one window-range share (windows lo..hi of 16) of a single 2^22-point MSM, five times: per-kernel latencies"""
import sys
import torch
sys.path.insert(0, ".")
import paillier_halo2_amd as pz
log_n = 22
lo, hi = int(sys.argv[1]), int(sys.argv[2])
n = 1 << log_n
eng = pz.Engine(0)
eng.bind_torch_stream()
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
def rand_fr(count):
    x = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
    x[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return x
ks = rand_fr(n)
d_b = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
eng.g1_fixed_base_mul_dev(ks.data_ptr(), n, d_b.data_ptr())
eng.sync()
tb = eng.load_bases_dev(d_b.data_ptr(), n)
d_s = rand_fr(n)
out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
for _ in range(5):
    eng.msm_dev(tb, d_s.data_ptr(), 1, n, 4 * n, out.data_ptr(), lo, hi)
torch.cuda.synchronize()
