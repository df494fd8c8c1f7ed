"""Stand-alone timing of the two multiplier-bound kernel families at the c2 shapes (one stream, nothing else on
the GPU): 64 polynomials x (iNTT 2^17 + coset extension to 2^19), and 256 full-width / 256 witness-like column MSMs at
2^17.  Prints ms per column.  Usage: python profiles/probes/kernel_times.py [reps]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import paillier_halo2_amd as pz
from paillier_halo2_amd import consts

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
eng = pz.Engine(0)
eng.bind_torch_stream()
k, n = 17, 1 << 17
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
def rand_fr(count):
    x = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device="cuda", generator=gen)
    x[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return x
nb = 64
cols = rand_fr(nb * n).view(nb, n, 4)
ext = torch.zeros((nb, 4 * n, 4), dtype=torch.int64, device="cuda")
w_inv = consts.fr_mont_limbs(pow(consts.fr_omega(k), -1, consts.FR_R))
n_inv = consts.fr_mont_limbs(pow(n, -1, consts.FR_R))
w_n = consts.fr_mont_limbs(consts.fr_omega(k))
w_ext = consts.fr_omega(k + 2)
gens = np.stack([consts.fr_mont_limbs(consts.FR_GENERATOR * pow(w_ext, r, consts.FR_R) % consts.FR_R) for r in range(4)])
def ntt_once():
    eng.ntt_dev(cols.data_ptr(), nb, 4 * n, w_inv, k, None, None)
    eng.ntt_extend_dev(cols.data_ptr(), nb, 4 * n, ext.data_ptr(), 16 * n, k, 2, w_n, gens, n_inv)
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
t_i = timeit(lambda: eng.ntt_dev(cols.data_ptr(), nb, 4 * n, w_inv, k, None, None), reps)
t_e = timeit(lambda: eng.ntt_extend_dev(cols.data_ptr(), nb, 4 * n, ext.data_ptr(), 16 * n, k, 2, w_n, gens, n_inv), reps)
print("NTT: iNTT 2^17 %.4f ms/col, extend->2^19 %.4f ms/col, both %.4f ms/col (x4930 = %.1f ms/proof)" % (t_i / nb, t_e / nb, (t_i + t_e) / nb, (t_i + t_e) / nb * 4930))
# MSM
d_b = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
eng.srs_setup_g1_dev(k, consts.fr_mont_limbs(123456789), consts.fr_mont_limbs(consts.fr_omega(k)), 0, d_b.data_ptr())
eng.sync()
bases = eng.load_bases_dev(d_b.data_ptr(), n)
nc = 256
full = rand_fr(nc * n).view(nc, n, 4)
out = torch.zeros((nc, 12), dtype=torch.int64, device="cuda")
t_f = timeit(lambda: eng.msm_dev(bases, full.data_ptr(), nc, n, 4 * n, out.data_ptr()), reps)
wit = full.clone()
u = torch.rand(nc * n, device="cuda", generator=gen).view(nc, n)
wit[:, :, 3] = 0
wit[:, :, 2] = torch.where(u >= 0.9, wit[:, :, 2] & 0x7F, torch.zeros_like(wit[:, :, 2]))
wit[:, :, 1] = torch.where(u >= 0.9, wit[:, :, 1], torch.zeros_like(wit[:, :, 1]))
wit[:, :, 0] = torch.where(u < 0.6, wit[:, :, 0] & 0xFFFF, wit[:, :, 0])
eng.fr_convert_dev(wit.data_ptr(), nc * n, True)
t_w = timeit(lambda: eng.msm_dev(bases, wit.data_ptr(), nc, n, 4 * n, out.data_ptr()), reps)
eng.timing_enable(True); eng.timing_reset()
eng.msm_dev(bases, full.data_ptr(), nc, n, 4 * n, out.data_ptr()); eng.sync()
acc_ms, _ = eng.timing_get(0); all_ms, _ = eng.timing_get(4)
print("MSM 2^17: full-width %.4f ms/col (accumulate %.4f, rest %.4f), witness-like mix %.4f ms/col" % (t_f / nc, acc_ms / nc, (all_ms - acc_ms) / nc, t_w / nc))
