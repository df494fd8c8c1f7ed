"""probe: cost of the prover-tail kernels at c2 sizes (k = 17, extended 2^19)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import paillier_halo2_amd as pz
from paillier_halo2_amd import consts

eng = pz.Engine(0); eng.bind_torch_stream()
k, log_e = 17, 2
n, N = 1 << k, 1 << (k + log_e)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
def rand(*shape):
    x = torch.randint(-(1 << 63), (1 << 63) - 1, shape + (4,), dtype=torch.int64, device="cuda", generator=gen)
    x[..., 3] &= 0x0FFFFFFFFFFFFFFF
    return x
one = consts.fr_mont_limbs(1); g7 = consts.fr_mont_limbs(7); w = consts.fr_mont_limbs(consts.fr_omega(k)); wx = consts.fr_mont_limbs(consts.fr_omega(k + log_e))
def t(name, fn, reps=3, per=1, unit=""):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    print("%-44s %8.3f ms  (%.3f ms per %s)" % (name, dt, dt / per, unit), flush=True)
nc = 64
adv, sel, h = rand(nc, N), rand(nc, N), rand(N)
t("quotient_gate 64 cols @2^19", lambda: eng.quotient_gate_dev(adv.data_ptr(), 4 * N, sel.data_ptr(), 4 * N, nc, k + log_e, 4, g7, h.data_ptr()), per=nc, unit="column")
t("quotient_finish @2^19", lambda: eng.quotient_finish_dev(h.data_ptr(), k, log_e, g7, wx))
z = rand(32, N); l = rand(3, N)
t("quotient_permutation 64 cols/32 sets @2^19", lambda: eng.quotient_permutation_dev(adv.data_ptr(), 4 * N, sel.data_ptr(), 4 * N, z.data_ptr(), 4 * N, 32, 2, 64, k + log_e, 4, 6, l[0].data_ptr(), l[1].data_ptr(), l[2].data_ptr(), g7, g7, g7, g7, wx, g7, h.data_ptr()), per=64, unit="column")
t("quotient_permutation 4 cols/2 sets @2^19", lambda: eng.quotient_permutation_dev(adv.data_ptr(), 4 * N, sel.data_ptr(), 4 * N, z.data_ptr(), 4 * N, 2, 2, 4, k + log_e, 4, 6, l[0].data_ptr(), l[1].data_ptr(), l[2].data_ptr(), g7, g7, g7, g7, wx, g7, h.data_ptr()), per=4, unit="column")
t("quotient_lookup 8 lookups @2^19", lambda: eng.quotient_lookup_dev(adv.data_ptr(), 4 * N, sel.data_ptr(), adv[8].data_ptr(), 4 * N, adv[16].data_ptr(), 4 * N, adv[24].data_ptr(), 4 * N, 8, k + log_e, 4, l[0].data_ptr(), l[1].data_ptr(), l[2].data_ptr(), g7, g7, g7, h.data_ptr()), per=8, unit="lookup")
cols, sig, zz = rand(2, n), rand(2, n), rand(n)
t("permutation_product 2 cols @2^17", lambda: eng.permutation_product_dev(cols.data_ptr(), 4 * n, sig.data_ptr(), 4 * n, 2, k, w, g7, g7, one, g7, one, zz.data_ptr()), per=1, unit="set")
a = rand(n)
t("batch_invert 2^17", lambda: eng.fr_batch_invert_dev(a.data_ptr(), n))
a8 = rand(64 * n)
t("batch_invert 64 x 2^17", lambda: eng.fr_batch_invert_dev(a8.data_ptr(), 64 * n), per=64, unit="column")
t("prefix_product 2^17", lambda: eng.fr_prefix_product_dev(a.data_ptr(), n, one, zz.data_ptr()))
p64, q64 = rand(64, n), rand(64, n)
t("kate_division 64 cols @2^17", lambda: eng.poly_div_linear_dev(p64.data_ptr(), 64, 4 * n, n, g7, q64.data_ptr(), 4 * n), per=64, unit="column")
ev = rand(64)
t("poly_eval 64 cols @2^17", lambda: eng.poly_eval_dev(p64.data_ptr(), 64, 4 * n, n, g7, ev.data_ptr()), per=64, unit="column")
rows = n - 9   # layout.row_budget(k).max_rows
lk = torch.zeros((84, rows, 4), dtype=torch.int64, device="cuda")
lk[:, :, 0] = torch.randint(0, 1 << 16, (84, rows), dtype=torch.int64, device="cuda", generator=gen)
eng.fr_convert_dev(lk.data_ptr(), 84 * rows, True)
tab = torch.zeros((rows, 4), dtype=torch.int64, device="cuda"); tab[: 1 << 16, 0] = torch.arange(1 << 16, device="cuda")
eng.fr_convert_dev(tab.data_ptr(), rows, True)
pi, pt = torch.zeros_like(lk), torch.zeros_like(lk)
t("lookup_permute 84 cols, 16-bit, @2^17", lambda: eng.lookup_permute_dev(lk.data_ptr(), 84, 4 * rows, tab.data_ptr(), rows, 16, pi.data_ptr(), pt.data_ptr(), 4 * rows), per=84, unit="column")
zl = torch.zeros_like(lk)
t("lookup_product 84 lookups @2^17", lambda: eng.lookup_product_dev(lk.data_ptr(), 4 * rows, tab.data_ptr(), pi.data_ptr(), 4 * rows, pt.data_ptr(), 4 * rows, 84, rows, g7, g7, one, zl.data_ptr(), 4 * rows), per=84, unit="lookup")
pc, ps, pz_ = rand(256, n), rand(256, n), rand(128, n)
t("permutation_product_sets 256 cols/128 sets @2^17", lambda: eng.permutation_product_sets_dev(pc.data_ptr(), 4 * n, ps.data_ptr(), 4 * n, 256, 2, k, n - 6, w, g7, g7, g7, pz_.data_ptr(), 4 * n), per=128, unit="set")
