"""Round 6, VERDICT r05 item 7: ONE bounded probe of the idle matrix pipes -- the constant products of a field multiplication on
v_mfma_i32_16x16x64_i8 (csrc/probe/pz_probe_mfma.hip) against f29_mulc on the VALU.  Prints: bit-exactness of the whole MFMA pipeline
against Python integers, then products per second of (0) f29_mulc, (1) the matrix side alone, (2) the VALU side of the MFMA formulation
alone, (3) the whole pipeline; and the keep / drop decision."""
import random
import sys

import numpy as np

sys.path.insert(0, ".")
from paillier_halo2_amd import consts, probe

R = consts.FR_R
dev = 0
rng = random.Random(0x6d666d61)
limbs = lambda v: [(v >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]
unl = lambda row: sum(int(x) << (29 * i) for i, x in enumerate(row))
bad = 0
for trial in range(6):
    w = rng.randrange(1, R) if trial else R - 1
    vals = [rng.randrange(0, 1 << 261) for _ in range(4093)] + [0, 1, (1 << 261) - 1]
    a = np.array([limbs(v) for v in vals], dtype=np.uint32)
    out, _ = probe.mulc_mfma(dev, a, w, R)
    got = [unl(r) for r in out]
    want = [v * w % R for v in vals]
    bad += sum(1 for g_, w_ in zip(got, want) if g_ != w_)
print("bit-exact vs Python integers: %d mismatches in %d products (6 constants, values below 2^261 incl. 0, 1, 2^261 - 1)" % (bad, 6 * 4096))
# rates
blocks, iters = 8192, 256
t0 = min(probe.ubench_fqmul_variant(dev, 5, blocks, iters) for _ in range(3))
r_mulc = blocks * 256 * iters * 2 / (t0 * 1e-3)
t1 = min(probe.ubench_mfma(dev, 1, blocks, iters) for _ in range(3))
r_mat = blocks * 4 * iters * 16 / (t1 * 1e-3)
t2 = min(probe.ubench_mfma(dev, 2, blocks, 64) for _ in range(3))
r_valu = blocks * 256 * 64 / (t2 * 1e-3)
cnt = 1 << 20
a = np.array([limbs(rng.randrange(0, 1 << 254)) for _ in range(4096)], dtype=np.uint32)
a = np.tile(a, (cnt // 4096, 1))
it3 = 16
_, t3 = probe.mulc_mfma(dev, a, rng.randrange(1, R), R, iters=it3)
r_pipe = cnt * it3 / (t3 * 1e-3)
print("(0) f29_mulc on the VALU (143 v_mad_u64_u32 per product)            %8.1f G products/s" % (r_mulc / 1e9))
print("(1) matrix side alone: 8 x v_mfma_i32_16x16x64_i8 per 16 products    %8.1f G products/s  (%.2f P byte-MAC/s)" % (r_mat / 1e9, r_mat * 8192 / 1e15))
print("(2) VALU side of the MFMA formulation alone (digits, carries, repack) %8.1f G products/s" % (r_valu / 1e9))
print("(3) whole MFMA pipeline, correct, LDS-staged (2^20 products x %d)     %8.1f G products/s" % (it3, r_pipe / 1e9))
best = min(r_mat, r_valu)
print("upper bound of any arrangement = min((1), (2)) = %.1f G/s = %.2f x f29_mulc -> %s" % (
    best / 1e9, best / r_mulc, "KEEP: build it into K2" if best > 1.3 * r_mulc else "DROP: the digit split and the carry recombination the byte formulation needs "
    "cost the VALU more than the whole 29-bit product they would replace; the matrix pipes stay idle by arithmetic, not by omission"))
