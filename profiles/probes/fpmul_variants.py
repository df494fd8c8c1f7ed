"""Fq products per second of every field-product variant (libpz_probe.so: pzp_ubench_fqmul_variant), at two occupancies"""
import sys
sys.path.insert(0, ".")
import paillier_halo2_amd as pz
from paillier_halo2_amd import probe
eng = pz.Engine(0)
names = {0: "fp_mul 8x32 asm", 1: "fp_mul 8x32 no wait states", 2: "9x29 plain C (probe)", 3: "f29_mul asm columns", 4: "f29_sqr asm columns"}
for blocks, iters in ((256 * 16, 256), (256 * 4, 1024)):
    out = {}
    for v, nm in names.items():
        ms = min(probe.ubench_fqmul_variant(0, v, blocks, iters) for _ in range(3))
        out[nm] = round(blocks * 256 * iters * 2 / (ms * 1e-3) / 1e9, 1)
    print("blocks=%d iters=%d:" % (blocks, iters), out)

for blocks, iters in ((2048, 4096), (8192, 1024)):
    ms = min(probe.ubench_mad(0, blocks, iters) for _ in range(3))
    ms2 = min(probe.ubench_mad_indep(0, blocks, iters) for _ in range(3))
    print("v_mad_u64_u32 issue rate, blocks=%d iters=%d: dependent multiplicand %.1f T/s, independent %.1f T/s" % (
        blocks, iters, blocks * 256 * iters * 8 / ms / 1e9, blocks * 256 * iters * 8 / ms2 / 1e9))
