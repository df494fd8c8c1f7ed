"""Fq products per second of every field-product variant (pz_ubench_fqmul_variant), at two occupancies"""
import sys
sys.path.insert(0, ".")
import paillier_halo2_amd as pz
eng = pz.Engine(0)
names = {0: "fp_mul 8x32 asm", 1: "fp_mul 8x32 no wait states", 2: "9x29 plain C (probe)", 3: "f29_mul asm columns", 4: "f29_sqr asm columns"}
for blocks, iters in ((256 * 16, 256), (256 * 4, 1024)):
    out = {}
    for v, nm in names.items():
        ms = min(eng.ubench_fqmul_variant(v, blocks, iters) for _ in range(3))
        out[nm] = round(blocks * 256 * iters * 2 / (ms * 1e-3) / 1e9, 1)
    print("blocks=%d iters=%d:" % (blocks, iters), out)
