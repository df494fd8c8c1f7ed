// fp29_probe.cuh -- MEASUREMENT PROBE, not used by any product kernel: the Fq Montgomery product in a
// reduced radix where v_mad_u64_u32's 64-bit addend absorbs every carry.
//   9 limbs x 29 bits (261 bits, R = 2^261 > 4p); a column of the product scan sums at most 18 products
//   below 2^58 plus a carry-in below 2^35: < 2^62.2, so NO mad can overflow its 64-bit accumulator and
//   there is no v_addc at all -- at the price of 81 + 81 + 9 = 171 multiplier instructions instead of
//   64 + 64 + 8 = 136, and one 64-bit shift + mask per column.
// Timed by pz_ubench_fqmul_variant(variant = 2); checked against Python ints by pz_fq_mul29 (tests/).
#pragma once
#include "fp.cuh"

struct Fq29 {
    u32 v[9];
};

// p in 29-bit limbs and -p^-1 mod 2^29
__device__ __forceinline__ constexpr u32 fq29_p(int i) {
    constexpr u32 p[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                          0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    return p[i];
}
static constexpr u32 FQ29_INV = 0x04866389u;   // (-p^-1 mod 2^32) & (2^29 - 1)
static constexpr u32 FQ29_MASK = 0x1fffffffu;

__device__ __forceinline__ Fq29 fq29_from_words(const u32 w[8]) {   // 256-bit little-endian -> 9 x 29
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        u64 two = (u64)w[wi] | ((u64)(wi + 1 < 8 ? w[wi + 1] : 0u) << 32);
        r.v[i] = (u32)(two >> sh) & FQ29_MASK;
    }
    return r;
}
__device__ __forceinline__ void fq29_to_words(const Fq29& a, u32 w[8]) {   // value < 2^256
    u64 acc = 0;
    int have = 0, wi = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        acc |= (u64)a.v[i] << have;
        have += 29;
        if (have >= 32) {
            if (wi < 8) w[wi++] = (u32)acc;
            acc >>= 32;
            have -= 32;
        }
    }
    if (wi < 8) w[wi] = (u32)acc;
}

// a * b * 2^-261 mod p, inputs with limbs < 2^29 and value < 2p; result < 2p, limbs < 2^29
__device__ __forceinline__ Fq29 fq29_mul(const Fq29& a, const Fq29& b) {
    u64 acc = 0;
    u32 m[9];
    Fq29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (u64)m[i] * fq29_p(k - i);
        m[k] = ((u32)acc * FQ29_INV) & FQ29_MASK;
        acc += (u64)m[k] * fq29_p(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            acc += (u64)m[i] * fq29_p(k - i);
        }
        r.v[k - 9] = (u32)acc & FQ29_MASK;
        acc >>= 29;
    }
    return r;
}
