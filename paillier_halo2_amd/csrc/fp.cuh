// fp.cuh -- BN254 base field Fq and scalar field Fr on gfx950, 8 x 32-bit limbs, Montgomery form
// with R = 2^256 (bit-identical to the 4 x u64 little-endian layout of halo2curves Fr/Fq that the
// C ABI in include/pz.h takes verbatim).
//
// Integer modular arithmetic: no MFMA.  The multiplier primitive of CDNA4's VALU is
// v_mad_u64_u32 (32x32+64 -> 64); everything here is written so hipcc lowers the inner products
// to that instruction with the carries kept in 64-bit accumulators.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

struct FqTag {};
struct FrTag {};

template <class Tag> struct FieldParams;

template <> struct FieldParams<FqTag> {
    static constexpr u32 INV = 0xe4866389u;  // -p^-1 mod 2^32
    __device__ __forceinline__ static constexpr u32 P(int i) {
        constexpr u32 p[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                              0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return p[i];
    }
    __device__ __forceinline__ static constexpr u32 P2(int i) {  // 2p: the lazy range bound (see below)
        constexpr u32 p[8] = {0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u,
                              0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};
        return p[i];
    }
    __device__ __forceinline__ static constexpr u32 R1(int i) {  // R mod p
        constexpr u32 r[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                              0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return r[i];
    }
    __device__ __forceinline__ static constexpr u32 R2(int i) {  // R^2 mod p
        constexpr u32 r[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                              0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return r[i];
    }
};

template <> struct FieldParams<FrTag> {
    static constexpr u32 INV = 0xefffffffu;
    __device__ __forceinline__ static constexpr u32 P(int i) {
        constexpr u32 p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                              0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return p[i];
    }
    __device__ __forceinline__ static constexpr u32 P2(int i) {  // 2p: the lazy range bound (see below)
        constexpr u32 p[8] = {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u,
                              0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};
        return p[i];
    }
    __device__ __forceinline__ static constexpr u32 R1(int i) {
        constexpr u32 r[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                              0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return r[i];
    }
    __device__ __forceinline__ static constexpr u32 R2(int i) {
        constexpr u32 r[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                              0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return r[i];
    }
};

template <class Tag> struct alignas(16) Fp {
    u32 v[8];
};
typedef Fp<FqTag> Fq;
typedef Fp<FrTag> Fr;

template <class T> __device__ __forceinline__ Fp<T> fp_zero() {
    Fp<T> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r;
}
template <class T> __device__ __forceinline__ Fp<T> fp_one() {  // Montgomery one
    Fp<T> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = FieldParams<T>::R1(i);
    return r;
}
// LAZY RANGE: inside the kernels a field element is any representative in [0, 2p).  R = 2^256 > 4p, so the Montgomery
// product of two such values is again below 2p WITHOUT the final conditional subtraction ((4p^2 + Rp)/R < 1.76p):
// fp_mul skips it (+5.6 % multiplications per second), fp_add / fp_sub fold to [0, 2p) with the same instruction count
// as the canonical versions, and memory only ever holds canonical values: fp_store subtracts p once if needed, so
// loads are canonical and nothing lazy crosses the ABI.  Zero tests on COMPUTED differences must accept both
// representatives of zero (0 and p): fp_is_zero.  fp_is_zero_exact is for values that are zero only when they were
// set to zero (loaded coordinates, the ZZ of the identity): a product of non-zero elements is never 0 or p.
template <class T> __device__ __forceinline__ bool fp_is_zero_exact(const Fp<T>& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i];
    return o == 0;
}
template <class T> __device__ __forceinline__ bool fp_is_zero(const Fp<T>& a) {
    u32 o = 0, q = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        o |= a.v[i];
        q |= a.v[i] ^ FieldParams<T>::P(i);
    }
    return o == 0 || q == 0;
}

// 256-bit add / subtract as 32-bit carry chains.  Written with __builtin_addc / __builtin_subc so hipcc emits
// v_add_co_u32 + 7 x v_addc_co_u32 (about 2.5 cycles each); the same arithmetic written on u64 lowers to
// v_lshl_add_u64, which issues at ~9.5 cycles on gfx950 (measured, scratch ubench) and made every modular
// add / subtract cost a fifth of a multiplication.
template <class T> __device__ __forceinline__ u32 fp_sub_p(u32 t[8], const Fp<T>& a) {  // t = a - p, returns borrow
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 co;
        t[i] = __builtin_subc(a.v[i], FieldParams<T>::P(i), c, &co);
        c = co;
    }
    return c;
}

// r = a - p if a >= p (a < 2p assumed)
template <class T> __device__ __forceinline__ void fp_reduce_once(Fp<T>& a) {
    u32 t[8];
    const u32 br = fp_sub_p(t, a);
#pragma unroll
    for (int i = 0; i < 8; ++i) a.v[i] = br ? a.v[i] : t[i];
}

// r = a - 2p if a >= 2p (a < 4p assumed)
template <class T> __device__ __forceinline__ void fp_reduce_2p(Fp<T>& a) {
    u32 t[8];
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 co;
        t[i] = __builtin_subc(a.v[i], FieldParams<T>::P2(i), c, &co);
        c = co;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) a.v[i] = c ? a.v[i] : t[i];
}
// the canonical representative of a lazy value
template <class T> __device__ __forceinline__ Fp<T> fp_canon(const Fp<T>& a) {
    Fp<T> r = a;
    fp_reduce_once(r);
    return r;
}
template <class T> __device__ __forceinline__ bool fp_eq(const Fp<T>& a, const Fp<T>& b) {
    const Fp<T> x = fp_canon(a), y = fp_canon(b);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= x.v[i] ^ y.v[i];
    return o == 0;
}

template <class T> __device__ __forceinline__ Fp<T> fp_add(const Fp<T>& a, const Fp<T>& b) {
    Fp<T> r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 co;
        r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &co);
        c = co;
    }
    // a, b < 2p and 4p < 2^256: no carry out of limb 7; back to [0, 2p)
    fp_reduce_2p(r);
    return r;
}

template <class T> __device__ __forceinline__ Fp<T> fp_sub(const Fp<T>& a, const Fp<T>& b) {
    Fp<T> r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 co;
        r.v[i] = __builtin_subc(a.v[i], b.v[i], c, &co);
        c = co;
    }
    const u32 mask = (u32)0 - c;  // borrow -> add 2p back: a - b + 2p lies in (0, 2p)
    u32 c2 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 co;
        r.v[i] = __builtin_addc(r.v[i], FieldParams<T>::P2(i) & mask, c2, &co);
        c2 = co;
    }
    return r;
}

template <class T> __device__ __forceinline__ Fp<T> fp_neg(const Fp<T>& a) {
    if (fp_is_zero_exact(a)) return a;  // (a == p stays p: also a representative of zero)
    Fp<T> r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 co;
        r.v[i] = __builtin_subc(FieldParams<T>::P2(i), a.v[i], c, &co);
        c = co;
    }
    return r;  // in (0, 2p)
}

template <class T> __device__ __forceinline__ Fp<T> fp_dbl(const Fp<T>& a) { return fp_add(a, a); }

// Montgomery product a*b*R^-1 mod p: finely integrated product scanning over 32-bit limbs with one 96-bit
// column accumulator.  Per term one v_mad_u64_u32 (the quarter-rate multiplier primitive; its carry-out
// goes to VCC) and one v_addc folding the carry into the third accumulator word: 128 mads + 8 v_mul_lo
// + 128 addc.  hipcc does not use the mad's carry-out from C (it re-derives carries with
// v_cmp_lt_u64 + v_cndmask, or keeps 64-bit adds + zero-extension moves: 580 instructions), hence inline
// asm, one statement per column (gen_fp_mul.py).  VALU->VALU register dependencies inside a statement are
// hardware-interlocked on gfx950; a carry-out (an SGPR pair or VCC written by a VALU) is NOT: it needs two wait
// states before a VALU reads it, which the generator provides by software-pipelining the carries (and an
// explicit s_nop 1 where a block has a single product).  Cost model measured on MI355X: DESIGN.md section 5.
#ifndef PZ_FP_MUL_PLAIN
#include "fp_mul_gen.cuh"
#else
// plain-C CIOS variant (what hipcc makes of it: 128 mads + ~450 moves / 64-bit adds); kept for A/B debugging
template <class T> __device__ __forceinline__ Fp<T> fp_mul(const Fp<T>& a, const Fp<T>& b) {
    u32 t[8];
    u32 t8 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 c = 0;
        const u32 bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c = (u64)a.v[j] * bi + t[j] + c;
            t[j] = (u32)c;
            c >>= 32;
        }
        c += t8;
        t8 = (u32)c;
        u32 t9 = (u32)(c >> 32);
        const u32 m = t[0] * FieldParams<T>::INV;
        c = (u64)m * FieldParams<T>::P(0) + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c = (u64)m * FieldParams<T>::P(j) + t[j] + c;
            t[j - 1] = (u32)c;
            c >>= 32;
        }
        c += t8;
        t[7] = (u32)c;
        t8 = t9 + (u32)(c >> 32);
    }
    Fp<T> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    fp_reduce_once(r);
    return r;
}
#endif

template <class T> __device__ __forceinline__ Fp<T> fp_sqr(const Fp<T>& a) { return fp_mul(a, a); }

#ifdef PZ_FP_MUL_PLAIN
template <class T> __device__ __forceinline__ Fp<T> fp_from_mont(const Fp<T>& a) {
    Fp<T> one = fp_zero<T>();
    one.v[0] = 1;
    return fp_mul(a, one);
}
#endif  // otherwise generated beside fp_mul (fp_mul_gen.cuh): the reduction alone, 72 mads instead of 128
template <class T> __device__ __forceinline__ Fp<T> fp_to_mont(const Fp<T>& a) {
    Fp<T> r2;
#pragma unroll
    for (int i = 0; i < 8; ++i) r2.v[i] = FieldParams<T>::R2(i);
    return fp_mul(a, r2);
}

// a^(p-2): not unrolled (a loop of 254 squarings), used once per output point / table entry
template <class T> __device__ __noinline__ Fp<T> fp_inv(const Fp<T>& a) {
    Fp<T> acc = fp_one<T>();
    u32 e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = FieldParams<T>::P(i);
    e[0] -= 2;  // p is odd and p[0] >= 2 for both fields
    for (int i = 253; i >= 0; --i) {
        acc = fp_sqr(acc);
        u32 w = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k == (i >> 5)) w = e[k];
        if ((w >> (i & 31)) & 1) acc = fp_mul(acc, a);
    }
    return acc;
}

// 32-byte global/LDS accessors (two 16-byte transactions per element)
template <class T> __device__ __forceinline__ Fp<T> fp_load(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    Fp<T> r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
// memory always holds the canonical representative (the ABI's layout; and what makes loads canonical)
template <class T> __device__ __forceinline__ void fp_store(void* p, const Fp<T>& x) {
    const Fp<T> a = fp_canon(x);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
