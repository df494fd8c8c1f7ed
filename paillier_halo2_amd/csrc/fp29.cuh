// fp29.cuh -- BN254 Fq / Fr in a REDUCED RADIX for the multiplier-bound kernels: 9 limbs x 29 bits in 32-bit
// registers, Montgomery form with R' = 2^261.
//
// Why (measured on MI355X): the only wide multiplier of CDNA4's VALU is v_mad_u64_u32
// (32 x 32 + 64 -> 64, carry-out to an SGPR pair).  On saturated 32-bit limbs every mad of a product scan can carry
// out of its 64-bit accumulator, so fp.cuh pays one v_addc per mad (128 + 128) plus wait states.  With 29-bit
// limbs a whole column of the scan -- at most 9 a_i*b_j and 9 m_i*p_j terms -- stays below 2^64 (bounds below):
// the mad's 64-bit addend absorbs every carry, no carry-out is ever read, and additions / subtractions become
// 9 independent 32-bit operations with NO carry chain.  171 multiplier instructions instead of 136, but ~210
// instructions per product instead of ~380: +15 % products per second before any tuning (pz_ubench_fqmul_variant).
//
// Representation and discipline
//   value  = sum v[i] * 2^(29 i); any representative modulo p is allowed, values stay far below 2^261.
//   limbs  : "tight" < 2^29 (products, loads, f29_carry results up to +7), "loose" anything that fits 32 bits.
//   f29_mul(a, b): needs  9 * max(a limb) * max(b limb) + 2^59.8 < 2^64  (the m*p part of a column is below
//                  2^29 * sum_j p_j < 2^59.8), e.g. tight x 2^31.0, or 2^30.3 x 2^30.3.  Result: tight limbs,
//                  value < va*vb / (169 p) + p   (R' = 2^261 > 169 p).
//   f29_add      : limb-wise.   f29_sub<K, LB>(a, b) = a - b + K*p with K*p spelled so that its low 8 limbs are
//                  >= 2^LB (>= every limb of b) and its value >= b's: no limb goes negative, no borrow chain.
//   f29_carry    : one parallel carry round (3 instructions per limb, no serial dependency): limbs < 2^29 + 8.
// Montgomery domains: memory and the C ABI hold x * 2^256 mod p (halo2curves' layout, "256-domain").  A Montgomery
// product by a TABLE constant kept as c * 2^261 leaves its other operand's domain unchanged, so NTT data never
// changes domain; the EC kernels convert bases once when the window table is built (f29_mul by 2^266).
#pragma once
#include "fp.cuh"

static constexpr u32 F29_MASK = 0x1fffffffu;

template <class T> struct alignas(4) F29 {
    u32 v[9];
};

template <class Tag> struct P29;
template <> struct P29<FqTag> {
    static constexpr u32 INV = 0x04866389u;  // -p^-1 mod 2^29
    __device__ __host__ __forceinline__ static constexpr u32 P(int i) {
        constexpr u32 p[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                              0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return p[i];
    }
    __device__ __forceinline__ static constexpr u32 ONE(int i) {   // 2^261 mod p
        constexpr u32 c[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                              0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return c[i];
    }
    __device__ __forceinline__ static constexpr u32 R266(int i) {  // 2^266 mod p: 256-domain -> 261-domain
        constexpr u32 c[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u,
                              0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
        return c[i];
    }
    __device__ __forceinline__ static constexpr u32 R256(int i) {  // 2^256 mod p: 261-domain -> 256-domain
        constexpr u32 c[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u,
                              0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
        return c[i];
    }
    __device__ __forceinline__ static constexpr u32 R517(int i) {  // 2^517 mod p: integer -> 256-domain
        constexpr u32 c[9] = {0x0f6b5c04u, 0x08ead878u, 0x1645525du, 0x1aefe9cdu, 0x09d605edu,
                              0x0483a115u, 0x0d08508bu, 0x0dba4804u, 0x001982b4u};
        return c[i];
    }
};
template <> struct P29<FrTag> {
    static constexpr u32 INV = 0x0fffffffu;
    __device__ __host__ __forceinline__ static constexpr u32 P(int i) {
        constexpr u32 p[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                              0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return p[i];
    }
    __device__ __forceinline__ static constexpr u32 ONE(int i) {
        constexpr u32 c[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                              0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return c[i];
    }
    __device__ __forceinline__ static constexpr u32 R266(int i) {
        constexpr u32 c[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau,
                              0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
        return c[i];
    }
    __device__ __forceinline__ static constexpr u32 R256(int i) {
        constexpr u32 c[9] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu,
                              0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
        return c[i];
    }
    __device__ __forceinline__ static constexpr u32 R517(int i) {
        constexpr u32 c[9] = {0x142db4dfu, 0x19d6990eu, 0x1472f48cu, 0x06dbe7e3u, 0x0b84d579u,
                              0x10f9faf7u, 0x121f4380u, 0x17a112deu, 0x001275c7u};
        return c[i];
    }
};

// limb i of K*p in strict 29-bit limbs (compile-time for constant i)
template <class T> __device__ __host__ __forceinline__ constexpr u32 f29_kp_limb(unsigned K, int i) {
    u64 carry = 0;
    u32 out = 0;
    for (int j = 0; j <= i; ++j) {
        const u64 t = (u64)P29<T>::P(j) * K + carry;
        out = (u32)(t & F29_MASK);
        carry = t >> 29;
        if (j == 8 && j == i) out = (u32)t;   // the top limb keeps everything (K*p < 2^261)
    }
    return out;
}
// limb i of pbar = 2^261 - p in strict 29-bit limbs (f29_mulc subtracts q * p as q * pbar modulo 2^261)
template <class T> __device__ __host__ __forceinline__ constexpr u32 f29_pbar_limb(int i) {
    u32 borrow = 0, out = 0;
    for (int j = 0; j <= i; ++j) {
        const u32 pj = P29<T>::P(j) + borrow;          // <= 2^29
        out = (0u - pj) & F29_MASK;
        borrow = pj != 0 ? 1u : 0u;
    }
    return out;
}
// K*p spelled with low limbs >= 2^LB: limb_i + 2^LB for i < 8, minus the 2^(LB-29) units borrowed by the limb below
template <class T, unsigned K, unsigned LB> __device__ __forceinline__ constexpr u32 f29_spread(int i) {
    return f29_kp_limb<T>(K, i) + (i < 8 ? (1u << LB) : 0u) - (i > 0 ? (1u << (LB - 29)) : 0u);
}

template <class T> __device__ __forceinline__ F29<T> f29_zero() {
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = 0;
    return r;
}
template <class T> __device__ __forceinline__ F29<T> f29_one() {   // 1 in the 261-domain
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = P29<T>::ONE(i);
    return r;
}
template <class T> __device__ __forceinline__ F29<T> f29_add(const F29<T>& a, const F29<T>& b) {
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// a + 2b
template <class T> __device__ __forceinline__ F29<T> f29_add2(const F29<T>& a, const F29<T>& b) {
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = (b.v[i] << 1) + a.v[i];
    return r;
}
template <class T> __device__ __forceinline__ F29<T> f29_dbl(const F29<T>& a) {
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] << 1;
    return r;
}
// a - b + K*p: b's limbs must be <= 2^LB - 1 + (limb of K*p) -- guaranteed below 2^LB -- and b's value <= K*p - 2^233
template <unsigned K, unsigned LB, class T> __device__ __forceinline__ F29<T> f29_sub(const F29<T>& a, const F29<T>& b) {
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = (a.v[i] + f29_spread<T, K, LB>(i)) - b.v[i];
    return r;
}
// K*p - b
template <unsigned K, unsigned LB, class T> __device__ __forceinline__ F29<T> f29_neg(const F29<T>& b) {
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = f29_spread<T, K, LB>(i) - b.v[i];
    return r;
}
// one parallel carry round: limbs < 2^29 + 8 afterwards (the top limb takes what is left; values stay < 2^261)
template <class T> __device__ __forceinline__ F29<T> f29_carry(const F29<T>& a) {
    F29<T> r;
    r.v[0] = a.v[0] & F29_MASK;
#pragma unroll
    for (int i = 1; i < 8; ++i) r.v[i] = (a.v[i] & F29_MASK) + (a.v[i - 1] >> 29);
    r.v[8] = a.v[8] + (a.v[7] >> 29);
    return r;
}
// exact normalisation: every limb below 2^29 (serial carry chain)
template <class T> __device__ __forceinline__ F29<T> f29_norm(const F29<T>& a) {
    F29<T> r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 t = a.v[i] + c;
        r.v[i] = t & F29_MASK;
        c = t >> 29;
    }
    r.v[8] = a.v[8] + c;
    return r;
}

// r = a - c if a >= c (strict limbs both), else a.  c given by a limb function.
template <class T, unsigned K> __device__ __forceinline__ void f29_cond_sub_kp(F29<T>& a) {
    u32 d[9];
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const u32 t = a.v[i] - f29_kp_limb<T>(K, i) - br;
        br = t >> 31;
        d[i] = i < 8 ? (t & F29_MASK) : t;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) a.v[i] = br ? a.v[i] : d[i];
}
// canonical representative in [0, p), strict limbs; the value must be below 2^(J+1) * p
template <unsigned J, class T> __device__ __forceinline__ F29<T> f29_canon(const F29<T>& a) {
    F29<T> r = f29_norm(a);
    if (J >= 6) f29_cond_sub_kp<T, 64>(r);
    if (J >= 5) f29_cond_sub_kp<T, 32>(r);
    if (J >= 4) f29_cond_sub_kp<T, 16>(r);
    if (J >= 3) f29_cond_sub_kp<T, 8>(r);
    if (J >= 2) f29_cond_sub_kp<T, 4>(r);
    if (J >= 1) f29_cond_sub_kp<T, 2>(r);
    f29_cond_sub_kp<T, 1>(r);
    return r;
}

// ---- canonicalisation through a quotient estimate (a value below 64p, loose limbs; the estimate's multiplier is exact for
// top limbs below 2^31): the top limb after the carry chain gives
// q = floor(top / (p_top + 1)) with q <= floor(V / p) <= q + 1 (V / p - top / (p_top + 1) < 33 / p_top ~ 1e-5), so V - q p lies in
// [0, 2p) and ONE conditional subtraction finishes -- ~125 cheap instructions instead of the ~250 of f29_canon<4>'s five-step
// ladder.  q p comes from a 33-row table in LDS (f29_qtab_fill; a row per q, strict limbs).  p_top = p >> 232 = 0x30644e for
// BOTH BN254 fields; 0xa948e6d8 = floor(2^53 / (p_top + 1)) + 1 makes (top * M) >> 53 the exact floor for top < 2^27.
#define F29_QTAB_ROWS 65u   // values below 64p (round 5: the constant-operand products leave < 3p, a pair of stages adds up to 8p)
#define F29_QTAB_WORDS (F29_QTAB_ROWS * 9u)
template <class T> __device__ __forceinline__ void f29_qtab_fill(u32* tab) {   // every thread of the workgroup; __syncthreads() afterwards
    static_assert((P29<T>::P(8) == 0x30644eu), "quotient-estimate constant assumes p >> 232 == 0x30644e");
    for (unsigned t = threadIdx.x; t < F29_QTAB_WORDS; t += blockDim.x) {
        const unsigned q = t / 9u, i = t % 9u;
        u64 carry = 0;
        u32 out = 0;
        for (unsigned j = 0; j <= i; ++j) {
            const u64 v = (u64)P29<T>::P((int)j) * q + carry;
            out = j == 8 ? (u32)v : (u32)(v & F29_MASK);
            carry = v >> 29;
        }
        tab[t] = out;
    }
}
template <class T> __device__ __forceinline__ F29<T> f29_canon_q(const F29<T>& a, const u32* __restrict__ tab) {
    const F29<T> n = f29_norm(a);
    const u32 q = __umulhi(n.v[8], 0xa948e6d8u) >> 21;
    const u32* t = tab + q * 9u;
    F29<T> r;
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const u32 d = n.v[i] - t[i] - br;
        br = d >> 31;
        r.v[i] = i < 8 ? (d & F29_MASK) : d;
    }
    f29_cond_sub_kp<T, 1>(r);
    return r;
}

// a == 0 (mod p)?  the value must be below 32 p.  Quick filter on the low limb: a = j*p  =>  a_0 * (-p^-1) = -j mod 2^29;
// only a hit (probability 2^-24 for a random element) pays the exact reduction.
template <class T> __device__ __forceinline__ bool f29_is_zero(const F29<T>& a) {
    const u32 j = (0u - a.v[0] * P29<T>::INV) & F29_MASK;   // = 2^29 - (a_0 * INV mod 2^29); 0 -> 0
    // a = j*p (0 <= j < 32) gives a_0 * INV = -j, i.e. this is j itself
    if (j >= 32u) return false;
    const F29<T> c = f29_canon<4>(a);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o |= c.v[i];
    return o == 0;
}
// exactly the all-zero limb pattern (values that are zero only when they were set to zero: loaded coordinates, ZZ)
template <class T> __device__ __forceinline__ bool f29_is_zero_exact(const F29<T>& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o |= a.v[i];
    return o == 0;
}

// ---- packing: 8 x 32-bit words (a 256-bit integer) <-> 9 x 29-bit limbs
template <class T> __device__ __forceinline__ F29<T> f29_unpack(const u32 w[8]) {
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, j = bit >> 5, o = bit & 31;
        u32 x;
        if (o == 0) x = w[j];
        else if (j + 1 < 8) x = __builtin_amdgcn_alignbit(w[j + 1], w[j], o);
        else x = w[j] >> o;
        r.v[i] = i < 8 ? (x & F29_MASK) : x;
    }
    return r;
}
// the limbs of x * 2^5 (x a 256-bit integer: the value stays below 2^261, NOT reduced -- up to 32 x): limb i = bits
// [29 i - 5, 29 i + 24) of x.  What it is for: memory holds field elements as x * 2^256 (the ABI's Montgomery form) while
// f29_mul divides by 2^261, so a product of two LOADED values needs one of them as x * 2^261 = (x * 2^256) * 2^5 to land in
// the 256-domain again; unpacking with the shift built in costs the same 17 instructions as f29_unpack.
template <class T> __device__ __forceinline__ F29<T> f29_unpack_shl5(const u32 w[8]) {
    F29<T> r;
    r.v[0] = (w[0] << 5) & F29_MASK;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int bit = 29 * i - 5, j = bit >> 5, o = bit & 31;
        u32 x;
        if (o == 0) x = w[j];
        else if (j + 1 < 8) x = __builtin_amdgcn_alignbit(w[j + 1], w[j], o);
        else x = w[j] >> o;
        r.v[i] = i < 8 ? (x & F29_MASK) : x;
    }
    return r;
}
// strict limbs, value < 2^256
template <class T> __device__ __forceinline__ void f29_pack(const F29<T>& a, u32 w[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int bit = 32 * j, i = bit / 29, o = bit - 29 * i;
        // word j = bits [32j, 32j+32): limb i from bit o, then limb i+1 from bit 0 (29 - o + 29 >= 32 always)
        w[j] = (a.v[i] >> o) | (a.v[i + 1] << (29 - o));
    }
}
template <class T> __device__ __forceinline__ F29<T> f29_from_fp(const Fp<T>& x) { return f29_unpack<T>(x.v); }
template <class T> __device__ __forceinline__ F29<T> f29_load(const void* p) { return f29_from_fp(fp_load<T>(p)); }
template <class T> __device__ __forceinline__ F29<T> f29_from_fp_shl5(const Fp<T>& x) { return f29_unpack_shl5<T>(x.v); }
template <class T> __device__ __forceinline__ F29<T> f29_load_shl5(const void* p) { return f29_from_fp_shl5(fp_load<T>(p)); }
// store the canonical representative as a 256-bit integer (value below 2^(J+1) p)
template <unsigned J, class T> __device__ __forceinline__ void f29_store(void* p, const F29<T>& x) {
    const F29<T> c = f29_canon<J>(x);
    u32 w[8];
    f29_pack(c, w);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// the same for a value that comes straight out of f29_mul / f29_mul2 / f29_dot4 (strict limbs, below 2p): no carry chain and
// ONE conditional subtraction -- 60 instructions against ~130 for f29_store<1>
template <class T> __device__ __forceinline__ void f29_store_product(void* p, const F29<T>& x) {
    F29<T> c = x;
    f29_cond_sub_kp<T, 1>(c);
    u32 w[8];
    f29_pack(c, w);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// the canonical representative as an 8 x 32-bit element (value below 2^(J+1) p)
template <unsigned J, class T> __device__ __forceinline__ Fp<T> f29_to_fp(const F29<T>& x) {
    const F29<T> c = f29_canon<J>(x);
    Fp<T> r;
    f29_pack(c, r.v);
    return r;
}
// internal 36-byte form (9 words, strict limbs not required): workspaces that never cross the ABI
template <class T> __device__ __forceinline__ F29<T> f29_load_raw(const u32* p) {
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = p[i];
    return r;
}
template <class T> __device__ __forceinline__ void f29_store_raw(u32* p, const F29<T>& a) {
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = a.v[i];
}

#include "fp29_gen.cuh"

template <class T> __device__ __forceinline__ F29<T> f29_const(u32 (*f)(int)) {
    F29<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = f(i);
    return r;
}
// domain changes (one Montgomery product each)
template <class T> __device__ __forceinline__ F29<T> f29_to_261(const F29<T>& x256) {
    F29<T> c;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.v[i] = P29<T>::R266(i);
    return f29_mul(x256, c);
}
template <class T> __device__ __forceinline__ F29<T> f29_to_256(const F29<T>& x261) {
    F29<T> c;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.v[i] = P29<T>::R256(i);
    return f29_mul(x261, c);
}
// a^(p-2) in the 261-domain (once per output point)
template <class T> __device__ __noinline__ F29<T> f29_inv(const F29<T>& a) {
    F29<T> acc = f29_one<T>();
    u32 e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = FieldParams<T>::P(i);
    e[0] -= 2;
    for (int i = 253; i >= 0; --i) {
        acc = f29_sqr(acc);
        u32 w = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k == (i >> 5)) w = e[k];
        if ((w >> (i & 31)) & 1) acc = f29_mul(acc, a);
    }
    return acc;
}

// A CONSTANT as f29_mulc takes it: c (the plain integer below p) and cq = floor(c * 2^261 / p), 9 + 9 limbs.  Built ONCE per table entry
// from the Montgomery form t = c * 2^256 mod p the power tables hold (binary long division: 261 shift / compare / subtract steps).
struct C18 {
    u32 w[9], q[9];
};
template <class T> __device__ inline void f29_cpair_from_mont(const Fp<T>& t, u32 out[18]) {
    const Fp<T> c = fp_from_mont(t);          // canonical plain value below p
    u64 rem[4], P[4], Q[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        rem[i] = (u64)c.v[2 * i] | ((u64)c.v[2 * i + 1] << 32);
        P[i] = (u64)FieldParams<T>::P(2 * i) | ((u64)FieldParams<T>::P(2 * i + 1) << 32);
    }
    for (int b = 0; b < 261; ++b) {
        // rem < p < 2^254: 2 rem fits 256 bits
        for (int i = 3; i > 0; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
        rem[0] <<= 1;
        for (int i = 4; i > 0; --i) Q[i] = (Q[i] << 1) | (Q[i - 1] >> 63);
        Q[0] <<= 1;
        bool ge = true;
        for (int i = 3; i >= 0; --i)
            if (rem[i] != P[i]) {
                ge = rem[i] > P[i];
                break;
            }
        if (ge) {
            u64 br = 0;
            for (int i = 0; i < 4; ++i) {
                const u64 x = rem[i] - P[i], b1 = rem[i] < P[i], y = x - br;
                br = b1 | (x < br);
                rem[i] = y;
            }
            Q[0] |= 1;
        }
    }
    const F29<T> cl = f29_unpack<T>(c.v);
#pragma unroll
    for (int i = 0; i < 9; ++i) out[i] = cl.v[i];
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, j = bit >> 6, o = bit & 63;
        u64 x = Q[j] >> o;
        if (o > 35 && j + 1 < 5) x |= Q[j + 1] << (64 - o);
        out[9 + i] = (u32)x & F29_MASK;
    }
}
