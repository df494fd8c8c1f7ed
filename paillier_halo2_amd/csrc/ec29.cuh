// ec29.cuh -- BN254 G1 (y^2 = x^3 + 3, a = 0) point arithmetic for the MSM kernels on the reduced-radix field of
// fp29.cuh (9 x 29-bit limbs, coordinates in the 2^261 Montgomery domain).
//
// Buckets are kept in extended-Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): bucket += affine point
// costs 8M + 2S, bucket + bucket 12M + 2S, both with a fully handled doubling / cancellation case (the result must be
// exact for ANY input: bases G, 2G, 3G, .. make acc == next point inside a bucket).  Identity: ZZ == 0 (all limbs).
//
// Limb / value invariants of a stored XYZZ point (what makes every f29_mul / f29_sub below legal, fp29.cuh):
//   X  : limbs < 2^29 + 8 (carried), value < 8p        Y : limbs < 2^30.6, value < 4p
//   ZZ, ZZZ : tight (< 2^29), value < 2p
// Affine points (table rows, results): tight limbs, canonical values.
#pragma once
#include "fp29.cuh"

typedef F29<FqTag> Fq29;

struct G1A29 {
    Fq29 x, y;
};
struct G1X29 {
    Fq29 x, y, zz, zzz;
};
// 36-byte-per-coordinate workspace images (never cross the ABI)
struct alignas(16) G1X29Raw {
    u32 w[36];
};
// the ABI's / the table's 64-byte affine point (256-bit canonical integers; the table keeps them in the 261-domain)
struct alignas(16) G1Aff64 {
    u32 w[16];
};

__device__ __forceinline__ bool a29_is_inf(const G1A29& p) { return f29_is_zero_exact(p.x) && f29_is_zero_exact(p.y); }
__device__ __forceinline__ bool x29_is_inf(const G1X29& p) { return f29_is_zero_exact(p.zz); }

__device__ __forceinline__ G1X29 x29_inf() {
    G1X29 r;
    r.x = f29_zero<FqTag>();
    r.y = f29_one<FqTag>();
    r.zz = f29_zero<FqTag>();
    r.zzz = f29_zero<FqTag>();
    return r;
}
__device__ __forceinline__ G1X29 x29_from_affine(const G1A29& p) {
    if (a29_is_inf(p)) return x29_inf();
    G1X29 r;
    r.x = p.x;
    r.y = p.y;
    r.zz = f29_one<FqTag>();
    r.zzz = f29_one<FqTag>();
    return r;
}

// 2 * p, p = (X, Y, ZZ, ZZZ) with the stored-point invariants; ZZ = ZZZ = 1 for an affine point   (dbl-2008-s-1)
__device__ __forceinline__ G1X29 x29_dbl_core(const Fq29& X, const Fq29& Y, const Fq29* ZZ, const Fq29* ZZZ) {
    G1X29 r;
    const Fq29 Yc = f29_carry(Y);
    const Fq29 U = f29_dbl(Yc);                        // limbs < 2^30 + 16, value < 8p
    const Fq29 V = f29_sqr(U);
    const Fq29 W = f29_mul(U, V);
    const Fq29 S = f29_mul(X, V);
    const Fq29 XX = f29_sqr(X);
    const Fq29 M = f29_carry(f29_add2(XX, XX));        // 3 X^2
    r.x = f29_carry(f29_sub<4, 30>(f29_sqr(M), f29_dbl(S)));
    const Fq29 t = f29_sub<8, 30>(S, r.x);
    r.y = f29_sub<2, 29>(f29_mul(M, t), f29_mul(W, Yc));
    r.zz = ZZ ? f29_mul(V, *ZZ) : V;
    r.zzz = ZZZ ? f29_mul(W, *ZZZ) : W;
    return r;
}
__device__ __noinline__ G1X29 x29_dbl(const G1X29 p) {
    if (x29_is_inf(p) || f29_is_zero(p.y)) return x29_inf();
    return x29_dbl_core(p.x, p.y, &p.zz, &p.zzz);
}
__device__ __noinline__ G1X29 x29_dbl_affine(const G1A29 p) {
    if (f29_is_zero(p.y)) return x29_inf();   // order-2 point: none on BN254 G1, kept for totality
    return x29_dbl_core(p.x, p.y, nullptr, nullptr);
}
// the acc == +-q cases of the mixed addition, out of line and by VALUE (nothing in the hot loop is address-taken)
__device__ __noinline__ G1X29 x29_add_affine_special(G1A29 q, bool same_y) {
    if (same_y) return x29_dbl_affine(q);
    return x29_inf();
}

// acc += q (affine, tight canonical coordinates; the caller has already negated y if needed)   (madd-2008-s)
__device__ __forceinline__ void x29_add_affine(G1X29& acc, const G1A29& q) {
    if (a29_is_inf(q)) return;
    if (x29_is_inf(acc)) {
        acc = x29_from_affine(q);
        return;
    }
    const Fq29 U2 = f29_mul(q.x, acc.zz);
    const Fq29 S2 = f29_mul(q.y, acc.zzz);
    const Fq29 P = f29_carry(f29_sub<8, 30>(U2, acc.x));    // value < 10p
    const Fq29 nY = f29_neg<4, 31>(acc.y);                  // 4p - Y1: limbs < 2^31.4, value < 4p
    const Fq29 R = f29_carry(f29_add(S2, nY));              // S2 - Y1 + 4p, value < 6p
    if (f29_is_zero(P)) {
        acc = x29_add_affine_special(q, f29_is_zero(R));
        return;
    }
    const Fq29 PP = f29_sqr(P);
    const Fq29 PPP = f29_mul(P, PP);
    const Fq29 Q = f29_mul(acc.x, PP);
    const Fq29 X3 = f29_carry(f29_sub<4, 31>(f29_sqr(R), f29_add2(PPP, Q)));   // R^2 - PPP - 2Q, value < 5.3p
    const Fq29 t = f29_sub<8, 30>(Q, X3);                                      // limbs < 2^31, value < 9.1p
    // Y3 = R t - Y1 PPP as ONE Montgomery reduction of R t + (4p - Y1) PPP (f29_mul2): the column bound needs the negated Y1
    // carried (9 (2^29+8) 2^31 + 9 (2^29+8) 2^29 + 2^59.8 < 2^63.6); 27 carry instructions for ~118 of a second reduction and
    // the limb-wise subtraction.  Value < (55 + 4.4) p^2 / (169 p) + p < 1.4p, limbs tight.
    const Fq29 Y3 = f29_mul2(R, t, f29_carry(nY), PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = f29_mul(acc.zz, PP);
    acc.zzz = f29_mul(acc.zzz, PPP);
}

__device__ __noinline__ G1X29 x29_add_special(G1X29 acc, bool same_y) {
    if (same_y) return x29_dbl(acc);
    return x29_inf();
}
// acc += q   (add-2008-s)
__device__ __forceinline__ void x29_add(G1X29& acc, const G1X29& q) {
    if (x29_is_inf(q)) return;
    if (x29_is_inf(acc)) {
        acc = q;
        return;
    }
    const Fq29 U1 = f29_mul(acc.x, q.zz);
    const Fq29 U2 = f29_mul(q.x, acc.zz);
    const Fq29 S1 = f29_mul(acc.y, q.zzz);
    const Fq29 S2 = f29_mul(q.y, acc.zzz);
    const Fq29 P = f29_carry(f29_sub<2, 29>(U2, U1));
    const Fq29 nS1 = f29_neg<2, 29>(S1);                    // 2p - S1: limbs < 2^30, value < 2p
    const Fq29 R = f29_carry(f29_add(S2, nS1));
    if (f29_is_zero(P)) {
        acc = x29_add_special(acc, f29_is_zero(R));
        return;
    }
    const Fq29 PP = f29_sqr(P);
    const Fq29 PPP = f29_mul(P, PP);
    const Fq29 Q = f29_mul(U1, PP);
    const Fq29 X3 = f29_carry(f29_sub<4, 31>(f29_sqr(R), f29_add2(PPP, Q)));
    const Fq29 t = f29_sub<8, 30>(Q, X3);
    // one reduction for R t - S1 PPP (see x29_add_affine): 9 (2^29+8) 2^31 + 9 2^30 2^29 + 2^59.8 < 2^63.9, no carry needed
    const Fq29 Y3 = f29_mul2(R, t, nS1, PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = f29_mul(f29_mul(acc.zz, q.zz), PP);
    acc.zzz = f29_mul(f29_mul(acc.zzz, q.zzz), PPP);
}

// -y of an affine point: 2p - y limb-wise from 2p spelled with low limbs >= 2^29 (9 subtractions, no borrow chain; 2p, not p:
// y's top limb may equal p's, and a limb must not go negative); the result's limbs are loose (< 2^30) -- a legal operand of
// the mixed addition's S2 = y * ZZZ (ZZZ is tight)
__device__ __forceinline__ Fq29 a29_neg_y(const Fq29& y) { return f29_neg<2, 29>(y); }

// ---- memory images
// table row / ABI affine point: two 256-bit integers
__device__ __forceinline__ G1A29 a29_load64(const void* p) {
    G1A29 r;
    r.x = f29_load<FqTag>(p);
    r.y = f29_load<FqTag>(reinterpret_cast<const char*>(p) + 32);
    return r;
}
// canonical store (coordinates below 4p)
__device__ __forceinline__ void a29_store64(void* p, const G1A29& a) {
    f29_store<1>(p, a.x);
    f29_store<1>(reinterpret_cast<char*>(p) + 32, a.y);
}
__device__ __forceinline__ G1X29 x29_load_raw(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint4 t = q[i];
        w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
    }
    G1X29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.v[i] = w[i];
        r.y.v[i] = w[9 + i];
        r.zz.v[i] = w[18 + i];
        r.zzz.v[i] = w[27 + i];
    }
    return r;
}
__device__ __forceinline__ void x29_store_raw(void* p, const G1X29& a) {
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        w[i] = a.x.v[i];
        w[9 + i] = a.y.v[i];
        w[18 + i] = a.zz.v[i];
        w[27 + i] = a.zzz.v[i];
    }
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 9; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// XYZZ (261-domain) -> the ABI's Jacobian point (x, y, z) in the 256-domain, canonical: (X*ZZ, Y*ZZZ, ZZ)
__device__ __forceinline__ void x29_store_jac(void* p, const G1X29& a) {
    char* c = reinterpret_cast<char*>(p);
    if (x29_is_inf(a)) {   // (0, R, 0) as the 32-bit path returns it
        Fp<FqTag> z = fp_zero<FqTag>();
        fp_store(c, z);
        fp_store(c + 32, fp_one<FqTag>());
        fp_store(c + 64, z);
        return;
    }
    f29_store<1>(c, f29_to_256(f29_mul(a.x, a.zz)));
    f29_store<1>(c + 32, f29_to_256(f29_mul(a.y, a.zzz)));
    f29_store<1>(c + 64, f29_to_256(a.zz));
}
// the ABI's Jacobian point -> XYZZ in the 261-domain
__device__ __forceinline__ G1X29 x29_load_jac(const void* p) {
    const char* c = reinterpret_cast<const char*>(p);
    const Fq29 z = f29_load<FqTag>(c + 64);
    if (f29_is_zero_exact(z)) return x29_inf();
    G1X29 r;
    r.x = f29_to_261(f29_load<FqTag>(c));
    r.y = f29_to_261(f29_load<FqTag>(c + 32));
    const Fq29 z1 = f29_to_261(z);
    r.zz = f29_sqr(z1);
    r.zzz = f29_mul(r.zz, z1);
    return r;
}
// XYZZ -> affine in the 261-domain (tight, value < 2p; canonicalised by the store)
__device__ __forceinline__ G1A29 x29_to_affine(const G1X29& p) {
    G1A29 r;
    if (x29_is_inf(p)) {
        r.x = f29_zero<FqTag>();
        r.y = f29_zero<FqTag>();
        return r;
    }
    const Fq29 t = f29_inv(f29_mul(p.zz, p.zzz));   // 1 / (ZZ * ZZZ)
    const Fq29 izz = f29_mul(t, p.zzz);
    const Fq29 izzz = f29_mul(t, p.zz);
    r.x = f29_mul(p.x, izz);
    r.y = f29_mul(p.y, izzz);
    return r;
}
// canonical, tight affine coordinates from an x29_to_affine result (so that the point can serve as a table row /
// an addend again: a29_neg_y and the exact identity test need canonical values)
__device__ __forceinline__ G1A29 a29_canon(const G1A29& a) {
    G1A29 r;
    r.x = f29_canon<1>(a.x);
    r.y = f29_canon<1>(a.y);
    return r;
}
