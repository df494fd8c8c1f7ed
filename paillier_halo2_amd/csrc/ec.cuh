// ec.cuh -- BN254 G1 (y^2 = x^3 + 3, a = 0) point arithmetic for the MSM kernels.
//
// Buckets are kept in extended-Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// a bucket += affine-point update costs 8M+2S, a bucket + bucket add 12M+2S, and both have a
// cheap, fully handled doubling / cancellation case, which matters here because the result must
// be exact for ANY input (e.g. bases G,2G,3G.. make acc == next point inside a bucket).
// Identity: ZZ == 0.  Affine identity (C ABI / halo2curves G1Affine): x == y == 0.
#pragma once
#include "fp.cuh"

struct alignas(16) G1Affine {
    Fq x, y;
};
struct alignas(16) G1Jac {
    Fq x, y, z;
};
struct alignas(16) G1X {
    Fq x, y, zz, zzz;
};

// exact tests: affine coordinates come from memory (canonical) or are set to zero for the identity; ZZ is a product of
// non-zero elements unless it was set to zero (fp.cuh, lazy range)
__device__ __forceinline__ bool aff_is_inf(const G1Affine& p) { return fp_is_zero_exact(p.x) && fp_is_zero_exact(p.y); }
__device__ __forceinline__ bool x_is_inf(const G1X& p) { return fp_is_zero_exact(p.zz); }

__device__ __forceinline__ G1X x_inf() {
    G1X r;
    r.x = fp_zero<FqTag>();
    r.y = fp_one<FqTag>();
    r.zz = fp_zero<FqTag>();
    r.zzz = fp_zero<FqTag>();
    return r;
}
__device__ __forceinline__ G1X x_from_affine(const G1Affine& p) {
    G1X r;
    if (aff_is_inf(p)) return x_inf();
    r.x = p.x;
    r.y = p.y;
    r.zz = fp_one<FqTag>();
    r.zzz = fp_one<FqTag>();
    return r;
}

// 2 * (affine p), p != identity   (mdbl-2008-s-1)
__device__ __noinline__ G1X x_dbl_affine(const G1Affine& p) {
    G1X r;
    if (fp_is_zero(p.y)) return x_inf();  // order-2 point: none on BN254 G1, kept for totality
    Fq U = fp_dbl(p.y);
    Fq V = fp_sqr(U);
    Fq W = fp_mul(U, V);
    Fq S = fp_mul(p.x, V);
    Fq xx = fp_sqr(p.x);
    Fq M = fp_add(fp_dbl(xx), xx);
    r.x = fp_sub(fp_sqr(M), fp_dbl(S));
    r.y = fp_sub(fp_mul(M, fp_sub(S, r.x)), fp_mul(W, p.y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// 2 * p   (dbl-2008-s-1)
__device__ __noinline__ G1X x_dbl(const G1X& p) {
    if (x_is_inf(p) || fp_is_zero(p.y)) return x_inf();
    G1X r;
    Fq U = fp_dbl(p.y);
    Fq V = fp_sqr(U);
    Fq W = fp_mul(U, V);
    Fq S = fp_mul(p.x, V);
    Fq xx = fp_sqr(p.x);
    Fq M = fp_add(fp_dbl(xx), xx);
    r.x = fp_sub(fp_sqr(M), fp_dbl(S));
    r.y = fp_sub(fp_mul(M, fp_sub(S, r.x)), fp_mul(W, p.y));
    r.zz = fp_mul(V, p.zz);
    r.zzz = fp_mul(W, p.zzz);
    return r;
}

// out = 2*q if same_y else identity: the acc == q / acc == -q cases of the mixed addition
// (arguments and result by VALUE: they travel in VGPRs, nothing in the caller's hot loop is address-taken)
__device__ __noinline__ G1X x_add_affine_special(G1Affine q, bool same_y) {
    if (same_y) return x_dbl_affine(q);
    return x_inf();
}

// acc += q (affine), q possibly negated by the caller beforehand.   (madd-2008-s)
__device__ __forceinline__ void x_add_affine(G1X& acc, const G1Affine& q) {
    if (aff_is_inf(q)) return;
    if (x_is_inf(acc)) {
        acc.x = q.x;
        acc.y = q.y;
        acc.zz = fp_one<FqTag>();
        acc.zzz = fp_one<FqTag>();
        return;
    }
    Fq U2 = fp_mul(q.x, acc.zz);
    Fq S2 = fp_mul(q.y, acc.zzz);
    Fq P = fp_sub(U2, acc.x);
    Fq R = fp_sub(S2, acc.y);
    if (fp_is_zero(P)) {
        // rare (acc == +-q): handled out of line with by-value arguments, so neither the accumulator nor the
        // loaded point ever has its address taken in the hot loop (an sret call on `acc` put it in scratch:
        // PMC showed 32 GB of scratch write-back per accumulate launch)
        acc = x_add_affine_special(q, fp_is_zero(R));
        return;
    }
    Fq PP = fp_sqr(P);
    Fq PPP = fp_mul(P, PP);
    Fq Q = fp_mul(acc.x, PP);
    Fq X3 = fp_sub(fp_sub(fp_sqr(R), PPP), fp_dbl(Q));
    Fq Y3 = fp_sub(fp_mul(R, fp_sub(Q, X3)), fp_mul(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fp_mul(acc.zz, PP);
    acc.zzz = fp_mul(acc.zzz, PPP);
}

// rare case of the full addition (acc == +-q), by value for the same reason as above
__device__ __noinline__ G1X x_add_special(G1X acc, bool same_y) {
    if (same_y) return x_dbl(acc);
    return x_inf();
}

// acc += q   (add-2008-s)
__device__ __forceinline__ void x_add(G1X& acc, const G1X& q) {
    if (x_is_inf(q)) return;
    if (x_is_inf(acc)) {
        acc = q;
        return;
    }
    Fq U1 = fp_mul(acc.x, q.zz);
    Fq U2 = fp_mul(q.x, acc.zz);
    Fq S1 = fp_mul(acc.y, q.zzz);
    Fq S2 = fp_mul(q.y, acc.zzz);
    Fq P = fp_sub(U2, U1);
    Fq R = fp_sub(S2, S1);
    if (fp_is_zero(P)) {
        acc = x_add_special(acc, fp_is_zero(R));
        return;
    }
    Fq PP = fp_sqr(P);
    Fq PPP = fp_mul(P, PP);
    Fq Q = fp_mul(U1, PP);
    Fq X3 = fp_sub(fp_sub(fp_sqr(R), PPP), fp_dbl(Q));
    Fq Y3 = fp_sub(fp_mul(R, fp_sub(Q, X3)), fp_mul(S1, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fp_mul(fp_mul(acc.zz, q.zz), PP);
    acc.zzz = fp_mul(fp_mul(acc.zzz, q.zzz), PPP);
}

// XYZZ -> Jacobian without inversion: (X*ZZ, Y*ZZZ, ZZ) since x = X*ZZ/ZZ^2, y = Y*ZZZ/ZZ^3
__device__ __forceinline__ G1Jac x_to_jac(const G1X& p) {
    G1Jac r;
    if (x_is_inf(p)) {
        r.x = fp_zero<FqTag>();
        r.y = fp_one<FqTag>();
        r.z = fp_zero<FqTag>();
        return r;
    }
    r.x = fp_mul(p.x, p.zz);
    r.y = fp_mul(p.y, p.zzz);
    r.z = p.zz;
    return r;
}
__device__ __forceinline__ G1X jac_to_x(const G1Jac& p) {
    G1X r;
    if (fp_is_zero(p.z)) return x_inf();
    r.x = p.x;
    r.y = p.y;
    r.zz = fp_sqr(p.z);
    r.zzz = fp_mul(r.zz, p.z);
    return r;
}
__device__ __forceinline__ G1Affine x_to_affine(const G1X& p) {
    G1Affine r;
    if (x_is_inf(p)) {
        r.x = fp_zero<FqTag>();
        r.y = fp_zero<FqTag>();
        return r;
    }
    // 1/ZZZ gives both: 1/ZZ = ZZZ^2/ZZ^3 * 1/ZZ ... use one inversion of ZZ*ZZZ
    Fq t = fp_inv(fp_mul(p.zz, p.zzz));   // 1/(ZZ*ZZZ)
    Fq izz = fp_mul(t, p.zzz);            // 1/ZZ
    Fq izzz = fp_mul(t, p.zz);            // 1/ZZZ
    r.x = fp_mul(p.x, izz);
    r.y = fp_mul(p.y, izzz);
    return r;
}

__device__ __forceinline__ G1Affine aff_load(const void* p) {
    G1Affine r;
    r.x = fp_load<FqTag>(p);
    r.y = fp_load<FqTag>(reinterpret_cast<const char*>(p) + 32);
    return r;
}
__device__ __forceinline__ void aff_store(void* p, const G1Affine& a) {
    fp_store(p, a.x);
    fp_store(reinterpret_cast<char*>(p) + 32, a.y);
}
__device__ __forceinline__ G1X x_load(const void* p) {
    G1X r;
    const char* c = reinterpret_cast<const char*>(p);
    r.x = fp_load<FqTag>(c);
    r.y = fp_load<FqTag>(c + 32);
    r.zz = fp_load<FqTag>(c + 64);
    r.zzz = fp_load<FqTag>(c + 96);
    return r;
}
__device__ __forceinline__ void x_store(void* p, const G1X& a) {
    char* c = reinterpret_cast<char*>(p);
    fp_store(c, a.x);
    fp_store(c + 32, a.y);
    fp_store(c + 64, a.zz);
    fp_store(c + 96, a.zzz);
}
__device__ __forceinline__ void jac_store(void* p, const G1Jac& a) {
    char* c = reinterpret_cast<char*>(p);
    fp_store(c, a.x);
    fp_store(c + 32, a.y);
    fp_store(c + 64, a.z);
}
