// ec29_quad.cuh -- QUAD-COOPERATIVE point addition / doubling for the latency-bound reductions of K1 (few columns: one large
// MSM, a rank's share of it).  There the chip is nearly empty and every XYZZ addition is a chain of 14 dependent field
// products on ONE lane (~6 us): the reduction of 2^15 buckets is ~45 such additions deep.  The products of one addition are
// only 4 deep (add-2008-s: {U1,U2,S1,S2} -> {PP,R^2,ZZ1 ZZ2,ZZZ1 ZZZ2} -> {PPP,Q,ZZ3,W} -> {R(Q-X3), S1 PPP, ZZZ3}), so the four
// lanes of a quad, all holding BOTH operands, each take one product of a level and exchange the results with DPP quad
// broadcasts (9 v_mov_dpp per value): 4 product latencies per addition instead of 14, 3 instead of 9 per doubling.
// Every lane of the quad ends with the full result (replicated).  Values, bounds and special cases are ec29.cuh's: the
// exceptional inputs (identity operands, P == 0) take the serial routines on all four lanes, so results are identical bit
// for bit to x29_add / x29_dbl.
#pragma once
#include "ec29.cuh"

__device__ __forceinline__ u32 quad_bcast(u32 v, int k) {   // lane k of every quad -> its four lanes
    switch (k) {
        case 0: return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xf, 0xf, true);
        case 1: return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xf, 0xf, true);
        case 2: return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xaa, 0xf, 0xf, true);
        default: return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xff, 0xf, 0xf, true);
    }
}
template <int K> __device__ __forceinline__ Fq29 quad_get(const Fq29& own) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        u32 v = quad_bcast(own.v[i], K);
        asm volatile("" : "+v"(v));   // keep the exchanged value a plain register: hipcc (ROCm 7.2) otherwise folds some of these
        r.v[i] = v;                   // v_mov_dpp into their consumers with the wrong lane select (seen: lane 1's value replaced by the lane's own)
    }
    return r;
}
// the operand this lane multiplies: one of four, by its position in the quad
__device__ __forceinline__ Fq29 quad_sel(unsigned l, const Fq29& a0, const Fq29& a1, const Fq29& a2, const Fq29& a3) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const u32 lo = l & 1u ? a1.v[i] : a0.v[i], hi = l & 1u ? a3.v[i] : a2.v[i];
        r.v[i] = l & 2u ? hi : lo;
    }
    return r;
}
// one level: lane l computes a_l * b_l; all four products come back to every lane
__device__ __forceinline__ void quad_mul4(unsigned l, const Fq29& a0, const Fq29& b0, const Fq29& a1, const Fq29& b1, const Fq29& a2,
                                          const Fq29& b2, const Fq29& a3, const Fq29& b3, Fq29& r0, Fq29& r1, Fq29& r2, Fq29& r3) {
    const Fq29 own = f29_mul(quad_sel(l, a0, a1, a2, a3), quad_sel(l, b0, b1, b2, b3));
    r0 = quad_get<0>(own);
    r1 = quad_get<1>(own);
    r2 = quad_get<2>(own);
    r3 = quad_get<3>(own);
}

// acc += q on the four lanes of a quad (both operands replicated in all four; so is the result).  EXEC must hold whole quads.
// Written without early returns: the DPP exchanges are convergent operations, every lane of the quad must reach each of them.
__device__ __forceinline__ void x29_add_quad(G1X29& acc, const G1X29& q) {
    const unsigned l = threadIdx.x & 3u;
    const bool q_inf = x29_is_inf(q), a_inf = x29_is_inf(acc);   // replicated operands: the same on all four lanes
    G1X29 r = acc;
    if (q_inf) {
        // nothing to add
    } else if (a_inf) {
        r = q;
    } else {
        Fq29 U1, U2, S1, S2;
        quad_mul4(l, acc.x, q.zz, q.x, acc.zz, acc.y, q.zzz, q.y, acc.zzz, U1, U2, S1, S2);
        const Fq29 P = f29_carry(f29_sub<2, 29>(U2, U1));
        const Fq29 R = f29_carry(f29_sub<2, 29>(S2, S1));
        if (f29_is_zero(P)) {
            r = x29_add_special(acc, f29_is_zero(R));
        } else {
            Fq29 PP, RR, ZZ12, ZZZ12;
            quad_mul4(l, P, P, R, R, acc.zz, q.zz, acc.zzz, q.zzz, PP, RR, ZZ12, ZZZ12);
            Fq29 PPP, Q, ZZ3, W;
            quad_mul4(l, P, PP, U1, PP, ZZ12, PP, ZZZ12, PP, PPP, Q, ZZ3, W);
            const Fq29 X3 = f29_carry(f29_sub<4, 31>(RR, f29_add2(PPP, Q)));
            const Fq29 t = f29_sub<8, 30>(Q, X3);
            Fq29 A, B, ZZZ3, dup;
            quad_mul4(l, R, t, S1, PPP, W, P, W, P, A, B, ZZZ3, dup);
            r.x = X3;
            r.y = f29_sub<2, 29>(A, B);
            r.zz = ZZ3;
            r.zzz = ZZZ3;
        }
    }
    acc = r;
}

// 2 p on the four lanes of a quad (dbl-2008-s-1: {V = U^2, XX = X^2} -> {W = U V, S = X V, M^2} -> {M (S - X3), W Y, V ZZ, W ZZZ})
__device__ __forceinline__ G1X29 x29_dbl_quad(const G1X29& p) {
    const unsigned l = threadIdx.x & 3u;
    if (x29_is_inf(p) || f29_is_zero(p.y)) return x29_inf();
    const Fq29 Yc = f29_carry(p.y);
    const Fq29 U = f29_dbl(Yc);
    Fq29 V, XX, d0, d1;
    quad_mul4(l, U, U, p.x, p.x, U, U, p.x, p.x, V, XX, d0, d1);
    const Fq29 M = f29_carry(f29_add2(XX, XX));
    Fq29 W, S, MM, d2;
    quad_mul4(l, U, V, p.x, V, M, M, M, M, W, S, MM, d2);
    G1X29 r;
    r.x = f29_carry(f29_sub<4, 30>(MM, f29_dbl(S)));
    const Fq29 t = f29_sub<8, 30>(S, r.x);
    Fq29 A, B, ZZ3, ZZZ3;
    quad_mul4(l, M, t, W, Yc, V, p.zz, W, p.zzz, A, B, ZZ3, ZZZ3);
    r.y = f29_sub<2, 29>(A, B);
    r.zz = ZZ3;
    r.zzz = ZZZ3;
    return r;
}
