// pz_bigint.hip -- K3: big-integer witness generation for c = g^m * r^n mod n^2.
// Replaces the num-bigint arithmetic inside biguint-halo2's BigUintChip::{mul_mod,
// pow_mod_fixed_exp} (call sites /root/reference/src/paillier.rs:51,55,57,81) and
// paillier_enc_native / paillier_add_native (paillier.rs:87-97): every step yields the exact
// quotient and remainder (q, r) of a*b by the modulus, because the circuit assigns both.
//
// Mapping to CDNA4: one 64-lane wavefront owns one big integer, lane j holds limbs [j*E, j*E+E)
// (E = 1: up to 64 x 64-bit limbs = 4096 bits, the 2048-bit-n case; E = 2: up to 128 limbs, the
// 3072-bit-n case).  A product is a lane-systolic schoolbook: per outer limb A_i (broadcast with
// v_readlane) every lane multiplies its limbs (v_mad_u64_u32), adds into its column sums and the
// column array shifts one lane down; carries are kept as per-column counters and resolved once
// per product with a ballot-based carry look-ahead (64-bit scalar add of generate/propagate
// masks).  Reduction is Barrett with a normalised modulus, so the exact quotient falls out; the
// reciprocal is computed on the device once per modulus by restoring division.
// The modexp chain (k_pow_mod_chain) is latency-bound by construction -- every step depends on the one before -- and one
// wavefront is ISSUE-bound inside a product (one VALU instruction per ~4.5 cycles: 64 outer limbs x ~22 instructions), so a
// product is worked by a TEAM of four wavefronts, one per SIMD of the CU: wave w takes the outer limbs [16w, 16w + 16) of a
// against all of b, the four partial products meet in LDS (one workgroup barrier per product, double-buffered) and every wave of
// the team sums them, so all four hold the full result and the O(n) parts of a step (shifts, additions, comparisons) simply run
// redundantly.  A chain is TWO workgroups (two CUs: with both on one CU their waves halve each other's issue slots): the SQUARER
// runs sq <- sq^2 for every exponent bit, never waits, and publishes every square through global memory (a release counter); the
// MULTIPLIER consumes the squares of the set bits (acc <- acc * sq_i) as they appear.  Squarers have the lower workgroup ids, so
// they are resident before any multiplier starts to wait.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pz_internal.h"

typedef uint32_t u32;
typedef uint64_t u64;

template <int E> struct LD {
    u64 v[E];
};

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u64 bcast64(u64 x, unsigned src) {
    u32 lo = __builtin_amdgcn_readlane((u32)x, src);
    u32 hi = __builtin_amdgcn_readlane((u32)(x >> 32), src);
    return ((u64)hi << 32) | lo;
}
// one-lane shifts of the whole wave as DPP moves (wave_shl:1 / wave_shr:1, bound_ctrl: the lane without a
// source reads 0): a VALU op of a few cycles.  __shfl_down/__shfl_up compile to ds_bpermute_b32, whose LDS-crossbar
// latency (~100 cycles) sat on the critical path of every iteration of the systolic product.
__device__ __forceinline__ u32 dpp_down1(u32 x) { return __builtin_amdgcn_update_dpp(0u, x, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ u32 dpp_up1(u32 x) { return __builtin_amdgcn_update_dpp(0u, x, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ u64 shfl_down1(u64 x) {  // lane j <- lane j+1, lane 63 <- 0
    return ((u64)dpp_down1((u32)(x >> 32)) << 32) | dpp_down1((u32)x);
}
__device__ __forceinline__ u64 shfl_up1(u64 x) {  // lane j <- lane j-1, lane 0 <- 0
    return ((u64)dpp_up1((u32)(x >> 32)) << 32) | dpp_up1((u32)x);
}

template <int E> __device__ __forceinline__ LD<E> ld_zero() {
    LD<E> r;
#pragma unroll
    for (int e = 0; e < E; ++e) r.v[e] = 0;
    return r;
}
template <int E> __device__ __forceinline__ LD<E> ld_small(u64 x) {  // the integer x
    LD<E> r = ld_zero<E>();
    if (lane_id() == 0) r.v[0] = x;
    return r;
}
template <int E> __device__ __forceinline__ bool ld_is_zero(const LD<E>& a) {
    u64 o = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) o |= a.v[e];
    return __ballot(o != 0) == 0;
}
// load `limbs` u64 limbs from memory (zero extended to the 64*E capacity)
template <int E> __device__ __forceinline__ LD<E> ld_load(const u64* p, unsigned limbs) {
    LD<E> r;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        unsigned idx = lane_id() * E + e;
        r.v[e] = idx < limbs ? p[idx] : 0;
    }
    return r;
}
template <int E> __device__ __forceinline__ void ld_store(u64* p, const LD<E>& a, unsigned limbs) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        unsigned idx = lane_id() * E + e;
        if (idx < limbs) p[idx] = a.v[e];
    }
}
// true if any limb with index >= limbs is non-zero
template <int E> __device__ __forceinline__ bool ld_exceeds(const LD<E>& a, unsigned limbs) {
    bool bad = false;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        unsigned idx = lane_id() * E + e;
        bad |= (idx >= limbs) && (a.v[e] != 0);
    }
    return __ballot(bad) != 0;
}

// s = a + b + cin ; returns carry out.  Local ripple, then a wave-wide carry look-ahead: with
// per-lane generate g and (exclusive) propagate p, the carries of the 64-"bit" addition
// (G|P) + G + cin are exactly the inter-lane carries.
template <int E> __device__ __forceinline__ bool ld_add(LD<E>& s, const LD<E>& a, const LD<E>& b, bool cin) {
    u64 c = 0;
    bool allones = true;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        u64 t = a.v[e] + b.v[e];
        u64 c1 = t < a.v[e];
        u64 t2 = t + c;
        u64 c2 = t2 < t;
        s.v[e] = t2;
        c = c1 | c2;
        allones = allones && (t2 == ~0ull);
    }
    const bool g = c != 0;
    const bool p = allones && !g;
    const u64 G = __ballot(g), P = __ballot(p);
    const u64 X = G | P, Y = G;
    const u64 t = X + Y;
    const u64 o1 = t < X;
    const u64 t2 = t + (cin ? 1ull : 0ull);
    const u64 o2 = t2 < t;
    const u64 cmask = t2 ^ P;  // bit j = carry into lane j
    u64 ci = (cmask >> lane_id()) & 1ull;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        u64 t3 = s.v[e] + ci;
        ci = t3 < s.v[e];
        s.v[e] = t3;
    }
    return (o1 | o2) != 0;
}
// d = a - b ; returns borrow out (true if a < b)
template <int E> __device__ __forceinline__ bool ld_sub(LD<E>& d, const LD<E>& a, const LD<E>& b) {
    LD<E> nb;
#pragma unroll
    for (int e = 0; e < E; ++e) nb.v[e] = ~b.v[e];
    return !ld_add(d, a, nb, true);
}

__device__ __forceinline__ void mul64(u64 a, u64 b, u64& hi, u64& lo) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = (u64)a0 * b0;
    const u64 t = (u64)a0 * b1 + (p00 >> 32);
    const u64 u = (u64)a1 * b0 + (u32)t;
    lo = (u << 32) | (u32)p00;
    hi = (u64)a1 * b1 + (t >> 32) + (u >> 32);
}

// full product of two capacity-sized integers: lo = low 64*E limbs, hi = high 64*E limbs
#ifndef PZ_K3_UNROLL
#define PZ_K3_UNROLL 64  // full unroll: lane indices become immediates and the next round's broadcasts + products overlap the carry chain (52 -> 34 ms per 2048-bit encrypt)
#endif
template <int E> __device__ __noinline__ void ld_mul(LD<E>& lo, LD<E>& hi, const LD<E> a, const LD<E> b) {
    u64 col[E];
    u32 cc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        col[e] = 0;
        cc[e] = 0;
        lo.v[e] = 0;
    }
    const unsigned lane = lane_id();
#pragma unroll PZ_K3_UNROLL
    for (unsigned jj = 0; jj < 64; ++jj) {
#pragma unroll
        for (int ee = 0; ee < E; ++ee) {
            const u64 A = bcast64(a.v[ee], jj);
            u64 pend = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                u64 ph, pl;
                mul64(A, b.v[e], ph, pl);
                u64 t = col[e] + pl;
                cc[e] += t < pl;
                col[e] = t;
                if (e + 1 < E) {
                    u64 t2 = col[e + 1] + ph;
                    cc[e + 1] += t2 < ph;
                    col[e + 1] = t2;
                } else {
                    pend = ph;
                }
            }
            const u64 emit = bcast64(col[0], 0);
            if (lane == jj) lo.v[ee] = emit;
            // shift the column array one limb down
            // (a column's overflow count is consumed here, by the lane that holds it, into the next
            // column arriving in the same slot; it does not travel with the column)
            const u64 n0 = shfl_down1(col[0]);
            u64 ncol[E];
            u32 ncc[E];
#pragma unroll
            for (int e = 0; e + 1 < E; ++e) {
                u64 t = col[e + 1] + cc[e];
                ncc[e] = (t < col[e + 1]);
                ncol[e] = t;
            }
            {
                u64 t = n0 + pend;
                u32 k = (t < pend);
                u64 t2 = t + cc[E - 1];
                k += t2 < t;
                ncol[E - 1] = t2;
                ncc[E - 1] = k;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                col[e] = ncol[e];
                cc[e] = ncc[e];
            }
        }
    }
    // resolve: hi = col + (cc shifted up by one limb)
    LD<E> cv, sh;
#pragma unroll
    for (int e = 0; e < E; ++e) cv.v[e] = col[e];
    sh.v[0] = shfl_up1((u64)cc[E - 1]);
#pragma unroll
    for (int e = 1; e < E; ++e) sh.v[e] = cc[e - 1];
    (void)ld_add(hi, cv, sh, false);
}

// ---- the same product by a team of TEAM wavefronts (k_pow_mod_chain).  P_w = sum_{i in slice w} a_i * b * 2^(64 (i - i0_w)), i0_w =
// w * C / TEAM: C / TEAM low limbs (emitted one per outer iteration) + C high limbs (the column array once its deferred carries are
// resolved) -- exactly ld_mul's loop over a quarter of the outer limbs.  X = sum_w 2^(64 i0_w) P_w: the low limbs are disjoint
// (limb m belongs to slice m / (C / TEAM)); high limb t of P_w has weight i0_w + C / TEAM + t.
#define K3_TEAM 4
template <int E> struct TeamBuf {         // one buffer of a team's exchange area (two per team: double-buffered)
    u64 lo[64 * E];                       // emitted low limbs, by absolute limb index
    u64 sink[64 + 64 / K3_TEAM];          // where lanes 1..63 drop their copy of the per-iteration store (no exec-mask juggling)
    u64 row[K3_TEAM][2 * 64 * E];         // wave w's resolved high limbs AT THEIR WEIGHT: [(w + 1) C / TEAM, ... + C); the rest of a row
                                          // is zero from the kernel's start and never written, so the sum below reads unconditionally
};
template <int E> struct TeamCtx {
    TeamBuf<E>* buf;    // this team's two buffers
    unsigned w;         // this wave's index in the team
    unsigned parity;    // which buffer the next product uses (advanced by every call, identically in all waves of the team)
#ifdef PZ_K3_PROF
    unsigned long long t_loop = 0, t_res = 0, t_bar = 0, t_comb = 0, n_mul = 0;
#endif
};
#ifdef PZ_K3_PROF
#define K3_T0() const unsigned long long t0_ = clock64()
#define K3_T(acc, prev) const unsigned long long acc##_now = clock64(); T.acc += acc##_now - prev
#else
#define K3_T0()
#define K3_T(acc, prev)
#endif
// (forceinline, like mul_mod below: a TeamCtx / BarrettCtx whose address reaches a real call lives in scratch memory, and every
// T.parity / B.mu access of the callee becomes a private-memory round trip -- ~2000 clock units per product were exactly that)
// Wait states inside the hand-written multi-instruction asm blocks.  gfx940 / gfx950 require two wait states between a VALU
// instruction that writes an SGPR or VCC (v_readlane, v_add_co / v_addc_co carry-out) and a VALU instruction that reads it (LLVM's
// GCNHazardRecognizer: VALUWriteSGPRVALURead, the `s_nop 1` hipcc pads its own carry chains with on this target).  The recogniser
// does not look inside asm strings, so the blocks carry the `s_nop 1` themselves.  Measured round 4: identical traces without them on
// every box (the carry looked interlocked) and 12 % less time per step -- an empirical observation for a lone wave per SIMD, not an
// ISA guarantee, and in the pipelined prover K3 shares its CUs with K1 / K2 waves.  -DPZ_K3_NO_WAITSTATES builds the nop-free chain
// for measurement only.
#ifdef PZ_K3_NO_WAITSTATES
#define K3_WS ""
#else
#define K3_WS "s_nop 1\n\t"
#endif
template <int E> __device__ __forceinline__ void ld_mul_team(TeamCtx<E>& T, LD<E>& lo, LD<E>& hi, const LD<E> a, const LD<E> b) {
    constexpr unsigned C = 64 * E, SL = 64 / K3_TEAM;   // SL outer lanes (SL * E limbs) per wave
    u64 col[E];
    u32 cc[E];
    LD<E> plo;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        col[e] = 0;
        cc[e] = 0;
        plo.v[e] = 0;
    }
    const unsigned lane = lane_id();
    const unsigned j0 = T.w * SL;
    K3_T0();
    TeamBuf<E>& B = T.buf[T.parity & 1u];
    T.parity ^= 1u;
    if constexpr (E == 1) {
        // The same systolic step on 32-bit words with explicit carries: hipcc's code for the u64 version below is ~38 VALU instructions
        // per outer limb (64-bit additions as v_lshl_add_u64 at ~9.5 cycles, carries by 64-bit compares); this one is 5 mads + 2 DPP
        // moves + 2 readlanes + 6 carry instructions.  Column of this lane: (c1:c0) + c2 * 2^64; the finished low limb of every
        // iteration is lane 0's column, stored straight into the team's exchange area.
        const u32 b0 = (u32)b.v[0], b1 = (u32)(b.v[0] >> 32);
        const u32 a_lo = (u32)a.v[0], a_hi = (u32)(a.v[0] >> 32);
        // (1) the SL products A_jj * b of this wave's outer limbs: four mads each, independent of the column chain and of each other,
        // so they issue back to back (a mad's result is ~20 cycles away: chained through the column they made the loop latency-bound)
        u32 p0[SL], p1[SL], p2[SL], p3[SL];
        const u32 one = 1u;
#pragma unroll
        for (unsigned jl = 0; jl < SL; ++jl) {
            const unsigned jj = j0 + jl;
            const u32 A0 = __builtin_amdgcn_readlane(a_lo, jj), A1 = __builtin_amdgcn_readlane(a_hi, jj);
            u64 xx = 0;
            asm(K3_WS "v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(xx) : "s"(A0), "v"(b0) : "vcc");   // A0 / A1 come from v_readlane
            u64 y = xx >> 32;
            asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y) : "s"(A0), "v"(b1) : "vcc");
            u64 z = (u32)y;
            asm(K3_WS "v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(z) : "s"(A1), "v"(b0) : "vcc");
            u64 w = y >> 32;
            asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "s"(A1), "v"(b1) : "vcc");
            const u32 zh = (u32)(z >> 32);
            asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(zh), "v"(one) : "vcc");   // A * b < 2^128: no carry out
            p0[jl] = (u32)xx;
            p1[jl] = (u32)z;
            p2[jl] = (u32)w;
            p3[jl] = (u32)(w >> 32);
        }
        // (2) the column chain: additions only, every carry consumer behind its two wait states (K3_WS above)
        u32 c0 = 0, c1 = 0, c2 = 0;
        volatile u64* lo_dst = lane == 0 ? &B.lo[j0] : &B.sink[lane];   // one ds_write_b64 with an immediate offset per iteration
#pragma unroll
        for (unsigned jl = 0; jl < SL; ++jl) {
            asm("v_add_co_u32 %0, vcc, %0, %3\n\t" K3_WS "v_addc_co_u32 %1, vcc, %1, %4, vcc\n\t" K3_WS "v_addc_co_u32 %2, vcc, 0, %2, vcc"
                : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(p0[jl]), "v"(p1[jl]) : "vcc");
            lo_dst[jl] = ((u64)c1 << 32) | c0;     // lane 0: the finished low limb of this outer limb; other lanes: into the sink
            // shift one limb down: new column = column of lane + 1 (without its overflow count) + high half of the product + this
            // lane's overflow count
            u32 t0 = dpp_down1(c0), t1 = dpp_down1(c1), k = 0;
            asm("v_add_co_u32 %0, vcc, %0, %3\n\t" K3_WS "v_addc_co_u32 %1, vcc, %1, %4, vcc\n\t" K3_WS "v_addc_co_u32 %2, vcc, 0, %2, vcc"
                : "+v"(t0), "+v"(t1), "+v"(k) : "v"(p2[jl]), "v"(p3[jl]) : "vcc");
            asm("v_add_co_u32 %0, vcc, %0, %3\n\t" K3_WS "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t" K3_WS "v_addc_co_u32 %2, vcc, 0, %2, vcc"
                : "+v"(t0), "+v"(t1), "+v"(k) : "v"(c2) : "vcc");
            c0 = t0;
            c1 = t1;
            c2 = k;
        }
        const u64 x = ((u64)c1 << 32) | c0;
        col[0] = x;
        cc[0] = c2;
    } else {
#ifndef PZ_K3_TEAM_UNROLL
#define PZ_K3_TEAM_UNROLL 16
#endif
#pragma unroll PZ_K3_TEAM_UNROLL
    for (unsigned jl = 0; jl < SL; ++jl) {
        const unsigned jj = j0 + jl;    // wave-uniform: v_readlane takes it from an SGPR
#pragma unroll
        for (int ee = 0; ee < E; ++ee) {
            const u64 A = bcast64(a.v[ee], jj);
            u64 pend = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                u64 ph, pl;
                mul64(A, b.v[e], ph, pl);
                u64 t = col[e] + pl;
                cc[e] += t < pl;
                col[e] = t;
                if (e + 1 < E) {
                    u64 t2 = col[e + 1] + ph;
                    cc[e + 1] += t2 < ph;
                    col[e + 1] = t2;
                } else {
                    pend = ph;
                }
            }
            const u64 emit = bcast64(col[0], 0);
            if (lane == jj) plo.v[ee] = emit;
            const u64 n0 = shfl_down1(col[0]);
            u64 ncol[E];
            u32 ncc[E];
#pragma unroll
            for (int e = 0; e + 1 < E; ++e) {
                u64 t = col[e + 1] + cc[e];
                ncc[e] = (t < col[e + 1]);
                ncol[e] = t;
            }
            {
                u64 t = n0 + pend;
                u32 k = (t < pend);
                u64 t2 = t + cc[E - 1];
                k += t2 < t;
                ncol[E - 1] = t2;
                ncc[E - 1] = k;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                col[e] = ncol[e];
                cc[e] = ncc[e];
            }
        }
    }
    }
    K3_T(t_loop, t0_);
    // this wave's high limbs: col + (cc shifted up by one limb); cannot carry out (P_w < 2^(64 (C + SL E)))
    LD<E> cv, sh, ph_;
#pragma unroll
    for (int e = 0; e < E; ++e) cv.v[e] = col[e];
    sh.v[0] = shfl_up1((u64)cc[E - 1]);
#pragma unroll
    for (int e = 1; e < E; ++e) sh.v[e] = cc[e - 1];
    (void)ld_add(ph_, cv, sh, false);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (E > 1 && lane >= j0 && lane < j0 + SL) B.lo[lane * E + e] = plo.v[e];   // (E == 1: stored inside the loop)
        B.row[T.w][(T.w + 1) * SL * E + lane * E + e] = ph_.v[e];
    }
    K3_T(t_res, t_loop_now);
    __syncthreads();   // the team is the workgroup
    K3_T(t_bar, t_res_now);
    // every wave sums the partials: limb m = lane * E + e of the low half, m + C of the high half.  The 64-bit terms are added as two
    // 32-bit digits into 64-bit accumulators (a mad by one: no carry out to catch, one issue slot per digit), then the digits are
    // put back together: limb + an overflow count below 2^3, resolved by two look-ahead additions
    LD<E> s_lo, s_hi, c_lo, c_hi;
    const u32 one_ = 1u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned m = lane * E + e;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u64 d0 = 0, d1 = 0;
            if (half == 0) {
                const u64 v = B.lo[m];
                d0 = (u32)v;
                d1 = v >> 32;
            }
#pragma unroll
            for (unsigned w = 0; w < K3_TEAM; ++w) {
                const u64 v = B.row[w][half * C + m];
                const u32 v0 = (u32)v, v1 = (u32)(v >> 32);
                asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d0) : "v"(v0), "v"(one_) : "vcc");
                asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d1) : "v"(v1), "v"(one_) : "vcc");
            }
            const u64 mid = (d0 >> 32) + (u32)d1;
            const u64 limb = (u64)(u32)d0 | (mid << 32), cnt = (mid >> 32) + (d1 >> 32);
            if (half == 0) { s_lo.v[e] = limb; c_lo.v[e] = cnt; }
            else { s_hi.v[e] = limb; c_hi.v[e] = cnt; }
        }
    }
    // X = S + (carry counts shifted up one limb): low half, then the high half with the low half's carry out
    LD<E> k_lo, k_hi;
    const u64 top_cnt = bcast64(c_lo.v[E - 1], 63);    // count of the low half's top limb moves into the high half's limb 0
    k_lo.v[0] = shfl_up1(c_lo.v[E - 1]);
    k_hi.v[0] = shfl_up1(c_hi.v[E - 1]);
    if (lane == 0) k_hi.v[0] = top_cnt;
#pragma unroll
    for (int e = 1; e < E; ++e) {
        k_lo.v[e] = c_lo.v[e - 1];
        k_hi.v[e] = c_hi.v[e - 1];
    }
    const bool co = ld_add(lo, s_lo, k_lo, false);
    (void)ld_add(hi, s_hi, k_hi, co);
    K3_T(t_comb, t_bar_now);
#ifdef PZ_K3_PROF
    T.n_mul++;
#endif
}

// per-wave LDS scratch: 4*64*E u64 (a 2C-limb value twice)
template <int E> struct BarrettCtx {
    LD<E> M;      // modulus << s (top bit of the capacity set)
    LD<E> mu;     // floor(2^(2N)/M') - 2^N, N = 64*64*E
    unsigned s;   // normalisation shift in bits
    unsigned limbs;  // real limb count L of operands / results
    volatile u64* sm;  // this wave's LDS scratch, 2*64*E limbs
};

// y = (x_hi:x_lo) << s, as two halves; returns true if non-zero bits were shifted out
template <int E>
__device__ __forceinline__ bool shl_2c(const BarrettCtx<E>& B, LD<E>& ylo, LD<E>& yhi, const LD<E>& xlo, const LD<E>& xhi) {
    constexpr unsigned C = 64 * E;
    const unsigned ls = B.s >> 6, bs = B.s & 63;
    const unsigned lane = lane_id();
    if (E == 1 && ls == 0) {
        // a shift below one limb (a modulus that nearly fills the capacity: n^2 of a full-size key): neighbours by DPP, no LDS round trip
        if (bs == 0) {
            ylo = xlo;
            yhi = xhi;
            return false;
        }
        const u64 pl = shfl_up1(xlo.v[0]);
        u64 ph = shfl_up1(xhi.v[0]);
        const u64 top_lo = bcast64(xlo.v[0], 63), top_hi = bcast64(xhi.v[0], 63);
        if (lane == 0) ph = top_lo;
        ylo.v[0] = (xlo.v[0] << bs) | (pl >> (64 - bs));
        yhi.v[0] = (xhi.v[0] << bs) | (ph >> (64 - bs));
        return (top_hi >> (64 - bs)) != 0;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        B.sm[lane * E + e] = xlo.v[e];
        B.sm[C + lane * E + e] = xhi.v[e];
    }
    __builtin_amdgcn_wave_barrier();
    bool lost = false;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned t = h * C + lane * E + e;
            u64 cur = t >= ls ? B.sm[t - ls] : 0;
            u64 prv = t >= ls + 1 ? B.sm[t - ls - 1] : 0;
            u64 v = bs ? (cur << bs) | (prv >> (64 - bs)) : cur;
            if (h == 0) ylo.v[e] = v; else yhi.v[e] = v;
            // source limb t is shifted (partly) out if t + ls >= 2C, or its top bs bits if t + ls == 2C-1
            const u64 src = B.sm[t];
            if (t + ls >= 2 * C) lost |= src != 0;
            else if (t + ls == 2 * C - 1 && bs) lost |= (src >> (64 - bs)) != 0;
        }
    }
    __builtin_amdgcn_wave_barrier();
    return __ballot(lost) != 0;
}
// y = x >> s (single capacity-sized value)
template <int E> __device__ __forceinline__ LD<E> shr_c(const BarrettCtx<E>& B, const LD<E>& x) {
    constexpr unsigned C = 64 * E;
    const unsigned ls = B.s >> 6, bs = B.s & 63;
    const unsigned lane = lane_id();
    if (E == 1 && ls == 0) {   // see shl_2c
        if (bs == 0) return x;
        LD<E> y;
        y.v[0] = (x.v[0] >> bs) | (shfl_down1(x.v[0]) << (64 - bs));
        return y;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) B.sm[lane * E + e] = x.v[e];
    __builtin_amdgcn_wave_barrier();
    LD<E> y;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned t = lane * E + e;
        u64 cur = t + ls < C ? B.sm[t + ls] : 0;
        u64 nxt = t + ls + 1 < C ? B.sm[t + ls + 1] : 0;
        y.v[e] = bs ? (cur >> bs) | (nxt << (64 - bs)) : cur;
    }
    __builtin_amdgcn_wave_barrier();
    return y;
}
// y = x << s within the capacity (bits shifted out are dropped; used for the modulus only)
template <int E> __device__ __forceinline__ LD<E> shl_c(const BarrettCtx<E>& B, const LD<E>& x) {
    const unsigned ls = B.s >> 6, bs = B.s & 63;
    const unsigned lane = lane_id();
#pragma unroll
    for (int e = 0; e < E; ++e) B.sm[lane * E + e] = x.v[e];
    __builtin_amdgcn_wave_barrier();
    LD<E> y;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned t = lane * E + e;
        u64 cur = t >= ls ? B.sm[t - ls] : 0;
        u64 prv = t >= ls + 1 ? B.sm[t - ls - 1] : 0;
        y.v[e] = bs ? (cur << bs) | (prv >> (64 - bs)) : cur;
    }
    __builtin_amdgcn_wave_barrier();
    return y;
}

// bit length of a capacity-sized integer (0 for zero)
template <int E> __device__ __forceinline__ unsigned ld_bitlen(const LD<E>& a) {
    unsigned best = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (a.v[e]) best = (lane_id() * E + e) * 64 + (64 - __clzll(a.v[e]));
    }
    // max over lanes
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    return best;
}

// Barrett constants for modulus m (must be non-zero): returns false if m == 0
template <int E> __device__ __forceinline__ bool barrett_setup(BarrettCtx<E>& B, const LD<E> m) {   // (inlined: B stays in registers)
    constexpr unsigned N = 64 * 64 * E;
    const unsigned bl = ld_bitlen(m);
    if (bl == 0) return false;
    B.s = N - bl;
    B.M = shl_c(B, m);
    // R0 = 2^N - M' ; mu' = floor(2^N * R0 / M') by N steps of restoring division
    LD<E> R, zero = ld_zero<E>();
    (void)ld_sub(R, zero, B.M);  // wraps to 2^N - M'
    LD<E> mu = ld_zero<E>();
    LD<E> d;
    if (!ld_sub(d, R, B.M)) {
        // R0 >= M' only when M' == 2^(N-1): reciprocal would be 2^N, clamp to 2^N - 1
#pragma unroll
        for (int e = 0; e < E; ++e) mu.v[e] = ~0ull;
        B.mu = mu;
        return true;
    }
    for (unsigned i = 0; i < N; ++i) {
        // R <<= 1 (top bit out in `top`)
        u64 carry_in = shfl_up1(R.v[E - 1] >> 63);
        const bool top = (bcast64(R.v[E - 1], 63) >> 63) != 0;
#pragma unroll
        for (int e = E - 1; e >= 1; --e) R.v[e] = (R.v[e] << 1) | (R.v[e - 1] >> 63);
        R.v[0] = (R.v[0] << 1) | carry_in;
        const bool borrow = ld_sub(d, R, B.M);
        const bool bit = top || !borrow;
        if (bit) R = d;
        // mu = (mu << 1) | bit
        u64 mc = shfl_up1(mu.v[E - 1] >> 63);
#pragma unroll
        for (int e = E - 1; e >= 1; --e) mu.v[e] = (mu.v[e] << 1) | (mu.v[e - 1] >> 63);
        mu.v[0] = (mu.v[0] << 1) | mc;
        if (bit && lane_id() == 0) mu.v[0] |= 1ull;
    }
    B.mu = mu;
    return true;
}

enum { ST_OK = 0, ST_RANGE = 1, ST_ZERO_MOD = 2, ST_INTERNAL = 4, ST_TIMEOUT = 8 };

// the product routine of a mul_mod: one wave on its own (k_mul_mod) or a team of waves (k_pow_mod_chain)
template <int E> struct SoloMul {
    __device__ __forceinline__ void operator()(LD<E>& lo, LD<E>& hi, const LD<E>& a, const LD<E>& b) { ld_mul(lo, hi, a, b); }
};
template <int E> struct TeamMul {
    TeamCtx<E>* T;
    __device__ __forceinline__ void operator()(LD<E>& lo, LD<E>& hi, const LD<E>& a, const LD<E>& b) { ld_mul_team(*T, lo, hi, a, b); }
};
#ifdef PZ_K3_PROF
__device__ unsigned long long g_k3_phase[16];
#define K3_STAMP(i)                                                             \
    do {                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                      \
        const unsigned long long now_ = clock64();                              \
        __builtin_amdgcn_sched_barrier(0);                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_k3_phase[i] += now_ - last_; \
        last_ = now_;                                                           \
    } while (0)
#define K3_STAMP0() unsigned long long last_ = clock64()
#else
#define K3_STAMP(i)
#define K3_STAMP0()
#endif
// (q, r) = divmod(a*b, modulus).  Returns status bits.
template <int E, class Mul>
__device__ __forceinline__ unsigned mul_mod(const BarrettCtx<E>& B, LD<E>& q, LD<E>& r, const LD<E> a, const LD<E> b, Mul ld_mul) {
    unsigned st = ST_OK;
    K3_STAMP0();
    LD<E> xlo, xhi;
    ld_mul(xlo, xhi, a, b);
    K3_STAMP(0);
    LD<E> ylo, yhi;
    if (shl_2c(B, ylo, yhi, xlo, xhi)) st |= ST_RANGE;  // quotient cannot fit the capacity
    K3_STAMP(1);
    // qhat = yhi + hi(yhi * mu)
    LD<E> plo, phi;
    ld_mul(plo, phi, yhi, B.mu);
    K3_STAMP(2);
    LD<E> qh;
    if (ld_add(qh, yhi, phi, false)) st |= ST_RANGE;
    K3_STAMP(3);
    // r' = y - qhat*M'  (low capacity limbs + one limb above)
    LD<E> zlo, zhi;
    ld_mul(zlo, zhi, qh, B.M);
    K3_STAMP(4);
    LD<E> rr;
    const bool b0 = ld_sub(rr, ylo, zlo);
    K3_STAMP(5);
    u64 top = bcast64(yhi.v[0], 0) - bcast64(zhi.v[0], 0) - (b0 ? 1ull : 0ull);
    for (int it = 0; it < 8; ++it) {
        LD<E> d;
        const bool borrow = ld_sub(d, rr, B.M);
        if (top == 0 && borrow) break;  // 0 <= r' < M'
        if (it == 7) { st |= ST_INTERNAL; break; }
        rr = d;
        top -= borrow ? 1ull : 0ull;
        LD<E> z = ld_zero<E>();
        if (ld_add(qh, qh, z, true)) st |= ST_RANGE;
    }
    K3_STAMP(6);
    r = shr_c(B, rr);
    q = qh;
    if (ld_exceeds(q, B.limbs)) st |= ST_RANGE;
    K3_STAMP(7);
    return st;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
struct ChainDesc {
    const u64* modulus;  // limbs_mod limbs; if square_modulus: n (limbs_mod = Ln) and the modulus is n*n
    const u64* base;     // limbs_base limbs
    const u64* exp;      // exp_limbs limbs
    u64* steps;          // optional trace output (a|b|q|r per step, L limbs each)
    u64* result;         // L limbs
    u32* n_steps;        // optional
    u32* status;         // status bits (atomicOr)
    u32 limbs_mod, limbs_base, exp_limbs;
    u32 L;               // operand/result limb count (limbs of the modulus n^2)
    u32 square_modulus;
    u32 steps_cap;
    u32 uniform_bits;    // != 0: pow_mod with the exponent's bits IN the circuit: exactly this many
                         // bits, per bit the step (acc, sq) then the step (sq, sq); acc takes the product only if the bit is set
    u64* sq_buf;         // hand-off from the squarer to the multiplier: square i at sq_buf + i * L (max exponent bits x L limbs)
    u32* ready;          // number of squares published so far (zeroed by the host before the launch)
    u32 spin_limit;      // polls a multiplier wave makes for one square before it gives up (ST_TIMEOUT); K3_SPIN_LIMIT unless a test lowers it
    u32 test_no_publish; // test hook (PZ_K3_TEST_NO_PUBLISH): the squarer never advances `ready`, so the bounded wait is exercised
};
#define K3_SPIN_LIMIT (1u << 27)   // x >= 100 ns per poll: tens of seconds, far beyond any chain

template <int E> __global__ __launch_bounds__(64 * K3_TEAM) void k_pow_mod_chain(const ChainDesc* __restrict__ descs, unsigned n_chains,
                                                                                    u32* __restrict__ ticket) {
    constexpr unsigned C = 64 * E;
    __shared__ u64 s_scratch[K3_TEAM][2 * C];   // per-wave shift scratch
    __shared__ TeamBuf<E> s_team[2];            // [parity]
    __shared__ unsigned s_role, s_dead_step;
    // Roles by ARRIVAL, not by workgroup id: the first n workgroups that start running are the squarers of chains 0..n-1, the next n the
    // multipliers.  A multiplier only ever waits for the squarer of its chain, whose ticket is lower, i.e. which has already started
    // (and a squarer waits for nobody), so forward progress needs nothing from the dispatcher but that a workgroup which has started
    // keeps running -- no assumption on the order in which workgroup ids are dispatched, or on how many are resident.
    if (threadIdx.x == 0) {
        s_role = atomicAdd(ticket, 1u);
        s_dead_step = 0;
    }
    __syncthreads();
    const unsigned role = s_role;
    const bool squarer = role < n_chains;
    const ChainDesc D = descs[squarer ? role : role - n_chains];
#ifdef PZ_K3_PROF
    const unsigned long long k_t0 = clock64();
#endif
    const unsigned wave = threadIdx.x >> 6;
    TeamCtx<E> T;
    T.buf = s_team;
    T.w = wave;
    T.parity = 0;
    TeamMul<E> tmul{&T};
    const bool writer = wave == 0;   // the four waves of the team hold the same values: one of them writes
    BarrettCtx<E> B;
    B.sm = s_scratch[wave];
    B.limbs = D.L;
    B.s = 0;
    unsigned st = ST_OK;
    {   // the exchange rows are zero outside the ranges the products write (TeamBuf::row)
        u64* z = (u64*)s_team;
        for (unsigned t = threadIdx.x; t < sizeof(s_team) / 8; t += blockDim.x) z[t] = 0;
        __syncthreads();
    }
    LD<E> m = ld_load<E>(D.modulus, D.limbs_mod);
    if (D.square_modulus) {
        LD<E> lo, hi;
        tmul(lo, hi, m, m);
        m = lo;
    }
    const bool mod_ok = barrett_setup(B, m);
    if (!mod_ok) st |= ST_ZERO_MOD;
    // exponent bit length (uniform)
    unsigned nbits = 0;
    for (int i = (int)D.exp_limbs - 1; i >= 0; --i) {
        u64 w = D.exp[i];
        if (w) {
            nbits = (unsigned)i * 64 + (64 - __clzll(w));
            break;
        }
    }
    LD<E> sq = ld_load<E>(D.base, D.limbs_base);
    LD<E> acc = ld_small<E>(1);
    unsigned step_idx = 0;  // index of this bit's first step
    const bool uni = D.uniform_bits != 0;
    if (uni) nbits = D.uniform_bits;
    if (mod_ok) {
        for (unsigned i = 0; i < nbits; ++i) {
            const bool bit = (i >> 6) < D.exp_limbs && ((D.exp[i >> 6] >> (i & 63)) & 1);
            // reference schedule (pow_mod_fixed_exp): squaring step, then the multiply step on set bits;
            // uniform schedule (pow_mod): multiply step for EVERY bit, then the squaring step
            const unsigned sq_slot = uni ? step_idx + 1 : step_idx, mul_slot = uni ? step_idx : step_idx + 1;
            if (squarer) {
                // publish sq_i: the limbs leave as agent-scope stores BEFORE the step, the counter follows AFTER it with RELEASE
                // semantics at agent scope (the multiplier reads it and then fences with ACQUIRE: a textbook release / acquire pair, no
                // reliance on how gfx950 orders relaxed accesses).  Measured (profiles/r05_k3_ab_waitstates_release.txt): the release
                // costs 0.1 us of a 8.9-us step -- by the time the step's products are done the limb stores have long been
                // acknowledged, so its write-back / wait finds nothing in flight.  (-DPZ_K3_RELAXED_PUBLISH: round 4's relaxed store
                // behind s_waitcnt vmcnt(0), for measurement.)
                if (writer) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const unsigned idx = lane_id() * E + e;
                        if (idx < D.L) __hip_atomic_store(D.sq_buf + (size_t)i * D.L + idx, sq.v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                LD<E> q, r;
                st |= mul_mod(B, q, r, sq, sq, tmul);
                if (writer && !D.test_no_publish) {
#ifndef PZ_K3_RELAXED_PUBLISH
                    // every lane of the writer wave stored limbs: all of them wait for their own stores, lane 0 then releases the counter
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane_id() == 0) __hip_atomic_store(D.ready, i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane_id() == 0) __hip_atomic_store(D.ready, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                }
                if (writer && D.steps && sq_slot < D.steps_cap) {
                    u64* o = D.steps + (size_t)sq_slot * 4 * D.L;
                    ld_store(o, sq, D.L);
                    ld_store(o + D.L, sq, D.L);
                    ld_store(o + 2 * D.L, q, D.L);
                    ld_store(o + 3 * D.L, r, D.L);
                }
                sq = r;
            } else if (bit || uni) {
                // wait for square i (every wave polls for itself: no workgroup barrier on this path); the squarer never waits for
                // anyone and was dispatched first, so the wait is bounded by its progress
                // (bounded: ~2^27 polls of >= 100 ns are tens of seconds, far beyond any chain -- a wave must have an exit it reaches
                // whatever happens to the other workgroup; the step then runs on stale limbs and the status says so)
                for (unsigned spins = 0; __hip_atomic_load(D.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= i; ++spins) {
                    if (spins > D.spin_limit) {
                        st |= ST_TIMEOUT;
                        s_dead_step = i + 1;   // every wave of the team gives up at the END of this step (see below): same value from all
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                // acquire at agent scope: orders the limb loads below after the counter's (the compiler and the hardware both), and
                // invalidates what this CU may hold of the hand-off lines
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                LD<E> cur;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const unsigned idx = lane_id() * E + e;
                    cur.v[e] = idx < D.L ? __hip_atomic_load(D.sq_buf + (size_t)i * D.L + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                }
                LD<E> q, r;
                st |= mul_mod(B, q, r, acc, cur, tmul);
                if (writer && D.steps && mul_slot < D.steps_cap) {
                    u64* o = D.steps + (size_t)mul_slot * 4 * D.L;
                    ld_store(o, acc, D.L);
                    ld_store(o + D.L, cur, D.L);
                    ld_store(o + 2 * D.L, q, D.L);
                    ld_store(o + 3 * D.L, r, D.L);
                }
                if (bit) acc = r;   // select(bit, muled, acc)
                // a wave whose wait ran out has said so BEFORE this step's product barriers; every wave reads the flag after them,
                // so the whole team leaves at the same step (a wave leaving alone would hang the others at the next barrier)
                const unsigned dead = *(volatile unsigned*)&s_dead_step;
                if (dead != 0 && dead <= i + 1) {
                    st |= ST_TIMEOUT;
                    break;
                }
            }
            step_idx += (bit || uni) ? 2 : 1;
        }
    }
    if (D.steps && step_idx > D.steps_cap && !(st & ST_TIMEOUT)) st |= ST_INTERNAL;
    if (!squarer && writer && !(st & ST_TIMEOUT)) {   // after a timeout neither the result nor the step count is written
        ld_store(D.result, acc, D.L);
        if (D.n_steps && lane_id() == 0) *D.n_steps = step_idx;
    }
    if (st && lane_id() == 0) atomicOr(D.status, st);
#ifdef PZ_K3_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        printf("K3 phases (cycles per step, squarer wg 0): mul1 %llu shl %llu mul2 %llu add %llu mul3 %llu sub %llu fixup %llu shr %llu | whole kernel %llu per bit\n",
               g_k3_phase[0] / nbits, g_k3_phase[1] / nbits, g_k3_phase[2] / nbits, g_k3_phase[3] / nbits, g_k3_phase[4] / nbits, g_k3_phase[5] / nbits,
               g_k3_phase[6] / nbits, g_k3_phase[7] / nbits, (unsigned long long)(clock64() - k_t0) / nbits);
        for (int i = 0; i < 16; ++i) g_k3_phase[i] = 0;
    }
    if (lane_id() == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == n_chains))
        printf("K3 prof wg %u: products %llu; cycles per product: loop %llu, resolve+lds %llu, barrier %llu, combine %llu; total cycles %llu\n", blockIdx.x,
               T.n_mul, T.t_loop / T.n_mul, T.t_res / T.n_mul, T.t_bar / T.n_mul, T.t_comb / T.n_mul, (unsigned long long)(clock64() - k_t0));
#endif
}

struct MulDesc {
    const u64 *a, *b, *modulus;
    u64 *q, *r;
    u64* step;  // optional a|b|q|r record
    u32* status;
    u32 limbs_a, limbs_b, limbs_mod, L, square_modulus;
};

template <int E> __global__ __launch_bounds__(64) void k_mul_mod(const MulDesc* __restrict__ descs) {
    constexpr unsigned C = 64 * E;
    __shared__ u64 s_scratch[2 * C];
    const MulDesc D = descs[blockIdx.x];
    BarrettCtx<E> B;
    B.sm = s_scratch;
    B.limbs = D.L;
    B.s = 0;
    unsigned st = ST_OK;
    LD<E> m = ld_load<E>(D.modulus, D.limbs_mod);
    if (D.square_modulus) {
        LD<E> lo, hi;
        ld_mul(lo, hi, m, m);
        m = lo;
    }
    if (!barrett_setup(B, m)) {
        if (lane_id() == 0) atomicOr(D.status, (u32)ST_ZERO_MOD);
        return;
    }
    LD<E> a = ld_load<E>(D.a, D.limbs_a), b = ld_load<E>(D.b, D.limbs_b);
    LD<E> q, r;
    st |= mul_mod(B, q, r, a, b, SoloMul<E>());
    if (D.q) ld_store(D.q, q, D.L);
    if (D.r) ld_store(D.r, r, D.L);
    if (D.step) {
        ld_store(D.step, a, D.L);
        ld_store(D.step + D.L, b, D.L);
        ld_store(D.step + 2 * D.L, q, D.L);
        ld_store(D.step + 3 * D.L, r, D.L);
    }
    if (st && lane_id() == 0) atomicOr(D.status, st);
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static int status_to_rc(pz_ctx* ctx, u32 st) {
    if (st & ST_TIMEOUT) {
        snprintf(ctx->hip_err, sizeof ctx->hip_err, "big-integer kernel: a multiplier workgroup's bounded wait for its squarer ran out (status %u)", st);
        return PZ_ERR_INTERNAL;
    }
    if (st & ST_ZERO_MOD) return PZ_ERR_ZERO_MODULUS;
    if (st & ST_RANGE) return PZ_ERR_RANGE;
    if (st & ST_INTERNAL) {
        snprintf(ctx->hip_err, sizeof ctx->hip_err, "big-integer kernel: internal consistency check failed (status %u)", st);
        return PZ_ERR_HIP;
    }
    return PZ_OK;
}

// The hand-off area of a launch: per chain `bits` squares of L limbs; then ONE block of counters for the whole call, cleared by a single
// memset in stream order -- per launch a role ticket, per chain the number of squares published (a 64-byte line each: a counter is polled
// by the four waves of one multiplier while its squarer writes it; neighbours in one line would share that traffic).
// The squares of a batch are bounded: at most K3_HANDOFF_CAP bytes of them exist at a time, larger batches run as several launches
// one after the other on the stream (the multiplier consumes squares strictly in order, but a ring with a consumer counter would make
// the squarer wait for a workgroup that may not have started: that is the one dependency the role tickets exclude).
#define K3_HANDOFF_CAP ((size_t)1 << 30)
static size_t chain_squares_bytes(unsigned bits, unsigned L) { return (size_t)bits * L * 8; }
static unsigned env_u32(const char* name, unsigned dflt) {
    const char* v = getenv(name);
    return v && *v ? (unsigned)strtoul(v, nullptr, 0) : dflt;
}
// Test hooks (tests/test_gpu_kernels.py::test_k3_bounded_wait_reports_internal, test_encrypt_batch_beyond_residency): honoured only
// when PZ_K3_TEST_HOOKS=1 is set as well, and read ONCE PER CALL (chain_handoff_alloc), never inside the per-chain loop -- a stray
// PZ_K3_* variable in a deployment changes nothing, and the production path makes one getenv per entry point.
struct K3Hooks {
    unsigned spin_limit = K3_SPIN_LIMIT, test_no_publish = 0;
    size_t cap_kib = 0;
};
static K3Hooks k3_hooks() {
    K3Hooks h;
    const char* on = getenv("PZ_K3_TEST_HOOKS");
    if (!(on && on[0] == '1')) return h;
    h.spin_limit = env_u32("PZ_K3_SPIN_LIMIT", K3_SPIN_LIMIT);
    h.test_no_publish = env_u32("PZ_K3_TEST_NO_PUBLISH", 0);
    h.cap_kib = env_u32("PZ_K3_HANDOFF_CAP_KIB", 0);   // a small cap splits a small batch into several launches
    return h;
}
static size_t chains_per_launch(unsigned bits, unsigned L, size_t n_chains, size_t cap_kib) {
    size_t per = (cap_kib ? cap_kib << 10 : K3_HANDOFF_CAP) / chain_squares_bytes(bits, L);
    per = per < 2 ? 2 : per & ~(size_t)1;   // the two chains of an instance stay in one launch
    return per < n_chains ? per : n_chains;
}
struct ChainHandoff {
    char* squares;     // chains_per_launch x chain_squares_bytes, reused by every launch of the call
    char* counters;    // n_launches tickets, then n_chains ready counters, 64 bytes apart
    size_t per_launch, n_launches;
    K3Hooks hooks;
};
static int chain_handoff_alloc(pz_ctx* ctx, size_t n_chains, unsigned bits, unsigned L, ChainHandoff* h) {
    h->hooks = k3_hooks();
    h->per_launch = chains_per_launch(bits, L, n_chains, h->hooks.cap_kib);
    h->n_launches = (n_chains + h->per_launch - 1) / h->per_launch;
    const size_t sq = h->per_launch * chain_squares_bytes(bits, L), cnt = (h->n_launches + n_chains) * 64;
    void* d;
    PZCHK(pz_ws_get(ctx, WS_K3, sq + cnt + 256, &d));
    h->squares = (char*)d;
    h->counters = (char*)d + ((sq + 255) & ~(size_t)255);
    HIPCHK(ctx, hipMemsetAsync(h->counters, 0, cnt, ctx->stream));
    return PZ_OK;
}
static void chain_handoff_set(ChainDesc& d, const ChainHandoff& h, size_t i, unsigned bits, unsigned L) {
    d.sq_buf = (u64*)(h.squares + (i % h.per_launch) * chain_squares_bytes(bits, L));
    d.ready = (u32*)(h.counters + (h.n_launches + i) * 64);
    d.spin_limit = h.hooks.spin_limit;
    d.test_no_publish = h.hooks.test_no_publish;
}
static int launch_chains(pz_ctx* ctx, const ChainDesc* d_descs, size_t n, unsigned L, const ChainHandoff& h) {
    // per launch: 2 m workgroups for m chains; who squares and who multiplies is decided by arrival (k_pow_mod_chain)
    for (size_t l = 0, off = 0; off < n; ++l, off += h.per_launch) {
        const size_t m = n - off < h.per_launch ? n - off : h.per_launch;
        u32* ticket = (u32*)(h.counters + l * 64);
        if (L <= 64) hipLaunchKernelGGL(k_pow_mod_chain<1>, dim3((unsigned)(2 * m)), dim3(64 * K3_TEAM), 0, ctx->stream, d_descs + off, (unsigned)m, ticket);
        else hipLaunchKernelGGL(k_pow_mod_chain<2>, dim3((unsigned)(2 * m)), dim3(64 * K3_TEAM), 0, ctx->stream, d_descs + off, (unsigned)m, ticket);
        HIPCHK(ctx, hipGetLastError());
    }
    return PZ_OK;
}
static int launch_muls(pz_ctx* ctx, const MulDesc* d_descs, size_t n, unsigned L) {
    if (L <= 64) hipLaunchKernelGGL(k_mul_mod<1>, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_descs);
    else hipLaunchKernelGGL(k_mul_mod<2>, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_descs);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

extern "C" int pz_mul_mod(pz_ctx* ctx, uint32_t limbs, const uint64_t* a, const uint64_t* b, const uint64_t* modulus,
                          uint64_t* q, uint64_t* r) {
    if (!ctx || !a || !b || !modulus || !q || !r || limbs == 0) return PZ_ERR_INVALID;
    if (limbs > 128) return PZ_ERR_UNSUPPORTED;
    PZ_ENTER(ctx);
    const size_t lb = (size_t)limbs * 8;
    void* d;
    PZCHK(pz_ws_get(ctx, WS_BIG_A, 5 * lb + 256, &d));
    char* base = (char*)d;
    u32* d_status = (u32*)(base + 5 * lb);
    MulDesc* d_desc = (MulDesc*)(base + 5 * lb + 64);
    HIPCHK(ctx, hipMemcpyAsync(base, a, lb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(base + lb, b, lb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(base + 2 * lb, modulus, lb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d_status, 0, 4, ctx->stream));
    MulDesc h{};
    h.a = (const u64*)base;
    h.b = (const u64*)(base + lb);
    h.modulus = (const u64*)(base + 2 * lb);
    h.q = (u64*)(base + 3 * lb);
    h.r = (u64*)(base + 4 * lb);
    h.step = nullptr;
    h.status = d_status;
    h.limbs_a = h.limbs_b = h.limbs_mod = h.L = limbs;
    h.square_modulus = 0;
    HIPCHK(ctx, hipMemcpyAsync(d_desc, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    PZCHK(launch_muls(ctx, d_desc, 1, limbs));
    u32 st = 0;
    HIPCHK(ctx, hipMemcpyAsync(q, base + 3 * lb, lb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(r, base + 4 * lb, lb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&st, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return status_to_rc(ctx, st);
}

extern "C" int pz_paillier_trace(pz_ctx* ctx, uint32_t limbs_n2, const uint64_t* n2, const uint64_t* base,
                                 const uint64_t* exp, uint32_t exp_limbs, uint64_t* steps_out, size_t* n_steps,
                                 uint64_t* result) {
    if (!ctx || !n2 || !base || !exp || !result || limbs_n2 == 0 || exp_limbs == 0) return PZ_ERR_INVALID;
    if (steps_out && !n_steps) return PZ_ERR_INVALID;
    if (limbs_n2 > 128) return PZ_ERR_UNSUPPORTED;
    PZ_ENTER(ctx);
    const unsigned L = limbs_n2;
    const size_t lb = (size_t)L * 8, eb = (size_t)exp_limbs * 8;
    // steps needed = bits(exp) + popcount(exp)
    size_t need = 0;
    {
        int top = (int)exp_limbs - 1;
        while (top >= 0 && exp[top] == 0) --top;
        if (top >= 0) need = (size_t)top * 64 + (64 - __builtin_clzll(exp[top]));
        for (uint32_t i = 0; i < exp_limbs; ++i) need += (size_t)__builtin_popcountll(exp[i]);
    }
    const size_t cap = steps_out ? *n_steps : 0;
    if (steps_out && cap < need) return PZ_ERR_CAPACITY;
    void *d_in, *d_steps = nullptr;
    PZCHK(pz_ws_get(ctx, WS_BIG_A, 3 * lb + eb + 512, &d_in));
    if (steps_out && need) PZCHK(pz_ws_get(ctx, WS_BIG_B, need * 4 * lb, &d_steps));
    char* p = (char*)d_in;
    HIPCHK(ctx, hipMemcpyAsync(p, n2, lb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(p + lb, base, lb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(p + 2 * lb, exp, eb, hipMemcpyHostToDevice, ctx->stream));
    u32* d_status = (u32*)(p + 3 * lb + eb);
    u32* d_nsteps = d_status + 1;
    ChainDesc* d_desc = (ChainDesc*)(p + 3 * lb + eb + 64);
    HIPCHK(ctx, hipMemsetAsync(d_status, 0, 8, ctx->stream));
    ChainDesc h{};
    h.modulus = (const u64*)p;
    h.base = (const u64*)(p + lb);
    h.exp = (const u64*)(p + 2 * lb);
    h.steps = (u64*)d_steps;
    h.result = (u64*)(p + 2 * lb + eb);
    h.n_steps = d_nsteps;
    h.status = d_status;
    h.limbs_mod = L;
    h.limbs_base = L;
    h.exp_limbs = exp_limbs;
    h.L = L;
    h.square_modulus = 0;
    h.steps_cap = (u32)need;
    ChainHandoff hb;
    PZCHK(chain_handoff_alloc(ctx, 1, 64 * exp_limbs, L, &hb));
    chain_handoff_set(h, hb, 0, 64 * exp_limbs, L);
    HIPCHK(ctx, hipMemcpyAsync(d_desc, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    {
        pz_timer tm(ctx, PZ_T_TRACE);
        PZCHK(launch_chains(ctx, d_desc, 1, L, hb));
    }
    u32 st[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(result, p + 2 * lb + eb, lb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(st, d_status, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (steps_out && need)
        HIPCHK(ctx, hipMemcpyAsync(steps_out, d_steps, need * 4 * lb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_steps) *n_steps = st[1];
    return status_to_rc(ctx, st[0]);
}

static int encrypt_impl(pz_ctx* ctx, uint32_t Ln, size_t batch, const uint64_t* n, const uint64_t* g,
                        const uint64_t* m, const uint64_t* r, uint64_t* steps_out, int steps_on_device,
                        size_t steps_cap, uint32_t* n_steps_g, uint32_t* n_steps_r, uint64_t* c_out, uint32_t uniform_m_bits = 0) {
    if (!ctx || !n || !g || !m || !r || !c_out || Ln == 0 || batch == 0) return PZ_ERR_INVALID;
    if (uniform_m_bits > 64 * Ln) return PZ_ERR_INVALID;
    const unsigned L = 2 * Ln;
    if (L > 128) return PZ_ERR_UNSUPPORTED;
    if (steps_cap > 0x7fffffffu) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    const size_t nb = (size_t)Ln * 8, lb = (size_t)L * 8;
    // exact per-instance step counts from the exponents (host side: they are public structure of the
    // circuit, paillier.rs:50,54)
    std::vector<uint32_t> ng(batch), nr(batch);
    for (size_t i = 0; i < batch; ++i) {
        auto count = [&](const uint64_t* e) {
            size_t need = 0;
            int top = (int)Ln - 1;
            while (top >= 0 && e[top] == 0) --top;
            if (top >= 0) need = (size_t)top * 64 + (64 - __builtin_clzll(e[top]));
            for (uint32_t k = 0; k < Ln; ++k) need += (size_t)__builtin_popcountll(e[k]);
            return (uint32_t)need;
        };
        ng[i] = uniform_m_bits ? 2 * uniform_m_bits : count(m + i * Ln);
        nr[i] = count(n + i * Ln);
        if (uniform_m_bits) {   // the message must fit the bits the circuit decomposes
            for (uint32_t b = uniform_m_bits; b < 64 * Ln; ++b)
                if ((m[i * Ln + (b >> 6)] >> (b & 63)) & 1) return PZ_ERR_MESSAGE_RANGE;
        }
        if (steps_out && (size_t)ng[i] + nr[i] + 1 > steps_cap) return PZ_ERR_CAPACITY;
    }
    // device staging: inputs n,g,m,r (batch x Ln each), results gm, rn, c (batch x L), status, descs
    const size_t in_bytes = 4 * batch * nb;
    const size_t res_bytes = 3 * batch * lb;
    const size_t desc_bytes = batch * (2 * sizeof(ChainDesc) + sizeof(MulDesc));
    void* d;
    PZCHK(pz_ws_get(ctx, WS_BIG_A, in_bytes + res_bytes + desc_bytes + 1024, &d));
    char* p = (char*)d;
    u64* d_n = (u64*)p;
    u64* d_g = (u64*)(p + batch * nb);
    u64* d_m = (u64*)(p + 2 * batch * nb);
    u64* d_r = (u64*)(p + 3 * batch * nb);
    u64* d_gm = (u64*)(p + in_bytes);
    u64* d_rn = (u64*)(p + in_bytes + batch * lb);
    u64* d_c = (u64*)(p + in_bytes + 2 * batch * lb);
    u32* d_status = (u32*)(p + in_bytes + res_bytes);
    char* d_descs = p + in_bytes + res_bytes + 256;
    u64* d_steps = nullptr;
    if (steps_out) {
        if (steps_on_device) d_steps = steps_out;
        else {
            void* t;
            PZCHK(pz_ws_get(ctx, WS_BIG_B, batch * steps_cap * 4 * lb, &t));
            d_steps = (u64*)t;
        }
    }
    HIPCHK(ctx, hipMemcpyAsync(d_n, n, batch * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_g, g, batch * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_m, m, batch * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_r, r, batch * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d_status, 0, 4, ctx->stream));
    std::vector<ChainDesc> ch(2 * batch);
    std::vector<MulDesc> mu(batch);
    ChainHandoff hb;
    PZCHK(chain_handoff_alloc(ctx, 2 * batch, 64 * Ln, L, &hb));
    for (size_t i = 0; i < batch; ++i) {
        u64* st_i = d_steps ? d_steps + i * steps_cap * 4 * L : nullptr;
        ChainDesc& a = ch[2 * i];
        a = ChainDesc{};
        a.modulus = d_n + i * Ln;
        a.base = d_g + i * Ln;
        a.exp = d_m + i * Ln;
        a.steps = st_i;
        a.result = d_gm + i * L;
        a.n_steps = nullptr;
        a.status = d_status;
        a.limbs_mod = Ln;
        a.limbs_base = Ln;
        a.exp_limbs = Ln;
        a.L = L;
        a.square_modulus = 1;
        a.steps_cap = ng[i];
        a.uniform_bits = uniform_m_bits;
        ChainDesc& b = ch[2 * i + 1];
        b = a;
        b.uniform_bits = 0;   // n is the public key: its bits stay circuit structure (pow_mod_fixed_exp)
        b.base = d_r + i * Ln;
        b.exp = d_n + i * Ln;
        b.steps = st_i ? st_i + (size_t)ng[i] * 4 * L : nullptr;
        b.result = d_rn + i * L;
        b.steps_cap = nr[i];
        chain_handoff_set(a, hb, 2 * i, 64 * Ln, L);
        chain_handoff_set(b, hb, 2 * i + 1, 64 * Ln, L);
        MulDesc& f = mu[i];
        f = MulDesc{};
        f.a = d_gm + i * L;
        f.b = d_rn + i * L;
        f.modulus = d_n + i * Ln;
        f.q = nullptr;
        f.r = d_c + i * L;
        f.step = st_i ? st_i + (size_t)(ng[i] + nr[i]) * 4 * L : nullptr;
        f.status = d_status;
        f.limbs_a = f.limbs_b = L;
        f.limbs_mod = Ln;
        f.L = L;
        f.square_modulus = 1;
    }
    ChainDesc* d_ch = (ChainDesc*)d_descs;
    MulDesc* d_mu = (MulDesc*)(d_descs + 2 * batch * sizeof(ChainDesc));
    HIPCHK(ctx, hipMemcpyAsync(d_ch, ch.data(), ch.size() * sizeof(ChainDesc), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_mu, mu.data(), mu.size() * sizeof(MulDesc), hipMemcpyHostToDevice, ctx->stream));
    {
        pz_timer tm(ctx, PZ_T_TRACE);
        PZCHK(launch_chains(ctx, d_ch, 2 * batch, L, hb));
        PZCHK(launch_muls(ctx, d_mu, batch, L));
    }
    u32 st = 0;
    HIPCHK(ctx, hipMemcpyAsync(c_out, d_c, batch * lb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&st, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (steps_out && !steps_on_device)
        for (size_t i = 0; i < batch; ++i) {
            size_t cnt = (size_t)ng[i] + nr[i] + 1;
            HIPCHK(ctx, hipMemcpyAsync(steps_out + i * steps_cap * 4 * L, d_steps + i * steps_cap * 4 * L,
                                       cnt * 4 * lb, hipMemcpyDeviceToHost, ctx->stream));
        }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < batch; ++i) {
        if (n_steps_g) n_steps_g[i] = ng[i];
        if (n_steps_r) n_steps_r[i] = nr[i];
    }
    return status_to_rc(ctx, st);
}

extern "C" int pz_paillier_encrypt(pz_ctx* ctx, uint32_t limbs_n, size_t batch, const uint64_t* n, const uint64_t* g,
                                   const uint64_t* m, const uint64_t* r, uint64_t* steps_out, size_t steps_cap,
                                   uint32_t* n_steps_g, uint32_t* n_steps_r, uint64_t* c_out) {
    return encrypt_impl(ctx, limbs_n, batch, n, g, m, r, steps_out, 0, steps_cap, n_steps_g, n_steps_r, c_out);
}
extern "C" int pz_paillier_encrypt_dev(pz_ctx* ctx, uint32_t limbs_n, size_t batch, const uint64_t* n,
                                       const uint64_t* g, const uint64_t* m, const uint64_t* r, uint64_t* d_steps_out,
                                       size_t steps_cap, uint32_t* n_steps_g, uint32_t* n_steps_r, uint64_t* c_out) {
    return encrypt_impl(ctx, limbs_n, batch, n, g, m, r, d_steps_out, 1, steps_cap, n_steps_g, n_steps_r, c_out);
}

// the UNIFORM-SHAPE encrypt circuit's witness.  The reference pulls the exponent m out of the
// witness (paillier.rs:50: pow_mod_fixed_exp), so its circuit -- hence vk / pk -- is shaped by the secret message; here
// g^m runs as BigUintChip::pow_mod over exactly `m_bits` in-circuit bits (per bit: mul_mod(acc, sq), select, square_mod),
// so every message of a key shares one circuit shape: 2 * m_bits steps for g^m, then the r^n chain (n is public: fixed
// exponent as in the reference) and the final mul_mod.  `batch` independent messages per call.  Not bit-compatible with
// the reference's circuit by construction (a different constraint system proving the same statement).
extern "C" int pz_paillier_encrypt_uniform(pz_ctx* ctx, uint32_t limbs_n, size_t batch, uint32_t m_bits, const uint64_t* n,
                                           const uint64_t* g, const uint64_t* m, const uint64_t* r, uint64_t* steps_out,
                                           size_t steps_cap, uint32_t* n_steps_g, uint32_t* n_steps_r, uint64_t* c_out) {
    if (m_bits == 0) return PZ_ERR_INVALID;
    return encrypt_impl(ctx, limbs_n, batch, n, g, m, r, steps_out, 0, steps_cap, n_steps_g, n_steps_r, c_out, m_bits);
}
extern "C" int pz_paillier_encrypt_uniform_dev(pz_ctx* ctx, uint32_t limbs_n, size_t batch, uint32_t m_bits, const uint64_t* n,
                                               const uint64_t* g, const uint64_t* m, const uint64_t* r, uint64_t* d_steps_out,
                                               size_t steps_cap, uint32_t* n_steps_g, uint32_t* n_steps_r, uint64_t* c_out) {
    if (m_bits == 0) return PZ_ERR_INVALID;
    return encrypt_impl(ctx, limbs_n, batch, n, g, m, r, d_steps_out, 1, steps_cap, n_steps_g, n_steps_r, c_out, m_bits);
}
