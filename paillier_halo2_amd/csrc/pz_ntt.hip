// pz_ntt.hip -- K2: radix-2 NTT over BN254 Fr, natural order in and out (== halo2curves best_fft,
// the routine behind EvaluationDomain::{fft, ifft, coeff_to_extended, extended_to_coeff} that
// /root/reference/src/bench.rs:161-171 reaches through create_proof).
//
// Decomposition (n = n1*n2 or n1*n2*n3, every factor <= 2^9): each pass does the small DFTs of
// one factor entirely inside LDS (radix-2 DIT stages between __syncthreads), so a column crosses
// HBM once per pass -- 2 passes up to 2^18, 3 up to 2^27:
//   pass A/B ("strided"): view [hi][R][lo], DFT along R (stride lo), multiply by the inter-pass
//                         twiddle omega^(tw_mul*l*k), write back in the same layout.  A block owns
//                         T consecutive l, so every global access is a T*32-byte run.
//   pass C  ("final")   : view [k1][k2][R], DFT along the contiguous R, write transposed to
//                         k1 + n1*k2 + n1*n2*k so the result lands in natural order; a block
//                         owns T consecutive k1 for one k2, so stores are T*32-byte runs.
// Algorithmic HBM bytes: 64 B per element (one 32 B read + one 32 B write), see DESIGN.md.
#include <stdlib.h>

#include <utility>
#include <vector>

#include "fp.cuh"
#include "fp29.cuh"
#include "pz_internal.h"

struct NttPass {
    unsigned logR;      // DFT size of this pass = 1 << logR
    size_t lo;          // contiguous inner extent (1 for the final pass)
    size_t hi;          // outer extent
    size_t tw_mul;      // inter-pass twiddle exponent multiplier (strided passes)
    size_t n;           // total transform size
    unsigned T;         // tile width
    size_t n1, n2;      // final pass: hi = n1*n2, output index = k1 + n1*k2 + hi*k
    unsigned swap;      // 1: grid.x = column, grid.y = tile (set by the launch helpers)
};

__device__ __forceinline__ unsigned bitrev32(unsigned x, unsigned bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

extern __shared__ __attribute__((aligned(16))) unsigned char pz_smem[];


// ------------------------------------------------------------------------------------------------
// The transforms on the reduced-radix field of fp29.cuh (9 x 29-bit limbs).  Data stays in the ABI's 2^256 Montgomery domain from
// load to store.  [round 5] EVERY product of a transform is by a constant known in advance (twiddle, coset power, scale), so the
// tables hold constant PAIRS (c, floor(c 2^261 / p): 72 B per entry, pz_get_pow_table_raw) and the kernels multiply with f29_mulc
// (Barrett / Shoup with a precomputed quotient: 143 multiplier instructions instead of the Montgomery product's 180; x * c in x's own
// domain, no Montgomery factor).
//   LDS tile: 9 words per element (36 B, odd word stride: conflict-free for consecutive elements).
//   Tile elements stay UNCARRIED between stages (limbs < 2^31.3: f29_mulc takes 2^31.8 x 2^29 limbs); only the two
//   operands of a radix-4 group that are added without being multiplied take one parallel carry round.  A product is tight and
//   below 3p, a subtraction adds 4p: values grow by at most 8p per pair of stages (14p in the first pair, whose operands come
//   straight from the load) -- below 46p after 9 stages, far from 2^261 = 169p, so nothing is reduced inside a pass.  A pass ends in
//   a product wherever the algorithm has one (inter-pass twiddle, post scale: below 3p) and in the quotient-estimate
//   canonicalisation (values below 64p) otherwise.
// ------------------------------------------------------------------------------------------------
typedef F29<FrTag> Fr29;
#ifndef NTT_WAVES
#define NTT_WAVES 4   // waves per SIMD the kernels are compiled for (128 VGPRs): with the 18-limb constant pairs hipcc otherwise takes 135
#endif
#define NTT_BOUNDS __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NTT_WAVES, NTT_WAVES)))

#ifndef NTT_LB
#define NTT_LB 2u    // elements a thread loads per batch of outstanding global loads (4 before the 72-byte constant pairs: the batch held 4 x 18 + 4 x 8 registers; measured 48.6 vs 49.3 us per polynomial)
#endif
// a tile element to HBM.  CANON: the ABI's canonical representative (value below 2^(J+1) p).  Otherwise the packed 256-bit
// integer as it is (tight limbs, below 2p after a product): the ping-pong buffer between two passes never crosses the ABI,
// and the next pass takes any tight value below 2p
template <bool CANON, unsigned J> __device__ __forceinline__ void ntt_store(Fp<FrTag>* p, const F29<FrTag>& x) {
    if (CANON) {
        f29_store<J>(p, x);
    } else {
        u32 w[8];
        f29_pack(x, w);
        uint4* q = reinterpret_cast<uint4*>(p);
        q[0] = make_uint4(w[0], w[1], w[2], w[3]);
        q[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// the last pass's store of a value that has no product in front of it (uncarried limbs, below 32p: the inter-pass product left
// it below 2p and every pair of stages adds at most 4p): canonical through the quotient estimate (fp29.cuh::f29_canon_q)
__device__ __forceinline__ void ntt_store_q(Fp<FrTag>* p, const F29<FrTag>& x, const u32* __restrict__ qtab) {
    const F29<FrTag> c = f29_canon_q(x, qtab);
    u32 w[8];
    f29_pack(c, w);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// words of an LDS tile of `elems` elements (pad element per 32 + one): the quotient table of the final kernels sits behind it
__device__ __host__ __forceinline__ unsigned ntt29_lds_words(unsigned elems) { return (elems + (elems >> 5) + 1u) * 9u; }

// tile index -> LDS word offset: 9 words per element and ONE PAD ELEMENT PER 32, which spreads the power-of-two
// element strides of the bit-reversed fill and of the first pair of stages over all banks (without it 69 % of the
// LDS cycles of these kernels were bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/)
__device__ __forceinline__ unsigned lds29_off(unsigned e) { return (e + (e >> 5)) * 9u; }   // tile indices are < 2^16: 32-bit arithmetic
__device__ __forceinline__ Fr29 lds29_get(const u32* sm, unsigned e) {
    Fr29 r;
    const u32* p = sm + lds29_off(e);
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = p[i];
    return r;
}
__device__ __forceinline__ void lds29_put(u32* sm, unsigned e, const Fr29& a) {
    u32* p = sm + lds29_off(e);
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = a.v[i];
}
// entry idx of a constant-pair table (18 x 29-bit limbs per entry, 72 B: pz_get_pow_table_raw): no unpacking
__device__ __forceinline__ C18 raw9_load(const u32* __restrict__ tab, size_t idx) {
    C18 r;
    const u32* p = tab + idx * 18u;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.w[i] = p[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) r.q[i] = p[9 + i];
    return r;
}
typedef C18 Raw9;
// x * c (mod p, below 3p, tight) for a table constant
__device__ __forceinline__ Fr29 mulc(const Fr29& x, const C18& c) {
    Fr29 w, q;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        w.v[i] = c.w[i];
        q.v[i] = c.q[i];
    }
    return f29_mulc(x, w, q);
}
__device__ __forceinline__ Fr29 mulc(const Fr29& x, const u32* __restrict__ tab, size_t idx) { return mulc(x, raw9_load(tab, idx)); }
// radix-2 butterfly on carried inputs (u any value, v the already multiplied, tight operand below 3p)
__device__ __forceinline__ void bf29(const Fr29& u, const Fr29& v, Fr29& sum, Fr29& diff) {
    sum = f29_add(u, v);
    diff = f29_sub<4, 29>(u, v);
}

// logR radix-2 DIT stages over an LDS tile holding T independent transforms, taken TWO STAGES AT A TIME: a thread
// owns the four elements {base, base+h, base+2h, base+3h} of a radix-4 group, does both layers in registers (same
// four products as two radix-2 layers) and the tile crosses LDS and a barrier once per pair of stages instead of
// once per stage.  Stages 0+1 cost one product per group (their other twiddles are 1).  An odd logR ends with one
// plain radix-2 stage.  element (j, t) lives at tile index j*sr + t*st.  Input must be stored bit-reversed in j.
// stw != nullptr: the stage twiddles come from a per-transform table instead of the omega table -- entry (2^s - 1) + pos of
// stage s holds K_s * omega^(pos * n / 2^(s+1)) with a constant K_s per stage (the coset shift of the extended transform's first
// pass absorbed into the stages, see get_ext_abs_tables): every stage then has real twiddles, stages 0 + 1 included.
__device__ __forceinline__ void lds_dit29(u32* sm, unsigned logR, unsigned T, unsigned sr, unsigned st, bool t_fastest,
                                          const u32* __restrict__ tw, size_t n_, const u32* __restrict__ stw = nullptr) {
    const unsigned n = (unsigned)n_;   // n <= 2^27: twiddle indices fit 32 bits
    const unsigned R = 1u << logR;
    const unsigned logT = 31u - (unsigned)__builtin_clz(T);
    unsigned s = 0;
    for (; s + 1 < logR; s += 2) {
        const unsigned h = 1u << s;
        const unsigned nq = (R >> 2) * T;
        for (unsigned id = threadIdx.x; id < nq; id += blockDim.x) {
            unsigned t, q;
            if (t_fastest) {
                t = id & (T - 1);
                q = id >> logT;
            } else {
                q = id & ((R >> 2) - 1);
                t = id >> (logR - 2);
            }
            const unsigned pos = q & (h - 1);
            const unsigned base = ((q >> s) << (s + 2)) + pos;
            const unsigned e0 = base * sr + t * st, dh = h * sr;
            Fr29 a0 = lds29_get(sm, e0), a1 = lds29_get(sm, e0 + dh), a2 = lds29_get(sm, e0 + 2 * dh), a3 = lds29_get(sm, e0 + 3 * dh);
            Fr29 b0, b1, b2, b3, o0, o1, o2, o3;
            if (s || stw) {  // kernel-uniform branch
                // tile elements arrive with limbs < 2^31.3 (uncarried outputs of the previous pair of stages): legal as
                // they are for the operands that get multiplied (a1, a3); the two that are only added (a0, a2) take one
                // parallel carry round each
                a0 = f29_carry(a0);
                a2 = f29_carry(a2);
                const C18 w = stw ? raw9_load(stw, (h - 1u) + pos) : raw9_load(tw, pos * (n >> (s + 1)));
                a1 = mulc(a1, w);
                a3 = mulc(a3, w);
                bf29(a0, a1, b0, b1);
                bf29(a2, a3, b2, b3);
                b2 = stw ? mulc(b2, stw, (2u * h - 1u) + pos) : mulc(b2, tw, pos * (n >> (s + 2)));
                b3 = stw ? mulc(b3, stw, (2u * h - 1u) + pos + h) : mulc(b3, tw, (pos + h) * (n >> (s + 2)));
                bf29(b0, b2, o0, o2);
            } else {
                // stages 0 + 1: operands straight from the load (tight, value < 3p); the first layer's twiddles and the
                // second layer's even twiddle are 1
                b0 = f29_add(a0, a1);
                b1 = f29_sub<4, 29>(a0, a1);
                b2 = f29_add(a2, a3);          // un-multiplied: limbs < 2^30, value < 6p -> subtrahend of f29_sub<8, 30>
                b3 = f29_sub<4, 29>(a2, a3);
                b3 = mulc(b3, tw, (pos + h) * (n >> (s + 2)));
                o0 = f29_add(b0, b2);
                o2 = f29_sub<8, 30>(b0, b2);
            }
            bf29(b1, b3, o1, o3);
            // outputs stay uncarried: limbs < 2^31.3, values grow by at most 8p per pair of stages (14p in the first)
            lds29_put(sm, e0, o0);
            lds29_put(sm, e0 + 2 * dh, o2);
            lds29_put(sm, e0 + dh, o1);
            lds29_put(sm, e0 + 3 * dh, o3);
        }
        __syncthreads();
    }
    if (s < logR) {
        const unsigned half = 1u << s;
        const unsigned nbf = (R >> 1) * T;
        for (unsigned id = threadIdx.x; id < nbf; id += blockDim.x) {
            unsigned t, bf;
            if (t_fastest) {
                t = id & (T - 1);
                bf = id >> logT;
            } else {
                bf = id & ((R >> 1) - 1);
                t = id >> (logR - 1);
            }
            const unsigned pos = bf & (half - 1);
            const unsigned i0 = ((bf >> s) << (s + 1)) + pos;
            const unsigned e0 = i0 * sr + t * st, e1 = e0 + half * sr;
            Fr29 u = lds29_get(sm, e0);
            Fr29 v = lds29_get(sm, e1);
            Fr29 o0, o1;
            if (s || stw) {
                u = f29_carry(u);
                v = stw ? mulc(v, stw, (half - 1u) + pos) : mulc(v, tw, pos * (n >> (s + 1)));
                bf29(u, v, o0, o1);
            } else {  // a single stage (logR == 1): operands straight from the load (tight, < 3p)
                o0 = f29_add(u, v);
                o1 = f29_sub<4, 29>(u, v);
            }
            lds29_put(sm, e0, o0);
            lds29_put(sm, e1, o1);
        }
        __syncthreads();
    }
}

// strided pass.  grid.x = hi * (lo / T), grid.y = column.  tw, pre: 261-domain tables
// PRE: 0 none; 1 every loaded element times a pre-scale table entry; 2 the coset shift ABSORBED (extended transform's first
// pass): no product at the load, stage twiddles from stw0 (per coset, (1 << logR) entries apart), and pre0 holds the fused
// inter-pass table c^col * omega^(col * k) indexed [k * lo + col] instead of the pre-scale table
template <int PRE>
__global__ NTT_BOUNDS void k_ntt_strided29(const Fr* in, Fr* out, size_t in_stride, size_t out_stride,
                                                       NttPass p, const u32* __restrict__ tw,
                                                       const u32* __restrict__ pre0, unsigned n_r, size_t pre_r_stride,
                                                       size_t out_r_stride, const u32* __restrict__ stw0,
                                                       const u32* __restrict__ tw_ep) {   // tw_ep: the omega table of the inter-pass product (may carry a scale)
    u32* sm = reinterpret_cast<u32*>(pz_smem);
    const unsigned R = 1u << p.logR, T = p.T;
    const unsigned logT = 31u - (unsigned)__builtin_clz(T);
    const size_t tiles = p.lo >> logT;   // lo, T powers of two
    // columns of a batch share the pre-scale and twiddle tables: with the column in grid.x (dispatched fastest) the
    // same tile of every column runs back to back and the table rows it needs stay in L2
    const size_t bx = p.swap ? blockIdx.y : blockIdx.x, by = p.swap ? blockIdx.x : blockIdx.y;
    const size_t h = bx / tiles, lt = bx % tiles;
    const Fr* src = in + by * in_stride;
    Fr* dst = out + by * out_stride;
    const size_t base = h * R * p.lo + lt * T;
    // global loads are issued NTT_LB elements at a time before anything waits on them (one exposed memory round trip per
    // batch instead of two per element: the load, then the pre-scale table row)
    const unsigned nelem = R * T;
    // n_r > 1: the first pass of the coset-extended transform -- the SAME input tile, pre-scaled by n_r different tables
    // (one per coset) into n_r outputs; the tile is re-read from L2, not from HBM, and one launch does the work of n_r
    Fr* const dst0 = dst;
    for (unsigned rr = 0; rr < n_r; ++rr) {
    const u32* __restrict__ pre = pre0 + (size_t)rr * pre_r_stride * 18u;
    dst = dst0 + (size_t)rr * out_r_stride;
    if (rr) __syncthreads();   // the previous coset's stores have read the tile
    for (unsigned i0 = 0; i0 < nelem; i0 += NTT_LB * 256u) {
        Fr raw[NTT_LB];
        Raw9 praw[NTT_LB];
#pragma unroll
        for (unsigned k = 0; k < NTT_LB; ++k) {
            const unsigned idx = (i0 + k * 256u + threadIdx.x) & (nelem - 1);   // lanes past the tile re-read one of its elements: no branch around the loads
            const size_t g = base + (size_t)(idx >> logT) * p.lo + (idx & (T - 1));
            raw[k] = fp_load<FrTag>(src + g);
            if (PRE == 1) praw[k] = raw9_load(pre, g);
        }
#pragma unroll
        for (unsigned k = 0; k < NTT_LB; ++k) {
            const unsigned idx = i0 + k * 256u + threadIdx.x;
            if (idx < nelem) {
                Fr29 x = f29_from_fp(raw[k]);
                if (PRE == 1) x = mulc(x, praw[k]);
                lds29_put(sm, bitrev32(idx >> logT, p.logR) * T + (idx & (T - 1)), x);
            }
        }
    }
    __syncthreads();
    lds_dit29(sm, p.logR, T, T, 1, true, tw, p.n, PRE == 2 ? stw0 + (size_t)rr * R * 18u : nullptr);
    for (unsigned i0 = 0; i0 < nelem; i0 += NTT_LB * 256u) {
        Raw9 traw[NTT_LB];
#pragma unroll
        for (unsigned k = 0; k < NTT_LB; ++k) {
            const unsigned idx = (i0 + k * 256u + threadIdx.x) & (nelem - 1);
            if (PRE == 2) traw[k] = raw9_load(pre, (size_t)(idx >> logT) * p.lo + (lt * T + (idx & (T - 1))));
            else traw[k] = raw9_load(tw_ep, p.tw_mul * (lt * T + (idx & (T - 1))) * (idx >> logT));  // exponent < n by construction
        }
#pragma unroll
        for (unsigned k = 0; k < NTT_LB; ++k) {
            const unsigned idx = i0 + k * 256u + threadIdx.x;
            if (idx < nelem) {
                const unsigned kk = idx >> logT, t = idx & (T - 1);
                // always through the product (tw[0] = 1): the result is below 3p (< 2^256: it packs) whatever the stages accumulated
                const Fr29 x = mulc(lds29_get(sm, kk * T + t), traw[k]);
                ntt_store<false, 1>(dst + base + (size_t)kk * p.lo + t, x);
            }
        }
    }
    }
}

// final pass.  grid.x = n2 * (n1 / T), grid.y = column
template <bool PRE>
__global__ NTT_BOUNDS void k_ntt_final29(const Fr* in, Fr* out, size_t in_stride, size_t out_stride,
                                                     NttPass p, const u32* __restrict__ tw, const u32* __restrict__ pre,
                                                     C18 post, int has_post) {
    u32* sm = reinterpret_cast<u32*>(pz_smem);
    const unsigned R = 1u << p.logR, T = p.T;
    const unsigned logT = 31u - (unsigned)__builtin_clz(T);
    const size_t tiles = p.n1 >> logT;
    const size_t bx = p.swap ? blockIdx.y : blockIdx.x, by = p.swap ? blockIdx.x : blockIdx.y;
    const size_t k2 = bx / tiles, k1_0 = (bx % tiles) * T;
    const Fr* src = in + by * in_stride;
    Fr* dst = out + by * out_stride;
    const unsigned nelem = R * T;
    for (unsigned i0 = 0; i0 < nelem; i0 += NTT_LB * 256u) {
        Fr raw[NTT_LB];
        Raw9 praw[NTT_LB];
#pragma unroll
        for (unsigned k = 0; k < NTT_LB; ++k) {
            const unsigned idx = (i0 + k * 256u + threadIdx.x) & (nelem - 1);
            const size_t g = ((k1_0 + (idx >> p.logR)) * p.n2 + k2) * R + (idx & (R - 1));
            raw[k] = fp_load<FrTag>(src + g);
            if (PRE) praw[k] = raw9_load(pre, g);
        }
#pragma unroll
        for (unsigned k = 0; k < NTT_LB; ++k) {
            const unsigned idx = i0 + k * 256u + threadIdx.x;
            if (idx < nelem) {
                Fr29 x = f29_from_fp(raw[k]);
                if (PRE) x = mulc(x, praw[k]);
                lds29_put(sm, (idx >> p.logR) * R + bitrev32(idx & (R - 1), p.logR), x);
            }
        }
    }
    u32* qtab = sm + ntt29_lds_words(nelem);   // behind the tile: q * p rows of the quotient-estimate canonicalisation
    if (!has_post) f29_qtab_fill<FrTag>(qtab);   // before the barrier that publishes the tile: lds_dit29 has none when logR == 0
    __syncthreads();
    lds_dit29(sm, p.logR, T, 1, R, false, tw, p.n);
    for (unsigned idx = threadIdx.x; idx < R * T; idx += blockDim.x) {
        unsigned k = idx >> logT, r = idx & (T - 1);
        Fr29 x = lds29_get(sm, r * R + k);
        Fr* o = dst + (k1_0 + r) + p.n1 * k2 + p.hi * (size_t)k;
        if (has_post) f29_store<1>(o, mulc(x, post));   // a product: strict limbs, below 3p
        else ntt_store_q(o, x, qtab);
    }
}

// final pass of the coset-EXTENDED transform (coeff_to_extended): the 2^e * n point NTT of a zero-extended
// n-coefficient polynomial is 2^e independent n-point NTTs of a[i] * (g * w_ext^r)^i, r < 2^e, interleaved
// as out[2^e * q + r].  The strided pass has already run per r (inputs at in + r * in_r_stride); this
// block finishes T rows for ALL r at once so every store is a full T * 2^e * 32-byte run.
template <bool PRE>
__global__ NTT_BOUNDS void k_ntt_final_ext29(const Fr* in, Fr* out, size_t in_stride, size_t in_r_stride,
                                                         size_t out_stride, NttPass p, unsigned log_e,
                                                         const u32* __restrict__ tw, const u32* __restrict__ pre,
                                                         size_t pre_r_stride) {
    u32* sm = reinterpret_cast<u32*>(pz_smem);
    const unsigned R = 1u << p.logR, T = p.T, E = 1u << log_e;
    const unsigned logT = 31u - (unsigned)__builtin_clz(T);
    const size_t tiles = p.n1 >> logT;
    const size_t bx = p.swap ? blockIdx.y : blockIdx.x, by = p.swap ? blockIdx.x : blockIdx.y;
    const size_t k2 = bx / tiles, k1_0 = (bx % tiles) * T;
    const Fr* src = in + by * in_stride;
    Fr* dst = out + by * out_stride;
    const unsigned nelem = R * T * E;
    for (unsigned i0 = 0; i0 < nelem; i0 += NTT_LB * 256u) {
        Fr raw[NTT_LB];
        Raw9 praw[NTT_LB];
#pragma unroll
        for (unsigned k = 0; k < NTT_LB; ++k) {
            const unsigned idx = (i0 + k * 256u + threadIdx.x) & (nelem - 1);
            const unsigned j = idx & (R - 1), rr = (idx >> p.logR) & (T - 1), r = idx >> (p.logR + logT);
            const size_t g = ((k1_0 + rr) * p.n2 + k2) * R + j;
            raw[k] = fp_load<FrTag>(src + (size_t)r * in_r_stride + g);
            if (PRE) praw[k] = raw9_load(pre, (size_t)r * pre_r_stride + g);
        }
#pragma unroll
        for (unsigned k = 0; k < NTT_LB; ++k) {
            const unsigned idx = i0 + k * 256u + threadIdx.x;
            if (idx < nelem) {
                const unsigned j = idx & (R - 1), rr = (idx >> p.logR) & (T - 1), r = idx >> (p.logR + logT);
                Fr29 x = f29_from_fp(raw[k]);
                if (PRE) x = mulc(x, praw[k]);
                lds29_put(sm, (r * T + rr) * R + bitrev32(j, p.logR), x);
            }
        }
    }
    u32* qtab = sm + ntt29_lds_words(nelem);
    f29_qtab_fill<FrTag>(qtab);   // before the barrier that publishes the tile (see k_ntt_final29)
    __syncthreads();
    lds_dit29(sm, p.logR, T * E, 1, R, false, tw, p.n);
    for (unsigned idx = threadIdx.x; idx < R * T * E; idx += blockDim.x) {
        const unsigned r = idx & (E - 1), rr = (idx >> log_e) & (T - 1), k = idx >> (log_e + logT);
        const Fr29 x = lds29_get(sm, (r * T + rr) * R + k);
        ntt_store_q(dst + (((k1_0 + rr) + p.n1 * k2 + p.hi * (size_t)k) << log_e) + r, x, qtab);
    }
}

// ------------------------------------------------------------------------------------------------
#ifndef PZ_NTT_LDS
#define PZ_NTT_LDS 32768
#endif
#define NTT29_ELEM 36u   // bytes of one tile element of the 29-bit kernels (9 words)
static size_t ntt29_lds_bytes(size_t elems) { return (elems + (elems >> 5) + 1) * NTT29_ELEM; }
static size_t ntt_tile_budget() {   // tile size in 32-byte units x 32 (PZ_NTT_TILE_KIB overrides for tuning: 16, 32, 64)
    static size_t v = 0;
    if (!v) {
        const char* e = getenv("PZ_NTT_TILE_KIB");
        v = e ? (size_t)atol(e) * 1024 : PZ_NTT_LDS;
        if (v < 4096) v = PZ_NTT_LDS;
    }
    return v;
}
static unsigned pick_tile(size_t extent, unsigned logR) {
    // 1024 elements per block (36 KiB of the 9-word elements: 4 blocks per CU); prefer 128-byte runs (T = 4) or more
    unsigned T = 16;
    while (T > 1 && (((size_t)32 << logR) * T > ntt_tile_budget() || T > extent)) T >>= 1;
    return T;
}

// value * 32 mod r on the host (canonical 4 x u64 in and out): a 256-domain Montgomery constant c * 2^256 becomes the
// 261-domain constant c * 2^261 the 29-bit kernels multiply by
static void fr_times32(const uint64_t in[4], uint64_t out[4]) {
    static const uint64_t R_[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    uint64_t a[4] = {in[0], in[1], in[2], in[3]};
    for (int k = 0; k < 5; ++k) {
        uint64_t d[4], c = 0;
        for (int i = 0; i < 4; ++i) {   // d = 2a (a < r < 2^254: no carry out)
            d[i] = (a[i] << 1) | c;
            c = a[i] >> 63;
        }
        uint64_t t[4], br = 0;
        for (int i = 0; i < 4; ++i) {   // t = d - r
            const uint64_t x = d[i] - R_[i], b1 = d[i] < R_[i], y = x - br;
            br = b1 | (x < br);
            t[i] = y;
        }
        for (int i = 0; i < 4; ++i) a[i] = br ? d[i] : t[i];
    }
    for (int i = 0; i < 4; ++i) out[i] = a[i];
}
// Montgomery one (R mod r) times 32: the first entry of a 261-domain power table
static const uint64_t* one261() {
    static uint64_t v[4];
    static bool init = false;
    if (!init) {
        static const uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
        fr_times32(ONE, v);
        init = true;
    }
    return v;
}

const uint64_t* pz_fr_one261() { return one261(); }   // for the other translation units' 261-domain power tables

static int launch_strided(pz_ctx* ctx, const Fr* in, Fr* out, size_t is, size_t os, size_t ncols, NttPass p,
                          const u32* tw, const u32* pre, unsigned n_r = 1, size_t pre_r_stride = 0, size_t out_r_stride = 0,
                          const u32* stw = nullptr, const u32* tw_ep = nullptr) {
    if (!tw_ep) tw_ep = tw;
    size_t blocks = p.hi * (p.lo / p.T);
    size_t lds = ntt29_lds_bytes(((size_t)1 << p.logR) * p.T);
    p.swap = (ncols > 1 && blocks <= 65535) ? 1u : 0u;
    const dim3 grid = p.swap ? dim3((unsigned)ncols, (unsigned)blocks) : dim3((unsigned)blocks, (unsigned)ncols);
    if (stw) hipLaunchKernelGGL(k_ntt_strided29<2>, grid, dim3(256), lds, ctx->stream, in, out, is, os, p, tw, pre, n_r, pre_r_stride, out_r_stride, stw, tw_ep);
    else if (pre) hipLaunchKernelGGL(k_ntt_strided29<1>, grid, dim3(256), lds, ctx->stream, in, out, is, os, p, tw, pre, n_r, pre_r_stride, out_r_stride, stw, tw_ep);
    else hipLaunchKernelGGL(k_ntt_strided29<0>, grid, dim3(256), lds, ctx->stream, in, out, is, os, p, tw, pre, n_r, pre_r_stride, out_r_stride, stw, tw_ep);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}
// host: a Montgomery constant (c * 2^256 mod r, canonical) -> the pair (c, floor(c 2^261 / r)) f29_mulc takes (the device tables get
// theirs from f29_cpair_from_mont; this is the same arithmetic for the one scalar a launch passes by value)
static void host_cpair(const uint64_t mont[4], C18& out) {
    static const uint64_t R_[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    auto ge = [&](const uint64_t a[4]) {
        for (int i = 3; i >= 0; --i)
            if (a[i] != R_[i]) return a[i] > R_[i];
        return true;
    };
    auto sub = [&](uint64_t a[4]) {
        unsigned __int128 br = 0;
        for (int i = 0; i < 4; ++i) {
            const unsigned __int128 d = (unsigned __int128)a[i] - R_[i] - br;
            a[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    };
    // c = mont * 2^-256 mod r: 256 halvings (x odd: (x + r) / 2)
    uint64_t c[4] = {mont[0], mont[1], mont[2], mont[3]};
    for (int b = 0; b < 256; ++b) {
        uint64_t top = 0;
        if (c[0] & 1) {
            unsigned __int128 cy = 0;
            for (int i = 0; i < 4; ++i) {
                const unsigned __int128 t = (unsigned __int128)c[i] + R_[i] + cy;
                c[i] = (uint64_t)t;
                cy = t >> 64;
            }
            top = (uint64_t)cy;
        }
        for (int i = 0; i < 3; ++i) c[i] = (c[i] >> 1) | (c[i + 1] << 63);
        c[3] = (c[3] >> 1) | (top << 63);
    }
    uint64_t rem[4] = {c[0], c[1], c[2], c[3]}, Q[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < 261; ++b) {
        for (int i = 3; i > 0; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
        rem[0] <<= 1;
        for (int i = 4; i > 0; --i) Q[i] = (Q[i] << 1) | (Q[i - 1] >> 63);
        Q[0] <<= 1;
        if (ge(rem)) {
            sub(rem);
            Q[0] |= 1;
        }
    }
    auto limbs = [](const uint64_t* x, int words, u32* o) {
        for (int i = 0; i < 9; ++i) {
            const int bit = 29 * i, j = bit >> 6, sh = bit & 63;
            uint64_t v = j < words ? x[j] >> sh : 0;
            if (sh > 35 && j + 1 < words) v |= x[j + 1] << (64 - sh);
            o[i] = (u32)v & F29_MASK;
        }
    };
    limbs(c, 4, out.w);
    limbs(Q, 5, out.q);
}

static int launch_final(pz_ctx* ctx, const Fr* in, Fr* out, size_t is, size_t os, size_t ncols, NttPass p,
                        const u32* tw, const u32* pre, const uint64_t* post_scale) {
    size_t blocks = p.n2 * (p.n1 / p.T);
    size_t lds = ntt29_lds_bytes(((size_t)1 << p.logR) * p.T) + F29_QTAB_WORDS * 4;
    C18 post;
    memset(&post, 0, sizeof post);
    if (post_scale) host_cpair(post_scale, post);
    p.swap = (ncols > 1 && blocks <= 65535) ? 1u : 0u;
    const dim3 grid = p.swap ? dim3((unsigned)ncols, (unsigned)blocks) : dim3((unsigned)blocks, (unsigned)ncols);
    if (pre) hipLaunchKernelGGL(k_ntt_final29<true>, grid, dim3(256), lds, ctx->stream, in, out, is, os, p, tw, pre, post, post_scale ? 1 : 0);
    else hipLaunchKernelGGL(k_ntt_final29<false>, grid, dim3(256), lds, ctx->stream, in, out, is, os, p, tw, pre, post, post_scale ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

extern "C" int pz_ntt_fr_dev(pz_ctx* ctx, uint64_t* d_a, size_t n_cols, size_t col_stride, const uint64_t omega[4],
                             uint32_t log_n, const uint64_t* pre_coset_g, const uint64_t* post_scale) {
    return pz_ntt_fr_to_dev(ctx, d_a, col_stride, d_a, col_stride, n_cols, omega, log_n, pre_coset_g, post_scale);
}

// out of place: the first pass reads d_in, the last one writes d_out (the same passes; in place when the two are equal)
extern "C" int pz_ntt_fr_to_dev(pz_ctx* ctx, const uint64_t* d_in, size_t in_stride, uint64_t* d_out, size_t out_stride, size_t n_cols,
                                const uint64_t omega[4], uint32_t log_n, const uint64_t* pre_coset_g, const uint64_t* post_scale) {
    if (!ctx || !omega || (n_cols && (!d_in || !d_out)) || log_n > 27) return PZ_ERR_INVALID;
    if (n_cols == 0) return PZ_OK;
    const size_t n = (size_t)1 << log_n;
    if (in_stride % 4 || in_stride < 4 * n || out_stride % 4 || out_stride < 4 * n) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    const size_t cs = out_stride / 4, is_ = in_stride / 4;  // column strides in elements
    Fr* a = reinterpret_cast<Fr*>(d_out);
    const Fr* ain = reinterpret_cast<const Fr*>(d_in);
    void* twv = nullptr;
    PZCHK(pz_get_pow_table_raw(ctx, omega, n, &twv));   // omega^i as constant pairs
    const u32* tw = (const u32*)twv;
    void* prev = nullptr;
    if (pre_coset_g) PZCHK(pz_get_pow_table_raw(ctx, pre_coset_g, n, &prev));
    const u32* pre = (const u32*)prev;

    const unsigned npass = log_n <= 9 ? 1 : (log_n <= 18 ? 2 : 3);
    // a post-scale (the 1/n of an inverse transform) rides on the first pass's inter-pass product -- its omega table taken with
    // the scale as first entry -- instead of costing the last pass a product per element
    const u32* tw_ep = nullptr;
    if (post_scale && npass > 1) {
        void* t;
        PZCHK(pz_get_pow_table_raw(ctx, omega, n, &t, post_scale));
        tw_ep = (const u32*)t;
        post_scale = nullptr;
    }
    unsigned lg[3] = {0, 0, 0};
    {
        unsigned rem = log_n;
        for (unsigned i = 0; i < npass; ++i) {
            lg[i] = (rem + (npass - i) - 1) / (npass - i);
            rem -= lg[i];
        }
    }
    // column groups: bound the ping-pong workspace to ~1 GiB and grid.y to 65535
    size_t group = n_cols;
    const size_t max_ws = (size_t)1 << 30;
    if (npass > 1 && group * n * 32 > max_ws) group = max_ws / (n * 32) ? max_ws / (n * 32) : 1;
    if (group > 32768) group = 32768;
    Fr* tmp = nullptr;
    if (npass > 1) {
        void* t;
        PZCHK(pz_ws_get(ctx, WS_NTT_TMP, group * n * 32, &t));
        tmp = (Fr*)t;
    }
    pz_timer tm(ctx, PZ_T_NTT);
    for (size_t c0 = 0; c0 < n_cols; c0 += group) {
        size_t nc = n_cols - c0 < group ? n_cols - c0 : group;
        Fr* ac = a + c0 * cs;
        const Fr* ic = ain + c0 * is_;
        if (npass == 1) {
            NttPass p{};
            p.logR = log_n; p.lo = 1; p.hi = 1; p.tw_mul = 0; p.n = n; p.T = 1; p.n1 = 1; p.n2 = 1;
            PZCHK(launch_final(ctx, ic, ac, is_, cs, nc, p, tw, pre, post_scale));
        } else if (npass == 2) {
            size_t n1 = (size_t)1 << lg[0], n2 = (size_t)1 << lg[1];
            NttPass pa{};
            pa.logR = lg[0]; pa.lo = n2; pa.hi = 1; pa.tw_mul = 1; pa.n = n; pa.T = pick_tile(n2, lg[0]);
            PZCHK(launch_strided(ctx, ic, tmp, is_, n, nc, pa, tw, pre, 1, 0, 0, nullptr, tw_ep));
            // tmp columns are packed with stride n
            NttPass pc{};
            pc.logR = lg[1]; pc.lo = 1; pc.hi = n1; pc.n = n; pc.n1 = n1; pc.n2 = 1; pc.T = pick_tile(n1, lg[1]);
            PZCHK(launch_final(ctx, tmp, ac, n, cs, nc, pc, tw, nullptr, post_scale));
        } else {
            size_t n1 = (size_t)1 << lg[0], n2 = (size_t)1 << lg[1], n3 = (size_t)1 << lg[2];
            NttPass pa{};
            pa.logR = lg[0]; pa.lo = n2 * n3; pa.hi = 1; pa.tw_mul = 1; pa.n = n; pa.T = pick_tile(n2 * n3, lg[0]);
            PZCHK(launch_strided(ctx, ic, tmp, is_, n, nc, pa, tw, pre, 1, 0, 0, nullptr, tw_ep));
            NttPass pb{};
            pb.logR = lg[1]; pb.lo = n3; pb.hi = n1; pb.tw_mul = n1; pb.n = n; pb.T = pick_tile(n3, lg[1]);
            PZCHK(launch_strided(ctx, tmp, tmp, n, n, nc, pb, tw, nullptr));
            NttPass pc{};
            pc.logR = lg[2]; pc.lo = 1; pc.hi = n1 * n2; pc.n = n; pc.n1 = n1; pc.n2 = n2; pc.T = pick_tile(n1, lg[2]);
            PZCHK(launch_final(ctx, tmp, ac, n, cs, nc, pc, tw, nullptr, post_scale));
        }
    }
    return PZ_OK;
}

// E pre-scale tables scale * gens[r]^i * 2^261 (the 261-domain constants of the 29-bit kernels), packed [E][n], cached
static int get_ext_pre_tables(pz_ctx* ctx, const uint64_t* coset_gens, size_t E, uint32_t log_n, const uint64_t* scale,
                              void** out) {
    const size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> key(coset_gens, coset_gens + 4 * E);
    key.push_back(log_n);
    if (scale) key.insert(key.end(), scale, scale + 4);
    for (auto& c : ctx->ext_tables)
        if (c.key == key) {
            *out = c.d;
            return PZ_OK;
        }
    void* prev = nullptr;
    HIPCHK(ctx, pz_hip_malloc(ctx, &prev, E * n * 72));   // constant pairs
    for (size_t r = 0; r < E; ++r) {
        void* t;
        PZCHK(pz_get_pow_table(ctx, coset_gens + 4 * r, n, &t, scale));
        PZCHK(pz_raw29_convert(ctx, t, (char*)prev + r * n * 72, n));
    }
    ctx->ext_tables.push_back(pz_ext_table{key, prev});
    *out = prev;
    return PZ_OK;
}

// coeff_to_extended in one call: d_ext[col][2^e * q + r] = sum_i d_coeff[col][i] * scale * (gens[r])^i * omega_n^(i q)
// with gens[r] = g * omega_ext^r supplied by the caller (2^log_e x 4 limbs, host), omega_n = omega_ext^(2^log_e).
// Saves the zero-fill, the copy and two butterfly layers of the generic zero-extended transform.
// ---- the coset shift of the extended transform ABSORBED into its first pass (no pre-scale product).
// The 4-step first pass is, for every column col < lo, the R-point DFT over j1 of a[j1 * lo + col] * c^(j1 * lo + col).  The factor
// c^col is constant inside the DFT: it moves into the inter-pass twiddle, E[k * lo + col] = c^col * omega^(col * k) (* scale).
// The factor (c^lo)^j1 turns the DFT into an evaluation on the coset c^lo * <omega_R>, and a radix-2 DIT does that with every
// stage's twiddles times a constant: stage s (blocks of 2^s joined into 2^(s+1)) uses K_s * omega^(pos * n / 2^(s+1)),
// K_s = c^(n / 2^(s+1)).  Stages 0 + 1 lose their trivial twiddles (0.75 products per element more), the pre-scale goes
// (1 product per element less).  Both tables in the 261-domain as raw limbs; S[(2^s - 1) + pos], R entries apart per coset.
__global__ void k_ext_epilogue_table(const Fr* __restrict__ cpow, const Fr* __restrict__ twm, size_t lo, size_t n, u32* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t k = i / lo, col = i % lo;
    u32 o[18];
    f29_cpair_from_mont(fp_mul(fp_load<FrTag>(cpow + col), fp_load<FrTag>(twm + col * k)), o);   // col * k < lo * R = n
#pragma unroll
    for (int j = 0; j < 18; ++j) out[i * 18 + j] = o[j];
}
__global__ void k_ext_stage_table(Fr c, const Fr* __restrict__ twm, size_t n, unsigned logR, u32* __restrict__ out) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, R = 1u << logR;
    if (t + 1 >= R) return;
    const unsigned s = 31u - (unsigned)__builtin_clz(t + 1), pos = t + 1 - (1u << s);
    // K_s = c^(n >> (s + 1)): a power of two, by squarings (Montgomery form)
    Fr K = c;
    for (size_t e = n >> (s + 1); e > 1; e >>= 1) K = fp_sqr(K);
    u32 o[18];
    f29_cpair_from_mont(fp_mul(K, fp_load<FrTag>(twm + (size_t)pos * (n >> (s + 1)))), o);
#pragma unroll
    for (int j = 0; j < 18; ++j) out[(size_t)t * 18 + j] = o[j];
}
// out_ep: [E][n][9], out_stw: [E][R][9]
static int get_ext_abs_tables(pz_ctx* ctx, const uint64_t* coset_gens, size_t E, uint32_t log_n, unsigned logR, const uint64_t* scale,
                              const uint64_t omega_n[4], const u32** out_ep, const u32** out_stw) {
    const size_t n = (size_t)1 << log_n, R = (size_t)1 << logR, lo = n >> logR;
    std::vector<uint64_t> key(coset_gens, coset_gens + 4 * E);
    key.push_back(log_n);
    key.push_back(0xab50000ull + logR);   // distinguishes these tables from the pre-scale ones of the same cosets
    key.insert(key.end(), omega_n, omega_n + 4);   // both tables are built from omega_n's powers (`tw`): part of the key
    if (scale) key.insert(key.end(), scale, scale + 4);
    for (auto& c : ctx->ext_tables)
        if (c.key == key) {
            *out_ep = (const u32*)c.d;
            *out_stw = (const u32*)c.d + E * n * 18;
            return PZ_OK;
        }
    void* d = nullptr;
    HIPCHK(ctx, pz_hip_malloc(ctx, &d, (E * n + E * R) * 72));
    u32* ep = (u32*)d;
    u32* stw = ep + E * n * 18;
    void* twm;
    PZCHK(pz_get_pow_table(ctx, omega_n, n, &twm));   // omega^i, Montgomery form: the builders multiply in that form and convert once
    for (size_t r = 0; r < E; ++r) {
        void* cp;
        PZCHK(pz_get_pow_table(ctx, coset_gens + 4 * r, lo, &cp, scale));   // c^col * scale
        hipLaunchKernelGGL(k_ext_epilogue_table, dim3(pz_div_up(n, 256)), dim3(256), 0, ctx->stream, (const Fr*)cp, (const Fr*)twm, lo, n, ep + r * n * 18);
        Fr c;
        memcpy(c.v, coset_gens + 4 * r, 32);
        hipLaunchKernelGGL(k_ext_stage_table, dim3(pz_div_up(R, 256)), dim3(256), 0, ctx->stream, c, (const Fr*)twm, n, logR, stw + r * R * 18);
        HIPCHK(ctx, hipGetLastError());
    }
    ctx->ext_tables.push_back(pz_ext_table{key, d});
    *out_ep = ep;
    *out_stw = stw;
    return PZ_OK;
}

extern "C" int pz_ntt_fr_extend_dev(pz_ctx* ctx, const uint64_t* d_coeff, size_t n_cols, size_t in_stride, uint64_t* d_ext,
                                    size_t out_stride, uint32_t log_n, uint32_t log_e, const uint64_t omega_n[4],
                                    const uint64_t* coset_gens, const uint64_t* scale) {
    if (!ctx || !omega_n || !coset_gens || (n_cols && (!d_coeff || !d_ext)) || log_e > 3) return PZ_ERR_INVALID;
    if (log_n > 27) return PZ_ERR_INVALID;
    if (n_cols == 0) return PZ_OK;
    const size_t n = (size_t)1 << log_n, E = (size_t)1 << log_e;
    if (in_stride % 4 || in_stride < 4 * n || out_stride % 4 || out_stride < 4 * n * E) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    const size_t is = in_stride / 4, os = out_stride / 4;
    void* twv = nullptr;
    PZCHK(pz_get_pow_table_raw(ctx, omega_n, n, &twv));
    const u32* tw = (const u32*)twv;
    const unsigned npass_ = log_n <= 9 ? 1 : (log_n <= 18 ? 2 : 3);
    // several passes: the coset shift is absorbed into the first pass's twiddles (get_ext_abs_tables); a single-pass transform
    // (log_n <= 9) keeps the pre-scale product (round 3's A/B switch PZ_NTT_COSET=pre is gone: -2.1 us per polynomial)
    const bool absorbed = npass_ > 1;
    const unsigned logR_a = npass_ == 2 ? (log_n + 1) / 2 : (log_n + 2) / 3;   // size of the first pass (lg0 / lg[0] below)
    const u32 *pre = nullptr, *stw = nullptr;
    if (absorbed) {
        PZCHK(get_ext_abs_tables(ctx, coset_gens, E, log_n, logR_a, scale, omega_n, &pre, &stw));
    } else {
        void* prev = nullptr;
        PZCHK(get_ext_pre_tables(ctx, coset_gens, E, log_n, scale, &prev));
        pre = (const u32*)prev;
    }
    const Fr* cin = (const Fr*)d_coeff;
    Fr* eout = (Fr*)d_ext;
    const unsigned npass = log_n <= 9 ? 1 : (log_n <= 18 ? 2 : 3);
    size_t group = n_cols;
    const size_t max_ws = (size_t)2 << 30;
    if (npass > 1 && group * n * E * 32 > max_ws) group = max_ws / (n * E * 32) ? max_ws / (n * E * 32) : 1;
    if (group > 32768) group = 32768;
    Fr* tmp = nullptr;
    if (npass > 1) {
        void* t;
        PZCHK(pz_ws_get(ctx, WS_NTT_TMP, group * n * E * 32, &t));
        tmp = (Fr*)t;
    }
    pz_timer tm(ctx, PZ_T_NTT);
    for (size_t c0 = 0; c0 < n_cols; c0 += group) {
        const size_t nc = n_cols - c0 < group ? n_cols - c0 : group;
        if (npass == 1) {
            NttPass p{};
            p.logR = log_n; p.lo = 1; p.hi = 1; p.n = n; p.T = 1; p.n1 = 1; p.n2 = 1;
            const size_t lds = ntt29_lds_bytes(((size_t)1 << p.logR) * p.T * E) + F29_QTAB_WORDS * 4;
            p.swap = 0;
            hipLaunchKernelGGL(k_ntt_final_ext29<true>, dim3(1, (unsigned)nc), dim3(256), lds, ctx->stream, cin + c0 * is,
                               eout + c0 * os, is, (size_t)0, os, p, log_e, tw, pre, n);
        } else if (npass == 3) {
            // 2^19 .. 2^27 (config c5 runs k = 19): two strided passes per coset, then the interleaving final pass
            unsigned lg[3], rem = log_n;
            for (unsigned i = 0; i < 3; ++i) {
                lg[i] = (rem + (3 - i) - 1) / (3 - i);
                rem -= lg[i];
            }
            const size_t n1 = (size_t)1 << lg[0], n2 = (size_t)1 << lg[1], n3 = (size_t)1 << lg[2];
            NttPass pa{};
            pa.logR = lg[0]; pa.lo = n2 * n3; pa.hi = 1; pa.tw_mul = 1; pa.n = n; pa.T = pick_tile(n2 * n3, lg[0]);
            NttPass pb{};
            pb.logR = lg[1]; pb.lo = n3; pb.hi = n1; pb.tw_mul = n1; pb.n = n; pb.T = pick_tile(n3, lg[1]);
            // tmp layout [r][col][n]: all cosets of the first pass in one launch, then the E * nc intermediate columns together
            PZCHK(launch_strided(ctx, cin + c0 * is, tmp, is, n, nc, pa, tw, pre, (unsigned)E, n, nc * n, stw));
            PZCHK(launch_strided(ctx, tmp, tmp, n, n, E * nc, pb, tw, nullptr));
            NttPass pc{};
            pc.logR = lg[2]; pc.lo = 1; pc.hi = n1 * n2; pc.n = n; pc.n1 = n1; pc.n2 = n2;
            unsigned T = 8;
            while (T > 1 && (((size_t)32 << lg[2]) * T * E > ntt_tile_budget() || T > n1)) T >>= 1;
            pc.T = T;
            const size_t lds = ntt29_lds_bytes(((size_t)1 << lg[2]) * T * E) + F29_QTAB_WORDS * 4;
            pc.swap = (nc > 1 && n2 * (n1 / T) <= 65535) ? 1u : 0u;
            hipLaunchKernelGGL(k_ntt_final_ext29<false>, pc.swap ? dim3((unsigned)nc, (unsigned)(n2 * (n1 / T))) : dim3((unsigned)(n2 * (n1 / T)), (unsigned)nc), dim3(256), lds, ctx->stream, tmp,
                               eout + c0 * os, n, nc * n, os, pc, log_e, tw, (const u32*)nullptr, (size_t)0);
        } else {
            unsigned lg0 = (log_n + 1) / 2, lg1 = log_n - lg0;
            const size_t n1 = (size_t)1 << lg0, n2 = (size_t)1 << lg1;
            NttPass pa{};
            pa.logR = lg0; pa.lo = n2; pa.hi = 1; pa.tw_mul = 1; pa.n = n; pa.T = pick_tile(n2, lg0);
            // all cosets in one launch: tmp layout [r][col][n]
            PZCHK(launch_strided(ctx, cin + c0 * is, tmp, is, n, nc, pa, tw, pre, (unsigned)E, n, nc * n, stw));
            NttPass pc{};
            pc.logR = lg1; pc.lo = 1; pc.hi = n1; pc.n = n; pc.n1 = n1; pc.n2 = 1;
            // 32 KiB of LDS per block (4 blocks per CU): with 2^e = 4 interleaved outputs a single row already
            // stores full 128-byte lines
            unsigned T = 8;
            while (T > 1 && (((size_t)32 << lg1) * T * E > ntt_tile_budget() || T > n1)) T >>= 1;
            pc.T = T;
            const size_t lds = ntt29_lds_bytes(((size_t)1 << lg1) * T * E) + F29_QTAB_WORDS * 4;
            pc.swap = nc > 1 ? 1u : 0u;
            hipLaunchKernelGGL(k_ntt_final_ext29<false>, pc.swap ? dim3((unsigned)nc, (unsigned)(n1 / T)) : dim3((unsigned)(n1 / T), (unsigned)nc), dim3(256), lds, ctx->stream, tmp,
                               eout + c0 * os, n, nc * n, os, pc, log_e, tw, (const u32*)nullptr, (size_t)0);
        }
        HIPCHK(ctx, hipGetLastError());
    }
    return PZ_OK;
}

// Lagrange values -> coefficients (in place, == EvaluationDomain::lagrange_to_coeff: best_fft(omega^-1) then * 1/n)
// AND -> extended coset values (== coeff_to_extended) in one call, sharing the fused kernel above.  omega_n_inv and
// n_inv are supplied by the caller (Montgomery, host) like every other constant of the ABI.
extern "C" int pz_ntt_fr_coeff_extend_dev(pz_ctx* ctx, uint64_t* d_values, size_t n_cols, size_t col_stride, uint64_t* d_ext,
                                          size_t out_stride, uint32_t log_n, uint32_t log_e, const uint64_t omega_n[4],
                                          const uint64_t omega_n_inv[4], const uint64_t n_inv[4], const uint64_t* coset_gens) {
    if (!ctx || !omega_n || !omega_n_inv || !n_inv || !coset_gens || (n_cols && (!d_values || !d_ext)) || log_e > 3 || log_n > 27)
        return PZ_ERR_INVALID;
    if (n_cols == 0) return PZ_OK;
    const size_t n = (size_t)1 << log_n, E = (size_t)1 << log_e;
    if (col_stride % 4 || col_stride < 4 * n || out_stride % 4 || out_stride < 4 * n * E) return PZ_ERR_INVALID;
    // Round 1 had a kernel fusing the inverse transform's final pass with the first pass of the coset transforms
    // (one third less traffic, bit-identical results); it measured no faster than the two calls -- the transforms are
    // multiplier-bound, DESIGN.md section 6.1 -- and was dropped when the kernels moved to the 29-bit field.
    PZCHK(pz_ntt_fr_dev(ctx, d_values, n_cols, col_stride, omega_n_inv, log_n, nullptr, n_inv));
    return pz_ntt_fr_extend_dev(ctx, d_values, n_cols, col_stride, d_ext, out_stride, log_n, log_e, omega_n, coset_gens, nullptr);
}

// host-pointer forms: == best_fft(a, omega, log_n) on each column
extern "C" int pz_ntt_fr_batch(pz_ctx* ctx, uint64_t* const* cols, size_t n_cols, const uint64_t omega[4],
                               uint32_t log_n) {
    if (!ctx || !omega || (n_cols && !cols) || log_n > 27) return PZ_ERR_INVALID;
    if (n_cols == 0) return PZ_OK;
    PZ_ENTER(ctx);
    const size_t n = (size_t)1 << log_n, bytes = n * 32;
    for (size_t j = 0; j < n_cols; ++j)
        if (!cols[j]) return PZ_ERR_INVALID;
    // column groups of ~128 MiB in three staging buffers: upload of group g+1 (io_h2d), kernels of group g (caller's stream)
    // and download of group g-1 (io_d2h) run at the same time -- PCIe is full duplex
    size_t group = ((size_t)1 << 27) / bytes;
    if (group == 0) group = 1;
    if (group > n_cols) group = n_cols;
    void* d;
    PZCHK(pz_ws_get(ctx, WS_IO_A, 3 * group * bytes, &d));
    PZCHK(pz_io_init(ctx));
    hipEvent_t* ev_in = ctx->io_ev;         // [3] group uploaded
    hipEvent_t* ev_done = ctx->io_ev + 3;   // [3] group transformed
    hipEvent_t* ev_out = ctx->io_ev + 6;    // [3] group downloaded (its staging buffer is free again)
    int rc = PZ_OK;
    size_t g = 0;
    for (size_t c0 = 0; c0 < n_cols && rc == PZ_OK; c0 += group, ++g) {
        const size_t nc = n_cols - c0 < group ? n_cols - c0 : group;
        const unsigned b = (unsigned)(g % 3);
        char* buf = (char*)d + b * group * bytes;
        hipError_t e = hipSuccess;
        if (g >= 3) e = hipStreamWaitEvent(ctx->io_h2d, ev_out[b], 0);
        for (size_t j = 0; j < nc && e == hipSuccess; ++j)
            e = hipMemcpyAsync(buf + j * bytes, cols[c0 + j], bytes, hipMemcpyHostToDevice, ctx->io_h2d);
        if (e == hipSuccess) e = hipEventRecord(ev_in[b], ctx->io_h2d);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ev_in[b], 0);
        if (e != hipSuccess) { rc = pz_hip_fail(ctx, e, "pz_ntt_fr_batch: upload"); break; }
        rc = pz_ntt_fr_dev(ctx, (uint64_t*)buf, nc, 4 * n, omega, log_n, nullptr, nullptr);
        if (rc != PZ_OK) break;
        e = hipEventRecord(ev_done[b], ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->io_d2h, ev_done[b], 0);
        for (size_t j = 0; j < nc && e == hipSuccess; ++j)
            e = hipMemcpyAsync(cols[c0 + j], buf + j * bytes, bytes, hipMemcpyDeviceToHost, ctx->io_d2h);
        if (e == hipSuccess) e = hipEventRecord(ev_out[b], ctx->io_d2h);
        if (e != hipSuccess) rc = pz_hip_fail(ctx, e, "pz_ntt_fr_batch: download");
    }
    // the host buffers belong to the caller again when this returns, whatever happened
    hipError_t e1 = hipStreamSynchronize(ctx->io_h2d), e2 = hipStreamSynchronize(ctx->stream), e3 = hipStreamSynchronize(ctx->io_d2h);
    if (rc == PZ_OK && (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess))
        rc = pz_hip_fail(ctx, e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3), "pz_ntt_fr_batch: synchronize");
    return rc;
}

extern "C" int pz_ntt_fr(pz_ctx* ctx, uint64_t* a, const uint64_t omega[4], uint32_t log_n) {
    if (!a) return PZ_ERR_INVALID;
    uint64_t* cols[1] = {a};
    return pz_ntt_fr_batch(ctx, cols, 1, omega, log_n);
}

__global__ void k_fr_convert(Fr* a, size_t n, int to_mont) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        Fr x = fp_load<FrTag>(a + i);
        fp_store(a + i, to_mont ? fp_to_mont(x) : fp_from_mont(x));
    }
}
extern "C" int pz_fr_convert_dev(pz_ctx* ctx, uint64_t* d_a, size_t n, int to_mont) {
    if (!ctx || (n && !d_a)) return PZ_ERR_INVALID;
    if (!n) return PZ_OK;
    PZ_ENTER(ctx);
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_fr_convert, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (Fr*)d_a, n, to_mont);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}
