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
#include <utility>
#include <vector>

#include "fp.cuh"
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

// logR radix-2 DIT stages over an LDS tile holding T independent transforms, taken TWO STAGES AT A TIME: a thread
// owns the four elements {base, base+h, base+2h, base+3h} of a radix-4 group, does both layers in registers (same
// four products as two radix-2 layers) and the tile crosses LDS and a barrier once per pair of stages instead of
// once per stage.  Stages 0+1 cost one product per group (their other twiddles are 1).  An odd logR ends with one
// plain radix-2 stage.  element (j, t) lives at sm[j*sr + t*st].  Input must be stored bit-reversed in j.
__device__ __forceinline__ void lds_dit(Fr* sm, unsigned logR, unsigned T, unsigned sr, unsigned st, bool t_fastest,
                                        const Fr* __restrict__ tw, size_t n) {
    const unsigned R = 1u << logR;
    const unsigned logT = 31u - (unsigned)__builtin_clz(T);  // T is a power of two: every index split below is a shift / mask
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
            Fr* p0 = sm + (size_t)base * sr + (size_t)t * st;
            Fr* p1 = p0 + (size_t)h * sr;
            Fr* p2 = p1 + (size_t)h * sr;
            Fr* p3 = p2 + (size_t)h * sr;
            Fr a0 = *p0, a1 = *p1, a2 = *p2, a3 = *p3;
            if (s) {  // kernel-uniform: the first layer of stages 0+1 has twiddle 1 everywhere
                const Fr w = fp_load<FrTag>(tw + (size_t)pos * (n >> (s + 1)));
                a1 = fp_mul(a1, w);
                a3 = fp_mul(a3, w);
            }
            const Fr b0 = fp_add(a0, a1), b1 = fp_sub(a0, a1);
            Fr b2 = fp_add(a2, a3), b3 = fp_sub(a2, a3);
            if (s) b2 = fp_mul(b2, fp_load<FrTag>(tw + (size_t)pos * (n >> (s + 2))));
            b3 = fp_mul(b3, fp_load<FrTag>(tw + (size_t)(pos + h) * (n >> (s + 2))));
            *p0 = fp_add(b0, b2);
            *p2 = fp_sub(b0, b2);
            *p1 = fp_add(b1, b3);
            *p3 = fp_sub(b1, b3);
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
            Fr* p0 = sm + (size_t)i0 * sr + (size_t)t * st;
            Fr* p1 = p0 + (size_t)half * sr;
            Fr u = *p0;
            Fr v = *p1;
            if (s) v = fp_mul(v, fp_load<FrTag>(tw + (size_t)pos * (n >> (s + 1))));
            *p0 = fp_add(u, v);
            *p1 = fp_sub(u, v);
        }
        __syncthreads();
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char pz_smem[];

// strided pass.  grid.x = hi * (lo / T), grid.y = column
__global__ __launch_bounds__(256) void k_ntt_strided(const Fr* in, Fr* out, size_t in_stride, size_t out_stride,
                                                     NttPass p, const Fr* __restrict__ tw,
                                                     const Fr* __restrict__ pre) {
    Fr* sm = reinterpret_cast<Fr*>(pz_smem);
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
    for (unsigned idx = threadIdx.x; idx < R * T; idx += blockDim.x) {
        unsigned j = idx >> logT, t = idx & (T - 1);
        size_t g = base + (size_t)j * p.lo + t;
        Fr x = fp_load<FrTag>(src + g);
        if (pre) x = fp_mul(x, fp_load<FrTag>(pre + g));
        sm[(size_t)bitrev32(j, p.logR) * T + t] = x;
    }
    __syncthreads();
    lds_dit(sm, p.logR, T, T, 1, true, tw, p.n);
    for (unsigned idx = threadIdx.x; idx < R * T; idx += blockDim.x) {
        unsigned k = idx >> logT, t = idx & (T - 1);
        Fr x = sm[(size_t)k * T + t];
        size_t e = p.tw_mul * (lt * T + t) * k;  // < n by construction
        if (e) x = fp_mul(x, fp_load<FrTag>(tw + e));
        fp_store(dst + base + (size_t)k * p.lo + t, x);
    }
}

// final pass.  grid.x = n2 * (n1 / T), grid.y = column
__global__ __launch_bounds__(256) void k_ntt_final(const Fr* in, Fr* out, size_t in_stride, size_t out_stride,
                                                   NttPass p, const Fr* __restrict__ tw, const Fr* __restrict__ pre,
                                                   Fr post, int has_post) {
    Fr* sm = reinterpret_cast<Fr*>(pz_smem);
    const unsigned R = 1u << p.logR, T = p.T;
    const unsigned logT = 31u - (unsigned)__builtin_clz(T);
    const size_t tiles = p.n1 >> logT;
    const size_t bx = p.swap ? blockIdx.y : blockIdx.x, by = p.swap ? blockIdx.x : blockIdx.y;
    const size_t k2 = bx / tiles, k1_0 = (bx % tiles) * T;
    const Fr* src = in + by * in_stride;
    Fr* dst = out + by * out_stride;
    for (unsigned idx = threadIdx.x; idx < R * T; idx += blockDim.x) {
        unsigned r = idx >> p.logR, j = idx & (R - 1);
        size_t g = ((k1_0 + r) * p.n2 + k2) * R + j;
        Fr x = fp_load<FrTag>(src + g);
        if (pre) x = fp_mul(x, fp_load<FrTag>(pre + g));
        sm[(size_t)r * R + bitrev32(j, p.logR)] = x;
    }
    __syncthreads();
    lds_dit(sm, p.logR, T, 1, R, false, tw, p.n);
    for (unsigned idx = threadIdx.x; idx < R * T; idx += blockDim.x) {
        unsigned k = idx >> logT, r = idx & (T - 1);
        Fr x = sm[(size_t)r * R + k];
        if (has_post) x = fp_mul(x, post);
        fp_store(dst + (k1_0 + r) + p.n1 * k2 + p.hi * (size_t)k, x);
    }
}

// final pass of the coset-EXTENDED transform (coeff_to_extended): the 2^e * n point NTT of a zero-extended
// n-coefficient polynomial is 2^e independent n-point NTTs of a[i] * (g * w_ext^r)^i, r < 2^e, interleaved
// as out[2^e * q + r].  The strided pass has already run per r (inputs at in + r * in_r_stride); this
// block finishes T rows for ALL r at once so every store is a full T * 2^e * 32-byte run.
__global__ __launch_bounds__(256) void k_ntt_final_ext(const Fr* in, Fr* out, size_t in_stride, size_t in_r_stride,
                                                       size_t out_stride, NttPass p, unsigned log_e,
                                                       const Fr* __restrict__ tw, const Fr* __restrict__ pre,
                                                       size_t pre_r_stride) {
    Fr* sm = reinterpret_cast<Fr*>(pz_smem);
    const unsigned R = 1u << p.logR, T = p.T, E = 1u << log_e;
    const unsigned logT = 31u - (unsigned)__builtin_clz(T);
    const size_t tiles = p.n1 >> logT;
    const size_t bx = p.swap ? blockIdx.y : blockIdx.x, by = p.swap ? blockIdx.x : blockIdx.y;
    const size_t k2 = bx / tiles, k1_0 = (bx % tiles) * T;
    const Fr* src = in + by * in_stride;
    Fr* dst = out + by * out_stride;
    for (unsigned idx = threadIdx.x; idx < R * T * E; idx += blockDim.x) {
        const unsigned j = idx & (R - 1), rr = (idx >> p.logR) & (T - 1), r = idx >> (p.logR + logT);
        const size_t g = ((k1_0 + rr) * p.n2 + k2) * R + j;
        Fr x = fp_load<FrTag>(src + (size_t)r * in_r_stride + g);
        if (pre) x = fp_mul(x, fp_load<FrTag>(pre + (size_t)r * pre_r_stride + g));
        sm[((size_t)r * T + rr) * R + bitrev32(j, p.logR)] = x;
    }
    __syncthreads();
    lds_dit(sm, p.logR, T * E, 1, R, false, tw, p.n);
    for (unsigned idx = threadIdx.x; idx < R * T * E; idx += blockDim.x) {
        const unsigned r = idx & (E - 1), rr = (idx >> log_e) & (T - 1), k = idx >> (log_e + logT);
        Fr x = sm[((size_t)r * T + rr) * R + k];
        fp_store(dst + (((k1_0 + rr) + p.n1 * k2 + p.hi * (size_t)k) << log_e) + r, x);
    }
}

// Lagrange values -> coefficients -> extended coset in ONE chain (what a prover does with every column): this kernel
// is the final pass of the INVERSE transform (rows of R contiguous elements) fused with the first, strided pass of the
// 2^e forward coset transforms.  With the inverse transform factored as n = n1 (strided first) x R (final), a block's
// T rows hold the coefficients t + n1*k (t = row, k < R) -- exactly the tile [k][t] the forward transform's strided
// pass over R needs -- so the coefficients go to HBM once (they are an output) and are never read back: per column
// 64 B * 2^17 less traffic for every one of the 2^e cosets, and 2^e + 1 launches fewer.
__global__ __launch_bounds__(256) void k_ntt_inv_final_fwd_first(const Fr* in, Fr* coeff_out, size_t in_stride,
                                                                 size_t coeff_stride, Fr* ext_tmp, size_t ext_e_stride,
                                                                 NttPass p, unsigned log_e, const Fr* __restrict__ tw_inv,
                                                                 const Fr* __restrict__ tw_fwd, const Fr* __restrict__ pre,
                                                                 Fr post) {
    Fr* sm = reinterpret_cast<Fr*>(pz_smem);
    const unsigned R = 1u << p.logR, T = p.T, E = 1u << log_e;
    const unsigned logT = 31u - (unsigned)__builtin_clz(T);
    const size_t n1 = p.n1;                       // rows of the final pass == lo of the forward strided pass
    const size_t col = blockIdx.x, k1_0 = (size_t)blockIdx.y * T;
    const Fr* src = in + col * in_stride;
    for (unsigned idx = threadIdx.x; idx < R * T; idx += blockDim.x) {
        const unsigned r = idx >> p.logR, j = idx & (R - 1);
        sm[(size_t)r * R + bitrev32(j, p.logR)] = fp_load<FrTag>(src + (k1_0 + r) * R + j);
    }
    __syncthreads();
    lds_dit(sm, p.logR, T, 1, R, false, tw_inv, p.n);
    // the block's coefficients, kept in registers for all cosets (PER = R*T/256 <= 4 elements per thread)
    constexpr unsigned MAXPER = 4;
    Fr v[MAXPER];
#pragma unroll
    for (unsigned i = 0; i < MAXPER; ++i) {
        const unsigned idx = threadIdx.x + i * 256;
        if (idx < R * T) {
            const unsigned k = idx >> logT, r = idx & (T - 1);
            const Fr x = fp_mul(sm[(size_t)r * R + k], post);
            v[i] = x;
            fp_store(coeff_out + col * coeff_stride + (k1_0 + r) + n1 * (size_t)k, x);
        }
    }
    for (unsigned e = 0; e < E; ++e) {
        __syncthreads();
#pragma unroll
        for (unsigned i = 0; i < MAXPER; ++i) {
            const unsigned idx = threadIdx.x + i * 256;
            if (idx < R * T) {
                const unsigned k = idx >> logT, r = idx & (T - 1);
                sm[(size_t)r * R + bitrev32(k, p.logR)] =
                    fp_mul(v[i], fp_load<FrTag>(pre + (size_t)e * p.n + (k1_0 + r) + n1 * (size_t)k));
            }
        }
        __syncthreads();
        lds_dit(sm, p.logR, T, 1, R, false, tw_fwd, p.n);
        Fr* dst = ext_tmp + (size_t)e * ext_e_stride + col * p.n;
#pragma unroll
        for (unsigned i = 0; i < MAXPER; ++i) {
            const unsigned idx = threadIdx.x + i * 256;
            if (idx < R * T) {
                const unsigned k = idx >> logT, r = idx & (T - 1);
                Fr x = sm[(size_t)r * R + k];
                const size_t ex = (k1_0 + r) * (size_t)k;  // inter-pass twiddle of the forward transform, < n
                if (ex) x = fp_mul(x, fp_load<FrTag>(tw_fwd + ex));
                fp_store(dst + (size_t)k * n1 + (k1_0 + r), x);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
#ifndef PZ_NTT_LDS
#define PZ_NTT_LDS 32768
#endif
static unsigned pick_tile(size_t extent, unsigned logR) {
    // LDS budget 64 KiB per block -> R*T*32 <= 65536; prefer 128-byte runs (T = 4) or more
    unsigned T = 8;
    while (T > 1 && (((size_t)32 << logR) * T > PZ_NTT_LDS || T > extent)) T >>= 1;
    return T;
}

static int launch_strided(pz_ctx* ctx, const Fr* in, Fr* out, size_t is, size_t os, size_t ncols, NttPass p,
                          const Fr* tw, const Fr* pre) {
    size_t blocks = p.hi * (p.lo / p.T);
    size_t lds = ((size_t)32 << p.logR) * p.T;
    p.swap = (ncols > 1 && blocks <= 65535) ? 1u : 0u;
    hipLaunchKernelGGL(k_ntt_strided, p.swap ? dim3((unsigned)ncols, (unsigned)blocks) : dim3((unsigned)blocks, (unsigned)ncols),
                       dim3(256), lds, ctx->stream, in, out, is, os, p, tw, pre);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}
static int launch_final(pz_ctx* ctx, const Fr* in, Fr* out, size_t is, size_t os, size_t ncols, NttPass p,
                        const Fr* tw, const Fr* pre, const uint64_t* post_scale) {
    size_t blocks = p.n2 * (p.n1 / p.T);
    size_t lds = ((size_t)32 << p.logR) * p.T;
    Fr post;
    memset(&post, 0, sizeof post);
    if (post_scale) memcpy(post.v, post_scale, 32);
    p.swap = (ncols > 1 && blocks <= 65535) ? 1u : 0u;
    hipLaunchKernelGGL(k_ntt_final, p.swap ? dim3((unsigned)ncols, (unsigned)blocks) : dim3((unsigned)blocks, (unsigned)ncols),
                       dim3(256), lds, ctx->stream, in, out, is, os, p, tw, pre, post, post_scale ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

extern "C" int pz_ntt_fr_dev(pz_ctx* ctx, uint64_t* d_a, size_t n_cols, size_t col_stride, const uint64_t omega[4],
                             uint32_t log_n, const uint64_t* pre_coset_g, const uint64_t* post_scale) {
    if (!ctx || !omega || (n_cols && !d_a) || log_n > 27) return PZ_ERR_INVALID;
    if (n_cols == 0) return PZ_OK;
    const size_t n = (size_t)1 << log_n;
    if (col_stride % 4 || col_stride < 4 * n) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    const size_t cs = col_stride / 4;  // column stride in elements
    Fr* a = reinterpret_cast<Fr*>(d_a);
    void* twv = nullptr;
    PZCHK(pz_get_pow_table(ctx, omega, n, &twv));
    const Fr* tw = (const Fr*)twv;
    void* prev = nullptr;
    if (pre_coset_g) PZCHK(pz_get_pow_table(ctx, pre_coset_g, n, &prev));
    const Fr* pre = (const Fr*)prev;

    const unsigned npass = log_n <= 9 ? 1 : (log_n <= 18 ? 2 : 3);
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
        if (npass == 1) {
            NttPass p{};
            p.logR = log_n; p.lo = 1; p.hi = 1; p.tw_mul = 0; p.n = n; p.T = 1; p.n1 = 1; p.n2 = 1;
            PZCHK(launch_final(ctx, ac, ac, cs, cs, nc, p, tw, pre, post_scale));
        } else if (npass == 2) {
            size_t n1 = (size_t)1 << lg[0], n2 = (size_t)1 << lg[1];
            NttPass pa{};
            pa.logR = lg[0]; pa.lo = n2; pa.hi = 1; pa.tw_mul = 1; pa.n = n; pa.T = pick_tile(n2, lg[0]);
            PZCHK(launch_strided(ctx, ac, tmp, cs, n, nc, pa, tw, pre));
            // tmp columns are packed with stride n
            NttPass pc{};
            pc.logR = lg[1]; pc.lo = 1; pc.hi = n1; pc.n = n; pc.n1 = n1; pc.n2 = 1; pc.T = pick_tile(n1, lg[1]);
            PZCHK(launch_final(ctx, tmp, ac, n, cs, nc, pc, tw, nullptr, post_scale));
        } else {
            size_t n1 = (size_t)1 << lg[0], n2 = (size_t)1 << lg[1], n3 = (size_t)1 << lg[2];
            NttPass pa{};
            pa.logR = lg[0]; pa.lo = n2 * n3; pa.hi = 1; pa.tw_mul = 1; pa.n = n; pa.T = pick_tile(n2 * n3, lg[0]);
            PZCHK(launch_strided(ctx, ac, tmp, cs, n, nc, pa, tw, pre));
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

// E pre-scale tables scale * gens[r]^i, packed [E][n], cached in the context
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
    HIPCHK(ctx, hipMalloc(&prev, E * n * 32));
    for (size_t r = 0; r < E; ++r) {
        void* t;
        PZCHK(pz_get_pow_table(ctx, coset_gens + 4 * r, n, &t, scale));
        HIPCHK(ctx, hipMemcpyAsync((char*)prev + r * n * 32, t, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    }
    ctx->ext_tables.push_back(pz_ext_table{key, prev});
    *out = prev;
    return PZ_OK;
}

// coeff_to_extended in one call: d_ext[col][2^e * q + r] = sum_i d_coeff[col][i] * scale * (gens[r])^i * omega_n^(i q)
// with gens[r] = g * omega_ext^r supplied by the caller (2^log_e x 4 limbs, host), omega_n = omega_ext^(2^log_e).
// Saves the zero-fill, the copy and two butterfly layers of the generic zero-extended transform.
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
    PZCHK(pz_get_pow_table(ctx, omega_n, n, &twv));
    const Fr* tw = (const Fr*)twv;
    void* prev = nullptr;
    PZCHK(get_ext_pre_tables(ctx, coset_gens, E, log_n, scale, &prev));
    const Fr* pre = (const Fr*)prev;
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
            const size_t lds = ((size_t)32 << p.logR) * p.T * E;
            p.swap = 0;
            hipLaunchKernelGGL(k_ntt_final_ext, dim3(1, (unsigned)nc), dim3(256), lds, ctx->stream, cin + c0 * is,
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
            for (size_t r = 0; r < E; ++r) {  // tmp layout [r][col][n]
                PZCHK(launch_strided(ctx, cin + c0 * is, tmp + r * nc * n, is, n, nc, pa, tw, pre + r * n));
                PZCHK(launch_strided(ctx, tmp + r * nc * n, tmp + r * nc * n, n, n, nc, pb, tw, nullptr));
            }
            NttPass pc{};
            pc.logR = lg[2]; pc.lo = 1; pc.hi = n1 * n2; pc.n = n; pc.n1 = n1; pc.n2 = n2;
            unsigned T = 8;
            while (T > 1 && (((size_t)32 << lg[2]) * T * E > 32768 || T > n1)) T >>= 1;
            pc.T = T;
            const size_t lds = ((size_t)32 << lg[2]) * T * E;
            pc.swap = (nc > 1 && n2 * (n1 / T) <= 65535) ? 1u : 0u;
            hipLaunchKernelGGL(k_ntt_final_ext, pc.swap ? dim3((unsigned)nc, (unsigned)(n2 * (n1 / T))) : dim3((unsigned)(n2 * (n1 / T)), (unsigned)nc), dim3(256), lds, ctx->stream, tmp,
                               eout + c0 * os, n, nc * n, os, pc, log_e, tw, (const Fr*)nullptr, (size_t)0);
        } else {
            unsigned lg0 = (log_n + 1) / 2, lg1 = log_n - lg0;
            const size_t n1 = (size_t)1 << lg0, n2 = (size_t)1 << lg1;
            NttPass pa{};
            pa.logR = lg0; pa.lo = n2; pa.hi = 1; pa.tw_mul = 1; pa.n = n; pa.T = pick_tile(n2, lg0);
            for (size_t r = 0; r < E; ++r)  // tmp layout [r][col][n]
                PZCHK(launch_strided(ctx, cin + c0 * is, tmp + r * nc * n, is, n, nc, pa, tw, pre + r * n));
            NttPass pc{};
            pc.logR = lg1; pc.lo = 1; pc.hi = n1; pc.n = n; pc.n1 = n1; pc.n2 = 1;
            // 32 KiB of LDS per block (4 blocks per CU): with 2^e = 4 interleaved outputs a single row already
            // stores full 128-byte lines
            unsigned T = 8;
            while (T > 1 && (((size_t)32 << lg1) * T * E > 32768 || T > n1)) T >>= 1;
            pc.T = T;
            const size_t lds = ((size_t)32 << lg1) * T * E;
            pc.swap = nc > 1 ? 1u : 0u;
            hipLaunchKernelGGL(k_ntt_final_ext, pc.swap ? dim3((unsigned)nc, (unsigned)(n1 / T)) : dim3((unsigned)(n1 / T), (unsigned)nc), dim3(256), lds, ctx->stream, tmp,
                               eout + c0 * os, n, nc * n, os, pc, log_e, tw, (const Fr*)nullptr, (size_t)0);
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
    if (log_n < 10 || log_n > 18) {  // one- and three-pass sizes: the two transforms back to back
        PZCHK(pz_ntt_fr_dev(ctx, d_values, n_cols, col_stride, omega_n_inv, log_n, nullptr, n_inv));
        return pz_ntt_fr_extend_dev(ctx, d_values, n_cols, col_stride, d_ext, out_stride, log_n, log_e, omega_n, coset_gens, nullptr);
    }
    PZ_ENTER(ctx);
    const size_t cs = col_stride / 4, os = out_stride / 4;
    void *twf, *twi, *prev;
    PZCHK(pz_get_pow_table(ctx, omega_n, n, &twf));
    PZCHK(pz_get_pow_table(ctx, omega_n_inv, n, &twi));
    PZCHK(get_ext_pre_tables(ctx, coset_gens, E, log_n, nullptr, &prev));
    // inverse transform: strided pass over 2^lgA first, final (fused) pass over R = 2^lgB rows; forward: strided over R
    // (fused), final over 2^lgA
    const unsigned lgB = (log_n + 1) / 2, lgA = log_n - lgB;
    const size_t nA = (size_t)1 << lgA, nB = (size_t)1 << lgB;
    size_t group = n_cols;
    const size_t max_ws = (size_t)2 << 30;
    if (group * n * (E + 1) * 32 > max_ws) group = max_ws / (n * (E + 1) * 32) ? max_ws / (n * (E + 1) * 32) : 1;
    if (group > 32768) group = 32768;
    void* t;
    PZCHK(pz_ws_get(ctx, WS_NTT_TMP, group * n * (E + 1) * 32, &t));
    Fr* tmp0 = (Fr*)t;                 // [col][n]: inverse transform after its strided pass
    Fr* tmpe = tmp0 + group * n;       // [e][col][n]: forward transforms after their strided pass
    Fr post;
    memcpy(post.v, n_inv, 32);
    Fr* a = (Fr*)d_values;
    Fr* eout = (Fr*)d_ext;
    pz_timer tm(ctx, PZ_T_NTT);
    for (size_t c0 = 0; c0 < n_cols; c0 += group) {
        const size_t nc = n_cols - c0 < group ? n_cols - c0 : group;
        NttPass pa{};
        pa.logR = lgA; pa.lo = nB; pa.hi = 1; pa.tw_mul = 1; pa.n = n; pa.T = pick_tile(nB, lgA);
        PZCHK(launch_strided(ctx, a + c0 * cs, tmp0, cs, n, nc, pa, (const Fr*)twi, nullptr));
        NttPass pf{};
        pf.logR = lgB; pf.lo = 1; pf.hi = nA; pf.n = n; pf.n1 = nA; pf.n2 = 1;
        unsigned T = 8;
        while (T > 1 && (((size_t)32 << lgB) * T > PZ_NTT_LDS || T > nA || (((size_t)T << lgB) > 1024))) T >>= 1;
        pf.T = T;
        const size_t lds = ((size_t)32 << lgB) * T;
        hipLaunchKernelGGL(k_ntt_inv_final_fwd_first, dim3((unsigned)nc, (unsigned)(nA / T)), dim3(256), lds, ctx->stream,
                           (const Fr*)tmp0, a + c0 * cs, n, cs, tmpe, nc * n, pf, log_e, (const Fr*)twi, (const Fr*)twf,
                           (const Fr*)prev, post);
        NttPass pc{};
        pc.logR = lgA; pc.lo = 1; pc.hi = nB; pc.n = n; pc.n1 = nB; pc.n2 = 1;
        unsigned Tc = 8;
        while (Tc > 1 && (((size_t)32 << lgA) * Tc * E > 32768 || Tc > nB)) Tc >>= 1;
        pc.T = Tc;
        pc.swap = nc > 1 ? 1u : 0u;
        const size_t ldsc = ((size_t)32 << lgA) * Tc * E;
        hipLaunchKernelGGL(k_ntt_final_ext, pc.swap ? dim3((unsigned)nc, (unsigned)(nB / Tc)) : dim3((unsigned)(nB / Tc), (unsigned)nc),
                           dim3(256), ldsc, ctx->stream, tmpe, eout + c0 * os, n, nc * n, os, pc, log_e, (const Fr*)twf,
                           (const Fr*)nullptr, (size_t)0);
        HIPCHK(ctx, hipGetLastError());
    }
    return PZ_OK;
}

// host-pointer forms: == best_fft(a, omega, log_n) on each column
extern "C" int pz_ntt_fr_batch(pz_ctx* ctx, uint64_t* const* cols, size_t n_cols, const uint64_t omega[4],
                               uint32_t log_n) {
    if (!ctx || !omega || (n_cols && !cols) || log_n > 27) return PZ_ERR_INVALID;
    if (n_cols == 0) return PZ_OK;
    PZ_ENTER(ctx);
    const size_t n = (size_t)1 << log_n, bytes = n * 32;
    // stage in groups of <= 1 GiB
    size_t group = ((size_t)1 << 30) / bytes;
    if (group == 0) group = 1;
    if (group > n_cols) group = n_cols;
    void* d;
    PZCHK(pz_ws_get(ctx, WS_IO_A, group * bytes, &d));
    for (size_t c0 = 0; c0 < n_cols; c0 += group) {
        size_t nc = n_cols - c0 < group ? n_cols - c0 : group;
        for (size_t j = 0; j < nc; ++j) {
            if (!cols[c0 + j]) return PZ_ERR_INVALID;
            HIPCHK(ctx, hipMemcpyAsync((char*)d + j * bytes, cols[c0 + j], bytes, hipMemcpyHostToDevice, ctx->stream));
        }
        PZCHK(pz_ntt_fr_dev(ctx, (uint64_t*)d, nc, 4 * n, omega, log_n, nullptr, nullptr));
        for (size_t j = 0; j < nc; ++j)
            HIPCHK(ctx, hipMemcpyAsync(cols[c0 + j], (char*)d + j * bytes, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PZ_OK;
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
