// pz_lookup.hip -- SURVEY.md section 8f rank 1, lookup argument: halo2's lookup::prover `permute_expression_pair`
// and the lookup grand product, for the shape halo2-lib's RangeChip produces (reached in the reference through
// create_proof, bench.rs:161-171; RangeChip is built at paillier.rs:168-169 / bench.rs:162-163): ONE input column
// of range-check digits per argument, looked up in the table column {0 .. 2^lookup_bits - 1} (zero-padded).
//
// halo2 sorts the input column and lays the table out beside it so that every row either starts a run of equal
// inputs (table cell == input cell) or takes a left-over table value, left-overs in ascending order.  With values
// below 2^value_bits this is a counting sort, not a comparison sort:
//   counts  cA[v], cS[v]               (global atomics; one histogram per column, one for the shared table)
//   scans   oA = offsets of the sorted input, dr = rank among the values present, lS = cS - [cA > 0] (a negative
//           entry means an input value is missing from the table: the reference's prover fails there), oL
//   emit    row i: v = the value whose run contains i (binary search in oA); A'[i] = v; S'[i] = v on the first row
//           of a run, otherwise the (i - dr[v] - 1)-th left-over (binary search in oL)
// then the product z[i+1] = z[i] (A[i] + beta)(S[i] + gamma) / ((A'[i] + beta)(S'[i] + gamma)) reuses the batch
// inversion and running product of pz_quotient.hip.  Values that are not canonical integers below 2^value_bits are
// reported as PZ_ERR_RANGE (general Fr-valued lookups would need a 256-bit key sort: not built).
#include "fp.cuh"
#include "pz_internal.h"

int pz_batch_invert_internal(pz_ctx* ctx, Fr* d_a, size_t n, Fr* d_mul_io);
int pz_prefix_product_batch_internal(pz_ctx* ctx, const Fr* d_a, size_t a_stride, size_t n_cols, size_t n, Fr z0, Fr* d_z,
                                     size_t z_stride);

// canonical value of x if it is an integer below M, else 0xffffffff
__device__ __forceinline__ u32 small_value(const Fr& mont, u32 M) {
    Fr c = fp_from_mont(mont);
    u32 hi = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) hi |= c.v[k];
    return (hi == 0 && c.v[0] < M) ? c.v[0] : 0xffffffffu;
}
__device__ __forceinline__ Fr fr_small(u32 v) {
    Fr x = fp_zero<FrTag>();
    if (!v) return x;
    x.v[0] = v;
    return fp_to_mont(x);
}

__global__ __launch_bounds__(256) void k_lk_hist(const Fr* __restrict__ cols, size_t cs, size_t rows, u32 M,
                                                 u32* __restrict__ counts, u32* __restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const u32 v = small_value(fp_load<FrTag>(cols + (size_t)blockIdx.y * cs + i), M);
    if (v == 0xffffffffu) atomicOr(flags, 1u);
    else atomicAdd(counts + (size_t)blockIdx.y * M + v, 1u);
}

// one workgroup per column: exclusive scans over the M values
__global__ __launch_bounds__(256) void k_lk_scan(const u32* __restrict__ cA, const u32* __restrict__ cS, u32 M,
                                                 u32* __restrict__ oA, u32* __restrict__ dr, u32* __restrict__ oL,
                                                 u32* __restrict__ flags) {
    __shared__ u32 s_a[256], s_d[256], s_l[256];
    const size_t col = blockIdx.x;
    const u32* a = cA + col * M;
    const u32 per = (M + 255) / 256, lo = threadIdx.x * per;
    u32 ta = 0, td = 0, tl = 0;
    bool missing = false;
    for (u32 k = 0; k < per; ++k) {
        const u32 v = lo + k;
        if (v < M) {
            const u32 ca = a[v], cs = cS[v], present = ca ? 1u : 0u;
            if (present > cs) missing = true;
            ta += ca;
            td += present;
            tl += cs - (present <= cs ? present : cs);
        }
    }
    if (missing) atomicOr(flags, 2u);
    s_a[threadIdx.x] = ta; s_d[threadIdx.x] = td; s_l[threadIdx.x] = tl;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 ra = 0, rd = 0, rl = 0;
        for (int k = 0; k < 256; ++k) {
            u32 x = s_a[k]; s_a[k] = ra; ra += x;
            x = s_d[k]; s_d[k] = rd; rd += x;
            x = s_l[k]; s_l[k] = rl; rl += x;
        }
    }
    __syncthreads();
    ta = s_a[threadIdx.x]; td = s_d[threadIdx.x]; tl = s_l[threadIdx.x];
    for (u32 k = 0; k < per; ++k) {
        const u32 v = lo + k;
        if (v < M) {
            const u32 ca = a[v], cs = cS[v], present = ca ? 1u : 0u;
            oA[col * (M + 1) + v] = ta;
            dr[col * M + v] = td;
            oL[col * (M + 1) + v] = tl;
            ta += ca;
            td += present;
            tl += cs - (present <= cs ? present : cs);
        }
    }
    if (threadIdx.x == 255) {
        oA[col * (M + 1) + M] = ta;
        oL[col * (M + 1) + M] = tl;
    }
}

// largest v < M with off[v] <= i   (off is non-decreasing, off[0] = 0)
__device__ __forceinline__ u32 run_of(const u32* __restrict__ off, u32 M, u32 i) {
    u32 lo = 0, hi = M;  // invariant: off[lo] <= i, (hi == M or off[hi] > i)
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_lk_emit(const u32* __restrict__ oA, const u32* __restrict__ dr,
                                                 const u32* __restrict__ oL, u32 M, size_t rows, Fr* __restrict__ pin,
                                                 Fr* __restrict__ ptab, size_t os) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const size_t col = blockIdx.y;
    const u32* offA = oA + col * (M + 1);
    const u32 v = run_of(offA, M, (u32)i);
    const Fr fv = fr_small(v);
    fp_store(pin + col * os + i, fv);
    if (offA[v] == (u32)i) {
        fp_store(ptab + col * os + i, fv);
    } else {
        const u32 r = (u32)i - dr[col * M + v] - 1;  // rank among the rows that repeat their predecessor
        fp_store(ptab + col * os + i, fr_small(run_of(oL + col * (M + 1), M, r)));
    }
}

extern "C" int pz_lookup_permute_dev(pz_ctx* ctx, const uint64_t* d_inputs, size_t n_cols, size_t col_stride,
                                     const uint64_t* d_table, size_t rows, uint32_t value_bits, uint64_t* d_perm_inputs,
                                     uint64_t* d_perm_tables, size_t out_stride) {
    if (!ctx || value_bits == 0 || value_bits > 24 || col_stride % 4 || out_stride % 4) return PZ_ERR_INVALID;
    if (n_cols && rows && (!d_inputs || !d_table || !d_perm_inputs || !d_perm_tables)) return PZ_ERR_INVALID;
    if (n_cols > 1 && (col_stride < 4 * rows || out_stride < 4 * rows)) return PZ_ERR_INVALID;
    if (n_cols == 0 || rows == 0) return PZ_OK;
    if (n_cols > 65535 || rows > 0xfffffff0u) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    const u32 M = 1u << value_bits;
    // workspace: cA [n_cols][M] | cS [M] | oA [n_cols][M+1] | dr [n_cols][M] | oL [n_cols][M+1] | flags
    const size_t words = n_cols * (size_t)M + M + 2 * n_cols * (size_t)(M + 1) + n_cols * (size_t)M + 4;
    void* ws;
    PZCHK(pz_ws_get(ctx, WS_BIG_C, words * 4, &ws));
    u32* cA = (u32*)ws;
    u32* cS = cA + n_cols * (size_t)M;
    u32* oA = cS + M;
    u32* dr = oA + n_cols * (size_t)(M + 1);
    u32* oL = dr + n_cols * (size_t)M;
    u32* flags = oL + n_cols * (size_t)(M + 1);
    HIPCHK(ctx, hipMemsetAsync(cA, 0, (n_cols + 1) * (size_t)M * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(flags, 0, 16, ctx->stream));
    hipLaunchKernelGGL(k_lk_hist, dim3(pz_div_up(rows, 256), (unsigned)n_cols), dim3(256), 0, ctx->stream,
                       (const Fr*)d_inputs, col_stride / 4, rows, M, cA, flags);
    hipLaunchKernelGGL(k_lk_hist, dim3(pz_div_up(rows, 256), 1), dim3(256), 0, ctx->stream, (const Fr*)d_table, (size_t)0,
                       rows, M, cS, flags);
    hipLaunchKernelGGL(k_lk_scan, dim3((unsigned)n_cols), dim3(256), 0, ctx->stream, (const u32*)cA, (const u32*)cS, M, oA,
                       dr, oL, flags);
    HIPCHK(ctx, hipGetLastError());
    u32 f = 0;
    HIPCHK(ctx, hipMemcpyAsync(&f, flags, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (f) return PZ_ERR_RANGE;  // a value outside [0, 2^value_bits), or an input value absent from the table
    hipLaunchKernelGGL(k_lk_emit, dim3(pz_div_up(rows, 256), (unsigned)n_cols), dim3(256), 0, ctx->stream, (const u32*)oA,
                       (const u32*)dr, (const u32*)oL, M, rows, (Fr*)d_perm_inputs, (Fr*)d_perm_tables, out_stride / 4);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

// ------------------------------------------------------------------------------------------------ lookup product
__global__ __launch_bounds__(256) void k_lk_terms(const Fr* __restrict__ A, size_t as, const Fr* __restrict__ S,
                                                  const Fr* __restrict__ Ap, size_t aps, const Fr* __restrict__ Sp, size_t sps,
                                                  size_t n, Fr beta, Fr gamma, Fr* __restrict__ num, Fr* __restrict__ den) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t k = blockIdx.y;
    fp_store(num + k * n + i, fp_mul(fp_add(fp_load<FrTag>(A + k * as + i), beta), fp_add(fp_load<FrTag>(S + i), gamma)));
    fp_store(den + k * n + i, fp_mul(fp_add(fp_load<FrTag>(Ap + k * aps + i), beta), fp_add(fp_load<FrTag>(Sp + k * sps + i), gamma)));
}

extern "C" int pz_lookup_product_dev(pz_ctx* ctx, const uint64_t* d_inputs, size_t input_stride, const uint64_t* d_table,
                                     const uint64_t* d_perm_inputs, size_t perm_input_stride, const uint64_t* d_perm_tables,
                                     size_t perm_table_stride, size_t n_lookups, size_t n, const uint64_t beta[4],
                                     const uint64_t gamma[4], const uint64_t z0[4], uint64_t* d_z, size_t z_stride) {
    if (!ctx || !beta || !gamma || !z0 || (n && n_lookups && (!d_inputs || !d_table || !d_perm_inputs || !d_perm_tables || !d_z)))
        return PZ_ERR_INVALID;
    if (input_stride % 4 || perm_input_stride % 4 || perm_table_stride % 4 || z_stride % 4 || n_lookups > 65535) return PZ_ERR_INVALID;
    if (n_lookups > 1 && (input_stride < 4 * n || perm_input_stride < 4 * n || perm_table_stride < 4 * n || z_stride < 4 * n))
        return PZ_ERR_INVALID;
    if (!n || !n_lookups) return PZ_OK;
    PZ_ENTER(ctx);
    void* ws;
    PZCHK(pz_ws_get(ctx, WS_BIG_C, 2 * n_lookups * n * 32, &ws));
    Fr* num = (Fr*)ws;
    Fr* den = num + n_lookups * n;
    Fr b, g, z;
    memcpy(b.v, beta, 32);
    memcpy(g.v, gamma, 32);
    memcpy(z.v, z0, 32);
    hipLaunchKernelGGL(k_lk_terms, dim3(pz_div_up(n, 256), (unsigned)n_lookups), dim3(256), 0, ctx->stream, (const Fr*)d_inputs,
                       input_stride / 4, (const Fr*)d_table, (const Fr*)d_perm_inputs, perm_input_stride / 4,
                       (const Fr*)d_perm_tables, perm_table_stride / 4, n, b, g, num, den);
    HIPCHK(ctx, hipGetLastError());
    PZCHK(pz_batch_invert_internal(ctx, den, n_lookups * n, num));   // num <- num / den
    return pz_prefix_product_batch_internal(ctx, num, n, n_lookups, n, z, (Fr*)d_z, z_stride / 4);
}
