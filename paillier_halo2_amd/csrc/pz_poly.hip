// pz_poly.hip -- first "next" rows of SURVEY.md section 8f, built on K1/K2's primitives:
//   pz_srs_setup_g1_dev : ParamsKZG::setup as halo2 does it when the toxic scalar s is known (gen_srs seeds it):
//                         g[i] = [s^i] G and g_lagrange[i] = [L_i(s)] G with L_i(s) = (s^n - 1) w^i / (n (s - w^i)),
//                         i.e. Fr arithmetic + fixed-base multiplication, all on the device.
//   pz_poly_eval_dev    : evaluation of coefficient-form columns at a point x (the "evals" phase before SHPLONK):
//                         sum_i c_i x^i with a cached x^i table, one multiplication per coefficient.
// Both are reached in the reference only through bench.rs:161-171 (gen_srs / create_proof inside bench_builder).
#include "ec.cuh"
#include "fp29.cuh"
#include "pz_internal.h"

__global__ void k_srs_multiplier(Fr s, Fr n_mont, unsigned k, Fr* out /* [0] = (s^n - 1)/n, [1] = s^n - 1 */) {
    if (blockIdx.x || threadIdx.x) return;
    Fr sn = s;
    for (unsigned i = 0; i < k; ++i) sn = fp_sqr(sn);
    Fr num = fp_sub(sn, fp_one<FrTag>());
    fp_store(out + 1, num);
    fp_store(out, fp_mul(num, fp_inv(n_mont)));
}

// l_i = mult * w^i / (s - w^i)
__global__ __launch_bounds__(128) void k_lagrange_scalars(const Fr* __restrict__ tw, Fr s, const Fr* __restrict__ mult,
                                                          Fr* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr w = fp_load<FrTag>(tw + i);
    Fr d = fp_sub(s, w);
    Fr m = fp_load<FrTag>(mult);
    fp_store(out + i, fp_mul(fp_mul(m, w), fp_inv(d)));
}

extern "C" int pz_g1_fixed_base_mul_dev(pz_ctx* ctx, const uint64_t* d_scalars, size_t n, uint64_t* d_out_affine);

extern "C" int pz_srs_setup_g1_dev(pz_ctx* ctx, uint32_t k, const uint64_t s[4], const uint64_t omega[4], uint64_t* d_g,
                                   uint64_t* d_g_lagrange) {
    if (!ctx || !s || !omega || (!d_g && !d_g_lagrange) || k > 26) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    const size_t n = (size_t)1 << k;
    void* spow;
    PZCHK(pz_get_pow_table(ctx, s, n, &spow));
    if (d_g) PZCHK(pz_g1_fixed_base_mul_dev(ctx, (const uint64_t*)spow, n, d_g));
    if (d_g_lagrange) {
        void *tw, *ws;
        PZCHK(pz_get_pow_table(ctx, omega, n, &tw));
        PZCHK(pz_ws_get(ctx, WS_IO_A, n * 32 + 64, &ws));
        Fr* mult = (Fr*)ws;
        Fr* lag = (Fr*)((char*)ws + 64);
        Fr sv, nm;
        memcpy(sv.v, s, 32);
        // Montgomery form of n = 2^k: computed on the device side as 2^k * R via doubling of R
        {
            // R mod r doubled k times on the host would need field arithmetic; do it in the kernel via to_mont
            Fr raw;
            memset(&raw, 0, sizeof raw);
            raw.v[k >> 5] = 1u << (k & 31);
            nm = raw;  // canonical 2^k; converted below
        }
        // canonical -> Montgomery for n: reuse the conversion kernel on a 1-element buffer
        HIPCHK(ctx, hipMemcpyAsync(mult + 1, &nm, 32, hipMemcpyHostToDevice, ctx->stream));
        PZCHK(pz_fr_convert_dev(ctx, (uint64_t*)(mult + 1), 1, 1));
        HIPCHK(ctx, hipMemcpyAsync(&nm, mult + 1, 32, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        hipLaunchKernelGGL(k_srs_multiplier, dim3(1), dim3(64), 0, ctx->stream, sv, nm, (unsigned)k, mult);
        Fr chk[2];
        HIPCHK(ctx, hipMemcpyAsync(chk, mult, 64, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        bool zero = true;
        for (int i = 0; i < 8; ++i) zero = zero && chk[1].v[i] == 0;
        if (zero) return PZ_ERR_INVALID;  // s lies in the evaluation domain (s^n == 1): halo2 special-cases it, not supported
        hipLaunchKernelGGL(k_lagrange_scalars, dim3(pz_div_up(n, 128)), dim3(128), 0, ctx->stream, (const Fr*)tw, sv,
                           (const Fr*)mult, lag, n);
        HIPCHK(ctx, hipGetLastError());
        PZCHK(pz_g1_fixed_base_mul_dev(ctx, (const uint64_t*)lag, n, d_g_lagrange));
    }
    return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- evaluation at a point
#define EVAL_CH 16u
__global__ void k_poly_eval_final(const Fr* __restrict__ partial, unsigned blocks_per_col, size_t n_cols, Fr* __restrict__ out) {
    size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    Fr acc = fp_zero<FrTag>();
    for (unsigned b = 0; b < blocks_per_col; ++b) acc = fp_add(acc, fp_load<FrTag>(partial + col * blocks_per_col + b));
    fp_store(out + col, acc);
}

extern "C" int pz_poly_eval_multi_dev(pz_ctx* ctx, const uint64_t* d_coeffs, size_t n_cols, size_t col_stride, size_t n,
                                      const uint64_t* xs, uint32_t n_points, uint64_t* d_out);
extern "C" int pz_poly_eval_dev(pz_ctx* ctx, const uint64_t* d_coeffs, size_t n_cols, size_t col_stride, size_t n,
                                const uint64_t x[4], uint64_t* d_out) {
    if (!ctx || !x || (n_cols && (!d_coeffs || !d_out)) || col_stride % 4 || (n_cols > 1 && col_stride < 4 * n)) return PZ_ERR_INVALID;
    if (n_cols == 0) return PZ_OK;
    if (n_cols > 65535) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    if (n == 0) {
        HIPCHK(ctx, hipMemsetAsync(d_out, 0, n_cols * 32, ctx->stream));
        return PZ_OK;
    }
    return pz_poly_eval_multi_dev(ctx, d_coeffs, n_cols, col_stride, n, x, 1, d_out);
}

// The same at up to four points at once -- a polynomial's rotation set {x, wx, w^2 x, ...} (halo2 evaluates every advice
// polynomial at each rotation its gates query): the coefficient is read ONCE and multiplied into P accumulators, against P
// cached power tables.  One point at a time the evaluations of a c2 proof read 98 GB of coefficients; this way 20 GB.
template <unsigned P>
__global__ __launch_bounds__(256) void k_poly_eval_partial_multi(const Fr* __restrict__ coeffs, size_t col_stride, size_t n,
                                                                 const Fr* __restrict__ xp0, const Fr* __restrict__ xp1,
                                                                 const Fr* __restrict__ xp2, const Fr* __restrict__ xp3,
                                                                 Fr* __restrict__ partial, unsigned blocks_per_col) {
    __shared__ Fr s_acc[256];
    const size_t col = blockIdx.y;
    const Fr* c = coeffs + col * col_stride;
    const Fr* xp[4] = {xp0, xp1, xp2, xp3};
    // a thread's EVAL_CH coefficients lie 256 apart, so a wave's load is 64 consecutive 32-byte elements (any four terms may share a
    // reduction: the sum does not care which).  With 16 CONSECUTIVE coefficients per thread every lane of a load sat in its own
    // 512-byte segment and the lines were fetched again for each of their four quarters.
    const size_t base = (size_t)blockIdx.x * 256 * EVAL_CH + threadIdx.x;
    // products on the 9 x 29-bit field (fp29.cuh): the power tables are kept in the 2^261 domain, so coefficient (2^256
    // domain) x table entry stays in the ABI's domain.  Four terms share ONE Montgomery reduction (f29_dot4: 4 x 81 + 90
    // multiplier instructions instead of 4 x 171); the EVAL_CH / 4 results are added limb-wise (limbs < 4 * 2^29, values < 8p)
    F29<FrTag> acc[P];
#pragma unroll
    for (unsigned q = 0; q < P; ++q) acc[q] = f29_zero<FrTag>();
#pragma unroll 2
    for (unsigned t = 0; t < EVAL_CH; t += 4) {
        if (base + (size_t)t * 256 >= n) break;
        F29<FrTag> v[4];
        size_t idx[4];
#pragma unroll
        for (unsigned k = 0; k < 4; ++k) {
            idx[k] = base + (size_t)(t + k) * 256;
            v[k] = idx[k] < n ? f29_load<FrTag>(c + idx[k]) : f29_zero<FrTag>();
        }
#pragma unroll
        for (unsigned q = 0; q < P; ++q) {
            F29<FrTag> x[4];
#pragma unroll
            for (unsigned k = 0; k < 4; ++k) x[k] = f29_load<FrTag>(xp[q] + (idx[k] < n ? idx[k] : n - 1));
            acc[q] = f29_add(acc[q], f29_dot4(v, x));
        }
    }
#pragma unroll
    for (unsigned q = 0; q < P; ++q) {
        s_acc[threadIdx.x] = f29_to_fp<5>(acc[q]);
        __syncthreads();
        for (unsigned off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) s_acc[threadIdx.x] = fp_add(s_acc[threadIdx.x], s_acc[threadIdx.x + off]);
            __syncthreads();
        }
        if (threadIdx.x == 0) fp_store(partial + ((col * P + q) * blocks_per_col) + blockIdx.x, s_acc[0]);
        __syncthreads();
    }
}

extern "C" int pz_poly_eval_multi_dev(pz_ctx* ctx, const uint64_t* d_coeffs, size_t n_cols, size_t col_stride, size_t n,
                                      const uint64_t* xs, uint32_t n_points, uint64_t* d_out) {
    if (!ctx || !xs || n_points == 0 || n_points > 4 || (n_cols && (!d_coeffs || !d_out)) || col_stride % 4 ||
        (n_cols > 1 && col_stride < 4 * n))
        return PZ_ERR_INVALID;
    if (n_cols == 0) return PZ_OK;
    if (n_cols > 65535) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    if (n == 0) {
        HIPCHK(ctx, hipMemsetAsync(d_out, 0, n_cols * (size_t)n_points * 32, ctx->stream));
        return PZ_OK;
    }
    void* xp[4] = {nullptr, nullptr, nullptr, nullptr};
    for (uint32_t q = 0; q < n_points; ++q) PZCHK(pz_get_pow_table(ctx, xs + 4 * q, n, &xp[q], pz_fr_one261()));
    for (uint32_t q = n_points; q < 4; ++q) xp[q] = xp[0];
    const unsigned bpc = pz_div_up(n, 256 * EVAL_CH);
    void* part;
    PZCHK(pz_ws_get(ctx, WS_IO_B, n_cols * (size_t)n_points * bpc * 32, &part));
    const dim3 grid(bpc, (unsigned)n_cols);
#define PZ_EVAL_LAUNCH(P_)                                                                                                   \
    hipLaunchKernelGGL(k_poly_eval_partial_multi<P_>, grid, dim3(256), 0, ctx->stream, (const Fr*)d_coeffs, col_stride / 4, n, \
                       (const Fr*)xp[0], (const Fr*)xp[1], (const Fr*)xp[2], (const Fr*)xp[3], (Fr*)part, bpc)
    switch (n_points) {
        case 1: PZ_EVAL_LAUNCH(1); break;
        case 2: PZ_EVAL_LAUNCH(2); break;
        case 3: PZ_EVAL_LAUNCH(3); break;
        default: PZ_EVAL_LAUNCH(4); break;
    }
#undef PZ_EVAL_LAUNCH
    // partial[(col * P + q)][block]: the reduction over blocks is the single-point one over n_cols * P rows
    hipLaunchKernelGGL(k_poly_eval_final, dim3(pz_div_up(n_cols * n_points, 64)), dim3(64), 0, ctx->stream, (const Fr*)part, bpc,
                       n_cols * (size_t)n_points, (Fr*)d_out);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- keygen
// permutation::keygen: the sigma polynomial of column j holds, at row i, the label delta^(col') * omega^(row') of the cell
// the copy-constraint cycle maps (j, i) to.  The cycles themselves are circuit structure (the reference's dependency
// builds them while synthesising); they arrive as two index arrays.
// (the map is the caller's device data: an image outside the m x n cells -- e.g. a map cut into column batches whose images still
// name columns of the whole permutation -- would index past the power tables; it is clamped and reported through the context's
// asynchronous-failure word: the next synchronising entry point returns PZ_ERR_ASYNC)
__global__ __launch_bounds__(256) void k_perm_sigma(const u32* __restrict__ map_col, const u32* __restrict__ map_row, size_t n,
                                                    size_t total, unsigned m, const Fr* __restrict__ wpow, const Fr* __restrict__ dpow,
                                                    Fr* __restrict__ sigma, size_t stride, volatile unsigned* err) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t j = t / n, i = t % n;
    u32 c = map_col[t], r = map_row[t];
    if (c >= m || r >= n) {
        *err = 1u;
        c = 0;
        r = 0;
    }
    fp_store(sigma + j * stride + i, fp_mul(fp_load<FrTag>(dpow + c), fp_load<FrTag>(wpow + r)));
}

extern "C" int pz_permutation_sigma_dev(pz_ctx* ctx, const uint32_t* d_map_col, const uint32_t* d_map_row, size_t m, uint32_t k,
                                        const uint64_t omega[4], const uint64_t delta[4], uint64_t* d_sigma, size_t sigma_stride) {
    if (!ctx || !d_map_col || !d_map_row || !omega || !delta || !d_sigma || k > 26 || m == 0 || m > 0xffffffu) return PZ_ERR_INVALID;
    const size_t n = (size_t)1 << k;
    if (sigma_stride % 4 || (m > 1 && sigma_stride < 4 * n)) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    void *wp, *dp;
    PZCHK(pz_get_pow_table(ctx, omega, n, &wp));
    PZCHK(pz_get_pow_table(ctx, delta, m, &dp));
    PZCHK(pz_async_err_init(ctx));
    hipLaunchKernelGGL(k_perm_sigma, dim3(pz_div_up(m * n, 256)), dim3(256), 0, ctx->stream, d_map_col, d_map_row, n, m * n, (unsigned)m,
                       (const Fr*)wp, (const Fr*)dp, (Fr*)d_sigma, sigma_stride / 4, ctx->async_err_d);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

// keygen_vk + keygen_pk for a batch of Lagrange-form fixed columns (selectors, constants, lookup table, sigma): their
// commitments (commit_lagrange), coefficient forms (in place) and extended-coset forms -- the proving key's polynomials,
// left RESIDENT in the caller's device buffers for every proof that follows.
extern "C" int pz_msm_g1_dev(pz_ctx* ctx, const pz_bases* bases, const uint64_t* d_scalars, size_t n_cols, size_t n, size_t col_stride,
                             uint32_t win_lo, uint32_t win_hi, uint64_t* d_out_jac);
// selectors cross the ABI as bytes (0 / 1) and become Lagrange-form fixed columns here: out[i] = mask[i] ? 1 : 0 in Montgomery form.
// HBM-bound elementwise work: 1 byte read, 32 written per element; a thread takes four consecutive elements (one 4-byte load, four
// 32-byte stores that the wave lays down as whole 128-byte lines).
__global__ __launch_bounds__(256) void k_fr_from_mask(const uint8_t* __restrict__ mask, size_t n, Fr* __restrict__ out) {
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    const Fr one = fp_one<FrTag>(), zero = fp_zero<FrTag>();
    if (i0 + 4 <= n && ((uintptr_t)(mask + i0) & 3) == 0) {
        const u32 w = *(const u32*)(mask + i0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fp_store(out + i0 + j, ((w >> (8 * j)) & 0xffu) ? one : zero);
    } else {
        for (size_t i = i0; i < n && i < i0 + 4; ++i) fp_store(out + i, mask[i] ? one : zero);
    }
}
extern "C" int pz_fr_from_mask_dev(pz_ctx* ctx, const uint8_t* d_mask, size_t n, uint64_t* d_out) {
    if (!ctx || (n && (!d_mask || !d_out))) return PZ_ERR_INVALID;
    if (!n) return PZ_OK;
    PZ_ENTER(ctx);
    hipLaunchKernelGGL(k_fr_from_mask, dim3((unsigned)pz_div_up(pz_div_up(n, 4), 256)), dim3(256), 0, ctx->stream, d_mask, n, (Fr*)d_out);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

extern "C" int pz_keygen_columns_dev(pz_ctx* ctx, const pz_bases* bases_lagrange, uint64_t* d_cols, size_t n_cols, size_t col_stride,
                                     uint32_t k, uint32_t log_e, const uint64_t omega_n[4], const uint64_t omega_n_inv[4],
                                     const uint64_t n_inv[4], const uint64_t* coset_gens, uint64_t* d_commit_jac, uint64_t* d_ext,
                                     size_t ext_stride) {
    if (!ctx || !bases_lagrange || !d_cols || !d_commit_jac || k > 26) return PZ_ERR_INVALID;
    const size_t n = (size_t)1 << k;
    if (bases_lagrange->n < n) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    PZCHK(pz_msm_g1_dev(ctx, bases_lagrange, d_cols, n_cols, n, col_stride, 0, bases_lagrange->nwin, d_commit_jac));
    PZCHK(pz_ntt_fr_dev(ctx, d_cols, n_cols, col_stride, omega_n_inv, k, nullptr, n_inv));
    if (d_ext) PZCHK(pz_ntt_fr_extend_dev(ctx, d_cols, n_cols, col_stride, d_ext, ext_stride, k, log_e, omega_n, coset_gens, nullptr));
    return PZ_OK;
}
