// pz_shplonk.hip -- SHPLONK multi-point opening on the device: the batching of halo2's
// multiopen::shplonk::prover that turns every (polynomial, point, evaluation) query of a proof into TWO polynomials
// whose commitments end the proof.  Reached in the reference through create_proof (/root/reference/src/bench.rs:165);
// the algorithm restates the halo2 dependency (tag [D]) and is pinned by the opening identity it must satisfy
// (tests/test_gpu_next_rows.py, oracle/pyref.py::shplonk_*).
//
// Queries are grouped by ROTATION SET (the set of points a polynomial is opened at).  With challenges y, v (begin) and
// u (finish), points T = union of all sets, Z_S(X) = prod_{s in S} (X - s):
//   C_k(X)  = sum_j y^j P_kj(X)                            the set's polynomials folded            (kept for finish)
//   R_k(X)  = the interpolation of sum_j y^j evals_kj over S_k (degree < |S_k|)
//   h(X)    = sum_k v^k (C_k(X) - R_k(X)) / Z_{S_k}(X)                                   -> first commitment
//   L(X)    = sum_k v^k z_k (C_k(X) - R_k(u)) - Z_T(u) h(X),   z_k = Z_{T \ S_k}(u)      (L(u) = 0)
//   h'(X)   = L(X) / (X - u) / z_0                                                        -> second commitment
// Every n-coefficient pass is a device kernel; the handful of scalars (interpolation, z_k, R_k(u)) are computed by a
// single lane on the device too -- no field arithmetic runs on the host.
#include <vector>

#include "fp29.cuh"
#include <memory>

#include "pz_internal.h"

#define SH_MAX_PTS 8u     // points per rotation set
#define SH_MAX_SETS 16u
#define SH_MAX_T 32u      // distinct points overall

struct pz_shplonk {
    size_t n = 0;
    unsigned n_sets = 0, n_t = 0;
    std::vector<unsigned> set_np, set_npt, pt_idx;   // polys per set, points per set, flattened point indices
    void* d_C = nullptr;        // n_sets x n: the folded polynomials C_k          (context workspace slots WS_SH_C /
    void* d_small = nullptr;    // device scalars: T points | y | R_k | evals ...   WS_SH_SMALL: one opening at a time per context)
    size_t off_T = 0, off_R = 0, off_out = 0;
    uint64_t v[4];
};

// C[i] = sum_j ypow[j] * P_j[i] over the polynomials of one set (addresses in plist).  A set holds thousands of polynomials
// (every advice column of the proof): grid.y cuts them into chunks of SH_FOLD_CHUNK so the dependent load -> product -> add
// chain of a lane is 32 long, not 6000 (the loop is latency-bound on the loads); k_sh_fold_sum adds the chunks' partials.
#define SH_FOLD_CHUNK 32u
// On the 29-bit field: the polynomial value is unpacked as v * 2^5 (f29_load_shl5) so that v * ypow stays in the ABI's domain, and
// four terms share one Montgomery reduction (f29_dot4).  ypow[0] = 1 is multiplied like any other power (exact).
__global__ __launch_bounds__(256) void k_sh_fold(const u64* __restrict__ plist, unsigned np, const Fr* __restrict__ ypow, size_t n,
                                                 Fr* __restrict__ part) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned j0 = blockIdx.y * SH_FOLD_CHUNK, j1 = j0 + SH_FOLD_CHUNK < np ? j0 + SH_FOLD_CHUNK : np;
    F29<FrTag> acc = f29_zero<FrTag>();
    for (unsigned j = j0; j < j1; j += 4) {
        F29<FrTag> v[4], y[4];
#pragma unroll
        for (unsigned k = 0; k < 4; ++k) {
            const bool in = j + k < j1;
            const unsigned jj = in ? j + k : j;
            v[k] = in ? f29_load_shl5<FrTag>(reinterpret_cast<const Fr*>(plist[jj]) + i) : f29_zero<FrTag>();
            y[k] = f29_load<FrTag>(ypow + jj);
        }
        acc = f29_carry(f29_add(acc, f29_dot4(v, y)));   // each term < 1.8p, tight; at most SH_FOLD_CHUNK / 4 = 8 of them
    }
    f29_store<3>(part + (size_t)blockIdx.y * n + i, acc);
}
__global__ __launch_bounds__(256) void k_sh_fold_sum(const Fr* __restrict__ part, unsigned n_chunks, size_t n, Fr* __restrict__ C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr acc = fp_load<FrTag>(part + i);
    for (unsigned c = 1; c < n_chunks; ++c) acc = fp_add(acc, fp_load<FrTag>(part + (size_t)c * n + i));
    fp_store(C + i, acc);
}
// e_t = sum_j y^j evals[j][t]: one workgroup per point of the set (a set holds thousands of polynomials)
__global__ __launch_bounds__(256) void k_sh_fold_evals(const Fr* __restrict__ evals, unsigned np, unsigned npt, const Fr* __restrict__ ypow,
                                                       Fr* __restrict__ e_out) {
    __shared__ Fr s_acc[256];
    const unsigned t = blockIdx.x;
    Fr acc = fp_zero<FrTag>();
    for (unsigned j = threadIdx.x; j < np; j += 256) acc = fp_add(acc, fp_mul(fp_load<FrTag>(evals + (size_t)j * npt + t), fp_load<FrTag>(ypow + j)));
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) s_acc[threadIdx.x] = fp_add(s_acc[threadIdx.x], s_acc[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) fp_store(e_out + t, s_acc[0]);
}
// R_k = interpolation of the folded evaluations e_t over the set's points (coefficients, ascending), ALL sets in one launch:
// workgroup = set, lane t = the basis polynomial of point t (its own inversion: a Fermat chain of ~380 dependent products is the
// whole cost, so the npt chains of a set and the sets themselves run side by side instead of one lane walking them in turn --
// 0.94 ms per set before, one chain's time for all sets now)
__global__ __launch_bounds__(64) void k_sh_interpolate(const Fr* __restrict__ T, const u32* __restrict__ idx_all, const u32* __restrict__ set_off,
                                                       const Fr* __restrict__ e_all, Fr* __restrict__ R_all) {
    __shared__ Fr s_r[SH_MAX_PTS][SH_MAX_PTS];
    const unsigned k = blockIdx.x, lo = set_off[k], npt = set_off[k + 1] - lo, t = threadIdx.x;
    const u32* idx = idx_all + lo;
    if (t < npt) {
        const Fr xt = fp_load<FrTag>(T + idx[t]);
        // basis polynomial prod_{s != t} (X - x_s) / (x_t - x_s), coefficients in b[]
        Fr b[SH_MAX_PTS];
        b[0] = fp_one<FrTag>();
        unsigned deg = 0;
        Fr den = fp_one<FrTag>();
        for (unsigned s2 = 0; s2 < npt; ++s2) {
            if (s2 == t) continue;
            const Fr xs = fp_load<FrTag>(T + idx[s2]);
            // b *= (X - x_s)
            b[deg + 1] = b[deg];
            for (unsigned q = deg; q > 0; --q) b[q] = fp_sub(b[q - 1], fp_mul(b[q], xs));
            b[0] = fp_neg(fp_mul(b[0], xs));
            ++deg;
            den = fp_mul(den, fp_sub(xt, xs));
        }
        const Fr c = fp_mul(fp_load<FrTag>(e_all + (size_t)k * SH_MAX_PTS + t), fp_inv(den));
        for (unsigned q = 0; q < npt; ++q) s_r[t][q] = fp_mul(b[q], c);   // deg == npt - 1
    }
    __syncthreads();
    if (t < npt) {
        Fr r = s_r[0][t];
        for (unsigned t2 = 1; t2 < npt; ++t2) r = fp_add(r, s_r[t2][t]);
        fp_store(R_all + (size_t)k * SH_MAX_PTS + t, r);
    }
}
// N[i] = C[i] - (i < npt ? R[i] : 0)
__global__ __launch_bounds__(256) void k_sh_numerator(const Fr* __restrict__ C, const Fr* __restrict__ R, unsigned npt, size_t n,
                                                      Fr* __restrict__ N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr c = fp_load<FrTag>(C + i);
    if (i < npt) c = fp_sub(c, fp_load<FrTag>(R + i));
    fp_store(N + i, c);
}
// h[i] = h[i] * v + q[i]
__global__ __launch_bounds__(256) void k_sh_horner(Fr* __restrict__ h, const Fr* __restrict__ q, Fr v, size_t n, int first) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr a = fp_load<FrTag>(q + i);
    if (!first) a = fp_add(fp_mul(fp_load<FrTag>(h + i), v), a);
    fp_store(h + i, a);
}
// one lane: out[0 .. n_sets) = v^k z_k, out[n_sets] = sum_k v^k z_k R_k(u), out[n_sets + 1] = Z_T(u), out[n_sets + 2] = 1 / z_0
__global__ void k_sh_finish_scalars(const Fr* __restrict__ T, unsigned n_t, const u32* __restrict__ idx, const u32* __restrict__ set_off,
                                    unsigned n_sets, const Fr* __restrict__ R, Fr u, Fr v, Fr* __restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    Fr zt = fp_one<FrTag>();
    for (unsigned t = 0; t < n_t; ++t) zt = fp_mul(zt, fp_sub(u, fp_load<FrTag>(T + t)));
    Fr vk = fp_one<FrTag>(), cterm = fp_zero<FrTag>(), z0 = fp_one<FrTag>();
    for (unsigned k = 0; k < n_sets; ++k) {
        const unsigned lo = set_off[k], hi = set_off[k + 1];
        Fr zk = fp_one<FrTag>();
        for (unsigned t = 0; t < n_t; ++t) {
            bool in = false;
            for (unsigned q = lo; q < hi; ++q) in = in || idx[q] == t;
            if (!in) zk = fp_mul(zk, fp_sub(u, fp_load<FrTag>(T + t)));
        }
        if (k == 0) z0 = zk;
        // R_k(u) by Horner over its hi - lo coefficients
        Fr ru = fp_zero<FrTag>();
        for (unsigned q = hi; q-- > lo;) ru = fp_add(fp_mul(ru, u), fp_load<FrTag>(R + (size_t)k * SH_MAX_PTS + (q - lo)));
        const Fr ck = fp_mul(vk, zk);
        fp_store(out + k, ck);
        cterm = fp_add(cterm, fp_mul(ck, ru));
        vk = fp_mul(vk, v);
    }
    fp_store(out + n_sets, cterm);
    fp_store(out + n_sets + 1, zt);
    fp_store(out + n_sets + 2, fp_inv(z0));
}
// L[i] = sum_k coef_k C_k[i] - zt h[i]  (- cterm at i = 0)
__global__ __launch_bounds__(256) void k_sh_linearise(const Fr* __restrict__ C, unsigned n_sets, size_t n, const Fr* __restrict__ sc,
                                                      const Fr* __restrict__ h, Fr* __restrict__ L) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr acc = fp_neg(fp_mul(fp_load<FrTag>(h + i), fp_load<FrTag>(sc + n_sets + 1)));
    for (unsigned k = 0; k < n_sets; ++k) acc = fp_add(acc, fp_mul(fp_load<FrTag>(C + (size_t)k * n + i), fp_load<FrTag>(sc + k)));
    if (i == 0) acc = fp_sub(acc, fp_load<FrTag>(sc + n_sets));
    fp_store(L + i, acc);
}
__global__ __launch_bounds__(256) void k_sh_scale(Fr* __restrict__ a, size_t n, const Fr* __restrict__ s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fp_store(a + i, fp_mul(fp_load<FrTag>(a + i), fp_load<FrTag>(s)));
}

static Fr fr_host(const uint64_t x[4]) {
    Fr r;
    memcpy(r.v, x, 32);
    return r;
}

extern "C" int pz_shplonk_free(pz_ctx* ctx, pz_shplonk* st) {
    (void)ctx;
    if (!st) return PZ_OK;
    delete st;   // the buffers are the context's grow-only workspace slots: nothing to free, no device synchronisation
    return PZ_OK;
}

extern "C" int pz_shplonk_begin_dev(pz_ctx* ctx, size_t n, uint32_t n_sets, const uint32_t* set_n_polys,
                                    const uint64_t* const* d_polys, const uint32_t* set_n_points, const uint32_t* point_idx,
                                    uint32_t n_points_total, const uint64_t* points, const uint64_t* evals, const uint64_t y[4],
                                    const uint64_t v[4], uint64_t* d_h, pz_shplonk** state) {
    if (!ctx || !n || !n_sets || n_sets > SH_MAX_SETS || !set_n_polys || !d_polys || !set_n_points || !point_idx || !points ||
        !evals || !y || !v || !d_h || !state || !n_points_total || n_points_total > SH_MAX_T)
        return PZ_ERR_INVALID;
    *state = nullptr;
    size_t tot_polys = 0, tot_pts = 0, tot_evals = 0, max_np = 0;
    for (unsigned k = 0; k < n_sets; ++k) {
        if (!set_n_polys[k] || !set_n_points[k] || set_n_points[k] > SH_MAX_PTS || set_n_points[k] >= n) return PZ_ERR_INVALID;
        for (unsigned q = 0; q < set_n_points[k]; ++q)
            if (point_idx[tot_pts + q] >= n_points_total) return PZ_ERR_INVALID;
        tot_polys += set_n_polys[k];
        tot_pts += set_n_points[k];
        tot_evals += (size_t)set_n_polys[k] * set_n_points[k];
        if (set_n_polys[k] > max_np) max_np = set_n_polys[k];
    }
    for (size_t j = 0; j < tot_polys; ++j)
        if (!d_polys[j]) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    std::unique_ptr<pz_shplonk> owner(new pz_shplonk());   // released into *state only on success: no early return leaks it
    pz_shplonk* st = owner.get();
    st->n = n;
    st->n_sets = n_sets;
    st->n_t = n_points_total;
    st->set_np.assign(set_n_polys, set_n_polys + n_sets);
    st->set_npt.assign(set_n_points, set_n_points + n_sets);
    st->pt_idx.assign(point_idx, point_idx + tot_pts);
    memcpy(st->v, v, 32);
    // small device block: [T points][ypow max_np][R: n_sets x SH_MAX_PTS][finish scalars: n_sets + 3][evals][poly pointers]
    // [point idx][set offsets]
    const size_t o_T = 0, o_y = o_T + n_points_total, o_R = o_y + max_np, o_out = o_R + (size_t)n_sets * SH_MAX_PTS,
                 o_e = o_out + n_sets + 3, o_ev = o_e + (size_t)n_sets * SH_MAX_PTS, fr_end = o_ev + tot_evals;
    const size_t b_ptr = fr_end * 32, b_idx = b_ptr + tot_polys * 8, b_off = b_idx + tot_pts * 4, b_end = b_off + (n_sets + 1) * 4;
    {
        int rc = pz_ws_get(ctx, WS_SH_SMALL, b_end + 64, &st->d_small);
        if (rc == PZ_OK) rc = pz_ws_get(ctx, WS_SH_C, (size_t)n_sets * n * 32, &st->d_C);
        if (rc != PZ_OK) return rc;
    }
    st->off_T = o_T; st->off_R = o_R; st->off_out = o_out;
    char* sm = (char*)st->d_small;
    Fr* fsm = (Fr*)sm;
    std::vector<uint32_t> offs(n_sets + 1, 0);
    for (unsigned k = 0; k < n_sets; ++k) offs[k + 1] = offs[k] + set_n_points[k];
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(fsm + o_T, points, (size_t)n_points_total * 32, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(fsm + o_ev, evals, tot_evals * 32, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(sm + b_ptr, d_polys, tot_polys * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(sm + b_idx, point_idx, tot_pts * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(sm + b_off, offs.data(), (n_sets + 1) * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipStreamSynchronize(s));   // the host arrays may go away after the call
    {   // y^j, j < max_np: the cached power-table kernel (parallel) rather than one lane walking thousands of products
        void* yp;
        PZCHK(pz_get_pow_table(ctx, y, max_np, &yp));
        HIPCHK(ctx, hipMemcpyAsync(fsm + o_y, yp, max_np * 32, hipMemcpyDeviceToDevice, s));
    }
    void* wsN;
    PZCHK(pz_ws_get(ctx, WS_NTT_TMP, n * 32, &wsN));   // (pz_poly_div_linear_dev owns WS_BIG_A)
    Fr* N = (Fr*)wsN;
    const unsigned gb = pz_div_up(n, 256);
    // sets from the last to the first: h = h * v + Q_k ends as sum_k v^k Q_k
    size_t p_off[SH_MAX_SETS + 1], e_off[SH_MAX_SETS + 1];
    p_off[0] = e_off[0] = 0;
    for (unsigned k = 0; k < n_sets; ++k) {
        p_off[k + 1] = p_off[k] + set_n_polys[k];
        e_off[k + 1] = e_off[k] + (size_t)set_n_polys[k] * set_n_points[k];
    }
    // the folded evaluations of every set, then every set's interpolation polynomial, before the per-set passes over the polynomials
    for (unsigned kk = 0; kk < n_sets; ++kk)
        hipLaunchKernelGGL(k_sh_fold_evals, dim3(set_n_points[kk]), dim3(256), 0, s, fsm + o_ev + e_off[kk], set_n_polys[kk], set_n_points[kk],
                           fsm + o_y, fsm + o_e + (size_t)kk * SH_MAX_PTS);
    hipLaunchKernelGGL(k_sh_interpolate, dim3(n_sets), dim3(64), 0, s, fsm + o_T, (const u32*)(sm + b_idx), (const u32*)(sm + b_off), fsm + o_e,
                       fsm + o_R);
    for (unsigned kk = n_sets; kk-- > 0;) {
        Fr* Ck = (Fr*)st->d_C + (size_t)kk * n;
        {
            const unsigned nch = pz_div_up(set_n_polys[kk], SH_FOLD_CHUNK);
            void* part;
            PZCHK(pz_ws_get(ctx, WS_BIG_C, (size_t)nch * n * 32, &part));
            hipLaunchKernelGGL(k_sh_fold, dim3(gb, nch), dim3(256), 0, s, (const u64*)(sm + b_ptr) + p_off[kk], set_n_polys[kk], fsm + o_y, n,
                               (Fr*)part);
            hipLaunchKernelGGL(k_sh_fold_sum, dim3(gb), dim3(256), 0, s, (const Fr*)part, nch, n, Ck);
        }
        hipLaunchKernelGGL(k_sh_numerator, dim3(gb), dim3(256), 0, s, Ck, fsm + o_R + (size_t)kk * SH_MAX_PTS, set_n_points[kk], n, N);
        HIPCHK(ctx, hipGetLastError());
        for (unsigned q = 0; q < set_n_points[kk]; ++q)   // exact division by (X - s) for every point of the set
            PZCHK(pz_poly_div_linear_dev(ctx, (const uint64_t*)N, 1, 4 * n, n, points + 4 * (size_t)point_idx[offs[kk] + q], (uint64_t*)N, 4 * n));
        hipLaunchKernelGGL(k_sh_horner, dim3(gb), dim3(256), 0, s, (Fr*)d_h, N, fr_host(v), n, kk == n_sets - 1 ? 1 : 0);
    }
    HIPCHK(ctx, hipGetLastError());
    *state = owner.release();
    return PZ_OK;
}

extern "C" int pz_shplonk_finish_dev(pz_ctx* ctx, pz_shplonk* st_in, const uint64_t u[4], const uint64_t* d_h, uint64_t* d_h2) {
    if (!ctx || !st_in) return PZ_ERR_INVALID;
    std::unique_ptr<pz_shplonk> owner(st_in);   // pz.h: finish frees the state -- on every path, errors included
    pz_shplonk* st = owner.get();
    if (!u || !d_h || !d_h2) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    const size_t n = st->n;
    hipStream_t s = ctx->stream;
    char* sm = (char*)st->d_small;
    Fr* fsm = (Fr*)sm;
    // byte offsets as laid out by begin
    size_t tot_polys = 0, tot_pts = 0, tot_evals = 0, max_np = 0;
    for (unsigned k = 0; k < st->n_sets; ++k) {
        tot_polys += st->set_np[k];
        tot_pts += st->set_npt[k];
        tot_evals += (size_t)st->set_np[k] * st->set_npt[k];
        if (st->set_np[k] > max_np) max_np = st->set_np[k];
    }
    const size_t o_ev = st->off_out + st->n_sets + 3 + (size_t)st->n_sets * SH_MAX_PTS, fr_end = o_ev + tot_evals;
    const size_t b_ptr = fr_end * 32, b_idx = b_ptr + tot_polys * 8, b_off = b_idx + tot_pts * 4;
    hipLaunchKernelGGL(k_sh_finish_scalars, dim3(1), dim3(64), 0, s, fsm + st->off_T, st->n_t, (const u32*)(sm + b_idx),
                       (const u32*)(sm + b_off), st->n_sets, fsm + st->off_R, fr_host(u), fr_host(st->v), fsm + st->off_out);
    const unsigned gb = pz_div_up(n, 256);
    hipLaunchKernelGGL(k_sh_linearise, dim3(gb), dim3(256), 0, s, (const Fr*)st->d_C, st->n_sets, n, fsm + st->off_out, (const Fr*)d_h,
                       (Fr*)d_h2);
    HIPCHK(ctx, hipGetLastError());
    PZCHK(pz_poly_div_linear_dev(ctx, d_h2, 1, 4 * n, n, u, d_h2, 4 * n));
    hipLaunchKernelGGL(k_sh_scale, dim3(gb), dim3(256), 0, s, (Fr*)d_h2, n, fsm + st->off_out + st->n_sets + 2);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;   // `owner` frees the state
}
