// pz_quotient.hip -- SURVEY.md section 8f rank 1 and 3: the prover steps that sit between K1/K2 and are the
// Amdahl remainder once commitments and NTTs run on the GPU.  In the reference they are reached only through
// bench.rs:161-171 (create_proof inside bench_builder); the formulas are the published halo2 protocol as
// halo2-axiom implements it (dependency behaviour, SURVEY tag [D]: results are pinned by the mathematics --
// recurrences, divisibility by X^n - 1, p(X) - p(x) = (X - x) q(X) -- not by reference fixtures).
//
//   pz_fr_batch_invert_dev       halo2 BatchInvert: a[i] <- 1/a[i], zeros stay zero (Montgomery's trick, 3 products
//                                per element + one inversion per K-element strided run)
//   pz_fr_prefix_product_dev     z[0] = z0, z[i+1] = z[i] * a[i]: the running product of both grand-product arguments
//   pz_permutation_product_dev   permutation::Argument::commit for one chunk of columns:
//                                z[i+1] = z[i] * prod_j (v_j[i] + beta*delta^j*w^i + gamma) / (v_j[i] + beta*sigma_j[i] + gamma)
//   pz_quotient_gate_dev         evaluate_h for halo2-lib's vertical gate q*(a0 + a1*a2 - a3) on the extended coset,
//                                Horner-folded in the challenge y over the advice columns
//   pz_quotient_finish_dev       division by the vanishing polynomial on the coset: only 2^e distinct values of x^n - 1
//   pz_fr_distribute_powers_dev  a[i] *= c * g^i (the coset un-scaling of extended_to_coeff)
//   pz_poly_div_linear_dev       kate_division: q(X) = (p(X) - p(x)) / (X - x), blocked Horner recurrence
//
// All of it is elementwise / scan work on 32-byte field elements: HBM-bound at >= 8 columns per pass, otherwise
// bound by the field products (3-5 per element).
#include "fp29.cuh"
#include "pz_internal.h"

static inline Fr fp_one_host() {  // Montgomery one of Fr (R mod r)
    static const uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    Fr r;
    memcpy(r.v, ONE, 32);
    return r;
}
static inline Fr fr_from_u64(const uint64_t x[4]) {
    Fr r;
    memcpy(r.v, x, 32);
    return r;
}

// ---- the 29-bit field for the row kernels (fp29.cuh)
typedef F29<FrTag> Fr29;
// a challenge (x * 2^256, canonical) as x * 2^261: the operand that keeps a product with a 256-domain value in the 256-domain
__device__ __forceinline__ Fr29 fr29_c261(const Fr& c) { return f29_to_261(f29_from_fp(c)); }
// 1 in the 256-domain
__device__ __forceinline__ Fr29 fr29_one256() {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = P29<FrTag>::R256(i);
    return r;
}

// ---- wave-uniform constants in SGPRs.  A challenge is the same for every lane; converted on the HOST into the nine 29-bit limbs
// the kernels multiply by (c * 2^(256 + k) mod r: k = 5 for the 261-domain image of a 256-domain value, k = 10 for beta, whose
// product with a plain 256-domain value must land in the 261-domain) and passed BY VALUE as a kernel argument, it lives in SGPRs:
// f29_mul_s / f29_mul2_s read it as the scalar operand of every mad, and the nine VGPRs per constant a per-lane copy took are
// free -- k_quotient_permutation carried five such copies (45 of its 158 VGPRs: the difference between 3 and 4 waves per SIMD).
struct S29 {
    u32 v[9];
};
static bool host_fr_canonical(const uint64_t c[4]) {   // c < r: what host_fr_shl (and every kernel that takes its limbs) assumes
    static const uint64_t R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    for (int i = 3; i >= 0; --i)
        if (c[i] != R[i]) return c[i] < R[i];
    return false;
}
static S29 host_fr_shl(const uint64_t c[4], unsigned k) {   // canonical c (Montgomery form, below r) -> limbs of c * 2^k mod r
    static const uint64_t R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    uint64_t x[4] = {c[0], c[1], c[2], c[3]};
    for (unsigned t = 0; t < k; ++t) {
        for (int i = 3; i > 0; --i) x[i] = (x[i] << 1) | (x[i - 1] >> 63);   // x < r < 2^254: no bit is lost
        x[0] <<= 1;
        bool ge = true;
        for (int i = 3; i >= 0; --i)
            if (x[i] != R[i]) { ge = x[i] > R[i]; break; }
        if (ge) {
            unsigned __int128 br = 0;
            for (int i = 0; i < 4; ++i) {
                const unsigned __int128 d = (unsigned __int128)x[i] - R[i] - br;
                x[i] = (uint64_t)d;
                br = (d >> 64) & 1;
            }
        }
    }
    S29 r;
    for (int i = 0; i < 9; ++i) {
        const unsigned bit = 29u * i, j = bit >> 6, o = bit & 63;
        uint64_t w = x[j] >> o;
        if (o > 35 && j + 1 < 4) w |= x[j + 1] << (64 - o);
        r.v[i] = (u32)(w & (i < 8 ? 0x1fffffffull : 0xffffffffull));
    }
    return r;
}
// host Montgomery product of two canonical Fr elements (4 x u64, R = 2^256): a handful of calls per entry point (challenge powers)
static void host_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    static const uint64_t R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    const uint64_t INV = 0xc2e1f593efffffffULL;   // -r^-1 mod 2^64
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (unsigned __int128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * INV;
        c = ((unsigned __int128)m * R[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) {
            c += (unsigned __int128)m * R[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    bool ge = t[4] != 0;
    if (!ge) {
        ge = true;
        for (int i = 3; i >= 0; --i)
            if (t[i] != R[i]) { ge = t[i] > R[i]; break; }
    }
    if (ge) {
        unsigned __int128 br = 0;
        for (int i = 0; i < 4; ++i) {
            const unsigned __int128 d = (unsigned __int128)t[i] - R[i] - br;
            t[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    memcpy(out, t, 32);
}
static void host_fr_pow(const uint64_t a[4], unsigned e, uint64_t out[4]) {   // a^e, Montgomery form in and out
    static const uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    uint64_t acc[4], sq[4];
    memcpy(acc, ONE, 32);
    memcpy(sq, a, 32);
    for (; e; e >>= 1) {
        if (e & 1) host_fr_mul(acc, sq, acc);
        host_fr_mul(sq, sq, sq);
    }
    memcpy(out, acc, 32);
}
// CPU-only consistency check of the three host helpers above (no device call; outside the pz_ ABI: tests/test_host_logic.py calls it
// through ctypes): powers of two computed by doubling (host_fr_shl), by Montgomery products (host_fr_mul) and by square-and-
// multiply (host_fr_pow) must coincide, limb for limb.  Returns 0, or the number of the first check that failed.
extern "C" int pzx_host_selftest(void) {
    static const uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    auto limbs_of = [](const uint64_t x[4]) { return host_fr_shl(x, 0); };
    auto same = [](const S29& a, const S29& b) { return memcmp(a.v, b.v, sizeof a.v) == 0; };
    uint64_t two[4], acc[4], pw[4];
    {   // 2 in Montgomery form = ONE + ONE mod r, built from the limbs of ONE doubled once
        const S29 d = host_fr_shl(ONE, 1);
        unsigned __int128 carry = 0;
        memset(two, 0, sizeof two);
        for (int i = 0; i < 9; ++i) {   // limbs -> 4 x u64
            const unsigned bit = 29u * i, j = bit >> 6, o = bit & 63;
            two[j] |= (uint64_t)d.v[i] << o;
            if (o > 35 && j + 1 < 4) two[j + 1] |= (uint64_t)d.v[i] >> (64 - o);
        }
        (void)carry;
    }
    if (!same(limbs_of(two), host_fr_shl(ONE, 1))) return 1;
    host_fr_mul(ONE, ONE, acc);
    if (memcmp(acc, ONE, 32)) return 2;                       // 1 * 1 = 1
    memcpy(acc, ONE, 32);
    for (unsigned k = 1; k <= 300; ++k) {                     // 2^k three ways (k beyond 254: the reductions are exercised)
        host_fr_mul(acc, two, acc);
        if (!same(limbs_of(acc), host_fr_shl(ONE, k))) return 100 + (int)k;
        if (k % 37 == 0 || k == 300) {
            host_fr_pow(two, k, pw);
            if (memcmp(pw, acc, 32)) return 1000 + (int)k;
        }
    }
    return 0;
}
// the helpers themselves for the CPU tests (compared there with Python integers)
extern "C" void pzx_host_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { host_fr_mul(a, b, out); }
extern "C" void pzx_host_fr_pow(const uint64_t a[4], unsigned e, uint64_t out[4]) { host_fr_pow(a, e, out); }
extern "C" void pzx_host_fr_shl(const uint64_t c[4], unsigned k, uint32_t out[9]) {
    const S29 r = host_fr_shl(c, k);
    memcpy(out, r.v, sizeof r.v);
}
__device__ __forceinline__ Fr29 f29_add_s(const Fr29& a, const u32 (&b)[9]) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b[i];
    return r;
}

// ---------------------------------------------------------------------------------------------- batch inversion
// thread t owns elements t, t+T, t+2T, ... (coalesced across the wave for every j)
// 29-bit field: the running products are kept in the 256-domain (acc * shl5(v)); the scratch array takes them packed but not
// canonical (a product is tight and below 2p < 2^256: no conditional subtraction on the way out).  f29_inv works in the
// 261-domain: fed acc * 2^256 it returns acc^-1 * 2^266, and one product by 1 * 2^256 brings that to acc^-1 * 2^261 -- the domain
// in which inv * scratch lands in the 256-domain and inv * shl5(v) stays where it is.
// mul_io != nullptr: the inverses are not stored; mul_io[i] <- mul_io[i] / a[i] instead (mul_io[i] <- 0 where a[i] = 0: the inverse of zero is zero), a stays as
// it was -- the numerator / denominator quotient of the grand-product arguments without a pass of its own
__global__ __launch_bounds__(256) void k_batch_invert(Fr* __restrict__ a, Fr* __restrict__ scratch, size_t n, size_t T,
                                                      unsigned K, Fr* __restrict__ mul_io) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    Fr29 acc = fr29_one256();
    for (unsigned j = 0; j < K; ++j) {
        const size_t i = t + (size_t)j * T;
        if (i >= n) break;
        const Fr v = fp_load<FrTag>(a + i);
        if (fp_is_zero(v)) continue;
        Fr w;
        f29_pack(acc, w.v);
        fp_store(scratch + i, w);
        acc = f29_mul(acc, f29_from_fp_shl5(v));
    }
    Fr29 inv = f29_mul(f29_inv(acc), fr29_one256());  // acc is a product of non-zero elements (or 1)
    for (unsigned j = K; j-- > 0;) {
        const size_t i = t + (size_t)j * T;
        if (i >= n) continue;
        const Fr v = fp_load<FrTag>(a + i);
        if (fp_is_zero(v)) {
            // a zero denominator inverts to zero (ff::BatchInvert skips it and leaves 0), so the quotient the fused path forms is 0
            // -- what batch_invert followed by the multiplication gave before the two were fused, and what the reference computes
            if (mul_io) fp_store(mul_io + i, fp_zero<FrTag>());
            continue;
        }
        const Fr29 ai = f29_mul(inv, f29_load<FrTag>(scratch + i));   // 1 / a[i], 256-domain
        if (mul_io) f29_store_product(mul_io + i, f29_mul(ai, f29_load_shl5<FrTag>(mul_io + i)));
        else f29_store_product(a + i, ai);
        inv = f29_mul(inv, f29_from_fp_shl5(v));
    }
}

int pz_batch_invert_internal(pz_ctx* ctx, Fr* d_a, size_t n, Fr* d_mul_io) {
    if (!n) return PZ_OK;
    void* scr;
    PZCHK(pz_ws_get(ctx, WS_BIG_A, n * 32, &scr));
    // one inversion (~380 products, every lane of the wave its own) per K elements against 3 products per element: K = 64 once
    // there are 2^14 threads, up to 512 while 2^18 threads (four waves per SIMD) remain
    unsigned K = 8;
    while (K < 64 && n / (K * 2) >= 16384) K *= 2;
    while (K < 512 && n / (K * 2) >= 262144) K *= 2;
    const size_t T = pz_div_up(n, K);
    hipLaunchKernelGGL(k_batch_invert, dim3(pz_div_up(T, 256)), dim3(256), 0, ctx->stream, d_a, (Fr*)scr, n, T, K, d_mul_io);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

extern "C" int pz_fr_batch_invert_dev(pz_ctx* ctx, uint64_t* d_a, size_t n) {
    if (!ctx || (n && !d_a)) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    return pz_batch_invert_internal(ctx, (Fr*)d_a, n, nullptr);
}

// ---------------------------------------------------------------------------------------------- prefix product
#define PP_K 16u
// (batched over grid.y = column.)  phase 1: per-thread run products, block-level exclusive scan;
// excl[t] = product of the block's earlier runs
// (29-bit field: the run product stays in the 256-domain through shl5-unpacked factors; the workgroup scan runs in the
// 261-domain, where products of two scanned values are closed, on raw limbs in LDS; `excl` keeps that domain -- packed, not
// canonical -- so that block prefix (256-domain) x excl lands in the 256-domain in k_pp_apply)
__global__ __launch_bounds__(256) void k_pp_local(const Fr* __restrict__ a, size_t as, size_t n, Fr* __restrict__ excl,
                                                  Fr* __restrict__ block_tot, unsigned nb) {
    __shared__ u32 s[9 * 256];   // limb-major: s[limb * 256 + thread]
    a += (size_t)blockIdx.y * as;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t lo = t * PP_K;
    Fr29 p = fr29_one256();
    for (unsigned j = 0; j < PP_K; ++j)
        if (lo + j < n) p = f29_mul(p, f29_load_shl5<FrTag>(a + lo + j));
    p = f29_to_261(p);
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i * 256 + threadIdx.x] = p.v[i];
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
        Fr29 u;
        if (threadIdx.x >= off) {
#pragma unroll
            for (int i = 0; i < 9; ++i) u.v[i] = s[i * 256 + threadIdx.x - off];
        }
        __syncthreads();
        if (threadIdx.x >= off) {
            p = f29_mul(u, p);
#pragma unroll
            for (int i = 0; i < 9; ++i) s[i * 256 + threadIdx.x] = p.v[i];
        }
        __syncthreads();
    }
    Fr29 e = f29_one<FrTag>();
    if (threadIdx.x) {
#pragma unroll
        for (int i = 0; i < 9; ++i) e.v[i] = s[i * 256 + threadIdx.x - 1];
    }
    Fr w;
    f29_pack(e, w.v);   // strict limbs, below 2p
    fp_store(excl + (size_t)blockIdx.y * nb * 256 + t, w);
    if (threadIdx.x == 255) f29_store_product(block_tot + (size_t)blockIdx.y * nb + blockIdx.x, f29_to_256(p));
}
// phase 2: exclusive scan of each column's block totals, seeded with z0 (a few dozen entries: one lane per column)
__global__ void k_pp_blocks(Fr* __restrict__ block_tot, unsigned nb, size_t n_cols, Fr z0) {
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    Fr* bt = block_tot + col * nb;
    Fr run = z0;
    for (unsigned b = 0; b < nb; ++b) {
        Fr t = fp_load<FrTag>(bt + b);
        fp_store(bt + b, run);
        run = fp_mul(run, t);
    }
}
// phase 3: z[i] = block prefix * thread prefix * run prefix
__global__ __launch_bounds__(256) void k_pp_apply(const Fr* __restrict__ a, size_t as, size_t n, const Fr* __restrict__ excl,
                                                  const Fr* __restrict__ block_tot, unsigned nb, Fr* __restrict__ z, size_t zs) {
    a += (size_t)blockIdx.y * as;
    z += (size_t)blockIdx.y * zs;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t lo = t * PP_K;
    if (lo >= n) return;
    Fr29 cur = f29_mul(f29_load<FrTag>(block_tot + (size_t)blockIdx.y * nb + blockIdx.x),
                       f29_load<FrTag>(excl + (size_t)blockIdx.y * nb * 256 + t));
    for (unsigned j = 0; j < PP_K && lo + j < n; ++j) {
        const Fr v = fp_load<FrTag>(a + lo + j);  // read before the store: z may alias a
        f29_store_product(z + lo + j, cur);
        cur = f29_mul(cur, f29_from_fp_shl5(v));
    }
}

int pz_prefix_product_batch_internal(pz_ctx* ctx, const Fr* d_a, size_t a_stride, size_t n_cols, size_t n, Fr z0, Fr* d_z,
                                     size_t z_stride) {
    if (!n || !n_cols) return PZ_OK;
    if (n_cols > 65535) return PZ_ERR_INVALID;
    const unsigned nb = pz_div_up(pz_div_up(n, PP_K), 256);
    void* ws;
    PZCHK(pz_ws_get(ctx, WS_BIG_B, n_cols * ((size_t)nb * 256 + nb) * 32, &ws));
    Fr* excl = (Fr*)ws;
    Fr* tot = excl + n_cols * (size_t)nb * 256;
    hipLaunchKernelGGL(k_pp_local, dim3(nb, (unsigned)n_cols), dim3(256), 0, ctx->stream, d_a, a_stride, n, excl, tot, nb);
    hipLaunchKernelGGL(k_pp_blocks, dim3(pz_div_up(n_cols, 64)), dim3(64), 0, ctx->stream, tot, nb, n_cols, z0);
    hipLaunchKernelGGL(k_pp_apply, dim3(nb, (unsigned)n_cols), dim3(256), 0, ctx->stream, d_a, a_stride, n, (const Fr*)excl,
                       (const Fr*)tot, nb, d_z, z_stride);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}
int pz_prefix_product_internal(pz_ctx* ctx, const Fr* d_a, size_t n, Fr z0, Fr* d_z) {
    return pz_prefix_product_batch_internal(ctx, d_a, n, 1, n, z0, d_z, n);
}

extern "C" int pz_fr_prefix_product_dev(pz_ctx* ctx, const uint64_t* d_a, size_t n, const uint64_t z0[4], uint64_t* d_z) {
    if (!ctx || !z0 || (n && (!d_a || !d_z))) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    return pz_prefix_product_internal(ctx, (const Fr*)d_a, n, fr_from_u64(z0), (Fr*)d_z);
}

// ---------------------------------------------------------------------------------------------- permutation product
// (29-bit field; domains as in k_quotient_permutation: v, gamma, beta * sigma and beta * delta^j * w^i carry 2^261, the running
// products 2^256)
__global__ __launch_bounds__(256) void k_perm_terms(const Fr* __restrict__ cols, size_t cs, const Fr* __restrict__ sigma,
                                                    size_t ss, unsigned m, size_t n, const Fr* __restrict__ wpow, Fr beta,
                                                    Fr gamma, Fr delta0, Fr delta, Fr* __restrict__ num,
                                                    Fr* __restrict__ den) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr29 g = fr29_c261(gamma), d = fr29_c261(delta), beta266 = f29_to_261(fr29_c261(beta));
    Fr29 nm = fr29_one256(), dn = nm;
    Fr29 bd = f29_mul(f29_mul(beta266, f29_from_fp(delta0)), f29_load_shl5<FrTag>(wpow + i));  // beta * delta^j0 * w^i, times delta per column
    for (unsigned j = 0; j < m; ++j) {
        const Fr29 v = f29_add(f29_load_shl5<FrTag>(cols + (size_t)j * cs + i), g);
        const Fr29 sg = f29_load<FrTag>(sigma + (size_t)j * ss + i);
        nm = f29_mul(nm, f29_add(v, bd));
        dn = f29_mul(dn, f29_add(v, f29_mul(beta266, sg)));
        bd = f29_mul(bd, d);
    }
    f29_store<0>(num + i, nm);   // one256 when m == 0, a product otherwise: strict limbs either way, but keep the general store
    f29_store<0>(den + i, dn);
}

extern "C" int pz_permutation_product_dev(pz_ctx* ctx, const uint64_t* d_cols, size_t col_stride, const uint64_t* d_sigma,
                                          size_t sigma_stride, size_t m, uint32_t log_n, const uint64_t omega[4],
                                          const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta_start[4],
                                          const uint64_t delta[4], const uint64_t z0[4], uint64_t* d_z) {
    if (!ctx || !omega || !beta || !gamma || !delta_start || !delta || !z0 || !d_z || log_n > 26) return PZ_ERR_INVALID;
    if (m && (!d_cols || !d_sigma)) return PZ_ERR_INVALID;
    if (col_stride % 4 || sigma_stride % 4 || m > 65535) return PZ_ERR_INVALID;
    const size_t n = (size_t)1 << log_n;
    if (m > 1 && (col_stride < 4 * n || sigma_stride < 4 * n)) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    void *wp, *ws;
    PZCHK(pz_get_pow_table(ctx, omega, n, &wp));
    PZCHK(pz_ws_get(ctx, WS_BIG_C, 2 * n * 32, &ws));
    Fr* num = (Fr*)ws;
    Fr* den = num + n;
    hipLaunchKernelGGL(k_perm_terms, dim3(pz_div_up(n, 256)), dim3(256), 0, ctx->stream, (const Fr*)d_cols, col_stride / 4,
                       (const Fr*)d_sigma, sigma_stride / 4, (unsigned)m, n, (const Fr*)wp, fr_from_u64(beta), fr_from_u64(gamma),
                       fr_from_u64(delta_start), fr_from_u64(delta), num, den);
    HIPCHK(ctx, hipGetLastError());
    PZCHK(pz_batch_invert_internal(ctx, den, n, num));   // num <- num / den
    return pz_prefix_product_internal(ctx, num, n, fr_from_u64(z0), (Fr*)d_z);
}

// all sets of the permutation argument at once: set j covers columns [j*chunk_len, ...), its product starts where the
// previous set's stood at row `usable_rows` (halo2: z_j[0] = z_{j-1}[u]).  The sets are computed independently from 1
// (one batched inversion, one batched scan) and chained afterwards by a scalar per set.
__global__ __launch_bounds__(256) void k_perm_terms_sets(const Fr* __restrict__ cols, size_t cs, const Fr* __restrict__ sigma,
                                                         size_t ss, unsigned m, unsigned chunk_len, size_t n,
                                                         const Fr* __restrict__ wpow, const Fr* __restrict__ dpow,
                                                         S29 beta266, S29 g, S29 d, Fr* __restrict__ num, Fr* __restrict__ den,
                                                         unsigned set0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned c0 = (set0 + blockIdx.y) * chunk_len, set = blockIdx.y;   // num / den hold the launch's sets only
    Fr29 nm = fr29_one256(), dn = nm;
    Fr29 bd = f29_mul(f29_mul_s(f29_load<FrTag>(dpow + c0), beta266.v), f29_load_shl5<FrTag>(wpow + i));
    // the next column's value and sigma are requested before the current column's products start
    Fr v_n = fp_load<FrTag>(cols + (size_t)c0 * cs + i), s_n = fp_load<FrTag>(sigma + (size_t)c0 * ss + i);
    for (unsigned c = c0; c < c0 + chunk_len && c < m; ++c) {
        const Fr29 v = f29_add_s(f29_from_fp_shl5(v_n), g.v);
        const Fr29 sg = f29_from_fp(s_n);
        const unsigned cn = (c + 1 < c0 + chunk_len && c + 1 < m) ? c + 1 : c;
        v_n = fp_load<FrTag>(cols + (size_t)cn * cs + i);
        s_n = fp_load<FrTag>(sigma + (size_t)cn * ss + i);
        nm = f29_mul(nm, f29_add(v, bd));
        dn = f29_mul(dn, f29_add(v, f29_mul_s(sg, beta266.v)));
        bd = f29_mul_s(bd, d.v);
    }
    f29_store_product(num + (size_t)set * n + i, nm);   // one256 (an empty set) or a product: strict limbs, below 2p
    f29_store_product(den + (size_t)set * n + i, dn);
}
// mult[j] = prod_{t < j} z_t[u]: the exclusive prefix product over the sets that chains them (z_j starts where z_{j-1} stood at row
// u).  One workgroup: thread runs over consecutive sets, a Hillis-Steele scan over the 256 run products (a single lane walking
// 128 dependent products took 145 us per call, 13 calls per proof)
__global__ __launch_bounds__(256) void k_perm_chain(const Fr* __restrict__ z, size_t zs, unsigned n_sets, size_t u, Fr* __restrict__ mult) {
    __shared__ Fr s_p[256];
    const unsigned per = (n_sets + 255u) / 256u, lo = threadIdx.x * per, hi = lo + per < n_sets ? lo + per : n_sets;
    Fr p = fp_one<FrTag>();
    for (unsigned j = lo; j < hi; ++j) p = fp_mul(p, fp_load<FrTag>(z + (size_t)j * zs + u));
    s_p[threadIdx.x] = p;
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {
        Fr v = fp_one<FrTag>();
        if (threadIdx.x >= off) v = s_p[threadIdx.x - off];
        __syncthreads();
        if (threadIdx.x >= off) {
            p = fp_mul(v, p);
            s_p[threadIdx.x] = p;
        }
        __syncthreads();
    }
    Fr c = threadIdx.x ? s_p[threadIdx.x - 1] : fp_one<FrTag>();
    for (unsigned j = lo; j < hi; ++j) {
        fp_store(mult + j, c);
        c = fp_mul(c, fp_load<FrTag>(z + (size_t)j * zs + u));
    }
}
__global__ __launch_bounds__(256) void k_scale_sets(Fr* __restrict__ z, size_t zs, size_t n, const Fr* __restrict__ mult) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || blockIdx.y == 0) return;
    Fr* p = z + (size_t)blockIdx.y * zs + i;
    fp_store(p, fp_mul(fp_load<FrTag>(p), fp_load<FrTag>(mult + blockIdx.y)));
}

extern "C" int pz_permutation_product_sets_dev(pz_ctx* ctx, const uint64_t* d_cols, size_t col_stride, const uint64_t* d_sigma,
                                               size_t sigma_stride, size_t m, uint32_t chunk_len, uint32_t log_n,
                                               size_t usable_rows, const uint64_t omega[4], const uint64_t beta[4],
                                               const uint64_t gamma[4], const uint64_t delta[4], uint64_t* d_z,
                                               size_t z_stride) {
    if (!ctx || !d_cols || !d_sigma || !d_z || !omega || !beta || !gamma || !delta || m == 0 || chunk_len == 0 || log_n > 26)
        return PZ_ERR_INVALID;
    const size_t n = (size_t)1 << log_n;
    const size_t n_sets = (m + chunk_len - 1) / chunk_len;
    if (col_stride % 4 || sigma_stride % 4 || z_stride % 4 || usable_rows >= n || n_sets > 65535 || m > 0xffffffu) return PZ_ERR_INVALID;
    if ((m > 1 && (col_stride < 4 * n || sigma_stride < 4 * n)) || (n_sets > 1 && z_stride < 4 * n)) return PZ_ERR_INVALID;
    if (!host_fr_canonical(beta) || !host_fr_canonical(gamma) || !host_fr_canonical(delta)) return PZ_ERR_INVALID;   // pz.h: challenges below r
    PZ_ENTER(ctx);
    void *wp, *dp, *ws, *mu;
    PZCHK(pz_get_pow_table(ctx, omega, n, &wp));
    PZCHK(pz_get_pow_table(ctx, delta, m, &dp));
    // numerators and denominators of a BATCH of sets at a time: the workspace is bounded (4 GiB of terms: 512 sets of 2^17 rows, 128 of
    // 2^19) instead of growing with the circuit -- at BASELINE config c5 (1200 sets of 2^19 rows) all sets at once would be 40 GB of
    // library workspace beside a proving key that has to share 288 GB with it.  Sets are independent until the chaining below.
    size_t sb = ((size_t)4 << 30) / (2 * n * 32);
    if (sb < 1) sb = 1;
    if (sb > n_sets) sb = n_sets;
    PZCHK(pz_ws_get(ctx, WS_BIG_C, 2 * sb * n * 32, &ws));
    PZCHK(pz_ws_get(ctx, WS_MISC, n_sets * 32, &mu));
    Fr* num = (Fr*)ws;
    Fr* den = num + sb * n;
    for (size_t s0 = 0; s0 < n_sets; s0 += sb) {
        const size_t ns = n_sets - s0 < sb ? n_sets - s0 : sb;
        hipLaunchKernelGGL(k_perm_terms_sets, dim3(pz_div_up(n, 256), (unsigned)ns), dim3(256), 0, ctx->stream,
                           (const Fr*)d_cols, col_stride / 4, (const Fr*)d_sigma, sigma_stride / 4, (unsigned)m, (unsigned)chunk_len, n,
                           (const Fr*)wp, (const Fr*)dp, host_fr_shl(beta, 10), host_fr_shl(gamma, 5), host_fr_shl(delta, 5), num, den,
                           (unsigned)s0);
        HIPCHK(ctx, hipGetLastError());
        PZCHK(pz_batch_invert_internal(ctx, den, ns * n, num));   // num <- num / den
        PZCHK(pz_prefix_product_batch_internal(ctx, num, n, ns, n, fp_one_host(), (Fr*)d_z + s0 * (z_stride / 4), z_stride / 4));
    }
    if (n_sets > 1) {
        hipLaunchKernelGGL(k_perm_chain, dim3(1), dim3(256), 0, ctx->stream, (const Fr*)d_z, z_stride / 4, (unsigned)n_sets,
                           usable_rows, (Fr*)mu);
        hipLaunchKernelGGL(k_scale_sets, dim3(pz_div_up(n, 256), (unsigned)n_sets), dim3(256), 0, ctx->stream, (Fr*)d_z,
                           z_stride / 4, n, (const Fr*)mu);
        HIPCHK(ctx, hipGetLastError());
    }
    return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- gate part of evaluate_h
// 29-bit field: a1 * shl5(a2) stays in the 256-domain, and acc * y + e * shl5(sel) is one reduction (f29_mul2) -- two reductions
// per column where the 32-bit-limb version paid three full products.  e = a0 + a1 a2 - a3 + 2p: limbs < 2^31, value < 4.2p.
__global__ __launch_bounds__(256) void k_quotient_gate(const Fr* __restrict__ adv, size_t as, const Fr* __restrict__ sel,
                                                       size_t ss, unsigned n_cols, size_t N, unsigned step, S29 y261,
                                                       Fr* __restrict__ h) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const size_t i1 = (i + step) & (N - 1), i2 = (i + 2 * (size_t)step) & (N - 1), i3 = (i + 3 * (size_t)step) & (N - 1);
    Fr29 acc = f29_load<FrTag>(h + i);
    for (unsigned j = 0; j < n_cols; ++j) {
        const Fr* a = adv + (size_t)j * as;
        const Fr29 a0 = f29_load<FrTag>(a + i), a1 = f29_load<FrTag>(a + i1), a3 = f29_load<FrTag>(a + i3);
        const Fr29 a2 = f29_load_shl5<FrTag>(a + i2), sl = f29_load_shl5<FrTag>(sel + (size_t)j * ss + i);
        const Fr29 e = f29_sub<2, 29>(f29_add(a0, f29_mul(a1, a2)), a3);
        acc = f29_mul2_s(acc, y261.v, e, sl);
    }
    f29_store<0>(h + i, acc);
}

extern "C" int pz_quotient_gate_dev(pz_ctx* ctx, const uint64_t* d_adv_ext, size_t adv_stride, const uint64_t* d_sel_ext,
                                    size_t sel_stride, size_t n_cols, uint32_t log_ext, uint32_t rot_step,
                                    const uint64_t y[4], uint64_t* d_h) {
    if (!ctx || !y || !d_h || log_ext > 28 || (n_cols && (!d_adv_ext || !d_sel_ext))) return PZ_ERR_INVALID;
    if (adv_stride % 4 || sel_stride % 4 || n_cols > 0xffffffffu) return PZ_ERR_INVALID;
    const size_t N = (size_t)1 << log_ext;
    if (rot_step == 0 || rot_step >= N) return PZ_ERR_INVALID;
    if (n_cols > 1 && (adv_stride < 4 * N || sel_stride < 4 * N)) return PZ_ERR_INVALID;
    if (!host_fr_canonical(y)) return PZ_ERR_INVALID;
    if (n_cols == 0) return PZ_OK;
    PZ_ENTER(ctx);
    hipLaunchKernelGGL(k_quotient_gate, dim3(pz_div_up(N, 256)), dim3(256), 0, ctx->stream, (const Fr*)d_adv_ext,
                       adv_stride / 4, (const Fr*)d_sel_ext, sel_stride / 4, (unsigned)n_cols, N, (unsigned)rot_step,
                       host_fr_shl(y, 5), (Fr*)d_h);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

// 1 / ((g * w_ext^r)^n - 1), r < E = 2^e: the vanishing polynomial takes only E values on the extended coset
__global__ void k_vanishing_inv(Fr g, Fr w_ext, unsigned log_n, unsigned E, Fr* __restrict__ out) {
    const unsigned r = threadIdx.x;
    if (blockIdx.x || r >= E) return;
    Fr x = g;
    for (unsigned k = 0; k < r; ++k) x = fp_mul(x, w_ext);
    for (unsigned k = 0; k < log_n; ++k) x = fp_sqr(x);
    fp_store(out + r, fp_inv(fp_sub(x, fp_one<FrTag>())));
}
__global__ __launch_bounds__(256) void k_scale_periodic(Fr* __restrict__ h, size_t N, const Fr* __restrict__ tab,
                                                        unsigned mask) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) fp_store(h + i, fp_mul(fp_load<FrTag>(h + i), fp_load<FrTag>(tab + (i & mask))));
}

extern "C" int pz_quotient_finish_dev(pz_ctx* ctx, uint64_t* d_h, uint32_t log_n, uint32_t log_e, const uint64_t coset_g[4],
                                      const uint64_t omega_ext[4]) {
    if (!ctx || !d_h || !coset_g || !omega_ext || log_e > 6 || log_n + log_e > 28) return PZ_ERR_INVALID;   // log_e = 0: one coset of <omega_n>, a single divisor
    PZ_ENTER(ctx);
    const unsigned E = 1u << log_e;
    const size_t N = (size_t)1 << (log_n + log_e);
    void* tab;
    PZCHK(pz_ws_get(ctx, WS_MISC, E * 32, &tab));
    hipLaunchKernelGGL(k_vanishing_inv, dim3(1), dim3(64), 0, ctx->stream, fr_from_u64(coset_g), fr_from_u64(omega_ext),
                       (unsigned)log_n, E, (Fr*)tab);
    hipLaunchKernelGGL(k_scale_periodic, dim3(pz_div_up(N, 256)), dim3(256), 0, ctx->stream, (Fr*)d_h, N, (const Fr*)tab,
                       E - 1);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- a[i] *= c * g^i
__global__ __launch_bounds__(256) void k_mul_table_cols(Fr* __restrict__ a, size_t cs, size_t n, const Fr* __restrict__ tab) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr* p = a + (size_t)blockIdx.y * cs + i;
    fp_store(p, fp_mul(fp_load<FrTag>(p), fp_load<FrTag>(tab + i)));
}

extern "C" int pz_fr_distribute_powers_dev(pz_ctx* ctx, uint64_t* d_a, size_t n_cols, size_t col_stride, size_t n,
                                           const uint64_t g[4], const uint64_t c[4]) {
    if (!ctx || !g || (n_cols && n && !d_a) || col_stride % 4 || (n_cols > 1 && col_stride < 4 * n)) return PZ_ERR_INVALID;
    if (n_cols == 0 || n == 0) return PZ_OK;
    if (n_cols > 65535) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    void* tab;
    PZCHK(pz_get_pow_table(ctx, g, n, &tab, c));
    hipLaunchKernelGGL(k_mul_table_cols, dim3(pz_div_up(n, 256), (unsigned)n_cols), dim3(256), 0, ctx->stream, (Fr*)d_a,
                       col_stride / 4, n, (const Fr*)tab);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- kate_division
// q_{n-2} = a_{n-1}, q_{i-1} = a_i + x * q_i  (the last step's value, p(x), is dropped).  Blocked: chunk c covers
// coefficients [lo_c, hi_c) = [c*K, (c+1)*K); L_c = sum_{i in c} a_i x^(i - lo_c);
// carry_in(c) = q_{hi_c - 1} = sum_{i >= hi_c} a_i x^(i - hi_c) = L_{c+1} + x^K carry_in(c+1), carry_in(top) = q_{n-1} = 0
#define KD_K 64u
__global__ __launch_bounds__(256) void k_kd_local(const Fr* __restrict__ a, size_t cs, size_t n, Fr x, Fr* __restrict__ L,
                                                  size_t n_chunks) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const Fr* p = a + (size_t)blockIdx.y * cs;
    const size_t lo = c * KD_K, hi = (lo + KD_K < n ? lo + KD_K : n);
    Fr v = fp_zero<FrTag>();
    for (size_t i = hi; i-- > lo;) v = fp_add(fp_mul(v, x), fp_load<FrTag>(p + i));
    fp_store(L + (size_t)blockIdx.y * n_chunks + c, v);
}
// one workgroup per column: thread t owns `per` consecutive chunks; its segment maps the carry entering at its top to
// the carry leaving at its bottom as out = V + X * in (V = the segment's own value, X = x^(K * chunks)); a suffix scan of
// these affine maps over the 256 threads gives every segment its carry-in, then the chunks are walked once more.
__global__ __launch_bounds__(256) void k_kd_carry(Fr* __restrict__ L, size_t n_chunks, Fr x) {
    __shared__ Fr s_v[256], s_x[256];
    Fr* l = L + (size_t)blockIdx.x * n_chunks;
    Fr xk = x;
    for (unsigned s = 1; s < KD_K; s <<= 1) xk = fp_sqr(xk);  // x^64
    const size_t per = (n_chunks + 255) / 256;
    const size_t lo = threadIdx.x * per, hi = lo + per < n_chunks ? lo + per : n_chunks;
    Fr V = fp_zero<FrTag>(), X = fp_one<FrTag>();
    for (size_t c = hi; c-- > lo && c < n_chunks;) {
        V = fp_add(fp_load<FrTag>(l + c), fp_mul(xk, V));
        X = fp_mul(X, xk);
    }
    s_v[threadIdx.x] = V;
    s_x[threadIdx.x] = X;
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {  // suffix scan: compose with the segments above
        Fr v = s_v[threadIdx.x], xx = s_x[threadIdx.x];
        Fr uv = fp_zero<FrTag>(), ux = fp_one<FrTag>();
        const bool has = threadIdx.x + off < 256;
        if (has) { uv = s_v[threadIdx.x + off]; ux = s_x[threadIdx.x + off]; }
        __syncthreads();
        if (has) {
            s_v[threadIdx.x] = fp_add(v, fp_mul(xx, uv));
            s_x[threadIdx.x] = fp_mul(xx, ux);
        }
        __syncthreads();
    }
    Fr carry = threadIdx.x + 1 < 256 ? s_v[threadIdx.x + 1] : fp_zero<FrTag>();
    for (size_t c = hi; c-- > lo && c < n_chunks;) {
        Fr t = fp_load<FrTag>(l + c);
        fp_store(l + c, carry);  // carry_in(c)
        carry = fp_add(t, fp_mul(xk, carry));
    }
}
// a and q may alias (SHPLONK divides in place): neither is __restrict__, and every coefficient is read before its slot is written
__global__ __launch_bounds__(256) void k_kd_apply(const Fr* a, size_t cs, size_t n, Fr x,
                                                  const Fr* __restrict__ L, size_t n_chunks, Fr* q, size_t qs) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const Fr* p = a + (size_t)blockIdx.y * cs;
    Fr* o = q + (size_t)blockIdx.y * qs;
    const size_t lo = c * KD_K, hi = (lo + KD_K < n ? lo + KD_K : n);
    Fr cur = fp_load<FrTag>(L + (size_t)blockIdx.y * n_chunks + c);  // q_{hi-1}
    for (size_t i = hi; i-- > lo;) {
        Fr ai = fp_load<FrTag>(p + i);         // read first: q may alias the coefficients
        fp_store(o + i, cur);                  // q_i (q_{n-1} = 0: the quotient has n-1 coefficients)
        cur = fp_add(ai, fp_mul(x, cur));      // q_{i-1}; the last one, p(x), is dropped
    }
}

extern "C" int pz_poly_div_linear_dev(pz_ctx* ctx, const uint64_t* d_coeffs, size_t n_cols, size_t col_stride, size_t n,
                                      const uint64_t x[4], uint64_t* d_q, size_t q_stride) {
    if (!ctx || !x || (n_cols && n && (!d_coeffs || !d_q)) || col_stride % 4 || q_stride % 4) return PZ_ERR_INVALID;
    if (n_cols > 1 && (col_stride < 4 * n || q_stride < 4 * n)) return PZ_ERR_INVALID;
    if (n_cols == 0 || n == 0) return PZ_OK;
    if (n_cols > 65535) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    const size_t nch = pz_div_up(n, KD_K);
    void* L;
    PZCHK(pz_ws_get(ctx, WS_BIG_A, n_cols * nch * 32, &L));
    Fr xv = fr_from_u64(x);
    hipLaunchKernelGGL(k_kd_local, dim3(pz_div_up(nch, 256), (unsigned)n_cols), dim3(256), 0, ctx->stream,
                       (const Fr*)d_coeffs, col_stride / 4, n, xv, (Fr*)L, nch);
    hipLaunchKernelGGL(k_kd_carry, dim3((unsigned)n_cols), dim3(256), 0, ctx->stream, (Fr*)L, nch, xv);
    hipLaunchKernelGGL(k_kd_apply, dim3(pz_div_up(nch, 256), (unsigned)n_cols), dim3(256), 0, ctx->stream,
                       (const Fr*)d_coeffs, col_stride / 4, n, xv, (const Fr*)L, nch, (Fr*)d_q, q_stride / 4);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- evaluate_h: permutation
// halo2 plonk/evaluation.rs, "Permutations" block, on the extended coset (point X_i = x0 * w_ext^i):
//   h = h*y + l0 (1 - z_0)
//   h = h*y + l_last (z_last^2 - z_last)
//   for sets j > 0:  h = h*y + l0 (z_j - z_{j-1}(w^-(blinding+1) X))
//   for every set j: h = h*y + l_active ( z_j(wX) prod_c (v_c + beta sigma_c + gamma) - z_j(X) prod_c (v_c + delta^c beta X + gamma) )
// with c running over the set's chunk of columns and delta^c continuing across sets.
struct PermQ {
    const Fr *cols, *sigma, *z, *l0, *llast, *lactive, *xpow;
    size_t cs, ss, zs, N;
    unsigned n_sets, chunk_len, m_total, step, last_rot;
    // a call may cover only the sets [set_lo, set_lo + n_sets) of n_sets_total (pz_quotient_permutation_part_dev: the prover streams
    // 64-column tiles through the extension and this kernel): z always holds ALL sets' products (row stride zs), cols / sigma start
    // at the call's first column, m_total counts the call's columns; head != 0 adds the boundary and chaining lines (first call only)
    unsigned set_lo, n_sets_total, head;
    S29 y, gamma, delta, beta266;   // challenges as SGPR-resident limbs (host_fr_shl): y, gamma, delta times 2^261, beta times 2^266
    S29 bx266;                      // beta * delta^(index of the call's first column) times 2^266: the identity term's start
    S29 y_chain, y_sets;            // y^(n_sets_total - 1) and y^n_sets times 2^261: the Horner steps of a whole GROUP of lines
};
// On the 29-bit field (fp29.cuh): ~210 instructions per product instead of ~380 on saturated 32-bit limbs.  Domains: memory
// holds x * 2^256; f29_mul divides by 2^261, so in every product exactly one operand carries the extra 2^5 -- a challenge
// converted once per thread (c261), or a loaded value unpacked through f29_load_shl5.  acc * y + term * l is one reduction
// (f29_mul2).  Limb / value bounds: products are tight (< 2^29, value < 2p); sums of up to three tight values and
// f29_sub<2, 29> results stay below 2^31, a legal operand against a tight one.
// A thread takes the TWO rows i and i + N/2: X_{i + N/2} = -X_i on the extended coset (w_ext^(N/2) = -1), so the identity term
// beta * delta^c * X of the second row is the first row's negated and the product cur * delta per column is paid once for the pair
// (3.5 instead of 4 products per row and column; the kernel is at its VALU floor -- profiles/r04_pmc_tail_sq_counters.txt -- so only
// fewer products shorten it).
__global__ __launch_bounds__(256) void k_quotient_permutation(PermQ q, Fr* __restrict__ h) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t half = q.N >> 1;
    if (i0 >= half) return;
    const size_t mask = q.N - 1;
    size_t row[2], i_next[2], i_last[2];
    Fr29 acc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        row[r] = i0 + (r ? half : 0);
        i_next[r] = (row[r] + q.step) & mask;
        i_last[r] = (row[r] + q.N - (size_t)q.last_rot * q.step) & mask;
        acc[r] = f29_load<FrTag>(h + row[r]);
    }
#pragma unroll
    for (int r = 0; r < 2 && q.head; ++r) {   // the boundary lines: l0 and l_last rows are live only here
        const size_t i = row[r];
        const Fr29 one = fr29_one256();
        const Fr29 l0 = f29_load_shl5<FrTag>(q.l0 + i);
        const Fr29 z_first = f29_load<FrTag>(q.z + i);
        acc[r] = f29_mul2_s(acc[r], q.y.v, f29_sub<2, 29>(one, z_first), l0);
        {
            const Fr29 ll = f29_load_shl5<FrTag>(q.llast + i);
            const Fr zl = fp_load<FrTag>(q.z + (size_t)(q.n_sets_total - 1) * q.zs + i);
            const Fr29 z_lastset = f29_from_fp(zl);
            acc[r] = f29_mul2_s(acc[r], q.y.v, f29_sub<2, 29>(f29_mul(z_lastset, f29_from_fp_shl5(zl)), z_lastset), ll);
        }
        // the n_sets - 1 chaining lines share their factor l0: sum_j y^(S-1-j) l0 d_j = l0 * (Horner of the d_j in y), so the group
        // costs one plain product per line and ONE two-term reduction, instead of a two-term reduction per line (same field value,
        // hence the same canonical h)
        if (q.n_sets_total > 1) {
            Fr29 t0 = f29_sub<2, 29>(f29_load<FrTag>(q.z + q.zs + i), f29_load<FrTag>(q.z + i_last[r]));
            for (unsigned j = 2; j < q.n_sets_total; ++j) {
                const Fr29 d = f29_sub<2, 29>(f29_load<FrTag>(q.z + (size_t)j * q.zs + i), f29_load<FrTag>(q.z + (size_t)(j - 1) * q.zs + i_last[r]));
                t0 = f29_add(f29_mul_s(t0, q.y.v), d);   // tight + (< 2^31): a legal operand of the next product
            }
            acc[r] = f29_mul2_s(acc[r], q.y_chain.v, t0, l0);
        }
    }
    // beta * X_i (261-domain), X_i = x0 * w_ext^i from the cached power table; the second row's is its negative
    Fr29 cur = f29_mul_s(f29_load<FrTag>(q.xpow + i0), q.bx266.v);
    unsigned c = 0;
    // the next column's values and sigmas are requested before the current column's products start
    Fr v_n[2], s_n[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        v_n[r] = fp_load<FrTag>(q.cols + row[r]);
        s_n[r] = fp_load<FrTag>(q.sigma + row[r]);
    }
    Fr29 hs[2];   // Horner in y of the sets' (left - right): the factor l_active is applied once to the whole group
    for (unsigned j = 0; j < q.n_sets; ++j) {
        Fr29 left[2], right[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            left[r] = f29_load<FrTag>(q.z + (size_t)(q.set_lo + j) * q.zs + i_next[r]);
            right[r] = f29_load<FrTag>(q.z + (size_t)(q.set_lo + j) * q.zs + row[r]);
        }
        for (unsigned t = 0; t < q.chunk_len && c < q.m_total; ++t, ++c) {
            const unsigned cn = c + 1 < q.m_total ? c + 1 : c;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const Fr29 v = f29_add_s(f29_from_fp_shl5(v_n[r]), q.gamma.v);   // (v + gamma) * 2^261
                const Fr29 sg = f29_from_fp(s_n[r]);
                v_n[r] = fp_load<FrTag>(q.cols + (size_t)cn * q.cs + row[r]);
                s_n[r] = fp_load<FrTag>(q.sigma + (size_t)cn * q.ss + row[r]);
                left[r] = f29_mul(left[r], f29_add(v, f29_mul_s(sg, q.beta266.v)));
                // v + beta delta^c X: row 0 adds cur, row 1 (X negated) subtracts it (cur is a product: tight, below 2p)
                right[r] = f29_mul(right[r], r ? f29_sub<2, 29>(v, cur) : f29_add(v, cur));
            }
            cur = f29_mul_s(cur, q.delta.v);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const Fr29 diff = f29_sub<2, 29>(left[r], right[r]);
            hs[r] = j ? f29_add(f29_mul_s(hs[r], q.y.v), diff) : diff;
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        acc[r] = f29_mul2_s(acc[r], q.y_sets.v, hs[r], f29_load_shl5<FrTag>(q.lactive + row[r]));
        f29_store<0>(h + row[r], acc[r]);
    }
}

extern "C" int pz_quotient_permutation_part_dev(pz_ctx* ctx, const uint64_t* d_cols_ext, size_t col_stride,
                                                const uint64_t* d_sigma_ext, size_t sigma_stride, const uint64_t* d_z_ext,
                                                size_t z_stride, uint32_t n_sets_total, uint32_t set_lo, uint32_t n_sets,
                                                uint32_t chunk_len, uint32_t m_cols, int head, uint32_t log_ext, uint32_t rot_step,
                                                uint32_t last_rotation, const uint64_t* d_l0, const uint64_t* d_l_last,
                                                const uint64_t* d_l_active, const uint64_t beta[4], const uint64_t gamma[4],
                                                const uint64_t delta[4], const uint64_t coset_g[4], const uint64_t omega_ext[4],
                                                const uint64_t y[4], uint64_t* d_h) {
    if (!ctx || !d_cols_ext || !d_sigma_ext || !d_z_ext || !d_l0 || !d_l_last || !d_l_active || !beta || !gamma || !delta ||
        !coset_g || !omega_ext || !y || !d_h)
        return PZ_ERR_INVALID;
    if (log_ext > 28 || log_ext == 0 || n_sets == 0 || chunk_len == 0 || m_cols == 0 || col_stride % 4 || sigma_stride % 4 || z_stride % 4)
        return PZ_ERR_INVALID;
    if (n_sets_total == 0 || (size_t)set_lo + n_sets > n_sets_total) return PZ_ERR_INVALID;
    const size_t N = (size_t)1 << log_ext;
    // the call's columns fill its sets; only the LAST set of the whole argument may be short
    if ((size_t)n_sets * chunk_len < m_cols || (size_t)(n_sets - 1) * chunk_len >= m_cols) return PZ_ERR_INVALID;
    if (set_lo + n_sets < n_sets_total && (size_t)n_sets * chunk_len != m_cols) return PZ_ERR_INVALID;
    if (rot_step == 0 || rot_step >= N || (size_t)last_rotation * rot_step >= N) return PZ_ERR_INVALID;
    if ((m_cols > 1 && (col_stride < 4 * N || sigma_stride < 4 * N)) || (n_sets_total > 1 && z_stride < 4 * N)) return PZ_ERR_INVALID;
    if (!host_fr_canonical(beta) || !host_fr_canonical(gamma) || !host_fr_canonical(delta) || !host_fr_canonical(y)) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    PermQ q;
    q.cols = (const Fr*)d_cols_ext; q.sigma = (const Fr*)d_sigma_ext; q.z = (const Fr*)d_z_ext;
    q.l0 = (const Fr*)d_l0; q.llast = (const Fr*)d_l_last; q.lactive = (const Fr*)d_l_active;
    q.cs = col_stride / 4; q.ss = sigma_stride / 4; q.zs = z_stride / 4; q.N = N;
    q.n_sets = n_sets; q.chunk_len = chunk_len; q.m_total = m_cols; q.step = rot_step; q.last_rot = last_rotation;
    q.set_lo = set_lo; q.n_sets_total = n_sets_total; q.head = head ? 1u : 0u;
    q.beta266 = host_fr_shl(beta, 10); q.gamma = host_fr_shl(gamma, 5); q.delta = host_fr_shl(delta, 5); q.y = host_fr_shl(y, 5);
    uint64_t yp[4], bx[4];
    host_fr_pow(delta, set_lo * chunk_len, bx);       // delta^(first column of the call)
    host_fr_mul(bx, beta, bx);
    q.bx266 = host_fr_shl(bx, 10);
    host_fr_pow(y, n_sets_total - 1, yp);
    q.y_chain = host_fr_shl(yp, 5);
    host_fr_pow(y, n_sets, yp);
    q.y_sets = host_fr_shl(yp, 5);
    void* xp;
    PZCHK(pz_get_pow_table(ctx, omega_ext, N, &xp, coset_g));   // X_i = coset_g * omega_ext^i, cached across calls
    q.xpow = (const Fr*)xp;
    hipLaunchKernelGGL(k_quotient_permutation, dim3(pz_div_up(N / 2, 256)), dim3(256), 0, ctx->stream, q, (Fr*)d_h);   // a thread per row PAIR
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}
extern "C" int pz_quotient_permutation_dev(pz_ctx* ctx, const uint64_t* d_cols_ext, size_t col_stride,
                                           const uint64_t* d_sigma_ext, size_t sigma_stride, const uint64_t* d_z_ext,
                                           size_t z_stride, uint32_t n_sets, uint32_t chunk_len, uint32_t m_total,
                                           uint32_t log_ext, uint32_t rot_step, uint32_t last_rotation,
                                           const uint64_t* d_l0, const uint64_t* d_l_last, const uint64_t* d_l_active,
                                           const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta[4],
                                           const uint64_t coset_g[4], const uint64_t omega_ext[4], const uint64_t y[4],
                                           uint64_t* d_h) {
    return pz_quotient_permutation_part_dev(ctx, d_cols_ext, col_stride, d_sigma_ext, sigma_stride, d_z_ext, z_stride, n_sets, 0, n_sets,
                                            chunk_len, m_total, 1, log_ext, rot_step, last_rotation, d_l0, d_l_last, d_l_active, beta, gamma,
                                            delta, coset_g, omega_ext, y, d_h);
}

// ---------------------------------------------------------------------------------------------- evaluate_h: lookups
// per lookup (input a, table s, permuted a', s', product z), in halo2's order:
//   h = h*y + l0 (1 - z);  h = h*y + l_last (z^2 - z)
//   h = h*y + l_active ( z(wX)(a' + beta)(s' + gamma) - z(X)(a + beta)(s + gamma) )
//   h = h*y + l0 (a' - s');  h = h*y + l_active (a' - s')(a' - a'(w^-1 X))
struct LookQ {
    const Fr *a, *s, *ap, *sp, *z, *l0, *llast, *lactive;
    size_t as, aps, sps, zs, N;
    unsigned n_lookups, step;
    S29 beta, gamma, y;   // times 2^261, SGPR-resident limbs (host_fr_shl)
};
// (29-bit field, domains as in k_quotient_permutation: beta, gamma, y and the l_* rows carry 2^261; every line of the argument is
// one f29_mul2 with acc * y)
__global__ __launch_bounds__(256) void k_quotient_lookup(LookQ q, Fr* __restrict__ h) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.N) return;
    const size_t mask = q.N - 1;
    const size_t i_next = (i + q.step) & mask, i_prev = (i + q.N - q.step) & mask;
    const Fr29 one = fr29_one256();
    const Fr29 l0 = f29_load_shl5<FrTag>(q.l0 + i), ll = f29_load_shl5<FrTag>(q.llast + i), la = f29_load_shl5<FrTag>(q.lactive + i);
    const Fr29 sg = f29_add_s(f29_load_shl5<FrTag>(q.s + i), q.gamma.v);   // (s + gamma) 2^261
    Fr29 acc = f29_load<FrTag>(h + i);
    for (unsigned k = 0; k < q.n_lookups; ++k) {
        const Fr zf = fp_load<FrTag>(q.z + (size_t)k * q.zs + i);
        const Fr29 z = f29_from_fp(zf), zn = f29_load<FrTag>(q.z + (size_t)k * q.zs + i_next);
        const Fr apf = fp_load<FrTag>(q.ap + (size_t)k * q.aps + i), spf = fp_load<FrTag>(q.sp + (size_t)k * q.sps + i);
        const Fr29 ap = f29_from_fp(apf), sp = f29_from_fp(spf);
        acc = f29_mul2_s(acc, q.y.v, f29_sub<2, 29>(one, z), l0);
        acc = f29_mul2_s(acc, q.y.v, f29_sub<2, 29>(f29_mul(z, f29_from_fp_shl5(zf)), z), ll);
        // (a' + beta)(s' + gamma) and (a + beta)(s + gamma): both factors carry 2^261, so does the product (loose x loose limbs:
        // 9 * 2^30 * 2^30 + 2^59.8 < 2^64; value < 8p), and z * that is back in the 256-domain
        const Fr29 f1 = f29_mul(f29_add_s(f29_from_fp_shl5(apf), q.beta.v), f29_add_s(f29_from_fp_shl5(spf), q.gamma.v));
        const Fr29 f2 = f29_mul(f29_add_s(f29_load_shl5<FrTag>(q.a + (size_t)k * q.as + i), q.beta.v), sg);
        acc = f29_mul2_s(acc, q.y.v, f29_sub<2, 29>(f29_mul(zn, f1), f29_mul(z, f2)), la);
        const Fr29 ams = f29_carry(f29_sub<2, 29>(ap, sp));   // a' - s' + 2p, limbs < 2^29 + 8
        acc = f29_mul2_s(acc, q.y.v, ams, l0);
        // (a' - a'(w^-1 X)) * 2^261 from the shl5 images (values below 32p each; + 64p keeps every limb and the value positive)
        const Fr29 dprev = f29_sub<64, 29>(f29_from_fp_shl5(apf), f29_load_shl5<FrTag>(q.ap + (size_t)k * q.aps + i_prev));
        acc = f29_mul2_s(acc, q.y.v, f29_mul(ams, dprev), la);
    }
    f29_store<0>(h + i, acc);
}

extern "C" int pz_quotient_lookup_dev(pz_ctx* ctx, const uint64_t* d_input_ext, size_t input_stride, const uint64_t* d_table_ext,
                                      const uint64_t* d_perm_input_ext, size_t perm_input_stride,
                                      const uint64_t* d_perm_table_ext, size_t perm_table_stride, const uint64_t* d_z_ext,
                                      size_t z_stride, uint32_t n_lookups, uint32_t log_ext, uint32_t rot_step,
                                      const uint64_t* d_l0, const uint64_t* d_l_last, const uint64_t* d_l_active,
                                      const uint64_t beta[4], const uint64_t gamma[4], const uint64_t y[4], uint64_t* d_h) {
    if (!ctx || !d_input_ext || !d_table_ext || !d_perm_input_ext || !d_perm_table_ext || !d_z_ext || !d_l0 || !d_l_last ||
        !d_l_active || !beta || !gamma || !y || !d_h)
        return PZ_ERR_INVALID;
    if (log_ext > 28 || input_stride % 4 || perm_input_stride % 4 || perm_table_stride % 4 || z_stride % 4) return PZ_ERR_INVALID;
    const size_t N = (size_t)1 << log_ext;
    if (rot_step == 0 || rot_step >= N) return PZ_ERR_INVALID;
    if (n_lookups > 1 && (input_stride < 4 * N || perm_input_stride < 4 * N || perm_table_stride < 4 * N || z_stride < 4 * N))
        return PZ_ERR_INVALID;
    if (!host_fr_canonical(beta) || !host_fr_canonical(gamma) || !host_fr_canonical(y)) return PZ_ERR_INVALID;
    if (n_lookups == 0) return PZ_OK;
    PZ_ENTER(ctx);
    LookQ q;
    q.a = (const Fr*)d_input_ext; q.s = (const Fr*)d_table_ext; q.ap = (const Fr*)d_perm_input_ext;
    q.sp = (const Fr*)d_perm_table_ext; q.z = (const Fr*)d_z_ext;
    q.l0 = (const Fr*)d_l0; q.llast = (const Fr*)d_l_last; q.lactive = (const Fr*)d_l_active;
    q.as = input_stride / 4; q.aps = perm_input_stride / 4; q.sps = perm_table_stride / 4; q.zs = z_stride / 4; q.N = N;
    q.n_lookups = n_lookups; q.step = rot_step;
    q.beta = host_fr_shl(beta, 5); q.gamma = host_fr_shl(gamma, 5); q.y = host_fr_shl(y, 5);
    hipLaunchKernelGGL(k_quotient_lookup, dim3(pz_div_up(N, 256)), dim3(256), 0, ctx->stream, q, (Fr*)d_h);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- linear combination
// out[i] = sum_j v^(n_cols-1-j) * p_j[i]  (Horner over the columns: acc = acc*v + p_j): the random linear
// combinations of the multiopen argument (SHPLONK / GWC fold the polynomials opened at one point set with powers
// of a challenge before the division by the set's vanishing factors, pz_poly_div_linear_dev per point).
__global__ __launch_bounds__(256) void k_lincomb(const Fr* __restrict__ p, size_t cs, unsigned n_cols, size_t n, Fr v,
                                                 Fr* __restrict__ out, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr acc = accumulate ? fp_load<FrTag>(out + i) : fp_zero<FrTag>();
    for (unsigned j = 0; j < n_cols; ++j) acc = fp_add(fp_mul(acc, v), fp_load<FrTag>(p + (size_t)j * cs + i));
    fp_store(out + i, acc);
}

extern "C" int pz_fr_lincomb_dev(pz_ctx* ctx, const uint64_t* d_polys, size_t n_cols, size_t col_stride, size_t n,
                                 const uint64_t v[4], uint64_t* d_out, int accumulate) {
    if (!ctx || !v || (n && (!d_out || (n_cols && !d_polys))) || col_stride % 4 || (n_cols > 1 && col_stride < 4 * n))
        return PZ_ERR_INVALID;
    if (n_cols > 0xffffffffu) return PZ_ERR_INVALID;
    if (n == 0) return PZ_OK;
    PZ_ENTER(ctx);
    hipLaunchKernelGGL(k_lincomb, dim3(pz_div_up(n, 256)), dim3(256), 0, ctx->stream, (const Fr*)d_polys, col_stride / 4,
                       (unsigned)n_cols, n, fr_from_u64(v), (Fr*)d_out, accumulate);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}
