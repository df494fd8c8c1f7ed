// pz_probe.hip -- MEASUREMENT PROBES (libpz_probe.so): issue-rate microbenchmarks and alternative field products.  NOT part of
// the product ABI (include/pz.h) and never linked into libpz_hip.so: bench.py loads this library only to measure the live
// v_mad_u64_u32 issue peak `roofline_int` is priced against; profiles/probes/*.py use the rest.  Each entry point takes the HIP
// device ordinal, runs on the null stream and returns elapsed device milliseconds of the second of two launches.
#define PZ_FP_MUL_VARIANTS 1
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../fp.cuh"
#include "../fp29_probe.cuh"
#include "../fp29.cuh"

// issue-rate probes --------------------------------------------------------------------------
__global__ void k_ubench_mad(u64* out, unsigned iters) {
    u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
    u64 x0 = a, x1 = b, x2 = a ^ b, x3 = a + b, x4 = a * 3, x5 = b * 5, x6 = a - b, x7 = ~a;
    for (unsigned i = 0; i < iters; ++i) {
        // 8 independent accumulators: measures issue throughput, not dependent latency
        x0 = (u64)a * (u32)x0 + x0;
        x1 = (u64)b * (u32)x1 + x1;
        x2 = (u64)a * (u32)x2 + x2;
        x3 = (u64)b * (u32)x3 + x3;
        x4 = (u64)a * (u32)x4 + x4;
        x5 = (u64)b * (u32)x5 + x5;
        x6 = (u64)a * (u32)x6 + x6;
        x7 = (u64)b * (u32)x7 + x7;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

// the same with multiplicands that do NOT depend on the accumulators (what a column of a field product looks like: only the
// 64-bit addend chains): the issue rate the 29-bit kernels actually see
__global__ void k_ubench_mad_indep(u64* out, unsigned iters) {
    u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u, c = a ^ 0x9e3779b9u, d = b + 0x7f4a7c15u;
    u64 x0 = a, x1 = b, x2 = a ^ b, x3 = a + b, x4 = a * 3, x5 = b * 5, x6 = a - b, x7 = ~a;
    for (unsigned i = 0; i < iters; ++i) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %9, %10, %1\n\tv_mad_u64_u32 %2, vcc, %10, %11, %2\n\t"
                     "v_mad_u64_u32 %3, vcc, %8, %11, %3\n\tv_mad_u64_u32 %4, vcc, %8, %10, %4\n\tv_mad_u64_u32 %5, vcc, %9, %11, %5\n\t"
                     "v_mad_u64_u32 %6, vcc, %8, %8, %6\n\tv_mad_u64_u32 %7, vcc, %9, %9, %7"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                     : "v"(a), "v"(b), "v"(c), "v"(d)
                     : "vcc");
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

__global__ void k_ubench_fqmul(Fq* out, unsigned iters) {
    Fq x = fp_one<FqTag>(), y = fp_one<FqTag>();
    x.v[0] ^= threadIdx.x;
    y.v[1] ^= blockIdx.x;
    for (unsigned i = 0; i < iters; ++i) {
        x = fp_mul(x, y);
        y = fp_mul(y, x);
    }
    fp_store(out + (size_t)blockIdx.x * blockDim.x + threadIdx.x, fp_add(x, y));
}

// the same chain with another product: 1 = round 1's back-to-back mad/addc pairs (no wait states -- timing only,
// its results are not trusted), 2 = the 9 x 29-bit no-carry product of fp29_probe.cuh
__global__ void k_ubench_fqmul_nowait(Fq* out, unsigned iters) {
    Fq x = fp_one<FqTag>(), y = fp_one<FqTag>();
    x.v[0] ^= threadIdx.x;
    y.v[1] ^= blockIdx.x;
    for (unsigned i = 0; i < iters; ++i) {
        x = fp_mul_nowait(x, y);
        y = fp_mul_nowait(y, x);
    }
    fp_store(out + (size_t)blockIdx.x * blockDim.x + threadIdx.x, fp_add(x, y));
}
__global__ void k_ubench_fqmul29(Fq* out, unsigned iters) {
    Fq x0 = fp_one<FqTag>(), y0 = fp_one<FqTag>();
    x0.v[0] ^= threadIdx.x;
    y0.v[1] ^= blockIdx.x;
    Fq29 x = fq29_from_words(x0.v), y = fq29_from_words(y0.v);
    for (unsigned i = 0; i < iters; ++i) {
        x = fq29_mul(x, y);
        y = fq29_mul(y, x);
    }
    Fq r;
    fq29_to_words(x, r.v);
    Fq r2;
    fq29_to_words(y, r2.v);
#pragma unroll
    for (int k = 0; k < 8; ++k) r.v[k] ^= r2.v[k];
    uint4* q = reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * blockDim.x + threadIdx.x);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
// variants 3 / 4: the production 29-bit product / square of fp29.cuh (asm columns)
template <int SQR> __global__ void k_ubench_f29(Fq* out, unsigned iters) {
    Fq x0 = fp_one<FqTag>(), y0 = fp_one<FqTag>();
    x0.v[0] ^= threadIdx.x;
    y0.v[1] ^= blockIdx.x;
    F29<FqTag> x = f29_from_fp(x0), y = f29_from_fp(y0);
    for (unsigned i = 0; i < iters; ++i) {
        if (SQR) {
            x = f29_sqr(y);
            y = f29_sqr(x);
        } else {
            x = f29_mul(x, y);
            y = f29_mul(y, x);
        }
    }
    f29_store<1>(out + (size_t)blockIdx.x * blockDim.x + threadIdx.x, f29_mul(x, y));
}
// variant 5: the constant-operand product f29_mulc (Barrett / Shoup: 143 multiplier instructions), the constants held in registers as a
// kernel keeps a loaded table entry; same dependency shape as variant 3 (two chains, each product waits for the previous one)
__global__ void k_ubench_f29c(Fq* out, unsigned iters) {
    Fq x0 = fp_one<FqTag>(), y0 = fp_one<FqTag>();
    x0.v[0] ^= threadIdx.x;
    y0.v[1] ^= blockIdx.x;
    F29<FqTag> x = f29_from_fp(x0), y = f29_from_fp(y0), w = f29_from_fp(y0), wq = f29_from_fp(x0);
    w.v[8] &= 0xffffu;
    for (unsigned i = 0; i < iters; ++i) {
        x = f29_mulc(x, w, wq);
        y = f29_mulc(y, wq, w);
    }
    f29_store<1>(out + (size_t)blockIdx.x * blockDim.x + threadIdx.x, f29_mul(x, y));
}
// one product of the probe, for its correctness check: out = a * b * 2^-261 mod p as a 256-bit integer below 2p
__global__ void k_fq_mul29(const u32* a, const u32* b, u32* out) {
    u32 aw[8], bw[8], rw[8];
    for (int k = 0; k < 8; ++k) {
        aw[k] = a[k];
        bw[k] = b[k];
    }
    Fq29 r = fq29_mul(fq29_from_words(aw), fq29_from_words(bw));
    fq29_to_words(r, rw);
    for (int k = 0; k < 8; ++k) out[k] = rw[k];
}


template <class K, class... A> static int timed_launch(int device, double* ms, K kern, dim3 g, dim3 b, A... args) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    hipLaunchKernelGGL(kern, g, b, 0, 0, args...);  // warm
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, g, b, 0, 0, args...);
    (void)hipEventRecord(e1, 0);
    int rc = hipEventSynchronize(e1) == hipSuccess ? 0 : -1;
    float f = 0;
    if (rc == 0 && hipEventElapsedTime(&f, e0, e1) != hipSuccess) rc = -1;
    *ms = f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
static void* probe_buf(int device, size_t bytes) {
    static void* d = nullptr;
    static size_t cap = 0;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    if (cap < bytes) {
        if (d) (void)hipFree(d);
        d = nullptr;
        cap = 0;
        if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
        cap = bytes;
    }
    return d;
}

extern "C" int pzp_ubench_mad(int device, uint32_t blocks, uint32_t iters, double* ms) {
    void* d = probe_buf(device, (size_t)blocks * 256 * 32);
    if (!d || !ms || !blocks) return -1;
    return timed_launch(device, ms, k_ubench_mad, dim3(blocks), dim3(256), (u64*)d, (unsigned)iters);
}
extern "C" int pzp_ubench_mad_indep(int device, uint32_t blocks, uint32_t iters, double* ms) {
    void* d = probe_buf(device, (size_t)blocks * 256 * 32);
    if (!d || !ms || !blocks) return -1;
    return timed_launch(device, ms, k_ubench_mad_indep, dim3(blocks), dim3(256), (u64*)d, (unsigned)iters);
}
// variant 0 = fp_mul 8 x 32 asm, 1 = the same without its wait states (timing only), 2 = 9 x 29 plain C, 3 = f29_mul (asm columns),
// 4 = f29_sqr
extern "C" int pzp_ubench_fqmul_variant(int device, int variant, uint32_t blocks, uint32_t iters, double* ms) {
    void* d = probe_buf(device, (size_t)blocks * 256 * 32);
    if (!d || !ms || !blocks || variant < 0 || variant > 5) return -1;
    if (variant == 5) return timed_launch(device, ms, k_ubench_f29c, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
    if (variant == 1) return timed_launch(device, ms, k_ubench_fqmul_nowait, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
    if (variant == 3) return timed_launch(device, ms, k_ubench_f29<0>, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
    if (variant == 4) return timed_launch(device, ms, k_ubench_f29<1>, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
    if (variant == 2) return timed_launch(device, ms, k_ubench_fqmul29, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
    return timed_launch(device, ms, k_ubench_fqmul, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
}
// one product of variant 2 for its correctness check: a, b 256-bit integers below 2p; out = a * b * 2^-261 mod p below 2p
extern "C" int pzp_fq_mul29(int device, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    void* d = probe_buf(device, 96);
    if (!d || !a || !b || !out) return -1;
    if (hipMemcpy(d, a, 32, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy((char*)d + 32, b, 32, hipMemcpyHostToDevice) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_fq_mul29, dim3(1), dim3(1), 0, 0, (const u32*)d, (const u32*)d + 8, (u32*)d + 16);
    if (hipMemcpy(out, (char*)d + 64, 32, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}

// ---- element-wise checks of the 29-bit field's building blocks on RAW limb inputs (loose limbs allowed: the tests feed the
// documented operand bounds, where a column of the product scan is closest to 2^64).  in: [count][k][9] words, out: [count][9].
//   op 0: f29_mul(a, b)   1: f29_sqr(a)   2: f29_mul2(a, b, c, d)   3: f29_dot4(a0..a3, b0..b3) (k = 8: a0 b0 a1 b1 ..)
//   op 4: f29_unpack_shl5 of an 8-word integer (k = 1, the ninth word ignored)   5: f29_canon<4> of a loose value (k = 1)
//   op 6: the canonical 8-word image f29_store_product writes for a strict value below 2p (k = 1; out words 0..7)
//   op 7: f29_canon_q of a loose value below 32p (k = 1), the quotient table in LDS as the transforms' last pass keeps it
template <class T> __global__ void k_f29_ops(int op, const u32* __restrict__ in, unsigned k, size_t count, u32* __restrict__ out) {
    __shared__ u32 qtab[F29_QTAB_WORDS];
    f29_qtab_fill<T>(qtab);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    F29<T> x[8];
    for (unsigned j = 0; j < k && j < 8; ++j) x[j] = f29_load_raw<T>(in + (i * k + j) * 9);
    F29<T> r = f29_zero<T>();
    if (op == 0) r = f29_mul(x[0], x[1]);
    else if (op == 1) r = f29_sqr(x[0]);
    else if (op == 2) r = f29_mul2(x[0], x[1], x[2], x[3]);
    else if (op == 3) {
        const F29<T> a[4] = {x[0], x[2], x[4], x[6]}, b[4] = {x[1], x[3], x[5], x[7]};
        r = f29_dot4(a, b);
    } else if (op == 4) {
        r = f29_unpack_shl5<T>(x[0].v);
    } else if (op == 5) {
        r = f29_canon<4>(x[0]);
    } else if (op == 6) {
        alignas(16) u32 w[8];
        f29_store_product(w, x[0]);
        for (int j = 0; j < 8; ++j) r.v[j] = w[j];
        r.v[8] = 0;
    } else if (op == 7) {
        r = f29_canon_q(x[0], qtab);
    } else if (op == 8) {
        r = f29_mulc(x[0], x[1], x[2]);     // a, w, wq = floor(w * 2^261 / p)
    }
    f29_store_raw(out + i * 9, r);
}
extern "C" int pzp_f29_ops(int device, int field, int op, const uint32_t* in, uint32_t k, size_t count, uint32_t* out) {
    if (!in || !out || !count || k == 0 || k > 8 || op < 0 || op > 8 || (field != 0 && field != 1)) return -1;
    const size_t in_b = count * k * 36, out_b = count * 36;
    void* d = probe_buf(device, in_b + out_b);
    if (!d) return -1;
    if (hipMemcpy(d, in, in_b, hipMemcpyHostToDevice) != hipSuccess) return -1;
    u32* d_out = (u32*)((char*)d + in_b);
    const dim3 g((unsigned)((count + 63) / 64)), b(64);
    if (field == 0) hipLaunchKernelGGL(k_f29_ops<FqTag>, g, b, 0, 0, op, (const u32*)d, (unsigned)k, count, d_out);
    else hipLaunchKernelGGL(k_f29_ops<FrTag>, g, b, 0, 0, op, (const u32*)d, (unsigned)k, count, d_out);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(out, d_out, out_b, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}
