// r03_occupancy_probe.hip -- MEASUREMENT PROBE (never linked into libpz_hip.so).
// How many waves per SIMD does the 29-bit Montgomery product need to keep v_mad_u64_u32 issuing back to back, and does a
// second, interleaved accumulator chain per lane (f29_mul2) buy that rate at lower occupancy?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I paillier_halo2_amd/csrc -I paillier_halo2_amd/csrc/probe \
//         profiles/probes/r03_occupancy_probe.hip -o gpurun_out/occ_probe && gpurun_out/occ_probe
// Occupancy is forced with dynamic LDS: 160 KiB / k bytes per 256-thread workgroup -> k workgroups per CU = k waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "fp29.cuh"
#include "fp29_dual_gen.cuh"

typedef F29<FqTag> Fq29;
extern __shared__ unsigned char smem[];

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ Fq29 seed(unsigned s) {
    Fq29 r;
    for (int i = 0; i < 9; ++i) r.v[i] = (s * 2654435761u + i * 40503u + 77u) & 0x0fffffffu;
    return r;
}

// one dependent chain of mads (multiplicands fixed, only the 64-bit addend chains): cycles per mad at a given occupancy
__global__ __launch_bounds__(256) void k_mad_chain(u64* out, unsigned iters, unsigned long long* cyc) {
    u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
    u64 x = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (unsigned i = 0; i < iters; ++i) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0"
                     : "+v"(x) : "v"(a), "v"(b) : "vcc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// two independent chains
__global__ __launch_bounds__(256) void k_mad_chain2(u64* out, unsigned iters, unsigned long long* cyc) {
    u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
    u64 x = a, y = b;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (unsigned i = 0; i < iters; ++i) {
        asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %1, vcc, %3, %3, %1\n\t"
                     "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %1, vcc, %3, %3, %1"
                     : "+v"(x), "+v"(y) : "v"(a), "v"(b) : "vcc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x ^ y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// MODE 0: one product chain per lane; 1: two chains, two f29_mul calls (the compiler's schedule); 2: two chains, f29_mul2
template <int MODE> __global__ __launch_bounds__(256) void k_mul_chain(u32* out, unsigned iters) {
    Fq29 x1 = seed(threadIdx.x), x2 = seed(threadIdx.x + 999u), y = seed(blockIdx.x + 31u);
    for (unsigned i = 0; i < iters; ++i) {
        if (MODE == 0) {
            x1 = f29_mul(x1, y);
            x1 = f29_mul(x1, y);
        } else if (MODE == 1) {
            x1 = f29_mul(x1, y);
            x2 = f29_mul(x2, y);
        } else {
            Fq29 r, s;
            f29_mul2(x1, y, x2, y, r, s);
            x1 = r;
            x2 = s;
        }
    }
    u32 o = 0;
    for (int i = 0; i < 9; ++i) o ^= x1.v[i] ^ x2.v[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = o;
}
// correctness of f29_mul2 against f29_mul (device-side compare; the field code itself is tested in tests/)
__global__ void k_check(unsigned* bad) {
    Fq29 a = seed(threadIdx.x), b = seed(threadIdx.x * 7u + 1u), c = seed(threadIdx.x + 123u), d = seed(threadIdx.x * 3u + 5u);
    Fq29 r, s;
    f29_mul2(a, b, c, d, r, s);
    const Fq29 r0 = f29_mul(a, b), s0 = f29_mul(c, d);
    for (int i = 0; i < 9; ++i)
        if (r.v[i] != r0.v[i] || s.v[i] != s0.v[i]) atomicAdd(bad, 1u);
}

template <class K, class... A> static float run(K kern, unsigned k, unsigned rounds, A... args) {
    const size_t lds = (163840 / k) & ~(size_t)255;
    CHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const unsigned blocks = 256 * k * rounds;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, args...);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, args...);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CHK(hipGetLastError());
    return best;
}

int main() {
    void *d_out, *d_cyc;
    unsigned* d_bad;
    CHK(hipMalloc(&d_out, (size_t)256 * 8 * 8 * 256 * 8));
    CHK(hipMalloc(&d_cyc, (size_t)256 * 8 * 8 * 8));
    CHK(hipMalloc(&d_bad, 4));
    CHK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, 0, d_bad);
    unsigned bad = 1;
    CHK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("f29_mul2 vs f29_mul mismatching limbs: %u\n", bad);
    const unsigned ks[] = {1, 2, 3, 4, 5, 6, 8};
    printf("# dependent v_mad_u64_u32 chains, cycles per mad per wave (s_memtime) and chip-wide T mads/s\n");
    for (unsigned k : ks) {
        const unsigned iters = 4096, rounds = 2;
        float ms1 = run(k_mad_chain, k, rounds, (u64*)d_out, iters, (unsigned long long*)d_cyc);
        unsigned long long c1 = 0;
        CHK(hipMemcpy(&c1, d_cyc, 8, hipMemcpyDeviceToHost));
        float ms2 = run(k_mad_chain2, k, rounds, (u64*)d_out, iters, (unsigned long long*)d_cyc);
        unsigned long long c2 = 0;
        CHK(hipMemcpy(&c2, d_cyc, 8, hipMemcpyDeviceToHost));
        const double n = 256.0 * k * rounds * 256 * iters * 8;
        printf("waves/SIMD %u: one chain %.2f memtime-ticks/mad, %.2f T/s | two chains %.2f ticks/mad, %.2f T/s\n", k,
               (double)c1 / (iters * 8.0), n / (ms1 * 1e-3) / 1e12, (double)c2 / (iters * 8.0), n / (ms2 * 1e-3) / 1e12);
    }
    printf("# f29_mul chains, G products/s chip-wide\n");
    for (unsigned k : ks) {
        const unsigned iters = 256, rounds = 4;
        const double n = 256.0 * k * rounds * 256 * iters * 2;
        float a = run(k_mul_chain<0>, k, rounds, (u32*)d_out, iters);
        float b = run(k_mul_chain<1>, k, rounds, (u32*)d_out, iters);
        float c = run(k_mul_chain<2>, k, rounds, (u32*)d_out, iters);
        printf("waves/SIMD %u: single chain %.1f | two chains (compiler) %.1f | two chains (f29_mul2 interleaved) %.1f\n", k,
               n / (a * 1e-3) / 1e9, n / (b * 1e-3) / 1e9, n / (c * 1e-3) / 1e9);
    }
    return 0;
}
