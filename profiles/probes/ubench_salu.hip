// scratch probe: can SALU bit-plane bookkeeping hide under v_mad_u64_u32 issue on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint64_t u64; typedef uint32_t u32;
#define MAD(acc, a, b) asm volatile("v_mad_u64_u32 %0, s[4:5], %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "s4", "s5")
#define SX(x, y) asm volatile("s_xor_b64 %0, %0, %1" : "+s"(x) : "s"(y) : "scc")
#define SA(x, y) asm volatile("s_and_b64 %0, %0, %1" : "+s"(x) : "s"(y) : "scc")

template <int MODE> __global__ void k(u64* out, unsigned iters, u64 seed) {
    u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
    u64 x0 = a, x1 = b, x2 = a ^ b, x3 = a + b;
    u64 s0 = seed, s1 = seed * 3, s2 = seed * 5, s3 = seed * 7, s4 = seed * 11, s5 = seed * 13, s6 = seed * 17, s7 = seed * 19;
    for (unsigned i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (MODE & 1) MAD(x0, a, b);
            if (MODE & 2) { SX(s0, s1); SA(s2, s3); SX(s4, s5); SA(s6, s7); if (MODE & 4) { SX(s1, s2); SX(s3, s4); } }
            if (MODE & 1) MAD(x1, a, b);
            if (MODE & 2) { SX(s1, s0); SA(s3, s2); SX(s5, s4); SA(s7, s6); if (MODE & 4) { SX(s5, s6); SX(s7, s0); } }
            if (MODE & 1) MAD(x2, a, b);
            if (MODE & 2) { SX(s0, s2); SA(s1, s3); SX(s4, s6); SA(s5, s7); if (MODE & 4) { SX(s2, s4); SX(s6, s1); } }
            if (MODE & 1) MAD(x3, a, b);
            if (MODE & 2) { SX(s2, s0); SA(s3, s1); SX(s6, s4); SA(s7, s5); if (MODE & 4) { SX(s4, s0); SX(s3, s7); } }
        }
    }
    u64 s = s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ s;
}

template <int MODE> double run(u64* d, unsigned blocks, unsigned iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10u, 0x1234567ull);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0x1234567ull);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    unsigned blocks = 256 * 16, iters = 4000; u64* d; hipMalloc(&d, (size_t)blocks * 256 * 8);
    double waves_per_simd = blocks * 4.0 / 1024.0;
    auto cyc = [&](double ms, double n_per_iter) { return ms * 1e-3 * 2.4e9 / (iters * n_per_iter * waves_per_simd); };
    double a = run<1>(d, blocks, iters), b = run<2>(d, blocks, iters), c = run<3>(d, blocks, iters), e = run<6>(d, blocks, iters), f = run<7>(d, blocks, iters);
    printf("mad only      : %.2f ms  %.2f cyc/mad (per SIMD, 2.4 GHz)\n", a, cyc(a, 8));
    printf("salu only (4/): %.2f ms  %.2f cyc/salu\n", b, cyc(b, 32));
    printf("mad + 4 salu  : %.2f ms  %.2f cyc/mad-group\n", c, cyc(c, 8));
    printf("salu only (6/): %.2f ms  %.2f cyc/salu\n", e, cyc(e, 48));
    printf("mad + 6 salu  : %.2f ms  %.2f cyc/mad-group\n", f, cyc(f, 8));
    return 0;
}
