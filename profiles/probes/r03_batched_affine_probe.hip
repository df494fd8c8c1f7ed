// r03_batched_affine_probe.hip -- MEASUREMENT PROBE: bucket accumulation by batched-affine additions against
// the shipped XYZZ mixed addition, on the same synthetic gather stream (random rows of a 134 MB window table, as k_msm_accumulate
// sees them for a 2^17-point SRS).
//   arm X  : acc (XYZZ) += row, one gather per addition: 8 products + 2 squares (ec29.cuh::x29_add_affine), the product kernel's loop
//   arm A0 : batched-affine with the inversion FREE (an upper bound on what the scheme can gain): per pair of rows
//            forward  d = x2 - x1, running product P *= d, prefix stored to LDS (36 B per pair and lane)
//            backward inv_j = Inv * prefix_{j-1}, Inv *= d_j, lambda = (y2 - y1) inv_j, x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1
//            = 3 + 2 products + 1 square per addition, both rows gathered twice (forward needs x only), result stored (64 B)
//            with M pairs per lane and batch (M = 4: 36 KiB of LDS per 256-thread workgroup, 4 workgroups per CU, the occupancy of arm X)
//   arm A1 : the same with a REAL inversion per lane and batch (Fermat, 254 squares + ~127 products per M additions)
// Exceptional cases (equal / opposite points, identities) are not handled in arms A*: they would only add work.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I paillier_halo2_amd/csrc profiles/probes/r03_batched_affine_probe.hip -o scratch/ba_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "ec29.cuh"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned rnd(unsigned s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__global__ void k_fill_table(G1Aff64* t, size_t n) {   // random canonical-looking coordinates (values < 2^252): arithmetic cost only
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned s = (unsigned)i * 2654435761u + 1u;
    for (int k = 0; k < 16; ++k) { s = rnd(s); t[i].w[k] = (k == 7 || k == 15) ? (s & 0x0fffffffu) : s; }
}

__global__ __launch_bounds__(256) void k_arm_xyzz(const G1Aff64* __restrict__ table, unsigned n_rows, unsigned adds, G1X29Raw* out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s = gid * 747796405u + 12345u;
    G1X29 acc = x29_inf();
    for (unsigned k = 0; k < adds; ++k) {
        s = rnd(s);
        G1A29 q = a29_load64(table + (s % n_rows));
        x29_add_affine(acc, q);
    }
    x29_store_raw(out + gid, acc);
}

template <int M, bool REAL_INV>
__global__ __launch_bounds__(256) void k_arm_affine(const G1Aff64* __restrict__ table, unsigned n_rows, unsigned adds, G1Aff64* out) {
    __shared__ u32 s_pre[M][9][256];   // prefix products, limb-major: conflict-free
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, t = threadIdx.x;
    unsigned s0 = gid * 747796405u + 12345u;
    Fq29 chk = f29_zero<FqTag>();
    for (unsigned base = 0; base < adds; base += M) {
        // forward: d_j = x2 - x1, prefix_j = d_0 ... d_j
        unsigned s = s0;
        Fq29 P = f29_one<FqTag>();
        for (int j = 0; j < M; ++j) {
            s = rnd(s); const unsigned r1 = s % n_rows;
            s = rnd(s); const unsigned r2 = s % n_rows;
            const Fq29 x1 = f29_load<FqTag>(table + r1), x2 = f29_load<FqTag>(table + r2);
            const Fq29 d = f29_carry(f29_sub<2, 29>(x2, x1));
            for (int i = 0; i < 9; ++i) s_pre[j][i][t] = P.v[i];     // prefix BEFORE d_j
            P = f29_mul(P, d);
        }
        Fq29 Inv = REAL_INV ? f29_inv(P) : P;      // A0: pretend the inverse is free
        // backward
        unsigned sb[2 * M];
        s = s0;
        for (int j = 0; j < 2 * M; ++j) { s = rnd(s); sb[j] = s % n_rows; }
        s0 = s;
        for (int j = M - 1; j >= 0; --j) {
            const G1A29 p1 = a29_load64(table + sb[2 * j]), p2 = a29_load64(table + sb[2 * j + 1]);
            Fq29 pre;
            for (int i = 0; i < 9; ++i) pre.v[i] = s_pre[j][i][t];
            const Fq29 d = f29_carry(f29_sub<2, 29>(p2.x, p1.x));
            const Fq29 inv = f29_mul(Inv, pre);
            Inv = f29_mul(Inv, d);
            const Fq29 lam = f29_mul(f29_carry(f29_sub<2, 29>(p2.y, p1.y)), inv);
            const Fq29 x3 = f29_carry(f29_sub<4, 30>(f29_sqr(lam), f29_add(p1.x, p2.x)));
            const Fq29 y3 = f29_sub<2, 29>(f29_mul(lam, f29_carry(f29_sub<8, 30>(p1.x, x3))), p1.y);
            G1A29 r;
            r.x = x3;
            r.y = f29_carry(y3);
            // the next tree level reads it back: stored as a 64-byte row (canonicalised like a table row)
            a29_store64(out + (size_t)gid * M + j, r);
            chk = f29_add(chk, r.x);
            chk = f29_carry(chk);
        }
    }
    if (chk.v[0] == 0xffffffffu) out[0].w[0] = chk.v[1];
}

template <class K, class... A> static float run(K kern, dim3 g, A... args) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, g, dim3(256), 0, 0, args...);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, g, dim3(256), 0, 0, args...);
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
    const unsigned n_rows = 16u << 17;          // the window table of a 2^17-point SRS: 16 windows x 2^17 rows x 64 B = 134 MB
    const unsigned lanes = 256 * 1024 * 4;      // 4 waves per SIMD, as k_msm_accumulate runs
    const unsigned adds = 64;                   // additions per lane
    G1Aff64* table;
    void* out;
    CHK(hipMalloc(&table, (size_t)n_rows * 64));
    CHK(hipMalloc(&out, (size_t)lanes * 4 * 144));
    hipLaunchKernelGGL(k_fill_table, dim3(n_rows / 256), dim3(256), 0, 0, table, (size_t)n_rows);
    const double total = (double)lanes * adds;
    const float x = run(k_arm_xyzz, dim3(lanes / 256), (const G1Aff64*)table, n_rows, adds, (G1X29Raw*)out);
    const float a0 = run(k_arm_affine<4, false>, dim3(lanes / 256), (const G1Aff64*)table, n_rows, adds, (G1Aff64*)out);
    const float a1 = run(k_arm_affine<4, true>, dim3(lanes / 256), (const G1Aff64*)table, n_rows, adds, (G1Aff64*)out);
    printf("additions per launch %.0f\n", total);
    printf("arm X  (XYZZ mixed addition, shipped)              %8.3f ms  %6.2f ns per 1000 additions\n", x, x * 1e6 / total * 1000);
    printf("arm A0 (batched affine, M = 4, inversion FREE)     %8.3f ms  %6.2f ns per 1000 additions  (%.2f x arm X)\n", a0, a0 * 1e6 / total * 1000, a0 / x);
    printf("arm A1 (batched affine, M = 4, Fermat per lane)    %8.3f ms  %6.2f ns per 1000 additions  (%.2f x arm X)\n", a1, a1 * 1e6 / total * 1000, a1 / x);
    return 0;
}
