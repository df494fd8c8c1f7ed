// pz_probe_mfma.hip -- MEASUREMENT PROBE (libpz_probe.so, not the product ABI): the one lever rounds 1-5 never tried -- the idle MATRIX
// pipes.  Every product of K2 is by a constant known in advance (twiddles, coset powers), and in f29_mulc's Barrett form ALL THREE of its
// big products are by constants: a * wq (quotient estimate), a * w and q * p (remainder).  A [batch x digits] . Toeplitz(constant)
// contraction is what v_mfma_i32_16x16x64_i8 computes (I8 MFMA: 2 x the BF16 rate, /opt/skills/guides/MI355X_MICROARCH.md:435), so:
//
//   digits   a value below 2^261 as 38 unsigned 7-bit digits (i8 operands are signed: 0..127 is what fits); 64 = the instruction's K
//   pass 1   A = digits of 16 elements [16 x 64];  B1 = [64 x 80]: columns 0..37 Toeplitz(w) (the LOW 38 columns of a * w), columns
//            38..79 Toeplitz(wq') for the product columns 34..75 of a * wq', wq' = floor(w 2^266 / p)      -> 5 MFMAs per 16 elements
//   VALU     carry-propagate the 42 high column sums: q = floor(a wq' / 2^266) (short of the true quotient by a few units), as digits
//   pass 2   A = digits of q;  B2 = [64 x 48]: Toeplitz(p), the low 38 columns of q * p                    -> 3 MFMAs per 16 elements
//   VALU     r = (a w - q p) mod 2^266 by a signed carry propagation of the 38 column differences, back to 9 x 29-bit limbs
//
// i.e. 8 MFMAs (131 072 byte-MACs) per 16 field products against f29_mulc's 143 v_mad_u64_u32 per product on the VALU.  The probe
// measures, on this device: (0) f29_mulc itself, (1) the MATRIX side alone (8 MFMAs per iteration, operands in registers), (2) the VALU
// side alone (digit split + the two carry propagations + repack, no data movement between lanes), (3) the whole pipeline as a correct
// kernel (LDS-staged transposes, one wave per 16 elements) checked bit for bit against Python integers.  If (2) alone is slower than
// (0), no arrangement of (1) can win: the decision is a measurement, not an estimate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

typedef uint32_t u32;
typedef uint64_t u64;
typedef int v4i __attribute__((ext_vector_type(4)));

#define ND 38          // 7-bit digits of a value below 2^266
#define N1 80          // pass-1 columns: 38 (a w low) + 42 (a wq' columns 34..75)
#define N2 48          // pass-2 columns: 38 (q p low), padded to three tiles

// ---- (3) the whole pipeline, correct: one wave = 16 elements
// a: [count][9] strict 29-bit limbs (value below 2^261); b1: [5][64] v4i, b2: [3][64] v4i fragments (host-built, see pzp_mfma_tables);
// out: [count][9] strict limbs of a * w mod p, CANONICAL (below p); p29: the modulus' nine 29-bit limbs
__global__ __launch_bounds__(64) void k_mulc_mfma(const u32* __restrict__ a, size_t count, const v4i* __restrict__ b1, const v4i* __restrict__ b2,
                                                  const u32* __restrict__ p29, u32* __restrict__ out, unsigned iters) {
    __shared__ __attribute__((aligned(16))) unsigned char dig[16][64];
    __shared__ int sums[16][N1];
    const unsigned l = threadIdx.x, e = l & 15, g = l >> 4;
    const size_t base = (size_t)blockIdx.x * 16;
    u32 limbs[9];
    for (int i = 0; i < 9; ++i) limbs[i] = (l < 16 && base + e < count) ? a[(base + e) * 9 + i] : 0u;
    v4i B1[5], B2[3];
    for (int t = 0; t < 5; ++t) B1[t] = b1[t * 64 + l];
    for (int t = 0; t < 3; ++t) B2[t] = b2[t * 64 + l];
    for (unsigned it = 0; it < iters; ++it) {
        // digits of a
        if (l < 16) {
            for (int d = 0; d < 64; ++d) {
                u32 v = 0;
                if (d < ND) {
                    const unsigned bit = 7u * d, li = bit / 29u, sh = bit % 29u;
                    u64 w = limbs[li];
                    if (li + 1 < 9) w |= (u64)limbs[li + 1] << 29;
                    v = (u32)(w >> sh) & 127u;
                }
                dig[e][d] = (unsigned char)v;
            }
        }
        __syncthreads();
        v4i A = *reinterpret_cast<const v4i*>(&dig[e][16 * g]);
        for (int t = 0; t < 5; ++t) {
            v4i acc = {0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B1[t], acc, 0, 0, 0);
            for (int r = 0; r < 4; ++r) sums[4 * g + r][16 * t + e] = acc[r];   // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
        }
        __syncthreads();
        // q = floor(a wq' / 2^266): carries run up from product column 34; digits from column 38 on
        if (l < 16) {
            long long carry = 0;
            for (int c = 34; c < 76; ++c) {
                const long long t = (long long)sums[e][38 + (c - 34)] + carry;
                carry = t >> 7;
                if (c >= 38) dig[e][c - 38] = (unsigned char)(t & 127);
            }
            for (int d = ND; d < 64; ++d) dig[e][d] = 0;
        }
        __syncthreads();
        A = *reinterpret_cast<const v4i*>(&dig[e][16 * g]);
        int lowaw[ND];   // (lane < 16) the low columns of a w, saved before the sums are reused
        if (l < 16)
            for (int k = 0; k < ND; ++k) lowaw[k] = sums[e][k];
        __syncthreads();
        for (int t = 0; t < 3; ++t) {
            v4i acc = {0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B2[t], acc, 0, 0, 0);
            for (int r = 0; r < 4; ++r) sums[4 * g + r][16 * t + e] = acc[r];
        }
        __syncthreads();
        if (l < 16) {
            // r = (a w - q p) mod 2^266 as 38 digits, then 9 x 29-bit limbs
            long long carry = 0;
            u32 r[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < ND; ++k) {
                const long long t = (long long)lowaw[k] - sums[e][k] + carry;
                carry = t >> 7;
                const u32 d = (u32)(t & 127);
                const unsigned bit = 7u * k, li = bit / 29u, sh = bit % 29u;
                const u64 w = (u64)d << sh;
                r[li] |= (u32)w & 0x1fffffffu;
                r[li + 1] |= (u32)(w >> 29);
            }
            // r is below a few p: canonical by conditional subtractions of p
            for (int round = 0; round < 16; ++round) {
                int ge = 1;
                for (int i = 8; i >= 0; --i)
                    if (r[i] != p29[i]) {
                        ge = r[i] > p29[i];
                        break;
                    }
                if (!ge) break;
                long long br = 0;
                for (int i = 0; i < 9; ++i) {
                    long long t = (long long)r[i] - p29[i] + br;
                    br = t >> 29;
                    r[i] = (u32)t & 0x1fffffffu;
                }
            }
            for (int i = 0; i < 9; ++i) limbs[i] = r[i];   // feeds the next iteration (timing) and the output
        }
        __syncthreads();
    }
    if (l < 16 && base + e < count)
        for (int i = 0; i < 9; ++i) out[(base + e) * 9 + i] = limbs[i];
}

// ---- (1) the matrix side alone: 8 MFMAs per iteration per wave (= 16 field products' worth), operands in registers
__global__ __launch_bounds__(256) void k_ubench_mfma8(int* out, unsigned iters) {
    v4i A = {(int)threadIdx.x * 0x01010101, (int)blockIdx.x, 0x12345678, 0x0f0e0d0c};
    v4i Bq = {0x01020304, 0x05060708, (int)threadIdx.x, 0x7f7e7d7c};
    v4i acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = (v4i){t, t, t, t};
    for (unsigned i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, Bq, acc[t], 0, 0, 0);
    }
    int s = 0;
    for (int t = 0; t < 8; ++t) s ^= acc[t][0] ^ acc[t][1] ^ acc[t][2] ^ acc[t][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- (2) the VALU side alone, per lane and element: 9 limbs -> 38 digits packed four to a word (what an MFMA operand takes), the
// carry propagation of 42 column sums into q's digits (packed again), the signed propagation of 38 column differences into 9 limbs.
// The column sums are synthesised from the digits (one add each: far below what a transposing load would cost): this is a LOWER bound
// of the VALU work the MFMA formulation needs around the matrix instructions.
__global__ __launch_bounds__(256) void k_ubench_mfma_valu(u32* out, unsigned iters) {
    u32 limbs[9];
    for (int i = 0; i < 9; ++i) limbs[i] = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + i * 97u) & 0x1fffffffu;
    for (unsigned it = 0; it < iters; ++it) {
        u32 dg[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const unsigned bit = 7u * d, li = bit / 29u, sh = bit % 29u;
            u64 w = limbs[li];
            if (li + 1 < 9) w |= (u64)limbs[li + 1] << 29;
            dg[d] = (u32)(w >> sh) & 127u;
        }
        u32 packed[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            u32 v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (4 * j + b < ND) v |= dg[4 * j + b] << (8 * b);
            packed[j] = v;
        }
        // q's digits from 42 column sums (stand-ins: sums of neighbouring digits, at most 2^20 like the real ones)
        int carry = 0;
        u32 qd[ND];
#pragma unroll
        for (int c = 0; c < 42; ++c) {
            const int t = (int)(dg[c % ND] * 4001u + packed[c % 10] % 65521u) + carry;
            carry = t >> 7;
            if (c >= 4) qd[c - 4] = (u32)t & 127u;
        }
        u32 qpacked[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            u32 v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (4 * j + b < ND) v |= qd[4 * j + b] << (8 * b);
            qpacked[j] = v;
        }
        // r's limbs from 38 column differences
        int cr = 0;
        u32 r[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int t = (int)(dg[k] * 3001u) - (int)(qd[k] * 2999u + (qpacked[k % 10] & 1023u)) + cr;
            cr = t >> 7;
            const u32 d = (u32)t & 127u;
            const unsigned bit = 7u * k, li = bit / 29u, sh = bit % 29u;
            const u64 w = (u64)d << sh;
            r[li] |= (u32)w & 0x1fffffffu;
            r[li + 1] |= (u32)(w >> 29);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) limbs[i] = r[i];
    }
    u32 s = 0;
    for (int i = 0; i < 9; ++i) s ^= limbs[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static void* mfma_buf(int device, size_t bytes) {
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
template <class F> static int timed(int device, double* ms, F launch) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    launch();
    (void)hipEventRecord(e0, 0);
    launch();
    (void)hipEventRecord(e1, 0);
    int rc = hipEventSynchronize(e1) == hipSuccess ? 0 : -1;
    float f = 0;
    if (rc == 0 && hipEventElapsedTime(&f, e0, e1) != hipSuccess) rc = -1;
    *ms = f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? rc : -1;
}

// which: 1 = the matrix side alone (blocks x 4 waves x iters x 16 products), 2 = the VALU side alone (blocks x 256 lanes x iters products)
extern "C" int pzp_ubench_mfma(int device, int which, uint32_t blocks, uint32_t iters, double* ms) {
    void* d = mfma_buf(device, (size_t)blocks * 256 * 4);
    if (!d || !ms || !blocks || (which != 1 && which != 2)) return -1;
    if (which == 1) return timed(device, ms, [&] { hipLaunchKernelGGL(k_ubench_mfma8, dim3(blocks), dim3(256), 0, 0, (int*)d, (unsigned)iters); });
    return timed(device, ms, [&] { hipLaunchKernelGGL(k_ubench_mfma_valu, dim3(blocks), dim3(256), 0, 0, (u32*)d, (unsigned)iters); });
}

// the whole pipeline: a [count][9] strict limbs; wd / wqd / pd: the 38 seven-bit digits of w, of wq' = floor(w 2^266 / p), of p; p29: p's
// nine limbs.  out [count][9]: a * w mod p, canonical.  iters > 1 feeds the result back (timing: *ms of the second of two launches).
extern "C" int pzp_mulc_mfma(int device, const uint32_t* a, size_t count, const uint8_t wd[ND], const uint8_t wqd[ND], const uint8_t pd[ND],
                             const uint32_t p29[9], uint32_t* out, uint32_t iters, double* ms) {
    if (!a || !count || !wd || !wqd || !pd || !p29 || !out || !iters) return -1;
    // B fragments: lane l of tile t holds B[k = 16 (l >> 4) + j][col = 16 t + (l & 15)], j = 0..15, four bytes to a word
    unsigned char B1[5][64][16], B2[3][64][16];
    memset(B1, 0, sizeof B1);
    memset(B2, 0, sizeof B2);
    for (int t = 0; t < 5; ++t)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 16; ++j) {
                const int k = 16 * (l >> 4) + j, n = 16 * t + (l & 15);
                if (k >= ND) continue;
                if (n < 38) {
                    if (n - k >= 0 && n - k < ND) B1[t][l][j] = wd[n - k];
                } else {
                    const int c = n - 38 + 34;
                    if (c - k >= 0 && c - k < ND) B1[t][l][j] = wqd[c - k];
                }
            }
    for (int t = 0; t < 3; ++t)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 16; ++j) {
                const int k = 16 * (l >> 4) + j, n = 16 * t + (l & 15);
                if (k < ND && n < ND && n - k >= 0) B2[t][l][j] = pd[n - k];
            }
    const size_t ab = count * 36;
    char* d = (char*)mfma_buf(device, 2 * ab + sizeof B1 + sizeof B2 + 64);
    if (!d) return -1;
    char *d_a = d, *d_o = d + ab, *d_b1 = d + 2 * ab, *d_b2 = d_b1 + sizeof B1, *d_p = d_b2 + sizeof B2;
    if (hipMemcpy(d_a, a, ab, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_b1, B1, sizeof B1, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_b2, B2, sizeof B2, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_p, p29, 36, hipMemcpyHostToDevice) != hipSuccess)
        return -1;
    const dim3 g((unsigned)((count + 15) / 16)), b(64);
    double t_ms = 0;
    const int rc = timed(device, &t_ms, [&] {
        hipLaunchKernelGGL(k_mulc_mfma, g, b, 0, 0, (const u32*)d_a, count, (const v4i*)d_b1, (const v4i*)d_b2, (const u32*)d_p, (u32*)d_o, (unsigned)iters);
    });
    if (rc != 0) return rc;
    if (ms) *ms = t_ms;
    if (hipMemcpy(out, d_o, ab, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}
