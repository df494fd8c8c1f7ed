// pz_core.hip -- context, workspaces, cached power tables, timing and the two issue-rate
// microbenchmarks of libpz_hip.so.  gfx950 only.
#define PZ_FP_MUL_VARIANTS 1
#include "fp.cuh"
#include "fp29_probe.cuh"
#include "fp29.cuh"
#include "pz_internal.h"

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// table[i] = base^i (Montgomery).  Thread t fills entries [t*CH, (t+1)*CH): start by
// square-and-multiply over the bits of t*CH, then CH-1 successive products.
__global__ void k_pow_table(Fr base, Fr init, Fr* table, size_t n, unsigned ch) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = t * ch;
    if (lo >= n) return;
    Fr acc = init;
    Fr sq = base;
    size_t e = lo;
    while (e) {
        if (e & 1) acc = fp_mul(acc, sq);
        sq = fp_sqr(sq);
        e >>= 1;
    }
    size_t hi = lo + ch < n ? lo + ch : n;
    for (size_t i = lo; i < hi; ++i) {
        fp_store(table + i, acc);
        acc = fp_mul(acc, base);
    }
}

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

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
extern "C" int pz_abi_version(void) { return 1; }

extern "C" const char* pz_strerror(int s) {
    switch (s) {
        case PZ_OK: return "ok";
        case PZ_ERR_INVALID: return "invalid argument";
        case PZ_ERR_HIP: return "HIP runtime error";
        case PZ_ERR_NO_DEVICE: return "no gfx950 device";
        case PZ_ERR_OOM: return "device out of memory";
        case PZ_ERR_ZERO_MODULUS: return "modulus is zero";
        case PZ_ERR_RANGE: return "quotient does not fit the assigned limb count";
        case PZ_ERR_UNSUPPORTED: return "unsupported configuration";
        case PZ_ERR_CAPACITY: return "output capacity too small";
        default: return "unknown pz_status";
    }
}

extern "C" const char* pz_last_hip_error(const pz_ctx* ctx) { return ctx ? ctx->hip_err : ""; }

extern "C" int pz_init(int n_devices, const int* device_ids, pz_ctx** out) {
    if (!out) return PZ_ERR_INVALID;
    *out = nullptr;
    if (n_devices != 1) return PZ_ERR_UNSUPPORTED;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PZ_ERR_NO_DEVICE;
    int dev = device_ids ? device_ids[0] : 0;
    if (dev < 0 || dev >= count) return PZ_ERR_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return PZ_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PZ_ERR_NO_DEVICE;  // code objects are gfx950 only
    pz_ctx* ctx = new pz_ctx();
    ctx->device = dev;
    ctx->cu_count = prop.multiProcessorCount;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return PZ_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return PZ_OK;
}

extern "C" int pz_free(pz_ctx* ctx) {
    if (!ctx) return PZ_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& w : ctx->ws)
        if (w.d) (void)hipFree(w.d);
    for (auto& t : ctx->pow_tables)
        if (t.d) (void)hipFree(t.d);
    for (auto& t : ctx->ext_tables)
        if (t.d) (void)hipFree(t.d);
    for (auto& v : ctx->ev)
        for (auto& p : v) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    for (auto& e : ctx->io_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->io_h2d) (void)hipStreamDestroy(ctx->io_h2d);
    if (ctx->io_d2h) (void)hipStreamDestroy(ctx->io_d2h);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return PZ_OK;
}

extern "C" int pz_set_stream(pz_ctx* ctx, void* s) {
    if (!ctx) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    hipStream_t ns = s ? (hipStream_t)s : ctx->own_stream;
    if (ns != ctx->stream) {
        // cached tables and workspaces may still be written / read by work queued on the old stream: order the
        // new stream after it (an event wait, no host synchronisation)
        hipEvent_t ev;
        HIPCHK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ns, ev, 0);
        (void)hipEventDestroy(ev);
        if (e != hipSuccess) return pz_hip_fail(ctx, e, "pz_set_stream: order the new stream after the old one");
        ctx->stream = ns;
    }
    return PZ_OK;
}

int pz_io_init(pz_ctx* ctx) {
    if (!ctx->io_h2d) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->io_h2d, hipStreamNonBlocking));
    if (!ctx->io_d2h) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->io_d2h, hipStreamNonBlocking));
    for (auto& e : ctx->io_ev)
        if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // the staging buffers may still be in use by work queued on the caller's stream (an earlier call): uploads start after it
    HIPCHK(ctx, hipEventRecord(ctx->io_ev[PZ_IO_EVENTS - 1], ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->io_h2d, ctx->io_ev[PZ_IO_EVENTS - 1], 0));
    return PZ_OK;
}

extern "C" int pz_sync(pz_ctx* ctx) {
    if (!ctx) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PZ_OK;
}

int pz_ws_get(pz_ctx* ctx, int slot, size_t bytes, void** out) {
    pz_wsbuf& w = ctx->ws[slot];
    if (w.cap < bytes) {
        // in-flight kernels may still use the old buffer
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (w.d) HIPCHK(ctx, hipFree(w.d));
        w.d = nullptr;
        w.cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIPCHK(ctx, hipMalloc(&w.d, want));
        w.cap = want;
    }
    *out = w.d;
    return PZ_OK;
}

int pz_get_pow_table(pz_ctx* ctx, const uint64_t base[4], size_t n, void** d_out, const uint64_t* init) {
    // Montgomery one (R mod r): the default first entry
    static const uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    if (!init) init = ONE;
    for (auto& t : ctx->pow_tables)
        if (t.n >= n && memcmp(t.base, base, 32) == 0 && memcmp(t.init, init, 32) == 0) {
            t.stamp = ++ctx->pow_clock;
            *d_out = t.d;
            return PZ_OK;
        }
    if (ctx->pow_tables.size() >= 48) {  // evict the least recently used table (kernels still reading it: drain first)
        size_t lru = 0;
        for (size_t k = 1; k < ctx->pow_tables.size(); ++k)
            if (ctx->pow_tables[k].stamp < ctx->pow_tables[lru].stamp) lru = k;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(ctx->pow_tables[lru].d));
        ctx->pow_tables.erase(ctx->pow_tables.begin() + lru);
    }
    pz_pow_table t;
    t.stamp = ++ctx->pow_clock;
    memcpy(t.base, base, 32);
    memcpy(t.init, init, 32);
    t.n = n;
    HIPCHK(ctx, hipMalloc(&t.d, n * 32));
    Fr b, i0;
    memcpy(b.v, base, 32);
    memcpy(i0.v, init, 32);
    const unsigned ch = 64;
    size_t threads = (n + ch - 1) / ch;
    hipLaunchKernelGGL(k_pow_table, dim3(pz_div_up(threads, 128)), dim3(128), 0, ctx->stream, b, i0, (Fr*)t.d, n, ch);
    HIPCHK(ctx, hipGetLastError());
    ctx->pow_tables.push_back(t);
    *d_out = t.d;
    return PZ_OK;
}

// ---- timing ------------------------------------------------------------------------------------
pz_timer::pz_timer(pz_ctx* c, int cls_) : ctx(c), cls(cls_), on(c->timing), idx(0) {
    if (!on) return;
    auto& v = ctx->ev[cls];
    if (ctx->ev_used[cls] == v.size()) {
        pz_event_pair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) {
            on = false;
            return;
        }
        v.push_back(p);
    }
    idx = ctx->ev_used[cls]++;
    (void)hipEventRecord(v[idx].a, ctx->stream);
}
pz_timer::~pz_timer() {
    if (on) (void)hipEventRecord(ctx->ev[cls][idx].b, ctx->stream);
}

static void pz_timing_collect(pz_ctx* ctx) {
    for (int c = 0; c < PZ_T_COUNT; ++c) {
        for (size_t i = 0; i < ctx->ev_used[c]; ++i) {
            float ms = 0;
            if (hipEventSynchronize(ctx->ev[c][i].b) == hipSuccess &&
                hipEventElapsedTime(&ms, ctx->ev[c][i].a, ctx->ev[c][i].b) == hipSuccess) {
                ctx->ev_ms[c] += ms;
                ctx->ev_n[c] += 1;
            }
        }
        ctx->ev_used[c] = 0;
    }
}

extern "C" int pz_timing_enable(pz_ctx* ctx, int on) {
    if (!ctx) return PZ_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!on && ctx->timing) pz_timing_collect(ctx);
    ctx->timing = on != 0;
    return PZ_OK;
}
extern "C" int pz_timing_reset(pz_ctx* ctx) {
    if (!ctx) return PZ_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    pz_timing_collect(ctx);
    for (int c = 0; c < PZ_T_COUNT; ++c) {
        ctx->ev_ms[c] = 0;
        ctx->ev_n[c] = 0;
    }
    return PZ_OK;
}
extern "C" int pz_timing_get(pz_ctx* ctx, int which, double* total_ms, uint64_t* launches) {
    if (!ctx || which < 0 || which >= PZ_T_COUNT) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    pz_timing_collect(ctx);
    if (total_ms) *total_ms = ctx->ev_ms[which];
    if (launches) *launches = ctx->ev_n[which];
    return PZ_OK;
}

// ---- microbenchmarks ---------------------------------------------------------------------------
template <class K, class... A>
static int timed_launch(pz_ctx* ctx, double* ms, K kern, dim3 g, dim3 b, A... args) {
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, g, b, 0, ctx->stream, args...);  // warm
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(kern, g, b, 0, ctx->stream, args...);
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(e1));
    float f = 0;
    HIPCHK(ctx, hipEventElapsedTime(&f, e0, e1));
    *ms = f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PZ_OK;
}

extern "C" int pz_ubench_mad(pz_ctx* ctx, uint32_t blocks, uint32_t iters, double* ms) {
    if (!ctx || !ms || !blocks) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    void* d;
    PZCHK(pz_ws_get(ctx, WS_MISC, (size_t)blocks * 256 * 32, &d));
    return timed_launch(ctx, ms, k_ubench_mad, dim3(blocks), dim3(256), (u64*)d, (unsigned)iters);
}
extern "C" int pz_ubench_mad_indep(pz_ctx* ctx, uint32_t blocks, uint32_t iters, double* ms) {
    if (!ctx || !ms || !blocks) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    void* d;
    PZCHK(pz_ws_get(ctx, WS_MISC, (size_t)blocks * 256 * 32, &d));
    return timed_launch(ctx, ms, k_ubench_mad_indep, dim3(blocks), dim3(256), (u64*)d, (unsigned)iters);
}
extern "C" int pz_ubench_fqmul_variant(pz_ctx* ctx, int variant, uint32_t blocks, uint32_t iters, double* ms) {
    if (!ctx || !ms || !blocks || variant < 0 || variant > 4) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    void* d;
    PZCHK(pz_ws_get(ctx, WS_MISC, (size_t)blocks * 256 * 32, &d));
    if (variant == 1) return timed_launch(ctx, ms, k_ubench_fqmul_nowait, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
    if (variant == 3) return timed_launch(ctx, ms, k_ubench_f29<0>, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
    if (variant == 4) return timed_launch(ctx, ms, k_ubench_f29<1>, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
    if (variant == 2) return timed_launch(ctx, ms, k_ubench_fqmul29, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
    return timed_launch(ctx, ms, k_ubench_fqmul, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
}
extern "C" int pz_fq_mul29(pz_ctx* ctx, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    if (!ctx || !a || !b || !out) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    void* d;
    PZCHK(pz_ws_get(ctx, WS_MISC, 96, &d));
    HIPCHK(ctx, hipMemcpyAsync(d, a, 32, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync((char*)d + 32, b, 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_fq_mul29, dim3(1), dim3(1), 0, ctx->stream, (const u32*)d, (const u32*)d + 8, (u32*)d + 16);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, (char*)d + 64, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PZ_OK;
}
extern "C" int pz_ubench_fqmul(pz_ctx* ctx, uint32_t blocks, uint32_t iters, double* ms) {
    if (!ctx || !ms || !blocks) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    void* d;
    PZCHK(pz_ws_get(ctx, WS_MISC, (size_t)blocks * 256 * 32, &d));
    return timed_launch(ctx, ms, k_ubench_fqmul, dim3(blocks), dim3(256), (Fq*)d, (unsigned)iters);
}
