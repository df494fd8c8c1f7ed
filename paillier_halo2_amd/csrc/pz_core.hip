// pz_core.hip -- context, device memory, workspaces, cached power tables and timing of libpz_hip.so.  gfx950 only.
// (the issue-rate microbenchmarks live in probe/pz_probe.hip -> libpz_probe.so, outside the product ABI)
#include "fp.cuh"
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

// Montgomery table entries -> the constant pairs the K2 kernels multiply by (f29_mulc, fp29.cuh): 18 words per entry, the plain value
// c and cq = floor(c * 2^261 / p) as 9 x 29-bit limbs each -- no unpack per use, and the product by a known constant costs 143
// multiplier instructions instead of a Montgomery product's 180
__global__ void k_raw29(const Fr* __restrict__ src, u32* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 o[18];
    f29_cpair_from_mont(fp_load<FrTag>(src + i), o);
#pragma unroll
    for (int j = 0; j < 18; ++j) dst[18 * i + j] = o[j];
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
// explicit ABI version: bumped whenever an entry point of include/pz.h is added, removed or changes meaning (1 = rounds 1-2;
// 3 = round 3: device-memory entry points added, measurement probes moved out to libpz_probe.so; 4 = round 4: PZ_ERR_ASYNC,
// pz_msm_g1_multi; 5 = round 5: PZ_ERR_INTERNAL, the break-point column layout and the connected-proof entry points)
extern "C" int pz_abi_version(void) { return PZ_ABI_VERSION; }

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
        case PZ_ERR_MESSAGE_RANGE: return "message does not fit the exponent bits of the uniform-shape circuit";
        case PZ_ERR_ASYNC: return "an asynchronous call found its inputs changed while it ran (or an internal invariant broken); its results are invalid";
        case PZ_ERR_INTERNAL: return "a bounded device-side wait ran out; the call's outputs are invalid, the context is intact";
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

static int arena_release(pz_ctx* ctx);
extern "C" int pz_free(pz_ctx* ctx) {
    if (!ctx) return PZ_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    pz_dev_cache_trim(ctx, 0);
    for (auto& w : ctx->ws)
        if (w.d) (void)pz_hip_free(w.d);
    for (auto& t : ctx->pow_tables) {
        if (t.d) (void)pz_hip_free(t.d);
        if (t.d_raw) (void)pz_hip_free(t.d_raw);
    }
    for (auto& t : ctx->ext_tables)
        if (t.d) (void)pz_hip_free(t.d);
    for (auto& v : ctx->ev)
        for (auto& p : v) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    for (auto& e : ctx->io_ev)
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 4; ++k) {
        if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
        if (ctx->stage_h[k]) (void)hipHostFree(ctx->stage_h[k]);
    }
    if (ctx->async_err_h) (void)hipHostFree((void*)ctx->async_err_h);
    if (ctx->io_h2d) (void)hipStreamDestroy(ctx->io_h2d);
    if (ctx->io_d2h) (void)hipStreamDestroy(ctx->io_d2h);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    (void)arena_release(ctx);   // (an arena other objects still hold blocks of -- bases tables, a proving key -- stays reserved)
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


// ---- pz_dev_arena: a sub-allocator over ONE reserved block --------------------------------------------------------------------
// Why: a caller that follows the reference's flow (src/paillier.rs:50-55 bakes the message into the circuit, so every message is a new key)
// builds and drops 120-250 GB of key, workspace and witness per proof.  Through the driver that is seconds per step (measured at config c5,
// prove_connected --fresh: 5.6 of 11.4 s per step in hipMalloc / hipFree once the block cache has to be released for the next phase's
// temporaries); carved out of one block it is map operations.  Address-ordered hole list, coalescing on free; large requests take the best
// fitting hole from its start, small ones (< 64 MiB: temporaries, tables) the highest hole from its end, so that short-lived small blocks
// do not pin holes between the long-lived large ones.  Blocks are 4 KiB-granular.  Memory is NOT zeroed (hipMalloc does not promise it
// either); PZ_DEV_ARENA_POISON=1 fills the block with 0xA5 at creation and every block again when it is freed -- the test suite runs
// under it to prove that nothing in the library depends on fresh pages being zero.
struct pz_arena {
    char* base = nullptr;
    size_t size = 0;
    int device = 0;
    bool poison = false;
    bool orphan = false;   // detached from its context while blocks were live: released when the last of them is freed
    std::mutex mu;
    std::map<size_t, size_t> holes;   // offset -> length
    std::map<size_t, size_t> live;    // offset -> length
    size_t used = 0, peak = 0, served = 0, missed = 0;
};
static std::mutex g_arena_mu;
static std::vector<pz_arena*> g_arenas;   // every live arena of the process: a block may be freed through another context than its own
#define PZ_ARENA_GRAIN ((size_t)4096)
#define PZ_ARENA_SMALL ((size_t)64 << 20)

static void* arena_take(pz_arena* a, size_t bytes) {
    const size_t need = (bytes + PZ_ARENA_GRAIN - 1) / PZ_ARENA_GRAIN * PZ_ARENA_GRAIN;
    std::lock_guard<std::mutex> lk(a->mu);
    size_t off = 0;
    bool found = false;
    if (need < PZ_ARENA_SMALL) {
        for (auto it = a->holes.rbegin(); it != a->holes.rend(); ++it)
            if (it->second >= need) {
                const size_t h_off = it->first, h_len = it->second;
                off = h_off + h_len - need;
                a->holes.erase(h_off);
                if (h_len > need) a->holes[h_off] = h_len - need;
                found = true;
                break;
            }
    } else {
        size_t best_off = 0, best_len = ~(size_t)0;
        for (auto& h : a->holes)
            if (h.second >= need && h.second < best_len) { best_off = h.first; best_len = h.second; }
        if (best_len != ~(size_t)0) {
            a->holes.erase(best_off);
            if (best_len > need) a->holes[best_off + need] = best_len - need;
            off = best_off;
            found = true;
        }
    }
    if (!found) { ++a->missed; return nullptr; }
    a->live[off] = need;
    a->used += need;
    if (a->used > a->peak) a->peak = a->used;
    ++a->served;
    return a->base + off;
}
// -> the arena `d` lies in (registered), or nullptr
static pz_arena* arena_of(const void* d) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    for (pz_arena* a : g_arenas)
        if ((const char*)d >= a->base && (const char*)d < a->base + a->size) return a;
    return nullptr;
}
static hipError_t arena_give(pz_arena* a, void* d) {
    size_t off = (size_t)((char*)d - a->base), len = 0;
    {
        std::lock_guard<std::mutex> lk(a->mu);
        auto it = a->live.find(off);
        if (it == a->live.end()) return hipErrorInvalidValue;   // not the start of a live block (double free)
        len = it->second;
    }
    // hipFree's semantics, which the library's callers rely on: nothing queued on the device still uses the block afterwards
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != a->device) (void)hipSetDevice(a->device);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && a->poison) {
        e = hipMemset(d, 0xA5, len);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (cur != a->device) (void)hipSetDevice(cur);
    if (e != hipSuccess) return e;
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(a->mu);
        if (a->live.erase(off) != 1) return hipErrorInvalidValue;   // two threads freed the same block: the second one loses
        a->used -= len;
        last = a->orphan && a->live.empty();
        auto nx = a->holes.lower_bound(off);
        if (nx != a->holes.end() && off + len == nx->first) {   // merge with the hole above
            len += nx->second;
            nx = a->holes.erase(nx);
        }
        bool merged = false;
        if (nx != a->holes.begin()) {                            // ... and below
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) {
                pv->second += len;
                merged = true;
            }
        }
        if (!merged) a->holes[off] = len;
    }
    if (last) {   // nobody can allocate from an orphan: this was its last user
        {
            std::lock_guard<std::mutex> lk(g_arena_mu);
            for (size_t i = 0; i < g_arenas.size(); ++i)
                if (g_arenas[i] == a) { g_arenas.erase(g_arenas.begin() + (long)i); break; }
        }
        (void)hipFree(a->base);
        delete a;
    }
    return hipSuccess;
}
hipError_t pz_hip_free(void* d) {
    if (!d) return hipSuccess;
    if (pz_arena* a = arena_of(d)) return arena_give(a, d);
    return hipFree(d);
}
size_t pz_mem_free_bytes(pz_ctx* ctx) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    if (ctx && ctx->arena) {
        std::lock_guard<std::mutex> lk(ctx->arena->mu);
        free_b += ctx->arena->size - ctx->arena->used;
    }
    return free_b;
}
// releases the context's arena if nothing lives in it; an arena with live blocks stays registered (their owners free them later through
// pz_hip_free) and is only detached from the context
static int arena_release(pz_ctx* ctx) {
    pz_arena* a = ctx->arena;
    if (!a) return PZ_OK;
    ctx->arena = nullptr;
    {
        std::lock_guard<std::mutex> lk(a->mu);
        if (!a->live.empty()) {
            a->orphan = true;
            return PZ_ERR_INVALID;
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        for (size_t i = 0; i < g_arenas.size(); ++i)
            if (g_arenas[i] == a) { g_arenas.erase(g_arenas.begin() + (long)i); break; }
    }
    (void)hipFree(a->base);
    delete a;
    return PZ_OK;
}
extern "C" int pz_dev_arena(pz_ctx* ctx, size_t bytes) {
    if (!ctx) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->arena) {
        const int rc = arena_release(ctx);
        if (rc != PZ_OK) {
            snprintf(ctx->hip_err, sizeof ctx->hip_err, "pz_dev_arena: blocks of the previous arena are still live; it stays reserved until they are freed");
            return rc;
        }
    }
    if (!bytes) return PZ_OK;
    pz_dev_cache_trim(ctx, 0);   // cached blocks are driver allocations the arena is about to replace
    pz_arena* a = new pz_arena();
    a->size = bytes / PZ_ARENA_GRAIN * PZ_ARENA_GRAIN;
    a->device = ctx->device;
    const char* po = getenv("PZ_DEV_ARENA_POISON");
    a->poison = po && po[0] == '1';
    void* base = nullptr;
    hipError_t e = a->size ? hipMalloc(&base, a->size) : hipErrorInvalidValue;
    if (e == hipSuccess && a->poison) {
        e = hipMemset(base, 0xA5, a->size);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) (void)hipFree(base);
    }
    if (e != hipSuccess) {
        delete a;
        return pz_hip_fail(ctx, e, "pz_dev_arena: hipMalloc of the arena");
    }
    a->base = (char*)base;
    a->holes[0] = a->size;
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        g_arenas.push_back(a);
    }
    ctx->arena = a;
    ctx->mem_avail = 0;
    return PZ_OK;
}
extern "C" int pz_dev_mem_info(pz_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
    if (!ctx) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    size_t f = 0, t = 0;
    HIPCHK(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return PZ_OK;
}
extern "C" int pz_dev_arena_info(pz_ctx* ctx, uint64_t out[6]) {
    if (!ctx || !out) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    memset(out, 0, 48);
    if (!ctx->arena) return PZ_OK;
    pz_arena* a = ctx->arena;
    std::lock_guard<std::mutex> lk(a->mu);
    size_t largest = 0;
    for (auto& h : a->holes)
        if (h.second > largest) largest = h.second;
    out[0] = a->size; out[1] = a->used; out[2] = a->peak; out[3] = largest; out[4] = a->served; out[5] = a->missed;
    return PZ_OK;
}

// ---- device memory for hosts that own no HIP runtime of their own (a Rust prover behind the FFI; tests/cpp) ------------
// Plain hipMalloc'd buffers; the `_dev` entry points take them as they are.  Transfers are ordered on the context's stream:
// pz_upload returns once the host buffer may be reused, pz_download once the data has arrived.
#define PZ_DEV_CACHE_MIN ((size_t)32 << 20)
void pz_dev_cache_trim(pz_ctx* ctx, size_t keep_bytes) {   // largest blocks go first
    while (ctx->dev_cache_bytes > keep_bytes && !ctx->dev_cache.empty()) {
        size_t big = 0;
        for (size_t i = 1; i < ctx->dev_cache.size(); ++i)
            if (ctx->dev_cache[i].second > ctx->dev_cache[big].second) big = i;
        (void)pz_hip_free(ctx->dev_cache[big].first);
        ctx->dev_cache_bytes -= ctx->dev_cache[big].second;
        ctx->dev_cache.erase(ctx->dev_cache.begin() + (long)big);
    }
}
hipError_t pz_hip_malloc(pz_ctx* ctx, void** d, size_t bytes) {
    if (ctx && ctx->arena && bytes) {
        if (void* p = arena_take(ctx->arena, bytes)) {
            *d = p;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(d, bytes);
    if (e == hipErrorOutOfMemory && ctx && !ctx->dev_cache.empty()) {
        (void)hipGetLastError();
        pz_dev_cache_trim(ctx, 0);
        e = hipMalloc(d, bytes);
    }
    return e;
}
extern "C" int pz_dev_cache_limit(pz_ctx* ctx, size_t max_bytes) {
    if (!ctx) return PZ_ERR_INVALID;
    PZ_ENTER(ctx);
    ctx->dev_cache_limit = max_bytes;
    if (ctx->dev_cache_bytes > max_bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        pz_dev_cache_trim(ctx, max_bytes);
    }
    return PZ_OK;
}
// The MSM's grow-only workspaces (up to PZ_MSM_WS_GIB = 48 GiB, sized when memory was plentiful) are a CACHE: when a caller's allocation
// does not fit beside them they go back, and the next pz_msm_g1* call sizes its column groups for what is free then.  Only the slots whose
// contents no entry point expects to survive its own return (the sort / partial-sum / tree buffers; NOT the SHPLONK state's or K3's).
static hipError_t ws_release_msm(pz_ctx* ctx) {
    static const int slots[] = {WS_HIST, WS_OFFS, WS_CURSOR, WS_ITEMS, WS_ENTRIES, WS_PARTIALS, WS_NODES_A, WS_NODES_B, WS_TOTALS, WS_ORDER};
    hipError_t e = hipStreamSynchronize(ctx->stream);   // queued kernels of this context may still use them
    if (e != hipSuccess) return e;
    for (int sl : slots) {
        pz_wsbuf& w = ctx->ws[sl];
        if (w.d) (void)pz_hip_free(w.d);
        w.d = nullptr;
        w.cap = 0;
    }
    ctx->mem_avail = 0;
    return hipSuccess;
}
static bool ws_msm_held(const pz_ctx* ctx) {
    return ctx->ws[WS_ENTRIES].cap || ctx->ws[WS_PARTIALS].cap || ctx->ws[WS_HIST].cap || ctx->ws[WS_ITEMS].cap;
}
extern "C" int pz_dev_alloc(pz_ctx* ctx, size_t bytes, void** d_out) {
    if (!ctx || !d_out) return PZ_ERR_INVALID;
    *d_out = nullptr;
    if (!bytes) return PZ_OK;
    PZ_ENTER(ctx);
    if (ctx->arena) {   // from the arena, else from the driver; the block cache is for contexts without one
        hipError_t e = pz_hip_malloc(ctx, d_out, bytes);
        if (e == hipErrorOutOfMemory && ws_msm_held(ctx)) {
            (void)hipGetLastError();
            HIPCHK(ctx, ws_release_msm(ctx));
            e = pz_hip_malloc(ctx, d_out, bytes);
        }
        HIPCHK(ctx, e);
        return PZ_OK;
    }
    if (ctx->dev_cache_limit && bytes >= PZ_DEV_CACHE_MIN) {
        int best = -1;   // best fit: the smallest cached block that holds the request and is at most an eighth larger
        for (size_t i = 0; i < ctx->dev_cache.size(); ++i) {
            const size_t c = ctx->dev_cache[i].second;
            if (c >= bytes && c - bytes <= bytes / 8 && (best < 0 || c < ctx->dev_cache[(size_t)best].second)) best = (int)i;
        }
        if (best >= 0) {
            *d_out = ctx->dev_cache[(size_t)best].first;
            ctx->dev_live[*d_out] = ctx->dev_cache[(size_t)best].second;
            ctx->dev_cache_bytes -= ctx->dev_cache[(size_t)best].second;
            ctx->dev_cache.erase(ctx->dev_cache.begin() + best);
            return PZ_OK;
        }
        // a miss.  Cached blocks a little SMALLER than the request (down to 7/8 of it) are what the same array was for a key of slightly
        // fewer columns: the block allocated now serves both sizes from here on, so they are obsolete -- released before the allocation.
        // Without this the cache converges to TWO blocks per array (one per size class on either side of a rounding boundary): 2 x 150 GB
        // at config c2 (found by the 40-message soak of prove_connected --fresh: out of memory at the fifth key)
        for (size_t i = 0; i < ctx->dev_cache.size();) {
            const size_t c = ctx->dev_cache[i].second;
            if (c < bytes && bytes - c <= bytes / 8) {
                (void)pz_hip_free(ctx->dev_cache[i].first);
                ctx->dev_cache_bytes -= c;
                ctx->dev_cache.erase(ctx->dev_cache.begin() + (long)i);
            } else {
                ++i;
            }
        }
    }
    {
        hipError_t e = pz_hip_malloc(ctx, d_out, bytes);
        if (e == hipErrorOutOfMemory && ws_msm_held(ctx)) {
            (void)hipGetLastError();
            HIPCHK(ctx, ws_release_msm(ctx));
            e = pz_hip_malloc(ctx, d_out, bytes);
        }
        HIPCHK(ctx, e);
    }
    if (ctx->dev_cache_limit && bytes >= PZ_DEV_CACHE_MIN) ctx->dev_live[*d_out] = bytes;
    return PZ_OK;
}
extern "C" int pz_dev_free(pz_ctx* ctx, void* d) {
    if (!ctx) return PZ_ERR_INVALID;
    if (!d) return PZ_OK;
    PZ_ENTER(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // kernels queued by this context may still use it
    if (pz_arena* a = arena_of(d)) {
        HIPCHK(ctx, arena_give(a, d));
        return PZ_OK;
    }
    auto it = ctx->dev_live.find(d);
    if (it != ctx->dev_live.end()) {
        const size_t bytes = it->second;
        ctx->dev_live.erase(it);
        if (ctx->dev_cache_limit && ctx->dev_cache_bytes + bytes <= ctx->dev_cache_limit) {
            ctx->dev_cache.push_back({d, bytes});
            ctx->dev_cache_bytes += bytes;
            return PZ_OK;
        }
    }
    HIPCHK(ctx, pz_hip_free(d));
    return PZ_OK;
}
extern "C" int pz_upload(pz_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !src))) return PZ_ERR_INVALID;
    if (!bytes) return PZ_OK;
    PZ_ENTER(ctx);
    HIPCHK(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PZ_OK;
}
extern "C" int pz_download(pz_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !d_src))) return PZ_ERR_INVALID;
    if (!bytes) return PZ_OK;
    PZ_ENTER(ctx);
    HIPCHK(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return pz_check_async(ctx);
}
extern "C" int pz_dev_memset(pz_ctx* ctx, void* d_dst, int byte_value, size_t bytes) {
    if (!ctx || (bytes && !d_dst)) return PZ_ERR_INVALID;
    if (!bytes) return PZ_OK;
    PZ_ENTER(ctx);
    HIPCHK(ctx, hipMemsetAsync(d_dst, byte_value, bytes, ctx->stream));
    return PZ_OK;
}
extern "C" int pz_dev_copy(pz_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !d_src))) return PZ_ERR_INVALID;
    if (!bytes) return PZ_OK;
    PZ_ENTER(ctx);
    HIPCHK(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PZ_OK;
}
// `rows` runs of `width` bytes, the runs dst_pitch / src_pitch bytes apart: the blinding rows of a proof's columns (a few rows at
// the end of every 2^k-row column) filled from one staged block
extern "C" int pz_dev_copy_2d(pz_ctx* ctx, void* d_dst, size_t dst_pitch, const void* d_src, size_t src_pitch, size_t width, size_t rows) {
    if (!ctx || (width && rows && (!d_dst || !d_src)) || dst_pitch < width || src_pitch < width) return PZ_ERR_INVALID;
    if (!width || !rows) return PZ_OK;
    PZ_ENTER(ctx);
    HIPCHK(ctx, hipMemcpy2DAsync(d_dst, dst_pitch, d_src, src_pitch, width, rows, hipMemcpyDeviceToDevice, ctx->stream));
    return PZ_OK;
}
// everything queued on `producer`'s stream so far happens before whatever `waiter` queues from now on (an event, no host
// synchronisation): how a host with several contexts -- witness / commitments / transforms, each with its own stream -- orders a
// buffer one context writes and another reads
extern "C" int pz_ctx_wait(pz_ctx* waiter, pz_ctx* producer) {
    if (!waiter || !producer) return PZ_ERR_INVALID;
    if (waiter == producer) return PZ_OK;
    if (waiter->device != producer->device) return PZ_ERR_UNSUPPORTED;
    // both contexts' mutexes, taken by std::scoped_lock's deadlock-avoiding algorithm: one thread per context issuing
    // pz_ctx_wait(a, b) and pz_ctx_wait(b, a) at the same time must not take them in opposite orders
    std::scoped_lock<std::recursive_mutex, std::recursive_mutex> both(waiter->mu, producer->mu);
    HIPCHK(waiter, hipSetDevice(waiter->device));
    hipEvent_t ev;
    HIPCHK(waiter, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, producer->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(waiter->stream, ev, 0);
    (void)hipEventDestroy(ev);
    if (e != hipSuccess) return pz_hip_fail(waiter, e, "pz_ctx_wait");
    return PZ_OK;
}

int pz_upload_small_async(pz_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    const unsigned k = ctx->stage_next++ & 3u;
    if (ctx->stage_ev[k]) HIPCHK(ctx, hipEventSynchronize(ctx->stage_ev[k]));   // the copy queued from this slot four calls ago
    else HIPCHK(ctx, hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    if (ctx->stage_cap[k] < bytes) {
        if (ctx->stage_h[k]) HIPCHK(ctx, hipHostFree(ctx->stage_h[k]));
        ctx->stage_h[k] = nullptr;
        ctx->stage_cap[k] = 0;
        HIPCHK(ctx, hipHostMalloc(&ctx->stage_h[k], bytes + 256, hipHostMallocDefault));
        ctx->stage_cap[k] = bytes + 256;
    }
    memcpy(ctx->stage_h[k], src, bytes);
    HIPCHK(ctx, hipMemcpyAsync(d_dst, ctx->stage_h[k], bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->stage_ev[k], ctx->stream));
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
    return pz_check_async(ctx);
}

int pz_async_err_init(pz_ctx* ctx) {
    if (ctx->async_err_h) return PZ_OK;
    void *h = nullptr, *d = nullptr;
    HIPCHK(ctx, hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *(volatile unsigned*)h = 0;
    hipError_t e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(h);
        return pz_hip_fail(ctx, e, "hipHostGetDevicePointer(async error flag)");
    }
    ctx->async_err_h = (volatile unsigned*)h;
    ctx->async_err_d = (volatile unsigned*)d;
    return PZ_OK;
}

int pz_check_async(pz_ctx* ctx) {
    if (!ctx->async_err_h || !*ctx->async_err_h) return PZ_OK;
    *ctx->async_err_h = 0;
    snprintf(ctx->hip_err, sizeof ctx->hip_err, "a sort kernel of pz_msm_g1* found an entry position outside its list: the scalars "
             "changed between the call's two passes over them (they must stay untouched until the context is synchronised)");
    return PZ_ERR_ASYNC;
}

int pz_ws_get(pz_ctx* ctx, int slot, size_t bytes, void** out) {
    pz_wsbuf& w = ctx->ws[slot];
    if (w.cap < bytes) {
        // in-flight kernels may still use the old buffer
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (w.d) HIPCHK(ctx, pz_hip_free(w.d));
        w.d = nullptr;
        w.cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIPCHK(ctx, pz_hip_malloc(ctx, &w.d, want));
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
    void* reuse = nullptr;
    void* reuse_raw = nullptr;
    size_t reuse_cap = 0;
    if (ctx->pow_tables.size() >= 48) {
        // evict the least recently used table.  Its buffer is handed to the new table when it is large enough: the kernel
        // that fills it runs on the context's stream, BEHIND every kernel still reading the old contents, so no host
        // synchronisation and no free / malloc pair (a prover draws fresh challenge points every proof: six new 2^k-entry
        // tables per proof went through hipStreamSynchronize + hipFree + hipMalloc here)
        size_t lru = 0;
        for (size_t k = 1; k < ctx->pow_tables.size(); ++k)
            if (ctx->pow_tables[k].stamp < ctx->pow_tables[lru].stamp) lru = k;
        if (ctx->pow_tables[lru].cap >= n) {
            reuse = ctx->pow_tables[lru].d;
            reuse_cap = ctx->pow_tables[lru].cap;
            reuse_raw = ctx->pow_tables[lru].d_raw;   // same capacity: rebuilt on demand (in stream order)
        } else {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            HIPCHK(ctx, pz_hip_free(ctx->pow_tables[lru].d));
            if (ctx->pow_tables[lru].d_raw) HIPCHK(ctx, pz_hip_free(ctx->pow_tables[lru].d_raw));
        }
        ctx->pow_tables.erase(ctx->pow_tables.begin() + lru);
    }
    pz_pow_table t;
    t.stamp = ++ctx->pow_clock;
    memcpy(t.base, base, 32);
    memcpy(t.init, init, 32);
    t.n = n;
    if (reuse) {
        t.d = reuse;
        t.cap = reuse_cap;
        t.d_raw = reuse_raw;
        t.raw_valid = false;
    } else {
        HIPCHK(ctx, pz_hip_malloc(ctx, &t.d, n * 32));
        t.cap = n;
    }
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

int pz_raw29_convert(pz_ctx* ctx, const void* d_src_fr, void* d_dst_u32, size_t count) {
    if (!count) return PZ_OK;
    hipLaunchKernelGGL(k_raw29, dim3(pz_div_up(count, 256)), dim3(256), 0, ctx->stream, (const Fr*)d_src_fr, (u32*)d_dst_u32, count);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

int pz_get_pow_table_raw(pz_ctx* ctx, const uint64_t base[4], size_t n, void** d_raw_out, const uint64_t* init) {
    void* d;
    PZCHK(pz_get_pow_table(ctx, base, n, &d, init));
    for (auto& t : ctx->pow_tables)
        if (t.d == d) {
            if (!t.d_raw) {
                HIPCHK(ctx, pz_hip_malloc(ctx, &t.d_raw, t.cap * 72));
                t.raw_valid = false;
            }
            if (!t.raw_valid) {
                PZCHK(pz_raw29_convert(ctx, t.d, t.d_raw, t.n));
                t.raw_valid = true;
            }
            *d_raw_out = t.d_raw;
            return PZ_OK;
        }
    return PZ_ERR_INVALID;
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

