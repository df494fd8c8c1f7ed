// pz_internal.h -- shared host-side plumbing of libpz_hip.so (context, workspaces, error handling,
// HIP-event timing).  Not part of the ABI; the ABI is include/pz.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/pz.h"

enum { PZ_T_MSM_ACC = 0, PZ_T_NTT = 1, PZ_T_TRACE = 2, PZ_T_EXPAND = 3, PZ_T_MSM_ALL = 4, PZ_T_MSM_SORT = 5, PZ_T_MSM_TREE = 6, PZ_T_COUNT = 7 };

struct pz_pow_table {   // cached table init * base^i, i < n  (twiddles omega^i, coset powers s*g^i)
    uint64_t base[4];
    uint64_t init[4];
    size_t n;
    size_t cap;         // entries the buffer holds (>= n: an evicted table's buffer is reused)
    void* d;            // cap x 32 B
    void* d_raw = nullptr;   // the same entries as constant pairs (c, floor(c 2^261 / p)), 18 x 29-bit limbs = 72 B each, built on demand for the K2 kernels
    bool raw_valid = false;
    uint64_t stamp;     // last use (LRU: per-proof challenge points would otherwise grow the cache without bound)
};

struct pz_ext_table {   // packed [2^e][n] pre-scale tables of pz_ntt_fr_extend_dev, keyed by its parameters
    std::vector<uint64_t> key;
    void* d;       // constant pairs, 72 B per entry: [2^e][n][18] u32
};

struct pz_wsbuf {
    void* d = nullptr;
    size_t cap = 0;
};

enum { WS_HIST = 0, WS_OFFS, WS_CURSOR, WS_ITEMS, WS_ENTRIES, WS_PARTIALS, WS_NODES_A, WS_NODES_B, WS_NTT_TMP,
       WS_IO_A, WS_IO_B, WS_IO_C, WS_BIG_A, WS_BIG_B, WS_BIG_C, WS_MISC, WS_TOTALS, WS_ORDER, WS_SH_C, WS_SH_SMALL, WS_ROWC, WS_SEL, WS_K3, WS_MULTI, WS_COUNT };

struct pz_event_pair {
    hipEvent_t a, b;
};

#define PZ_IO_EVENTS 10
struct pz_arena;
struct pz_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char hip_err[256] = {0};
    pz_wsbuf ws[WS_COUNT];
    std::vector<pz_pow_table> pow_tables;
    uint64_t pow_clock = 0;
    std::vector<pz_ext_table> ext_tables;
    bool timing = false;
    std::vector<pz_event_pair> ev[PZ_T_COUNT];
    size_t ev_used[PZ_T_COUNT] = {0};
    double ev_ms[PZ_T_COUNT] = {0};
    uint64_t ev_n[PZ_T_COUNT] = {0};
    int cu_count = 256;
    size_t mem_avail = 0;      // (free + held) / 2 as last queried by pz_msm_g1_dev; hipMemGetInfo drains the queue on ROCm,
    unsigned mem_avail_age = 0; // so it is asked again only every 64 calls
    // host-pointer entry points (pz_msm_g1_batch, pz_ntt_fr_batch): copy engines of their own, so that the PCIe transfers
    // of the next / previous column group run beside the kernels of the current one (pz_io_init creates them on first use)
    hipStream_t io_h2d = nullptr, io_d2h = nullptr;
    hipEvent_t io_ev[PZ_IO_EVENTS] = {};
    // small pinned staging blocks for host arguments of asynchronous entry points (pz_upload_small_async): a ring of four, a
    // slot is reused only after the copy recorded on it has completed
    void* stage_h[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_cap[4] = {0, 0, 0, 0};
    hipEvent_t stage_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned stage_next = 0;
    // asynchronous-failure flag: one word of pinned host memory a kernel sets when it finds an internal invariant broken (the
    // scatter kernels' position checks, pz_msm.hip); read and cleared by pz_check_async after a synchronisation
    volatile unsigned* async_err_h = nullptr;
    volatile unsigned* async_err_d = nullptr;   // the same word as the device addresses it
    // pz_dev_alloc / pz_dev_free block cache (pz_dev_cache_limit; off by default): freed blocks of 32 MiB and more are kept, up to
    // dev_cache_limit bytes, and handed to the next request of (nearly) that size -- a caller that builds a 116-GB proving key per
    // message pays the driver's allocation cost once, not per key
    size_t dev_cache_limit = 0, dev_cache_bytes = 0;
    std::vector<std::pair<void*, size_t>> dev_cache;
    std::map<void*, size_t> dev_live;   // sizes of the live pz_dev_alloc blocks (only tracked while the cache is on)
    // pz_dev_arena: one reserved block of device memory this context's allocations (pz_dev_alloc and the library's own buffers)
    // are carved from, so that a caller who builds and drops a proving key per message makes no driver allocation call after start-up
    struct pz_arena* arena = nullptr;
    std::recursive_mutex mu;   // one context is serialised internally: entry points may be called from any thread
};

struct pz_bases {
    size_t n = 0;          // points
    uint32_t c = 0;        // window bits
    uint32_t nwin = 0;     // floor(253/c)+1
    void* d_table = nullptr;  // nwin x n affine points (64 B), window-major
    int lagrange = 0;
    int device = 0;
};

static inline int pz_hip_fail(pz_ctx* ctx, hipError_t e, const char* what) {
    if (ctx) snprintf(ctx->hip_err, sizeof ctx->hip_err, "%s: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? PZ_ERR_OOM : PZ_ERR_HIP;
}
#define HIPCHK(ctx, x)                                              \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) return pz_hip_fail((ctx), e_, #x);    \
    } while (0)
#define PZCHK(x)                    \
    do {                            \
        int rc_ = (x);              \
        if (rc_ != PZ_OK) return rc_; \
    } while (0)

// first statement of every entry point that touches the device: serialise on the context (workspaces, caches and the
// stream order belong to it; entry points call each other, hence recursive) and select its device for this thread
#define PZ_ENTER(ctx)                                           \
    std::lock_guard<std::recursive_mutex> pz_lock_((ctx)->mu);  \
    HIPCHK((ctx), hipSetDevice((ctx)->device))

// grow-only workspace slot; contents are NOT preserved across growth
int pz_ws_get(pz_ctx* ctx, int slot, size_t bytes, void** out);
// hipMalloc for the library's own buffers: on out-of-memory the pz_dev_alloc block cache is released and the request repeated once
hipError_t pz_hip_malloc(pz_ctx* ctx, void** d, size_t bytes);
// the counterpart: a block of ANY context's arena goes back to its arena (after a device synchronisation: hipFree's own semantics, the
// callers rely on it), anything else to hipFree
hipError_t pz_hip_free(void* d);
// free device memory as the library should plan with it: the driver's figure plus what this context's arena has free
size_t pz_mem_free_bytes(pz_ctx* ctx);
void pz_dev_cache_trim(pz_ctx* ctx, size_t keep_bytes);
// cached base^i table (device, Fr Montgomery)
int pz_get_pow_table(pz_ctx* ctx, const uint64_t base[4], size_t n, void** d_out, const uint64_t* init = nullptr);
// the same table as raw 9 x 29-bit limbs per entry (n x 9 u32), cached beside it
int pz_get_pow_table_raw(pz_ctx* ctx, const uint64_t base[4], size_t n, void** d_raw_out, const uint64_t* init = nullptr);
// dst[9 i + j] = limb j (29 bits; the top limb takes the rest) of the 256-bit integer src[i]: asynchronous on the context's stream
int pz_raw29_convert(pz_ctx* ctx, const void* d_src_fr, void* d_dst_u32, size_t count);

// timing scopes: record an event pair around a kernel-class region on ctx->stream
struct pz_timer {
    pz_ctx* ctx;
    int cls;
    bool on;
    size_t idx;
    pz_timer(pz_ctx* c, int cls_);
    ~pz_timer();
};

int pz_async_err_init(pz_ctx* ctx);    // allocates the flag on first use
int pz_check_async(pz_ctx* ctx);       // after a stream synchronisation: PZ_ERR_ASYNC (and the flag cleared) if a kernel raised it
const uint64_t* pz_fr_one261();   // Montgomery one times 32: first entry of a power table kept in the 2^261 domain (fp29.cuh)
int pz_io_init(pz_ctx* ctx);
// asynchronous host -> device copy of a SMALL host argument (pageable memory the caller may free on return): staged through a
// pinned block of the context (ring of four; a block is reused only after the copy queued from it has completed)
int pz_upload_small_async(pz_ctx* ctx, void* d_dst, const void* src, size_t bytes);   // streams + events of the host-pointer pipelines; orders io_h2d after ctx->stream

static inline unsigned pz_div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
