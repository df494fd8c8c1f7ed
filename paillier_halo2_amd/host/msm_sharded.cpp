// msm_sharded.cpp -- config c4 from plain C++ over the C ABI (include/pz.h): ONE large G1 multi-scalar multiplication sharded over
// N contexts, one host thread per context -- what a Rust host without torch builds from N pz_ctx (one per device id),
// pz_msm_g1_dev(win_lo, win_hi) on its window range (north_star's split) or on its point range (SURVEY 8e's alternative), a
// 96-byte download per rank and pz_g1_sum in rank order.  No torch, no RCCL, no HIP call of its own:
// inside one process the "exchange" is N x 96 bytes through host memory; across processes the same 96 bytes travel by the host's
// own transport (or torch.distributed, paillier_halo2_amd/dist.py).
//
// usage: msm_sharded <log_n> <n_ctx> <windows|points> [seed] [steps] [dev0,dev1,...]
//   device list shorter than n_ctx is cycled (a one-GPU box runs all contexts on device 0: the shares then time-slice, the result
//   is the same).  Bases P_i = [s + i t] G (fixed-base multiplication on the device), scalars from a counter-mode splitmix64 below
//   2^252, so the caller (tests/test_gpu_cpp_mirror.py) can recompute sum_i c_i (s + i t) and the expected point itself.
// prints one JSON line: the affine result of the whole MSM on one context and of the folded shares (hex limbs), times.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pz.h"

#define CK(x)                                                                                             \
    do {                                                                                                  \
        int rc_ = (x);                                                                                    \
        if (rc_ != PZ_OK) {                                                                               \
            fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, pz_strerror(rc_));     \
            exit(2);                                                                                      \
        }                                                                                                 \
    } while (0)

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// contiguous, disjoint, exhaustive split of [0, units): rank r gets [lo, hi)  (== paillier_halo2_amd/dist.py::window_range)
static void unit_range(size_t units, size_t rank, size_t world, size_t* lo, size_t* hi) {
    *lo = rank * units / world;
    *hi = (rank + 1) * units / world;
}

struct Rank {
    pz_ctx* ctx = nullptr;
    pz_bases* bases = nullptr;
    void *d_bases = nullptr, *d_scalars = nullptr, *d_out = nullptr;
    size_t lo = 0, hi = 0;       // this rank's units (windows or points)
    uint64_t part[12] = {0};
};

static void hex256(const uint64_t* w, char* out) { snprintf(out, 65, "%016llx%016llx%016llx%016llx", (unsigned long long)w[3], (unsigned long long)w[2], (unsigned long long)w[1], (unsigned long long)w[0]); }

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: msm_sharded <log_n> <n_ctx> <windows|points> [seed] [steps] [devices]\n"); return 2; }
    const unsigned log_n = (unsigned)atoi(argv[1]);
    const size_t world = (size_t)atoi(argv[2]);
    const bool by_points = !strcmp(argv[3], "points");
    const uint64_t seed = argc > 4 ? strtoull(argv[4], nullptr, 0) : 0x5045;
    const size_t steps = argc > 5 ? (size_t)atoi(argv[5]) : 3;
    std::vector<int> devs;
    if (argc > 6) for (char* t = strtok(argv[6], ","); t; t = strtok(nullptr, ",")) devs.push_back(atoi(t));
    if (devs.empty()) devs.push_back(0);
    if (log_n < 1 || log_n > 24 || world < 1 || world > 64) { fprintf(stderr, "bad arguments\n"); return 2; }
    if (pz_abi_version() != PZ_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const size_t n = (size_t)1 << log_n;

    // ---- inputs on the host: discrete logs s + i t of the bases (canonical), scalars below 2^252 (canonical)
    const uint64_t s_lo = splitmix64(seed) | 1, t_lo = splitmix64(seed + 1) >> 8;    // s = s_lo + 2^200, t = t_lo: s + n t < r
    std::vector<uint64_t> dl(n * 4), sc(n * 4);
    for (size_t i = 0; i < n; ++i) {
        const unsigned __int128 v = (unsigned __int128)s_lo + (unsigned __int128)t_lo * i;
        dl[4 * i] = (uint64_t)v;
        dl[4 * i + 1] = (uint64_t)(v >> 64);
        dl[4 * i + 2] = 0;
        dl[4 * i + 3] = 1ull << 8;   // + 2^200
        for (int k = 0; k < 4; ++k) sc[4 * i + k] = splitmix64(seed * 0x100000001B3ull + 4 * i + k);
        sc[4 * i + 3] &= 0x0FFFFFFFFFFFFFFFull;
    }

    // ---- reference: the whole MSM on one context
    pz_ctx* c0;
    int d0 = devs[0];
    CK(pz_init(1, &d0, &c0));
    void *d_dl, *d_b, *d_s, *d_o;
    CK(pz_dev_alloc(c0, n * 32, &d_dl));
    CK(pz_dev_alloc(c0, n * 64, &d_b));
    CK(pz_dev_alloc(c0, n * 32, &d_s));
    CK(pz_dev_alloc(c0, 96, &d_o));
    CK(pz_upload(c0, d_dl, dl.data(), n * 32));
    CK(pz_fr_convert_dev(c0, (uint64_t*)d_dl, n, 1));
    CK(pz_g1_fixed_base_mul_dev(c0, (const uint64_t*)d_dl, n, (uint64_t*)d_b));
    CK(pz_upload(c0, d_s, sc.data(), n * 32));
    CK(pz_fr_convert_dev(c0, (uint64_t*)d_s, n, 1));   // the ABI takes Montgomery form
    CK(pz_sync(c0));
    std::vector<uint64_t> h_bases(n * 8), h_sc(n * 4);
    CK(pz_download(c0, h_bases.data(), d_b, n * 64));
    CK(pz_download(c0, h_sc.data(), d_s, n * 32));
    pz_bases* b0;
    CK(pz_bases_load_g1(c0, (const uint64_t*)d_b, n, 1, 0, &b0));
    uint32_t nwin = 0, cbits = 0;
    CK(pz_bases_info(b0, nullptr, &cbits, &nwin));
    uint64_t whole[12], whole_aff[8];
    double t_whole = 0;
    for (size_t it = 0; it < steps + 1; ++it) {   // first pass warms tables / workspaces
        const auto t0 = std::chrono::steady_clock::now();
        CK(pz_msm_g1_dev(c0, b0, (const uint64_t*)d_s, 1, n, 4 * n, 0, nwin, (uint64_t*)d_o));
        CK(pz_download(c0, whole, d_o, 96));
        if (it) t_whole += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    CK(pz_g1_normalize(c0, whole, 1, whole_aff));

    // ---- the shares: one context per rank, each with what the rank of an N-GPU run holds
    std::vector<Rank> ranks(world);
    for (size_t r = 0; r < world; ++r) {
        Rank& R = ranks[r];
        int dev = devs[r % devs.size()];
        CK(pz_init(1, &dev, &R.ctx));
        unit_range(by_points ? n : nwin, r, world, &R.lo, &R.hi);
        CK(pz_dev_alloc(R.ctx, 96, &R.d_out));
        if (by_points) {   // its own N / world bases and scalars
            const size_t cnt = R.hi - R.lo;
            if (cnt) {
                CK(pz_dev_alloc(R.ctx, cnt * 64, &R.d_bases));
                CK(pz_dev_alloc(R.ctx, cnt * 32, &R.d_scalars));
                CK(pz_upload(R.ctx, R.d_bases, h_bases.data() + R.lo * 8, cnt * 64));
                CK(pz_upload(R.ctx, R.d_scalars, h_sc.data() + R.lo * 4, cnt * 32));
                CK(pz_bases_load_g1(R.ctx, (const uint64_t*)R.d_bases, cnt, 1, 0, &R.bases));
            }
        } else {           // all bases and all scalars, a window range
            CK(pz_dev_alloc(R.ctx, n * 64, &R.d_bases));
            CK(pz_dev_alloc(R.ctx, n * 32, &R.d_scalars));
            CK(pz_upload(R.ctx, R.d_bases, h_bases.data(), n * 64));
            CK(pz_upload(R.ctx, R.d_scalars, h_sc.data(), n * 32));
            CK(pz_bases_load_g1(R.ctx, (const uint64_t*)R.d_bases, n, 1, cbits, &R.bases));
        }
    }
    auto share = [&](size_t r) {
        Rank& R = ranks[r];
        if (by_points) {
            if (!R.bases) { memset(R.part, 0, 96); return; }   // Jacobian identity: z = 0
            uint32_t w = 0;
            CK(pz_bases_info(R.bases, nullptr, nullptr, &w));
            CK(pz_msm_g1_dev(R.ctx, R.bases, (const uint64_t*)R.d_scalars, 1, R.hi - R.lo, 4 * (R.hi - R.lo), 0, w, (uint64_t*)R.d_out));
        } else {
            CK(pz_msm_g1_dev(R.ctx, R.bases, (const uint64_t*)R.d_scalars, 1, n, 4 * n, (uint32_t)R.lo, (uint32_t)R.hi, (uint64_t*)R.d_out));
        }
        CK(pz_download(R.ctx, R.part, R.d_out, 96));   // the 96 bytes a rank contributes to the exchange
    };
    uint64_t folded[12], folded_aff[8];
    double t_shares = 0;
    for (size_t it = 0; it < steps + 1; ++it) {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (size_t r = 0; r < world; ++r) th.emplace_back(share, r);   // one host thread per context, as one process per GPU would run
        for (auto& x : th) x.join();
        std::vector<uint64_t> parts(world * 12);
        for (size_t r = 0; r < world; ++r) memcpy(&parts[12 * r], ranks[r].part, 96);   // rank order == the fixed fold order
        CK(pz_g1_sum(c0, parts.data(), world, folded));
        if (it) t_shares += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    CK(pz_g1_normalize(c0, folded, 1, folded_aff));
    // ---- the same in ONE call of the ABI (pz_msm_g1_multi: the shares queued from this thread, partials folded inside)
    uint64_t multi[12], multi_aff[8];
    double t_multi = 0;
    {
        std::vector<pz_ctx*> cs(world);
        std::vector<const pz_bases*> bs(world);
        std::vector<const uint64_t*> ss(world);
        std::vector<size_t> ns(world);
        for (size_t r = 0; r < world; ++r) {
            cs[r] = ranks[r].ctx;
            bs[r] = ranks[r].bases;
            ss[r] = (const uint64_t*)ranks[r].d_scalars;
            ns[r] = by_points ? ranks[r].hi - ranks[r].lo : n;
        }
        for (size_t it = 0; it < steps + 1; ++it) {
            const auto t0 = std::chrono::steady_clock::now();
            CK(pz_msm_g1_multi(cs.data(), bs.data(), ss.data(), ns.data(), world, by_points ? 1 : 0, multi));
            if (it) t_multi += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        CK(pz_g1_normalize(c0, multi, 1, multi_aff));
    }
    char hx[4][65];
    hex256(whole_aff, hx[0]); hex256(whole_aff + 4, hx[1]); hex256(folded_aff, hx[2]); hex256(folded_aff + 4, hx[3]);
    printf("{\"log_n\": %u, \"contexts\": %zu, \"split\": \"%s\", \"window_bits\": %u, \"n_windows\": %u, \"devices\": %zu, \"equal\": %s, \"multi_call_equal\": %s, "
           "\"whole_ms\": %.4f, \"sharded_ms\": %.4f, \"multi_call_ms\": %.4f, \"s_lo\": \"%llu\", \"t_lo\": \"%llu\", \"seed\": \"%llu\", "
           "\"whole_affine_mont\": [\"%s\", \"%s\"], \"sharded_affine_mont\": [\"%s\", \"%s\"]}\n",
           log_n, world, by_points ? "points" : "windows", cbits, nwin, devs.size(), memcmp(whole_aff, folded_aff, 64) ? "false" : "true",
           memcmp(whole_aff, multi_aff, 64) ? "false" : "true", t_whole / steps * 1e3, t_shares / steps * 1e3, t_multi / steps * 1e3, (unsigned long long)s_lo, (unsigned long long)t_lo, (unsigned long long)seed, hx[0], hx[1], hx[2],
           hx[3]);
    for (auto& R : ranks) {
        if (R.bases) pz_bases_free(R.ctx, R.bases);
        for (void* d : {R.d_bases, R.d_scalars, R.d_out}) CK(pz_dev_free(R.ctx, d));
        pz_free(R.ctx);
    }
    pz_bases_free(c0, b0);
    for (void* d : {d_dl, d_b, d_s, d_o}) CK(pz_dev_free(c0, d));
    pz_free(c0);
    return (memcmp(whole_aff, folded_aff, 64) || memcmp(whole_aff, multi_aff, 64)) ? 1 : 0;
}
