// prove_c2.cpp -- the device-resident hot path of one encrypt proof, driven from plain C++ over the C ABI (include/pz.h):
// no torch, no HIP call of its own.  It is what a reference prover patched at points C and D of INTEGRATION.md does between
// bench.rs:161 and :171 (bench_builder -> create_proof): K3 trace -> K4 columns in HBM -> K1 commit_lagrange of every column
// + the full-width commitments of the later phases -> K2 lagrange_to_coeff / coeff_to_extended of every polynomial, with the
// witness of proof i+1 produced beside the commitments and transforms of proof i (three contexts ordered by pz_ctx_wait).
// The work per step is EXACTLY bench.py's ProofWorkload (same counts, same call sizes, same pools); bench.py writes the job
// file, runs this binary and reports its proofs/s as `dropin_device_resident` beside its own `value`.
//
// job file: little-endian u64 words
//   [0] magic 0x335a50  [1] enc_bits  [2] k  [3] lookup_bits  [4] n_steps  [5] msm_full  [6] polys  [7] pool  [8] ntt_batch
//   [9] steps  [10] warmup  [11] log_e  [12] seed  [13] max_rows (rows a column is filled to)  [14] minimum_rows (calculate_params' argument:
//   fixes the column COUNT; paillier_halo2_amd/layout.py RowBudget)
//   then n | g | m | r (Ln words each), res | n^2 (2 Ln words each), s_toxic, omega_n, omega_n_inv, n_inv (4 words each,
//   Montgomery), coset_gens (2^log_e x 4 words), then [n_extra] and per extra message of the SAME circuit shape: m | r (Ln
//   words each), res (2 Ln words)
// Step i proves message i % (1 + n_extra) in witness slot i & 1, as bench.py does.  After the timed loop, untimed, a few more
// PIPELINED steps keep their outputs and are compared with a SERIAL recomputation (one context, synchronised after every call):
// every advice / lookup commitment in affine form and the coefficient + extended forms of sampled columns, bit for bit; the line
// says "verified" and carries a hash of each message's commitments that bench.py compares with its own.
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pz.h"

#define CK(x)                                                                                               \
    do {                                                                                                    \
        int rc_ = (x);                                                                                      \
        if (rc_ != PZ_OK) {                                                                                 \
            fprintf(stderr, "%s:%d: %s -> %d (%s) %s\n", __FILE__, __LINE__, #x, rc_, pz_strerror(rc_), "");   \
            exit(2);                                                                                        \
        }                                                                                                   \
    } while (0)

struct Job {
    uint64_t enc_bits, k, lb, n_steps, msm_full, polys, pool, ntt_batch, steps, warmup, log_e, seed;
    uint64_t max_rows, minimum_rows;   // the row budget (paillier_halo2_amd/layout.py RowBudget): rows a column is filled to / calculate_params' argument
    std::vector<uint64_t> n, g, m, r, res, n2;
    uint64_t s_toxic[4], omega[4], omega_inv[4], n_inv[4];
    std::vector<uint64_t> gens;
    std::vector<std::vector<uint64_t>> vm, vr, vres;   // messages of the same shape: [0] = m, r, res above
};

static Job read_job(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    std::vector<uint64_t> w;
    uint64_t buf[512];
    size_t got;
    while ((got = fread(buf, 8, 512, f)) > 0) w.insert(w.end(), buf, buf + got);
    fclose(f);
    if (w.size() < 15 || w[0] != 0x335a50) { fprintf(stderr, "bad job file\n"); exit(2); }
    Job j;
    j.enc_bits = w[1]; j.k = w[2]; j.lb = w[3]; j.n_steps = w[4]; j.msm_full = w[5]; j.polys = w[6]; j.pool = w[7];
    j.ntt_batch = w[8]; j.steps = w[9]; j.warmup = w[10]; j.log_e = w[11]; j.seed = w[12];
    j.max_rows = w[13]; j.minimum_rows = w[14];
    if (j.max_rows < 8 || j.max_rows > ((uint64_t)1 << j.k) - 7 || j.minimum_rows >= ((uint64_t)1 << j.k)) { fprintf(stderr, "bad row budget\n"); exit(2); }
    size_t Ln = j.enc_bits / 64, p = 15;
    auto take = [&](std::vector<uint64_t>& v, size_t cnt) { v.assign(w.begin() + p, w.begin() + p + cnt); p += cnt; };
    take(j.n, Ln); take(j.g, Ln); take(j.m, Ln); take(j.r, Ln); take(j.res, 2 * Ln); take(j.n2, 2 * Ln);
    memcpy(j.s_toxic, &w[p], 32); p += 4;
    memcpy(j.omega, &w[p], 32); p += 4;
    memcpy(j.omega_inv, &w[p], 32); p += 4;
    memcpy(j.n_inv, &w[p], 32); p += 4;
    take(j.gens, 4u << j.log_e);
    j.vm.push_back(j.m); j.vr.push_back(j.r); j.vres.push_back(j.res);
    if (p < w.size()) {
        const size_t extra = w[p++];
        for (size_t e = 0; e < extra; ++e) {
            std::vector<uint64_t> a, b, c;
            take(a, Ln); take(b, Ln); take(c, 2 * Ln);
            j.vm.push_back(a); j.vr.push_back(b); j.vres.push_back(c);
        }
    }
    if (p != w.size()) { fprintf(stderr, "job file length\n"); exit(2); }
    return j;
}

// pool of uniformly random field elements below 2^252 (valid Montgomery representatives), like bench.py's _rand_fr
// FNV-1a over little-endian 64-bit words (the same walk as bench.py's fnv1a64)
static uint64_t fnv1a64(const uint64_t* w, size_t n_words) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (size_t i = 0; i < n_words; ++i) h = (h ^ w[i]) * 0x100000001B3ull;
    return h;
}

static void fill_pool(pz_ctx* ctx, void* d, size_t elems, uint64_t seed) {
    const size_t chunk = (size_t)1 << 22;
    std::vector<uint64_t> h(chunk * 4);
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (size_t e0 = 0; e0 < elems; e0 += chunk) {
        const size_t ne = elems - e0 < chunk ? elems - e0 : chunk;
        for (size_t i = 0; i < ne * 4; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            h[i] = (i & 3) == 3 ? (s & 0x0FFFFFFFFFFFFFFFull) : s;
        }
        CK(pz_upload(ctx, (char*)d + e0 * 32, h.data(), ne * 32));
    }
}

// all replicas enter the timed region together (and leave the warm-up together): N provers on N devices -- or, on a one-GPU box,
// N provers time-slicing device 0 -- are timed as ONE job
struct Gate {
    std::mutex m;
    std::condition_variable cv;
    size_t n, waiting = 0, round = 0;
    explicit Gate(size_t n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> l(m);
        const size_t r = round;
        if (++waiting == n) { waiting = 0; ++round; cv.notify_all(); }
        else cv.wait(l, [&] { return round != r; });
    }
};

struct Result {
    double value = 0, dt = 0;
    bool verified = false;
    std::string json;
};

// one prover: three contexts on `device`, the two-slot pipeline, J.steps timed steps, then the verification
static void prove(const Job& J, int device, Gate& gate, Result& out) {
    const size_t Ln = J.enc_bits / 64, L = 2 * Ln, n = (size_t)1 << J.k, rows = J.max_rows, E = (size_t)1 << J.log_e;
    pz_ctx *ctx, *ctxw, *ctxn;   // commitments / witness / transforms
    CK(pz_init(1, &device, &ctx));
    CK(pz_init(1, &device, &ctxw));
    CK(pz_init(1, &device, &ctxn));
    // circuit shape
    size_t adv_cells, lk_cells;
    uint32_t ng = 0, nr = 0;
    // the trace lengths are data (m, n): run the trace once to learn them, as the reference's keygen synthesis does
    void* d_steps[2];
    for (int s = 0; s < 2; ++s) CK(pz_dev_alloc(ctxw, J.n_steps * 4 * L * 8, &d_steps[s]));
    std::vector<uint64_t> c_out(L);
    CK(pz_paillier_encrypt_dev(ctxw, (uint32_t)Ln, 1, J.n.data(), J.g.data(), J.m.data(), J.r.data(), (uint64_t*)d_steps[0], J.n_steps, &ng,
                               &nr, c_out.data()));
    if (memcmp(c_out.data(), J.res.data(), L * 8) != 0) { fprintf(stderr, "ciphertext mismatch\n"); exit(2); }
    if ((size_t)ng + nr + 1 != J.n_steps) { fprintf(stderr, "step count mismatch\n"); exit(2); }
    CK(pz_circuit_cells(0, (uint32_t)Ln, 64, (uint32_t)J.lb, ng, nr, &adv_cells, &lk_cells));
    // configured columns = calculate_params(Some(minimum_rows)): at least the columns a cut at `rows` fills (the rest stay empty)
    const size_t count_rows = n - J.minimum_rows;
    auto cols_for = [&](size_t cells) { const size_t a = (cells + rows - 1) / rows, b = (cells + count_rows - 1) / count_rows; return a > b ? a : b; };
    const size_t adv_cols = cols_for(adv_cells), lk_cols = cols_for(lk_cells);
    void *d_adv[2], *d_lk[2], *d_mod, *d_out_adv, *d_out_full, *d_lagr, *d_pool_f, *d_pool_n, *d_ext;
    for (int s = 0; s < 2; ++s) {
        CK(pz_dev_alloc(ctxw, adv_cols * n * 32, &d_adv[s]));
        CK(pz_dev_alloc(ctxw, lk_cols * n * 32, &d_lk[s]));
        CK(pz_dev_memset(ctxw, d_adv[s], 0, adv_cols * n * 32));   // blinding rows stay zero
        CK(pz_dev_memset(ctxw, d_lk[s], 0, lk_cols * n * 32));
    }
    CK(pz_dev_alloc(ctxw, L * 8, &d_mod));
    CK(pz_upload(ctxw, d_mod, J.n2.data(), L * 8));
    CK(pz_dev_alloc(ctx, (adv_cols > lk_cols ? adv_cols : lk_cols) * 96, &d_out_adv));
    CK(pz_dev_alloc(ctx, J.msm_full * 96, &d_out_full));
    // the Lagrange-basis SRS of ParamsKZG::setup, derived on the device, and its window table
    CK(pz_dev_alloc(ctx, n * 64, &d_lagr));
    CK(pz_srs_setup_g1_dev(ctx, (uint32_t)J.k, J.s_toxic, J.omega, nullptr, (uint64_t*)d_lagr));
    CK(pz_sync(ctx));
    pz_bases* bases;
    CK(pz_bases_load_g1(ctx, (const uint64_t*)d_lagr, n, 1, 0, &bases));
    uint32_t nwin = 0;
    CK(pz_bases_info(bases, nullptr, nullptr, &nwin));
    CK(pz_dev_free(ctx, d_lagr));
    // resident pools: full-width scalars of the later phases' commitments; the polynomials the transforms run on
    CK(pz_dev_alloc(ctx, J.pool * n * 32, &d_pool_f));
    CK(pz_dev_alloc(ctxn, J.pool * n * 32, &d_pool_n));
    fill_pool(ctx, d_pool_f, J.pool * n, J.seed);
    fill_pool(ctxn, d_pool_n, J.pool * n, J.seed + 1);
    const size_t ext_cols = J.ntt_batch > 64 ? J.ntt_batch : 64;
    CK(pz_dev_alloc(ctxn, ext_cols * n * E * 32, &d_ext));
    const size_t NV = J.vm.size();
    std::vector<std::vector<uint64_t>> inputs(NV);
    for (size_t v = 0; v < NV; ++v)
        for (const auto* x : {&J.n, &J.g, &J.vm[v], &J.vr[v], &J.vres[v]}) inputs[v].insert(inputs[v].end(), x->begin(), x->end());
    // what a verification step keeps: commitments of its own, sampled columns' coefficient and extended forms
    struct Keep {
        void *adv = nullptr, *lk = nullptr, *coef = nullptr, *ext = nullptr;
    };
    const size_t samp_buf[5] = {0, 0, 0, 1, 1};
    const size_t samp_col[5] = {0, adv_cols / 2, adv_cols - 1, 0, lk_cols ? lk_cols - 1 : 0};
    const size_t NS = lk_cols ? 5 : 3;
    auto new_keep = [&](pz_ctx* c) {
        Keep k;
        CK(pz_dev_alloc(c, adv_cols * 96, &k.adv));
        CK(pz_dev_alloc(c, (lk_cols ? lk_cols : 1) * 96, &k.lk));
        CK(pz_dev_alloc(c, NS * n * 32, &k.coef));
        CK(pz_dev_alloc(c, NS * n * E * 32, &k.ext));
        return k;
    };

    // K3 + K4 of message v into witness slot `slot`, on context c (the witness context in the pipeline)
    auto produce_on = [&](pz_ctx* c, int slot, size_t v) {
        CK(pz_paillier_encrypt_dev(c, (uint32_t)Ln, 1, J.n.data(), J.g.data(), J.vm[v].data(), J.vr[v].data(), (uint64_t*)d_steps[slot], J.n_steps,
                                   &ng, &nr, c_out.data()));
        if (memcmp(c_out.data(), J.vres[v].data(), L * 8) != 0) { fprintf(stderr, "ciphertext mismatch (message %zu)\n", v); exit(2); }
        CK(pz_circuit_expand_dev(c, 0, (uint32_t)Ln, 64, (uint32_t)J.lb, inputs[v].data(), (const uint64_t*)d_steps[slot], ng, nr,
                                 (const uint64_t*)d_mod, (uint64_t*)d_adv[slot], (uint64_t*)d_lk[slot], rows, n));
    };
    auto produce = [&](int slot, size_t v) { produce_on(ctxw, slot, v); };
    auto consume = [&](int slot, const Keep* keep) {   // K1 on the commitment context, K2 on the transform context
        CK(pz_msm_g1_dev(ctx, bases, (const uint64_t*)d_adv[slot], adv_cols, n, 4 * n, 0, nwin, (uint64_t*)(keep ? keep->adv : d_out_adv)));
        CK(pz_msm_g1_dev(ctx, bases, (const uint64_t*)d_lk[slot], lk_cols, n, 4 * n, 0, nwin, (uint64_t*)(keep ? keep->lk : d_out_adv)));
        for (size_t done = 0; done < J.msm_full;) {
            const size_t nc = J.msm_full - done < J.pool ? J.msm_full - done : J.pool;
            CK(pz_msm_g1_dev(ctx, bases, (const uint64_t*)d_pool_f, nc, n, 4 * n, 0, nwin, (uint64_t*)d_out_full + done * 12));
            done += nc;
        }
        // K2: first the proof's own advice / lookup columns (lagrange_to_coeff out of place into the transform buffer: the
        // coefficient form is its own allocation in a prover too), then the pool polynomials of the later phases
        size_t done = 0;
        const void* own[2] = {d_adv[slot], d_lk[slot]};
        const size_t own_cols[2] = {adv_cols, lk_cols};
        for (int b = 0; b < 2; ++b)
            for (size_t c0 = 0; c0 < own_cols[b] && done < J.polys;) {
                size_t nc = own_cols[b] - c0 < J.ntt_batch ? own_cols[b] - c0 : J.ntt_batch;
                if (nc > J.polys - done) nc = J.polys - done;
                CK(pz_ntt_fr_to_dev(ctxn, (const uint64_t*)own[b] + c0 * n * 4, 4 * n, (uint64_t*)d_pool_n, 4 * n, nc, J.omega_inv, (uint32_t)J.k,
                                    nullptr, J.n_inv));   // lagrange_to_coeff: the 1/n belongs to the inverse transform (and costs nothing there)
                CK(pz_ntt_fr_extend_dev(ctxn, (const uint64_t*)d_pool_n, nc, 4 * n, (uint64_t*)d_ext, 4 * n * E, (uint32_t)J.k, (uint32_t)J.log_e,
                                        J.omega, J.gens.data(), nullptr));
                if (keep)   // sampled columns of this batch, copied out behind the transforms (same context, same stream)
                    for (size_t s_ = 0; s_ < NS; ++s_)
                        if (samp_buf[s_] == (size_t)b && samp_col[s_] >= c0 && samp_col[s_] < c0 + nc) {
                            CK(pz_dev_copy(ctxn, (char*)keep->coef + s_ * n * 32, (const char*)d_pool_n + (samp_col[s_] - c0) * n * 32, n * 32));
                            CK(pz_dev_copy(ctxn, (char*)keep->ext + s_ * n * E * 32, (const char*)d_ext + (samp_col[s_] - c0) * n * E * 32, n * E * 32));
                        }
                c0 += nc;
                done += nc;
            }
        for (; done < J.polys;) {
            const size_t nc = J.polys - done < J.ntt_batch ? J.polys - done : J.ntt_batch;
            size_t off = done % J.pool;
            if (off + nc > J.pool) off = 0;
            uint64_t* src = (uint64_t*)d_pool_n + off * n * 4;
            CK(pz_ntt_fr_dev(ctxn, src, nc, 4 * n, J.omega_inv, (uint32_t)J.k, nullptr, J.n_inv));
            CK(pz_ntt_fr_extend_dev(ctxn, src, nc, 4 * n, (uint64_t*)d_ext, 4 * n * E, (uint32_t)J.k, (uint32_t)J.log_e, J.omega, J.gens.data(),
                                    nullptr));
            done += nc;
        }
    };
    size_t steps_done = 0;
    const char* drop = getenv("PZ_PROVE_DROP_EDGE");   // negative test of the verification only: "ready" / "free"
    const bool drop_ready = drop && !strcmp(drop, "ready"), drop_free = drop && !strcmp(drop, "free");
    auto run = [&](size_t steps, const std::vector<Keep>* keeps) {
        if (!steps) return;
        const size_t base = steps_done;
        steps_done += steps;
        produce(0, base % NV);
        for (size_t i = 0; i < steps; ++i) {
            if (!drop_free) {
                CK(pz_ctx_wait(ctxw, ctx));    // the witness of proof i+1 may overwrite its slot once proof i-1's commitments ...
                CK(pz_ctx_wait(ctxw, ctxn));   // ... and transforms have read it
            }
            if (!drop_ready) {
                CK(pz_ctx_wait(ctx, ctxw));    // proof i's commitments and transforms read the columns K4 wrote
                CK(pz_ctx_wait(ctxn, ctxw));
            }
            consume((int)(i & 1), keeps ? &(*keeps)[i] : nullptr);
            if (i + 1 < steps) produce((int)((i + 1) & 1), (base + i + 1) % NV);
        }
    };
    auto sync_all = [&]() { CK(pz_sync(ctxw)); CK(pz_sync(ctx)); CK(pz_sync(ctxn)); };
    run(J.warmup, nullptr);
    sync_all();
    gate.wait();
    const auto t0 = std::chrono::steady_clock::now();
    run(J.steps, nullptr);
    sync_all();
    gate.wait();   // the job ends when its slowest replica does
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // ---- verification (untimed): VSTEPS more pipelined steps keeping their outputs, against a serial recomputation
    const size_t VSTEPS = 4;
    std::vector<Keep> keeps;
    for (size_t i = 0; i < VSTEPS; ++i) keeps.push_back(new_keep(ctx));
    const size_t vbase = steps_done;
    run(VSTEPS, &keeps);
    sync_all();
    bool verified = true;
    char mismatch[256] = "";
    auto fail = [&](size_t step, size_t v, const char* what, size_t idx) {
        if (verified) snprintf(mismatch, sizeof mismatch, "step %zu (message %zu): %s %zu", step, v, what, idx);
        verified = false;
    };
    auto affine = [&](const void* d_jac, size_t cnt) {
        std::vector<uint64_t> jac(cnt * 12), aff(cnt * 8);
        if (!cnt) return aff;
        CK(pz_download(ctx, jac.data(), d_jac, cnt * 96));
        CK(pz_g1_normalize(ctx, jac.data(), cnt, aff.data()));
        return aff;
    };
    std::vector<Keep> refs(NV);
    std::vector<std::vector<uint64_t>> ref_adv(NV), ref_lk(NV);
    std::vector<bool> have(NV, false);
    std::vector<uint64_t> hashes(NV, 0);
    std::vector<uint64_t> ha, hb;
    for (size_t i = 0; i < VSTEPS; ++i) {
        const size_t v = (vbase + i) % NV;
        if (!have[v]) {   // serial: ONE context, a synchronisation after every call
            have[v] = true;
            refs[v] = new_keep(ctx);
            produce_on(ctx, 0, v);
            CK(pz_sync(ctx));
            CK(pz_msm_g1_dev(ctx, bases, (const uint64_t*)d_adv[0], adv_cols, n, 4 * n, 0, nwin, (uint64_t*)refs[v].adv));
            CK(pz_sync(ctx));
            CK(pz_msm_g1_dev(ctx, bases, (const uint64_t*)d_lk[0], lk_cols, n, 4 * n, 0, nwin, (uint64_t*)refs[v].lk));
            CK(pz_sync(ctx));
            for (size_t s_ = 0; s_ < NS; ++s_) {
                const uint64_t* col = (const uint64_t*)(samp_buf[s_] ? d_lk[0] : d_adv[0]) + samp_col[s_] * n * 4;
                uint64_t* cf = (uint64_t*)refs[v].coef + s_ * n * 4;
                CK(pz_ntt_fr_to_dev(ctx, col, 4 * n, cf, 4 * n, 1, J.omega_inv, (uint32_t)J.k, nullptr, J.n_inv));
                CK(pz_sync(ctx));
                CK(pz_ntt_fr_extend_dev(ctx, cf, 1, 4 * n, (uint64_t*)refs[v].ext + s_ * n * E * 4, 4 * n * E, (uint32_t)J.k, (uint32_t)J.log_e, J.omega,
                                        J.gens.data(), nullptr));
                CK(pz_sync(ctx));
            }
            ref_adv[v] = affine(refs[v].adv, adv_cols);
            ref_lk[v] = affine(refs[v].lk, lk_cols);
            std::vector<uint64_t> both(ref_adv[v]);
            both.insert(both.end(), ref_lk[v].begin(), ref_lk[v].end());
            hashes[v] = fnv1a64(both.data(), both.size());
        }
        const std::vector<uint64_t> ga = affine(keeps[i].adv, adv_cols), gl = affine(keeps[i].lk, lk_cols);
        for (size_t c = 0; c < adv_cols; ++c)
            if (memcmp(&ga[c * 8], &ref_adv[v][c * 8], 64)) { fail(i, v, "advice commitment", c); break; }
        for (size_t c = 0; c < lk_cols; ++c)
            if (memcmp(&gl[c * 8], &ref_lk[v][c * 8], 64)) { fail(i, v, "lookup commitment", c); break; }
        for (size_t s_ = 0; s_ < NS; ++s_) {
            ha.resize(n * E * 4); hb.resize(n * E * 4);
            CK(pz_download(ctx, ha.data(), (const char*)keeps[i].coef + s_ * n * 32, n * 32));
            CK(pz_download(ctx, hb.data(), (const char*)refs[v].coef + s_ * n * 32, n * 32));
            if (memcmp(ha.data(), hb.data(), n * 32)) fail(i, v, "coefficient form of sample", s_);
            CK(pz_download(ctx, ha.data(), (const char*)keeps[i].ext + s_ * n * E * 32, n * E * 32));
            CK(pz_download(ctx, hb.data(), (const char*)refs[v].ext + s_ * n * E * 32, n * E * 32));
            if (memcmp(ha.data(), hb.data(), n * E * 32)) fail(i, v, "extended form of sample", s_);
        }
    }
    // one commitment back to the host: the data a transcript would absorb
    std::vector<uint64_t> first(12);
    CK(pz_download(ctx, first.data(), d_out_adv, 96));
    char buf[2048];
    int len = snprintf(buf, sizeof buf,
           "{\"value\": %.6f, \"unit\": \"proofs/s\", \"steps\": %zu, \"warmup\": %zu, \"ms_per_step\": %.3f, \"advice_cols\": %zu, "
           "\"lookup_cols\": %zu, \"msm_full\": %zu, \"polys\": %zu, \"mul_mod_steps\": %zu, \"first_commitment_x_limb0\": %llu, \"device\": %d, "
           "\"verified\": %s, \"pipelined_steps_checked\": %zu, \"messages\": %zu, \"mismatch\": \"%s\", \"commitment_hash_by_message\": {",
           J.steps / dt, (size_t)J.steps, (size_t)J.warmup, dt / J.steps * 1e3, adv_cols, lk_cols, (size_t)J.msm_full, (size_t)J.polys,
           (size_t)J.n_steps, (unsigned long long)first[0], device, verified ? "true" : "false", VSTEPS, NV, mismatch);
    bool first_h = true;
    for (size_t v = 0; v < NV; ++v)
        if (have[v]) {
            len += snprintf(buf + len, sizeof buf - len, "%s\"%zu\": \"%016llx\"", first_h ? "" : ", ", v, (unsigned long long)hashes[v]);
            first_h = false;
        }
    snprintf(buf + len, sizeof buf - len, "}}");
    out.json = buf;
    out.value = J.steps / dt;
    out.dt = dt;
    out.verified = verified;
    pz_bases_free(ctx, bases);
    pz_free(ctxn);
    pz_free(ctxw);
    pz_free(ctx);
}

// usage: prove_c2 <job file> [replicas] [dev0,dev1,...]
//   replicas > 1: config c5 / the weak-scaling headline from plain C++ -- N independent provers (one host thread each, its own three
//   contexts) on the listed devices (cycled: a one-GPU box runs them all on device 0), entering the timed region together; the line
//   carries the aggregate proofs/s and every replica's own line.  Replicas are proofs of the same job: their commitment hashes agree.
int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: prove_c2 <job file> [replicas] [devices]\n"); return 2; }
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // more than four streams in this process
    const Job J = read_job(argv[1]);
    if (pz_abi_version() != PZ_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    const size_t replicas = argc > 2 ? (size_t)atoi(argv[2]) : 1;
    std::vector<int> devs;
    if (argc > 3) for (char* t = strtok(argv[3], ","); t; t = strtok(nullptr, ",")) devs.push_back(atoi(t));
    if (devs.empty()) devs.push_back(0);
    if (replicas < 1 || replicas > 64) { fprintf(stderr, "bad replica count\n"); return 2; }
    Gate gate(replicas);
    std::vector<Result> res(replicas);
    std::vector<std::thread> th;
    for (size_t r = 0; r < replicas; ++r) th.emplace_back([&, r] { prove(J, devs[r % devs.size()], gate, res[r]); });
    for (auto& t : th) t.join();
    if (replicas == 1) {
        printf("%s\n", res[0].json.c_str());
        return 0;
    }
    double dt = 0;
    bool all = true;
    for (auto& r : res) { dt = r.dt > dt ? r.dt : dt; all = all && r.verified; }
    printf("{\"value\": %.6f, \"unit\": \"proofs/s\", \"replicas\": %zu, \"devices\": %zu, \"steps\": %zu, \"ms_per_step\": %.3f, \"verified\": %s, "
           "\"per_replica\": [", replicas * J.steps / dt, replicas, devs.size(), (size_t)J.steps, dt / J.steps * 1e3, all ? "true" : "false");
    for (size_t r = 0; r < replicas; ++r) printf("%s%s", r ? ", " : "", res[r].json.c_str());
    printf("]}\n");
    return 0;
}
