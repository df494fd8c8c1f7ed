// prove_c2.cpp -- the device-resident hot path of one encrypt proof, driven from plain C++ over the C ABI (include/pz.h):
// no torch, no HIP call of its own.  It is what a reference prover patched at points C and D of INTEGRATION.md does between
// bench.rs:161 and :171 (bench_builder -> create_proof): K3 trace -> K4 columns in HBM -> K1 commit_lagrange of every column
// + the full-width commitments of the later phases -> K2 lagrange_to_coeff / coeff_to_extended of every polynomial, with the
// witness of proof i+1 produced beside the commitments and transforms of proof i (three contexts ordered by pz_ctx_wait).
// The work per step is EXACTLY bench.py's ProofWorkload (same counts, same call sizes, same pools); bench.py writes the job
// file, runs this binary and reports its proofs/s as `dropin_device_resident` beside its own `value`.
//
// job file: little-endian u64 words
//   [0] magic 0x325a50  [1] enc_bits  [2] k  [3] lookup_bits  [4] n_steps  [5] msm_full  [6] polys  [7] pool  [8] ntt_batch
//   [9] steps  [10] warmup  [11] log_e  [12] seed
//   then n | g | m | r (Ln words each), res | n^2 (2 Ln words each), s_toxic, omega_n, omega_n_inv, n_inv (4 words each,
//   Montgomery), coset_gens (2^log_e x 4 words)
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    std::vector<uint64_t> n, g, m, r, res, n2;
    uint64_t s_toxic[4], omega[4], omega_inv[4], n_inv[4];
    std::vector<uint64_t> gens;
};

static Job read_job(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    std::vector<uint64_t> w;
    uint64_t buf[512];
    size_t got;
    while ((got = fread(buf, 8, 512, f)) > 0) w.insert(w.end(), buf, buf + got);
    fclose(f);
    if (w.size() < 13 || w[0] != 0x325a50) { fprintf(stderr, "bad job file\n"); exit(2); }
    Job j;
    j.enc_bits = w[1]; j.k = w[2]; j.lb = w[3]; j.n_steps = w[4]; j.msm_full = w[5]; j.polys = w[6]; j.pool = w[7];
    j.ntt_batch = w[8]; j.steps = w[9]; j.warmup = w[10]; j.log_e = w[11]; j.seed = w[12];
    size_t Ln = j.enc_bits / 64, p = 13;
    auto take = [&](std::vector<uint64_t>& v, size_t cnt) { v.assign(w.begin() + p, w.begin() + p + cnt); p += cnt; };
    take(j.n, Ln); take(j.g, Ln); take(j.m, Ln); take(j.r, Ln); take(j.res, 2 * Ln); take(j.n2, 2 * Ln);
    memcpy(j.s_toxic, &w[p], 32); p += 4;
    memcpy(j.omega, &w[p], 32); p += 4;
    memcpy(j.omega_inv, &w[p], 32); p += 4;
    memcpy(j.n_inv, &w[p], 32); p += 4;
    take(j.gens, 4u << j.log_e);
    if (p != w.size()) { fprintf(stderr, "job file length\n"); exit(2); }
    return j;
}

// pool of uniformly random field elements below 2^252 (valid Montgomery representatives), like bench.py's _rand_fr
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

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: prove_c2 <job file>\n"); return 2; }
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // more than four streams in this process
    const Job J = read_job(argv[1]);
    const size_t Ln = J.enc_bits / 64, L = 2 * Ln, n = (size_t)1 << J.k, rows = n - 10, E = (size_t)1 << J.log_e;
    if (pz_abi_version() != PZ_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    pz_ctx *ctx, *ctxw, *ctxn;   // commitments / witness / transforms
    CK(pz_init(1, nullptr, &ctx));
    CK(pz_init(1, nullptr, &ctxw));
    CK(pz_init(1, nullptr, &ctxn));
    // circuit shape
    size_t adv_cells, lk_cells;
    uint32_t ng = 0, nr = 0;
    // the trace lengths are data (m, n): run the trace once to learn them, as the reference's keygen synthesis does
    void* d_steps[2];
    for (int s = 0; s < 2; ++s) CK(pz_dev_alloc(ctxw, J.n_steps * 4 * L * 8, &d_steps[s]));
    std::vector<uint64_t> c_out(L);
    CK(pz_paillier_encrypt_dev(ctxw, (uint32_t)Ln, 1, J.n.data(), J.g.data(), J.m.data(), J.r.data(), (uint64_t*)d_steps[0], J.n_steps, &ng,
                               &nr, c_out.data()));
    if (memcmp(c_out.data(), J.res.data(), L * 8) != 0) { fprintf(stderr, "ciphertext mismatch\n"); return 2; }
    if ((size_t)ng + nr + 1 != J.n_steps) { fprintf(stderr, "step count mismatch\n"); return 2; }
    CK(pz_circuit_cells(0, (uint32_t)Ln, 64, (uint32_t)J.lb, ng, nr, &adv_cells, &lk_cells));
    const size_t adv_cols = (adv_cells + rows - 1) / rows, lk_cols = (lk_cells + rows - 1) / rows;
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
    std::vector<uint64_t> inputs;
    for (const auto* v : {&J.n, &J.g, &J.m, &J.r, &J.res}) inputs.insert(inputs.end(), v->begin(), v->end());

    auto produce = [&](int slot) {   // K3 + K4 on the witness context
        CK(pz_paillier_encrypt_dev(ctxw, (uint32_t)Ln, 1, J.n.data(), J.g.data(), J.m.data(), J.r.data(), (uint64_t*)d_steps[slot], J.n_steps,
                                   &ng, &nr, c_out.data()));
        CK(pz_circuit_expand_dev(ctxw, 0, (uint32_t)Ln, 64, (uint32_t)J.lb, inputs.data(), (const uint64_t*)d_steps[slot], ng, nr,
                                 (const uint64_t*)d_mod, (uint64_t*)d_adv[slot], (uint64_t*)d_lk[slot], rows, n));
    };
    auto consume = [&](int slot) {   // K1 on the commitment context, K2 on the transform context
        CK(pz_msm_g1_dev(ctx, bases, (const uint64_t*)d_adv[slot], adv_cols, n, 4 * n, 0, nwin, (uint64_t*)d_out_adv));
        CK(pz_msm_g1_dev(ctx, bases, (const uint64_t*)d_lk[slot], lk_cols, n, 4 * n, 0, nwin, (uint64_t*)d_out_adv));
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
    auto run = [&](size_t steps) {
        if (!steps) return;
        produce(0);
        for (size_t i = 0; i < steps; ++i) {
            CK(pz_ctx_wait(ctxw, ctx));    // the witness of proof i+1 may overwrite its slot once proof i-1's commitments ...
            CK(pz_ctx_wait(ctxw, ctxn));   // ... and transforms have read it
            CK(pz_ctx_wait(ctx, ctxw));    // proof i's commitments and transforms read the columns K4 wrote
            CK(pz_ctx_wait(ctxn, ctxw));
            consume((int)(i & 1));
            if (i + 1 < steps) produce((int)((i + 1) & 1));
        }
    };
    auto sync_all = [&]() { CK(pz_sync(ctxw)); CK(pz_sync(ctx)); CK(pz_sync(ctxn)); };
    run(J.warmup);
    sync_all();
    const auto t0 = std::chrono::steady_clock::now();
    run(J.steps);
    sync_all();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // one commitment back to the host: the data a transcript would absorb
    std::vector<uint64_t> first(12);
    CK(pz_download(ctx, first.data(), d_out_adv, 96));
    printf("{\"value\": %.6f, \"unit\": \"proofs/s\", \"steps\": %zu, \"warmup\": %zu, \"ms_per_step\": %.3f, \"advice_cols\": %zu, "
           "\"lookup_cols\": %zu, \"msm_full\": %zu, \"polys\": %zu, \"mul_mod_steps\": %zu, \"first_commitment_x_limb0\": %llu}\n",
           J.steps / dt, (size_t)J.steps, (size_t)J.warmup, dt / J.steps * 1e3, adv_cols, lk_cols, (size_t)J.msm_full, (size_t)J.polys,
           (size_t)J.n_steps, (unsigned long long)first[0]);
    pz_bases_free(ctx, bases);
    pz_free(ctxn);
    pz_free(ctxw);
    pz_free(ctx);
    return 0;
}
