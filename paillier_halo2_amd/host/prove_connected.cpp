// prove_connected.cpp -- ONE CONNECTED PROOF from plain C++ over the C ABI (include/pz.h): K3 trace -> K4 columns in halo2-lib's
// break-point layout -> keygen -> create_proof (host/create_proof.hpp) with a hashing transcript, the proof written to a file.  No
// torch, no HIP call of its own, no oracle: the compiled-language counterpart of bench_connected.ConnectedWorkload, what a reference
// prover patched at point D of INTEGRATION.md runs between /root/reference/src/bench.rs:161 and :171.
//
// usage: prove_connected <job file> <proof file>
// job file: little-endian u64 words
//   [0] magic 0x435a50  [1] enc_bits  [2] k  [3] lookup_bits  [4] max_rows  [5] blinding_factors  [6] n_adv  [7] n_lk  [8] n_constants
//   [9] kind (0 encrypt, 2 encrypt_uniform)  [10] n_steps_g  [11] n_steps_r  [12] seed  [13] proofs  [14] tile  [15] n_messages
//   then n | g (Ln words each), n^2 (2 Ln), s_toxic (4, Montgomery), starts (n_adv + 1), constants (4 x n_constants, canonical),
//   per message m | r (Ln each), then BYTES: selectors [n_adv][2^k] (padded to 8), then u32 map_col [m][2^k], u32 map_row [m][2^k]
// The circuit structure is an INPUT (paillier_halo2_amd/circuit_structure.py writes it): the dependency's keygen knows it.
// proof file: records  [name_len u64][name bytes, padded to 8][kind u64: 0 points (8 words), 1 evaluations, 2 challenge][count u64]
//   [per-item words u64][data]; proof p's records are prefixed "p<p>/"; the verifying key's commitments are "vk/fixed", "vk/sigma".
// stdout: one JSON line with the timings.
//
// usage: prove_connected --fresh <params file> <proof file>
//   A NEW KEY PAIR AND MESSAGE PER PROOF with the reference's circuit, from compiled code alone: the message's bits are circuit structure
//   (paillier.rs:50-55), so every step generates the structure ON THE DEVICE (pz_circuit_structure_dev), runs keygen on its device arrays,
//   writes the witness (K3 + K4 in the structure's break-point layout) and proves -- what bench.rs:161-171 pays per message.
//   params file (u64 words): [0] magic 0x465a50  [1] enc_bits  [2] k  [3] lookup_bits  [4] minimum_rows  [5] blinding_factors  [6] seed
//   [7] steps  [8] tile  then s_toxic (4, Montgomery), then per step n | g | m | r (Ln words each).
//   Device memory comes from one arena (pz_dev_arena; PZ_PROVE_ARENA_GB): re-allocating 116-250 GB per message through the driver costs seconds.
//
// PZ_PROVE_VIA_STEPPER=1 (job-file mode): the SAME proofs through the library's entry points instead of this file's own composition --
//   pz_pk_create (the structure's host arrays, as halo2's Assembly would hand them over) and pz_proof_begin ... pz_proof_open_finish, one call
//   per transcript round -- with the NEXT proof's witness written by a second host thread on a second context while the stepper runs
//   (the two-context recipe of rust/pz-rt prove_pipelined).
#include <chrono>
#include <thread>

#include "create_proof.hpp"

using namespace pzp;

static std::vector<uint64_t> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0 || sz % 8) { fprintf(stderr, "job file length\n"); exit(2); }
    std::vector<uint64_t> w((size_t)sz / 8);
    if (fread(w.data(), 8, w.size(), f) != w.size()) { fprintf(stderr, "short read\n"); exit(2); }
    fclose(f);
    return w;
}

struct Out {
    FILE* f;
    void rec(const std::string& name, uint64_t kind, uint64_t count, uint64_t per, const uint64_t* data) {
        const uint64_t len = name.size();
        std::string padded = name;
        padded.resize((len + 7) / 8 * 8, '\0');
        fwrite(&len, 8, 1, f);
        fwrite(padded.data(), 1, padded.size(), f);
        fwrite(&kind, 8, 1, f);
        fwrite(&count, 8, 1, f);
        fwrite(&per, 8, 1, f);
        fwrite(data, 8, count * per, f);
    }
};

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void write_proof(Out& out, const std::string& pre, const Proof& pr, const Transcript& tr) {
    for (auto& c : pr.commitments) out.rec(pre + "c/" + c.first, 0, c.second.size() / 8, 8, c.second.data());
    for (size_t f = 0; f < pr.evals.size(); ++f) {
        const uint64_t per = 4ull * pr.eval_points[f].second;
        out.rec(pre + "e/" + pr.evals[f].first, 1, pr.evals[f].second.size() / per, per, pr.evals[f].second.data());
    }
    for (auto& c : tr.drawn) out.rec(pre + "ch/" + c.first, 2, 1, 4, c.second.v);
}

static int fresh_main(const char* params_path, const char* proof_path) {
    const std::vector<uint64_t> w = slurp(params_path);
    if (w.size() < 13 || w[0] != 0x465a50) { fprintf(stderr, "bad params file\n"); return 2; }
    const uint64_t enc_bits = w[1];
    const unsigned k = (unsigned)w[2], lb = (unsigned)w[3], bf = (unsigned)w[5];
    const size_t minimum_rows = w[4], seed = w[6], steps = w[7], tile = w[8];
    const size_t Ln = enc_bits / 64, L = 2 * Ln, n = (size_t)1 << k;
    if (k < 4 || k > 24 || !Ln || !steps || tile % CHUNK || w.size() != 13 + steps * 4 * Ln) { fprintf(stderr, "params\n"); return 2; }
    const uint64_t* s_tox = &w[9];
    Ctx cx;
    cx.round_cols = true;
    int dev = 0;
    PZP_CK(pz_init(1, &dev, &cx.c));
    // device memory: by default ONE arena over (almost) all free memory, so that a step -- structure, key, workspace, witness: 120 GB at
    // c2, 250 GB at c5, all dropped again at its end -- makes no driver allocation call (pz.h pz_dev_arena).  PZ_PROVE_ARENA_GB=<n> sizes
    // it; PZ_PROVE_ARENA_GB=0 = the older plan: driver allocations recycled through the block cache
    uint64_t arena_gb = 0;
    {
        size_t free_b = 0;
        PZP_CK(pz_dev_mem_info(cx.c, &free_b, nullptr));
        const char* e = getenv("PZ_PROVE_ARENA_GB");
        const size_t reserve = (size_t)4 << 30;   // left to the runtime (code objects, scratch, the other context's small buffers)
        size_t want = e ? (size_t)strtoull(e, nullptr, 10) << 30 : (free_b > reserve ? free_b - reserve : 0);
        if (want && want + ((size_t)1 << 30) > free_b) want = free_b > ((size_t)1 << 30) ? free_b - ((size_t)1 << 30) : 0;
        if (want) {
            PZP_CK(pz_dev_arena(cx.c, want));
            arena_gb = want >> 30;
        } else {
            PZP_CK(pz_dev_cache_limit(cx.c, ~(size_t)0));     // a released key's blocks serve the next key (pz.h)
        }
    }
    const Fr om = pzh::omega(k);
    void *d_g = nullptr, *d_gl = nullptr;
    PZP_CK(pz_dev_alloc(cx.c, n * 64, &d_g));
    PZP_CK(pz_dev_alloc(cx.c, n * 64, &d_gl));
    PZP_CK(pz_srs_setup_g1_dev(cx.c, k, s_tox, om.v, (uint64_t*)d_g, (uint64_t*)d_gl));
    PZP_CK(pz_sync(cx.c));
    pz_bases *bl = nullptr, *bm = nullptr;
    PZP_CK(pz_bases_load_g1(cx.c, (const uint64_t*)d_gl, n, 1, 0, &bl));
    PZP_CK(pz_bases_load_g1(cx.c, (const uint64_t*)d_g, n, 1, 0, &bm));
    pz_dev_free(cx.c, d_g);
    pz_dev_free(cx.c, d_gl);
    Out out{fopen(proof_path, "wb")};
    if (!out.f) { perror(proof_path); return 2; }
    bool degree_ok = true;
    const char* sk = getenv("PZ_PROVE_STREAMED_KEY");      // R: the streamed proving key (0 at config c5)
    const size_t ext_res = sk && *sk ? strtoull(sk, nullptr, 10) : EXT_ALL;
    double sum_after_first = 0, structure_ms = 0, keygen_ms = 0, witness_ms = 0, prove_ms = 0;
    size_t last_adv = 0, last_lk = 0;
    for (size_t si = 0; si < steps; ++si) {
        const uint64_t *vn = &w[13 + si * 4 * Ln], *vg = vn + Ln, *vm = vg + Ln, *vr = vm + Ln;
        PZP_CK(pz_sync(cx.c));
        const double t0 = now_ms();
        // ---- the structure of THIS message's circuit, on the device
        pz_structure* ps = nullptr;
        PZP_CK(pz_circuit_structure_dev(cx.c, 0, (uint32_t)Ln, 64, lb, k, vm, vn, minimum_rows, bf, &ps));
        Structure st;
        st.k = k; st.lookup_bits = lb; st.blinding_factors = bf;
        size_t filled = 0, n_const = 0, cells = 0, lks = 0, ng = 0, nr = 0;
        PZP_CK(pz_structure_info(ps, &st.n_adv, &filled, &st.n_lk, &st.max_rows, &n_const, &cells, &lks, &ng, &nr));
        const uint64_t *consts = nullptr, *d_starts = nullptr, *starts_h = nullptr;
        PZP_CK(pz_structure_arrays(ps, &st.d_selectors, &st.d_map_col, &st.d_map_row, &d_starts, &consts, &starts_h));
        st.constants.assign(consts, consts + 4 * n_const);
        const size_t A = st.n_adv, m = st.m();
        const double t1 = now_ms();
        // ---- keygen on the structure's device arrays; the break points stay (K4), the rest of the structure goes
        ProvingKey* pk = keygen(cx, std::move(st), bl, bm, ext_res);
        pk->st.d_selectors = nullptr; pk->st.d_map_col = pk->st.d_map_row = nullptr;
        uint64_t* d_starts_own = cx.alloc(cx.cols(A + 1));
        PZP_CK(pz_dev_copy(cx.c, d_starts_own, d_starts, (A + 1) * 8));
        PZP_CK(pz_sync(cx.c));
        PZP_CK(pz_structure_free(ps));
        Workspace ws = make_workspace(cx, *pk, tile);
        PZP_CK(pz_sync(cx.c));
        const double t2 = now_ms();
        // ---- the witness: K3 -> K4 in the structure's break-point layout
        uint64_t* d_cols = cx.alloc(cx.cols(m) * n * 4);
        uint64_t* d_steps = cx.alloc((ng + nr + 1 + 64) / 64 * 64 * 4 * L);
        uint64_t* d_mod = cx.alloc(L);
        std::vector<uint64_t> n2(L), c_out(L);
        {   // n^2 by schoolbook (host, 2 Ln words): the modulus K4's refresh cells need
            std::fill(n2.begin(), n2.end(), 0);
            for (size_t i = 0; i < Ln; ++i) {
                unsigned __int128 carry = 0;
                for (size_t j = 0; j < Ln; ++j) {
                    const unsigned __int128 t = (unsigned __int128)vn[i] * vn[j] + n2[i + j] + carry;
                    n2[i + j] = (uint64_t)t;
                    carry = t >> 64;
                }
                n2[i + Ln] = (uint64_t)carry;
            }
        }
        PZP_CK(pz_upload(cx.c, d_mod, n2.data(), L * 8));
        uint32_t sg = 0, sr = 0;
        PZP_CK(pz_paillier_encrypt_dev(cx.c, (uint32_t)Ln, 1, vn, vg, vm, vr, d_steps, ng + nr + 1, &sg, &sr, c_out.data()));
        if (sg != ng || sr != nr) { fprintf(stderr, "trace shape (%u, %u) is not the structure's (%zu, %zu)\n", sg, sr, ng, nr); return 2; }
        std::vector<uint64_t> inputs;
        inputs.insert(inputs.end(), vn, vn + Ln); inputs.insert(inputs.end(), vg, vg + Ln); inputs.insert(inputs.end(), vm, vm + Ln);
        inputs.insert(inputs.end(), vr, vr + Ln); inputs.insert(inputs.end(), c_out.begin(), c_out.end());
        PZP_CK(pz_circuit_expand_cols_dev(cx.c, 0, (uint32_t)Ln, 64, lb, inputs.data(), d_steps, ng, nr, d_mod, d_cols, d_cols + A * n * 4, d_starts_own, A,
                                          pk->st.max_rows, pk->st.max_rows, n));
        PZP_CK(pz_sync(cx.c));
        const double t3 = now_ms();
        Transcript tr((uint64_t)si);
        Proof pr = create_proof(cx, *pk, ws, d_cols, tr, seed + si);
        PZP_CK(pz_sync(cx.c));
        const double t4 = now_ms();
        degree_ok = degree_ok && pr.h_degree_ok;
        const std::string pre = "p" + std::to_string(si) + "/";
        const uint64_t shape[8] = {A, pk->st.n_lk, m, pk->n_sets, pk->st.max_rows, filled, ng, nr};
        out.rec(pre + "shape", 3, 1, 8, shape);
        out.rec(pre + "vk/fixed", 0, pk->F, 8, pk->fixed_commit.data());
        out.rec(pre + "vk/sigma", 0, m, 8, pk->sigma_commit.data());
        out.rec(pre + "ciphertext", 3, 1, L, c_out.data());
        const uint64_t flags[2] = {pr.h_degree_ok, si};
        out.rec(pre + "flags", 3, 1, 2, flags);
        write_proof(out, pre, pr, tr);
        last_adv = A; last_lk = pk->st.n_lk;
        delete pk;
        cx.release();                       // the key's and the workspace's blocks: kept by the library's block cache for the next message
        PZP_CK(pz_sync(cx.c));
        const double t5 = now_ms();
        if (si) {
            sum_after_first += t5 - t0;
            structure_ms += t1 - t0; keygen_ms += t2 - t1; witness_ms += t3 - t2; prove_ms += t4 - t3;
        }
        fprintf(stderr, "[fresh %zu] structure %.0f + keygen %.0f + witness %.0f + proof %.0f + release %.0f = %.0f ms (n_adv %zu)\n", si, t1 - t0, t2 - t1,
                t3 - t2, t4 - t3, t5 - t4, t5 - t0, A);
    }
    fclose(out.f);
    const double d = steps > 1 ? (double)(steps - 1) : 1.0;
    uint64_t ai[6] = {0, 0, 0, 0, 0, 0};
    PZP_CK(pz_dev_arena_info(cx.c, ai));
    printf("{\"mode\": \"fresh_message\", \"steps\": %zu, \"k\": %u, \"enc_bits\": %llu, \"minimum_rows\": %zu, \"n_adv_last\": %zu, \"n_lk_last\": %zu, "
           "\"mean_step_ms\": %.1f, \"of_which\": {\"structure_ms\": %.1f, \"keygen_and_workspace_ms\": %.1f, \"witness_ms\": %.1f, \"proof_ms\": %.1f}, "
           "\"arena\": {\"gib\": %llu, \"peak_gb\": %.2f, \"requests_served\": %llu, \"requests_passed_to_the_driver\": %llu}, \"quotient_degree_ok\": %s}\n",
           steps, k, (unsigned long long)enc_bits, minimum_rows, last_adv, last_lk, sum_after_first / d, structure_ms / d, keygen_ms / d, witness_ms / d,
           prove_ms / d, (unsigned long long)arena_gb, (double)ai[2] / 1e9, (unsigned long long)ai[4], (unsigned long long)ai[5], degree_ok ? "true" : "false");
    pz_bases_free(cx.c, bl);
    pz_bases_free(cx.c, bm);
    pz_free(cx.c);
    return degree_ok ? 0 : 1;
}

// the job's proofs through pz_pk_create + pz_proof_*: -> exit code.  `produce(pi, ctx)` writes proof pi's witness into slot pi & 1 on ctx.
static int stepper_main(Ctx& cx, int dev, Structure& st, const pz_bases* bl, const pz_bases* bm, size_t n_const, size_t tile, size_t proofs, size_t seed,
                        size_t ext_res, const std::function<void(size_t, pz_ctx*, uint64_t*, std::vector<uint64_t>&)>& produce, size_t n_msg, size_t L,
                        double t_setup, const char* proof_path) {
    const size_t n = (size_t)1 << st.k, A = st.n_adv, Lk = st.n_lk, m = st.m();
    const unsigned k = st.k;
    const double t_keygen = now_ms();
    pz_pk* pk = nullptr;
    PZP_CK(pz_pk_create(cx.c, bl, bm, st.k, st.lookup_bits, st.blinding_factors, st.max_rows, A, Lk, st.selectors.data(), st.constants.data(), n_const,
                        st.map_col.data(), st.map_row.data(), tile, ext_res, &pk));
    const double keygen_ms = now_ms() - t_keygen;
    size_t F = 0, m_ = 0, S = 0, bw = 0, ew = 0;
    PZP_CK(pz_pk_info(pk, &F, &m_, &S, &bw, &ew));
    std::vector<uint64_t> vkf(8 * F), vks(8 * m);
    PZP_CK(pz_pk_commitments(pk, vkf.data(), vks.data()));
    Out out{fopen(proof_path, "wb")};
    if (!out.f) { perror(proof_path); return 2; }
    out.rec("vk/fixed", 0, F, 8, vkf.data());
    out.rec("vk/sigma", 0, m, 8, vks.data());
    Ctx cw;
    PZP_CK(pz_init(1, &dev, &cw.c));
    uint64_t* d_slot[2] = {cx.alloc(m * n * 4), cx.alloc(m * n * 4)};
    std::vector<std::vector<uint64_t>> ciphertexts(proofs, std::vector<uint64_t>(L));
    produce(0, cw.c, d_slot[0], ciphertexts[0]);
    PZP_CK(pz_sync(cw.c));
    struct FamW { const char* name; size_t count, pts; };
    const FamW fams[10] = {{"advice", A, 4}, {"lookup_advice", Lk + 1, 1}, {"fixed", F, 1}, {"sigma", m, 1}, {"perm_z", S, 3}, {"lookup_z", Lk, 2},
                           {"perm_inputs", Lk, 2}, {"perm_tables", Lk, 1}, {"random", 1, 1}, {"h", 1, 1}};
    double best = 1e30, sum_after_first = 0;
    bool degree_ok = true;
    for (size_t pi = 0; pi < proofs; ++pi) {
        PZP_CK(pz_sync(cx.c));
        const double t0 = now_ms();
        std::thread wt;
        if (pi + 1 < proofs)   // the next proof's K3 + K4: another thread, another context, the other slot
            wt = std::thread([&, pi] {
                produce(pi + 1, cw.c, d_slot[(pi + 1) & 1], ciphertexts[pi + 1]);
                PZP_CK(pz_sync(cw.c));
            });
        uint64_t* d_cols = d_slot[pi & 1];
        Transcript tr((uint64_t)pi);
        pz_proof* pr = nullptr;
        std::vector<uint64_t> adv(8 * (A + Lk)), ap(8 * Lk), sp(8 * Lk), cz(8 * S), czl(8 * Lk), crnd(8), ch(24), ev(ew), w1(8), w2(8);
        PZP_CK(pz_proof_begin(pk, d_cols, seed + pi, nullptr, PZ_BLINDING_SEEDED_TEST_STREAM, &pr, adv.data()));
        tr.common_points(adv.data(), A + Lk);
        const Fr theta = tr.squeeze("theta");
        PZP_CK(pz_proof_lookups(pr, theta.v, ap.data(), sp.data()));
        tr.common_points(ap.data(), Lk); tr.common_points(sp.data(), Lk);
        const Fr beta = tr.squeeze("beta"), gamma = tr.squeeze("gamma");
        PZP_CK(pz_proof_products(pr, beta.v, gamma.v, cz.data(), czl.data(), crnd.data()));
        tr.common_points(cz.data(), S); tr.common_points(czl.data(), Lk); tr.common_points(crnd.data(), 1);
        const Fr y = tr.squeeze("y");
        PZP_CK(pz_proof_quotient(pr, y.v, ch.data()));
        tr.common_points(ch.data(), 3);
        const Fr x = tr.squeeze("x");
        PZP_CK(pz_proof_evaluate(pr, x.v, ev.data()));
        tr.common_scalars(ev.data(), ev.size() / 4 - 1);      // (h's value is the last element: the verifier computes it itself)
        const Fr shy = tr.squeeze("sh_y"), shv = tr.squeeze("sh_v");
        PZP_CK(pz_proof_open_begin(pr, shy.v, shv.v, w1.data()));
        tr.common_points(w1.data(), 1);
        const Fr shu = tr.squeeze("sh_u");
        int ok = 0;
        PZP_CK(pz_proof_open_finish(pr, shu.v, w2.data(), &ok));
        PZP_CK(pz_proof_free(pr));
        if (wt.joinable()) wt.join();
        const double t2 = now_ms();
        if (t2 - t0 < best) best = t2 - t0;
        if (pi) sum_after_first += t2 - t0;
        degree_ok = degree_ok && ok;
        const std::string pre = "p" + std::to_string(pi) + "/";
        out.rec(pre + "ciphertext", 3, 1, L, ciphertexts[pi].data());
        const uint64_t flags[2] = {(uint64_t)ok, pi % n_msg};
        out.rec(pre + "flags", 3, 1, 2, flags);
        out.rec(pre + "c/advice", 0, A, 8, adv.data());
        out.rec(pre + "c/lookup_advice", 0, Lk, 8, adv.data() + 8 * A);
        out.rec(pre + "c/perm_inputs", 0, Lk, 8, ap.data());
        out.rec(pre + "c/perm_tables", 0, Lk, 8, sp.data());
        out.rec(pre + "c/perm_z", 0, S, 8, cz.data());
        out.rec(pre + "c/lookup_z", 0, Lk, 8, czl.data());
        out.rec(pre + "c/random", 0, 1, 8, crnd.data());
        out.rec(pre + "c/h", 0, 3, 8, ch.data());
        out.rec(pre + "c/w1", 0, 1, 8, w1.data());
        out.rec(pre + "c/w2", 0, 1, 8, w2.data());
        size_t off = 0;
        for (const FamW& f : fams) {
            out.rec(pre + "e/" + f.name, 1, f.count, 4 * f.pts, ev.data() + off);
            off += f.count * f.pts * 4;
        }
        for (auto& c : tr.drawn) out.rec(pre + "ch/" + c.first, 2, 1, 4, c.second.v);
    }
    fclose(out.f);
    printf("{\"proofs\": %zu, \"k\": %u, \"n_adv\": %zu, \"n_lk\": %zu, \"via\": \"pz_pk_create + pz_proof_* (the library's stepper), next witness on a second thread and "
           "context\", \"streamed_key\": %s, \"pipelined_witness\": true, \"keygen_ms\": %.1f, \"setup_ms\": %.1f, \"best_proof_ms\": %.2f, \"mean_proof_ms\": %.2f, "
           "\"of_which_witness_ms\": 0.0, \"quotient_degree_ok\": %s}\n",
           proofs, k, A, Lk, ext_res == EXT_ALL ? "false" : "true", keygen_ms, t_keygen - t_setup, best,
           proofs > 1 ? sum_after_first / (double)(proofs - 1) : best, degree_ok ? "true" : "false");
    PZP_CK(pz_pk_free(pk));
    cx.release();
    pz_free(cw.c);
    return degree_ok ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc == 4 && !strcmp(argv[1], "--fresh")) return fresh_main(argv[2], argv[3]);
    if (argc != 3) { fprintf(stderr, "usage: %s <job file> <proof file>  |  %s --fresh <params file> <proof file>\n", argv[0], argv[0]); return 2; }
    const std::vector<uint64_t> w = slurp(argv[1]);
    if (w.size() < 16 || w[0] != 0x435a50) { fprintf(stderr, "bad job file\n"); return 2; }
    Structure st;
    const uint64_t enc_bits = w[1];
    st.k = (unsigned)w[2]; st.lookup_bits = (unsigned)w[3]; st.max_rows = w[4]; st.blinding_factors = (unsigned)w[5];
    st.n_adv = w[6]; st.n_lk = w[7];
    const size_t n_const = w[8], kind = w[9], ng = w[10], nr = w[11], seed = w[12], proofs = w[13], tile = w[14], n_msg = w[15];
    const size_t Ln = enc_bits / 64, L = 2 * Ln, n = (size_t)1 << st.k, m = st.m(), A = st.n_adv;
    if (st.k < 4 || st.k > 24 || !Ln || !A || st.max_rows + st.blinding_factors + 1 > n || tile % CHUNK || !n_msg || !proofs) {
        fprintf(stderr, "job parameters\n");
        return 2;
    }
    size_t p = 16;
    auto take = [&](std::vector<uint64_t>& v, size_t cnt) {
        if (p + cnt > w.size()) { fprintf(stderr, "job file truncated\n"); exit(2); }
        v.assign(w.begin() + p, w.begin() + p + cnt);
        p += cnt;
    };
    std::vector<uint64_t> vn, vg, n2, s_tox;
    take(vn, Ln); take(vg, Ln); take(n2, L); take(s_tox, 4);
    take(st.starts, A + 1);
    take(st.constants, 4 * n_const);
    std::vector<std::vector<uint64_t>> vm(n_msg), vr(n_msg);
    for (size_t v = 0; v < n_msg; ++v) { take(vm[v], Ln); take(vr[v], Ln); }
    const size_t sel_words = (A * n + 7) / 8, map_words = (m * n + 1) / 2;
    if (p + sel_words + 2 * map_words != w.size()) { fprintf(stderr, "job file length (%zu + %zu + 2 x %zu != %zu)\n", p, sel_words, map_words, w.size()); return 2; }
    st.selectors.assign((const uint8_t*)&w[p], (const uint8_t*)&w[p] + A * n);
    p += sel_words;
    st.map_col.assign((const uint32_t*)&w[p], (const uint32_t*)&w[p] + m * n);
    p += map_words;
    st.map_row.assign((const uint32_t*)&w[p], (const uint32_t*)&w[p] + m * n);
    for (size_t j = 0; j + 1 < st.starts.size(); ++j)
        if (st.starts[j] > st.starts[j + 1] || st.starts[j + 1] - st.starts[j] > st.max_rows) { fprintf(stderr, "break points\n"); return 2; }

    Ctx cx;
    int dev = 0;
    PZP_CK(pz_init(1, &dev, &cx.c));
    // SRS: monomial and Lagrange bases from the seeded scalar (ParamsKZG::setup, as gen_srs does)
    const double t_setup = now_ms();
    const Fr om = pzh::omega(st.k);
    void *d_g = nullptr, *d_gl = nullptr;
    PZP_CK(pz_dev_alloc(cx.c, n * 64, &d_g));
    PZP_CK(pz_dev_alloc(cx.c, n * 64, &d_gl));
    PZP_CK(pz_srs_setup_g1_dev(cx.c, st.k, s_tox.data(), om.v, (uint64_t*)d_g, (uint64_t*)d_gl));
    PZP_CK(pz_sync(cx.c));
    pz_bases *bl = nullptr, *bm = nullptr;
    PZP_CK(pz_bases_load_g1(cx.c, (const uint64_t*)d_gl, n, 1, 0, &bl));
    PZP_CK(pz_bases_load_g1(cx.c, (const uint64_t*)d_g, n, 1, 0, &bm));
    pz_dev_free(cx.c, d_g);
    pz_dev_free(cx.c, d_gl);
    // PZ_PROVE_STREAMED_KEY=R: the streamed proving key (only the first R permuted columns keep their extended forms; 0 at config c5)
    const char* sk = getenv("PZ_PROVE_STREAMED_KEY");
    const size_t ext_res = sk && *sk ? strtoull(sk, nullptr, 10) : EXT_ALL;
    if (const char* vs = getenv("PZ_PROVE_VIA_STEPPER")) {
        if (vs[0] == '1') {
            // the witness of proof pi on the given context into the given columns (its own trace / modulus / break-point buffers)
            uint64_t* s_steps = cx.alloc((ng + nr + 1) * 4 * L);
            uint64_t* s_mod = cx.alloc(L);
            uint64_t* s_starts = cx.alloc(A + 1);
            PZP_CK(pz_upload(cx.c, s_mod, n2.data(), L * 8));
            PZP_CK(pz_upload(cx.c, s_starts, st.starts.data(), (A + 1) * 8));
            PZP_CK(pz_sync(cx.c));
            const uint32_t lb_ = st.lookup_bits;
            const size_t max_rows_ = st.max_rows;
            auto produce_s = [&, lb_, max_rows_](size_t pi, pz_ctx* c, uint64_t* d_cols, std::vector<uint64_t>& c_out) {
                const size_t v = pi % n_msg;
                PZP_CK(pz_dev_memset(c, d_cols, 0, m * n * 32));
                uint32_t sg = 0, sr = 0;
                if (kind == 2)
                    PZP_CK(pz_paillier_encrypt_uniform_dev(c, (uint32_t)Ln, 1, (uint32_t)enc_bits, vn.data(), vg.data(), vm[v].data(), vr[v].data(), s_steps,
                                                           ng + nr + 1, &sg, &sr, c_out.data()));
                else
                    PZP_CK(pz_paillier_encrypt_dev(c, (uint32_t)Ln, 1, vn.data(), vg.data(), vm[v].data(), vr[v].data(), s_steps, ng + nr + 1, &sg, &sr,
                                                   c_out.data()));
                std::vector<uint64_t> inputs;
                for (const auto* x : {&vn, &vg, &vm[v], &vr[v], &c_out}) inputs.insert(inputs.end(), x->begin(), x->end());
                PZP_CK(pz_circuit_expand_cols_dev(c, (int)kind, (uint32_t)Ln, 64, lb_, inputs.data(), s_steps, ng, nr, s_mod, d_cols, d_cols + A * n * 4,
                                                  s_starts, A, max_rows_, max_rows_, n));
            };
            return stepper_main(cx, dev, st, bl, bm, n_const, tile, proofs, seed, ext_res, produce_s, n_msg, L, t_setup, argv[2]);
        }
    }
    const double t_keygen = now_ms();
    ProvingKey* pk = keygen(cx, std::move(st), bl, bm, ext_res);   // st's arrays now live in (or were released by) the key; its scalar fields stay readable
    PZP_CK(pz_sync(cx.c));
    const double keygen_ms = now_ms() - t_keygen;
    Workspace ws = make_workspace(cx, *pk, tile);
    // two witness slots: proof i + 1's K3 + K4 run on a second context under proof i's advice commitments (PZ_PROVE_PIPELINE=0: one
    // context, one slot, everything in order)
    const char* pe = getenv("PZ_PROVE_PIPELINE");
    const bool pipeline = !(pe && pe[0] == '0') && !getenv("PZ_PROVE_TAMPER_WORD");
    Ctx cw;   // the witness context
    if (pipeline) PZP_CK(pz_init(1, &dev, &cw.c));
    pz_ctx* wctx = pipeline ? cw.c : cx.c;
    uint64_t* d_slot[2] = {cx.alloc(m * n * 4), pipeline ? cx.alloc(m * n * 4) : nullptr};
    uint64_t* d_steps = cx.alloc((ng + nr + 1) * 4 * L);
    uint64_t* d_mod = cx.alloc(L);
    uint64_t* d_starts = cx.alloc(A + 1);
    PZP_CK(pz_upload(cx.c, d_mod, n2.data(), L * 8));
    PZP_CK(pz_upload(cx.c, d_starts, pk->st.starts.data(), (A + 1) * 8));
    PZP_CK(pz_sync(cx.c));

    Out out{fopen(argv[2], "wb")};
    if (!out.f) { perror(argv[2]); return 2; }
    out.rec("vk/fixed", 0, pk->F, 8, pk->fixed_commit.data());
    out.rec("vk/sigma", 0, m, 8, pk->sigma_commit.data());
    double best = 1e30, witness_ms = 0, sum_after_first = 0;
    bool degree_ok = true;
    std::vector<std::vector<uint64_t>> ciphertexts(proofs, std::vector<uint64_t>(L));
    size_t produced = 0;
    // K3 + K4 of proof `pi` into its slot, on the witness context.  No device-side wait on the prover's context: the proof that last read
    // this slot (pi - 2) ended with a host synchronisation before proof pi - 1 began, and a pz_ctx_wait HERE would order the witness
    // behind the advice commitments just queued -- the very work it is meant to run under
    auto produce = [&](size_t pi) {
        const size_t v = pi % n_msg;
        uint64_t* d_cols = d_slot[pipeline ? pi & 1 : 0];
        PZP_CK(pz_dev_memset(wctx, d_cols, 0, m * n * 32));
        uint32_t sg = 0, sr = 0;
        std::vector<uint64_t>& c_out = ciphertexts[pi];
        if (kind == 2)
            PZP_CK(pz_paillier_encrypt_uniform_dev(wctx, (uint32_t)Ln, 1, (uint32_t)enc_bits, vn.data(), vg.data(), vm[v].data(), vr[v].data(), d_steps,
                                                   ng + nr + 1, &sg, &sr, c_out.data()));
        else
            PZP_CK(pz_paillier_encrypt_dev(wctx, (uint32_t)Ln, 1, vn.data(), vg.data(), vm[v].data(), vr[v].data(), d_steps, ng + nr + 1, &sg, &sr,
                                           c_out.data()));
        if (sg != ng || sr != nr) { fprintf(stderr, "trace shape (%u, %u) is not the structure's (%zu, %zu)\n", sg, sr, ng, nr); exit(2); }
        std::vector<uint64_t> inputs;
        for (const auto* x : {&vn, &vg, &vm[v], &vr[v], &c_out}) inputs.insert(inputs.end(), x->begin(), x->end());
        PZP_CK(pz_circuit_expand_cols_dev(wctx, (int)kind, (uint32_t)Ln, 64, st.lookup_bits, inputs.data(), d_steps, ng, nr, d_mod, d_cols,
                                          d_cols + A * n * 4, d_starts, A, st.max_rows, st.max_rows, n));
        produced = pi + 1;
    };
    for (size_t pi = 0; pi < proofs; ++pi) {
        const size_t v = pi % n_msg;
        uint64_t* d_cols = d_slot[pipeline ? pi & 1 : 0];
        PZP_CK(pz_sync(cx.c));
        const double t0 = now_ms();
        if (produced <= pi) produce(pi);
        if (pipeline) PZP_CK(pz_ctx_wait(cx.c, wctx));   // the columns are complete before the prover reads them
        if (const char* tw = getenv("PZ_PROVE_TAMPER_WORD")) {   // tests' negative control: flip one bit of one witness word
            const size_t idx = strtoull(tw, nullptr, 10);
            uint64_t word = 0;
            if (idx >= m * n * 4) { fprintf(stderr, "tamper index\n"); return 2; }
            PZP_CK(pz_download(cx.c, &word, d_cols + idx, 8));
            word ^= 1;
            PZP_CK(pz_upload(cx.c, d_cols + idx, &word, 8));
        }
        PZP_CK(pz_sync(cx.c));
        const double t1 = now_ms();
        Transcript tr((uint64_t)pi);
        std::function<void()> hook;
        if (pipeline && pi + 1 < proofs) hook = [&, pi] { produce(pi + 1); };
        Proof pr = create_proof(cx, *pk, ws, d_cols, tr, seed + pi, hook);
        PZP_CK(pz_sync(cx.c));
        const double t2 = now_ms();
        if (t2 - t0 < best) { best = t2 - t0; witness_ms = t1 - t0; }
        if (pi) sum_after_first += t2 - t0;   // the first proof grows the library's workspaces
        degree_ok = degree_ok && pr.h_degree_ok;
        const std::string pre = "p" + std::to_string(pi) + "/";
        out.rec(pre + "ciphertext", 3, 1, L, ciphertexts[pi].data());
        const uint64_t flags[2] = {pr.h_degree_ok, v};
        out.rec(pre + "flags", 3, 1, 2, flags);
        for (auto& c : pr.commitments) out.rec(pre + "c/" + c.first, 0, c.second.size() / 8, 8, c.second.data());
        for (size_t f = 0; f < pr.evals.size(); ++f) {
            const uint64_t per = 4ull * pr.eval_points[f].second;
            out.rec(pre + "e/" + pr.evals[f].first, 1, pr.evals[f].second.size() / per, per, pr.evals[f].second.data());
        }
        for (auto& c : tr.drawn) out.rec(pre + "ch/" + c.first, 2, 1, 4, c.second.v);
    }
    if (pipeline) PZP_CK(pz_sync(wctx));
    fclose(out.f);
    printf("{\"proofs\": %zu, \"k\": %u, \"enc_bits\": %llu, \"n_adv\": %zu, \"n_lk\": %zu, \"cosets\": 3, \"streamed_key\": %s, \"pipelined_witness\": %s, \"keygen_ms\": %.1f, \"setup_ms\": %.1f, "
           "\"best_proof_ms\": %.2f, \"mean_proof_ms\": %.2f, \"of_which_witness_ms\": %.2f, \"quotient_degree_ok\": %s}\n",
           proofs, st.k, (unsigned long long)enc_bits, A, st.n_lk, pk->streamed ? "true" : "false", pipeline ? "true" : "false", keygen_ms, t_keygen - t_setup, best,
           proofs > 1 ? sum_after_first / (double)(proofs - 1) : best, witness_ms, degree_ok ? "true" : "false");
    cx.release();
    if (pipeline) pz_free(cw.c);
    pz_bases_free(cx.c, bl);
    pz_bases_free(cx.c, bm);
    delete pk;
    pz_free(cx.c);
    return degree_ok ? 0 : 1;
}
