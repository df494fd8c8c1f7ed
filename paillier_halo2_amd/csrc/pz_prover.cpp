// pz_prover.cpp -- patch point D as ENTRY POINTS: keygen and create_proof's phases (paillier_halo2_amd/host/create_proof.hpp, the code
// tests/cpp/prove_connected runs) behind the C ABI, one call per phase -- between two calls the caller's transcript absorbs what the
// phase handed back and squeezes the next challenge (/root/reference/src/bench.rs:161-171: bench_builder -> keygen -> gen_proof ->
// halo2-axiom's create_proof).  Pure host composition of the library's own entry points: no kernel of its own, no HIP call.
// Nothing is thrown across the ABI: a failed step surfaces as its status, the proof handle stays valid for pz_proof_free only.
#define PZP_THROW
#include "../host/create_proof.hpp"

#include <atomic>
#include <memory>
#include <new>

struct pz_pk {
    pzp::Ctx mem;   // the key's and the workspace's device memory (pz_dev_alloc through the creating context), released by pz_pk_free
    pzp::ProvingKey* key = nullptr;
    pzp::Workspace ws;
    // one proof in flight per key (it owns the workspace).  Claimed by an atomic test-and-set: pz.h invites calls from any thread
    // (halo2's rayon workers), and two pz_proof_begin on one key must not both pass the check
    std::atomic<bool> busy{false};
};
struct pz_proof {
    pz_pk* pk;
    pzp::Session* se;
    bool failed = false;   // sticky: after a failed phase only pz_proof_free is valid (pz.h)
};

namespace {
bool fr_canonical(const uint64_t c[4]) {   // challenges cross the ABI as Montgomery representatives below r, like every field element
    for (int i = 3; i >= 0; --i)
        if (c[i] != pzh::FR_MOD[i]) return c[i] < pzh::FR_MOD[i];
    return false;
}
pzh::Fr fr_of(const uint64_t c[4]) {
    pzh::Fr r;
    memcpy(r.v, c, 32);
    return r;
}
template <class F> int guarded_raw(F&& f) {
    try {
        f();
        return PZ_OK;
    } catch (const pzp::PzpError& e) {
        return e.rc;
    } catch (const std::bad_alloc&) {
        return PZ_ERR_OOM;
    } catch (...) {
        return PZ_ERR_INTERNAL;
    }
}
// a phase of an open proof: refused once an earlier phase has failed (the session would run on half-built state), and a failure here is
// remembered
template <class F> int phase(pz_proof* pr, F&& f) {
    if (pr->failed) return PZ_ERR_INVALID;
    const int rc = guarded_raw(f);
    if (rc != PZ_OK) pr->failed = true;
    return rc;
}
size_t evals_words(const pzp::ProvingKey& pk) {
    const size_t A = pk.st.n_adv, Lk = pk.st.n_lk, m = pk.st.m(), S = pk.n_sets;
    return 4 * (4 * A + (Lk + 1) + pk.F + m + 3 * S + 2 * Lk + 2 * Lk + Lk + 1 + 1);
}
}   // namespace

namespace {
// the two forms of pz_pk_create: the structure's arrays in host memory (uploaded as they are: selectors as bytes) or already on the device
int pk_create(pz_ctx* ctx, const pz_bases* bases_lagrange, const pz_bases* bases_monomial, uint32_t k, uint32_t lookup_bits,
              uint32_t blinding_factors, size_t max_rows, size_t n_adv, size_t n_lk, const uint8_t* selectors, const uint64_t* constants,
              size_t n_constants, const uint32_t* map_col, const uint32_t* map_row, size_t tile, size_t ext_resident_cols, bool on_device,
              pz_pk** out) {
    if (!ctx || !bases_lagrange || !bases_monomial || !selectors || !map_col || !map_row || !out || (n_constants && !constants)) return PZ_ERR_INVALID;
    // (n_lk = 0: a circuit without range-check lookups is not a halo2-lib circuit; the composition assumes at least one lookup column)
    if (k < 4 || k > 24 || !n_adv || !n_lk || lookup_bits >= k || tile == 0 || tile % pzp::CHUNK) return PZ_ERR_INVALID;
    const size_t n = (size_t)1 << k;
    if (max_rows + blinding_factors + 1 > n || n_constants > max_rows) return PZ_ERR_INVALID;
    if ((n_adv + n_lk + 1) > ((size_t)1 << 32) / n) return PZ_ERR_UNSUPPORTED;   // the copy-constraint map addresses cells with 32 bits
    size_t np_l = 0, np_m = 0;
    if (pz_bases_info(bases_lagrange, &np_l, nullptr, nullptr) != PZ_OK || pz_bases_info(bases_monomial, &np_m, nullptr, nullptr) != PZ_OK) return PZ_ERR_INVALID;
    if (np_l < n || np_m < n) return PZ_ERR_INVALID;
    *out = nullptr;
    pz_pk* pk = new (std::nothrow) pz_pk;
    if (!pk) return PZ_ERR_OOM;
    pk->mem.c = ctx;
    pk->mem.round_cols = true;   // identical block sizes for keys of nearly equal shape: what pz_dev_cache_limit's cache can serve
    const int rc = guarded_raw([&] {
        pzp::Structure st;
        st.k = k; st.lookup_bits = lookup_bits; st.blinding_factors = blinding_factors; st.max_rows = max_rows; st.n_adv = n_adv; st.n_lk = n_lk;
        const size_t m = st.m();
        st.constants.assign(constants, constants + 4 * n_constants);
        if (on_device) {
            st.d_selectors = selectors; st.d_map_col = map_col; st.d_map_row = map_row;
        } else {   // ONE host copy (moved into the key and released by keygen once it is on the device); selectors travel as bytes
            st.selectors.assign(selectors, selectors + n_adv * n);
            st.map_col.assign(map_col, map_col + m * n);
            st.map_row.assign(map_row, map_row + m * n);
        }
        pk->key = pzp::keygen(pk->mem, std::move(st), bases_lagrange, bases_monomial, ext_resident_cols);
        pk->key->st.d_selectors = nullptr; pk->key->st.d_map_col = pk->key->st.d_map_row = nullptr;   // the caller's arrays are not kept
        pk->ws = pzp::make_workspace(pk->mem, *pk->key, tile);
        PZP_CK(pz_sync(ctx));
    });
    if (rc != PZ_OK) {
        (void)pz_sync(ctx);
        pk->mem.release();
        delete pk->key;
        delete pk;
        return rc;
    }
    *out = pk;
    return PZ_OK;
}
}   // namespace

extern "C" int pz_pk_create(pz_ctx* ctx, const pz_bases* bases_lagrange, const pz_bases* bases_monomial, uint32_t k, uint32_t lookup_bits,
                            uint32_t blinding_factors, size_t max_rows, size_t n_adv, size_t n_lk, const uint8_t* selectors,
                            const uint64_t* constants, size_t n_constants, const uint32_t* map_col, const uint32_t* map_row, size_t tile,
                            size_t ext_resident_cols, pz_pk** out) {
    return pk_create(ctx, bases_lagrange, bases_monomial, k, lookup_bits, blinding_factors, max_rows, n_adv, n_lk, selectors, constants, n_constants,
                     map_col, map_row, tile, ext_resident_cols, false, out);
}
extern "C" int pz_pk_create_dev(pz_ctx* ctx, const pz_bases* bases_lagrange, const pz_bases* bases_monomial, uint32_t k, uint32_t lookup_bits,
                                uint32_t blinding_factors, size_t max_rows, size_t n_adv, size_t n_lk, const uint8_t* d_selectors,
                                const uint64_t* constants, size_t n_constants, const uint32_t* d_map_col, const uint32_t* d_map_row, size_t tile,
                                size_t ext_resident_cols, pz_pk** out) {
    return pk_create(ctx, bases_lagrange, bases_monomial, k, lookup_bits, blinding_factors, max_rows, n_adv, n_lk, d_selectors, constants,
                     n_constants, d_map_col, d_map_row, tile, ext_resident_cols, true, out);
}

extern "C" int pz_pk_info(const pz_pk* pk, size_t* n_fixed, size_t* n_perm_cols, size_t* n_sets, size_t* blinding_words, size_t* evals_words_out) {
    if (!pk || !pk->key) return PZ_ERR_INVALID;
    if (n_fixed) *n_fixed = pk->key->F;
    if (n_perm_cols) *n_perm_cols = pk->key->st.m();
    if (n_sets) *n_sets = pk->key->n_sets;
    if (blinding_words) *blinding_words = pzp::blinding_words(*pk->key);
    if (evals_words_out) *evals_words_out = evals_words(*pk->key);
    return PZ_OK;
}

extern "C" int pz_pk_commitments(const pz_pk* pk, uint64_t* fixed_affine, uint64_t* sigma_affine) {
    if (!pk || !pk->key || !fixed_affine || !sigma_affine) return PZ_ERR_INVALID;
    memcpy(fixed_affine, pk->key->fixed_commit.data(), pk->key->fixed_commit.size() * 8);
    memcpy(sigma_affine, pk->key->sigma_commit.data(), pk->key->sigma_commit.size() * 8);
    return PZ_OK;
}

extern "C" int pz_pk_free(pz_pk* pk) {
    if (!pk) return PZ_OK;
    if (pk->busy.load()) return PZ_ERR_INVALID;   // a proof still holds the workspace: pz_proof_free first
    if (pk->mem.c) (void)pz_sync(pk->mem.c);
    pk->mem.release();
    delete pk->key;
    delete pk;
    return PZ_OK;
}

extern "C" int pz_proof_begin(pz_pk* pk, uint64_t* d_cols, uint64_t seed, const uint64_t* blinding, size_t n_blinding, pz_proof** out,
                              uint64_t* advice_affine) {
    if (!pk || !pk->key || !d_cols || !out || !advice_affine) return PZ_ERR_INVALID;
    if (blinding ? n_blinding < pzp::blinding_words(*pk->key) : n_blinding != PZ_BLINDING_SEEDED_TEST_STREAM) return PZ_ERR_INVALID;
    *out = nullptr;
    bool idle = false;
    if (!pk->busy.compare_exchange_strong(idle, true)) return PZ_ERR_INVALID;   // exactly one caller claims the key
    pz_proof* pr = new (std::nothrow) pz_proof{pk, nullptr};
    if (!pr) {
        pk->busy.store(false);
        return PZ_ERR_OOM;
    }
    const int rc = guarded_raw([&] {
        pr->se = new pzp::Session(pk->mem, *pk->key, pk->ws, d_cols, blinding ? pzp::Rng(blinding, n_blinding) : pzp::Rng(seed));
        pr->se->advice(advice_affine);
    });
    if (rc != PZ_OK) {
        delete pr->se;
        delete pr;
        pk->busy.store(false);
        return rc;
    }
    *out = pr;
    return PZ_OK;
}

extern "C" int pz_proof_lookups(pz_proof* pr, const uint64_t theta[4], uint64_t* perm_inputs_affine, uint64_t* perm_tables_affine) {
    if (!pr || !pr->se || !theta || !perm_inputs_affine || !perm_tables_affine || !fr_canonical(theta)) return PZ_ERR_INVALID;
    return phase(pr, [&] { pr->se->lookups(perm_inputs_affine, perm_tables_affine); });
}

extern "C" int pz_proof_products(pz_proof* pr, const uint64_t beta[4], const uint64_t gamma[4], uint64_t* perm_z_affine, uint64_t* lookup_z_affine,
                                 uint64_t* random_affine) {
    if (!pr || !pr->se || !beta || !gamma || !perm_z_affine || !lookup_z_affine || !random_affine || !fr_canonical(beta) || !fr_canonical(gamma))
        return PZ_ERR_INVALID;
    return phase(pr, [&] { pr->se->products(fr_of(beta), fr_of(gamma), perm_z_affine, lookup_z_affine, random_affine); });
}

extern "C" int pz_proof_quotient(pz_proof* pr, const uint64_t y[4], uint64_t* h_affine) {
    if (!pr || !pr->se || !y || !h_affine || !fr_canonical(y)) return PZ_ERR_INVALID;
    return phase(pr, [&] { pr->se->quotient(fr_of(y), h_affine); });
}

extern "C" int pz_proof_evaluate(pz_proof* pr, const uint64_t x[4], uint64_t* evals) {
    if (!pr || !pr->se || !x || !evals || !fr_canonical(x)) return PZ_ERR_INVALID;
    return phase(pr, [&] {
        pr->se->evaluate(fr_of(x));
        size_t off = 0;
        for (const auto& e : pr->se->ev) {
            memcpy(evals + off, e.data(), e.size() * 8);
            off += e.size();
        }
    });
}

extern "C" int pz_proof_open_begin(pz_proof* pr, const uint64_t y[4], const uint64_t v[4], uint64_t* w1_affine) {
    if (!pr || !pr->se || !y || !v || !w1_affine || !fr_canonical(y) || !fr_canonical(v)) return PZ_ERR_INVALID;
    return phase(pr, [&] { pr->se->open_begin(fr_of(y), fr_of(v), w1_affine); });
}

extern "C" int pz_proof_open_finish(pz_proof* pr, const uint64_t u[4], uint64_t* w2_affine, int* quotient_degree_ok) {
    if (!pr || !pr->se || !u || !w2_affine || !quotient_degree_ok || !fr_canonical(u)) return PZ_ERR_INVALID;
    return phase(pr, [&] { *quotient_degree_ok = pr->se->open_finish(fr_of(u), w2_affine) ? 1 : 0; });
}

extern "C" int pz_proof_free(pz_proof* pr) {
    if (!pr) return PZ_OK;
    if (pr->pk && pr->pk->mem.c) (void)pz_sync(pr->pk->mem.c);
    delete pr->se;
    if (pr->pk) pr->pk->busy.store(false);
    delete pr;
    return PZ_OK;
}
