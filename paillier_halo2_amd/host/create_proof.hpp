// create_proof.hpp -- the prover's phases composed in create_proof's order from plain C++ over the C ABI (include/pz.h) only: no torch,
// no HIP call of its own.  The compiled-language counterpart of paillier_halo2_amd/prover.py (same phases, same entry points, same
// three-coset quotient: DESIGN.md section 6.3) -- what a reference prover patched at point D of INTEGRATION.md runs between
// /root/reference/src/bench.rs:161 and :171 (bench_builder -> keygen -> gen_proof):
//
//   keygen        fixed columns (selectors, constants, table) and the sigma polynomials of the circuit's copy constraints: commitments,
//                 coefficient forms, extended forms on the quotient's three cosets, resident in HBM
//   create_proof  advice commitments -> permuted lookup columns -> grand products -> quotient (tiles of columns extended and folded as
//                 produced) -> h pieces -> evaluations -> SHPLONK, a transcript round trip (synchronising download + hash) per phase
//
// The circuit STRUCTURE (selectors, copy-constraint map, constants, break points) is an input, as in prover.py: the dependency's keygen
// knows it; tests hand over what paillier_halo2_amd/circuit_structure.py generates.  Device memory through pz_dev_alloc only.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pz.h"
#include "fr_host.hpp"
#include "transcript.hpp"

namespace pzp {
using pzh::Fr;

// a failed entry point: the drivers print and exit; inside the library (PZP_THROW: csrc/pz_prover.cpp) it becomes an exception that the C
// entry point catches and returns as its status -- nothing is thrown across the ABI
struct PzpError {
    int rc;
};
#ifdef PZP_THROW
#define PZP_FAIL(rc_, what_) throw ::pzp::PzpError{rc_}
#else
#define PZP_FAIL(rc_, what_)                                                                               \
    do {                                                                                                   \
        fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, what_, rc_, pz_strerror(rc_));       \
        exit(2);                                                                                           \
    } while (0)
#endif
#define PZP_CK(x)                           \
    do {                                    \
        int rc_ = (x);                      \
        if (rc_ != PZ_OK) PZP_FAIL(rc_, #x); \
    } while (0)

static const unsigned CHUNK = 2;   // permutation columns per grand product: degree - 2

struct Structure {   // what keygen derives from the circuit (INPUT)
    unsigned k = 0, lookup_bits = 0, blinding_factors = 6;
    size_t max_rows = 0, n_adv = 0, n_lk = 0;
    std::vector<uint8_t> selectors;          // [n_adv][2^k]
    std::vector<uint64_t> constants;         // canonical integers, 4 words each
    std::vector<uint32_t> map_col, map_row;  // [m][2^k]
    std::vector<uint64_t> starts;            // n_adv + 1 (the break-point layout K4 writes the advice stream in)
    // the same three arrays ALREADY ON THE DEVICE (pz_pk_create_dev: a structure generated there, or uploaded once by the caller):
    // when set they are used as they are and the host vectors above stay empty -- no host copy, no PCIe crossing
    const uint8_t* d_selectors = nullptr;
    const uint32_t *d_map_col = nullptr, *d_map_row = nullptr;
    size_t m() const { return n_adv + n_lk + 1; }
};

struct Part {   // a part of the quotient's domain (prover.py Domain._parts): part A = g <w_2n>, part B = g w_4n <w_n>
    unsigned log_e;
    size_t size;
    Fr coset_g, omega, omega_inv, size_inv;
    std::vector<uint64_t> gens;   // 2^log_e x 4 words
};

struct Dev {   // a device buffer of Fr elements (or points), freed with the key / workspace
    uint64_t* p = nullptr;
    size_t words = 0;
};

struct Domain {
    unsigned k, bf;
    size_t n, usable;
    Fr omega, omega_inv, n_inv, omega4, coset_g;
    std::vector<Part> parts;
    Domain(unsigned k_, unsigned bf_) : k(k_), bf(bf_) {
        n = (size_t)1 << k;
        usable = n - (bf + 1);
        omega = pzh::omega(k);
        omega_inv = pzh::inv(omega);
        n_inv = pzh::inv(pzh::from_u64(n));
        omega4 = pzh::omega(k + 2);
        coset_g = pzh::zeta();
        auto mk = [&](unsigned log_e, const Fr& cg, const Fr& om) {
            Part p;
            p.log_e = log_e;
            p.size = n << log_e;
            p.coset_g = cg;
            p.omega = om;
            p.omega_inv = pzh::inv(om);
            p.size_inv = pzh::inv(pzh::from_u64(p.size));
            Fr x = cg;
            for (unsigned r = 0; r < (1u << log_e); ++r) {
                p.gens.insert(p.gens.end(), x.v, x.v + 4);
                x = pzh::mul(x, om);
            }
            return p;
        };
        parts.push_back(mk(1, coset_g, pzh::mul(omega4, omega4)));
        parts.push_back(mk(0, pzh::mul(coset_g, omega4), omega));
    }
};

class Ctx {
  public:
    pz_ctx* c = nullptr;
    std::vector<void*> owned;
    // round_cols: column counts of the big arrays are rounded up to 64, so that keys whose column counts differ by a few (a new message of
    // the reference's circuit moves them by a handful) ask for IDENTICAL block sizes -- with the library's block cache on
    // (pz_dev_cache_limit) a released key's blocks then serve the next key instead of going back to the driver (seconds per 116-GB key)
    bool round_cols = false;
    size_t cols(size_t x) const { return round_cols ? (x + 63) / 64 * 64 : x; }
    uint64_t* alloc(size_t words) {
        void* d = nullptr;
        PZP_CK(pz_dev_alloc(c, words * 8, &d));
        owned.push_back(d);
        PZP_CK(pz_dev_memset(c, d, 0, words * 8));
        return (uint64_t*)d;
    }
    void release() {
        for (void* d : owned) pz_dev_free(c, d);
        owned.clear();
    }
};

struct ProvingKey {
    Structure st;
    Domain dom;
    size_t n_sets = 0, F = 0;
    const pz_bases *bl = nullptr, *bm = nullptr;
    uint64_t *fixed_coeff = nullptr, *sigma_coeff = nullptr, *sigma_lagrange = nullptr, *const_lagrange = nullptr, *table_lagrange = nullptr;
    uint64_t *fixed_ext[2] = {nullptr, nullptr}, *sigma_ext[2] = {nullptr, nullptr}, *l_ext[2] = {nullptr, nullptr};
    uint64_t* table_ext[2] = {nullptr, nullptr};   // the lookup table on the quotient's domain: one column, always resident
    // the STREAMED proving key (prover.py ProvingKey.ext_resident_cols): extended forms are resident for selector j < res_fixed and sigma
    // j < res_sigma only; create_proof re-extends the rest per tile from the coefficient forms.  Resident mode: res_fixed = n_adv,
    // res_sigma = m (what halo2's ProvingKey keeps).  The proof is byte for byte the same either way.
    size_t res_fixed = 0, res_sigma = 0;
    bool streamed = false;
    std::vector<uint64_t> fixed_commit, sigma_commit;   // affine, 8 words each
    explicit ProvingKey(Structure&& s) : st(std::move(s)), dom(st.k, st.blinding_factors) {}
};

inline void upload_mont(Ctx& cx, uint64_t* d, const std::vector<Fr>& v) {
    PZP_CK(pz_upload(cx.c, d, v.data(), v.size() * 32));
}

// st is CONSUMED (moved into the key; its large arrays -- selectors, the copy-constraint map -- are released once they are on the device:
// at config c2 they are 0.4 + 2 x 1.6 GB of host memory).  A failing step throws (PZP_THROW) or exits: the key under construction and the
// map's device buffers are released on the way out (the buffers of `cx` are the caller's to release: Ctx::release)
static const size_t EXT_ALL = ~(size_t)0;   // pz.h PZ_PK_EXT_ALL

// device buffers that do not outlive a scope (freed on every path out of it, a throwing PZP_CK included)
struct Scratch {
    pz_ctx* c;
    std::vector<void*> bufs;
    explicit Scratch(pz_ctx* c_) : c(c_) {}
    void* get(size_t bytes) {
        void* d = nullptr;
        PZP_CK(pz_dev_alloc(c, bytes, &d));
        bufs.push_back(d);
        return d;
    }
    ~Scratch() {
        for (void* d : bufs) pz_dev_free(c, d);
    }
};

inline ProvingKey* keygen(Ctx& cx, Structure&& st_in, const pz_bases* bl, const pz_bases* bm, size_t ext_resident_cols = EXT_ALL) {
    std::unique_ptr<ProvingKey> pk(new ProvingKey(std::move(st_in)));
    const Structure& st = pk->st;
    Domain& d = pk->dom;
    const size_t n = d.n, A = st.n_adv, m = st.m(), F = A + 2;
    pk->bl = bl; pk->bm = bm; pk->F = F;
    pk->n_sets = (m + CHUNK - 1) / CHUNK;
    pk->streamed = ext_resident_cols < m;
    pk->res_sigma = pk->streamed ? ext_resident_cols : m;
    pk->res_fixed = pk->res_sigma < A ? pk->res_sigma : A;
    // fixed columns [selectors | constants | table], Lagrange form.  The selectors are bytes (0 / 1): they cross PCIe as bytes (or are on the
    // device already) and become field elements THERE (pz_fr_from_mask_dev) -- built on the host as 32-byte elements they were 12.7 GB of
    // uploads at config c2, most of pz_pk_create's 5 s
    uint64_t* fixed = cx.alloc(cx.cols(F) * n * 4);
    {
        Scratch tmp(cx.c);
        const uint8_t* d_sel = st.d_selectors;
        if (!d_sel) {
            void* up = tmp.get(A * n);
            PZP_CK(pz_upload(cx.c, up, st.selectors.data(), A * n));
            d_sel = (const uint8_t*)up;
        }
        PZP_CK(pz_fr_from_mask_dev(cx.c, d_sel, A * n, fixed));
        PZP_CK(pz_sync(cx.c));
    }
    {
        std::vector<Fr> col(n);
        const Fr zero = {{0, 0, 0, 0}};
        for (size_t i = 0; i < n; ++i) col[i] = i < st.constants.size() / 4 ? pzh::from_raw(&st.constants[4 * i]) : zero;
        upload_mont(cx, fixed + A * n * 4, col);
        for (size_t i = 0; i < n; ++i) col[i] = i < ((size_t)1 << st.lookup_bits) ? pzh::from_u64(i) : zero;
        upload_mont(cx, fixed + (A + 1) * n * 4, col);
    }
    pk->const_lagrange = cx.alloc(n * 4);
    pk->table_lagrange = cx.alloc(n * 4);
    PZP_CK(pz_dev_copy(cx.c, pk->const_lagrange, fixed + A * n * 4, n * 32));
    PZP_CK(pz_dev_copy(cx.c, pk->table_lagrange, fixed + (A + 1) * n * 4, n * 32));
    // sigma from the copy-constraint map (one call over all m columns)
    uint64_t* sigma = cx.alloc(cx.cols(m) * n * 4);
    {
        Scratch tmp(cx.c);
        const uint32_t *dmc = st.d_map_col, *dmr = st.d_map_row;
        if (!dmc || !dmr) {
            void *mc = tmp.get(m * n * 4), *mr = tmp.get(m * n * 4);
            PZP_CK(pz_upload(cx.c, mc, st.map_col.data(), m * n * 4));
            PZP_CK(pz_upload(cx.c, mr, st.map_row.data(), m * n * 4));
            dmc = (const uint32_t*)mc;
            dmr = (const uint32_t*)mr;
        }
        const Fr delta = pzh::delta();
        PZP_CK(pz_permutation_sigma_dev(cx.c, dmc, dmr, m, st.k, d.omega.v, delta.v, sigma, 4 * n));
        PZP_CK(pz_sync(cx.c));
    }
    {   // the host copies of the structure's large arrays are done with
        Structure& s_ = pk->st;
        std::vector<uint8_t>().swap(s_.selectors);
        std::vector<uint32_t>().swap(s_.map_col);
        std::vector<uint32_t>().swap(s_.map_row);
    }
    pk->sigma_lagrange = cx.alloc(cx.cols(m) * n * 4);
    PZP_CK(pz_dev_copy(cx.c, pk->sigma_lagrange, sigma, m * n * 32));
    // keygen_vk + keygen_pk: commitments, coefficient forms in place, extended forms per part
    uint64_t* com_f = cx.alloc(cx.cols(F) * 12);
    uint64_t* com_s = cx.alloc(cx.cols(m) * 12);
    for (int pi = 0; pi < 2; ++pi) {
        pk->fixed_ext[pi] = cx.alloc((pk->res_fixed ? cx.cols(pk->res_fixed) : 1) * d.parts[pi].size * 4);
        pk->sigma_ext[pi] = cx.alloc((pk->res_sigma ? cx.cols(pk->res_sigma) : 1) * d.parts[pi].size * 4);
        pk->l_ext[pi] = cx.alloc(3 * d.parts[pi].size * 4);
        pk->table_ext[pi] = cx.alloc(d.parts[pi].size * 4);
    }
    struct Job { uint64_t* cols; size_t cnt; uint64_t* com; uint64_t** ext; size_t res; };
    Job jobs[2] = {{fixed, F, com_f, pk->fixed_ext, pk->res_fixed}, {sigma, m, com_s, pk->sigma_ext, pk->res_sigma}};
    for (auto& jb : jobs) {
        for (size_t c0 = 0; c0 < jb.cnt; c0 += 256) {
            const size_t cnt = jb.cnt - c0 < 256 ? jb.cnt - c0 : 256;
            // a batch that straddles the resident prefix runs as two calls: with and without the extended forms kept
            const size_t r_here = jb.res <= c0 ? 0 : (jb.res - c0 < cnt ? jb.res - c0 : cnt);
            const size_t lo[2] = {c0, c0 + r_here}, bc[2] = {r_here, cnt - r_here};
            for (int keep = 1; keep >= 0; --keep) {
                const size_t b0 = lo[1 - keep], bn = bc[1 - keep];
                if (!bn) continue;
                const Part& A_ = d.parts[0];
                PZP_CK(pz_keygen_columns_dev(cx.c, bl, jb.cols + b0 * n * 4, bn, 4 * n, st.k, A_.log_e, d.omega.v, d.omega_inv.v, d.n_inv.v,
                                             A_.gens.data(), jb.com + b0 * 12, keep ? jb.ext[0] + b0 * A_.size * 4 : nullptr, 4 * A_.size));
                if (!keep) continue;
                const Part& B_ = d.parts[1];
                PZP_CK(pz_ntt_fr_extend_dev(cx.c, jb.cols + b0 * n * 4, bn, 4 * n, jb.ext[1] + b0 * B_.size * 4, 4 * B_.size, st.k, B_.log_e, d.omega.v,
                                            B_.gens.data(), nullptr));
            }
        }
    }
    pk->fixed_coeff = fixed;
    pk->sigma_coeff = sigma;
    for (int pi = 0; pi < 2; ++pi)
        PZP_CK(pz_ntt_fr_extend_dev(cx.c, fixed + (A + 1) * n * 4, 1, 4 * n, pk->table_ext[pi], 4 * d.parts[pi].size, st.k, d.parts[pi].log_e,
                                    d.omega.v, d.parts[pi].gens.data(), nullptr));
    // l_0, l_last, l_active
    {
        uint64_t* lrows = cx.alloc(3 * n * 4);
        std::vector<Fr> col(n);
        const Fr zero = {{0, 0, 0, 0}};
        for (int w = 0; w < 3; ++w) {
            for (size_t i = 0; i < n; ++i) col[i] = (w == 0 ? i == 0 : w == 1 ? i == d.usable : i < d.usable) ? pzh::FR_ONE : zero;
            upload_mont(cx, lrows + (size_t)w * n * 4, col);
        }
        PZP_CK(pz_ntt_fr_dev(cx.c, lrows, 3, 4 * n, d.omega_inv.v, st.k, nullptr, d.n_inv.v));
        for (int pi = 0; pi < 2; ++pi)
            PZP_CK(pz_ntt_fr_extend_dev(cx.c, lrows, 3, 4 * n, pk->l_ext[pi], 4 * d.parts[pi].size, st.k, d.parts[pi].log_e, d.omega.v,
                                        d.parts[pi].gens.data(), nullptr));
    }
    // the verifying key's commitments, affine
    std::vector<uint64_t> jac(12 * (F > m ? F : m));
    pk->fixed_commit.resize(8 * F);
    pk->sigma_commit.resize(8 * m);
    PZP_CK(pz_download(cx.c, jac.data(), com_f, F * 96));
    PZP_CK(pz_g1_normalize(cx.c, jac.data(), F, pk->fixed_commit.data()));
    PZP_CK(pz_download(cx.c, jac.data(), com_s, m * 96));
    PZP_CK(pz_g1_normalize(cx.c, jac.data(), m, pk->sigma_commit.data()));
    return pk.release();
}

struct Proof {
    std::vector<std::pair<std::string, std::vector<uint64_t>>> commitments;   // family -> affine points (8 words each)
    std::vector<std::pair<std::string, std::vector<uint64_t>>> evals;         // family -> [count][points][4] Montgomery
    std::vector<std::pair<std::string, uint32_t>> eval_points;                // family -> points per polynomial
    bool h_degree_ok = false;
};

struct Workspace {
    uint64_t *Ap, *Sp, *Zl, *Z, *z_ext[2], *ext[2], *lk_ext[2][4], *hh[2], *hp[2], *h, *tmp, *rnd, *hcomb, *w1, *w2, *blind, *out12, *evals;
    uint64_t* key_ext[2] = {nullptr, nullptr};   // streamed proving key: the tile of selectors / of sigma columns re-extended per step
    size_t tile, lt;
};

inline Workspace make_workspace(Ctx& cx, const ProvingKey& pk, size_t tile = 64) {
    // (with cx.round_cols the per-proof buffers below are sized by rounded counts too, so a following key finds them in the block cache)
    const Domain& d = pk.dom;
    const size_t n = d.n, Lk = pk.st.n_lk, S = pk.n_sets, m = pk.st.m();
    Workspace w;
    w.tile = tile;
    w.lt = tile < Lk ? tile : Lk;
    w.Ap = cx.alloc(cx.cols(Lk) * n * 4); w.Sp = cx.alloc(cx.cols(Lk) * n * 4); w.Zl = cx.alloc(cx.cols(Lk) * n * 4);
    w.Z = cx.alloc(cx.cols(S) * n * 4);
    // the grand products on the quotient's domain: all sets of ONE part at once (the chaining lines read z_{j-1} beside z_j); the parts are
    // worked one after the other, so one buffer of the larger part serves both
    const size_t big = d.parts[0].size > d.parts[1].size ? d.parts[0].size : d.parts[1].size;
    w.z_ext[0] = w.z_ext[1] = cx.alloc(cx.cols(S) * big * 4);
    if (pk.streamed)
        for (int q = 0; q < 2; ++q) w.key_ext[q] = cx.alloc(tile * big * 4);
    for (int pi = 0; pi < 2; ++pi) {
        const size_t Np = d.parts[pi].size;
        w.ext[pi] = cx.alloc(tile * Np * 4);
        for (int q = 0; q < 4; ++q) w.lk_ext[pi][q] = cx.alloc(w.lt * Np * 4);
        w.hh[pi] = cx.alloc(2 * Np * 4);
        w.hp[pi] = cx.alloc(Np * 4);
    }
    w.h = cx.alloc(4 * n * 4);
    w.tmp = cx.alloc(2 * n * 4);
    w.rnd = cx.alloc(n * 4);
    w.hcomb = cx.alloc(n * 4);
    w.w1 = cx.alloc(n * 4);
    w.w2 = cx.alloc(n * 4);
    w.blind = cx.alloc((cx.cols(m) + cx.cols(S) + 3 * cx.cols(Lk)) * (d.bf + 1) * 4 + n * 4);
    {   // the largest batch of commitments a phase leaves there: advice (m - 1), lookups (2 Lk), products (S + Lk + 1), the pieces (3)
        size_t pts = m - 1;
        if (2 * Lk > pts) pts = 2 * Lk;
        if (S + Lk + 1 > pts) pts = S + Lk + 1;
        if (pts < 3) pts = 3;
        w.out12 = cx.alloc(cx.cols(pts) * 12);
    }
    w.evals = cx.alloc((4 * cx.cols(m) + 4 * cx.cols(S) + 16) * 4 * 4);
    return w;
}

// blinding values: UNIFORM field elements (as Montgomery representatives: any value below r is one), drawn by rejection from 254-bit
// candidates (accepted with probability r / 2^254 = 0.756) -- from the caller's random words when given (a production prover hands over
// OS randomness: pz_pk_info says how many, margin included), else from a seeded xorshift (tests, benches: not zero-knowledge)
struct Rng {
    uint64_t s;
    const uint64_t* words = nullptr;
    size_t n_words = 0, used = 0;
    explicit Rng(uint64_t seed) : s(seed * 0x9e3779b97f4a7c15ULL + 1) {}
    Rng(const uint64_t* w, size_t n) : s(1), words(w), n_words(n) {}
    uint64_t next() {
        if (words) {
            if (used >= n_words) PZP_FAIL(PZ_ERR_INVALID, "blinding words exhausted");
            return words[used++];
        }
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return s;
    }
    void fill(std::vector<uint64_t>& v) {   // v: 4 words per element
        for (size_t i = 0; i + 3 < v.size(); i += 4) {
            for (;;) {
                for (int j = 0; j < 4; ++j) v[i + j] = next();
                v[i + 3] &= 0x3fffffffffffffffULL;
                bool below = false;
                for (int j = 3; j >= 0; --j)
                    if (v[i + j] != pzh::FR_MOD[j]) {
                        below = v[i + j] < pzh::FR_MOD[j];
                        break;
                    }
                if (below) break;
            }
        }
    }
};
// 64-bit words of randomness one proof may consume (blinding rows of the advice / permuted / product columns + the random polynomial):
// twice the words of the elements drawn -- the rejection sampling takes 1.32 draws per element on average, and a factor 2 over tens of
// thousands of elements is out of reach of its fluctuations (a caller's array that does run out: PZ_ERR_INVALID, nothing reused)
inline size_t blinding_words(const ProvingKey& pk) {
    const size_t b = pk.dom.bf + 1, W = pk.st.n_adv + pk.st.n_lk, Lk = pk.st.n_lk, S = pk.n_sets;
    return 2 * 4 * (W * b + 2 * Lk * b + (S + Lk) * (b - 1) + pk.dom.n);
}

// rows [row0, n) of `count` columns (stride n elements) <- random elements
inline void blind_rows(Ctx& cx, Workspace& w, Rng& rng, uint64_t* d_cols, size_t count, size_t n, size_t row0) {
    const size_t rows = n - row0;
    std::vector<uint64_t> host(count * rows * 4);
    rng.fill(host);
    PZP_CK(pz_upload(cx.c, w.blind, host.data(), host.size() * 8));
    PZP_CK(pz_dev_copy_2d(cx.c, d_cols + row0 * 4, n * 32, w.blind, rows * 32, rows * 32, count));
}

inline Fr pow_small(const Fr& a, uint64_t e) { return pzh::pow_u64(a, e); }

// the families of polynomials a proof opens, in the order their evaluations are handed over (and absorbed); points = indices into the six
// rotation points {x, wx, w^2 x, w^3 x, w^-(bf+1) x, w^-1 x}.  "lookup_advice" carries the constants column as its last polynomial.
struct Fam {
    const char* name;
    const uint64_t* polys;
    size_t count;
    std::vector<int> idx;
};

// ONE PROOF IN FLIGHT, phase by phase: each method runs a phase on the device and hands back what the transcript must absorb before the
// next challenge exists (affine commitments, 8 words each; evaluations, 4 words each, Montgomery).  Methods must be called in order.
// d_cols: [m][2^k] elements: advice then lookup-advice columns as K4 wrote them; consumed (ends in coefficient form).
struct Session {
    Ctx& cx;
    ProvingKey& pk;
    Workspace& w;
    uint64_t* d_cols;
    Rng rng;
    Fr beta, gamma, y, x, xs[6];
    uint64_t* pieces = nullptr;
    std::vector<Fam> fams;
    std::vector<std::vector<uint64_t>> ev;
    pz_shplonk* state = nullptr;
    int phase = 0;

    Session(Ctx& cx_, ProvingKey& pk_, Workspace& w_, uint64_t* d_cols_, Rng rng_) : cx(cx_), pk(pk_), w(w_), d_cols(d_cols_), rng(rng_) {}
    ~Session() {
        if (state) pz_shplonk_free(cx.c, state);
    }
    // phases run in order, each once: a phase is entered only from its predecessor's COMPLETION (done(p) is the last statement of phase
    // p), so after a phase that failed half-way every later call is refused instead of running on half-built state
    void expect(int p) {
        if (phase != p) PZP_FAIL(PZ_ERR_INVALID, "proof phases out of order");
        phase = -1;   // in progress: re-entering or skipping ahead is refused until done(p)
    }
    void done(int p) { phase = p + 1; }
    void commit(const pz_bases* b, const uint64_t* t, size_t count, uint64_t* out) {
        uint32_t nw = 0, cb = 0;
        size_t np = 0;
        PZP_CK(pz_bases_info(b, &np, &cb, &nw));
        PZP_CK(pz_msm_g1_dev(cx.c, b, t, count, pk.dom.n, 4 * pk.dom.n, 0, nw, out));
    }
    void affine(const uint64_t* d_jac, size_t count, uint64_t* out) {   // the synchronising download of a phase's commitments
        std::vector<uint64_t> jac(12 * count);
        PZP_CK(pz_download(cx.c, jac.data(), d_jac, count * 96));
        PZP_CK(pz_g1_normalize(cx.c, jac.data(), count, out));
    }

    // ---- 1. advice: out (n_adv + n_lk) x 8.  after_launch: called once the commitments are queued and before the host waits for them --
    // the caller's chance to queue independent work on ANOTHER context (the next proof's K3 + K4) under this proof's largest batch
    void advice(uint64_t* out_affine, const std::function<void()>& after_launch = nullptr) {
        expect(0);
        const Domain& d = pk.dom;
        const size_t n = d.n, W = pk.st.n_adv + pk.st.n_lk;
        blind_rows(cx, w, rng, d_cols, W, n, d.usable);
        PZP_CK(pz_dev_copy(cx.c, d_cols + W * n * 4, pk.const_lagrange, n * 32));
        commit(pk.bl, d_cols, W, w.out12);
        if (after_launch) after_launch();
        affine(w.out12, W, out_affine);
        done(0);
    }
    // ---- 2. lookups (one expression each side: theta does not enter).  out: n_lk x 8 each
    void lookups(uint64_t* out_inputs, uint64_t* out_tables) {
        expect(1);
        const Domain& d = pk.dom;
        const size_t n = d.n, Lk = pk.st.n_lk;
        uint64_t* lk_in = d_cols + pk.st.n_adv * n * 4;
        PZP_CK(pz_lookup_permute_dev(cx.c, lk_in, Lk, 4 * n, pk.table_lagrange, d.usable, pk.st.lookup_bits, w.Ap, w.Sp, 4 * n));
        blind_rows(cx, w, rng, w.Ap, Lk, n, d.usable);
        blind_rows(cx, w, rng, w.Sp, Lk, n, d.usable);
        commit(pk.bl, w.Ap, Lk, w.out12);
        commit(pk.bl, w.Sp, Lk, w.out12 + Lk * 12);
        affine(w.out12, Lk, out_inputs);
        affine(w.out12 + Lk * 12, Lk, out_tables);
        done(1);
    }
    // ---- 3. grand products + the vanishing argument's random polynomial.  out: n_sets x 8, n_lk x 8, 8
    void products(const Fr& beta_, const Fr& gamma_, uint64_t* out_z, uint64_t* out_zl, uint64_t* out_random) {
        expect(2);
        beta = beta_; gamma = gamma_;
        const Domain& d = pk.dom;
        const size_t n = d.n, u = d.usable, Lk = pk.st.n_lk, S = pk.n_sets;
        const Fr delta = pzh::delta();
        uint64_t* lk_in = d_cols + pk.st.n_adv * n * 4;
        PZP_CK(pz_permutation_product_sets_dev(cx.c, d_cols, 4 * n, pk.sigma_lagrange, 4 * n, pk.st.m(), CHUNK, d.k, u, d.omega.v, beta.v, gamma.v,
                                               delta.v, w.Z, 4 * n));
        blind_rows(cx, w, rng, w.Z, S, n, u + 1);
        PZP_CK(pz_lookup_product_dev(cx.c, lk_in, 4 * n, pk.table_lagrange, w.Ap, 4 * n, w.Sp, 4 * n, Lk, n, beta.v, gamma.v, pzh::FR_ONE.v, w.Zl,
                                     4 * n));
        blind_rows(cx, w, rng, w.Zl, Lk, n, u + 1);
        blind_rows(cx, w, rng, w.rnd, 1, n, 0);
        commit(pk.bl, w.Z, S, w.out12);
        commit(pk.bl, w.Zl, Lk, w.out12 + S * 12);
        commit(pk.bm, w.rnd, 1, w.out12 + (S + Lk) * 12);
        affine(w.out12, S, out_z);
        affine(w.out12 + S * 12, Lk, out_zl);
        affine(w.out12 + (S + Lk) * 12, 1, out_random);
        done(2);
    }
    // ---- 4. quotient.  out: 3 x 8 (the pieces h_0, h_1, h_2)
    void quotient(const Fr& y_, uint64_t* out_h) {
        expect(3);
        y = y_;
        const Structure& st = pk.st;
        const Domain& d = pk.dom;
        const size_t n = d.n, A = st.n_adv, Lk = st.n_lk, m = st.m(), S = pk.n_sets, tile = w.tile;
        const unsigned k = d.k, bf = d.bf;
        const Fr delta = pzh::delta();
        auto to_coeff = [&](uint64_t* t, size_t cnt) {
            for (size_t c0 = 0; c0 < cnt; c0 += tile)
                PZP_CK(pz_ntt_fr_dev(cx.c, t + c0 * n * 4, cnt - c0 < tile ? cnt - c0 : tile, 4 * n, d.omega_inv.v, k, nullptr, d.n_inv.v));
        };
        to_coeff(d_cols, m); to_coeff(w.Ap, Lk); to_coeff(w.Sp, Lk); to_coeff(w.Z, S); to_coeff(w.Zl, Lk);
        const size_t n_perm_lines = 2 + (S - 1) + S;
        const Fr y_lines = pow_small(y, n_perm_lines);
        for (int pi = 0; pi < 2; ++pi) {
            const Part& pt = d.parts[pi];
            const size_t Np = pt.size;
            const unsigned lg = k + pt.log_e, rot = 1u << pt.log_e;
            auto extend = [&](const uint64_t* src, size_t cnt, uint64_t* dst) {
                PZP_CK(pz_ntt_fr_extend_dev(cx.c, src, cnt, 4 * n, dst, 4 * Np, k, pt.log_e, d.omega.v, pt.gens.data(), nullptr));
            };
            for (size_t s0 = 0; s0 < S; s0 += tile) extend(w.Z + s0 * n * 4, S - s0 < tile ? S - s0 : tile, w.z_ext[pi] + s0 * Np * 4);
            PZP_CK(pz_dev_memset(cx.c, w.hh[pi], 0, 2 * Np * 32));
            uint64_t *hg = w.hh[pi], *hp = w.hh[pi] + Np * 4;
            const uint64_t *l0 = pk.l_ext[pi], *llast = pk.l_ext[pi] + Np * 4, *lact = pk.l_ext[pi] + 2 * Np * 4;
            // the extended forms of key columns [c0, c0 + cnt): resident, or (streamed proving key) re-extended into the tile buffer
            auto key_tile = [&](const uint64_t* coeff, const uint64_t* resident, size_t res, int which, size_t c0, size_t cnt) -> const uint64_t* {
                if (c0 + cnt <= res) return resident + c0 * Np * 4;
                extend(coeff + c0 * n * 4, cnt, w.key_ext[which]);
                return w.key_ext[which];
            };
            for (size_t c0 = 0; c0 < m; c0 += tile) {
                const size_t cnt = m - c0 < tile ? m - c0 : tile;
                extend(d_cols + c0 * n * 4, cnt, w.ext[pi]);
                const size_t na = c0 >= A ? 0 : (A - c0 < cnt ? A - c0 : cnt);
                if (na)
                    PZP_CK(pz_quotient_gate_dev(cx.c, w.ext[pi], 4 * Np, key_tile(pk.fixed_coeff, pk.fixed_ext[pi], pk.res_fixed, 0, c0, na), 4 * Np, na,
                                                lg, rot, y.v, hg));
                PZP_CK(pz_quotient_permutation_part_dev(cx.c, w.ext[pi], 4 * Np, key_tile(pk.sigma_coeff, pk.sigma_ext[pi], pk.res_sigma, 1, c0, cnt),
                                                        4 * Np, w.z_ext[pi], 4 * Np,
                                                        (uint32_t)S, (uint32_t)(c0 / CHUNK), (uint32_t)((cnt + CHUNK - 1) / CHUNK), CHUNK, (uint32_t)cnt,
                                                        c0 == 0, lg, rot, bf + 1, l0, llast, lact, beta.v, gamma.v, delta.v, pt.coset_g.v, pt.omega.v,
                                                        y.v, hp));
            }
            uint64_t* hq = w.hp[pi];
            PZP_CK(pz_fr_lincomb_dev(cx.c, w.hh[pi], 2, 4 * Np, Np, y_lines.v, hq, 0));
            for (size_t l0_ = 0; l0_ < Lk; l0_ += w.lt) {
                const size_t cnt = Lk - l0_ < w.lt ? Lk - l0_ : w.lt;
                extend(d_cols + (A + l0_) * n * 4, cnt, w.lk_ext[pi][0]);
                extend(w.Ap + l0_ * n * 4, cnt, w.lk_ext[pi][1]);
                extend(w.Sp + l0_ * n * 4, cnt, w.lk_ext[pi][2]);
                extend(w.Zl + l0_ * n * 4, cnt, w.lk_ext[pi][3]);
                PZP_CK(pz_quotient_lookup_dev(cx.c, w.lk_ext[pi][0], 4 * Np, pk.table_ext[pi], w.lk_ext[pi][1], 4 * Np,
                                              w.lk_ext[pi][2], 4 * Np, w.lk_ext[pi][3], 4 * Np, (uint32_t)cnt, lg, rot, l0, llast, lact, beta.v, gamma.v,
                                              y.v, hq));
            }
            PZP_CK(pz_quotient_finish_dev(cx.c, hq, k, pt.log_e, pt.coset_g.v, pt.omega.v));
            PZP_CK(pz_ntt_fr_dev(cx.c, hq, 1, 4 * Np, pt.omega_inv.v, lg, nullptr, pt.size_inv.v));
            const Fr cg_inv = pzh::inv(pt.coset_g);
            PZP_CK(pz_fr_distribute_powers_dev(cx.c, hq, 1, 4 * Np, Np, cg_inv.v, nullptr));
        }
        // the quotient from three cosets (prover.py): [U | h_1] on part A, V on part B
        pieces = w.h;
        {
            const Fr g2n = pow_small(pzh::mul(d.coset_g, d.coset_g), n);
            const Fr lam = pow_small(d.parts[1].coset_g, n);
            uint64_t *U = w.hp[0], *h1 = w.hp[0] + n * 4, *V = w.hp[1];
            uint64_t *t0 = w.tmp, *t1 = w.tmp + n * 4;
            PZP_CK(pz_dev_copy(cx.c, pieces + n * 4, h1, n * 32));
            PZP_CK(pz_dev_copy(cx.c, t0, h1, n * 32));
            PZP_CK(pz_dev_copy(cx.c, t1, V, n * 32));
            const Fr mlam = pzh::neg(lam), m1 = pzh::neg(pzh::FR_ONE), mg2n = pzh::neg(g2n);
            PZP_CK(pz_fr_lincomb_dev(cx.c, w.tmp, 2, 4 * n, n, mlam.v, pieces + 2 * n * 4, 0));      // T = V - lam h_1
            PZP_CK(pz_dev_copy(cx.c, t0, pieces + 2 * n * 4, n * 32));
            PZP_CK(pz_dev_copy(cx.c, t1, U, n * 32));
            PZP_CK(pz_fr_lincomb_dev(cx.c, w.tmp, 2, 4 * n, n, m1.v, pieces + 2 * n * 4, 0));        // U - T
            const Fr half = pzh::inv(pzh::add(g2n, g2n));
            PZP_CK(pz_fr_distribute_powers_dev(cx.c, pieces + 2 * n * 4, 1, 4 * n, n, pzh::FR_ONE.v, half.v));   // h_2
            PZP_CK(pz_dev_copy(cx.c, t0, pieces + 2 * n * 4, n * 32));
            PZP_CK(pz_dev_copy(cx.c, t1, U, n * 32));
            PZP_CK(pz_fr_lincomb_dev(cx.c, w.tmp, 2, 4 * n, n, mg2n.v, pieces, 0));                  // h_0 = U - g^2n h_2
        }
        commit(pk.bm, pieces, 3, w.out12);
        affine(w.out12, 3, out_h);
        done(3);
    }
    // ---- 5. evaluations at x and its rotations: ev[f] = [count][points][4] per family of `fams`
    void evaluate(const Fr& x_) {
        expect(4);
        x = x_;
        const Structure& st = pk.st;
        const Domain& d = pk.dom;
        const size_t n = d.n, A = st.n_adv, Lk = st.n_lk, m = st.m(), S = pk.n_sets;
        xs[0] = x;
        xs[1] = pzh::mul(x, d.omega);
        xs[2] = pzh::mul(xs[1], d.omega);
        xs[3] = pzh::mul(xs[2], d.omega);
        xs[4] = pzh::mul(x, pow_small(d.omega_inv, d.bf + 1));
        xs[5] = pzh::mul(x, d.omega_inv);
        const Fr xn = pow_small(x, n);
        PZP_CK(pz_fr_lincomb_dev(cx.c, pieces + 2 * n * 4, 1, 4 * n, n, xn.v, w.hcomb, 0));
        PZP_CK(pz_fr_lincomb_dev(cx.c, pieces + 1 * n * 4, 1, 4 * n, n, xn.v, w.hcomb, 1));
        PZP_CK(pz_fr_lincomb_dev(cx.c, pieces, 1, 4 * n, n, xn.v, w.hcomb, 1));
        fams = {{"advice", d_cols, A, {0, 1, 2, 3}}, {"lookup_advice", d_cols + A * n * 4, Lk + 1, {0}}, {"fixed", pk.fixed_coeff, pk.F, {0}},
                {"sigma", pk.sigma_coeff, m, {0}}, {"perm_z", w.Z, S, {0, 1, 4}}, {"lookup_z", w.Zl, Lk, {0, 1}}, {"perm_inputs", w.Ap, Lk, {0, 5}},
                {"perm_tables", w.Sp, Lk, {0}}, {"random", w.rnd, 1, {0}}, {"h", w.hcomb, 1, {0}}};
        ev.assign(fams.size(), {});
        for (size_t f = 0; f < fams.size(); ++f) {
            const Fam& fm = fams[f];
            std::vector<uint64_t> pts;
            for (int i : fm.idx) pts.insert(pts.end(), xs[i].v, xs[i].v + 4);
            PZP_CK(pz_poly_eval_multi_dev(cx.c, fm.polys, fm.count, 4 * n, n, pts.data(), (uint32_t)fm.idx.size(), w.evals));
            ev[f].resize(fm.count * fm.idx.size() * 4);
            PZP_CK(pz_download(cx.c, ev[f].data(), w.evals, ev[f].size() * 8));
        }
        done(4);
    }
    // ---- 6. SHPLONK: the rotation sets in prover.py's query_layout order.  out: 8 each
    void open_begin(const Fr& shy, const Fr& shv, uint64_t* out_w1) {
        expect(5);
        const size_t n = pk.dom.n, A = pk.st.n_adv, Lk = pk.st.n_lk, m = pk.st.m(), S = pk.n_sets, F = pk.F;
        auto fam_of = [&](const char* nm) -> size_t {
            for (size_t f = 0; f < fams.size(); ++f)
                if (!strcmp(fams[f].name, nm)) return f;
            return 0;
        };
        struct Member { size_t fam, idx; };
        std::vector<std::pair<std::vector<uint32_t>, std::vector<Member>>> sets;
        {
            std::vector<Member> s0;
            for (size_t i = 0; i < Lk; ++i) s0.push_back({fam_of("lookup_advice"), i});
            for (size_t i = 0; i < F; ++i) s0.push_back({fam_of("fixed"), i});
            for (size_t i = 0; i < m; ++i) s0.push_back({fam_of("sigma"), i});
            for (size_t i = 0; i < Lk; ++i) s0.push_back({fam_of("perm_tables"), i});
            s0.push_back({fam_of("h"), 0});
            s0.push_back({fam_of("random"), 0});
            sets.push_back({{0}, s0});
            std::vector<Member> s1;
            for (size_t i = 0; i < A; ++i) s1.push_back({fam_of("advice"), i});
            sets.push_back({{0, 1, 2, 3}, s1});
            if (S > 1) {
                std::vector<Member> s2;
                for (size_t i = 0; i + 1 < S; ++i) s2.push_back({fam_of("perm_z"), i});
                sets.push_back({{0, 1, 4}, s2});
            }
            std::vector<Member> s3;
            s3.push_back({fam_of("perm_z"), S - 1});
            for (size_t i = 0; i < Lk; ++i) s3.push_back({fam_of("lookup_z"), i});
            sets.push_back({{0, 1}, s3});
            std::vector<Member> s4;
            for (size_t i = 0; i < Lk; ++i) s4.push_back({fam_of("perm_inputs"), i});
            sets.push_back({{0, 5}, s4});
        }
        std::vector<uint32_t> set_n_polys, set_n_points, point_idx;
        std::vector<const uint64_t*> polys;
        std::vector<uint64_t> evals_flat, points;
        for (int i = 0; i < 6; ++i) points.insert(points.end(), xs[i].v, xs[i].v + 4);
        for (auto& sp : sets) {
            set_n_polys.push_back((uint32_t)sp.second.size());
            set_n_points.push_back((uint32_t)sp.first.size());
            for (uint32_t pi_ : sp.first) point_idx.push_back(pi_);
            for (const Member& mb : sp.second) {
                const Fam& fm = fams[mb.fam];
                polys.push_back(fm.polys + mb.idx * n * 4);
                const size_t npts = fm.idx.size();
                for (size_t q = 0; q < sp.first.size(); ++q)   // the set's points are a prefix of the family's
                    evals_flat.insert(evals_flat.end(), &ev[mb.fam][(mb.idx * npts + q) * 4], &ev[mb.fam][(mb.idx * npts + q) * 4] + 4);
            }
        }
        PZP_CK(pz_shplonk_begin_dev(cx.c, n, (uint32_t)sets.size(), set_n_polys.data(), polys.data(), set_n_points.data(), point_idx.data(), 6,
                                    points.data(), evals_flat.data(), shy.v, shv.v, w.w1, &state));
        commit(pk.bm, w.w1, 1, w.out12);
        affine(w.out12, 1, out_w1);
        done(5);
    }
    // -> whether the quotient has degree <= 3n - 4 (the top three coefficients of h_2 vanish): false for an unsatisfied witness
    bool open_finish(const Fr& shu, uint64_t* out_w2) {
        expect(6);
        const size_t n = pk.dom.n;
        pz_shplonk* st_ = state;
        state = nullptr;                                   // pz_shplonk_finish_dev frees it on every path
        PZP_CK(pz_shplonk_finish_dev(cx.c, st_, shu.v, w.w1, w.w2));
        commit(pk.bm, w.w2, 1, w.out12);
        affine(w.out12, 1, out_w2);
        uint64_t top[12];
        PZP_CK(pz_download(cx.c, top, pieces + (3 * n - 3) * 4, 96));
        done(6);
        for (int i = 0; i < 12; ++i)
            if (top[i]) return false;
        return true;
    }
};

// the phases behind a transcript (the drivers' form): every phase's hand-over is absorbed, the next challenge squeezed
inline Proof create_proof(Ctx& cx, ProvingKey& pk, Workspace& w, uint64_t* d_cols, Transcript& tr, uint64_t seed,
                          const std::function<void()>& after_advice_launch = nullptr) {
    const size_t A = pk.st.n_adv, Lk = pk.st.n_lk, S = pk.n_sets, W = A + Lk;
    Session se(cx, pk, w, d_cols, Rng(seed));
    Proof pr;
    auto keep = [&](const char* name, const std::vector<uint64_t>& aff) {
        tr.common_points(aff.data(), aff.size() / 8);
        pr.commitments.push_back({name, aff});
    };
    std::vector<uint64_t> a(8 * W), b, c;
    se.advice(a.data(), after_advice_launch);
    tr.common_points(a.data(), W);
    pr.commitments.push_back({"advice", std::vector<uint64_t>(a.begin(), a.begin() + 8 * A)});
    pr.commitments.push_back({"lookup_advice", std::vector<uint64_t>(a.begin() + 8 * A, a.end())});
    tr.squeeze("theta");
    a.assign(8 * Lk, 0); b.assign(8 * Lk, 0);
    se.lookups(a.data(), b.data());
    keep("perm_inputs", a);
    keep("perm_tables", b);
    const Fr beta = tr.squeeze("beta"), gamma = tr.squeeze("gamma");
    a.assign(8 * S, 0); b.assign(8 * Lk, 0); c.assign(8, 0);
    se.products(beta, gamma, a.data(), b.data(), c.data());
    keep("perm_z", a);
    keep("lookup_z", b);
    keep("random", c);
    const Fr y = tr.squeeze("y");
    a.assign(24, 0);
    se.quotient(y, a.data());
    keep("h", a);
    const Fr x = tr.squeeze("x");
    se.evaluate(x);
    for (size_t f = 0; f < se.fams.size(); ++f) {
        pr.evals.push_back({se.fams[f].name, se.ev[f]});
        pr.eval_points.push_back({se.fams[f].name, (uint32_t)se.fams[f].idx.size()});
        if (strcmp(se.fams[f].name, "h")) tr.common_scalars(se.ev[f].data(), se.ev[f].size() / 4);
    }
    const Fr shy = tr.squeeze("sh_y"), shv = tr.squeeze("sh_v");
    a.assign(8, 0);
    se.open_begin(shy, shv, a.data());
    keep("w1", a);
    const Fr shu = tr.squeeze("sh_u");
    pr.h_degree_ok = se.open_finish(shu, a.data());
    keep("w2", a);
    return pr;
}

}   // namespace pzp
