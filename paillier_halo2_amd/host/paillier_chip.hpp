// paillier_chip.hpp -- C++ mirror of the reference's chip interface for the hot path, on top of the C ABI
// (include/pz.h).  Rust is not available in this image, the reference is compiled code, so the host side
// above the ABI is C++ with the reference's names, argument meaning and error behaviour:
//
//   reference (Rust)                                          here
//   ------------------------------------------------------    ------------------------------------------
//   halo2_base::Context                                        pz::Context      (device-resident step tape)
//   halo2_base::gates::RangeChip                               pz::RangeChip    {lookup_bits}
//   biguint_halo2::big_uint::chip::BigUintChip                 pz::BigUintChip  ::construct / assign_integer /
//     (call sites paillier.rs:39-57, bench.rs:40-74)             square / refresh / mul_mod / pow_mod_fixed_exp /
//                                                                assert_equal_fresh
//   AssignedBigUint<F, Fresh|Muled>, RefreshAux                pz::AssignedBigUint, pz::RefreshAux
//   paillier.rs:6-9   EncryptionPublicKeyAssigned              pz::EncryptionPublicKeyAssigned
//   paillier.rs:11-20 PaillierChip / construct                 pz::PaillierChip / construct
//   paillier.rs:22-30 get_biguint                              PaillierChip::get_biguint
//   paillier.rs:32-60 encrypt, :62-85 add                      PaillierChip::encrypt / add
//   paillier.rs:87-97 paillier_enc_native / _add_native        pz::paillier_enc_native / paillier_add_native
//   bench.rs:11-31    input structs, :33-117 drivers           pz::PaillierEncryptionInput ... paillier_enc_test ...
//
// Where the reference pushes cells into a CPU Context one by one, this Context records a TAPE of the chip
// operations in call order (assign_integer, square, refresh, load_zero, load_constant, mul_mod steps produced by the
// K3 kernels, assert_equal_fresh) with the cells each one pushes, and expands the whole tape to the advice / lookup
// cell streams on the device (synthesize_circuit -> pz_circuit_expand_dev).  Results<> carry plonk::Error's role; unwrap() throws where Rust
// would panic; value mismatches throw like the reference's assert_eq! (paillier.rs:158-163).
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pz.h"
#include "biguint.hpp"

namespace pz {

struct Error {  // halo2_proofs::plonk::Error stand-in
    int status = 0;
    std::string msg;
};
template <class T> struct Result {
    bool ok = false;
    T val{};
    Error err;
    static Result Ok(T v) { Result r; r.ok = true; r.val = std::move(v); return r; }
    static Result Err(int st, const std::string& m) { Result r; r.err.status = st; r.err.msg = m; return r; }
    T unwrap() const {
        if (!ok) throw std::runtime_error("called unwrap() on an Err: " + err.msg);  // Rust: panic
        return val;
    }
};

struct RangeChip {
    unsigned lookup_bits;
};

// device context: owns the pz_ctx and the step tape of the circuit being built
class Context {
  public:
    explicit Context(int device = 0) {
        int dev = device;
        int rc = pz_init(1, &dev, &ctx_);
        if (rc != PZ_OK) throw std::runtime_error(std::string("pz_init: ") + pz_strerror(rc));
    }
    ~Context() { if (ctx_) pz_free(ctx_); }
    Context(const Context&) = delete;
    pz_ctx* raw() { return ctx_; }
    // ---- the operation tape: what the reference's Context would hold, operation by operation
    enum Op { ASSIGN = 0, SQUARE = 1, REFRESH = 2, CONST_CELL = 3, MUL_MOD = 4, ASSERT_EQUAL = 5 };
    struct OpRec {
        Op op;
        unsigned limbs;      // limb count the operation works on
        size_t count;        // MUL_MOD: number of steps; otherwise 1
    };
    void set_shape(unsigned limb_bits, unsigned lookup_bits) { W_ = limb_bits; lookup_bits_ = lookup_bits; }
    // record an operation and the cells it pushes (pz_op_cells: the same arithmetic pz_circuit_cells sums)
    void record(Op op, unsigned limbs, size_t count = 1) {
        size_t a = 0, l = 0;
        int rc = pz_op_cells((int)op, limbs, W_, lookup_bits_, &a, &l);
        if (rc != PZ_OK) throw std::runtime_error(std::string("pz_op_cells: ") + pz_strerror(rc));
        advice_cells_ += a * count;
        lookup_cells_ += l * count;
        if (!ops_.empty() && ops_.back().op == op && op == MUL_MOD && ops_.back().limbs == limbs) ops_.back().count += count;
        else ops_.push_back(OpRec{op, limbs, count});
    }
    size_t load_zero() { record(CONST_CELL, 1); return zero_cells_++; }            // paillier.rs:47 ctx.load_zero(): one advice cell
    size_t load_constant_one() { record(CONST_CELL, 1); return zero_cells_++; }    // assign_constant(1) of pow_mod_fixed_exp
    const std::vector<OpRec>& ops() const { return ops_; }
    size_t advice_cells() const { return advice_cells_; }
    size_t lookup_cells() const { return lookup_cells_; }
    unsigned lookup_bits() const { return lookup_bits_; }
    // tape of mul_mod steps in emission order: (a|b|q|r), `words` 64-bit words each, whatever the circuit's limb
    // width is; all steps of one circuit share the limb count, the limb width and the modulus
    void push_steps(const std::vector<uint64_t>& steps, unsigned words, unsigned limbs, unsigned limb_bits, const BigUint& modulus) {
        if (L_ && (L_ != limbs || W_ != limb_bits || modulus_ != modulus)) throw std::logic_error("one Context holds steps of one modulus");
        L_ = limbs;
        W_ = limb_bits;
        words_ = words;
        modulus_ = modulus;
        tape_.insert(tape_.end(), steps.begin(), steps.end());
        record(MUL_MOD, limbs, steps.size() / (4 * (size_t)words));
    }
    size_t n_steps() const { return words_ ? tape_.size() / (4 * words_) : 0; }
    const std::vector<uint64_t>& tape() const { return tape_; }
    unsigned limbs() const { return L_; }        // circuit limbs per big integer
    unsigned limb_bits() const { return W_; }
    unsigned words() const { return words_; }    // 64-bit words per big integer of a step record
    const BigUint& modulus() const { return modulus_; }

  private:
    pz_ctx* ctx_ = nullptr;
    size_t zero_cells_ = 0, advice_cells_ = 0, lookup_cells_ = 0;
    unsigned L_ = 0, W_ = 64, words_ = 0, lookup_bits_ = 0;
    std::vector<OpRec> ops_;
    BigUint modulus_;
    std::vector<uint64_t> tape_;
};

struct Fresh {};
struct Muled {};

struct RefreshAux {   // paillier.rs:40-44
    unsigned limb_bits, num_limbs_l, num_limbs_r;
    std::vector<uint8_t> increased_limbs_vec;   // extra limbs each product limb's maximal value spills into
    static RefreshAux new_(unsigned limb_bits, unsigned l, unsigned r) {
        RefreshAux a{limb_bits, l, r, std::vector<uint8_t>(256)};
        uint32_t n = 0;
        int rc = pz_refresh_aux(limb_bits, l, r, a.increased_limbs_vec.data(), 256, &n);
        if (rc != PZ_OK) throw std::runtime_error(std::string("RefreshAux::new: ") + pz_strerror(rc));
        a.increased_limbs_vec.resize(n);
        return a;
    }
};

template <class Kind> class AssignedBigUint {
  public:
    AssignedBigUint() {}
    AssignedBigUint(BigUint v, unsigned num_limbs, unsigned max_limb_bits)
        : value_(std::move(v)), num_limbs_(num_limbs), max_limb_bits_(max_limb_bits) {}
    const BigUint& value() const { return value_; }
    unsigned num_limbs() const { return num_limbs_; }
    unsigned max_limb_bits() const { return max_limb_bits_; }
    // the circuit's limbs (max_limb_bits wide, limbs()[0] least significant): what the reference's limbs() cells hold
    std::vector<BigUint> limbs() const {
        if (value_.bits() > (size_t)num_limbs_ * max_limb_bits_) throw std::range_error("integer does not fit the limb count");
        std::vector<BigUint> r;
        for (unsigned i = 0; i < num_limbs_; ++i) r.push_back((value_ >> ((size_t)i * max_limb_bits_)).low_bits(max_limb_bits_));
        return r;
    }
    // the same integer as the C ABI takes it: little-endian 64-bit words covering num_limbs * max_limb_bits bits
    unsigned num_words() const { return (num_limbs_ * max_limb_bits_ + 63) / 64; }
    std::vector<uint64_t> words() const {
        if (value_.bits() > (size_t)num_limbs_ * max_limb_bits_) throw std::range_error("integer does not fit the limb count");
        return value_.to_limbs(num_words());
    }
    AssignedBigUint extend_limbs(unsigned extra, size_t /*zero_cell*/) const {   // paillier.rs:49,53,79-80
        return AssignedBigUint(value_, num_limbs_ + extra, max_limb_bits_);
    }

  private:
    BigUint value_;
    unsigned num_limbs_ = 0, max_limb_bits_ = 64;
};

class BigUintChip {
  public:
    const RangeChip* range;
    unsigned limb_bits;
    static BigUintChip construct(const RangeChip* range, unsigned limb_bits) {
        if (limb_bits < 16 || limb_bits > 90) throw std::invalid_argument("limb_bits outside 16..90 (K4 carries a limb in two 64-bit words)");
        return BigUintChip{range, limb_bits};
    }
    // assign_integer(ctx, Value::known(v), bit_len): bit_len must be a multiple of limb_bits, v must fit (its limbs'
    // range checks would fail otherwise); pushes the limbs and one range check each
    Result<AssignedBigUint<Fresh>> assign_integer(Context& ctx, const BigUint& v, unsigned bit_len) const {
        if (bit_len % limb_bits) return Result<AssignedBigUint<Fresh>>::Err(PZ_ERR_INVALID, "bit_len % limb_bits != 0");
        if (v.bits() > bit_len) return Result<AssignedBigUint<Fresh>>::Err(PZ_ERR_RANGE, "value exceeds bit_len");
        ctx.set_shape(limb_bits, range->lookup_bits);
        ctx.record(Context::ASSIGN, bit_len / limb_bits);
        return Result<AssignedBigUint<Fresh>>::Ok(AssignedBigUint<Fresh>(v, bit_len / limb_bits, limb_bits));
    }
    // square = mul(a, a): unreduced limb products (2 l - 1 limbs of up to 2 W + log2(l) bits); the cells are the limb
    // convolution K4 writes on the device
    Result<AssignedBigUint<Muled>> square(Context& ctx, const AssignedBigUint<Fresh>& a) const {  // paillier.rs:39
        ctx.record(Context::SQUARE, a.num_limbs());
        unsigned lg = 0;
        while ((1u << lg) < a.num_limbs()) ++lg;
        return Result<AssignedBigUint<Muled>>::Ok(AssignedBigUint<Muled>(a.value() * a.value(), 2 * a.num_limbs() - 1,
                                                                         2 * limb_bits + lg));
    }
    // refresh: cut every product limb back to limb_bits-wide limbs, carrying into the limbs above as aux says
    Result<AssignedBigUint<Fresh>> refresh(Context& ctx, const AssignedBigUint<Muled>& a, const RefreshAux& aux) const {  // :45
        using R = Result<AssignedBigUint<Fresh>>;
        if (aux.limb_bits != limb_bits) return R::Err(PZ_ERR_INVALID, "refresh: aux.limb_bits != limb_bits");
        if (a.num_limbs() != aux.num_limbs_l + aux.num_limbs_r - 1) return R::Err(PZ_ERR_INVALID, "refresh: limb count does not match aux");
        if (aux.num_limbs_l != aux.num_limbs_r) return R::Err(PZ_ERR_UNSUPPORTED, "refresh: only squares are expanded on the device");
        const unsigned fresh = (unsigned)aux.increased_limbs_vec.size();
        if (a.value().bits() > (size_t)fresh * limb_bits) return R::Err(PZ_ERR_RANGE, "refresh: value exceeds the refreshed limbs");
        ctx.record(Context::REFRESH, aux.num_limbs_l);
        return R::Ok(AssignedBigUint<Fresh>(a.value(), fresh, limb_bits));
    }
    // mul_mod(ctx, a, b, n): witness (q, r) from the K3 kernel, step recorded for K4
    Result<AssignedBigUint<Fresh>> mul_mod(Context& ctx, const AssignedBigUint<Fresh>& a, const AssignedBigUint<Fresh>& b,
                                           const AssignedBigUint<Fresh>& n) const {
        const unsigned L = n.num_limbs();
        if (a.num_limbs() != L || b.num_limbs() != L) return Result<AssignedBigUint<Fresh>>::Err(PZ_ERR_INVALID, "limb count mismatch");
        const unsigned Wd = n.num_words();
        std::vector<uint64_t> av = a.words(), bv = b.words(), nv = n.words(), q(Wd), r(Wd);
        int rc = pz_mul_mod(ctx.raw(), Wd, av.data(), bv.data(), nv.data(), q.data(), r.data());
        if (rc != PZ_OK) return Result<AssignedBigUint<Fresh>>::Err(rc, std::string("pz_mul_mod: ") + pz_strerror(rc));
        BigUint qv = BigUint::from_limbs(q.data(), Wd);
        if (qv.bits() > (size_t)L * limb_bits)   // the circuit assigns q with L limbs: its range check would fail
            return Result<AssignedBigUint<Fresh>>::Err(PZ_ERR_RANGE, "quotient does not fit the assigned limb count");
        std::vector<uint64_t> step;
        step.insert(step.end(), av.begin(), av.end());
        step.insert(step.end(), bv.begin(), bv.end());
        step.insert(step.end(), q.begin(), q.end());
        step.insert(step.end(), r.begin(), r.end());
        ctx.push_steps(step, Wd, L, limb_bits, n.value());
        return Result<AssignedBigUint<Fresh>>::Ok(AssignedBigUint<Fresh>(BigUint::from_limbs(r.data(), Wd), L, limb_bits));
    }
    // pow_mod_fixed_exp(ctx, a, e, n): e is a native BigUint -- the exponent's bits shape the circuit
    Result<AssignedBigUint<Fresh>> pow_mod_fixed_exp(Context& ctx, const AssignedBigUint<Fresh>& a, const BigUint& e,
                                                     const AssignedBigUint<Fresh>& n) const {
        const unsigned L = n.num_limbs();
        if (a.num_limbs() != L) return Result<AssignedBigUint<Fresh>>::Err(PZ_ERR_INVALID, "limb count mismatch");
        (void)ctx.load_constant_one();   // acc = assign_constant(1) ...
        (void)ctx.load_zero();           // ... extended with the zero cell
        const unsigned Wd = n.num_words();
        std::vector<uint64_t> av = a.words(), nv = n.words(), res(Wd);
        const unsigned el = e.l.empty() ? 1 : (unsigned)e.l.size();
        std::vector<uint64_t> ev = e.to_limbs(el);
        size_t cap = e.bits();
        for (uint64_t w : ev) cap += (size_t)__builtin_popcountll(w);
        std::vector<uint64_t> steps(std::max<size_t>(cap, 1) * 4 * Wd);
        size_t ns = cap;
        int rc = pz_paillier_trace(ctx.raw(), Wd, nv.data(), av.data(), ev.data(), el, steps.data(), &ns, res.data());
        if (rc != PZ_OK) return Result<AssignedBigUint<Fresh>>::Err(rc, std::string("pz_paillier_trace: ") + pz_strerror(rc));
        steps.resize(ns * 4 * Wd);
        if (ns) ctx.push_steps(steps, Wd, L, limb_bits, n.value());
        return Result<AssignedBigUint<Fresh>>::Ok(AssignedBigUint<Fresh>(BigUint::from_limbs(res.data(), Wd), L, limb_bits));
    }
    // is_equal_fresh limb by limb (load_zero, load_constant(1), is_equal + and per limb), result constrained to 1
    Result<bool> assert_equal_fresh(Context& ctx, const AssignedBigUint<Fresh>& a, const AssignedBigUint<Fresh>& b) const {
        const unsigned max_n = std::max(a.num_limbs(), b.num_limbs());
        ctx.record(Context::ASSERT_EQUAL, max_n);
        std::vector<BigUint> al = a.extend_limbs(max_n - a.num_limbs(), 0).limbs(), bl = b.extend_limbs(max_n - b.num_limbs(), 0).limbs();
        bool eq = true;
        for (unsigned i = 0; i < max_n; ++i) eq = eq && al[i] == bl[i];
        if (!eq) return Result<bool>::Err(PZ_ERR_INVALID, "assert_equal_fresh: constraint not satisfied");
        return Result<bool>::Ok(true);
    }
};

struct EncryptionPublicKeyAssigned {  // paillier.rs:6-9
    AssignedBigUint<Fresh> n, g;
};

class PaillierChip {  // paillier.rs:11-15
  public:
    const BigUintChip* biguint;
    unsigned enc_bits;  // stored, never read by encrypt/add -- as in the reference
    static PaillierChip construct(const BigUintChip* biguint, unsigned enc_bits) { return PaillierChip{biguint, enc_bits}; }

    // paillier.rs:22-30: fold limbs MSB -> LSB with shift max_limb_bits
    BigUint get_biguint(const AssignedBigUint<Fresh>& assigned) const {
        BigUint acc;
        std::vector<BigUint> limbs = assigned.limbs();
        for (size_t i = limbs.size(); i-- > 0;) acc = (acc << assigned.max_limb_bits()) + limbs[i];
        return acc;
    }

    // paillier.rs:32-60
    Result<AssignedBigUint<Fresh>> encrypt(Context& ctx, const EncryptionPublicKeyAssigned& pk_enc, const AssignedBigUint<Fresh>& m,
                                           const AssignedBigUint<Fresh>& r) const {
        using R = Result<AssignedBigUint<Fresh>>;
        auto n2m = biguint->square(ctx, pk_enc.n);
        if (!n2m.ok) return R::Err(n2m.err.status, n2m.err.msg);
        RefreshAux aux = RefreshAux::new_(biguint->limb_bits, pk_enc.n.num_limbs(), pk_enc.n.num_limbs());
        auto n2r = biguint->refresh(ctx, n2m.val, aux);
        if (!n2r.ok) return R::Err(n2r.err.status, n2r.err.msg);
        const AssignedBigUint<Fresh>& n2 = n2r.val;
        size_t zero_value = ctx.load_zero();
        auto g_extended = pk_enc.g.extend_limbs(n2.num_limbs() - pk_enc.g.num_limbs(), zero_value);
        BigUint m_biguint = get_biguint(m);
        auto gm = biguint->pow_mod_fixed_exp(ctx, g_extended, m_biguint, n2);
        if (!gm.ok) return gm;
        auto r_extended = r.extend_limbs(n2.num_limbs() - r.num_limbs(), zero_value);
        BigUint n_biguint = get_biguint(pk_enc.n);
        auto rn = biguint->pow_mod_fixed_exp(ctx, r_extended, n_biguint, n2);
        if (!rn.ok) return rn;
        return biguint->mul_mod(ctx, gm.val, rn.val, n2);
    }

    // paillier.rs:62-85 (pk_enc.g unused, as in the reference)
    Result<AssignedBigUint<Fresh>> add(Context& ctx, const EncryptionPublicKeyAssigned& pk_enc, const AssignedBigUint<Fresh>& c1,
                                       const AssignedBigUint<Fresh>& c2) const {
        using R = Result<AssignedBigUint<Fresh>>;
        auto n2m = biguint->square(ctx, pk_enc.n);
        if (!n2m.ok) return R::Err(n2m.err.status, n2m.err.msg);
        RefreshAux aux = RefreshAux::new_(biguint->limb_bits, pk_enc.n.num_limbs(), pk_enc.n.num_limbs());
        auto n2r = biguint->refresh(ctx, n2m.val, aux);
        if (!n2r.ok) return R::Err(n2r.err.status, n2r.err.msg);
        const AssignedBigUint<Fresh>& n2 = n2r.val;
        size_t zero_value = ctx.load_zero();
        auto c1e = c1.extend_limbs(n2.num_limbs() - c1.num_limbs(), zero_value);
        auto c2e = c2.extend_limbs(n2.num_limbs() - c2.num_limbs(), zero_value);
        return biguint->mul_mod(ctx, c1e, c2e, n2);
    }
};

// paillier.rs:87-92: (g^m * r^n) mod n^2 -- one pz_paillier_encrypt call without a trace
inline BigUint paillier_enc_native(Context& ctx, const BigUint& n, const BigUint& g, const BigUint& m, const BigUint& r) {
    size_t Ln = std::max<size_t>(1, std::max(std::max(n.l.size(), g.l.size()), std::max(m.l.size(), r.l.size())));
    std::vector<uint64_t> nv = n.to_limbs(Ln), gv = g.to_limbs(Ln), mv = m.to_limbs(Ln), rv = r.to_limbs(Ln), c(2 * Ln);
    int rc = pz_paillier_encrypt(ctx.raw(), (uint32_t)Ln, 1, nv.data(), gv.data(), mv.data(), rv.data(), nullptr, 0, nullptr,
                                 nullptr, c.data());
    if (rc != PZ_OK) throw std::runtime_error(std::string("paillier_enc_native: ") + pz_strerror(rc));  // Rust: % 0 panics
    return BigUint::from_limbs(c.data(), 2 * Ln);
}
// paillier.rs:94-97
inline BigUint paillier_add_native(Context& ctx, const BigUint& n, const BigUint& c1, const BigUint& c2) {
    BigUint n2 = n * n;
    size_t L = std::max<size_t>(1, std::max(n2.l.size(), std::max(c1.l.size(), c2.l.size())));
    std::vector<uint64_t> nv = n2.to_limbs(L), a = c1.to_limbs(L), b = c2.to_limbs(L), q(L), r(L);
    int rc = pz_mul_mod(ctx.raw(), (uint32_t)L, a.data(), b.data(), nv.data(), q.data(), r.data());
    if (rc != PZ_OK) throw std::runtime_error(std::string("paillier_add_native: ") + pz_strerror(rc));
    return BigUint::from_limbs(r.data(), L);
}

// bench.rs:11-31
struct PaillierEncryptionInput {
    unsigned enc_bits, limb_bits;
    BigUint n, g, m, r, res;
};
struct PaillierAddCipherInput {
    unsigned limb_bits, enc_bits;
    BigUint n, g, c1, c2, res;
};

// bench.rs:33-75: assign n, g, m, r -> encrypt -> assign res at 2*enc_bits -> value assert -> assert_equal_fresh
inline void paillier_enc_test(Context& ctx, const RangeChip& range, const PaillierEncryptionInput& input) {
    BigUintChip biguint_chip = BigUintChip::construct(&range, input.limb_bits);
    PaillierChip paillier_chip = PaillierChip::construct(&biguint_chip, input.enc_bits);
    auto n_assigned = biguint_chip.assign_integer(ctx, input.n, input.enc_bits).unwrap();
    auto g_assigned = biguint_chip.assign_integer(ctx, input.g, input.enc_bits).unwrap();
    EncryptionPublicKeyAssigned pk_enc{n_assigned, g_assigned};
    auto m_assigned = biguint_chip.assign_integer(ctx, input.m, input.enc_bits).unwrap();
    auto r_assigned = biguint_chip.assign_integer(ctx, input.r, input.enc_bits).unwrap();
    auto c_assigned = paillier_chip.encrypt(ctx, pk_enc, m_assigned, r_assigned).unwrap();
    auto res_assigned = biguint_chip.assign_integer(ctx, input.res, input.enc_bits * 2).unwrap();
    if (c_assigned.value() != res_assigned.value()) throw std::runtime_error("assertion failed: `(left == right)` (paillier_enc_test)");
    biguint_chip.assert_equal_fresh(ctx, c_assigned, res_assigned).unwrap();
}
// bench.rs:77-117
inline void paillier_enc_add_test(Context& ctx, const RangeChip& range, const PaillierAddCipherInput& input) {
    BigUintChip biguint_chip = BigUintChip::construct(&range, input.limb_bits);
    PaillierChip paillier_chip = PaillierChip::construct(&biguint_chip, input.enc_bits);
    auto n_assigned = biguint_chip.assign_integer(ctx, input.n, input.enc_bits).unwrap();
    auto g_assigned = biguint_chip.assign_integer(ctx, input.g, input.enc_bits).unwrap();
    EncryptionPublicKeyAssigned pk_enc{n_assigned, g_assigned};
    auto c1_assigned = biguint_chip.assign_integer(ctx, input.c1, input.enc_bits).unwrap();
    auto c2_assigned = biguint_chip.assign_integer(ctx, input.c2, input.enc_bits).unwrap();
    auto res = paillier_chip.add(ctx, pk_enc, c1_assigned, c2_assigned).unwrap();
    auto res_assigned = biguint_chip.assign_integer(ctx, input.res, input.enc_bits * 2).unwrap();
    if (res.value() != res_assigned.value()) throw std::runtime_error("assertion failed: `(left == right)` (paillier_enc_add_test)");
    biguint_chip.assert_equal_fresh(ctx, res, res_assigned).unwrap();
}

// K4 over the mul_mod steps alone (the bulk of the stream; the caller places the buffers)
inline int synthesize_witness(Context& ctx, const RangeChip& range, uint64_t* d_steps, uint64_t* d_modulus, uint64_t* d_advice,
                              uint64_t* d_lookup) {
    return pz_witness_expand_dev(ctx.raw(), ctx.limbs(), ctx.limb_bits(), range.lookup_bits, d_steps, ctx.n_steps(), d_modulus, d_advice,
                                 d_lookup);
}

// The whole tape of one of the two drivers expanded on the device (pz_circuit_expand_dev).  kind 0 = paillier_enc_test
// (x, y = m, r), 1 = paillier_enc_add_test (x, y = c1, c2).  The tape must have the driver's shape; its cell total equals
// pz_circuit_cells (checked).  d_steps / d_modulus / d_advice / d_lookup: device buffers of the caller.
inline int synthesize_circuit(Context& ctx, int kind, unsigned enc_bits, const BigUint& n, const BigUint& g, const BigUint& x,
                              const BigUint& y, const BigUint& res, size_t n_steps_g, size_t n_steps_r, uint64_t* d_steps,
                              uint64_t* d_modulus, uint64_t* d_advice, uint64_t* d_lookup) {
    const unsigned W = ctx.limb_bits(), Ln = enc_bits / W, wn = (Ln * W + 63) / 64, wr = (2 * Ln * W + 63) / 64;
    size_t a = 0, l = 0;
    int rc = pz_circuit_cells(kind, Ln, W, ctx.lookup_bits(), n_steps_g, n_steps_r, &a, &l);
    if (rc != PZ_OK) return rc;
    if (a != ctx.advice_cells() || l != ctx.lookup_cells()) return PZ_ERR_INVALID;   // the tape is not this driver's
    std::vector<uint64_t> in;
    for (const BigUint* v : {&n, &g, &x, &y}) {
        std::vector<uint64_t> w = v->to_limbs(wn);
        in.insert(in.end(), w.begin(), w.end());
    }
    std::vector<uint64_t> w = res.to_limbs(wr);
    in.insert(in.end(), w.begin(), w.end());
    return pz_circuit_expand_dev(ctx.raw(), kind, Ln, W, ctx.lookup_bits(), in.data(), d_steps, n_steps_g, n_steps_r, d_modulus,
                                 d_advice, d_lookup, 0, 0);
}

}  // namespace pz
