// C++ mirror of the reference's own tests, reading like them:
//   test_paillier_encryption   /root/reference/src/paillier.rs:113-182  (ENC_BIT_LEN 128, LIMB_BIT_LEN 64)
//   test_encryption_addition   /root/reference/src/paillier.rs:184-259  (ENC_BIT_LEN 264, LIMB_BIT_LEN 88)
//   bench_paillier_enc         /root/reference/src/bench.rs:137-179     (k = 14, lookup_bits = 13)
//   bench_paillier_enc_add     /root/reference/src/bench.rs:181-222
// plus the 2048-bit key of BASELINE config c2.  Inputs come from a SEEDED generator (the reference uses
// thread_rng, so its failures are not reproducible).  Expected values come from the C oracle
// (oracle/pz_oracle.c -- test infrastructure), the values under test from the HIP kernels via the chip API.
#include <cstdio>
#include <cstring>
#include <random>

#include "../../paillier_halo2_amd/host/paillier_chip.hpp"

extern "C" int ora_paillier_enc(uint32_t Ln, const uint64_t* n, const uint64_t* g, const uint64_t* m, const uint64_t* r, uint64_t* c_out);
extern "C" int ora_mul_mod_step(uint32_t L, const uint64_t* a, const uint64_t* b, const uint64_t* mod, uint64_t* q, uint64_t* r);

using namespace pz;

static std::mt19937_64 rng(0x5042);
static BigUint gen_biguint(unsigned bits) {  // like num_bigint::RandBigInt::gen_biguint: uniform below 2^bits
    std::vector<uint64_t> v((bits + 63) / 64);
    for (auto& w : v) w = rng();
    if (bits % 64) v.back() &= (1ull << (bits % 64)) - 1;
    return BigUint::from_limbs(v.data(), v.size());
}
static BigUint oracle_enc(const BigUint& n, const BigUint& g, const BigUint& m, const BigUint& r, unsigned Ln) {
    std::vector<uint64_t> c(2 * Ln);
    auto nv = n.to_limbs(Ln), gv = g.to_limbs(Ln), mv = m.to_limbs(Ln), rv = r.to_limbs(Ln);
    if (ora_paillier_enc(Ln, nv.data(), gv.data(), mv.data(), rv.data(), c.data()) != 0) throw std::runtime_error("oracle");
    return BigUint::from_limbs(c.data(), c.size());
}
static BigUint oracle_add(const BigUint& n, const BigUint& c1, const BigUint& c2, unsigned L) {
    BigUint n2 = n * n;
    auto nv = n2.to_limbs(L), a = c1.to_limbs(L), b = c2.to_limbs(L);
    std::vector<uint64_t> q(L), r(L);
    if (ora_mul_mod_step(L, a.data(), b.data(), nv.data(), q.data(), r.data()) != 0) throw std::runtime_error("oracle");
    return BigUint::from_limbs(r.data(), L);
}

static int failures = 0;
#define CHECK(cond, what)                                          \
    do {                                                           \
        if (!(cond)) { std::printf("FAIL %s\n", what); ++failures; } \
        else std::printf("ok   %s\n", what);                       \
    } while (0)

static void test_paillier_encryption(unsigned enc_bits, unsigned lookup_bits) {
    Context ctx(0);
    RangeChip range{lookup_bits};
    BigUint n = gen_biguint(enc_bits), g = gen_biguint(enc_bits), m = gen_biguint(enc_bits), r = gen_biguint(enc_bits);
    if (n.is_zero()) n = BigUint(3);
    BigUint res = oracle_enc(n, g, m, r, enc_bits / 64);
    paillier_enc_test(ctx, range, PaillierEncryptionInput{enc_bits, 64, n, g, m, r, res});  // throws on any mismatch
    char name[96];
    std::snprintf(name, sizeof name, "paillier_enc_test enc_bits=%u (%zu mul_mod steps on the tape)", enc_bits, ctx.n_steps());
    CHECK(ctx.n_steps() == m.bits() + n.bits() + (size_t)[&] { size_t p = 0; for (auto w : m.l) p += __builtin_popcountll(w); for (auto w : n.l) p += __builtin_popcountll(w); return p; }() + 1, name);
    CHECK(paillier_enc_native(ctx, n, g, m, r) == res, "paillier_enc_native == oracle");
}

static void test_encryption_addition(unsigned enc_bits, unsigned lookup_bits, unsigned limb_bits = 64) {
    Context ctx(0);
    RangeChip range{lookup_bits};
    BigUint n = gen_biguint(enc_bits), g = gen_biguint(enc_bits), c1 = gen_biguint(enc_bits), c2 = gen_biguint(enc_bits);
    if (n.is_zero()) n = BigUint(3);
    BigUint res = oracle_add(n, c1, c2, (2 * enc_bits + 63) / 64);
    paillier_enc_add_test(ctx, range, PaillierAddCipherInput{limb_bits, enc_bits, n, g, c1, c2, res});
    char name[96];
    std::snprintf(name, sizeof name, "paillier_enc_add_test enc_bits=%u limb_bits=%u: one mul_mod step, %u limbs in %u words", enc_bits,
                  limb_bits, ctx.limbs(), ctx.words());
    CHECK(ctx.n_steps() == 1 && ctx.limbs() == 2 * enc_bits / limb_bits && ctx.limb_bits() == limb_bits, name);
    CHECK(paillier_add_native(ctx, n, c1, c2) == res, "paillier_add_native == oracle");
}

static void test_error_behaviour() {
    Context ctx(0);
    RangeChip range{15};
    BigUintChip chip = BigUintChip::construct(&range, 64);
    CHECK(!chip.assign_integer(ctx, gen_biguint(130) + (BigUint(1) << 129), 128).ok, "assign_integer rejects a value wider than bit_len");
    CHECK(!chip.assign_integer(ctx, BigUint(5), 100).ok, "assign_integer rejects bit_len not a multiple of limb_bits");
    {
        BigUintChip c88 = BigUintChip::construct(&range, 88);
        BigUint v = gen_biguint(264);
        auto a88 = c88.assign_integer(ctx, v, 264).unwrap();
        BigUint acc;
        auto lm = a88.limbs();
        for (size_t i = lm.size(); i-- > 0;) acc = (acc << 88) + lm[i];
        CHECK(lm.size() == 3 && acc == v && a88.num_words() == 5, "88-bit limbs: 3 limbs of a 264-bit integer fold back (get_biguint, paillier.rs:22-30)");
    }
    auto a = chip.assign_integer(ctx, BigUint(7), 128).unwrap(), z = chip.assign_integer(ctx, BigUint(), 128).unwrap();
    auto bad = chip.mul_mod(ctx, a, a, z);
    CHECK(!bad.ok && bad.err.status == PZ_ERR_ZERO_MODULUS, "mul_mod by modulus 0 -> PZ_ERR_ZERO_MODULUS (reference: BigUint % 0 panics, paillier.rs:91)");
    bool threw = false;
    try { bad.unwrap(); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw, "unwrap() on Err throws (Rust: panic)");
    threw = false;
    try {
        PaillierEncryptionInput in{128, 64, BigUint(15), BigUint(2), BigUint(3), BigUint(4), BigUint(1)};  // wrong res
        paillier_enc_test(ctx, range, in);
    } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw, "value mismatch throws like the reference's assert_eq! (paillier.rs:158-163)");
}

int main() {
    try {
        test_paillier_encryption(128, 15);  // paillier.rs:113-182
        test_encryption_addition(264, 15, 88);  // paillier.rs:184-259: 264-bit key, 88-bit limbs
        test_encryption_addition(256, 15);
        test_paillier_encryption(128, 13);  // bench.rs:137-179
        test_encryption_addition(128, 13);  // bench.rs:181-222
        test_paillier_encryption(2048, 16); // BASELINE config c2 key size
        test_encryption_addition(2048, 14); // config c3
        test_error_behaviour();
    } catch (const std::exception& e) {
        std::printf("FAIL exception: %s\n", e.what());
        return 2;
    }
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "ALL OK", failures);
    return failures ? 1 : 0;
}
