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

#include <hip/hip_runtime_api.h>

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

static size_t popcount(const BigUint& v) {
    size_t p = 0;
    for (auto w : v.l) p += (size_t)__builtin_popcountll(w);
    return p;
}

// the tape the driver recorded == the whole circuit's stream (layout.py::circuit_cells / pz_circuit_cells), and its
// expansion on the device ends in assert_equal_fresh's result bit: Montgomery 1 for the honest witness
static void check_tape(Context& ctx, int kind, unsigned enc_bits, const BigUint& n, const BigUint& g, const BigUint& x, const BigUint& y,
                       const BigUint& res, size_t ng, size_t nr, const char* what) {
    static const uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    const unsigned W = ctx.limb_bits(), Ln = enc_bits / W;
    size_t a = 0, l = 0;
    int rc = pz_circuit_cells(kind, Ln, W, ctx.lookup_bits(), ng, nr, &a, &l);
    char name[160];
    std::snprintf(name, sizeof name, "%s: tape of %zu operations = %zu advice + %zu lookup cells == pz_circuit_cells", what, ctx.ops().size(),
                  ctx.advice_cells(), ctx.lookup_cells());
    CHECK(rc == PZ_OK && a == ctx.advice_cells() && l == ctx.lookup_cells(), name);
    // operation order of the driver: 4 assigns, square, refresh, load_zero, [const, zero, steps] x 2, final step, assign, assert
    std::vector<int> want = {0, 0, 0, 0, 1, 2, 3};
    if (kind == 0) want.insert(want.end(), {3, 3, 4, 3, 3, 4});   // the final mul_mod merges into the r^n run of steps
    else want.push_back(4);
    want.insert(want.end(), {0, 5});
    std::vector<int> got;
    for (auto& o : ctx.ops()) got.push_back((int)o.op);
    CHECK(got == want, "operation order of the tape (bench.rs:33-75 / 77-117, paillier.rs:32-85)");
    uint64_t *d_steps = nullptr, *d_mod = nullptr, *d_adv = nullptr, *d_lk = nullptr;
    std::vector<uint64_t> mod = ctx.modulus().to_limbs(ctx.words());
    bool ok = hipMalloc((void**)&d_steps, ctx.tape().size() * 8) == hipSuccess && hipMalloc((void**)&d_mod, mod.size() * 8) == hipSuccess &&
              hipMalloc((void**)&d_adv, a * 32) == hipSuccess && hipMalloc((void**)&d_lk, l * 32 + 32) == hipSuccess;
    ok = ok && hipMemcpy(d_steps, ctx.tape().data(), ctx.tape().size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_mod, mod.data(), mod.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    uint64_t last[4] = {0, 0, 0, 0}, last_bad[4] = {1, 1, 1, 1};
    if (ok) {
        ok = synthesize_circuit(ctx, kind, enc_bits, n, g, x, y, res, ng, nr, d_steps, d_mod, d_adv, d_lk) == PZ_OK && pz_sync(ctx.raw()) == PZ_OK &&
             hipMemcpy(last, d_adv + 4 * (a - 1), 32, hipMemcpyDeviceToHost) == hipSuccess;
        BigUint wrong = res + BigUint(1);
        ok = ok && synthesize_circuit(ctx, kind, enc_bits, n, g, x, y, wrong, ng, nr, d_steps, d_mod, d_adv, d_lk) == PZ_OK && pz_sync(ctx.raw()) == PZ_OK &&
             hipMemcpy(last_bad, d_adv + 4 * (a - 1), 32, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d_steps); (void)hipFree(d_mod); (void)hipFree(d_adv); (void)hipFree(d_lk);
    CHECK(ok && std::memcmp(last, ONE, 32) == 0, "device expansion of the whole tape ends in assert_equal_fresh == 1");
    CHECK(ok && (last_bad[0] | last_bad[1] | last_bad[2] | last_bad[3]) == 0, "a wrong `res` expands to assert_equal_fresh == 0");
}

static void test_paillier_encryption(unsigned enc_bits, unsigned lookup_bits) {
    Context ctx(0);
    RangeChip range{lookup_bits};
    BigUint n = gen_biguint(enc_bits), g = gen_biguint(enc_bits), m = gen_biguint(enc_bits), r = gen_biguint(enc_bits);
    if (n.is_zero()) n = BigUint(3);
    if (enc_bits > 512) m = m.low_bits(48);   // keeps the 2048-bit tape's cell stream (12.7 GB at full width) small
    BigUint res = oracle_enc(n, g, m, r, enc_bits / 64);
    paillier_enc_test(ctx, range, PaillierEncryptionInput{enc_bits, 64, n, g, m, r, res});  // throws on any mismatch
    char name[96];
    std::snprintf(name, sizeof name, "paillier_enc_test enc_bits=%u (%zu mul_mod steps on the tape)", enc_bits, ctx.n_steps());
    const size_t ng = m.bits() + popcount(m), nr = n.bits() + popcount(n);
    CHECK(ctx.n_steps() == ng + nr + 1, name);
    CHECK(paillier_enc_native(ctx, n, g, m, r) == res, "paillier_enc_native == oracle");
    if (enc_bits <= 512 || nr < 4200) check_tape(ctx, 0, enc_bits, n, g, m, r, res, ng, nr, "paillier_enc_test");
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
    check_tape(ctx, 1, enc_bits, n, g, c1, c2, res, 0, 0, "paillier_enc_add_test");
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
