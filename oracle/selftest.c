/* selftest.c -- drives the C restatement (pz_oracle.c) through its exported entry points on small seeded inputs so the
 * whole file can be run under AddressSanitizer / UBSan on the CPU (`make -C oracle asan`): GPU sanitizers are not
 * available on the pool, so memory errors in the CHECKER are hunted here.  TEST INFRASTRUCTURE ONLY, like the rest of
 * oracle/.  Self-consistency checked: walk bases lie on the curve; MSM(k, G..) agrees with per-point mul + add;
 * NTT followed by the inverse NTT and the 1/n scale is the identity; a*b == q*mod + r for every traced step. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
int ora_fr_mul(const u64 a[4], const u64 b[4], u64 out[4]);
int ora_mont_convert(u64 *a, size_t n, int which, int to_mont);
int ora_g1_normalize(const u64 jac[12], u64 aff[8]);
int ora_g1_mul(const u64 base_aff[8], const u64 k[4], u64 out_jac[12]);
int ora_g1_add(const u64 a[12], const u64 b[12], u64 out[12]);
int ora_g1_on_curve(const u64 aff[8]);
int ora_walk_bases(u64 *out_aff, size_t n, const u64 s[4], const u64 t[4]);
int ora_msm_g1(const u64 *scalars, const u64 *bases, size_t n, int threads, u64 out_jac[12]);
int ora_ntt_fr(u64 *a_, const u64 omega_[4], uint32_t log_n, int threads);
int ora_fr_scale(u64 *a_, size_t n, const u64 scale[4]);
int ora_mul_mod_step(uint32_t L, const u64 *a, const u64 *b, const u64 *mod, u64 *q, u64 *r);
int ora_pow_mod_trace(uint32_t L, const u64 *mod, const u64 *base, const u64 *exp, uint32_t exp_limbs, u64 *steps,
                      size_t *n_steps, u64 *result);
int ora_paillier_enc(uint32_t Ln, const u64 *n, const u64 *g, const u64 *m, const u64 *r, u64 *c_out);

static u64 rs = 0x9E3779B97F4A7C15ull;
static u64 rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static void rnd_fr(u64 x[4]) { for (int i = 0; i < 4; ++i) x[i] = rnd(); x[3] &= 0x0FFFFFFFFFFFFFFFull; }
static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { printf("FAIL %s\n", msg); ++fails; } else printf("ok   %s\n", msg); } while (0)

/* schoolbook check a*b == q*mod + r on L limbs */
static int step_ok(uint32_t L, const u64 *a, const u64 *b, const u64 *q, const u64 *r, const u64 *mod) {
    u64 *l = calloc(2 * L + 1, 8), *rr = calloc(2 * L + 1, 8);
    for (uint32_t i = 0; i < L; ++i) {
        unsigned __int128 c = 0, d = 0;
        for (uint32_t j = 0; j < L; ++j) {
            c += (unsigned __int128)a[i] * b[j] + l[i + j]; l[i + j] = (u64)c; c >>= 64;
            d += (unsigned __int128)q[i] * mod[j] + rr[i + j]; rr[i + j] = (u64)d; d >>= 64;
        }
        l[i + L] += (u64)c; rr[i + L] += (u64)d;
    }
    unsigned __int128 c = 0;
    for (uint32_t i = 0; i < 2 * L; ++i) { c += (unsigned __int128)rr[i] + (i < L ? r[i] : 0); rr[i] = (u64)c; c >>= 64; }
    int ok = memcmp(l, rr, 2 * L * 8) == 0;
    free(l); free(rr);
    return ok;
}

int main(void) {
    /* G1: walk bases, MSM vs mul+add */
    enum { N = 37 };
    u64 s[4], t[4], *bases = malloc(N * 64), *sc = malloc(N * 32);
    rnd_fr(s); rnd_fr(t);
    CHECK(ora_walk_bases(bases, N, s, t) == 0, "walk_bases");
    int on = 1;
    for (int i = 0; i < N; ++i) on &= ora_g1_on_curve(bases + 8 * i) == 1;
    CHECK(on, "walk bases on curve");
    for (int i = 0; i < N; ++i) rnd_fr(sc + 4 * i);
    memset(sc, 0, 32); memset(sc + 4, 0, 32); sc[4] = 1;  /* scalars 0 and 1 among them */
    u64 *sc_canon = malloc(N * 32);
    memcpy(sc_canon, sc, N * 32);
    ora_mont_convert(sc, N, 1, 1);   /* best_multiexp takes Fr in Montgomery form */
    u64 msm[12], acc[12], tmp[12], a1[8], a2[8];
    CHECK(ora_msm_g1(sc, bases, N, 2, msm) == 0, "msm");
    memset(acc, 0, sizeof acc);
    int first = 1;
    for (int i = 0; i < N; ++i) {
        ora_g1_mul(bases + 8 * i, sc_canon + 4 * i, tmp);
        if (first) { memcpy(acc, tmp, sizeof acc); first = 0; } else ora_g1_add(acc, tmp, acc);
    }
    ora_g1_normalize(msm, a1); ora_g1_normalize(acc, a2);
    CHECK(memcmp(a1, a2, 64) == 0, "msm == sum of scalar multiplications");
    /* NTT round trip at 2^9 */
    enum { LOGN = 9, NN = 1 << LOGN };
    u64 *v = malloc(NN * 32), *w = malloc(NN * 32);
    for (int i = 0; i < NN; ++i) rnd_fr(v + 4 * i);
    ora_mont_convert(v, NN, 1, 1);
    memcpy(w, v, NN * 32);
    /* omega = 7^((r-1)/2^9): computed by repeated squaring of ROOT_OF_UNITY (2^28-th root), Montgomery form via convert */
    u64 root[4] = {0xd34f1ed960c37c9cull, 0x3215cf6dd39329c8ull, 0x98865ea93dd31f74ull, 0x03ddb9f5166d18b7ull};
    ora_mont_convert(root, 1, 1, 1);
    for (int i = 0; i < 28 - LOGN; ++i) ora_fr_mul(root, root, root);
    u64 inv[4]; memcpy(inv, root, 32);
    for (int i = 0; i < NN - 2; ++i) ora_fr_mul(inv, root, inv);   /* omega^(n-1) = omega^-1 */
    CHECK(ora_ntt_fr(w, root, LOGN, 2) == 0 && ora_ntt_fr(w, inv, LOGN, 1) == 0, "ntt forward + inverse");
    /* w == n * v: scale v by n (Montgomery n via repeated doubling of one) and compare */
    u64 nm[4] = {NN, 0, 0, 0};
    ora_mont_convert(nm, 1, 1, 1);
    ora_fr_scale(v, NN, nm);
    CHECK(memcmp(v, w, NN * 32) == 0, "intt(ntt(v)) == n * v");
    /* big-integer trace at 4 limbs + a short Paillier encryption at 2 limbs */
    enum { L = 4 };
    u64 mod[L], base[L], ex[2], res[L], steps[4 * L * 260];
    for (int i = 0; i < L; ++i) { mod[i] = rnd(); base[i] = rnd(); }
    mod[L - 1] |= 1ull << 63; base[L - 1] &= mod[L - 1] - 1;
    ex[0] = rnd(); ex[1] = rnd() & 0xffff;
    size_t ns = 260;
    CHECK(ora_pow_mod_trace(L, mod, base, ex, 2, steps, &ns, res) == 0 && ns > 64 && ns <= 260, "pow_mod_trace");
    int all = 1;
    for (size_t k = 0; k < ns; ++k) all &= step_ok(L, steps + 4 * L * k, steps + 4 * L * k + L, steps + 4 * L * k + 2 * L, steps + 4 * L * k + 3 * L, mod);
    CHECK(all, "every traced step satisfies a*b == q*mod + r");
    u64 n2[2] = {rnd() | 1, rnd() | (1ull << 63)}, g2[2] = {rnd(), rnd() >> 1}, m2[2] = {rnd(), 0}, r2[2] = {rnd(), rnd() >> 1}, c[4];
    CHECK(ora_paillier_enc(2, n2, g2, m2, r2, c) == 0, "paillier_enc");
    u64 zero[L] = {0}, q[L], r[L];
    CHECK(ora_mul_mod_step(L, base, base, zero, q, r) != 0, "modulus 0 is reported, not divided by");
    free(bases); free(sc); free(sc_canon); free(v); free(w);
    printf("%s (%d failures)\n", fails ? "FAILED" : "ALL OK", fails);
    return fails ? 1 : 0;
}
