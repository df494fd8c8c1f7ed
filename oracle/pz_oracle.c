/* pz_oracle.c -- CPU restatement ("port") of the reference hot path, plain C11 + OpenMP.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product (paillier_halo2_amd/, include/pz.h) never calls it.
 *
 * PARITY STATUS: "parity unpinned" at the byte level -- the reference ships no golden vectors and
 * cannot be built here.  Every routine below computes a mathematically unique
 * value and is cross-checked against oracle/pyref.py (Python ints) in tests/test_oracle.py.
 *
 * What each routine restates (the algorithms live in un-vendored dependencies of the reference;
 * named here with the pin the reference gives -- none: Cargo.toml:9-11 has no rev, Cargo.lock is
 * git-ignored):
 *   ora_msm_g1        halo2curves `best_multiexp` / `multiexp_serial` (bucket method, window
 *                     c = 1 | 3 | ceil(ln n), 256/c+1 segments MSB first, running-sum bucket fold,
 *                     per-thread chunking and a final sum of chunk results)
 *                     -- reached from /root/reference/src/bench.rs:161-171 via create_proof.
 *   ora_ntt_fr        halo2curves `best_fft` (bit-reversal, twiddle table, log_n butterfly layers)
 *                     -- reached from the same call site via EvaluationDomain.
 *   ora_mul_mod_step  BigUintChip::mul_mod witness part: full=a*b; (q,r)=div_rem(full, n)
 *                     (num-bigint 0.4.4 in the reference, Cargo.toml:12) -- call sites
 *                     /root/reference/src/paillier.rs:57,81.
 *   ora_pow_mod_trace BigUintChip::pow_mod_fixed_exp schedule (LSB->MSB, square every bit, multiply on
 *                     set bits, acc = 1) -- call sites paillier.rs:51,55.
 *   ora_paillier_enc  paillier_enc_native, paillier.rs:87-92.
 *
 * Data layout: field elements are 4 x u64 little-endian limbs in Montgomery form (R = 2^256), the
 * in-memory layout of halo2curves Fr/Fq; G1 affine = {x,y} (identity all-zero); Jacobian = {x,y,z}
 * (identity z = 0).  Big integers are little-endian u64 limb arrays.
 */
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

typedef struct { u64 l[4]; } fe;

typedef struct {
    u64 p[4];   /* modulus */
    u64 inv;    /* -p^-1 mod 2^64 */
    u64 r1[4];  /* R mod p */
    u64 r2[4];  /* R^2 mod p */
} field_t;

static const field_t FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0x87d20782e4866389ULL,
    {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
    {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};

static const field_t FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0xc2e1f593efffffffULL,
    {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
    {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};

int ora_num_threads(void);

/* ---------------------------------------------------------------- field arithmetic */
static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int geq4(const u64 *a, const u64 *b) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline void sub4(u64 *r, const u64 *a, const u64 *b) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a[i] - b[i] - (u64)br;
        r[i] = (u64)t;
        br = (t >> 64) & 1;
    }
}
static inline void fe_add(fe *r, const fe *a, const fe *b, const field_t *F) {
    u128 c = 0;
    u64 t[4];
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        t[i] = (u64)c;
        c >>= 64;
    }
    if (c || geq4(t, F->p)) sub4(t, t, F->p);
    memcpy(r->l, t, 32);
}
static inline void fe_sub(fe *r, const fe *a, const fe *b, const field_t *F) {
    u64 t[4];
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - (u64)br;
        t[i] = (u64)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)t[i] + F->p[i];
            t[i] = (u64)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, 32);
}
static inline void fe_neg(fe *r, const fe *a, const field_t *F) {
    if (fe_is_zero(a)) { *r = *a; return; }
    sub4(r->l, F->p, a->l);
}
/* Montgomery product, coarsely integrated operand scanning */
static inline void fe_mul(fe *r, const fe *a, const fe *b, const field_t *F) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        u64 m = t[0] * F->inv;
        c = (u128)m * F->p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * F->p[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    if (t[4] || geq4(t, F->p)) sub4(t, t, F->p);
    memcpy(r->l, t, 32);
}
static inline void fe_sqr(fe *r, const fe *a, const field_t *F) { fe_mul(r, a, a, F); }
static inline void fe_dbl(fe *r, const fe *a, const field_t *F) { fe_add(r, a, a, F); }
static void fe_from_mont(fe *r, const fe *a, const field_t *F) {
    fe one = {{1, 0, 0, 0}};
    fe_mul(r, a, &one, F);
}
static void fe_to_mont(fe *r, const fe *a, const field_t *F) {
    fe r2;
    memcpy(r2.l, F->r2, 32);
    fe_mul(r, a, &r2, F);
}
static void fe_pow(fe *r, const fe *a, const u64 e[4], const field_t *F) {
    fe acc;
    memcpy(acc.l, F->r1, 32);
    for (int i = 255; i >= 0; --i) {
        fe_sqr(&acc, &acc, F);
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, a, F);
    }
    *r = acc;
}
static void fe_inv(fe *r, const fe *a, const field_t *F) {
    u64 e[4];
    u64 two[4] = {2, 0, 0, 0};
    sub4(e, F->p, two);
    fe_pow(r, a, e, F);
}

/* ---------------------------------------------------------------- G1 (a = 0, b = 3) */
typedef struct { fe x, y; } aff_t;
typedef struct { fe x, y, z; } jac_t;

static inline int aff_is_inf(const aff_t *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline void jac_set_inf(jac_t *p) { memset(p, 0, sizeof *p); memcpy(p->y.l, FQ.r1, 32); }

static void jac_double(jac_t *r, const jac_t *p) {
    if (fe_is_zero(&p->z) || fe_is_zero(&p->y)) { jac_set_inf(r); return; }
    const field_t *F = &FQ;
    fe A, B, C, D, E, Fv, t;
    fe_sqr(&A, &p->x, F);
    fe_sqr(&B, &p->y, F);
    fe_sqr(&C, &B, F);
    fe_add(&t, &p->x, &B, F);
    fe_sqr(&t, &t, F);
    fe_sub(&t, &t, &A, F);
    fe_sub(&t, &t, &C, F);
    fe_dbl(&D, &t, F);
    fe_dbl(&E, &A, F);
    fe_add(&E, &E, &A, F);
    fe_sqr(&Fv, &E, F);
    fe z3;
    fe_mul(&z3, &p->y, &p->z, F);
    fe_dbl(&z3, &z3, F);
    fe x3;
    fe_dbl(&t, &D, F);
    fe_sub(&x3, &Fv, &t, F);
    fe y3;
    fe_sub(&t, &D, &x3, F);
    fe_mul(&y3, &E, &t, F);
    fe_dbl(&C, &C, F);
    fe_dbl(&C, &C, F);
    fe_dbl(&C, &C, F);
    fe_sub(&y3, &y3, &C, F);
    r->x = x3; r->y = y3; r->z = z3;
}

static void jac_add_mixed(jac_t *r, const jac_t *p, const aff_t *q) {
    if (aff_is_inf(q)) { *r = *p; return; }
    if (fe_is_zero(&p->z)) {
        r->x = q->x; r->y = q->y; memcpy(r->z.l, FQ.r1, 32);
        return;
    }
    const field_t *F = &FQ;
    fe z1z1, u2, s2, h, rr, hh, hhh, v, t;
    fe_sqr(&z1z1, &p->z, F);
    fe_mul(&u2, &q->x, &z1z1, F);
    fe_mul(&s2, &q->y, &p->z, F);
    fe_mul(&s2, &s2, &z1z1, F);
    if (fe_eq(&u2, &p->x)) {
        if (fe_eq(&s2, &p->y)) { jac_double(r, p); return; }
        jac_set_inf(r);
        return;
    }
    fe_sub(&h, &u2, &p->x, F);
    fe_sub(&rr, &s2, &p->y, F);
    fe_sqr(&hh, &h, F);
    fe_mul(&hhh, &hh, &h, F);
    fe_mul(&v, &p->x, &hh, F);
    fe x3, y3, z3;
    fe_sqr(&x3, &rr, F);
    fe_sub(&x3, &x3, &hhh, F);
    fe_dbl(&t, &v, F);
    fe_sub(&x3, &x3, &t, F);
    fe_sub(&t, &v, &x3, F);
    fe_mul(&y3, &rr, &t, F);
    fe_mul(&t, &p->y, &hhh, F);
    fe_sub(&y3, &y3, &t, F);
    fe_mul(&z3, &p->z, &h, F);
    r->x = x3; r->y = y3; r->z = z3;
}

static void jac_add(jac_t *r, const jac_t *p, const jac_t *q) {
    if (fe_is_zero(&p->z)) { *r = *q; return; }
    if (fe_is_zero(&q->z)) { *r = *p; return; }
    const field_t *F = &FQ;
    fe z1z1, z2z2, u1, u2, s1, s2, h, rr, hh, hhh, v, t;
    fe_sqr(&z1z1, &p->z, F);
    fe_sqr(&z2z2, &q->z, F);
    fe_mul(&u1, &p->x, &z2z2, F);
    fe_mul(&u2, &q->x, &z1z1, F);
    fe_mul(&s1, &p->y, &q->z, F);
    fe_mul(&s1, &s1, &z2z2, F);
    fe_mul(&s2, &q->y, &p->z, F);
    fe_mul(&s2, &s2, &z1z1, F);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { jac_double(r, p); return; }
        jac_set_inf(r);
        return;
    }
    fe_sub(&h, &u2, &u1, F);
    fe_sub(&rr, &s2, &s1, F);
    fe_sqr(&hh, &h, F);
    fe_mul(&hhh, &hh, &h, F);
    fe_mul(&v, &u1, &hh, F);
    fe x3, y3, z3;
    fe_sqr(&x3, &rr, F);
    fe_sub(&x3, &x3, &hhh, F);
    fe_dbl(&t, &v, F);
    fe_sub(&x3, &x3, &t, F);
    fe_sub(&t, &v, &x3, F);
    fe_mul(&y3, &rr, &t, F);
    fe_mul(&t, &s1, &hhh, F);
    fe_sub(&y3, &y3, &t, F);
    fe_mul(&z3, &p->z, &q->z, F);
    fe_mul(&z3, &z3, &h, F);
    r->x = x3; r->y = y3; r->z = z3;
}

static void jac_to_aff(aff_t *r, const jac_t *p) {
    if (fe_is_zero(&p->z)) { memset(r, 0, sizeof *r); return; }
    fe zi, zi2, zi3;
    fe_inv(&zi, &p->z, &FQ);
    fe_sqr(&zi2, &zi, &FQ);
    fe_mul(&zi3, &zi2, &zi, &FQ);
    fe_mul(&r->x, &p->x, &zi2, &FQ);
    fe_mul(&r->y, &p->y, &zi3, &FQ);
}

/* k: canonical (non-Montgomery) 256-bit scalar */
static void jac_mul_aff(jac_t *r, const aff_t *p, const u64 k[4]) {
    jac_t acc;
    jac_set_inf(&acc);
    for (int i = 255; i >= 0; --i) {
        jac_double(&acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) jac_add_mixed(&acc, &acc, p);
    }
    *r = acc;
}

/* ---------------------------------------------------------------- exported: field helpers */
int ora_fr_mul(const u64 a[4], const u64 b[4], u64 out[4]) {
    fe_mul((fe *)out, (const fe *)a, (const fe *)b, &FR);
    return 0;
}
int ora_fq_mul(const u64 a[4], const u64 b[4], u64 out[4]) {
    fe_mul((fe *)out, (const fe *)a, (const fe *)b, &FQ);
    return 0;
}
/* which: 0 = Fq, 1 = Fr.  to_mont != 0: canonical -> Montgomery, else the inverse. n elements. */
int ora_mont_convert(u64 *a, size_t n, int which, int to_mont) {
    const field_t *F = which ? &FR : &FQ;
    for (size_t i = 0; i < n; ++i) {
        fe *e = (fe *)(a + 4 * i);
        if (to_mont) fe_to_mont(e, e, F); else fe_from_mont(e, e, F);
    }
    return 0;
}

int ora_g1_normalize(const u64 jac[12], u64 aff[8]) {
    jac_to_aff((aff_t *)aff, (const jac_t *)jac);
    return 0;
}
/* out_jac = [k] base, k canonical */
int ora_g1_mul(const u64 base_aff[8], const u64 k[4], u64 out_jac[12]) {
    jac_mul_aff((jac_t *)out_jac, (const aff_t *)base_aff, k);
    return 0;
}
int ora_g1_add(const u64 a[12], const u64 b[12], u64 out[12]) {
    jac_t r;
    jac_add(&r, (const jac_t *)a, (const jac_t *)b);
    memcpy(out, &r, sizeof r);
    return 0;
}
int ora_g1_on_curve(const u64 aff[8]) {
    const aff_t *p = (const aff_t *)aff;
    if (aff_is_inf(p)) return 1;
    fe y2, x3, b3 = {{3, 0, 0, 0}};
    fe_to_mont(&b3, &b3, &FQ);
    fe_sqr(&y2, &p->y, &FQ);
    fe_sqr(&x3, &p->x, &FQ);
    fe_mul(&x3, &x3, &p->x, &FQ);
    fe_add(&x3, &x3, &b3, &FQ);
    return fe_eq(&y2, &x3);
}

/* P_i = [s + i t] G, i < n, affine Montgomery out.  s, t canonical scalars. */
int ora_walk_bases(u64 *out_aff, size_t n, const u64 s[4], const u64 t[4]) {
    aff_t G;
    fe one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}};
    fe_to_mont(&G.x, &one, &FQ);
    fe_to_mont(&G.y, &two, &FQ);
    jac_t cur, step_j;
    aff_t step;
    jac_mul_aff(&cur, &G, s);
    jac_mul_aff(&step_j, &G, t);
    jac_to_aff(&step, &step_j);
    enum { CH = 1024 };
    jac_t *buf = (jac_t *)malloc(sizeof(jac_t) * CH);
    fe *pref = (fe *)malloc(sizeof(fe) * CH);
    if (!buf || !pref) return -1;
    for (size_t base = 0; base < n; base += CH) {
        size_t m = n - base < CH ? n - base : CH;
        for (size_t i = 0; i < m; ++i) {
            buf[i] = cur;
            jac_add_mixed(&cur, &cur, &step);
        }
        /* batch inversion of z (identity points get z := 1 in the product) */
        fe acc;
        memcpy(acc.l, FQ.r1, 32);
        for (size_t i = 0; i < m; ++i) {
            pref[i] = acc;
            if (!fe_is_zero(&buf[i].z)) fe_mul(&acc, &acc, &buf[i].z, &FQ);
        }
        fe inv;
        fe_inv(&inv, &acc, &FQ);
        for (size_t i = m; i-- > 0;) {
            aff_t *o = (aff_t *)(out_aff + 8 * (base + i));
            if (fe_is_zero(&buf[i].z)) { memset(o, 0, sizeof *o); continue; }
            fe zi, zi2, zi3;
            fe_mul(&zi, &inv, &pref[i], &FQ);
            fe_mul(&inv, &inv, &buf[i].z, &FQ);
            fe_sqr(&zi2, &zi, &FQ);
            fe_mul(&zi3, &zi2, &zi, &FQ);
            fe_mul(&o->x, &buf[i].x, &zi2, &FQ);
            fe_mul(&o->y, &buf[i].y, &zi3, &FQ);
        }
    }
    free(buf);
    free(pref);
    return 0;
}

/* ---------------------------------------------------------------- MSM: best_multiexp restated */
static inline unsigned get_at(unsigned segment, unsigned c, const u64 k[4]) {
    unsigned skip_bits = segment * c;
    if (skip_bits >= 256) return 0;
    unsigned limb = skip_bits >> 6, off = skip_bits & 63;
    u64 v = k[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
    return (unsigned)(v & ((1ULL << c) - 1));
}

/* One thread's share, as halo2curves' multiexp_serial: window bits c ~ ln(n), windows from the top, bucket running sums.  Two
 * standard refinements of the same algorithm keep this CPU baseline honest ("signed-digit Pippenger with a
 * per-thread window choice"): digits are recoded to (-2^(c-1), 2^(c-1)] with a carry into the next window, so a window has
 * 2^(c-1) buckets (a negative digit adds the negated point), and c is taken one larger than ln(n) since buckets cost half.
 * The VALUE is the same group element; Jacobian representatives are never compared (tests normalise). */
static void multiexp_serial(const fe *coeffs_canon, const aff_t *bases, size_t n, jac_t *acc) {
    unsigned c;
    if (n < 4) c = 1;
    else if (n < 32) c = 3;
    else c = (unsigned)ceil(log((double)n)) + 1;
    if (c > 16) c = 16;
    unsigned segments = 256 / c + 2;                 /* + 1 for the final carry */
    size_t nb = (size_t)1 << (c - 1);
    jac_t *buckets = (jac_t *)malloc(sizeof(jac_t) * nb);
    /* signed digits of every scalar, window-major would cost n * segments bytes * 2: recode on the fly per window instead,
     * carrying per scalar in a byte array (the carry into window s depends only on windows below s: walk windows bottom-up
     * once to record the carries) */
    unsigned char *carry_in = (unsigned char *)calloc(n * (size_t)segments, 1);
    for (size_t i = 0; i < n; ++i) {
        unsigned carry = 0;
        for (unsigned seg = 0; seg < segments; ++seg) {
            carry_in[(size_t)seg * n + i] = (unsigned char)carry;
            unsigned d = get_at(seg, c, coeffs_canon[i].l) + carry;
            carry = d > (1u << (c - 1)) ? 1u : 0u;
        }
    }
    for (unsigned seg = segments; seg-- > 0;) {
        for (unsigned i = 0; i < c; ++i) jac_double(acc, acc);
        for (size_t b = 0; b < nb; ++b) jac_set_inf(&buckets[b]);
        for (size_t i = 0; i < n; ++i) {
            unsigned d = get_at(seg, c, coeffs_canon[i].l) + carry_in[(size_t)seg * n + i];
            if (!d) continue;
            if (d > (1u << (c - 1))) {               /* digit d - 2^c < 0: add -P to bucket 2^c - d */
                if (d == (1u << c)) continue;        /* 2^c - 2^c = 0 with a carry out: nothing in this window */
                aff_t neg = bases[i];
                if (!aff_is_inf(&neg)) fe_neg(&neg.y, &neg.y, &FQ);
                jac_add_mixed(&buckets[(1u << c) - d - 1], &buckets[(1u << c) - d - 1], &neg);
            } else {
                jac_add_mixed(&buckets[d - 1], &buckets[d - 1], &bases[i]);
            }
        }
        jac_t run;
        jac_set_inf(&run);
        for (size_t b = nb; b-- > 0;) {
            jac_add(&run, &run, &buckets[b]);
            jac_add(acc, acc, &run);
        }
    }
    free(carry_in);
    free(buckets);
}

/* scalars: n x 4 Montgomery Fr; bases: n x 8 Montgomery affine; threads <= 0 -> all cores */
int ora_msm_g1(const u64 *scalars, const u64 *bases, size_t n, int threads, u64 out_jac[12]) {
    jac_t total;
    jac_set_inf(&total);
    if (n == 0) { memcpy(out_jac, &total, sizeof total); return 0; }
    fe *canon = (fe *)malloc(sizeof(fe) * n);
    if (!canon) return -1;
#ifdef _OPENMP
    if (threads <= 0) threads = ora_num_threads();
#else
    threads = 1;
#endif
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long i = 0; i < (long)n; ++i) fe_from_mont(&canon[i], (const fe *)(scalars + 4 * i), &FR);
    if (n > (size_t)threads) {
        size_t chunk = n / (size_t)threads;
        size_t nchunks = (n + chunk - 1) / chunk;
        jac_t *res = (jac_t *)malloc(sizeof(jac_t) * nchunks);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (long ci = 0; ci < (long)nchunks; ++ci) {
            size_t lo = (size_t)ci * chunk;
            size_t m = n - lo < chunk ? n - lo : chunk;
            jac_set_inf(&res[ci]);
            multiexp_serial(canon + lo, (const aff_t *)bases + lo, m, &res[ci]);
        }
        for (size_t ci = 0; ci < nchunks; ++ci) jac_add(&total, &total, &res[ci]);
        free(res);
    } else {
        multiexp_serial(canon, (const aff_t *)bases, n, &total);
    }
    free(canon);
    memcpy(out_jac, &total, sizeof total);
    return 0;
}

/* ---------------------------------------------------------------- NTT: best_fft restated */
static inline size_t bitrev(size_t k, unsigned l) {
    size_t r = 0;
    for (unsigned i = 0; i < l; ++i) { r = (r << 1) | (k & 1); k >>= 1; }
    return r;
}

/* The butterflies are exactly best_fft's serial layers (bit reversal, then log_n radix-2 DIT layers over a twiddle table
 * omega^i); only the ORDER differs between threads: the first log_n - log_t layers never leave a contiguous block of n / 2^log_t
 * elements, so each thread runs them on its own blocks without a barrier (cache-resident), and only the last log_t layers are
 * swept by all threads together -- halo2curves' best_fft splits the same way (recursive halves per thread).  The field
 * operations are associative-free (each element's value is the same function of the inputs in any order): results are
 * bit-identical to the serial layers. */
int ora_ntt_fr(u64 *a_, const u64 omega_[4], uint32_t log_n, int threads) {
    fe *a = (fe *)a_;
    size_t n = (size_t)1 << log_n;
    const field_t *F = &FR;
#ifdef _OPENMP
    if (threads <= 0) threads = ora_num_threads();
#else
    threads = 1;
#endif
    if (n < 4096) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (long k = 0; k < (long)n; ++k) {
        size_t rk = bitrev((size_t)k, log_n);
        if ((size_t)k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; }
    }
    if (log_n == 0) return 0;
    size_t half = n / 2;
    fe *tw = (fe *)malloc(sizeof(fe) * half);
    if (!tw) return -1;
    {   /* omega^i, i < n/2: blocks of 1024 started by square-and-multiply, then successive products */
        const size_t TB = 1024;
        size_t nblk = (half + TB - 1) / TB;
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
        for (long b = 0; b < (long)nblk; ++b) {
            size_t lo = (size_t)b * TB, hi = lo + TB < half ? lo + TB : half;
            u64 e[4] = {(u64)lo, 0, 0, 0};
            fe w;
            fe_pow(&w, (const fe *)omega_, e, F);
            for (size_t i = lo; i < hi; ++i) {
                tw[i] = w;
                fe_mul(&w, &w, (const fe *)omega_, F);
            }
        }
    }
    /* blocks of 2^log_b elements per task for the barrier-free layers */
    uint32_t log_b = log_n;
    if (threads > 1) {
        uint32_t log_t = 0;
        while ((1u << log_t) < (unsigned)threads * 4u) ++log_t;   /* a few blocks per thread */
        log_b = log_n > log_t ? log_n - log_t : 0;
        if (log_b > 14) log_b = 14;                               /* 512 KiB of elements: stays in a core's L2 */
    }
    size_t nb = n >> log_b, bsz = (size_t)1 << log_b;
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (long b = 0; b < (long)nb; ++b) {
        fe *base = a + (size_t)b * bsz;
        size_t chunk = 2, tchunk = half;
        for (uint32_t layer = 0; layer < log_b; ++layer) {
            size_t hc = chunk / 2;
            for (size_t blk = 0; blk < bsz; blk += chunk)
                for (size_t i = 0; i < hc; ++i) {
                    fe *lo = base + blk + i, *hi = lo + hc;
                    fe t;
                    if (i == 0) t = *hi; else fe_mul(&t, hi, &tw[i * tchunk], F);
                    fe u = *lo;
                    fe_add(lo, &u, &t, F);
                    fe_sub(hi, &u, &t, F);
                }
            chunk *= 2;
            tchunk /= 2;
        }
    }
    size_t chunk = (size_t)2 << log_b, tchunk = half >> log_b;
    for (uint32_t layer = log_b; layer < log_n; ++layer) {
        size_t hc = chunk / 2;
        size_t nblk = n / chunk;
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
        for (long bj = 0; bj < (long)(nblk * hc); ++bj) {
            size_t blk = (size_t)bj / hc, i = (size_t)bj % hc;
            fe *lo = a + blk * chunk + i;
            fe *hi = lo + hc;
            fe t;
            if (i == 0) t = *hi; else fe_mul(&t, hi, &tw[i * tchunk], F);
            fe u = *lo;
            fe_add(lo, &u, &t, F);
            fe_sub(hi, &u, &t, F);
        }
        chunk *= 2;
        tchunk /= 2;
    }
    free(tw);
    return 0;
}

/* a[i] *= scale (Montgomery) -- the ifft divisor step */
int ora_fr_scale(u64 *a_, size_t n, const u64 scale[4]) {
    fe *a = (fe *)a_;
    for (size_t i = 0; i < n; ++i) fe_mul(&a[i], &a[i], (const fe *)scale, &FR);
    return 0;
}
/* field-multiplication throughput of this host (bench.py's cpu_baseline leg prices the prover phases that are plain field arithmetic --
 * grand products, evaluate_h, evaluations, the SHPLONK fold -- with it): `reps` passes of a[i] = a[i] * b[i] over n elements on `threads`
 * OpenMP threads, the way halo2's parallelize() splits such loops.  -> seconds (n * reps Montgomery multiplications). */
double ora_fr_mul_seconds(size_t n, int reps, int threads) {
    fe *a = (fe *)malloc(n * sizeof(fe)), *b = (fe *)malloc(n * sizeof(fe));
    if (!a || !b) {
        free(a);
        free(b);
        return -1.0;
    }
    for (size_t i = 0; i < n; ++i) {
        memcpy(a[i].l, FR.r1, 32);
        memcpy(b[i].l, FR.r1, 32);
        a[i].l[0] ^= (u64)i * 0x9e3779b97f4a7c15ULL;
        a[i].l[3] &= 0x0fffffffffffffffULL;
        b[i].l[1] ^= (u64)(i + 7) * 0xbf58476d1ce4e5b9ULL;
        b[i].l[3] &= 0x0fffffffffffffffULL;
    }
    if (threads < 1) threads = 1;
    const double t0 = omp_get_wtime();
    for (int r = 0; r < reps; ++r) {
#pragma omp parallel for num_threads(threads) schedule(static)
        for (long i = 0; i < (long)n; ++i) fe_mul(&a[i], &a[i], &b[i], &FR);
    }
    const double dt = omp_get_wtime() - t0;
    volatile u64 sink = a[n / 2].l[0];
    (void)sink;
    free(a);
    free(b);
    return dt;
}
/* a[i] *= g^i (distribute_powers) */
int ora_fr_distribute_powers(u64 *a_, size_t n, const u64 g[4]) {
    fe *a = (fe *)a_;
    fe cur;
    memcpy(cur.l, FR.r1, 32);
    for (size_t i = 0; i < n; ++i) {
        fe_mul(&a[i], &a[i], &cur, &FR);
        fe_mul(&cur, &cur, (const fe *)g, &FR);
    }
    return 0;
}

/* ---------------------------------------------------------------- big integers (u64 LE limbs) */
static size_t bn_len(const u64 *a, size_t n) {
    while (n && a[n - 1] == 0) --n;
    return n;
}
static void bn_mul(u64 *r, const u64 *a, size_t na, const u64 *b, size_t nb) {
    memset(r, 0, 8 * (na + nb));
    for (size_t i = 0; i < na; ++i) {
        u128 c = 0;
        u64 ai = a[i];
        if (!ai) continue;
        for (size_t j = 0; j < nb; ++j) {
            c += (u128)ai * b[j] + r[i + j];
            r[i + j] = (u64)c;
            c >>= 64;
        }
        r[i + nb] = (u64)c;
    }
}
/* Knuth algorithm D: u (nu limbs) / v (nv significant limbs, v != 0) -> q (nu limbs), rem (nv limbs) */
static int bn_divrem(u64 *q, u64 *rem, const u64 *u, size_t nu, const u64 *v, size_t nv_full) {
    size_t nv = bn_len(v, nv_full);
    if (nv == 0) return -1;
    memset(q, 0, 8 * nu);
    memset(rem, 0, 8 * nv_full);
    size_t mu = bn_len(u, nu);
    if (mu < nv) { memcpy(rem, u, 8 * mu); return 0; }
    if (nv == 1) {
        u128 r = 0;
        for (size_t i = mu; i-- > 0;) {
            u128 cur = (r << 64) | u[i];
            q[i] = (u64)(cur / v[0]);
            r = cur % v[0];
        }
        rem[0] = (u64)r;
        return 0;
    }
    int s = __builtin_clzll(v[nv - 1]);
    u64 *vn = (u64 *)malloc(8 * nv);
    u64 *un = (u64 *)malloc(8 * (mu + 1));
    for (size_t i = nv - 1; i > 0; --i) vn[i] = s ? (v[i] << s) | (v[i - 1] >> (64 - s)) : v[i];
    vn[0] = v[0] << s;
    un[mu] = s ? u[mu - 1] >> (64 - s) : 0;
    for (size_t i = mu - 1; i > 0; --i) un[i] = s ? (u[i] << s) | (u[i - 1] >> (64 - s)) : u[i];
    un[0] = u[0] << s;
    for (size_t j = mu - nv + 1; j-- > 0;) {
        u128 num = ((u128)un[j + nv] << 64) | un[j + nv - 1];
        u128 qhat = num / vn[nv - 1];
        u128 rhat = num % vn[nv - 1];
        while ((qhat >> 64) || (u128)(u64)qhat * vn[nv - 2] > ((rhat << 64) | un[j + nv - 2])) {
            qhat -= 1;
            rhat += vn[nv - 1];
            if (rhat >> 64) break;
        }
        u128 borrow = 0, carry = 0;
        for (size_t i = 0; i < nv; ++i) {
            carry += (u128)(u64)qhat * vn[i];
            u128 t = (u128)un[i + j] - (u64)carry - (u64)borrow;
            un[i + j] = (u64)t;
            borrow = (t >> 64) & 1;
            carry >>= 64;
        }
        u128 t = (u128)un[j + nv] - (u64)carry - (u64)borrow;
        un[j + nv] = (u64)t;
        if ((t >> 64) & 1) {
            qhat -= 1;
            u128 c = 0;
            for (size_t i = 0; i < nv; ++i) {
                c += (u128)un[i + j] + vn[i];
                un[i + j] = (u64)c;
                c >>= 64;
            }
            un[j + nv] += (u64)c;
        }
        q[j] = (u64)qhat;
    }
    for (size_t i = 0; i < nv; ++i)
        rem[i] = s ? (un[i] >> s) | (un[i + 1] << (64 - s)) : un[i];
    free(vn);
    free(un);
    return 0;
}

/* a, b, mod: L limbs; q: L limbs; r: L limbs.  Returns -1 if mod == 0, -2 if q >= 2^(64 L)
   (the circuit assigns q with L limbs, so such an instance is unsatisfiable in the reference). */
int ora_mul_mod_step(uint32_t L, const u64 *a, const u64 *b, const u64 *mod, u64 *q, u64 *r) {
    u64 *full = (u64 *)malloc(8 * 2 * L);
    u64 *qq = (u64 *)malloc(8 * 2 * L);
    bn_mul(full, a, L, b, L);
    int rc = bn_divrem(qq, r, full, 2 * L, mod, L);
    if (rc == 0) {
        if (bn_len(qq, 2 * L) > L) rc = -2;
        memcpy(q, qq, 8 * L);
    }
    free(full);
    free(qq);
    return rc;
}

/* pow_mod_fixed_exp step trace.  steps_out: per step 4*L limbs laid out a|b|q|r.  The caller
   sizes it for 2*bits(e) steps.  result: L limbs. */
int ora_pow_mod_trace(uint32_t L, const u64 *mod, const u64 *base, const u64 *exp, uint32_t exp_limbs,
                      u64 *steps_out, size_t *n_steps, u64 *result) {
    size_t el = bn_len(exp, exp_limbs);
    size_t nbits = el ? 64 * (el - 1) + (64 - (size_t)__builtin_clzll(exp[el - 1])) : 0;
    u64 *acc = (u64 *)calloc(L, 8);
    u64 *sq = (u64 *)malloc(8 * L);
    acc[0] = 1;
    memcpy(sq, base, 8 * L);
    size_t ns = 0;
    int rc = 0;
    for (size_t i = 0; i < nbits && rc == 0; ++i) {
        u64 *st = steps_out + ns * 4 * L;
        memcpy(st, sq, 8 * L);
        memcpy(st + L, sq, 8 * L);
        rc = ora_mul_mod_step(L, sq, sq, mod, st + 2 * L, st + 3 * L);
        if (rc) break;
        ++ns;
        if ((exp[i >> 6] >> (i & 63)) & 1) {
            u64 *st2 = steps_out + ns * 4 * L;
            memcpy(st2, acc, 8 * L);
            memcpy(st2 + L, st, 8 * L); /* cur = the pre-square value */
            rc = ora_mul_mod_step(L, acc, st, mod, st2 + 2 * L, st2 + 3 * L);
            if (rc) break;
            memcpy(acc, st2 + 3 * L, 8 * L);
            ++ns;
        }
        memcpy(sq, st + 3 * L, 8 * L);
    }
    *n_steps = ns;
    memcpy(result, acc, 8 * L);
    free(acc);
    free(sq);
    return rc;
}

/* paillier_enc_native with n given in Ln limbs; g, m, r in Ln limbs; out c in 2*Ln limbs.
   Uses the same square-and-multiply schedule (value is schedule independent). */
int ora_paillier_enc(uint32_t Ln, const u64 *n, const u64 *g, const u64 *m, const u64 *r, u64 *c_out) {
    uint32_t L = 2 * Ln;
    u64 *n2 = (u64 *)malloc(8 * L);
    u64 *ge = (u64 *)calloc(L, 8), *re = (u64 *)calloc(L, 8);
    u64 *gm = (u64 *)malloc(8 * L), *rn = (u64 *)malloc(8 * L), *q = (u64 *)malloc(8 * L);
    bn_mul(n2, n, Ln, n, Ln);
    memcpy(ge, g, 8 * Ln);
    memcpy(re, r, 8 * Ln);
    size_t cap = 2 * 64 * (size_t)Ln + 2;
    u64 *steps = (u64 *)malloc(8 * 4 * L * cap);
    size_t ns;
    int rc = ora_pow_mod_trace(L, n2, ge, m, Ln, steps, &ns, gm);
    if (!rc) rc = ora_pow_mod_trace(L, n2, re, n, Ln, steps, &ns, rn);
    if (!rc) rc = ora_mul_mod_step(L, gm, rn, n2, q, c_out);
    free(n2); free(ge); free(re); free(gm); free(rn); free(q); free(steps);
    return rc;
}

/* threads worth starting: the OpenMP default, capped by the CPU quota of the cgroup (a GPU box shows 256 logical CPUs and grants
 * 16: 256 threads on a 16-CPU quota spend their time throttled at the barriers) */
int ora_num_threads(void) {
#ifdef _OPENMP
    static int cached = 0;
    if (cached) return cached;
    int t = omp_get_max_threads();
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            long quota = atol(q);
            int cpus = (int)((quota + period - 1) / period);
            if (cpus >= 1 && cpus < t) t = cpus;
        }
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {
        long quota = -1, period = 100000;
        if (fscanf(f, "%ld", &quota) != 1) quota = -1;
        fclose(f);
        FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (g) { if (fscanf(g, "%ld", &period) != 1) period = 100000; fclose(g); }
        if (quota > 0 && period > 0) {
            int cpus = (int)((quota + period - 1) / period);
            if (cpus >= 1 && cpus < t) t = cpus;
        }
    }
    cached = t < 1 ? 1 : t;
    return cached;
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- MockProver analogue on a whole cell stream
 * halo2-lib has ONE gate, q * (a + b*c - d) = 0 on four vertically consecutive cells.  cells: n
 * Montgomery Fr elements in stream order; sel[i] != 0 where the selector is enabled (the window starts at cell i; i + 3 < n).
 * Returns the number of windows whose identity fails, *first_bad = the first such offset (n if none).  Montgomery forms:
 * mont(b) * mont(c) = mont(bc), so the identity is checked in the Montgomery domain as it stands.
 * (reference: base_test().expect_satisfied(true) -> MockProver, /root/reference/src/paillier.rs:167-171) */
size_t ora_check_gates(const u64 *cells, const uint8_t *sel, size_t n, size_t *first_bad) {
    size_t bad = 0, first = n;
#pragma omp parallel for schedule(static) reduction(+ : bad) reduction(min : first)
    for (size_t i = 0; i < n; ++i) {
        if (!sel[i]) continue;
        if (i + 3 >= n) { bad++; if (i < first) first = i; continue; }
        const fe *a = (const fe *)(cells + 4 * i), *b = a + 1, *c = a + 2, *d = a + 3;
        fe t, u;
        fe_mul(&t, b, c, &FR);
        fe_add(&u, a, &t, &FR);
        fe_sub(&t, &u, d, &FR);
        if (!fe_is_zero(&t)) { bad++; if (i < first) first = i; }
    }
    if (first_bad) *first_bad = first;
    return bad;
}
/* lookup-advice cells (range-check digits) must be canonical integers below 2^bits: the table the reference's RangeChip
 * looks them up in is [0, 2^lookup_bits).  Returns the number of cells outside it. */
size_t ora_check_range(const u64 *cells, size_t n, uint32_t bits, size_t *first_bad) {
    size_t bad = 0, first = n;
#pragma omp parallel for schedule(static) reduction(+ : bad) reduction(min : first)
    for (size_t i = 0; i < n; ++i) {
        fe v;
        fe_from_mont(&v, (const fe *)(cells + 4 * i), &FR);
        int ok = v.l[1] == 0 && v.l[2] == 0 && v.l[3] == 0 && (bits >= 64 || (v.l[0] >> bits) == 0);
        if (!ok) { bad++; if (i < first) first = i; }
    }
    if (first_bad) *first_bad = first;
    return bad;
}
