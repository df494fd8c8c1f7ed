// fr_host.hpp -- BN254 Fr on the HOST, for the handful of scalars a prover derives between kernel calls (powers of challenges, domain
// constants, their inverses).  4 x u64 little-endian limbs in MONTGOMERY form (R = 2^256): the layout of halo2curves' Fr and of every
// field element that crosses include/pz.h.  Not a compute path: bulk arithmetic lives in libpz_hip.so.
#pragma once
#include <cstdint>
#include <cstring>

namespace pzh {

struct Fr {
    uint64_t v[4];
    bool operator==(const Fr& o) const { return !memcmp(v, o.v, 32); }
};

static const uint64_t FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t FR_INV = 0xc2e1f593efffffffULL;   // -r^-1 mod 2^64
static const Fr FR_R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};   // 2^512 mod r
static const Fr FR_ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};  // 2^256 mod r
// ROOT_OF_UNITY of halo2curves (order 2^28), as a canonical integer
static const uint64_t FR_ROOT_RAW[4] = {0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL};

inline bool ge_mod(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i)
        if (a[i] != FR_MOD[i]) return a[i] > FR_MOD[i];
    return true;
}
inline void sub_mod(uint64_t a[4]) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned __int128 d = (unsigned __int128)a[i] - FR_MOD[i] - br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
// Montgomery product a * b / 2^256 mod r (CIOS)
inline Fr mul(const Fr& a, const Fr& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (unsigned __int128)a.v[j] * b.v[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * FR_INV;
        c = (unsigned __int128)m * FR_MOD[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (unsigned __int128)m * FR_MOD[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || ge_mod(r.v)) sub_mod(r.v);
    return r;
}
inline Fr from_raw(const uint64_t c[4]) {   // canonical integer below r -> Montgomery
    Fr x;
    memcpy(x.v, c, 32);
    return mul(x, FR_R2);
}
inline Fr from_u64(uint64_t c) {
    const uint64_t w[4] = {c, 0, 0, 0};
    return from_raw(w);
}
inline void to_raw(const Fr& a, uint64_t out[4]) {   // Montgomery -> canonical integer
    const Fr one = {{1, 0, 0, 0}};
    const Fr r = mul(a, one);
    memcpy(out, r.v, 32);
}
inline Fr add(const Fr& a, const Fr& b) {
    Fr r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (unsigned __int128)a.v[i] + b.v[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || ge_mod(r.v)) sub_mod(r.v);
    return r;
}
inline Fr neg(const Fr& a) {
    bool zero = !(a.v[0] | a.v[1] | a.v[2] | a.v[3]);
    if (zero) return a;
    Fr r;
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned __int128 d = (unsigned __int128)FR_MOD[i] - a.v[i] - br;
        r.v[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    return r;
}
inline Fr pow(const Fr& a, const uint64_t e[4]) {
    Fr acc = FR_ONE, sq = a;
    for (int i = 0; i < 256; ++i) {
        if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, sq);
        sq = mul(sq, sq);
    }
    return acc;
}
inline Fr pow_u64(const Fr& a, uint64_t e) {
    const uint64_t w[4] = {e, 0, 0, 0};
    return pow(a, w);
}
inline Fr inv(const Fr& a) {   // a^(r - 2)
    uint64_t e[4] = {FR_MOD[0] - 2, FR_MOD[1], FR_MOD[2], FR_MOD[3]};
    return pow(a, e);
}
// generator of the 2^log_n domain: ROOT_OF_UNITY^(2^(28 - log_n))
inline Fr omega(unsigned log_n) {
    Fr w = from_raw(FR_ROOT_RAW);
    for (unsigned i = log_n; i < 28; ++i) w = mul(w, w);
    return w;
}
// 7^((r - 1) / 3): the coset generator ZETA of halo2's EvaluationDomain; 7^(2^28): DELTA
inline Fr zeta() {
    // (r - 1) / 3
    uint64_t e[4];
    unsigned __int128 rem = 0;
    uint64_t m1[4] = {FR_MOD[0] - 1, FR_MOD[1], FR_MOD[2], FR_MOD[3]};
    for (int i = 3; i >= 0; --i) {
        const unsigned __int128 cur = (rem << 64) | m1[i];
        e[i] = (uint64_t)(cur / 3);
        rem = cur % 3;
    }
    return pow(from_u64(7), e);
}
inline Fr delta() {
    Fr d = from_u64(7);
    for (int i = 0; i < 28; ++i) d = mul(d, d);
    return d;
}

}   // namespace pzh
