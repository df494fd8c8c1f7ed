// blake2b.hpp -- BLAKE2b-512 (RFC 7693), unkeyed, with a 16-byte personalisation: the hash behind the drivers' Fiat-Shamir transcript
// (create_proof.hpp::Transcript).  halo2's transcript (`Blake2bWrite`, halo2_proofs transcript/blake2b.rs [D]) is this hash personalised
// "Halo2-Transcript"; a challenge there is the digest of a CLONE of the running state, which is why the state is copyable and `digest()`
// is const.  Host code, a few MB per proof; pinned by RFC 7693's "abc" vector and against Python's hashlib (tests/test_cpp_host_field.py).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace pzh {

class Blake2b {
public:
    explicit Blake2b(const char personal[16] = nullptr) {
        static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        memcpy(h_, iv, 64);
        h_[0] ^= 0x01010000ULL ^ 64;   // parameter block: digest length 64, no key, fanout 1, depth 1
        if (personal) {
            uint64_t p[2];
            memcpy(p, personal, 16);
            h_[6] ^= p[0];
            h_[7] ^= p[1];
        }
    }
    void update(const void* data, size_t len) {
        const uint8_t* in = (const uint8_t*)data;
        while (len) {
            if (fill_ == 128) {   // the buffer is only compressed once more input arrives: the last block is flagged as final
                bump(128);
                compress(buf_, false);
                fill_ = 0;
            }
            const size_t take = len < 128 - fill_ ? len : 128 - fill_;
            memcpy(buf_ + fill_, in, take);
            fill_ += take;
            in += take;
            len -= take;
        }
    }
    void digest(uint8_t out[64]) const {   // of everything absorbed so far; the running state is untouched
        Blake2b c = *this;
        c.bump(c.fill_);
        memset(c.buf_ + c.fill_, 0, 128 - c.fill_);
        c.compress(c.buf_, true);
        memcpy(out, c.h_, 64);
    }

private:
    uint64_t h_[8], t_[2] = {0, 0};
    uint8_t buf_[128];
    size_t fill_ = 0;

    void bump(size_t n) {
        t_[0] += n;
        if (t_[0] < n) ++t_[1];
    }
    static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress(const uint8_t* block, bool last) {
        static const uint8_t sigma[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        uint64_t m[16], v[16];
        memcpy(m, block, 128);
        memcpy(v, h_, 64);
        memcpy(v + 8, iv, 64);
        v[12] ^= t_[0];
        v[13] ^= t_[1];
        if (last) v[14] = ~v[14];
        auto g = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x;
            v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y;
            v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; ++r) {
            const uint8_t* s = sigma[r];
            g(0, 4, 8, 12, m[s[0]], m[s[1]]);
            g(1, 5, 9, 13, m[s[2]], m[s[3]]);
            g(2, 6, 10, 14, m[s[4]], m[s[5]]);
            g(3, 7, 11, 15, m[s[6]], m[s[7]]);
            g(0, 5, 10, 15, m[s[8]], m[s[9]]);
            g(1, 6, 11, 12, m[s[10]], m[s[11]]);
            g(2, 7, 8, 13, m[s[12]], m[s[13]]);
            g(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; ++i) h_[i] ^= v[i] ^ v[i + 8];
    }
};

}   // namespace pzh
