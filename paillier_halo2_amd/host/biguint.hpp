// biguint.hpp -- minimal host-side arbitrary-precision unsigned integer for the C++ mirror of the
// reference's chip interface.  It only carries VALUES between API calls (limb split / join, equality,
// n*n, hex I/O); every modular multiplication / exponentiation of the hot path goes through the C ABI
// (include/pz.h) to the HIP kernels.  Little-endian u64 limbs, like num-bigint's BigUint semantics the
// reference uses (Cargo.toml:12).
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace pz {

class BigUint {
  public:
    std::vector<uint64_t> l;  // no trailing zero limbs
    BigUint() {}
    BigUint(uint64_t v) { if (v) l.push_back(v); }
    static BigUint from_limbs(const uint64_t* p, size_t n) {
        BigUint r;
        r.l.assign(p, p + n);
        r.trim();
        return r;
    }
    static BigUint from_hex(const std::string& s) {
        BigUint r;
        for (char c : s) {
            int v = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
            if (v < 0) throw std::invalid_argument("hex");
            r = (r << 4) + BigUint((uint64_t)v);
        }
        return r;
    }
    std::string to_hex() const {
        if (l.empty()) return "0";
        static const char* d = "0123456789abcdef";
        std::string s;
        for (size_t i = l.size(); i-- > 0;)
            for (int k = 60; k >= 0; k -= 4) s.push_back(d[(l[i] >> k) & 15]);
        size_t p = s.find_first_not_of('0');
        return s.substr(p);
    }
    void trim() { while (!l.empty() && l.back() == 0) l.pop_back(); }
    bool is_zero() const { return l.empty(); }
    size_t bits() const { return l.empty() ? 0 : 64 * (l.size() - 1) + (64 - (size_t)__builtin_clzll(l.back())); }
    bool bit(size_t i) const { return i / 64 < l.size() && ((l[i / 64] >> (i % 64)) & 1); }
    std::vector<uint64_t> to_limbs(size_t n) const {  // zero-extended; throws if it does not fit
        if (l.size() > n) throw std::range_error("integer does not fit the limb count");
        std::vector<uint64_t> r(l);
        r.resize(n, 0);
        return r;
    }
    friend bool operator==(const BigUint& a, const BigUint& b) { return a.l == b.l; }
    friend bool operator!=(const BigUint& a, const BigUint& b) { return !(a == b); }
    friend bool operator<(const BigUint& a, const BigUint& b) {
        if (a.l.size() != b.l.size()) return a.l.size() < b.l.size();
        for (size_t i = a.l.size(); i-- > 0;)
            if (a.l[i] != b.l[i]) return a.l[i] < b.l[i];
        return false;
    }
    friend BigUint operator+(const BigUint& a, const BigUint& b) {
        BigUint r;
        unsigned __int128 c = 0;
        size_t n = std::max(a.l.size(), b.l.size());
        for (size_t i = 0; i < n; ++i) {
            c += (unsigned __int128)(i < a.l.size() ? a.l[i] : 0) + (i < b.l.size() ? b.l[i] : 0);
            r.l.push_back((uint64_t)c);
            c >>= 64;
        }
        if (c) r.l.push_back((uint64_t)c);
        return r;
    }
    friend BigUint operator<<(const BigUint& a, size_t s) {
        if (a.l.empty()) return a;
        BigUint r;
        r.l.assign(s / 64, 0);
        unsigned sh = s % 64;
        uint64_t carry = 0;
        for (uint64_t v : a.l) {
            r.l.push_back(sh ? (v << sh) | carry : v);
            carry = sh ? v >> (64 - sh) : 0;
        }
        if (carry) r.l.push_back(carry);
        return r;
    }
    friend BigUint operator>>(const BigUint& a, size_t s) {
        BigUint r;
        const size_t w = s / 64;
        const unsigned sh = s % 64;
        for (size_t i = w; i < a.l.size(); ++i) {
            uint64_t v = a.l[i] >> sh;
            if (sh && i + 1 < a.l.size()) v |= a.l[i + 1] << (64 - sh);
            r.l.push_back(v);
        }
        r.trim();
        return r;
    }
    BigUint low_bits(size_t n) const {  // value mod 2^n
        BigUint r;
        for (size_t i = 0; i < l.size() && 64 * i < n; ++i)
            r.l.push_back(n - 64 * i >= 64 ? l[i] : l[i] & ((1ull << (n - 64 * i)) - 1));
        r.trim();
        return r;
    }
    friend BigUint operator*(const BigUint& a, const BigUint& b) {  // schoolbook: only for n*n-sized setup values
        BigUint r;
        if (a.l.empty() || b.l.empty()) return r;
        r.l.assign(a.l.size() + b.l.size(), 0);
        for (size_t i = 0; i < a.l.size(); ++i) {
            unsigned __int128 c = 0;
            for (size_t j = 0; j < b.l.size(); ++j) {
                c += (unsigned __int128)a.l[i] * b.l[j] + r.l[i + j];
                r.l[i + j] = (uint64_t)c;
                c >>= 64;
            }
            r.l[i + b.l.size()] = (uint64_t)c;
        }
        r.trim();
        return r;
    }
};

}  // namespace pz
