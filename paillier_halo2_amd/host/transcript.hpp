// transcript.hpp -- the Fiat-Shamir transcript of the compiled provers' drivers (create_proof.hpp, prove_connected.cpp): host code between
// the phases of a proof, no device work.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "blake2b.hpp"
#include "fr_host.hpp"

namespace pzp {
using pzh::Fr;

// every phase's commitments come to the host in affine form (a synchronising download) and are hashed; a challenge is
// the hash of everything absorbed so far.  halo2's Blake2b transcript [D] (halo2_proofs transcript/blake2b.rs) in its PRIMITIVES:
// BLAKE2b-512 personalised "Halo2-Transcript", one domain byte in front of every item (0 challenge, 1 point, 2 scalar), a challenge = the
// digest of a CLONE of the running state read as a 512-bit little-endian integer mod r (`from_uniform_bytes`).  Not its byte format: a field
// element enters as the 4 Montgomery words it crosses include/pz.h in (x then y for a point), families in this prover's order, and the
// seed stands where halo2 absorbs the verifying key's digest.  paillier_halo2_amd/prover.py::HashTranscript is the same function;
// oracle/verifier.py::replay_challenges re-derives every challenge from a proof's commitments and evaluations.
struct Transcript {
    pzh::Blake2b h{"Halo2-Transcript"};
    std::vector<std::pair<std::string, Fr>> drawn;       // (name, canonical value as 4 words): what the checker compares its replay with
    Transcript(const void* seed, size_t bytes) { h.update(seed, bytes); }
    explicit Transcript(uint64_t seed) { h.update(&seed, 8); }
    void items(uint8_t tag, const uint64_t* v, size_t count, size_t words) {
        std::vector<uint8_t> buf(count * (1 + 8 * words));
        uint8_t* o = buf.data();
        for (size_t i = 0; i < count; ++i, o += 1 + 8 * words) {
            o[0] = tag;
            memcpy(o + 1, v + i * words, 8 * words);
        }
        h.update(buf.data(), buf.size());
    }
    void common_points(const uint64_t* aff, size_t count) { items(1, aff, count, 8); }
    void common_scalars(const uint64_t* s, size_t count) { items(2, s, count, 4); }
    Fr squeeze(const char* name) {   // -> Montgomery form; the canonical words are recorded for the checker
        const uint8_t tag = 0;
        h.update(&tag, 1);
        uint64_t d[8];
        h.digest((uint8_t*)d);
        // (lo + hi * 2^256) mod r: from_raw multiplies by 2^512 / 2^256, and the Montgomery form of 2^256 is 2^512 mod r itself
        const Fr c = pzh::add(pzh::from_raw(d), pzh::mul(pzh::from_raw(d + 4), pzh::FR_R2));
        Fr raw;
        pzh::to_raw(c, raw.v);
        drawn.push_back({name, raw});
        return c;
    }
};

}   // namespace pzp
