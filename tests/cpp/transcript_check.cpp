// prints BLAKE2b-512 digests and transcript challenges for fixed inputs; tests/test_cpp_host_field.py compares them with hashlib and
// with paillier_halo2_amd/prover.py::HashTranscript.  CPU only, no library.
#include <cstdio>
#include <string>

#include "../../paillier_halo2_amd/host/transcript.hpp"

static void hex(const char* name, const uint8_t* d, size_t n) {
    printf("%s ", name);
    for (size_t i = 0; i < n; ++i) printf("%02x", d[i]);
    printf("\n");
}

int main() {
    uint8_t d[64];
    {
        pzh::Blake2b h;   // RFC 7693 appendix A: BLAKE2b-512("abc")
        h.update("abc", 3);
        h.digest(d);
        hex("abc", d, 64);
    }
    {
        pzh::Blake2b h;
        h.digest(d);
        hex("empty", d, 64);
    }
    for (size_t len : {1u, 127u, 128u, 129u, 255u, 256u, 257u, 1000u, 4099u}) {   // block boundaries, fed in uneven pieces
        std::string s(len, 0);
        for (size_t i = 0; i < len; ++i) s[i] = (char)(i * 131 + 7);
        pzh::Blake2b h("Halo2-Transcript");
        size_t pos = 0, step = 1;
        while (pos < len) {
            const size_t take = step < len - pos ? step : len - pos;
            h.update(s.data() + pos, take);
            pos += take;
            step = step * 3 + 1;
        }
        h.digest(d);
        hex(("pers" + std::to_string(len)).c_str(), d, 64);
    }
    // a transcript: seed 5; three points, a challenge, two scalars, two challenges (a challenge also enters the state)
    pzp::Transcript tr((uint64_t)5);
    uint64_t pts[24], sc[8];
    for (int i = 0; i < 24; ++i) pts[i] = 0x9e3779b97f4a7c15ULL * (i + 1);
    for (int i = 0; i < 8; ++i) sc[i] = 0xbf58476d1ce4e5b9ULL * (i + 3);
    tr.common_points(pts, 3);
    tr.squeeze("a");
    tr.common_scalars(sc, 2);
    tr.squeeze("b");
    tr.squeeze("c");
    for (auto& c : tr.drawn) hex(("ch_" + c.first).c_str(), (const uint8_t*)c.second.v, 32);
    return 0;
}
