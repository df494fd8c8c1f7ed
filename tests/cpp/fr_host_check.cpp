// prints the host field constants of paillier_halo2_amd/host/fr_host.hpp as canonical hex integers (tests/test_cpp_host_field.py compares
// them with the Python constants; no GPU, no library)
#include <cstdio>

#include "../../paillier_halo2_amd/host/fr_host.hpp"

static void show(const char* name, const pzh::Fr& a) {
    uint64_t c[4];
    pzh::to_raw(a, c);
    printf("%s %016llx%016llx%016llx%016llx\n", name, (unsigned long long)c[3], (unsigned long long)c[2], (unsigned long long)c[1], (unsigned long long)c[0]);
}

int main() {
    using namespace pzh;
    char name[32];
    for (unsigned k = 1; k <= 28; ++k) {
        snprintf(name, sizeof name, "omega%u", k);
        show(name, omega(k));
    }
    show("zeta", zeta());
    show("delta", delta());
    show("one", FR_ONE);
    const Fr seven = from_u64(7);
    show("inv7", inv(seven));
    show("neg7", neg(seven));
    show("pow7_1000003", pow_u64(seven, 1000003));
    Fr x = from_u64(0x123456789abcdefULL), acc = FR_ONE;
    for (int i = 0; i < 100; ++i) {   // a chain mixing every operation
        acc = add(mul(acc, x), neg(from_u64(i)));
        x = mul(x, x);
    }
    show("chain", acc);
    return 0;
}
