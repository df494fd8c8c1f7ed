// pz_witness.hip -- K4: expansion of mul_mod steps into advice cells (placeholder until the
// expansion kernels land; the entry points exist so the ABI is complete and fail loudly).
#include "pz_internal.h"

extern "C" int pz_witness_cells_per_step(uint32_t, uint32_t, uint32_t, size_t*, size_t*) { return PZ_ERR_UNSUPPORTED; }
extern "C" int pz_witness_expand_dev(pz_ctx*, uint32_t, uint32_t, uint32_t, const uint64_t*, size_t, const uint64_t*,
                                     uint64_t*, uint64_t*) {
    return PZ_ERR_UNSUPPORTED;
}
