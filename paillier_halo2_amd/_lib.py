"""ctypes loader for paillier_halo2_amd/csrc/libpz_hip.so (the C ABI of include/pz.h).

Plumbing only: no arithmetic lives here.  The library is hand-written HIP for gfx950; if it is
missing or cannot be loaded the import fails loudly -- there is no CPU fallback and nothing in
this package imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libpz_hip.so")
PROBE_SO_PATH = os.path.join(CSRC, "libpz_probe.so")   # measurement probes, not part of the product ABI (probe.py)

U64P = C.POINTER(C.c_uint64)
U32P = C.POINTER(C.c_uint32)
VP = C.c_void_p

# name -> (restype, argtypes): every entry point include/pz.h declares
SIGNATURES = {
    "pz_init": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(VP)]),
    "pz_free": (C.c_int, [VP]),
    "pz_strerror": (C.c_char_p, [C.c_int]),
    "pz_last_hip_error": (C.c_char_p, [VP]),
    "pz_set_stream": (C.c_int, [VP, VP]),
    "pz_sync": (C.c_int, [VP]),
    "pz_abi_version": (C.c_int, []),
    "pz_dev_alloc": (C.c_int, [VP, C.c_size_t, C.POINTER(VP)]),
    "pz_dev_cache_limit": (C.c_int, [VP, C.c_size_t]),
    "pz_dev_arena": (C.c_int, [VP, C.c_size_t]),
    "pz_dev_arena_info": (C.c_int, [VP, C.POINTER(C.c_uint64)]),
    "pz_dev_mem_info": (C.c_int, [VP, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "pz_dev_free": (C.c_int, [VP, VP]),
    "pz_upload": (C.c_int, [VP, VP, VP, C.c_size_t]),
    "pz_download": (C.c_int, [VP, VP, VP, C.c_size_t]),
    "pz_dev_memset": (C.c_int, [VP, VP, C.c_int, C.c_size_t]),
    "pz_dev_copy": (C.c_int, [VP, VP, VP, C.c_size_t]),
    "pz_dev_copy_2d": (C.c_int, [VP, VP, C.c_size_t, VP, C.c_size_t, C.c_size_t, C.c_size_t]),
    "pz_ctx_wait": (C.c_int, [VP, VP]),
    "pz_srs_load_g1": (C.c_int, [VP, C.c_uint32, VP, C.c_int, C.POINTER(VP)]),
    "pz_bases_load_g1": (C.c_int, [VP, VP, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(VP)]),
    "pz_bases_free": (C.c_int, [VP, VP]),
    "pz_bases_info": (C.c_int, [VP, C.POINTER(C.c_size_t), U32P, U32P]),
    "pz_msm_g1": (C.c_int, [VP, VP, VP, C.c_size_t, VP]),
    "pz_msm_g1_batch": (C.c_int, [VP, VP, C.POINTER(VP), C.c_size_t, C.c_size_t, VP]),
    "pz_msm_g1_dev": (C.c_int, [VP, VP, VP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, VP]),
    "pz_g1_sum": (C.c_int, [VP, VP, C.c_size_t, VP]),
    "pz_g1_sum_dev": (C.c_int, [VP, VP, C.c_size_t, VP]),
    "pz_msm_g1_multi": (C.c_int, [VP, VP, VP, VP, C.c_size_t, C.c_int, VP]),
    "pz_g1_normalize": (C.c_int, [VP, VP, C.c_size_t, VP]),
    "pz_g1_fixed_base_mul": (C.c_int, [VP, VP, C.c_size_t, VP]),
    "pz_g1_fixed_base_mul_dev": (C.c_int, [VP, VP, C.c_size_t, VP]),
    "pz_ntt_fr": (C.c_int, [VP, VP, VP, C.c_uint32]),
    "pz_ntt_fr_batch": (C.c_int, [VP, C.POINTER(VP), C.c_size_t, VP, C.c_uint32]),
    "pz_ntt_fr_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_size_t, VP, C.c_uint32, VP, VP]),
    "pz_ntt_fr_to_dev": (C.c_int, [VP, VP, C.c_size_t, VP, C.c_size_t, C.c_size_t, VP, C.c_uint32, VP, VP]),
    "pz_ntt_fr_extend_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_size_t, VP, C.c_size_t, C.c_uint32, C.c_uint32, VP, VP, VP]),
    "pz_ntt_fr_coeff_extend_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_size_t, VP, C.c_size_t, C.c_uint32, C.c_uint32, VP, VP, VP,
                                             VP]),
    "pz_fr_convert_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_int]),
    "pz_fr_from_mask_dev": (C.c_int, [VP, VP, C.c_size_t, VP]),
    "pz_mul_mod": (C.c_int, [VP, C.c_uint32, VP, VP, VP, VP, VP]),
    "pz_paillier_trace": (C.c_int, [VP, C.c_uint32, VP, VP, VP, C.c_uint32, VP, C.POINTER(C.c_size_t), VP]),
    "pz_paillier_encrypt": (C.c_int, [VP, C.c_uint32, C.c_size_t, VP, VP, VP, VP, VP, C.c_size_t, VP, VP, VP]),
    "pz_paillier_encrypt_dev": (C.c_int, [VP, C.c_uint32, C.c_size_t, VP, VP, VP, VP, VP, C.c_size_t, VP, VP, VP]),
    "pz_paillier_encrypt_uniform": (C.c_int, [VP, C.c_uint32, C.c_size_t, C.c_uint32, VP, VP, VP, VP, VP, C.c_size_t, VP, VP, VP]),
    "pz_paillier_encrypt_uniform_dev": (C.c_int, [VP, C.c_uint32, C.c_size_t, C.c_uint32, VP, VP, VP, VP, VP, C.c_size_t, VP, VP, VP]),
    "pz_witness_cells_per_step": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_size_t)]),
    "pz_witness_expand": (C.c_int, [VP, C.c_uint32, C.c_uint32, C.c_uint32, VP, C.c_size_t, VP, VP, VP]),
    "pz_witness_expand_dev": (C.c_int, [VP, C.c_uint32, C.c_uint32, C.c_uint32, VP, C.c_size_t, VP, VP, VP]),
    "pz_refresh_aux": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, VP, C.c_uint32, U32P]),
    "pz_op_cells": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "pz_circuit_cells": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_size_t)]),
    "pz_circuit_expand_dev": (C.c_int, [VP, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, VP, VP, C.c_size_t, C.c_size_t, VP, VP, VP,
                                        C.c_size_t, C.c_size_t]),
    "pz_circuit_break_points": (C.c_int, [VP, C.c_size_t, C.c_size_t, VP, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pz_circuit_expand_cols_dev": (C.c_int, [VP, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, VP, VP, C.c_size_t, C.c_size_t, VP, VP, VP,
                                             VP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    "pz_srs_setup_g1_dev": (C.c_int, [VP, C.c_uint32, VP, VP, VP, VP]),
    "pz_srs_lagrange_from_monomial_dev": (C.c_int, [VP, C.c_uint32, VP, VP, VP, VP]),
    "pz_permutation_sigma_dev": (C.c_int, [VP, VP, VP, C.c_size_t, C.c_uint32, VP, VP, VP, C.c_size_t]),
    "pz_keygen_columns_dev": (C.c_int, [VP, VP, VP, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, VP, VP, VP, VP, VP, VP, C.c_size_t]),
    "pz_poly_eval_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_size_t, C.c_size_t, VP, VP]),
    "pz_poly_eval_multi_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_size_t, C.c_size_t, VP, C.c_uint32, VP]),
    "pz_g1_check_dev": (C.c_int, [VP, VP, C.c_size_t, C.POINTER(C.c_uint64)]),
    "pz_fr_batch_invert_dev": (C.c_int, [VP, VP, C.c_size_t]),
    "pz_fr_prefix_product_dev": (C.c_int, [VP, VP, C.c_size_t, VP, VP]),
    "pz_permutation_product_dev": (C.c_int, [VP, VP, C.c_size_t, VP, C.c_size_t, C.c_size_t, C.c_uint32, VP, VP, VP, VP,
                                             VP, VP, VP]),
    "pz_lookup_permute_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_size_t, VP, C.c_size_t, C.c_uint32, VP, VP, C.c_size_t]),
    "pz_lookup_product_dev": (C.c_int, [VP, VP, C.c_size_t, VP, VP, C.c_size_t, VP, C.c_size_t, C.c_size_t, C.c_size_t, VP, VP,
                                        VP, VP, C.c_size_t]),
    "pz_permutation_product_sets_dev": (C.c_int, [VP, VP, C.c_size_t, VP, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32,
                                                  C.c_size_t, VP, VP, VP, VP, VP, C.c_size_t]),
    "pz_quotient_gate_dev": (C.c_int, [VP, VP, C.c_size_t, VP, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, VP, VP]),
    "pz_quotient_permutation_dev": (C.c_int, [VP, VP, C.c_size_t, VP, C.c_size_t, VP, C.c_size_t, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, VP, VP, VP, VP, VP, VP, VP, VP,
                                              VP, VP]),
    "pz_quotient_permutation_part_dev": (C.c_int, [VP, VP, C.c_size_t, VP, C.c_size_t, VP, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                   C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, VP, VP, VP, VP, VP,
                                                   VP, VP, VP, VP, VP]),
    "pz_quotient_lookup_dev": (C.c_int, [VP, VP, C.c_size_t, VP, VP, C.c_size_t, VP, C.c_size_t, VP, C.c_size_t, C.c_uint32,
                                         C.c_uint32, C.c_uint32, VP, VP, VP, VP, VP, VP, VP]),
    "pz_quotient_finish_dev": (C.c_int, [VP, VP, C.c_uint32, C.c_uint32, VP, VP]),
    "pz_fr_distribute_powers_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_size_t, C.c_size_t, VP, VP]),
    "pz_fr_lincomb_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_size_t, C.c_size_t, VP, VP, C.c_int]),
    "pz_poly_div_linear_dev": (C.c_int, [VP, VP, C.c_size_t, C.c_size_t, C.c_size_t, VP, VP, C.c_size_t]),
    "pz_shplonk_begin_dev": (C.c_int, [VP, C.c_size_t, C.c_uint32, U32P, C.POINTER(VP), U32P, U32P, C.c_uint32, VP, VP, VP, VP, VP,
                                       C.POINTER(VP)]),
    "pz_shplonk_finish_dev": (C.c_int, [VP, VP, VP, VP, VP]),
    "pz_shplonk_free": (C.c_int, [VP, VP]),
    # patch point D as entry points: keygen + create_proof, one call per transcript round
    "pz_circuit_structure_dev": (C.c_int, [VP, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, VP, VP, C.c_size_t, C.c_uint32, C.POINTER(VP)]),
    "pz_structure_info": (C.c_int, [VP] + [C.POINTER(C.c_size_t)] * 9),
    "pz_structure_arrays": (C.c_int, [VP] + [C.POINTER(VP)] * 6),
    "pz_structure_free": (C.c_int, [VP]),
    "pz_pk_create": (C.c_int, [VP, VP, VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.c_size_t, VP, VP, C.c_size_t, VP, VP,
                               C.c_size_t, C.c_size_t, C.POINTER(VP)]),
    "pz_pk_create_dev": (C.c_int, [VP, VP, VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.c_size_t, VP, VP, C.c_size_t, VP, VP,
                                   C.c_size_t, C.c_size_t, C.POINTER(VP)]),
    "pz_pk_info": (C.c_int, [VP, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                             C.POINTER(C.c_size_t)]),
    "pz_pk_commitments": (C.c_int, [VP, VP, VP]),
    "pz_pk_free": (C.c_int, [VP]),
    "pz_proof_begin": (C.c_int, [VP, VP, C.c_uint64, VP, C.c_size_t, C.POINTER(VP), VP]),
    "pz_proof_lookups": (C.c_int, [VP, VP, VP, VP]),
    "pz_proof_products": (C.c_int, [VP, VP, VP, VP, VP, VP]),
    "pz_proof_quotient": (C.c_int, [VP, VP, VP]),
    "pz_proof_evaluate": (C.c_int, [VP, VP, VP]),
    "pz_proof_open_begin": (C.c_int, [VP, VP, VP, VP]),
    "pz_proof_open_finish": (C.c_int, [VP, VP, VP, C.POINTER(C.c_int)]),
    "pz_proof_free": (C.c_int, [VP]),
    "pz_timing_enable": (C.c_int, [VP, C.c_int]),
    "pz_timing_reset": (C.c_int, [VP]),
    "pz_timing_get": (C.c_int, [VP, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    if not os.path.exists(SO_PATH):
        raise RuntimeError("libpz_hip.so was not produced by the build")
    return SO_PATH


def lib():
    """Load the HIP library; raises if it is absent (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        # torch bundles its own HIP runtime (same SONAME, libamdhip64.so.7).  Two HIP/HSA runtimes in one
        # process cannot both open the GPU, so when torch is installed make sure ITS runtime is the one
        # already mapped before libpz_hip.so's NEEDED entry is resolved.  Standalone (Rust/C++) users of
        # the C ABI simply get the system ROCm runtime.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the C ABI itself
            pass
        l = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError here == ABI symbol missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class PzError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        msg = lib().pz_strerror(status).decode()
        super().__init__(f"{where}: pz_status {status} ({msg}){(' -- ' + detail) if detail else ''}")


# status codes mirrored from include/pz.h
PZ_OK = 0
PZ_ERR_INVALID = -1
PZ_ERR_HIP = -2
PZ_ERR_NO_DEVICE = -3
PZ_ERR_OOM = -4
PZ_ERR_ZERO_MODULUS = -5
PZ_ERR_RANGE = -6
PZ_ERR_UNSUPPORTED = -7
PZ_ERR_CAPACITY = -8
PZ_ERR_MESSAGE_RANGE = -9
PZ_ERR_ASYNC = -10
PZ_ERR_INTERNAL = -11
