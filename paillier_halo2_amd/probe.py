"""ctypes loader of csrc/libpz_probe.so: issue-rate microbenchmarks and alternative field products (measurement only).
Not part of the product ABI (include/pz.h); bench.py uses mad_indep for the live multiplier-issue peak, profiles/probes/ the rest."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

_probe = None


def lib():
    global _probe
    if _probe is None:
        if not os.path.exists(_lib.PROBE_SO_PATH):
            raise RuntimeError(f"{_lib.PROBE_SO_PATH} is missing: build it with `make -C paillier_halo2_amd/csrc`")
        _lib.lib()   # the product library first: torch's HIP runtime is then the one already mapped (see _lib.lib)
        l = C.CDLL(_lib.PROBE_SO_PATH)
        for name, args in (("pzp_ubench_mad", [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
                           ("pzp_ubench_mad_indep", [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
                           ("pzp_ubench_fqmul_variant", [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
                           ("pzp_fq_mul29", [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
                           ("pzp_f29_ops", [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p]),
                           ("pzp_ubench_mfma", [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
                           ("pzp_mulc_mfma", [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                              C.POINTER(C.c_double)])):
            fn = getattr(l, name)
            fn.restype = C.c_int
            fn.argtypes = args
        _probe = l
    return _probe


def _ms(fn, *args) -> float:
    ms = C.c_double()
    if fn(*args, C.byref(ms)) != 0:
        raise RuntimeError("probe launch failed")
    return ms.value


def ubench_mad(device: int, blocks: int, iters: int) -> float:
    return _ms(lib().pzp_ubench_mad, device, blocks, iters)


def ubench_mad_indep(device: int, blocks: int, iters: int) -> float:
    return _ms(lib().pzp_ubench_mad_indep, device, blocks, iters)


def ubench_fqmul_variant(device: int, variant: int, blocks: int, iters: int) -> float:
    return _ms(lib().pzp_ubench_fqmul_variant, device, variant, blocks, iters)


def fq_mul29(device: int, a, b) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(4)
    out = np.zeros(4, dtype=np.uint64)
    if lib().pzp_fq_mul29(device, a.ctypes.data, b.ctypes.data, out.ctypes.data) != 0:
        raise RuntimeError("probe launch failed")
    return out


F29_OPS = {"mul": (0, 2), "sqr": (1, 1), "mul2": (2, 4), "dot4": (3, 8), "unpack_shl5": (4, 1), "canon4": (5, 1), "store_product": (6, 1), "canon_q": (7, 1), "mulc": (8, 3)}


def f29_ops(device: int, field: str, op: str, limbs) -> np.ndarray:
    """element-wise check entry of the 29-bit field's building blocks (csrc/fp29.cuh) on RAW limb operands:
    limbs [count][k][9] uint32 (loose limbs allowed) -> [count][9]; field 'fq' | 'fr'; op a key of F29_OPS"""
    code, k = F29_OPS[op]
    a = np.ascontiguousarray(limbs, dtype=np.uint32)
    assert a.ndim == 3 and a.shape[1] == k and a.shape[2] == 9, a.shape
    out = np.zeros((a.shape[0], 9), dtype=np.uint32)
    if lib().pzp_f29_ops(device, 0 if field == "fq" else 1, code, a.ctypes.data, k, a.shape[0], out.ctypes.data) != 0:
        raise RuntimeError("probe launch failed")
    return out


def ubench_mfma(device: int, which: int, blocks: int, iters: int) -> float:
    """csrc/probe/pz_probe_mfma.hip: which = 1 the matrix side alone (8 x v_mfma_i32_16x16x64_i8 per iteration per wave = 16 field products'
    worth), 2 the VALU side alone (digit split + two carry propagations + repack per lane and iteration).  -> ms"""
    return _ms(lib().pzp_ubench_mfma, device, which, blocks, iters)


def mulc_mfma(device: int, a_limbs, w: int, p: int, iters: int = 1):
    """a * w mod p for a batch through the I8-MFMA formulation of the constant product (the whole pipeline, one wave per 16 elements):
    a_limbs [count][9] strict 29-bit limbs of values below 2^261, w < p.  -> ([count][9] canonical limbs, ms of one launch)"""
    a = np.ascontiguousarray(a_limbs, dtype=np.uint32)
    assert a.ndim == 2 and a.shape[1] == 9
    dig = lambda v: np.array([(v >> (7 * i)) & 127 for i in range(38)], dtype=np.uint8)
    wq = (w << 266) // p
    assert wq < (1 << 266)
    p29 = np.array([(p >> (29 * i)) & ((1 << 29) - 1) for i in range(9)], dtype=np.uint32)
    wd, wqd, pd = dig(w), dig(wq), dig(p)
    out = np.zeros_like(a)
    ms = C.c_double()
    if lib().pzp_mulc_mfma(device, a.ctypes.data, a.shape[0], wd.ctypes.data, wqd.ctypes.data, pd.ctypes.data, p29.ctypes.data, out.ctypes.data, iters,
                           C.byref(ms)) != 0:
        raise RuntimeError("probe launch failed")
    return out, ms.value
