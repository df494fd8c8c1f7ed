"""ctypes binding of oracle/libpz_oracle.so (the C restatement, see pz_oracle.c header).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from paillier_halo2_amd/.

All arrays are numpy uint64: field elements (n,4) Montgomery little-endian limbs, affine
points (n,8), Jacobian points (12,), big integers (L,) little-endian u64 limbs.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpz_oracle.so")
_lib = None

U64P = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
        os.path.join(_HERE, "pz_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpz_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.ora_msm_g1.argtypes = [U64P, U64P, C.c_size_t, C.c_int, U64P]
        _lib.ora_ntt_fr.argtypes = [U64P, U64P, C.c_uint32, C.c_int]
        _lib.ora_walk_bases.argtypes = [U64P, C.c_size_t, U64P, U64P]
        _lib.ora_mul_mod_step.argtypes = [C.c_uint32, U64P, U64P, U64P, U64P, U64P]
        _lib.ora_pow_mod_trace.argtypes = [C.c_uint32, U64P, U64P, U64P, C.c_uint32, U64P,
                                           C.POINTER(C.c_size_t), U64P]
        _lib.ora_paillier_enc.argtypes = [C.c_uint32, U64P, U64P, U64P, U64P, U64P]
        _lib.ora_mont_convert.argtypes = [U64P, C.c_size_t, C.c_int, C.c_int]
        _lib.ora_g1_normalize.argtypes = [U64P, U64P]
        _lib.ora_g1_mul.argtypes = [U64P, U64P, U64P]
        _lib.ora_g1_add.argtypes = [U64P, U64P, U64P]
        _lib.ora_g1_on_curve.argtypes = [U64P]
        _lib.ora_fr_mul.argtypes = [U64P, U64P, U64P]
        _lib.ora_fq_mul.argtypes = [U64P, U64P, U64P]
        _lib.ora_fr_scale.argtypes = [U64P, C.c_size_t, U64P]
        _lib.ora_fr_distribute_powers.argtypes = [U64P, C.c_size_t, U64P]
        _lib.ora_check_gates.argtypes = [U64P, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t)]
        _lib.ora_check_gates.restype = C.c_size_t
        _lib.ora_check_range.argtypes = [U64P, C.c_size_t, C.c_uint32, C.POINTER(C.c_size_t)]
        _lib.ora_check_range.restype = C.c_size_t
        _lib.ora_fr_mul_seconds.argtypes = [C.c_size_t, C.c_int, C.c_int]
        _lib.ora_fr_mul_seconds.restype = C.c_double
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(U64P)


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


# ------------------------------------------------------------------ conversions
def int_to_limbs(x: int, n: int) -> np.ndarray:
    assert 0 <= x < (1 << (64 * n))
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def limbs_to_int(a) -> int:
    acc = 0
    for i, l in enumerate(np.asarray(a).reshape(-1).tolist()):
        acc |= int(l) << (64 * i)
    return acc


def ints_to_array(xs, n: int) -> np.ndarray:
    out = np.empty((len(xs), n), dtype=np.uint64)
    for i, x in enumerate(xs):
        out[i] = int_to_limbs(x, n)
    return out


def to_mont(arr, which: str) -> np.ndarray:
    """canonical (n,4) -> Montgomery (n,4); which in {'fq','fr'}"""
    a = _c(arr).copy().reshape(-1, 4)
    lib().ora_mont_convert(_p(a), a.shape[0], 1 if which == "fr" else 0, 1)
    return a


def from_mont(arr, which: str) -> np.ndarray:
    a = _c(arr).copy().reshape(-1, 4)
    lib().ora_mont_convert(_p(a), a.shape[0], 1 if which == "fr" else 0, 0)
    return a


def fr_ints_to_mont(xs) -> np.ndarray:
    return to_mont(ints_to_array(xs, 4), "fr")


def fr_mont_to_ints(arr):
    c = from_mont(arr, "fr")
    return [limbs_to_int(r) for r in c]


def affine_ints_to_mont(pts) -> np.ndarray:
    """[(x,y)] canonical ints -> (n,8) Montgomery; identity (0,0) stays all-zero."""
    flat = []
    for (x, y) in pts:
        flat.append(x)
        flat.append(y)
    return to_mont(ints_to_array(flat, 4), "fq").reshape(-1, 8)


def affine_mont_to_ints(arr):
    c = from_mont(np.asarray(arr).reshape(-1, 4), "fq")
    vals = [limbs_to_int(r) for r in c]
    return [(vals[2 * i], vals[2 * i + 1]) for i in range(len(vals) // 2)]


# ------------------------------------------------------------------ group / field ops
def msm_g1(scalars_mont, bases_mont, threads: int = 0) -> np.ndarray:
    s = _c(scalars_mont).reshape(-1, 4)
    b = _c(bases_mont).reshape(-1, 8)
    assert s.shape[0] == b.shape[0]
    out = np.zeros(12, dtype=np.uint64)
    rc = lib().ora_msm_g1(_p(s), _p(b), s.shape[0], threads, _p(out))
    assert rc == 0
    return out


def g1_normalize(jac) -> np.ndarray:
    j = _c(jac).reshape(12)
    out = np.zeros(8, dtype=np.uint64)
    lib().ora_g1_normalize(_p(j), _p(out))
    return out


def g1_add(a, b) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    lib().ora_g1_add(_p(_c(a).reshape(12)), _p(_c(b).reshape(12)), _p(out))
    return out


def g1_mul(base_aff_mont, k: int) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    lib().ora_g1_mul(_p(_c(base_aff_mont).reshape(8)), _p(int_to_limbs(k, 4)), _p(out))
    return out


def g1_on_curve(aff) -> bool:
    return bool(lib().ora_g1_on_curve(_p(_c(aff).reshape(8))))


def walk_bases(n: int, s: int, t: int) -> np.ndarray:
    out = np.zeros((n, 8), dtype=np.uint64)
    rc = lib().ora_walk_bases(_p(out), n, _p(int_to_limbs(s, 4)), _p(int_to_limbs(t, 4)))
    assert rc == 0
    return out


def ntt_fr(a_mont, omega_mont, log_n: int, threads: int = 0) -> np.ndarray:
    a = _c(a_mont).copy().reshape(-1, 4)
    assert a.shape[0] == 1 << log_n
    rc = lib().ora_ntt_fr(_p(a), _p(_c(omega_mont).reshape(4)), log_n, threads)
    assert rc == 0
    return a


def fr_mul_rate(n: int = 1 << 20, reps: int = 8, threads: int = 0) -> float:
    """Montgomery multiplications over Fr per second on `threads` OpenMP threads (0: the cgroup quota, ora_num_threads)"""
    L = lib()
    th = threads or L.ora_num_threads()
    L.ora_fr_mul_seconds(n, 1, th)
    dt = L.ora_fr_mul_seconds(n, reps, th)
    return n * reps / dt


def fr_scale(a_mont, scale_mont) -> np.ndarray:
    a = _c(a_mont).copy().reshape(-1, 4)
    lib().ora_fr_scale(_p(a), a.shape[0], _p(_c(scale_mont).reshape(4)))
    return a


def fr_distribute_powers(a_mont, g_mont) -> np.ndarray:
    a = _c(a_mont).copy().reshape(-1, 4)
    lib().ora_fr_distribute_powers(_p(a), a.shape[0], _p(_c(g_mont).reshape(4)))
    return a


# ------------------------------------------------------------------ big integers
def mul_mod_step(L: int, a: int, b: int, mod: int):
    q = np.zeros(L, dtype=np.uint64)
    r = np.zeros(L, dtype=np.uint64)
    rc = lib().ora_mul_mod_step(L, _p(int_to_limbs(a, L)), _p(int_to_limbs(b, L)), _p(int_to_limbs(mod, L)),
                                _p(q), _p(r))
    return rc, limbs_to_int(q), limbs_to_int(r)


def pow_mod_trace(L: int, mod: int, base: int, exp: int, exp_limbs: int):
    """returns (rc, result int, steps ndarray (n_steps, 4, L))"""
    cap = 2 * max(1, exp.bit_length()) + 2
    steps = np.zeros((cap, 4, L), dtype=np.uint64)
    ns = C.c_size_t(0)
    res = np.zeros(L, dtype=np.uint64)
    rc = lib().ora_pow_mod_trace(L, _p(int_to_limbs(mod, L)), _p(int_to_limbs(base, L)),
                                 _p(int_to_limbs(exp, exp_limbs)), exp_limbs, _p(steps), C.byref(ns), _p(res))
    return rc, limbs_to_int(res), steps[: ns.value]


def paillier_enc(Ln: int, n: int, g: int, m: int, r: int) -> int:
    out = np.zeros(2 * Ln, dtype=np.uint64)
    rc = lib().ora_paillier_enc(Ln, _p(int_to_limbs(n, Ln)), _p(int_to_limbs(g, Ln)), _p(int_to_limbs(m, Ln)),
                                _p(int_to_limbs(r, Ln)), _p(out))
    assert rc == 0, rc
    return limbs_to_int(out)


# ------------------------------------------------------------------ MockProver analogue on whole cell streams
def check_gates(cells_mont, sel) -> tuple:
    """cells: (n,4) Montgomery Fr in stream order; sel: (n,) uint8 selector (1 where a gate window [a,b,c,d] starts).
    -> (number of windows with a + b*c != d, offset of the first one or n)"""
    c = _c(cells_mont).reshape(-1, 4)
    s = np.ascontiguousarray(sel, dtype=np.uint8)
    assert s.shape[0] == c.shape[0]
    first = C.c_size_t(0)
    bad = lib().ora_check_gates(_p(c), s.ctypes.data_as(C.POINTER(C.c_uint8)), c.shape[0], C.byref(first))
    return int(bad), int(first.value)


def check_range(cells_mont, bits: int) -> tuple:
    """-> (number of cells that are not canonical integers below 2^bits, offset of the first one or n)"""
    c = _c(cells_mont).reshape(-1, 4)
    first = C.c_size_t(0)
    bad = lib().ora_check_range(_p(c), c.shape[0], bits, C.byref(first))
    return int(bad), int(first.value)
