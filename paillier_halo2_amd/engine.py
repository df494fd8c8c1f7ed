"""Thin host-side handle over the C ABI (include/pz.h): owns a pz_ctx, marshals numpy arrays
(host-pointer entry points) and torch CUDA tensors (device-pointer entry points).

All arithmetic happens in libpz_hip.so on the MI355X.  Array conventions are the ABI's: uint64,
field elements (…,4) Montgomery limbs, affine points (…,8), Jacobian points (…,12), big
integers little-endian u64 limbs.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import PzError, VP


def _np(a, shape_last: Optional[int] = None) -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=np.uint64)
    if shape_last is not None and (arr.ndim == 0 or arr.shape[-1] != shape_last):
        arr = arr.reshape(-1, shape_last)
    return arr


def _ptr(a: np.ndarray) -> VP:
    return VP(a.ctypes.data)


class Bases:
    """A device-resident window-shifted base table (pz_bases)."""

    def __init__(self, engine: "Engine", handle: VP):
        self.engine = engine
        self.handle = handle
        n = C.c_size_t()
        c = C.c_uint32()
        w = C.c_uint32()
        engine._chk(_lib.lib().pz_bases_info(handle, C.byref(n), C.byref(c), C.byref(w)), "pz_bases_info")
        self.n_points = n.value
        self.window_bits = c.value
        self.n_windows = w.value

    def free(self):
        if self.handle is not None:
            _lib.lib().pz_bases_free(self.engine.ctx, self.handle)
            self.handle = None


class Engine:
    """One pz_ctx == one GPU (one process per GPU; ranks are joined by RCCL above this layer)."""

    def __init__(self, device: int = 0):
        self.L = _lib.lib()
        self.ctx = VP()
        dev = (C.c_int * 1)(device)
        rc = self.L.pz_init(1, dev, C.byref(self.ctx))
        if rc != 0:
            raise PzError(rc, "pz_init")
        self.device = device

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc: int, where: str):
        if rc != 0:
            detail = self.L.pz_last_hip_error(self.ctx).decode() if self.ctx else ""
            raise PzError(rc, where, detail)

    def close(self):
        if self.ctx:
            self.L.pz_free(self.ctx)
            self.ctx = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle: int = 0):
        """stream_handle: a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); 0 = own."""
        self._chk(self.L.pz_set_stream(self.ctx, VP(stream_handle)), "pz_set_stream")

    def bind_torch_stream(self, stream=None):
        """Run the library on a torch stream (a fresh one by default) and make that stream torch's current one, so
        torch fills / copies / event waits and the library's kernels are ordered.  torch's DEFAULT stream is the NULL
        stream, whose handle 0 means "the library's own non-blocking stream" to pz_set_stream -- binding to it
        orders nothing."""
        import torch

        if stream is None:
            stream = torch.cuda.Stream(device=self.device)
        assert stream.cuda_stream != 0
        torch.cuda.set_stream(stream)
        self.set_stream(stream.cuda_stream)
        self._torch_stream = stream
        return stream

    def sync(self):
        self._chk(self.L.pz_sync(self.ctx), "pz_sync")

    # ------------------------------------------------------------------ K1: MSM
    def load_bases(self, bases_affine, window_bits: int = 0) -> Bases:
        b = _np(bases_affine, 8)
        h = VP()
        self._chk(self.L.pz_bases_load_g1(self.ctx, _ptr(b), b.shape[0], 0, window_bits, C.byref(h)),
                  "pz_bases_load_g1")
        return Bases(self, h)

    def load_bases_dev(self, d_ptr: int, n_points: int, window_bits: int = 0) -> Bases:
        h = VP()
        self._chk(self.L.pz_bases_load_g1(self.ctx, VP(d_ptr), n_points, 1, window_bits, C.byref(h)),
                  "pz_bases_load_g1(dev)")
        return Bases(self, h)

    def srs_load_g1(self, k: int, bases_affine, lagrange: bool = False) -> Bases:
        b = _np(bases_affine, 8)
        assert b.shape[0] == 1 << k
        h = VP()
        self._chk(self.L.pz_srs_load_g1(self.ctx, k, _ptr(b), int(lagrange), C.byref(h)), "pz_srs_load_g1")
        return Bases(self, h)

    def msm(self, bases: Bases, scalars) -> np.ndarray:
        s = _np(scalars, 4)
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.pz_msm_g1(self.ctx, bases.handle, _ptr(s) if s.size else VP(), s.shape[0] if s.size else 0,
                                   _ptr(out)), "pz_msm_g1")
        return out

    def msm_batch(self, bases: Bases, cols: Sequence[np.ndarray]) -> np.ndarray:
        cols = [_np(c, 4) for c in cols]
        n = cols[0].shape[0] if cols else 0
        assert all(c.shape[0] == n for c in cols)
        arr = (VP * len(cols))(*[_ptr(c) for c in cols])
        out = np.zeros((len(cols), 12), dtype=np.uint64)
        self._chk(self.L.pz_msm_g1_batch(self.ctx, bases.handle, arr, len(cols), n, _ptr(out)), "pz_msm_g1_batch")
        return out

    def msm_dev(self, bases: Bases, d_scalars: int, n_cols: int, n: int, col_stride_u64: int, d_out: int,
                win_lo: int = 0, win_hi: Optional[int] = None):
        if win_hi is None:
            win_hi = bases.n_windows
        self._chk(self.L.pz_msm_g1_dev(self.ctx, bases.handle, VP(d_scalars), n_cols, n, col_stride_u64, win_lo,
                                       win_hi, VP(d_out)), "pz_msm_g1_dev")

    @staticmethod
    def msm_multi(engines: Sequence["Engine"], bases: Sequence[Bases], d_scalars: Sequence[int], n_per_ctx: Sequence[int],
                  split_points: bool) -> np.ndarray:
        """pz_msm_g1_multi: ONE MSM over several contexts (one per GPU; several on one device for rehearsal).  d_scalars: device
        pointers, one per context.  Returns the Jacobian point (12,)."""
        k = len(engines)
        assert len(bases) == len(d_scalars) == len(n_per_ctx) == k and k > 0
        ctxs = (VP * k)(*[e.ctx for e in engines])
        bs = (VP * k)(*[(b.handle if b is not None else VP()) for b in bases])
        sc = (VP * k)(*[VP(p) for p in d_scalars])
        ns = (C.c_size_t * k)(*n_per_ctx)
        out = np.zeros(12, dtype=np.uint64)
        engines[0]._chk(engines[0].L.pz_msm_g1_multi(ctxs, bs, sc, ns, k, int(split_points), _ptr(out)), "pz_msm_g1_multi")
        return out

    def g1_sum(self, jac) -> np.ndarray:
        j = _np(jac, 12)
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.pz_g1_sum(self.ctx, _ptr(j), j.shape[0], _ptr(out)), "pz_g1_sum")
        return out

    def g1_sum_dev(self, d_jac: int, n: int, d_out: int):
        self._chk(self.L.pz_g1_sum_dev(self.ctx, VP(d_jac), n, VP(d_out)), "pz_g1_sum_dev")

    def g1_normalize(self, jac) -> np.ndarray:
        j = _np(jac, 12)
        out = np.zeros((j.shape[0], 8), dtype=np.uint64)
        self._chk(self.L.pz_g1_normalize(self.ctx, _ptr(j), j.shape[0], _ptr(out)), "pz_g1_normalize")
        return out

    def g1_fixed_base_mul(self, scalars) -> np.ndarray:
        s = _np(scalars, 4)
        out = np.zeros((s.shape[0], 8), dtype=np.uint64)
        self._chk(self.L.pz_g1_fixed_base_mul(self.ctx, _ptr(s), s.shape[0], _ptr(out)), "pz_g1_fixed_base_mul")
        return out

    def g1_fixed_base_mul_dev(self, d_scalars: int, n: int, d_out: int):
        self._chk(self.L.pz_g1_fixed_base_mul_dev(self.ctx, VP(d_scalars), n, VP(d_out)), "pz_g1_fixed_base_mul_dev")

    # ------------------------------------------------------------------ K2: NTT
    def ntt(self, a, omega, log_n: int) -> np.ndarray:
        x = _np(a, 4).copy()
        assert x.shape[0] == 1 << log_n
        w = _np(omega).reshape(4)
        self._chk(self.L.pz_ntt_fr(self.ctx, _ptr(x), _ptr(w), log_n), "pz_ntt_fr")
        return x

    def ntt_batch(self, cols: Sequence[np.ndarray], omega, log_n: int):
        cols = [_np(c, 4).copy() for c in cols]
        arr = (VP * len(cols))(*[_ptr(c) for c in cols])
        w = _np(omega).reshape(4)
        self._chk(self.L.pz_ntt_fr_batch(self.ctx, arr, len(cols), _ptr(w), log_n), "pz_ntt_fr_batch")
        return cols

    def ntt_batch_inplace(self, cols: Sequence[np.ndarray], omega, log_n: int):
        """host-pointer pz_ntt_fr_batch IN PLACE on the caller's arrays (uint64, contiguous): no staging copies on the host"""
        arr = (VP * len(cols))(*[_ptr(c) for c in cols])
        w = _np(omega).reshape(4)
        self._chk(self.L.pz_ntt_fr_batch(self.ctx, arr, len(cols), _ptr(w), log_n), "pz_ntt_fr_batch")

    def ntt_to_dev(self, d_in: int, in_stride_u64: int, d_out: int, out_stride_u64: int, n_cols: int, omega, log_n: int,
                   pre_coset_g=None, post_scale=None):
        """out of place: reads d_in, writes d_out (pz_ntt_fr_to_dev)"""
        w = _np(omega).reshape(4)
        g = _np(pre_coset_g).reshape(4) if pre_coset_g is not None else None
        s = _np(post_scale).reshape(4) if post_scale is not None else None
        self._chk(self.L.pz_ntt_fr_to_dev(self.ctx, VP(d_in), in_stride_u64, VP(d_out), out_stride_u64, n_cols, _ptr(w), log_n,
                                          _ptr(g) if g is not None else VP(), _ptr(s) if s is not None else VP()),
                  "pz_ntt_fr_to_dev")

    def ntt_dev(self, d_a: int, n_cols: int, col_stride_u64: int, omega, log_n: int, pre_coset_g=None,
                post_scale=None):
        w = _np(omega).reshape(4)
        g = _np(pre_coset_g).reshape(4) if pre_coset_g is not None else None
        s = _np(post_scale).reshape(4) if post_scale is not None else None
        self._chk(self.L.pz_ntt_fr_dev(self.ctx, VP(d_a), n_cols, col_stride_u64, _ptr(w), log_n,
                                       _ptr(g) if g is not None else VP(), _ptr(s) if s is not None else VP()),
                  "pz_ntt_fr_dev")

    def ntt_extend_dev(self, d_coeff: int, n_cols: int, in_stride_u64: int, d_ext: int, out_stride_u64: int, log_n: int,
                       log_e: int, omega_n, coset_gens, scale=None):
        w = _np(omega_n).reshape(4)
        g = _np(coset_gens).reshape(-1)
        assert g.size == 4 << log_e
        sc = _np(scale).reshape(4) if scale is not None else None
        self._chk(self.L.pz_ntt_fr_extend_dev(self.ctx, VP(d_coeff), n_cols, in_stride_u64, VP(d_ext), out_stride_u64,
                                              log_n, log_e, _ptr(w), _ptr(g), _ptr(sc) if sc is not None else VP()),
                  "pz_ntt_fr_extend_dev")

    def ntt_coeff_extend_dev(self, d_values: int, n_cols: int, col_stride_u64: int, d_ext: int, out_stride_u64: int, log_n: int,
                             log_e: int, omega_n, omega_n_inv, n_inv, coset_gens):
        g = _np(coset_gens).reshape(-1)
        assert g.size == 4 << log_e
        self._chk(self.L.pz_ntt_fr_coeff_extend_dev(self.ctx, VP(d_values), n_cols, col_stride_u64, VP(d_ext), out_stride_u64, log_n,
                                                    log_e, self._fr1(omega_n), self._fr1(omega_n_inv), self._fr1(n_inv), _ptr(g)),
                  "pz_ntt_fr_coeff_extend_dev")

    def fr_convert_dev(self, d_a: int, n: int, to_mont: bool = True):
        self._chk(self.L.pz_fr_convert_dev(self.ctx, VP(d_a), n, int(to_mont)), "pz_fr_convert_dev")

    def paillier_encrypt_dev(self, limbs_n: int, n, g, m, r, d_steps: int, steps_cap: int):
        """steps stay on the device (d_steps: batch x steps_cap x 4 x 2*limbs_n u64). Returns (c, ng, nr)."""
        n, g, m, r = (_np(x).reshape(-1, limbs_n) for x in (n, g, m, r))
        batch = n.shape[0]
        ng = np.zeros(batch, dtype=np.uint32)
        nr = np.zeros(batch, dtype=np.uint32)
        c = np.zeros((batch, 2 * limbs_n), dtype=np.uint64)
        self._chk(self.L.pz_paillier_encrypt_dev(self.ctx, limbs_n, batch, _ptr(n), _ptr(g), _ptr(m), _ptr(r),
                                                 VP(d_steps), steps_cap, VP(ng.ctypes.data), VP(nr.ctypes.data),
                                                 _ptr(c)), "pz_paillier_encrypt_dev")
        return c, ng, nr

    # ------------------------------------------------------------------ K3: big integers
    def mul_mod(self, limbs: int, a, b, modulus) -> Tuple[np.ndarray, np.ndarray]:
        a, b, m = (_np(x).reshape(limbs) for x in (a, b, modulus))
        q = np.zeros(limbs, dtype=np.uint64)
        r = np.zeros(limbs, dtype=np.uint64)
        self._chk(self.L.pz_mul_mod(self.ctx, limbs, _ptr(a), _ptr(b), _ptr(m), _ptr(q), _ptr(r)), "pz_mul_mod")
        return q, r

    def paillier_trace(self, limbs_n2: int, n2, base, exp, exp_limbs: int, want_steps: bool = True):
        n2, base = (_np(x).reshape(limbs_n2) for x in (n2, base))
        exp = _np(exp).reshape(exp_limbs)
        e_int = 0
        for i, l in enumerate(exp.tolist()):
            e_int |= int(l) << (64 * i)
        cap = e_int.bit_length() + bin(e_int).count("1")
        steps = np.zeros((max(cap, 1), 4, limbs_n2), dtype=np.uint64) if want_steps else None
        ns = C.c_size_t(cap)
        res = np.zeros(limbs_n2, dtype=np.uint64)
        self._chk(self.L.pz_paillier_trace(self.ctx, limbs_n2, _ptr(n2), _ptr(base), _ptr(exp), exp_limbs,
                                           _ptr(steps) if want_steps else VP(), C.byref(ns), _ptr(res)),
                  "pz_paillier_trace")
        return res, (steps[: ns.value] if want_steps else None), ns.value

    def paillier_encrypt(self, limbs_n: int, n, g, m, r, want_steps: bool = True):
        """Batch form: n, g, m, r are (batch, limbs_n).  Returns (c (batch, 2*limbs_n), steps or None,
        n_steps_g, n_steps_r)."""
        n, g, m, r = (_np(x).reshape(-1, limbs_n) for x in (n, g, m, r))
        batch = n.shape[0]
        L = 2 * limbs_n
        cap = 0
        steps = None
        if want_steps:
            for i in range(batch):
                bits = 0
                for arr in (m[i], n[i]):
                    e_int = 0
                    for k, l in enumerate(arr.tolist()):
                        e_int |= int(l) << (64 * k)
                    bits += e_int.bit_length() + bin(e_int).count("1")
                cap = max(cap, bits + 1)
            steps = np.zeros((batch, cap, 4, L), dtype=np.uint64)
        ng = np.zeros(batch, dtype=np.uint32)
        nr = np.zeros(batch, dtype=np.uint32)
        c = np.zeros((batch, L), dtype=np.uint64)
        self._chk(self.L.pz_paillier_encrypt(self.ctx, limbs_n, batch, _ptr(n), _ptr(g), _ptr(m), _ptr(r),
                                             _ptr(steps) if want_steps else VP(), cap, VP(ng.ctypes.data),
                                             VP(nr.ctypes.data), _ptr(c)), "pz_paillier_encrypt")
        return c, steps, ng, nr

    def paillier_encrypt_uniform(self, limbs_n: int, m_bits: int, n, g, m, r, want_steps: bool = True):
        """uniform-shape circuit (pz.h): batch form, 2*m_bits steps for g^m whatever the messages are.
        Returns (c, steps or None, n_steps_g, n_steps_r)."""
        n, g, m, r = (_np(x).reshape(-1, limbs_n) for x in (n, g, m, r))
        batch = n.shape[0]
        L = 2 * limbs_n
        cap, steps = 0, None
        if want_steps:
            for i in range(batch):
                e_int = 0
                for k, l in enumerate(n[i].tolist()):
                    e_int |= int(l) << (64 * k)
                cap = max(cap, 2 * m_bits + e_int.bit_length() + bin(e_int).count("1") + 1)
            steps = np.zeros((batch, cap, 4, L), dtype=np.uint64)
        ng = np.zeros(batch, dtype=np.uint32)
        nr = np.zeros(batch, dtype=np.uint32)
        c = np.zeros((batch, L), dtype=np.uint64)
        self._chk(self.L.pz_paillier_encrypt_uniform(self.ctx, limbs_n, batch, m_bits, _ptr(n), _ptr(g), _ptr(m), _ptr(r),
                                                     _ptr(steps) if want_steps else VP(), cap, VP(ng.ctypes.data), VP(nr.ctypes.data),
                                                     _ptr(c)), "pz_paillier_encrypt_uniform")
        return c, steps, ng, nr

    def paillier_encrypt_uniform_dev(self, limbs_n: int, m_bits: int, n, g, m, r, d_steps: int, steps_cap: int):
        """uniform-shape trace, steps stay on the device (d_steps: batch x steps_cap x 4 x 2*limbs_n u64). Returns (c, ng, nr)."""
        n, g, m, r = (_np(x).reshape(-1, limbs_n) for x in (n, g, m, r))
        batch = n.shape[0]
        ng = np.zeros(batch, dtype=np.uint32)
        nr = np.zeros(batch, dtype=np.uint32)
        c = np.zeros((batch, 2 * limbs_n), dtype=np.uint64)
        self._chk(self.L.pz_paillier_encrypt_uniform_dev(self.ctx, limbs_n, batch, m_bits, _ptr(n), _ptr(g), _ptr(m), _ptr(r),
                                                         VP(d_steps), steps_cap, VP(ng.ctypes.data), VP(nr.ctypes.data), _ptr(c)),
                  "pz_paillier_encrypt_uniform_dev")
        return c, ng, nr

    # ------------------------------------------------------------------ K4: witness expansion
    def witness_cells_per_step(self, limbs: int, limb_bits: int, lookup_bits: int) -> Tuple[int, int]:
        a = C.c_size_t()
        l = C.c_size_t()
        self._chk(self.L.pz_witness_cells_per_step(limbs, limb_bits, lookup_bits, C.byref(a), C.byref(l)),
                  "pz_witness_cells_per_step")
        return a.value, l.value

    def witness_expand(self, limbs: int, limb_bits: int, lookup_bits: int, steps, modulus, want_lookup: bool = True):
        """host form: steps (n_steps, 4, words) u64, modulus (words,) -> (advice (n_steps, cells, 4), lookup or None)"""
        st = np.ascontiguousarray(steps, dtype=np.uint64)
        n_steps = st.shape[0]
        md = np.ascontiguousarray(modulus, dtype=np.uint64)
        adv_n, lk_n = self.witness_cells_per_step(limbs, limb_bits, lookup_bits)
        adv = np.zeros((n_steps, adv_n, 4), dtype=np.uint64)
        lk = np.zeros((n_steps, lk_n, 4), dtype=np.uint64) if want_lookup else None
        self._chk(self.L.pz_witness_expand(self.ctx, limbs, limb_bits, lookup_bits, _ptr(st), n_steps, _ptr(md), _ptr(adv),
                                           _ptr(lk) if lk is not None else VP()), "pz_witness_expand")
        return adv, lk

    def witness_expand_dev(self, limbs: int, limb_bits: int, lookup_bits: int, d_steps: int, n_steps: int,
                           d_modulus: int, d_advice: int, d_lookup: int = 0):
        self._chk(self.L.pz_witness_expand_dev(self.ctx, limbs, limb_bits, lookup_bits, VP(d_steps), n_steps,
                                               VP(d_modulus), VP(d_advice), VP(d_lookup)), "pz_witness_expand_dev")

    def circuit_cells(self, kind: int, limbs_n: int, limb_bits: int, lookup_bits: int, n_steps_g: int = 0, n_steps_r: int = 0):
        a = C.c_size_t()
        l = C.c_size_t()
        self._chk(self.L.pz_circuit_cells(kind, limbs_n, limb_bits, lookup_bits, n_steps_g, n_steps_r, C.byref(a), C.byref(l)),
                  "pz_circuit_cells")
        return a.value, l.value

    def circuit_expand_dev(self, kind: int, limbs_n: int, limb_bits: int, lookup_bits: int, inputs, d_steps: int, n_steps_g: int,
                           n_steps_r: int, d_modulus: int, d_advice: int, d_lookup: int = 0, rows: int = 0, col_stride: int = 0):
        """inputs: host uint64 array n | g | x | y | res (pz.h); writes the whole circuit's cell stream, dense or cut into
        columns of `rows` cells stored `col_stride` elements apart"""
        inp = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1)
        self._chk(self.L.pz_circuit_expand_dev(self.ctx, kind, limbs_n, limb_bits, lookup_bits, _ptr(inp), VP(d_steps), n_steps_g,
                                               n_steps_r, VP(d_modulus), VP(d_advice), VP(d_lookup), rows, col_stride),
                  "pz_circuit_expand_dev")

    def circuit_expand_cols_dev(self, kind: int, limbs_n: int, limb_bits: int, lookup_bits: int, inputs, d_steps: int, n_steps_g: int,
                                n_steps_r: int, d_modulus: int, d_advice: int, d_lookup: int, d_col_starts: int, n_adv_cols: int,
                                max_rows: int, lookup_rows: int, col_stride: int):
        """the whole circuit's cell stream with the advice cells in halo2-lib's break-point column layout (pz.h); d_col_starts: device
        array of n_adv_cols + 1 uint64 from layout.break_points / pz_circuit_break_points"""
        inp = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1)
        self._chk(self.L.pz_circuit_expand_cols_dev(self.ctx, kind, limbs_n, limb_bits, lookup_bits, _ptr(inp), VP(d_steps), n_steps_g,
                                                    n_steps_r, VP(d_modulus), VP(d_advice), VP(d_lookup), VP(d_col_starts), n_adv_cols,
                                                    max_rows, lookup_rows, col_stride), "pz_circuit_expand_cols_dev")

    # ------------------------------------------------------------------ "next" rows: SRS setup, evaluation at a point
    def srs_setup_g1_dev(self, k: int, s, omega, d_g: int = 0, d_g_lagrange: int = 0):
        self._chk(self.L.pz_srs_setup_g1_dev(self.ctx, k, _ptr(_np(s).reshape(4)), _ptr(_np(omega).reshape(4)), VP(d_g),
                                             VP(d_g_lagrange)), "pz_srs_setup_g1_dev")

    def srs_lagrange_from_monomial_dev(self, k: int, omega_inv, n_inv, d_g: int, d_g_lagrange: int):
        self._chk(self.L.pz_srs_lagrange_from_monomial_dev(self.ctx, k, self._fr1(omega_inv), self._fr1(n_inv), VP(d_g), VP(d_g_lagrange)),
                  "pz_srs_lagrange_from_monomial_dev")

    def permutation_sigma_dev(self, d_map_col: int, d_map_row: int, m: int, k: int, omega, delta, d_sigma: int, sigma_stride_u64: int):
        self._chk(self.L.pz_permutation_sigma_dev(self.ctx, VP(d_map_col), VP(d_map_row), m, k, self._fr1(omega), self._fr1(delta),
                                                  VP(d_sigma), sigma_stride_u64), "pz_permutation_sigma_dev")

    def keygen_columns_dev(self, bases: Bases, d_cols: int, n_cols: int, col_stride_u64: int, k: int, log_e: int, omega_n, omega_n_inv,
                           n_inv, coset_gens, d_commit: int, d_ext: int = 0, ext_stride_u64: int = 0):
        g = _np(coset_gens).reshape(-1)
        self._chk(self.L.pz_keygen_columns_dev(self.ctx, bases.handle, VP(d_cols), n_cols, col_stride_u64, k, log_e, self._fr1(omega_n),
                                               self._fr1(omega_n_inv), self._fr1(n_inv), _ptr(g), VP(d_commit), VP(d_ext), ext_stride_u64),
                  "pz_keygen_columns_dev")

    def g1_check_dev(self, d_points: int, n: int) -> int:
        """number of points (device, affine) that are not on the curve"""
        bad = C.c_uint64()
        self._chk(self.L.pz_g1_check_dev(self.ctx, VP(d_points), n, C.byref(bad)), "pz_g1_check_dev")
        return bad.value

    def poly_eval_dev(self, d_coeffs: int, n_cols: int, col_stride_u64: int, n: int, x, d_out: int):
        self._chk(self.L.pz_poly_eval_dev(self.ctx, VP(d_coeffs), n_cols, col_stride_u64, n, _ptr(_np(x).reshape(4)),
                                          VP(d_out)), "pz_poly_eval_dev")

    def poly_eval_multi_dev(self, d_coeffs: int, n_cols: int, col_stride_u64: int, n: int, xs, d_out: int):
        """xs: (n_points, 4), n_points <= 4; d_out: (n_cols, n_points, 4) on the device"""
        x = _np(xs).reshape(-1, 4)
        self._chk(self.L.pz_poly_eval_multi_dev(self.ctx, VP(d_coeffs), n_cols, col_stride_u64, n, _ptr(x), x.shape[0], VP(d_out)),
                  "pz_poly_eval_multi_dev")

    # ------------------------------------------------------------------ "next" rows: products, quotient, openings
    def _fr1(self, x):
        """host pointer to one Fr element; the array is kept referenced until a few calls later (a temporary
        converted from a list would otherwise be freed before the C call reads it)"""
        a = _np(x).reshape(4)
        keep = self.__dict__.setdefault("_keep", [])
        keep.append(a)
        del keep[:-32]
        return _ptr(a)

    def fr_batch_invert_dev(self, d_a: int, n: int):
        self._chk(self.L.pz_fr_batch_invert_dev(self.ctx, VP(d_a), n), "pz_fr_batch_invert_dev")

    def fr_prefix_product_dev(self, d_a: int, n: int, z0, d_z: int):
        self._chk(self.L.pz_fr_prefix_product_dev(self.ctx, VP(d_a), n, self._fr1(z0), VP(d_z)), "pz_fr_prefix_product_dev")

    def permutation_product_dev(self, d_cols: int, col_stride_u64: int, d_sigma: int, sigma_stride_u64: int, m: int,
                                log_n: int, omega, beta, gamma, delta_start, delta, z0, d_z: int):
        self._chk(self.L.pz_permutation_product_dev(self.ctx, VP(d_cols), col_stride_u64, VP(d_sigma), sigma_stride_u64, m,
                                                    log_n, self._fr1(omega), self._fr1(beta), self._fr1(gamma),
                                                    self._fr1(delta_start), self._fr1(delta), self._fr1(z0), VP(d_z)),
                  "pz_permutation_product_dev")

    def lookup_permute_dev(self, d_inputs: int, n_cols: int, col_stride_u64: int, d_table: int, rows: int, value_bits: int,
                           d_perm_inputs: int, d_perm_tables: int, out_stride_u64: int):
        self._chk(self.L.pz_lookup_permute_dev(self.ctx, VP(d_inputs), n_cols, col_stride_u64, VP(d_table), rows, value_bits,
                                               VP(d_perm_inputs), VP(d_perm_tables), out_stride_u64), "pz_lookup_permute_dev")

    def lookup_product_dev(self, d_inputs: int, input_stride_u64: int, d_table: int, d_perm_inputs: int,
                           perm_input_stride_u64: int, d_perm_tables: int, perm_table_stride_u64: int, n_lookups: int, n: int,
                           beta, gamma, z0, d_z: int, z_stride_u64: int):
        self._chk(self.L.pz_lookup_product_dev(self.ctx, VP(d_inputs), input_stride_u64, VP(d_table), VP(d_perm_inputs),
                                               perm_input_stride_u64, VP(d_perm_tables), perm_table_stride_u64, n_lookups, n,
                                               self._fr1(beta), self._fr1(gamma), self._fr1(z0), VP(d_z), z_stride_u64),
                  "pz_lookup_product_dev")

    def permutation_product_sets_dev(self, d_cols: int, col_stride_u64: int, d_sigma: int, sigma_stride_u64: int, m: int,
                                     chunk_len: int, log_n: int, usable_rows: int, omega, beta, gamma, delta, d_z: int,
                                     z_stride_u64: int):
        self._chk(self.L.pz_permutation_product_sets_dev(self.ctx, VP(d_cols), col_stride_u64, VP(d_sigma), sigma_stride_u64, m,
                                                         chunk_len, log_n, usable_rows, self._fr1(omega), self._fr1(beta),
                                                         self._fr1(gamma), self._fr1(delta), VP(d_z), z_stride_u64),
                  "pz_permutation_product_sets_dev")

    def quotient_gate_dev(self, d_adv_ext: int, adv_stride_u64: int, d_sel_ext: int, sel_stride_u64: int, n_cols: int,
                          log_ext: int, rot_step: int, y, d_h: int):
        self._chk(self.L.pz_quotient_gate_dev(self.ctx, VP(d_adv_ext), adv_stride_u64, VP(d_sel_ext), sel_stride_u64,
                                              n_cols, log_ext, rot_step, self._fr1(y), VP(d_h)), "pz_quotient_gate_dev")

    def quotient_permutation_dev(self, d_cols_ext: int, col_stride_u64: int, d_sigma_ext: int, sigma_stride_u64: int,
                                 d_z_ext: int, z_stride_u64: int, n_sets: int, chunk_len: int, m_total: int, log_ext: int,
                                 rot_step: int, last_rotation: int, d_l0: int, d_l_last: int, d_l_active: int, beta, gamma,
                                 delta, coset_g, omega_ext, y, d_h: int):
        self._chk(self.L.pz_quotient_permutation_dev(
            self.ctx, VP(d_cols_ext), col_stride_u64, VP(d_sigma_ext), sigma_stride_u64, VP(d_z_ext), z_stride_u64, n_sets,
            chunk_len, m_total, log_ext, rot_step, last_rotation, VP(d_l0), VP(d_l_last), VP(d_l_active), self._fr1(beta),
            self._fr1(gamma), self._fr1(delta), self._fr1(coset_g), self._fr1(omega_ext), self._fr1(y), VP(d_h)),
            "pz_quotient_permutation_dev")

    def quotient_permutation_part_dev(self, d_cols_ext: int, col_stride_u64: int, d_sigma_ext: int, sigma_stride_u64: int, d_z_ext: int,
                                      z_stride_u64: int, n_sets_total: int, set_lo: int, n_sets: int, chunk_len: int, m_cols: int,
                                      head: bool, log_ext: int, rot_step: int, last_rotation: int, d_l0: int, d_l_last: int,
                                      d_l_active: int, beta, gamma, delta, coset_g, omega_ext, y, d_h: int):
        """the permutation lines of evaluate_h for the sets [set_lo, set_lo + n_sets) (pz.h): tiles of extended columns in order"""
        self._chk(self.L.pz_quotient_permutation_part_dev(
            self.ctx, VP(d_cols_ext), col_stride_u64, VP(d_sigma_ext), sigma_stride_u64, VP(d_z_ext), z_stride_u64, n_sets_total, set_lo,
            n_sets, chunk_len, m_cols, int(head), log_ext, rot_step, last_rotation, VP(d_l0), VP(d_l_last), VP(d_l_active),
            self._fr1(beta), self._fr1(gamma), self._fr1(delta), self._fr1(coset_g), self._fr1(omega_ext), self._fr1(y), VP(d_h)),
            "pz_quotient_permutation_part_dev")

    def quotient_lookup_dev(self, d_input_ext: int, input_stride_u64: int, d_table_ext: int, d_perm_input_ext: int,
                            perm_input_stride_u64: int, d_perm_table_ext: int, perm_table_stride_u64: int, d_z_ext: int,
                            z_stride_u64: int, n_lookups: int, log_ext: int, rot_step: int, d_l0: int, d_l_last: int,
                            d_l_active: int, beta, gamma, y, d_h: int):
        self._chk(self.L.pz_quotient_lookup_dev(
            self.ctx, VP(d_input_ext), input_stride_u64, VP(d_table_ext), VP(d_perm_input_ext), perm_input_stride_u64,
            VP(d_perm_table_ext), perm_table_stride_u64, VP(d_z_ext), z_stride_u64, n_lookups, log_ext, rot_step, VP(d_l0),
            VP(d_l_last), VP(d_l_active), self._fr1(beta), self._fr1(gamma), self._fr1(y), VP(d_h)), "pz_quotient_lookup_dev")

    def quotient_finish_dev(self, d_h: int, log_n: int, log_e: int, coset_g, omega_ext):
        self._chk(self.L.pz_quotient_finish_dev(self.ctx, VP(d_h), log_n, log_e, self._fr1(coset_g), self._fr1(omega_ext)),
                  "pz_quotient_finish_dev")

    def fr_distribute_powers_dev(self, d_a: int, n_cols: int, col_stride_u64: int, n: int, g, c=None):
        self._chk(self.L.pz_fr_distribute_powers_dev(self.ctx, VP(d_a), n_cols, col_stride_u64, n, self._fr1(g),
                                                     self._fr1(c) if c is not None else None),
                  "pz_fr_distribute_powers_dev")

    def fr_lincomb_dev(self, d_polys: int, n_cols: int, col_stride_u64: int, n: int, v, d_out: int, accumulate: bool = False):
        self._chk(self.L.pz_fr_lincomb_dev(self.ctx, VP(d_polys), n_cols, col_stride_u64, n, self._fr1(v), VP(d_out),
                                           int(accumulate)), "pz_fr_lincomb_dev")

    def poly_div_linear_dev(self, d_coeffs: int, n_cols: int, col_stride_u64: int, n: int, x, d_q: int, q_stride_u64: int):
        self._chk(self.L.pz_poly_div_linear_dev(self.ctx, VP(d_coeffs), n_cols, col_stride_u64, n, self._fr1(x), VP(d_q),
                                                q_stride_u64), "pz_poly_div_linear_dev")

    def shplonk_begin_dev(self, n: int, sets, points, y, v, d_h: int):
        """sets: list of (list of device pointers, list of point indices, evals array (n_polys, n_points, 4)); points: (T, 4).
        Returns the opaque state for shplonk_finish_dev."""
        n_sets = len(sets)
        npol = (C.c_uint32 * n_sets)(*[len(s_[0]) for s_ in sets])
        npts = (C.c_uint32 * n_sets)(*[len(s_[1]) for s_ in sets])
        ptrs = [p for s_ in sets for p in s_[0]]
        parr = (VP * len(ptrs))(*[VP(p) for p in ptrs])
        idx = [i for s_ in sets for i in s_[1]]
        iarr = (C.c_uint32 * len(idx))(*idx)
        pts = _np(points, 4)
        ev = np.ascontiguousarray(np.concatenate([_np(s_[2]).reshape(-1, 4) for s_ in sets]), dtype=np.uint64)
        st = VP()
        self._chk(self.L.pz_shplonk_begin_dev(self.ctx, n, n_sets, npol, parr, npts, iarr, pts.shape[0], _ptr(pts), _ptr(ev), self._fr1(y),
                                              self._fr1(v), VP(d_h), C.byref(st)), "pz_shplonk_begin_dev")
        return st

    def shplonk_finish_dev(self, state, u, d_h: int, d_h2: int):
        self._chk(self.L.pz_shplonk_finish_dev(self.ctx, state, self._fr1(u), VP(d_h), VP(d_h2)), "pz_shplonk_finish_dev")

    # ------------------------------------------------------------------ measurement
    def timing_enable(self, on: bool = True):
        self._chk(self.L.pz_timing_enable(self.ctx, int(on)), "pz_timing_enable")

    def timing_reset(self):
        self._chk(self.L.pz_timing_reset(self.ctx), "pz_timing_reset")

    def timing_get(self, which: int) -> Tuple[float, int]:
        ms = C.c_double()
        n = C.c_uint64()
        self._chk(self.L.pz_timing_get(self.ctx, which, C.byref(ms), C.byref(n)), "pz_timing_get")
        return ms.value, n.value

    # ------------------------------------------------------------------ device memory / context ordering (plain-pointer hosts)
    def dev_alloc(self, nbytes: int) -> int:
        d = VP()
        self._chk(self.L.pz_dev_alloc(self.ctx, nbytes, C.byref(d)), "pz_dev_alloc")
        return d.value or 0

    def dev_free(self, d: int):
        self._chk(self.L.pz_dev_free(self.ctx, VP(d)), "pz_dev_free")

    def dev_arena(self, nbytes: int):
        """pz_dev_arena: reserve one block this context's later allocations (pz_dev_alloc and the library's own buffers) are carved from;
        0 releases it"""
        self._chk(self.L.pz_dev_arena(self.ctx, C.c_size_t(nbytes)), "pz_dev_arena")

    def dev_arena_info(self) -> dict:
        out = (C.c_uint64 * 6)()
        self._chk(self.L.pz_dev_arena_info(self.ctx, out), "pz_dev_arena_info")
        return dict(zip(("bytes", "used", "peak", "largest_hole", "served", "missed"), (int(v) for v in out)))

    def dev_mem_info(self):
        """-> (free, total) bytes as the driver reports them"""
        f, t = C.c_size_t(0), C.c_size_t(0)
        self._chk(self.L.pz_dev_mem_info(self.ctx, C.byref(f), C.byref(t)), "pz_dev_mem_info")
        return int(f.value), int(t.value)

    def upload(self, d_dst: int, arr):
        a = np.ascontiguousarray(arr)
        self._chk(self.L.pz_upload(self.ctx, VP(d_dst), VP(a.ctypes.data), a.nbytes), "pz_upload")

    def download(self, d_src: int, shape, dtype=np.uint64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self._chk(self.L.pz_download(self.ctx, VP(out.ctypes.data), VP(d_src), out.nbytes), "pz_download")
        return out

    def dev_memset(self, d: int, value: int, nbytes: int):
        self._chk(self.L.pz_dev_memset(self.ctx, VP(d), value, nbytes), "pz_dev_memset")

    def dev_copy(self, d_dst: int, d_src: int, nbytes: int):
        self._chk(self.L.pz_dev_copy(self.ctx, VP(d_dst), VP(d_src), nbytes), "pz_dev_copy")

    def dev_copy_2d(self, d_dst: int, dst_pitch: int, d_src: int, src_pitch: int, width: int, rows: int):
        """`rows` runs of `width` bytes, the runs dst_pitch / src_pitch bytes apart (device to device, on the context's stream)"""
        self._chk(self.L.pz_dev_copy_2d(self.ctx, VP(d_dst), dst_pitch, VP(d_src), src_pitch, width, rows), "pz_dev_copy_2d")

    def wait_for(self, other: "Engine"):
        """everything `other` has queued so far happens before what this engine queues from now on"""
        self._chk(self.L.pz_ctx_wait(self.ctx, other.ctx), "pz_ctx_wait")


T_MSM_ACC, T_NTT, T_TRACE, T_EXPAND, T_MSM_ALL, T_MSM_SORT, T_MSM_TREE = 0, 1, 2, 3, 4, 5, 6
