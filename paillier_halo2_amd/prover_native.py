"""The library's own keygen / create_proof (include/pz.h "patch point D as entry points": pz_pk_* / pz_proof_*, one call per transcript
round -- the composition of host/create_proof.hpp behind the C ABI), driven from Python: what a reference prover patched at
/root/reference/src/bench.rs:161-171 calls with halo2's transcript in between.  Same inputs and the same Proof layout as prover.py, so
the same checker verifies both."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import consts
from ._lib import VP
from .engine import Bases, Engine
from .prover import CHUNK, Challenges, CircuitStructure, Proof, Transcript

M = consts.fr_mont_limbs


def _host(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else a


def _p(a: np.ndarray):
    return VP(a.ctypes.data)


class NativeKey:
    """pz_pk: the proving key resident on the device + one proof's workspace; serves one proof at a time"""

    def __init__(self, eng: Engine, st: CircuitStructure, bases_lagrange: Bases, bases_monomial: Bases, tile: int = 64,
                 ext_resident_cols: Optional[int] = None):
        """st's selectors / map_col / map_row: numpy arrays (pz_pk_create: uploaded by the library, selectors as bytes) or torch tensors
        already on the device (pz_pk_create_dev: used where they are -- circuit_structure.columns(keep_on_device=True)).
        ext_resident_cols: None = the extended key resident (PZ_PK_EXT_ALL); R = the streamed proving key (pz.h)."""
        self.eng, self.st = eng, st
        n = 1 << st.k
        consts_w = np.zeros((len(st.constants), 4), dtype=np.uint64)
        for i, v in enumerate(st.constants):
            consts_w[i] = consts.int_to_limbs(int(v) % consts.FR_R, 4)
        h = VP()
        R = (1 << 64) - 1 if ext_resident_cols is None else int(ext_resident_cols)
        on_dev = hasattr(st.selectors, "is_cuda") and st.selectors.is_cuda
        if on_dev:
            import torch

            sel = st.selectors.to(torch.uint8).contiguous()
            mc, mr = st.map_col.to(torch.int32).contiguous(), st.map_row.to(torch.int32).contiguous()
            assert tuple(sel.shape) == (st.n_adv, n) and tuple(mc.shape) == (st.m, n) == tuple(mr.shape)
            eng._chk(eng.L.pz_pk_create_dev(eng.ctx, bases_lagrange.handle, bases_monomial.handle, st.k, st.lookup_bits, st.blinding_factors,
                                            st.max_rows, st.n_adv, st.n_lk, VP(sel.data_ptr()), _p(consts_w), len(st.constants), VP(mc.data_ptr()),
                                            VP(mr.data_ptr()), tile, R, C.byref(h)), "pz_pk_create_dev")
        else:
            sel = np.ascontiguousarray(_host(st.selectors), dtype=np.uint8)
            mc = np.ascontiguousarray(_host(st.map_col)).view(np.uint32)
            mr = np.ascontiguousarray(_host(st.map_row)).view(np.uint32)
            assert sel.shape == (st.n_adv, n) and mc.shape == (st.m, n) == mr.shape
            eng._chk(eng.L.pz_pk_create(eng.ctx, bases_lagrange.handle, bases_monomial.handle, st.k, st.lookup_bits, st.blinding_factors, st.max_rows,
                                        st.n_adv, st.n_lk, _p(sel), _p(consts_w), len(st.constants), _p(mc), _p(mr), tile, R, C.byref(h)), "pz_pk_create")
        self.handle = h
        out = [C.c_size_t() for _ in range(5)]
        eng._chk(eng.L.pz_pk_info(h, *[C.byref(x) for x in out]), "pz_pk_info")
        self.n_fixed, self.m, self.n_sets, self.blinding_words, self.evals_words = (int(x.value) for x in out)
        assert self.m == st.m and self.n_sets == -(-st.m // CHUNK)

    @classmethod
    def from_device(cls, eng: Engine, ns: "NativeStructure", bases_lagrange: Bases, bases_monomial: Bases, tile: int = 64,
                    ext_resident_cols: Optional[int] = None) -> "NativeKey":
        """pz_pk_create_dev on a NativeStructure's device arrays (no Python-side structure at all)"""
        self = cls.__new__(cls)
        self.eng = eng
        self.st = CircuitStructure(k=ns.k, lookup_bits=ns.lookup_bits, max_rows=ns.max_rows, blinding_factors=ns.blinding_factors,
                                   selectors=np.zeros((ns.n_adv, 0), dtype=np.uint8), n_lk=ns.n_lk, constants=ns.constants(), map_col=None, map_row=None,
                                   minimum_rows=ns.minimum_rows, n_adv_used=ns.n_adv_used)
        h = VP()
        R = (1 << 64) - 1 if ext_resident_cols is None else int(ext_resident_cols)
        eng._chk(eng.L.pz_pk_create_dev(eng.ctx, bases_lagrange.handle, bases_monomial.handle, ns.k, ns.lookup_bits, ns.blinding_factors, ns.max_rows,
                                        ns.n_adv, ns.n_lk, VP(ns.d_selectors), VP(ns._constants), ns.n_constants, VP(ns.d_map_col), VP(ns.d_map_row),
                                        tile, R, C.byref(h)), "pz_pk_create_dev")
        self.handle = h
        out = [C.c_size_t() for _ in range(5)]
        eng._chk(eng.L.pz_pk_info(h, *[C.byref(x) for x in out]), "pz_pk_info")
        self.n_fixed, self.m, self.n_sets, self.blinding_words, self.evals_words = (int(x.value) for x in out)
        return self

    def vk_commitments(self) -> Dict[str, np.ndarray]:
        f, s = np.zeros((self.n_fixed, 8), dtype=np.uint64), np.zeros((self.m, 8), dtype=np.uint64)
        self.eng._chk(self.eng.L.pz_pk_commitments(self.handle, _p(f), _p(s)), "pz_pk_commitments")
        return {"fixed": f, "sigma": s}

    def free(self):
        if self.handle:
            self.eng._chk(self.eng.L.pz_pk_free(self.handle), "pz_pk_free")
            self.handle = None


def create_proof(key: NativeKey, d_cols: int, tr, seed: int = 0, blinding: Optional[np.ndarray] = None) -> Proof:
    """d_cols: device pointer of [m][2^k][4] words (the K4 columns; consumed).  tr: prover.Transcript / HashTranscript / Challenges.
    blinding: optional uint64 array of key.blinding_words caller-supplied random words (else the library's seeded stream)"""
    eng, st, L = key.eng, key.st, key.eng.L
    if isinstance(tr, Challenges):
        tr = Transcript(tr)
    A, Lk, m, S = st.n_adv, st.n_lk, st.m, key.n_sets
    z = lambda cnt: np.zeros((cnt, 8), dtype=np.uint64)
    h = VP()
    adv = z(A + Lk)
    bl = np.ascontiguousarray(blinding, dtype=np.uint64) if blinding is not None else None
    # no caller randomness: the library's seeded stream must be asked for by name (pz.h PZ_BLINDING_SEEDED_TEST_STREAM: tests, benches)
    eng._chk(L.pz_proof_begin(key.handle, VP(d_cols), seed, _p(bl) if bl is not None else None, (1 << 64) - 1 if bl is None else bl.size, C.byref(h), _p(adv)),
             "pz_proof_begin")
    try:
        pr = Proof()
        tr.absorb_affine(adv)
        c_theta = M(tr.squeeze("theta"))         # (the limb arrays are named: a pointer into a temporary would dangle)
        ap, sp = z(Lk), z(Lk)
        eng._chk(L.pz_proof_lookups(h, _p(c_theta), _p(ap), _p(sp)), "pz_proof_lookups")
        tr.absorb_affine(ap, sp)
        c_beta, c_gamma = M(tr.squeeze("beta")), M(tr.squeeze("gamma"))
        cz, czl, crnd = z(S), z(Lk), z(1)
        eng._chk(L.pz_proof_products(h, _p(c_beta), _p(c_gamma), _p(cz), _p(czl), _p(crnd)), "pz_proof_products")
        tr.absorb_affine(cz, czl, crnd)
        c_y = M(tr.squeeze("y"))
        ch_ = z(3)
        eng._chk(L.pz_proof_quotient(h, _p(c_y), _p(ch_)), "pz_proof_quotient")
        tr.absorb_affine(ch_)
        c_x = M(tr.squeeze("x"))
        ev = np.zeros(key.evals_words, dtype=np.uint64)
        eng._chk(L.pz_proof_evaluate(h, _p(c_x), _p(ev)), "pz_proof_evaluate")
        fams = (("advice", A, 4), ("lookup_advice", Lk + 1, 1), ("fixed", key.n_fixed, 1), ("sigma", m, 1), ("perm_z", S, 3), ("lookup_z", Lk, 2),
                ("perm_inputs", Lk, 2), ("perm_tables", Lk, 1), ("random", 1, 1), ("h", 1, 1))
        off, evs = 0, {}
        for name, cnt, pts in fams:
            evs[name] = ev[off:off + cnt * pts * 4].reshape(cnt, pts, 4)
            off += cnt * pts * 4
        assert off == ev.size
        tr.absorb_scalars(*[evs[f] for f, _, _ in fams if f != "h"])
        c_shy, c_shv = M(tr.squeeze("sh_y")), M(tr.squeeze("sh_v"))
        w1, w2 = z(1), z(1)
        eng._chk(L.pz_proof_open_begin(h, _p(c_shy), _p(c_shv), _p(w1)), "pz_proof_open_begin")
        tr.absorb_affine(w1)
        c_shu = M(tr.squeeze("sh_u"))
        ok = C.c_int(0)
        eng._chk(L.pz_proof_open_finish(h, _p(c_shu), _p(w2), C.byref(ok)), "pz_proof_open_finish")
        pr.commitments = {"advice": adv[:A], "lookup_advice": adv[A:], "perm_inputs": ap, "perm_tables": sp, "perm_z": cz, "lookup_z": czl,
                          "random": crnd, "h": ch_, "w1": w1, "w2": w2}
        pr.evals = dict(evs)
        pr.evals["constants"] = evs["lookup_advice"][Lk:]
        pr.evals["lookup_advice"] = evs["lookup_advice"][:Lk]
        pr.h_degree_ok = bool(ok.value)
        return pr
    finally:
        eng._chk(L.pz_proof_free(h), "pz_proof_free")


class NativeStructure:
    """pz_structure: the circuit structure of the reference's drivers generated by the library on the device (csrc/pz_structure.hip,
    include/pz.h pz_circuit_structure_dev) -- the compiled counterpart of circuit_structure.stream_structure + columns.
    kind: "encrypt" | "add" | "encrypt_uniform"; exp_g / exp_r: the message m and the modulus n (integers; only their bits are used)."""

    KINDS = {"encrypt": 0, "add": 1, "encrypt_uniform": 2}

    def __init__(self, eng: Engine, kind: str, enc_bits: int, limb_bits: int, lookup_bits: int, k: int, exp_g: int = 0, exp_r: int = 0,
                 minimum_rows: int = 20, blinding_factors: int = 6):
        self.eng, self.k, self.lookup_bits, self.blinding_factors, self.minimum_rows = eng, k, lookup_bits, blinding_factors, minimum_rows
        Ln = enc_bits // limb_bits
        ew = -(-Ln * limb_bits // 64)
        wg, wr = self._words(exp_g, ew), self._words(exp_r, ew)
        h = VP()
        eng._chk(eng.L.pz_circuit_structure_dev(eng.ctx, self.KINDS[kind], Ln, limb_bits, lookup_bits, k, _p(wg), _p(wr), minimum_rows,
                                                blinding_factors, C.byref(h)), "pz_circuit_structure_dev")
        self.handle = h
        out = [C.c_size_t() for _ in range(9)]
        eng._chk(eng.L.pz_structure_info(h, *[C.byref(x) for x in out]), "pz_structure_info")
        (self.n_adv, self.n_adv_used, self.n_lk, self.max_rows, self.n_constants, self.n_cells, self.n_lookups, self.n_steps_g,
         self.n_steps_r) = (int(x.value) for x in out)
        ptrs = [VP() for _ in range(6)]
        eng._chk(eng.L.pz_structure_arrays(h, *[C.byref(x) for x in ptrs]), "pz_structure_arrays")
        self.d_selectors, self.d_map_col, self.d_map_row, self.d_starts, self._constants, self._starts_host = (int(x.value or 0) for x in ptrs)
        self.m = self.n_adv + self.n_lk + 1

    @staticmethod
    def _words(x: int, Ln: int):
        """an exponent as exactly ceil(limbs_n * limb_bits / 64) 64-bit words (pz.h; the exponent's BITS are what the structure uses)"""
        assert x.bit_length() <= 64 * Ln
        return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(Ln)], dtype=np.uint64)

    def constants(self):
        if not self.n_constants:
            return []
        a = np.ctypeslib.as_array(C.cast(self._constants, C.POINTER(C.c_uint64)), shape=(self.n_constants, 4))
        return [sum(int(w) << (64 * j) for j, w in enumerate(row)) for row in a]

    def starts(self) -> np.ndarray:
        return np.ctypeslib.as_array(C.cast(self._starts_host, C.POINTER(C.c_uint64)), shape=(self.n_adv + 1,)).copy()

    def download(self):
        """-> (selectors u8 [n_adv][n], map_col u32 [m][n], map_row u32 [m][n]) on the host (tests)"""
        n = 1 << self.k
        sel = np.zeros((self.n_adv, n), dtype=np.uint8)
        mc, mr = np.zeros((self.m, n), dtype=np.uint32), np.zeros((self.m, n), dtype=np.uint32)
        for dst, src in ((sel, self.d_selectors), (mc, self.d_map_col), (mr, self.d_map_row)):
            self.eng._chk(self.eng.L.pz_download(self.eng.ctx, _p(dst), VP(src), dst.nbytes), "pz_download")
        return sel, mc, mr

    def key(self, bases_lagrange: Bases, bases_monomial: Bases, tile: int = 64, ext_resident_cols: Optional[int] = None) -> "NativeKey":
        """pz_pk_create_dev on the structure's own device arrays"""
        return NativeKey.from_device(self.eng, self, bases_lagrange, bases_monomial, tile, ext_resident_cols)

    def free(self):
        if self.handle:
            self.eng._chk(self.eng.L.pz_structure_free(self.handle), "pz_structure_free")
            self.handle = None
