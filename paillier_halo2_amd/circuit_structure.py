"""The STRUCTURE of the reference's circuits as halo2's keygen sees it -- selector positions, copy constraints, constants, lookup
sources, the break-point column layout -- generated at any size without touching a witness value.

In the reference this is what `keygen_vk` / `keygen_pk` extract by running `synthesize` of halo2-lib's builder over the drivers
(/root/reference/src/bench.rs:33-117 with PaillierChip::{encrypt, add}, src/paillier.rs:32-85; reached from bench.rs:161-175): the
structure is a function of the SHAPE only -- key size, limb width, lookup bits, and the bits of the two fixed exponents
(paillier.rs:50-55 pulls m and n out of the witness, so they shape the circuit).  K4 (csrc/pz_witness.hip) writes the VALUES of the
same cells; this module says which cells are tied together, so that prover.keygen can build the sigma polynomials and the
selectors of the very circuit K4 fills (the connected proof of tests/test_gpu_connected_proof.py and bench.py's with_next_rows).

How: one value-free walk over a single mul_mod block records, per cell, the cell it copies (or which operand limb, or which
constant), the gate windows and the looked-up cells; the circuit's ~6000 identical blocks are that template tiled with numpy, the
operand limbs resolved per step from pow_mod_fixed_exp's schedule.  Equality classes become cycles of sigma by one sort (torch, on
the GPU when there is one: 4 x 10^8 cells at config c2).

The per-primitive patterns restate halo2-lib / biguint-halo2 [D] like K4's: layout parity with the
reference's floating dependency versions is unpinned; tests/test_circuit_structure.py holds this module against the oracle's
independent restatement (oracle/pyref.py::expand_circuit_cells_wired, gate_mask_circuit) cell for cell.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import layout
from .prover import CircuitStructure


class _Walk:
    """records cells in emission order: what each copies, constants, gate windows, lookups.  A cell reference is a global stream
    index (int) or, inside the block template, an operand limb ('a' | 'b' | 'n', j)."""

    def __init__(self, base: int = 0):
        self.base = base
        self.src: List = []        # -1 | int (absolute) | (kind, j)
        self.cval: List = []       # None | int
        self.gates: List[int] = []     # local positions
        self.lk: List = []         # cell references, in lookup-stream order

    def put(self, src=-1) -> int:
        self.src.append(-1 if src is None else src)
        self.cval.append(None)
        return self.base + len(self.src) - 1

    def putc(self, v: int) -> int:
        self.src.append(-1)
        self.cval.append(v)
        return self.base + len(self.src) - 1

    def pair(self, src, copy: int):
        if src is None:
            return
        assert self.src[copy - self.base] == -1
        self.src[copy - self.base] = src

    def gate(self, cell: int):
        self.gates.append(cell - self.base)

    @property
    def n(self) -> int:
        return len(self.src)


def _range_check(w: _Walk, xcell, bits: int, lb: int):
    """RangeChip::range_check: digits + running recomposition, then the top digit's tail gate; -> the cell holding the value"""
    k = -(-bits // lb)
    rem = bits % lb
    last_cell = xcell
    holder = xcell
    if k > 1:
        c0 = w.put()
        w.lk.append(c0)
        w.gate(c0)
        acc_cell = c0
        for gi in range(1, k):
            last_cell = w.put()
            w.lk.append(last_cell)
            w.putc(1 << (lb * gi))
            acc_cell = w.put()
            if gi < k - 1:
                w.gate(acc_cell)
        w.pair(xcell, acc_cell)
        holder = acc_cell
    else:
        w.lk.append(xcell)
    if rem == 1:
        z = w.putc(0)
        w.gate(z)
        w.put(last_cell); w.put(last_cell); w.put(last_cell)
    elif rem > 1:
        z = w.putc(0)
        w.gate(z)
        w.put(last_cell); w.putc(1 << (lb - rem)); w.lk.append(w.put())
    return holder


def _assign(w: _Walk, nl: int, limb_bits: int, lb: int) -> List[int]:
    cells = [w.put() for _ in range(nl)]
    for c in cells:
        _range_check(w, c, limb_bits, lb)
    return cells


def _mul_cells(w: _Walk, xs: Sequence, ys: Sequence, D: int) -> List[int]:
    zc = w.putc(0)
    xe = list(xs) + [zc] * (D - len(xs))
    ye = list(ys) + [zc] * (D - len(ys))
    prod = []
    for i in range(D):
        z = w.putc(0)
        w.gate(z)
        cell = None
        for j in range(i + 1):
            w.put(xe[j]); w.put(ye[i - j])
            cell = w.put()
            if j < i:
                w.gate(cell)
        prod.append(cell)
    return prod


def _is_equal(w: _Walk, xc, yc) -> None:
    c_d = w.put()
    w.gate(c_d)
    w.put(yc); w.putc(1); w.put(xc)
    c_z = w.put()
    w.gate(c_z)
    c_a = w.put(c_d)
    w.put(); w.putc(1)
    z2 = w.putc(0)
    w.gate(z2)
    w.put(c_a); w.put(c_z); w.putc(0)


def _div_mod(w: _Walk, vcell, limb_bits: int) -> Tuple[int, int]:
    c_qd = w.put()
    c_rd = w.put()
    z = w.putc(0)
    w.gate(z)
    w.put(c_qd); w.putc(1 << limb_bits)
    c_pr = w.put()
    d = w.put()
    w.gate(d)
    w.put(c_pr); w.putc(1); w.put(vcell)
    _is_equal(w, c_rd, None)
    return c_qd, c_rd


def _mul_mod(w: _Walk, a: Sequence, b: Sequence, nfresh: Sequence, L: int, limb_bits: int, lb: int) -> List[int]:
    """BigUintChip::mul_mod: assign q, n, r; the two limb convolutions; qn + r; is_equal_muled's carry chain; r < n.  -> r's cells"""
    ql = _assign(w, L, limb_bits, lb)
    nl = _assign(w, L, limb_bits, lb)
    rl = _assign(w, L, limb_bits, lb)
    for c, srcn in zip(nl, nfresh):
        w.pair(srcn, c)
    D = 2 * L - 1
    p_ab = _mul_cells(w, a, b, D)
    p_qn = _mul_cells(w, ql, nl, D)
    qnr = list(p_qn)
    for i in range(L):
        g = w.put(p_qn[i])
        w.gate(g)
        w.putc(1); w.put(rl[i])
        qnr[i] = w.put()
    m = (1 << limb_bits) - 1
    MAX = L * m * m + m
    cb = (2 * MAX).bit_length() - limb_bits
    c_zero = w.putc(0)
    c_one = w.putc(1)
    carry, accx, eq_cell = c_zero, c_zero, c_one
    for i in range(D):
        c_diff = w.put()
        w.gate(c_diff)
        w.put(qnr[i]); w.putc(1); w.put(p_ab[i])
        g = w.put(c_diff)
        w.gate(g)
        w.put(carry); w.putc(1)
        s1 = w.put()
        w.gate(s1)
        w.putc(MAX); w.putc(1)
        c_s = w.put()
        new_carry, cmod = _div_mod(w, c_s, limb_bits)
        g = w.put(accx)
        w.gate(g)
        w.putc(1); w.putc(MAX)
        c_t = w.put()
        q_acc, mod_acc = _div_mod(w, c_t, limb_bits)
        _is_equal(w, cmod, mod_acc)
        g = w.putc(0)
        w.gate(g)
        w.put(eq_cell); w.put()
        eq_cell = w.put()
        accx = q_acc
        if i < D - 1:
            _range_check(w, new_carry, cb, lb)
        else:
            _is_equal(w, new_carry, accx)
            g = w.putc(0)
            w.gate(g)
            w.put(eq_cell); w.put()
            eq_cell = w.put()
        carry = new_carry
    borrow = c_zero
    for i in range(L):
        g = w.put(nl[i])
        w.gate(g)
        w.putc(1); w.put(borrow); w.put()
        w.put()
        c_lt = w.put()
        c_out = w.put()
        g = w.put(rl[i])
        w.gate(g)
        w.put(c_lt); w.putc(1 << limb_bits); w.put()
        _range_check(w, c_out, limb_bits, lb)
        borrow = c_lt
    w.put(borrow)
    return rl


@dataclass
class _Template:
    cells: int
    self_or_local: np.ndarray        # int64 [cells]: local source index, or the cell's own index
    ext: Dict[str, Tuple[np.ndarray, np.ndarray]]   # kind -> (positions, limb index)
    const_pos: np.ndarray            # positions of constant cells
    const_val: List[int]             # their values
    gates: np.ndarray
    lk: np.ndarray                   # local positions, lookup-stream order
    r_cells: np.ndarray              # local positions of r's limb cells


def _block_template(L: int, limb_bits: int, lb: int) -> _Template:
    w = _Walk()
    A = [("a", j) for j in range(L)]
    B = [("b", j) for j in range(L)]
    Nf = [("n", j) for j in range(L)]
    rl = _mul_mod(w, A, B, Nf, L, limb_bits, lb)
    n = w.n
    sol = np.arange(n, dtype=np.int64)
    ext: Dict[str, Tuple[List[int], List[int]]] = {"a": ([], []), "b": ([], []), "n": ([], [])}
    for i, s in enumerate(w.src):
        if isinstance(s, tuple):
            ext[s[0]][0].append(i)
            ext[s[0]][1].append(s[1])
        elif s >= 0:
            sol[i] = s
    cpos = [i for i, v in enumerate(w.cval) if v is not None]
    assert all(isinstance(c, int) for c in w.lk)
    return _Template(n, sol, {k: (np.asarray(p, dtype=np.int64), np.asarray(j, dtype=np.int64)) for k, (p, j) in ext.items()},
                     np.asarray(cpos, dtype=np.int64), [w.cval[i] for i in cpos], np.asarray(w.gates, dtype=np.int64),
                     np.asarray(w.lk, dtype=np.int64), np.asarray(rl, dtype=np.int64))


def _uniform_template(L: int, limb_bits: int, lb: int) -> _Template:
    """one exponent bit of the uniform-shape circuit (pz_paillier_encrypt_uniform, circuit kind 2): mul_mod(acc, sq), the limb-wise
    select(bit, product, acc) = 8 cells per limb [d | 1 | acc | prod | acc | bit | d | out] with gates at 0 and 4, square_mod(sq).
    Operand limbs: 'a' = acc, 'b' = sq, 'n' = the refreshed n^2, 's' = the bit's cell.  r_cells holds the NEW acc (select outputs)
    then the NEW sq (the square's remainder), L cells each."""
    w = _Walk()
    A = [("a", j) for j in range(L)]
    B = [("b", j) for j in range(L)]
    Nf = [("n", j) for j in range(L)]
    mul = _mul_mod(w, A, B, Nf, L, limb_bits, lb)
    outs = []
    for t in range(L):
        c_d = w.put()
        w.gate(c_d)
        w.putc(1); w.put(("a", t)); w.put(mul[t])
        g = w.put(("a", t))
        w.gate(g)
        w.put(("s", 0)); w.put(c_d)
        outs.append(w.put())
    sq = _mul_mod(w, B, B, Nf, L, limb_bits, lb)
    n = w.n
    sol = np.arange(n, dtype=np.int64)
    ext: Dict[str, Tuple[List[int], List[int]]] = {"a": ([], []), "b": ([], []), "n": ([], []), "s": ([], [])}
    for i, s_ in enumerate(w.src):
        if isinstance(s_, tuple):
            ext[s_[0]][0].append(i)
            ext[s_[0]][1].append(s_[1])
        elif s_ >= 0:
            sol[i] = s_
    cpos = [i for i, v in enumerate(w.cval) if v is not None]
    return _Template(n, sol, {k: (np.asarray(p, dtype=np.int64), np.asarray(j, dtype=np.int64)) for k, (p, j) in ext.items()},
                     np.asarray(cpos, dtype=np.int64), [w.cval[i] for i in cpos], np.asarray(w.gates, dtype=np.int64),
                     np.asarray(w.lk, dtype=np.int64), np.asarray(outs + sq, dtype=np.int64))


@dataclass
class StructureArrays:
    """stream-level structure: everything indexed by the advice stream's cell index"""
    n_cells: int
    src: np.ndarray              # int64 [n_cells]: the advice cell each cell copies (itself if none); constants point at -(1 + const id)
    gate_mask: np.ndarray        # uint8 [n_cells]
    lookup_src: np.ndarray       # int64 [n_lookups]
    constants: List[int]         # distinct constants in order of first use
    result_cell: int
    n_steps_g: int
    n_steps_r: int


def _exp_bits(e: int) -> List[int]:
    return [(e >> i) & 1 for i in range(e.bit_length())]


def stream_structure(kind: str, enc_bits: int, limb_bits: int, lb: int, exp_g: int = 0, exp_r: int = 0, device: Optional[str] = None) -> StructureArrays:
    """kind 'encrypt': exp_g = the message m, exp_r = the modulus n -- only their BITS are used, as in the reference's circuit
    (pow_mod_fixed_exp, paillier.rs:50-55); kind 'add': no exponents; kind 'encrypt_uniform' (the uniform-shape circuit, SURVEY 8f rank
    4): exp_g is ignored -- the message's bits are witness cells, ONE structure serves every message of a key.
    device: None -> numpy arrays; a torch device ("cuda") -> the template is tiled THERE and `src` / `lookup_src` are tensors on it (at
    config c2 the tiled arrays are 3.2 GB: 0.9 s of host numpy against a few ms; `columns` takes either)."""
    dev = None
    if device is not None:
        import torch

        dev = torch.device(device)

    def tile(bases: np.ndarray, sol: np.ndarray, assigns=()) -> "np.ndarray | torch.Tensor":
        """flattened [len(bases)][len(sol)] array of bases[i] + sol[j], columns `pos` overwritten by `vals` ([ns][len(pos)] or [len(pos)])"""
        if dev is None:
            t_ = bases[:, None] + sol[None, :]
            for pos, vals in assigns:
                t_[:, pos] = vals
            return t_.reshape(-1)
        up = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int64).to(dev)
        t_ = up(bases)[:, None] + up(sol)[None, :]
        for pos, vals in assigns:
            v = up(vals)
            t_[:, up(pos)] = v if v.dim() == 2 else v[None, :].expand(t_.shape[0], -1)
        return t_.reshape(-1)

    Ln = enc_bits // limb_bits
    L = 2 * Ln
    tm = _block_template(L, limb_bits, lb)
    const_id: Dict[int, int] = {}
    constants: List[int] = []

    def cid(v: int) -> int:
        if v not in const_id:
            const_id[v] = len(constants)
            constants.append(v)
        return const_id[v]

    # ---- prefix: the four assign_integer, square, refresh, load_zero (global indices from 0)
    w = _Walk()
    n_c = _assign(w, Ln, limb_bits, lb)
    g_c = _assign(w, Ln, limb_bits, lb)
    x_c = _assign(w, Ln, limb_bits, lb)
    y_c = _assign(w, Ln, limb_bits, lb)
    prod = _mul_cells(w, n_c, n_c, 2 * Ln - 1)
    inc = layout.refresh_aux(limb_bits, Ln, Ln)
    w.putc(0)
    cur: List = list(prod) + [None] * (len(inc) - len(prod))
    for i in range(len(inc)):
        limb = cur[i]
        for j in range(inc[i] + 1):
            qd, rd = _div_mod(w, limb, limb_bits)
            if j == 0:
                cur[i] = rd
            else:
                g = w.put(cur[i + j])
                w.gate(g)
                w.putc(1); w.put(rd)
                cur[i + j] = w.put()
            limb = qd
    fresh = []
    for c in cur:
        holder = _range_check(w, c, limb_bits, lb)
        fresh.append(c if c is not None else holder)
    zero = w.putc(0)
    ext_l = lambda limbs: list(limbs) + [zero] * (L - len(limbs))
    parts_src: List[np.ndarray] = []
    parts_mask: List[np.ndarray] = []
    parts_lk: List[np.ndarray] = []

    def flush(walk: _Walk):
        """a finished Python-walked part -> arrays"""
        n = walk.n
        s = np.arange(walk.base, walk.base + n, dtype=np.int64)
        for i, v in enumerate(walk.src):
            if v != -1:
                s[i] = v
        for i, v in enumerate(walk.cval):
            if v is not None:
                s[i] = -(1 + cid(v))
        mk = np.zeros(n, dtype=np.uint8)
        mk[np.asarray(walk.gates, dtype=np.int64)] = 1
        parts_src.append(s)
        parts_mask.append(mk)
        parts_lk.append(np.asarray(walk.lk, dtype=np.int64))
        return walk.base + n

    off = flush(w)
    tm_const_ids = np.asarray([cid(v) for v in tm.const_val], dtype=np.int64)
    tm_mask = np.zeros(tm.cells, dtype=np.uint8)
    tm_mask[tm.gates] = 1
    fresh_arr = np.asarray(fresh, dtype=np.int64)

    def blocks(off: int, a_cells: np.ndarray, b_cells: np.ndarray):
        """ns mul_mod blocks starting at stream index `off`; a_cells / b_cells: int64 [ns][L] global cells of the operands' limbs"""
        ns = a_cells.shape[0]
        bases = off + tm.cells * np.arange(ns, dtype=np.int64)
        assigns = []
        for kind_, cells_ in (("a", a_cells), ("b", b_cells)):
            pos, j = tm.ext[kind_]
            assigns.append((pos, cells_[:, j]))
        pos, j = tm.ext["n"]
        assigns.append((pos, fresh_arr[j]))
        assigns.append((tm.const_pos, -(1 + tm_const_ids)))
        parts_src.append(tile(bases, tm.self_or_local, assigns))
        parts_mask.append(np.tile(tm_mask, ns))
        parts_lk.append(tile(bases, tm.lk))
        return off + ns * tm.cells, bases[:, None] + tm.r_cells[None, :]      # r cells of every block

    n_steps = [0, 0]
    if kind in ("encrypt", "encrypt_uniform"):
        results = []
        chains = [(ext_l(g_c), exp_g), (ext_l(y_c), exp_r)]
        if kind == "encrypt_uniform":
            # g^m over ALL enc_bits bits of m IN the circuit: assign_constant(1), load_zero, then per limb of m num_to_bits and per bit
            # the (mul_mod, select, square_mod) block -- the same shape for every message
            W_ = limb_bits
            ut = _uniform_template(L, limb_bits, lb)
            ut_const = np.asarray([cid(v) for v in ut.const_val], dtype=np.int64)
            ut_mask = np.zeros(ut.cells, dtype=np.uint8)
            ut_mask[ut.gates] = 1
            wc = _Walk(off)
            one = wc.putc(1)
            z2 = wc.putc(0)
            off = flush(wc)
            acc_cells = np.asarray([one] + [z2] * (L - 1), dtype=np.int64)
            sq_cells = np.asarray(ext_l(g_c), dtype=np.int64)
            for li in range(Ln):
                wb = _Walk(off)
                bit_cells = [wb.put()]
                wb.gate(bit_cells[0])
                acc_cell = bit_cells[0]
                for i in range(1, W_):
                    bit_cells.append(wb.put())
                    wb.putc(1 << i)
                    acc_cell = wb.put()
                    if i < W_ - 1:
                        wb.gate(acc_cell)
                wb.pair(x_c[li], acc_cell)
                for bc in bit_cells:
                    g_ = wb.putc(0)
                    wb.gate(g_)
                    wb.put(bc); wb.put(bc); wb.put(bc)
                off = flush(wb)
                # the limb's W_ bits: blocks chained through acc (select outputs) and sq (square remainders)
                bases = off + ut.cells * np.arange(W_, dtype=np.int64)
                new_acc = bases[:, None] + ut.r_cells[None, :L]
                new_sq = bases[:, None] + ut.r_cells[None, L:]
                a_cells = np.concatenate([acc_cells[None, :], new_acc[:-1]])
                b_cells = np.concatenate([sq_cells[None, :], new_sq[:-1]])
                assigns = []
                for kind_, cells_ in (("a", a_cells), ("b", b_cells)):
                    pos, j = ut.ext[kind_]
                    assigns.append((pos, cells_[:, j]))
                pos, j = ut.ext["n"]
                assigns.append((pos, fresh_arr[j]))
                pos, _ = ut.ext["s"]
                assigns.append((pos, np.repeat(np.asarray(bit_cells, dtype=np.int64)[:, None], len(pos), axis=1)))
                assigns.append((ut.const_pos, -(1 + ut_const)))
                parts_src.append(tile(bases, ut.self_or_local, assigns))
                parts_mask.append(np.tile(ut_mask, W_))
                parts_lk.append(tile(bases, ut.lk))
                off += W_ * ut.cells
                acc_cells, sq_cells = new_acc[-1], new_sq[-1]
            n_steps[0] = 2 * Ln * W_
            results.append(acc_cells)
            chains = [None, chains[1]]
        for ci, chain in enumerate(chains):
            if chain is None:
                continue
            base_limbs, e = chain
            wc = _Walk(off)
            one = wc.putc(1)
            z2 = wc.putc(0)
            off = flush(wc)
            bits = _exp_bits(e)
            # pow_mod_fixed_exp's schedule: per bit the squaring step (cur, cur); on a set bit then (acc, cur).  Which BLOCK produced
            # each operand is structure: block indices are assigned first, the operand cells follow from them
            ns = len(bits) + sum(bits)
            n_steps[ci] = ns
            a_blk = np.empty(ns, dtype=np.int64)      # producing block of operand a (-1: the base, -2: the constant one)
            b_blk = np.empty(ns, dtype=np.int64)
            sq_blk, acc_blk, t = -1, -2, 0
            for bit in bits:
                a_blk[t] = b_blk[t] = sq_blk
                cur_blk, sq_blk = sq_blk, t
                t += 1
                if bit:
                    a_blk[t], b_blk[t] = acc_blk, cur_blk
                    acc_blk = t
                    t += 1
            r_of = off + tm.cells * np.arange(ns, dtype=np.int64)[:, None] + tm.r_cells[None, :]
            special = np.stack([np.asarray([one] + [z2] * (L - 1), dtype=np.int64), np.asarray(base_limbs, dtype=np.int64)])   # -2, -1
            table = np.concatenate([special, r_of]) if ns else special
            a_cells, b_cells = table[a_blk + 2], table[b_blk + 2]
            if ns:
                off, _ = blocks(off, a_cells, b_cells)
            results.append(table[acc_blk + 2])
        gm, rn = results
        off, rfin = blocks(off, np.asarray(gm, dtype=np.int64)[None, :], np.asarray(rn, dtype=np.int64)[None, :])
    else:
        off, rfin = blocks(off, np.asarray(ext_l(x_c), dtype=np.int64)[None, :], np.asarray(ext_l(y_c), dtype=np.int64)[None, :])
    c_limbs = rfin[0].tolist()
    ws = _Walk(off)
    res_c = _assign(ws, L, limb_bits, lb)
    g0 = ws.putc(0)
    eq_cell = ws.putc(1)
    for cc, rc in zip(c_limbs, res_c):
        _is_equal(ws, cc, rc)
        g = ws.putc(0)
        ws.gate(g)
        ws.put(eq_cell); ws.put()
        eq_cell = ws.put()
    off = flush(ws)
    if dev is None:
        src, lookup_src = np.concatenate(parts_src), np.concatenate(parts_lk)
    else:
        cat = lambda parts: torch.cat([p_ if isinstance(p_, torch.Tensor) else torch.as_tensor(p_, dtype=torch.int64).to(dev) for p_ in parts])
        src, lookup_src = cat(parts_src), cat(parts_lk)
    src[eq_cell] = -(1 + cid(1))          # assert_equal_fresh's result is constrained to the constant 1 (bench.rs:74)
    return StructureArrays(n_cells=off, src=src, gate_mask=np.concatenate(parts_mask), lookup_src=lookup_src,
                           constants=constants, result_cell=eq_cell, n_steps_g=n_steps[0], n_steps_r=n_steps[1])


def columns(sa: StructureArrays, k: int, lb: int, minimum_rows: int = layout.MINIMUM_ROWS_BENCH, blinding_factors: int = layout.BLINDING_FACTORS,
            device: Optional[str] = None, keep_on_device: bool = False, break_rows: Optional[int] = None):
    """stream structure -> (CircuitStructure for prover.keygen, starts).  The permutation covers [advice | lookup advice | constants];
    every equality class becomes one cycle of sigma (cells in increasing (column, row) order).
    minimum_rows: the argument of the tester's calculate_params (layout.RowBudget): it fixes the NUMBER of advice / lookup-advice columns
    (20 on the reference's bench path, /root/reference/src/bench.rs:161-171; 9 under MockProver, src/paillier.rs:167-171); columns are
    FILLED to 2^k - (blinding_factors + 3) rows (break_rows overrides), so with 20 the last configured column can stay empty -- it is a
    column of the circuit all the same (selector all zero, identity permutation, committed and opened).
    starts: n_adv + 1 break points -- what K4 takes (pz_circuit_expand_cols_dev); a configured column the cells do not reach starts and
    ends at the stream's end.
    keep_on_device: selectors / map_col / map_row stay tensors on `device` (uint8 / int32) for prover.keygen instead of travelling to the
    host and back (4 GB each way at config c2)."""
    import torch

    n = 1 << k
    rb = layout.row_budget(k, minimum_rows, blinding_factors, break_rows)
    max_rows = rb.max_rows
    assert max_rows <= n - (blinding_factors + 1)
    starts = layout.break_points(sa.gate_mask, max_rows).astype(np.int64)
    A_used = starts.shape[0] - 1
    NC, NL, NK = sa.n_cells, sa.lookup_src.shape[0], len(sa.constants)
    A = rb.columns_for(NC, filled=A_used)
    Lk = rb.columns_for(NL)
    m = A + Lk + 1
    assert NK <= max_rows
    if device is None:
        device = "cuda" if (torch.cuda.is_available() and NC > (1 << 22)) else "cpu"
    dev = torch.device(device)
    T = NC + NL + NK + (A_used - 1)
    st = torch.from_numpy(starts).to(dev)
    # ---- node -> flat position (column * n + row)
    pos = torch.empty(T, dtype=torch.int64, device=dev)
    c = torch.arange(NC, dtype=torch.int64, device=dev)
    col = torch.searchsorted(st, c, right=True) - 1
    col.clamp_(max=A_used - 1)
    pos[:NC] = col * n + (c - st[col])
    del c, col
    t = torch.arange(NL, dtype=torch.int64, device=dev)
    pos[NC:NC + NL] = (A + t // max_rows) * n + t % max_rows
    del t
    pos[NC + NL:NC + NL + NK] = (A + Lk) * n + torch.arange(NK, dtype=torch.int64, device=dev)
    j = torch.arange(1, A_used, dtype=torch.int64, device=dev)
    pos[NC + NL + NK:] = (j - 1) * n + (st[j] - st[j - 1])
    # ---- what every node copies
    src = torch.arange(T, dtype=torch.int64, device=dev)
    s_adv = torch.as_tensor(sa.src).to(dev)
    src[:NC] = torch.where(s_adv < 0, NC + NL - 1 - s_adv, s_adv)          # -(1 + id) -> constant node NC + NL + id
    del s_adv
    src[NC:NC + NL] = torch.as_tensor(sa.lookup_src).to(dev)
    src[NC + NL + NK:] = st[j]
    del j
    while True:                                                           # roots by pointer jumping (chains are a few links long)
        nxt = src[src]
        if torch.equal(nxt, src):
            break
        src = nxt
    del nxt
    # ---- cycles: nodes sorted by (root, position); each maps to its successor, the last of a class to the first
    key_root = src
    touched = torch.zeros(T, dtype=torch.bool, device=dev)
    ids = torch.arange(T, dtype=torch.int64, device=dev)
    nonroot = key_root != ids
    touched[nonroot] = True
    touched[key_root[nonroot]] = True
    del nonroot
    members = ids[touched]
    del ids, touched
    r_m, p_m = key_root[members], pos[members]
    del key_root
    # sorted by (class, position): one sort of the combined key (class < T < 2^31, position < m n < 2^31 at every supported size)
    span = m * n
    assert T < (1 << 31) and span < (1 << 31)
    key, _ = torch.sort(r_m * span + p_m)
    r_m, p_m = key // span, key % span
    del key, members
    first = torch.ones_like(r_m, dtype=torch.bool)
    first[1:] = r_m[1:] != r_m[:-1]
    firsts = torch.nonzero(first).view(-1)                 # index of every class's first member ...
    start_of = firsts[torch.cumsum(first.to(torch.int64), 0) - 1]   # ... broadcast over its members (a cumsum, not torch.cummax: that
    #                                                                   scan-with-indices kernel took 0.9 s on 2 x 10^8 elements)
    nxt_p = torch.empty_like(p_m)
    nxt_p[:-1] = p_m[1:]
    last = torch.ones_like(first)
    last[:-1] = first[1:]
    nxt_p[last] = p_m[start_of[last]]
    image = torch.arange(m * n, dtype=torch.int64, device=dev)
    image[p_m] = nxt_p
    map_col, map_row = (image // n).to(torch.int32).view(m, n), (image % n).to(torch.int32).view(m, n)
    del image, p_m, nxt_p, r_m
    # ---- selectors: the gate of a shared break cell is enabled in the column it starts
    gi = torch.nonzero(torch.from_numpy(sa.gate_mask).to(dev)).view(-1)
    sel = torch.zeros(A * n, dtype=torch.uint8, device=dev)
    sel[pos[gi]] = 1
    selectors = sel.view(A, n)
    if not keep_on_device:
        map_col = map_col.cpu().numpy().view(np.uint32)
        map_row = map_row.cpu().numpy().view(np.uint32)
        selectors = selectors.cpu().numpy()
    cs = CircuitStructure(k=k, lookup_bits=lb, max_rows=max_rows, blinding_factors=blinding_factors, selectors=selectors, n_lk=Lk,
                          constants=list(sa.constants), map_col=map_col, map_row=map_row, minimum_rows=minimum_rows, n_adv_used=A_used)
    starts = np.concatenate([starts, np.full(A - A_used, NC, dtype=np.int64)])
    return cs, starts.astype(np.uint64)
