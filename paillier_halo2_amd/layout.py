"""Shape arithmetic of the Paillier-encrypt circuit: how many advice / lookup cells the
BigUintChip operations behind PaillierChip::encrypt (paillier.rs:32-60) push into halo2-lib's
Context, hence how many 2^k-row columns, MSMs and NTTs one proof needs.

The cell-emitting code lives in un-vendored dependencies (biguint-halo2, halo2-base, halo2-ecc:
Cargo.toml:9-11, no rev) -- SURVEY.md tags it [D].  The counts below restate the halo2-rsa
BigUintChip lineage on halo2-lib v0.4 gate primitives and are the SPEC the K4 expansion kernel
(csrc/pz_witness.hip) implements; DESIGN.md section 4 lists the per-primitive cell patterns.  Layout parity
with the reference's dependency versions is unpinned (they float); the VALUES in the cells are
determined by the step trace.
"""
from __future__ import annotations

import math
from dataclasses import dataclass


# ---- the row budget of a column (halo2-lib's tester, reached from /root/reference/src/bench.rs:161-171 and src/paillier.rs:167-171) [D]
BLINDING_FACTORS = 6          # cs.blinding_factors() of halo2-lib's circuits: max(3, the basic gate's 4 rotations) + 2
MINIMUM_ROWS_BENCH = 20       # bench_builder: builder.calculate_params(Some(20))   
MINIMUM_ROWS_MOCK = 9         # base_test().run(..): builder.calculate_params(Some(9)) before MockProver::run


@dataclass(frozen=True)
class RowBudget:
    """Two different row counts shape a halo2-lib circuit, and the reference's two test paths differ in the first:
    count_rows = 2^k - minimum_rows   what calculate_params(Some(minimum_rows)) divides the cell totals by: the NUMBER of advice /
                                      lookup-advice columns the circuit is configured with (20 on the bench path, 9 under MockProver);
    max_rows   = 2^k - unusable_rows  how far a column is FILLED: FlexGateConfig::max_rows = 2^k - cs.minimum_rows(), and
                                      cs.minimum_rows() = blinding_factors + 3 = 9 -- what assign_with_constraints breaks columns at.
    With minimum_rows = 20 the configured column count can exceed the columns the cells fill (the last one stays empty: it is still
    committed, opened and part of the permutation).  Both numbers restate dependency behaviour (halo2-lib v0.4 lineage, SURVEY tag
    [D]: unpinned); `break_rows` overrides max_rows for a dependency version that fills columns to 2^k - minimum_rows instead."""
    k: int
    minimum_rows: int
    unusable_rows: int

    @property
    def n(self) -> int:
        return 1 << self.k

    @property
    def count_rows(self) -> int:
        return self.n - self.minimum_rows

    @property
    def max_rows(self) -> int:
        return self.n - self.unusable_rows

    def columns_for(self, cells: int, filled: int | None = None) -> int:
        """configured columns for a stream of `cells` cells of which the fill needs `filled` (default: a plain cut at max_rows)"""
        used = -(-cells // self.max_rows) if filled is None else filled
        return max(used, -(-cells // self.count_rows))


def row_budget(k: int, minimum_rows: int = MINIMUM_ROWS_BENCH, blinding_factors: int = BLINDING_FACTORS, break_rows: int | None = None) -> RowBudget:
    unusable = blinding_factors + 3 if break_rows is None else (1 << k) - break_rows
    assert unusable >= blinding_factors + 1 and minimum_rows >= 0
    return RowBudget(k, minimum_rows, unusable)


def range_check_cells(bits: int, lookup_bits: int):
    """halo2-lib RangeChip::range_check(a, bits): (advice cells, lookup cells)."""
    k = -(-bits // lookup_bits)
    rem = bits % lookup_bits
    adv = 0 if k == 1 else 1 + 3 * (k - 1)
    lk = k
    if rem == 1:
        adv += 4  # assert_bit gate
    elif rem > 1:
        adv += 4  # gate.mul(last_limb, 2^(lookup_bits-rem))
        lk += 1
    return adv, lk


def mul_word_max_bits(limb_bits: int, min_n: int) -> int:
    """bits_size(2 * (min_n*(2^limb_bits-1)^2 + (2^limb_bits-1)))  (is_equal_muled's carry bound)"""
    m = (1 << limb_bits) - 1
    return (2 * (min_n * m * m + m)).bit_length()


@dataclass
class StepCells:
    advice: int
    lookup: int
    # offsets of the segments inside one step's advice block (the K4 kernel's layout)
    seg: dict


def mul_mod_cells(limbs: int, limb_bits: int, lookup_bits: int) -> StepCells:
    """Cells one BigUintChip::mul_mod(a, b, n) emits, all operands `limbs` limbs."""
    L = limbs
    D = 2 * L - 1  # limbs of a product
    seg = {}
    off = 0
    rc_adv, rc_lk = range_check_cells(limb_bits, lookup_bits)
    # 1. assign_integer(q), assign_integer(n), assign_integer(r): L witness cells + L range checks each
    seg["assign"] = off
    off += 3 * (L + L * rc_adv)
    lookups = 3 * L * rc_lk
    # 2. mul(a, b) and mul(q, n): load_zero + truncated mul_no_carry over D limbs
    per_mul = 1 + sum(1 + 3 * (i + 1) for i in range(D))
    seg["mul_ab"] = off
    off += per_mul
    seg["mul_qn"] = off
    off += per_mul
    # 3. qn + r: L gate.add
    seg["add_r"] = off
    off += 4 * L
    # 4. is_equal_muled over D limbs
    cb = mul_word_max_bits(limb_bits, L) - limb_bits
    cb_adv, cb_lk = range_check_cells(cb, lookup_bits)
    per_limb = 4 + 7 + 22 + 4 + 22 + 12 + 4  # sub, sum3, div_mod, add, div_mod, is_equal, and
    seg["eq"] = off
    off += 2 + D * per_limb + (D - 1) * cb_adv + (12 + 4)  # load_zero/one + limbs + carry checks + final carry
    lookups += (D - 1) * cb_lk
    # 5. r < n : big_less_than over L limbs (sub with borrow, one range check per limb)
    lt_adv, lt_lk = range_check_cells(limb_bits, lookup_bits)
    seg["lt"] = off
    off += L * (11 + lt_adv) + 1
    lookups += L * lt_lk
    seg["end"] = off
    return StepCells(off, lookups, seg)


def refresh_aux(limb_bits: int, num_limbs_l: int, num_limbs_r: int):
    """RefreshAux::new(limb_bits, l, r).increased_limbs_vec (paillier.rs:40-44): how many further limbs the maximal
    value of each product limb spills into when it is cut back to limb_bits-wide limbs; its length is the limb count
    of the refreshed integer (l + r for the shapes of this circuit)."""
    mx = (1 << limb_bits) - 1
    d = num_limbs_l + num_limbs_r - 1
    muled = [sum(1 for j in range(num_limbs_l) if 0 <= i - j < num_limbs_r) * mx * mx for i in range(d)]
    inc = []
    cur = 0
    while cur < len(muled):
        chunks = max(1, -(-muled[cur].bit_length() // limb_bits))
        inc.append(chunks - 1)
        val = muled[cur]
        for i in range(chunks):
            piece, val = val & mx, val >> limb_bits
            if cur + i < len(muled):
                muled[cur + i] = piece if i == 0 else muled[cur + i] + piece
            else:
                muled.append(piece)
        cur += 1
    return inc


def assign_cells(num_limbs: int, limb_bits: int, lookup_bits: int):
    """assign_integer: the limbs as witnesses, then one range check each -> (advice, lookup)"""
    a, l = range_check_cells(limb_bits, lookup_bits)
    return num_limbs * (1 + a), num_limbs * l


def square_cells(limbs: int) -> int:
    """square(n) = mul(n, n): load_zero + truncated mul_no_carry over 2 limbs - 1 product limbs"""
    return 1 + sum(1 + 3 * (i + 1) for i in range(2 * limbs - 1))


def refresh_cells(inc, limb_bits: int, lookup_bits: int):
    a, l = range_check_cells(limb_bits, lookup_bits)
    adv = 1 + sum((k + 1) * 22 + k * 4 for k in inc) + len(inc) * a
    return adv, len(inc) * l


def assert_equal_cells(limbs: int) -> int:
    return 2 + 16 * limbs


@dataclass
class CircuitCells:
    advice: int
    lookup: int
    seg: dict   # name -> (advice offset, lookup offset), in emission order


def circuit_cells(kind: str, limbs_n: int, limb_bits: int, lookup_bits: int, n_steps_g: int = 0, n_steps_r: int = 0) -> CircuitCells:
    """The whole cell stream of paillier_enc_test (bench.rs:33-75, kind 'encrypt') or paillier_enc_add_test
    (bench.rs:77-117, kind 'add'), operation by operation in call order -- what pz_circuit_expand_dev writes."""
    Ln, L = limbs_n, 2 * limbs_n
    mm = mul_mod_cells(L, limb_bits, lookup_bits)
    seg, a, l = {}, 0, 0

    def put(name, da, dl=0):
        nonlocal a, l
        seg[name] = (a, l)
        a += da
        l += dl

    for name in ("assign_n", "assign_g", "assign_x", "assign_y"):
        put(name, *assign_cells(Ln, limb_bits, lookup_bits))
    put("square", square_cells(Ln))
    put("refresh", *refresh_cells(refresh_aux(limb_bits, Ln, Ln), limb_bits, lookup_bits))
    put("load_zero", 1)
    if kind == "encrypt":
        put("pow_g", 2 + n_steps_g * mm.advice, n_steps_g * mm.lookup)
        put("pow_r", 2 + n_steps_r * mm.advice, n_steps_r * mm.lookup)
    elif kind == "encrypt_uniform":
        # g^m as pow_mod over Ln * limb_bits in-circuit bits: per limb of m num_to_bits (7 W - 2 cells), per
        # bit two mul_mods and a limb-wise select (8 cells per limb); the step count is fixed by the key size
        m_bits = Ln * limb_bits
        assert n_steps_g in (0, 2 * m_bits)
        put("pow_g", 2 + Ln * (7 * limb_bits - 2) + m_bits * (2 * mm.advice + 8 * L), 2 * m_bits * mm.lookup)
        put("pow_r", 2 + n_steps_r * mm.advice, n_steps_r * mm.lookup)
    put("final", mm.advice, mm.lookup)
    put("assign_res", *assign_cells(L, limb_bits, lookup_bits))
    put("assert_equal", assert_equal_cells(L))
    seg["end"] = (a, l)
    return CircuitCells(a, l, seg)


@dataclass
class ProofShape:
    k: int
    lookup_bits: int
    limbs: int
    n_steps: int
    cells_per_step: int
    lookups_per_step: int
    advice_cols: int
    lookup_cols: int
    perm_cols: int
    msm_witness: int      # commit_lagrange of advice columns (short scalars)
    msm_lookup: int       # commit_lagrange of lookup-advice columns (lookup_bits-bit scalars)
    msm_full: int         # permuted lookup columns, lookup/permutation products, h pieces, openings
    polys: int            # polynomials taken Lagrange -> coeff -> extended coset
    ext_k: int
    advice_cells: int = 0   # the whole circuit's stream (circuit_cells)
    lookup_cells: int = 0


def encrypt_proof_shape(enc_bits: int, k: int, n_steps: int, limb_bits: int = 64, lookup_bits: int | None = None,
                        minimum_rows: int = MINIMUM_ROWS_BENCH, max_degree: int = 4, kind: str = "encrypt", n_steps_g: int | None = None,
                        blinding_factors: int = BLINDING_FACTORS) -> ProofShape:
    """Column / MSM / NTT counts of one encrypt (or add) proof (SURVEY.md section 3.4's table, made concrete).  n_steps
    counts every mul_mod of the circuit (both chains + the final one); the split between the chains only moves the four
    constant cells of pow_mod_fixed_exp and does not change any count."""
    if lookup_bits is None:
        lookup_bits = k - 1  # the reference's pattern: paillier.rs:168-169, bench.rs:162-163
    Ln = enc_bits // limb_bits
    L = 2 * Ln
    sc = mul_mod_cells(L, limb_bits, lookup_bits)
    rb = row_budget(k, minimum_rows, blinding_factors)
    if kind == "encrypt":
        ng = (n_steps - 1) // 2 if n_steps_g is None else n_steps_g
        cc = circuit_cells("encrypt", Ln, limb_bits, lookup_bits, ng, n_steps - 1 - ng)
    elif kind == "encrypt_uniform":   # g^m takes 2 * (Ln * limb_bits) steps whatever the message is
        ng = 2 * Ln * limb_bits
        cc = circuit_cells("encrypt_uniform", Ln, limb_bits, lookup_bits, ng, n_steps - 1 - ng)
    else:
        cc = circuit_cells("add", Ln, limb_bits, lookup_bits)
    A = rb.columns_for(cc.advice)     # (plain cut; the break-point layout of circuit_structure.columns loses 1-3 rows per column)
    Lk = rb.columns_for(cc.lookup)
    P = math.ceil((A + Lk + 1) / (max_degree - 2))
    return ProofShape(k=k, lookup_bits=lookup_bits, limbs=L, n_steps=n_steps, cells_per_step=sc.advice,
                      lookups_per_step=sc.lookup, advice_cols=A, lookup_cols=Lk, perm_cols=P, msm_witness=A,
                      msm_lookup=Lk, msm_full=3 * Lk + P + 1 + (max_degree - 1) + 2, polys=A + 4 * Lk + P,
                      ext_k=k + 2, advice_cells=cc.advice, lookup_cells=cc.lookup)


def break_points(gate_mask, max_rows: int):
    """halo2-lib's break-point column layout of an advice stream (pz.h pz_circuit_break_points; host logic inside the C ABI library,
    no device needed): gate_mask = uint8 array, 1 where a gate window starts.  -> uint64 array of n_cols + 1 stream indices:
    column j holds the cells starts[j] .. starts[j + 1] from row 0 on (the cell it ends with is also row 0 of column j + 1)."""
    import ctypes as C

    import numpy as np

    from . import _lib

    m = np.ascontiguousarray(gate_mask, dtype=np.uint8)
    L = _lib.lib()
    nc = C.c_size_t()
    rc = L.pz_circuit_break_points(C.c_void_p(m.ctypes.data), m.shape[0], max_rows, None, 0, C.byref(nc))
    if rc != 0:
        raise _lib.PzError(rc, "pz_circuit_break_points")
    out = np.zeros(nc.value + 1, dtype=np.uint64)
    rc = L.pz_circuit_break_points(C.c_void_p(m.ctypes.data), m.shape[0], max_rows, C.c_void_p(out.ctypes.data), out.shape[0], C.byref(nc))
    if rc != 0:
        raise _lib.PzError(rc, "pz_circuit_break_points")
    return out
