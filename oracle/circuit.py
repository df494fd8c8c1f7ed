"""TEST INFRASTRUCTURE (oracle): the circuit STRUCTURE halo2-lib's keygen would hand the prover for the reference's drivers
(/root/reference/src/bench.rs:33-117 paillier_enc_test / paillier_enc_add_test with PaillierChip::{encrypt, add},
src/paillier.rs:32-85) -- selector positions, the break-point column layout, the copy constraints, the constants column and the
lookup table -- in COLUMN coordinates, together with the column values the witness must have.

It is derived from oracle/pyref.py's restatement of the dependency's cell patterns (`expand_circuit_cells_wired`,
`gate_mask_circuit`; SURVEY tag [D]: cell-layout parity with the reference's floating dependency versions is unpinned).  The
product never imports this module: the prover (paillier_halo2_amd/prover.py) takes a structure as INPUT, the way the reference's
create_proof takes a ProvingKey built by keygen_vk / keygen_pk (bench.rs:161-175).  tests/ build that input here.

What is restated here, with its source in the dependency:
  * break points -- halo2-lib `assign_with_constraints` (flex_gate/threads/single_phase.rs [D]): walking the Context's cells with a
    row offset r, cell i is assigned at (column, r); if (q[i] && r + 4 > max_rows) || r >= max_rows - 1 the column ends there, the
    cell is assigned AGAIN at row 0 of the next basic-gate column with an equality constraint, and its gate (if q[i]) is enabled
    in the new column;
  * constants -- `assign_raw_constants`: every distinct constant gets one cell of a fixed column (equality-enabled) and every
    advice cell loaded as that constant is constrained equal to it;
  * lookups -- `RangeChip` pushes the cells to look up to `cells_to_lookup`; they are copied in order into lookup-enabled advice
    columns (equality constraint with the source cell); the table column holds 0 .. 2^lookup_bits - 1 and zeros after it;
  * the permutation covers [advice columns | lookup-advice columns | the constants column], in that order.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

from . import pyref as P

R = P.FR_R


def break_points(mask, max_rows: int) -> List[int]:
    """-> starts (n_cols + 1 stream indices); independent restatement of pz_circuit_break_points (compared in tests)"""
    n = len(mask)
    starts = [0]
    s = 0
    while True:
        r = 0
        i = s
        end = None
        while i < n:
            r = i - s
            if (mask[i] and r + 4 > max_rows) or r >= max_rows - 1:
                end = i
                break
            i += 1
        if end is None:
            break
        s = end
        starts.append(s)
    starts.append(n)
    return starts


@dataclass
class Structure:
    k: int
    lookup_bits: int
    max_rows: int
    blinding_factors: int
    starts: List[int]                       # advice break-point layout
    n_adv: int
    n_lk: int
    selectors: np.ndarray                   # uint8 [n_adv][2^k]
    constants: List[int]                    # the constants column, row i
    table: List[int]                        # the lookup table column (2^k rows)
    map_col: np.ndarray                     # uint32 [m][2^k]: the permutation sigma as (column, row) of the image
    map_row: np.ndarray
    equalities: List[Tuple[Tuple[int, int], Tuple[int, int]]]
    adv_cols: List[List[int]]               # expected witness: advice columns (usable rows only, zero above the stream)
    lk_cols: List[List[int]]
    n_cells: int = 0
    n_lookups: int = 0
    satisfied: int = 1
    info: Dict = field(default_factory=dict)

    @property
    def n(self):
        return 1 << self.k

    @property
    def usable(self):
        return self.n - (self.blinding_factors + 1)

    @property
    def m(self):
        return self.n_adv + self.n_lk + 1

    def pos(self, c: int) -> Tuple[int, int]:
        """stream index of an advice cell -> (column, row) (for a shared break cell: row 0 of the later column)"""
        import bisect

        j = bisect.bisect_right(self.starts, c) - 1
        used = self.info.get("n_adv_used", self.n_adv)
        if j >= used:
            j = used - 1
        return j, c - self.starts[j]


def build(kind: str, n: int, g: int, x: int, y: int, res: int, enc_bits: int, limb_bits: int, lb: int, k: int,
          minimum_rows: int = 20, blinding_factors: int = 6, break_rows: int | None = None) -> Structure:
    """structure + expected witness columns of one encrypt / add circuit instance.
    minimum_rows: the argument of halo2-lib's `calculate_params` [D] -- it fixes the NUMBER of advice / lookup-advice columns as
    ceil(cells / (2^k - minimum_rows)): 20 on the reference's bench path (bench_builder, reached from /root/reference/src/bench.rs:161-171),
    9 under MockProver (src/paillier.rs:167-171).  Columns are FILLED to max_rows = 2^k - cs.minimum_rows() = 2^k - (blinding_factors + 3)
    (FlexGateConfig::max_rows [D]; break_rows overrides), so configured columns beyond the ones the cells fill stay empty."""
    N = 1 << k
    usable = N - (blinding_factors + 1)
    max_rows = N - (blinding_factors + 3) if break_rows is None else break_rows
    count_rows = N - minimum_rows
    assert max_rows <= usable and count_rows > 0
    W = P.expand_circuit_cells_wired(kind, n, g, x, y, res, enc_bits, limb_bits, lb, full=True)
    adv = W["advice"]
    if kind == "encrypt":
        ng = x.bit_length() + bin(x).count("1")
        nr = n.bit_length() + bin(n).count("1")
    elif kind == "encrypt_uniform":
        ng = 2 * enc_bits
        nr = n.bit_length() + bin(n).count("1")
    else:
        ng = nr = 0
    if kind == "encrypt_uniform":
        gl, total = P.gate_offsets_uniform_circuit(enc_bits, limb_bits, lb, nr)
        mask = np.zeros(total, dtype=np.uint8)
        mask[np.asarray(gl, dtype=np.int64)] = 1
    else:
        mask, total = P.gate_mask_circuit(kind, enc_bits, limb_bits, lb, ng, nr)
    assert total == len(adv)
    starts = break_points(mask, max_rows)
    A_used = len(starts) - 1
    A = max(A_used, -(-len(adv) // count_rows))     # configured columns (calculate_params); the ones past A_used stay empty
    lk_src = W["lookup_src"]
    Lk = max(-(-len(lk_src) // max_rows), -(-len(lk_src) // count_rows))
    m = A + Lk + 1
    st = Structure(k=k, lookup_bits=lb, max_rows=max_rows, blinding_factors=blinding_factors, starts=starts, n_adv=A, n_lk=Lk,
                   selectors=np.zeros((A, N), dtype=np.uint8), constants=[], table=[], map_col=None, map_row=None, equalities=[],
                   adv_cols=[], lk_cols=[], n_cells=len(adv), n_lookups=len(lk_src), satisfied=W["satisfied"])
    st.info["n_adv_used"], st.info["minimum_rows"] = A_used, minimum_rows
    st.starts = starts + [len(adv)] * (A - A_used)      # n_adv + 1 entries: an empty configured column starts (and ends) at the stream's end
    # columns and selectors
    for j in range(A):
        if j >= A_used:
            st.adv_cols.append([0] * N)
            continue
        lo, hi = starts[j], starts[j + 1]
        last = hi if j + 1 < A_used else hi - 1     # a non-final column also holds the cell the next one starts with
        col = adv[lo:last + 1]
        assert len(col) <= max_rows, (j, len(col))
        st.adv_cols.append(col + [0] * (N - len(col)))
        st.selectors[j, : hi - lo] = mask[lo:hi]    # the shared last cell's gate is enabled in the NEXT column
        if j + 1 < A_used:
            st.equalities.append(((j + 1, 0), (j, hi - lo)))
    lk_vals = [adv[c] for c in lk_src]
    for j in range(Lk):
        col = lk_vals[j * max_rows:(j + 1) * max_rows]
        st.lk_cols.append(col + [0] * (N - len(col)))
    # constants column
    const_row: Dict[int, int] = {}
    for idx, v in W["consts"]:
        if v not in const_row:
            const_row[v] = len(st.constants)
            st.constants.append(v)
    assert len(st.constants) <= max_rows
    for idx, v in W["consts"]:
        st.equalities.append((st.pos(idx), (A + Lk, const_row[v])))
    for src, cp in W["pairs"]:
        st.equalities.append((st.pos(src), st.pos(cp)))
    for t, c in enumerate(lk_src):
        st.equalities.append((st.pos(c), (A + t // max_rows, t % max_rows)))
    # the circuit's output bit is constrained to the constant 1 (assert_equal_fresh, bench.rs:74): only an honest witness satisfies it
    st.info["result_cell"] = st.pos(W["result_cell"])
    st.equalities.append((st.pos(W["result_cell"]), (A + Lk, const_row[1])))
    st.table = [i if i < (1 << lb) else 0 for i in range(N)]
    st.map_col, st.map_row = permutation_from_equalities(st.equalities, m, N)
    return st


def permutation_from_equalities(eqs, m: int, N: int):
    """equality constraints -> a permutation whose cycles are the equivalence classes (halo2's Assembly merges cycles pair by pair;
    any permutation with these cycles proves the same statement -- the ORDER inside a cycle changes sigma and the vk, not validity).
    -> (map_col, map_row) uint32 [m][N], identity outside the classes"""
    parent: Dict[Tuple[int, int], Tuple[int, int]] = {}

    def find(a):
        root = a
        while parent.get(root, root) != root:
            root = parent[root]
        while parent.get(a, a) != root:
            parent[a], a = root, parent[a]
        return root

    for a, b in eqs:
        ra, rb = find(a), find(b)
        parent.setdefault(ra, ra)
        parent.setdefault(rb, rb)
        if ra != rb:
            parent[rb] = ra
    classes: Dict[Tuple[int, int], List[Tuple[int, int]]] = {}
    for cell in list(parent.keys()):
        classes.setdefault(find(cell), []).append(cell)
    map_col = np.repeat(np.arange(m, dtype=np.uint32), N).reshape(m, N)
    map_row = np.tile(np.arange(N, dtype=np.uint32), m).reshape(m, N)
    for cells in classes.values():
        cells.sort()
        for a, b in zip(cells, cells[1:] + cells[:1]):
            map_col[a[0], a[1]] = b[0]
            map_row[a[0], a[1]] = b[1]
    return map_col, map_row


def perm_columns(st: Structure) -> List[List[int]]:
    """the permuted columns' values in permutation order: advice, lookup advice, constants"""
    N = st.n
    return st.adv_cols + st.lk_cols + [st.constants + [0] * (N - len(st.constants))]


def mock_prover(st: Structure, cols: List[List[int]] | None = None) -> List[str]:
    """MockProver analogue in column form (the reference's tests call expect_satisfied(true): src/paillier.rs:167-171): every enabled
    gate inside its column, every equality constraint, every lookup cell in the table.  -> list of failures (empty = satisfied)"""
    if cols is None:
        cols = perm_columns(st)
    bad: List[str] = []
    A, Lk = st.n_adv, st.n_lk
    for j in range(A):
        rows = np.nonzero(st.selectors[j])[0]
        c = cols[j]
        for r in rows.tolist():
            if (c[r] + c[r + 1] * c[r + 2] - c[r + 3]) % R != 0:
                bad.append("gate %d:%d" % (j, r))
    for (ca, ra), (cb, rb) in st.equalities:
        if cols[ca][ra] != cols[cb][rb]:
            bad.append("copy %d:%d != %d:%d" % (ca, ra, cb, rb))
    tab = set(st.table)
    for j in range(Lk):
        for r, v in enumerate(cols[A + j][: st.usable]):
            if v not in tab:
                bad.append("lookup %d:%d" % (j, r))
    return bad
