"""The prover's phases composed in create_proof's order, on the device, over the C ABI (include/pz.h) only.

The reference reaches halo2-axiom's `keygen_vk` / `keygen_pk` / `create_proof` through halo2-lib's tester
(/root/reference/src/bench.rs:161-171: `bench_builder` -> keygen -> `gen_proof`); its circuit is halo2-lib's: one vertical gate
q (a + b c - d) per basic-gate advice column, range-check lookups of lookup-enabled advice columns against one table column, a
permutation over [advice | lookup advice | constants].  This module runs that flow as ONE dataflow -- every
later phase consumes the earlier phases' outputs, nothing is synthetic:

    keygen:        fixed columns (selectors, constants, table) and the sigma polynomials of the circuit's copy constraints:
                   commitments, coefficient forms, extended-coset forms, all resident in HBM (pz_permutation_sigma_dev,
                   pz_keygen_columns_dev)
    create_proof:  1 advice + lookup-advice commitments (blinding rows filled)                        K1 pz_msm_g1_dev
                   2 permute_expression_pair -> A', S' commitments                                    pz_lookup_permute_dev
                   3 permutation products Z_j, lookup products Z commitments                          pz_permutation_product_sets_dev ...
                   4 vanishing argument's random polynomial commitment
                   5 every polynomial Lagrange -> coefficients (K2), then TILES of columns -> extended coset -> custom-gate and
                     permutation lines of evaluate_h as each tile is produced; lookup lines; division by X^n - 1; extended -> coeff;
                     h pieces committed                                                               pz_ntt_fr_*  pz_quotient_*
                   6 evaluations at x and its rotations                                                pz_poly_eval_multi_dev
                   7 SHPLONK over the real coefficient forms                                           pz_shplonk_*

What stays outside: the transcript (challenges are INPUTS here: the caller hashes the commitments each phase
hands back -- that is the host round trip between phases), the verifier / G2 side, and the circuit STRUCTURE itself (selector
positions, copy constraints, break points: the dependency's keygen knows them; they are an input, `CircuitStructure`).

Memory plan: the proving key's extended forms are RESIDENT (at config c2: 3033 selectors + 3118 sigma
columns x 2^19 x 32 B = 103 GB of the 288 GB on halo2's 4n-point domain, 77 GB on the three cosets used here); the proof's own columns are extended tile by tile (`tile` columns at a time) and never
exist on the extended domain all at once -- only the grand products Z (which the chaining lines read across sets) do.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import consts
from .engine import Bases, Engine

FR = consts.FR_R
M = consts.fr_mont_limbs

LOG_E = 2                      # cs.degree() = 4 for halo2-lib's gate + lookup -> quotient on the 4n coset, 3 pieces
CHUNK = 2                      # permutation columns per grand product: degree - 2
ZETA = pow(consts.FR_GENERATOR, (FR - 1) // 3, FR)          # the coset generator halo2's EvaluationDomain uses (g_coset)
DELTA = pow(consts.FR_GENERATOR, 1 << consts.FR_S, FR)      # halo2curves Fr::DELTA


@dataclass
class CircuitStructure:
    """what keygen derives from the circuit (INPUT; in the reference the dependency's synthesize / Assembly produce it)"""
    k: int
    lookup_bits: int
    max_rows: int                  # rows a column is filled to (halo2-lib: FlexGateConfig::max_rows = 2^k - cs.minimum_rows(); layout.RowBudget)
    blinding_factors: int          # cs.blinding_factors(): 6 for halo2-lib's 4-rotation gate
    selectors: np.ndarray          # uint8 [n_adv][2^k]  (selectors / map_col / map_row: numpy, or torch tensors already on the device)
    n_lk: int
    constants: Sequence[int]       # the constants fixed column (row i), canonical integers
    map_col: np.ndarray            # uint32 [m][2^k]: sigma as the (column, row) every cell maps to; m = n_adv + n_lk + 1
    map_row: np.ndarray
    table: Optional[Sequence[int]] = None      # default: 0 .. 2^lookup_bits - 1, then zeros
    minimum_rows: Optional[int] = None         # the calculate_params argument the column COUNT came from (layout.RowBudget; informational)
    n_adv_used: Optional[int] = None           # advice columns the cells fill (<= n_adv: K4's break-point table has n_adv_used + 1 entries)

    @property
    def n_adv(self) -> int:
        return int(self.selectors.shape[0])

    @property
    def m(self) -> int:
        return self.n_adv + self.n_lk + 1


def _torch():
    import torch

    return torch


def _zeros(*shape):
    torch = _torch()
    return torch.zeros(shape, dtype=torch.int64, device="cuda")


def _zeros_cap(count: int, *shape, q: int = 64):
    """[count, *shape] zeros inside an allocation of ceil(count / q) * q rows: keys of nearly equal column counts (a new message of the
    reference's circuit moves the advice column count by a few) then ask torch's caching allocator for IDENTICAL block sizes, so the
    next key's 100-GB tensors reuse the previous key's blocks instead of fragmenting them (a re-malloc of the key costs 4-5 s)"""
    return _zeros(-(-count // q) * q, *shape)[:count]


def _ints_to_dev_mont(eng: Engine, vals: Sequence[int]):
    """canonical integers -> Montgomery elements on the device (upload of raw limbs + pz_fr_convert_dev)"""
    torch = _torch()
    n = len(vals)
    raw = np.zeros((n, 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        v %= FR
        if v:
            for j in range(4):
                raw[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    t = torch.from_numpy(raw.view(np.int64)).cuda()
    eng.fr_convert_dev(t.data_ptr(), n, True)
    return t


_R3 = FR >> 192                  # the top 64-bit word of r


def _random_fr(gen, *shape):
    """blinding values: UNIFORM field elements as Montgomery representatives, by rejection from 254-bit candidates WITHOUT a host round trip
    (a synchronising test in the middle of a phase would stall the launch queue): eight candidates per element, the first whose top word
    is below r's is taken (that drops a 2^-62 fraction of the field -- the values whose top word EQUALS r's); an element whose eight
    candidates all fail (probability 0.244^8 = 1.3e-5) falls back to a 252-bit value"""
    torch = _torch()
    K = 8
    c = torch.randint(-(1 << 63), (1 << 63) - 1, (*shape, K, 4), dtype=torch.int64, device="cuda", generator=gen)
    c[..., 3] &= 0x3FFFFFFFFFFFFFFF
    ok = c[..., 3] < _R3                                            # [..., K]
    first = ok.to(torch.int8).argmax(dim=-1, keepdim=True)          # the first valid candidate (0 if none)
    t = torch.gather(c, -2, first.unsqueeze(-1).expand(*shape, 1, 4)).squeeze(-2)
    none = ~ok.any(dim=-1)
    t[..., 3] = torch.where(none, t[..., 3] & 0x0FFFFFFFFFFFFFFF, t[..., 3])
    return t


@dataclass
class Domain:
    k: int
    bf: int
    cosets: int = 3

    def __post_init__(self):
        self.n = 1 << self.k
        self.N = self.n << LOG_E
        self.E = 1 << LOG_E
        self.usable = self.n - (self.bf + 1)
        self.omega = consts.fr_omega(self.k)
        self.omega_inv = pow(self.omega, -1, FR)
        self.n_inv = pow(self.n, -1, FR)
        self.omega_ext = consts.fr_omega(self.k + LOG_E)
        self.omega_ext_inv = pow(self.omega_ext, -1, FR)
        self.N_inv = pow(self.N, -1, FR)
        self.coset_g = ZETA
        self.gens = np.stack([M(self.coset_g * pow(self.omega_ext, r, FR) % FR) for r in range(self.E)])
        self.parts = self._parts(self.cosets)

    def _parts(self, cosets: int):
        """the points the quotient is evaluated on.  halo2 takes the whole extended coset g <w_4n> (4n points); the quotient has degree
        below 3n, so THREE of its four cosets of <w_n> determine it: part A = g <w_2n> (the cosets r = 0 and r = 2 interleaved: a coset
        of the 2n-th roots), part B = g w_4n <w_n> (r = 1).  Every kernel of evaluate_h is generic in (log of the domain, rows per
        step, coset generator, domain generator), so a part is just another call; what changes is the way back to coefficients
        (create_proof, "the quotient from three cosets").  cosets = 4: halo2's own domain as one part (kept for the A/B and the tests)."""
        w4, g = self.omega_ext, self.coset_g
        mk = lambda log_e, cg, om: dict(log_e=log_e, E=1 << log_e, size=self.n << log_e, coset_g=cg, omega=om, omega_inv=pow(om, -1, FR),
                                        size_inv=pow(self.n << log_e, -1, FR),
                                        gens=np.stack([M(cg * pow(om, r, FR) % FR) for r in range(1 << log_e)]))
        if cosets == 4:
            return [mk(2, g, w4)]
        assert cosets == 3
        return [mk(1, g, w4 * w4 % FR), mk(0, g * w4 % FR, pow(w4, 4, FR))]


class ProvingKey:
    """fixed + permutation polynomials in all three forms, resident in HBM; the SRS tables.
    ext_resident_cols: how many of the permuted columns' EXTENDED key forms (selector j and sigma j for j < R) stay resident.  None = all of
    them (halo2's ProvingKey: fixed_cosets / permutation.cosets; at config c2 77 GB); an integer R = the STREAMED proving key: only the
    coefficient forms of the rest are kept and create_proof re-extends them per tile beside the advice tile they are folded with (R = 0
    at BASELINE config c5, where the extended key would be 239 GB: DESIGN.md section 6.3).  The proof is byte for byte the same: the tile's
    values come from the same transform either way.  The lookup table's extended form (one column) is always resident."""

    def __init__(self, eng: Engine, st: CircuitStructure, bases_lagrange: Bases, bases_monomial: Bases, cosets: int = 3,
                 ext_resident_cols: Optional[int] = None):
        torch = _torch()
        self.eng, self.st = eng, st
        self.dom = d = Domain(st.k, st.blinding_factors, cosets)
        self.bases_lagrange, self.bases_monomial = bases_lagrange, bases_monomial
        n, N, A, m = d.n, d.N, st.n_adv, st.m
        assert st.max_rows <= d.usable and tuple(st.map_col.shape) == (m, n) and tuple(st.selectors.shape) == (A, n)
        self.n_sets = -(-m // CHUNK)
        one = torch.from_numpy(M(1).view(np.int64)).cuda()
        zero4 = torch.zeros_like(one)
        GB = 256                                   # columns per keygen call (bounds the library's MSM / NTT workspaces)
        # ---- fixed columns: [selectors | constants | table], Lagrange form
        F = A + 2
        fixed = _zeros_cap(F, n, 4)
        for a0 in range(0, A, GB):
            sel = st.selectors[a0:a0 + GB]          # numpy, or a tensor already on the device (circuit_structure.columns(keep_on_device=True))
            sel = sel.cuda() if isinstance(sel, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(sel)).cuda()
            fixed[a0:a0 + sel.shape[0]] = torch.where(sel.bool().unsqueeze(-1), one, zero4)
            del sel
        consts_col = list(st.constants) + [0] * (n - len(st.constants))
        fixed[A] = _ints_to_dev_mont(eng, consts_col)
        table = list(st.table) if st.table is not None else [i if i < (1 << st.lookup_bits) else 0 for i in range(n)]
        fixed[A + 1] = _ints_to_dev_mont(eng, table)
        self.const_lagrange = fixed[A].clone()
        self.table_lagrange = fixed[A + 1].clone()
        # ---- sigma polynomials from the copy-constraint map
        # (one call over all m columns: the images' column indices run over the whole permutation, and the entry point sizes its
        # table of delta powers by the m it is given)
        sigma = _zeros_cap(m, n, 4)
        as_dev = lambda a: a.to(torch.int32).cuda().contiguous() if isinstance(a, torch.Tensor) else torch.from_numpy(a.view(np.int32)).cuda()
        d_mc, d_mr = as_dev(st.map_col), as_dev(st.map_row)
        eng.permutation_sigma_dev(d_mc.data_ptr(), d_mr.data_ptr(), m, st.k, M(d.omega), M(DELTA), sigma.data_ptr(), 4 * n)
        eng.sync()
        del d_mc, d_mr
        self.sigma_lagrange = _zeros_cap(m, n, 4).copy_(sigma)
        # ---- keygen_vk + keygen_pk: commitments, coefficient forms (in place), extended forms
        self.fixed_commit = _zeros(F, 12)
        self.sigma_commit = _zeros(m, 12)
        P0 = d.parts[0]
        self.streamed = ext_resident_cols is not None and ext_resident_cols < m
        R = m if not self.streamed else max(0, int(ext_resident_cols))
        self.ext_resident = (min(R, A), R)                                     # resident extended columns of (selectors, sigma)
        Rf, Rs = self.ext_resident
        if not self.streamed:
            Rf = F                                                             # (resident mode keeps the constants and table columns in the same tensor)
        self.fixed_ext = [_zeros_cap(Rf, pt["size"], 4) for pt in d.parts]     # per part of the quotient's domain (Domain._parts)
        self.sigma_ext = [_zeros_cap(Rs, pt["size"], 4) for pt in d.parts]
        for t_, cnt_all, com, ext, Rx in ((fixed, F, self.fixed_commit, self.fixed_ext, Rf), (sigma, m, self.sigma_commit, self.sigma_ext, Rs)):
            for c0 in range(0, cnt_all, GB):
                cnt = min(GB, cnt_all - c0)
                # (columns, whether their extended forms are kept): a batch that straddles the resident prefix runs as two calls
                r_here = max(0, min(cnt, Rx - c0))
                for b0, bc, keep in ((c0, r_here, True), (c0 + r_here, cnt - r_here, False)):
                    if not bc:
                        continue
                    eng.keygen_columns_dev(bases_lagrange, t_[b0].data_ptr(), bc, 4 * n, st.k, P0["log_e"], M(d.omega), M(d.omega_inv), M(d.n_inv),
                                           P0["gens"], com[b0].data_ptr(), ext[0][b0].data_ptr() if keep else None, 4 * P0["size"])
                    for pi in range(1, len(d.parts) if keep else 0):     # the further parts from the coefficient form keygen_columns_dev left in place
                        pt = d.parts[pi]
                        eng.ntt_extend_dev(t_[b0].data_ptr(), bc, 4 * n, ext[pi][b0].data_ptr(), 4 * pt["size"], st.k, pt["log_e"], M(d.omega),
                                           pt["gens"], None)
        self.fixed_coeff = fixed
        self.sigma_coeff = sigma
        # the lookup table on the quotient's domain: one column, always resident (every lookup tile of evaluate_h reads it)
        self.table_ext = [_zeros(1, pt["size"], 4) for pt in d.parts]
        for pi, pt in enumerate(d.parts):
            eng.ntt_extend_dev(fixed[A + 1].data_ptr(), 1, 4 * n, self.table_ext[pi].data_ptr(), 4 * pt["size"], st.k, pt["log_e"], M(d.omega), pt["gens"], None)
        # ---- l_0, l_last, l_active on the extended coset
        u = d.usable
        lrows = _zeros(3, n, 4)
        lrows[0, 0] = one
        lrows[1, u] = one
        lrows[2, :u] = one
        eng.ntt_dev(lrows.data_ptr(), 3, 4 * n, M(d.omega_inv), st.k, None, M(d.n_inv))
        self.l_ext = [_zeros(3, pt["size"], 4) for pt in d.parts]
        for pi, pt in enumerate(d.parts):
            eng.ntt_extend_dev(lrows.data_ptr(), 3, 4 * n, self.l_ext[pi].data_ptr(), 4 * pt["size"], st.k, pt["log_e"], M(d.omega), pt["gens"], None)
        eng.sync()

    def vk_commitments(self) -> Dict[str, np.ndarray]:
        e = self.eng
        e.sync()
        jac = lambda t: e.g1_normalize(t.cpu().numpy().view(np.uint64))
        return {"fixed": jac(self.fixed_commit), "sigma": jac(self.sigma_commit)}


def keygen(eng: Engine, st: CircuitStructure, bases_lagrange: Bases, bases_monomial: Bases, cosets: int = 3,
           ext_resident_cols: Optional[int] = None) -> ProvingKey:
    return ProvingKey(eng, st, bases_lagrange, bases_monomial, cosets, ext_resident_cols)


@dataclass
class Challenges:
    """the transcript's outputs (canonical integers): theta is unused by single-expression lookups but drawn all the same"""
    theta: int
    beta: int
    gamma: int
    y: int
    x: int
    sh_y: int      # SHPLONK's y, v, u
    sh_v: int
    sh_u: int
    transcript_seed: Optional[bytes] = None      # HashTranscript: what a verifier needs to replay the transcript; None = fixed challenges


class Transcript:
    """fixed challenges (tests): nothing is hashed, the phases still hand their commitments over"""

    def __init__(self, ch: Challenges):
        self.ch = ch

    def absorb_points(self, eng: Engine, *jac_tensors):
        """-> the affine forms (one array per tensor): the phase's synchronising download; create_proof keeps them for the proof"""
        return [eng.g1_normalize(t.cpu().numpy().view(np.uint64)) for t in jac_tensors]

    def absorb_affine(self, *host_arrays):
        pass

    def absorb_scalars(self, *host_arrays):
        pass

    def squeeze(self, name: str) -> int:
        return getattr(self.ch, name)


class HashTranscript(Transcript):
    """the drivers' Fiat-Shamir transcript: halo2's Blake2b transcript [D] (halo2_proofs transcript/blake2b.rs) in its PRIMITIVES -- BLAKE2b-512
    personalised "Halo2-Transcript", one domain byte in front of every item (0 challenge, 1 point, 2 scalar), a challenge = the digest of a
    clone of the running state read as a 512-bit little-endian integer mod r -- and in its DATAFLOW: every phase's commitments are brought to
    the host in affine form (a synchronising download: the prover cannot start the next phase before the challenge exists).  Not halo2's
    byte format: a field element enters as the 4 Montgomery words it crosses include/pz.h in (x then y for a point), families in this
    prover's order, and `seed` stands where halo2 absorbs the verifying key's digest.  host/transcript.hpp is the same function (checked
    value for value by tests/test_cpp_host_field.py); oracle/verifier.py::replay_challenges re-derives every challenge from a proof."""

    PERSONAL = b"Halo2-Transcript"

    def __init__(self, seed: bytes = b"pz"):
        import hashlib

        self.seed = bytes(seed)
        self.h = hashlib.blake2b(self.seed, digest_size=64, person=self.PERSONAL)
        self.drawn: Dict[str, int] = {}

    def _items(self, tag: int, a: np.ndarray, words: int):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, words)
        buf = np.empty((a.shape[0], 1 + 8 * words), dtype=np.uint8)
        buf[:, 0] = tag
        buf[:, 1:] = a.view(np.uint8).reshape(a.shape[0], 8 * words)
        self.h.update(buf.tobytes())

    def absorb_affine(self, *host_arrays):
        """affine points already on the host: (count, 8) Montgomery words each"""
        for a in host_arrays:
            self._items(1, a, 8)

    def absorb_points(self, eng: Engine, *jac_tensors):
        out = [eng.g1_normalize(t.cpu().numpy().view(np.uint64)) for t in jac_tensors]
        self.absorb_affine(*out)
        return out

    def absorb_scalars(self, *host_arrays):
        for a in host_arrays:
            self._items(2, a, 4)

    def squeeze(self, name: str) -> int:
        self.h.update(b"\0")
        v = int.from_bytes(self.h.digest(), "little") % FR          # (digest() leaves the running state as it is: halo2 hashes a clone)
        self.drawn[name] = v
        return v

    def challenges(self) -> Challenges:
        return Challenges(**self.drawn, transcript_seed=self.seed)


@dataclass
class Proof:
    commitments: Dict[str, np.ndarray] = field(default_factory=dict)     # name -> affine points (count, 8), Montgomery
    evals: Dict[str, np.ndarray] = field(default_factory=dict)           # name -> (count, n_points, 4), Montgomery
    h_degree_ok: bool = False
    h_top: Optional[np.ndarray] = None                                   # coefficients 3n-3 .. 4n-1 of the quotient (must be zero)


def rotation_points(dom: Domain, x: int) -> List[int]:
    """the six points of halo2-lib circuits' queries: x, wx, w^2 x, w^3 x (the gate's rotations), w^-(bf+1) x (x_last), w^-1 x"""
    w = dom.omega
    return [x % FR, x * w % FR, x * pow(w, 2, FR) % FR, x * pow(w, 3, FR) % FR, x * pow(w, -(dom.bf + 1), FR) % FR, x * pow(w, -1, FR) % FR]


def query_layout(A: int, Lk: int, m: int, S: int):
    """the multi-point opening's rotation sets: [(point indices into rotation_points, [(polynomial family, index), ...])], in the order
    create_proof hands them to SHPLONK (a verifier must fold commitments and evaluations in the same order)"""
    F = A + 2
    sets = [([0], [("lookup_advice", i) for i in range(Lk)] + [("fixed", i) for i in range(F)] + [("sigma", i) for i in range(m)]
             + [("perm_tables", i) for i in range(Lk)] + [("h", 0), ("random", 0)]),
            ([0, 1, 2, 3], [("advice", i) for i in range(A)])]
    if S > 1:
        sets.append(([0, 1, 4], [("perm_z", i) for i in range(S - 1)]))
    sets.append(([0, 1], [("perm_z", S - 1)] + [("lookup_z", i) for i in range(Lk)]))
    sets.append(([0, 5], [("perm_inputs", i) for i in range(Lk)]))
    return sets


class Workspace:
    """the per-proof device buffers of create_proof, allocated once per proving key and reused by every proof (at config c2: the
    grand products on the extended domain are 26 GB; everything else is a few GB)"""

    def __init__(self, pk: ProvingKey, tile: int = 64, lookup_tile: Optional[int] = None):
        d, st = pk.dom, pk.st
        n, N, Lk, S = d.n, d.N, st.n_lk, pk.n_sets
        assert tile % CHUNK == 0
        self.tile = tile
        self.Ap, self.Sp, self.Zl = _zeros_cap(Lk, n, 4, q=8), _zeros_cap(Lk, n, 4, q=8), _zeros_cap(Lk, n, 4, q=8)
        self.Z = _zeros_cap(S, n, 4)
        lt = min(tile, Lk, lookup_tile or tile)
        # the grand products on the quotient's domain: ALL sets of one part at once (the chaining lines read z_{j-1} beside z_j); the parts
        # are worked one after the other, so ONE buffer of the largest part serves them all (at c2: 13 GB instead of 20)
        big = max(pt["size"] for pt in d.parts)
        self._z_ext_flat = _zeros_cap(S, big, 4).view(-1)
        self.z_ext = [self._z_ext_flat[: S * pt["size"] * 4].view(S, pt["size"], 4) for pt in d.parts]
        self.ext = [_zeros(tile, pt["size"], 4) for pt in d.parts]
        # streamed proving key: the tile of selectors and the tile of sigma columns, re-extended beside the advice tile (one buffer each)
        if pk.streamed:
            self._key_tiles = [_zeros(tile, big, 4).view(-1) for _ in range(2)]
            self.key_ext = [[kt[: tile * pt["size"] * 4].view(tile, pt["size"], 4) for kt in self._key_tiles] for pt in d.parts]
        self.lk_ext = [[_zeros(lt, pt["size"], 4) for _ in range(4)] for pt in d.parts]
        self.hh = [_zeros(2, pt["size"], 4) for pt in d.parts]
        self.hp = [_zeros(pt["size"], 4) for pt in d.parts]
        self.h = _zeros(N, 4)                     # the quotient's coefficients: pieces h_0, h_1, h_2 (and the vanishing h_3 with 4 cosets)
        self.tmp = _zeros(2, n, 4)
        self.rnd = _zeros(1, n, 4)
        self.hcomb = _zeros(n, 4)
        self.w1, self.w2 = _zeros(n, 4), _zeros(n, 4)


def create_proof(pk: ProvingKey, cols, tr, seed: Optional[int] = 0, tile: int = 64, hooks=None, ws: Optional[Workspace] = None,
                 timings: Optional[Dict[str, float]] = None, eng: Optional[Engine] = None) -> Proof:
    """cols: int64 CUDA tensor [m][2^k][4]: the advice columns then the lookup-advice columns as K4 wrote them (rows >= max_rows
    zero); the last column (constants) and the blinding rows are filled here; cols is consumed (it ends up in coefficient form).
    tr: a Transcript (or plain Challenges).  Runs on the engine's stream (bind_torch_stream).
    seed: of the device generator the blinding values come from (tests and benches want reproducible proofs); None = 64 bits of OS
    randomness per proof (os.urandom) -- what a caller who needs the proof to be zero-knowledge passes.
    hooks: optional dict of callables name -> f(tensors) applied to intermediate device buffers (the tests' tamper points).
    timings: if given, phase -> milliseconds of wall time (each phase ends with the transcript's synchronising download).
    eng: the context (stream, library workspaces) this proof runs on -- default the key's; a second proof in flight beside this one
    takes its own (bench_connected.run_in_flight); hooks["on_phase"](name) is called as each phase ends (there: the stage hand-over)."""
    import time

    torch = _torch()
    eng, st, d = eng or pk.eng, pk.st, pk.dom
    n, N, k, u, bf = d.n, d.N, d.k, d.usable, d.bf
    A, Lk, m, S = st.n_adv, st.n_lk, st.m, pk.n_sets
    W = A + Lk
    if isinstance(tr, Challenges):
        tr = Transcript(tr)
    if ws is None:
        ws = Workspace(pk, tile)
    tile = ws.tile
    assert tuple(cols.shape) == (m, n, 4)
    hooks = hooks or {}
    gen = torch.Generator(device="cuda")
    if seed is None:
        import os

        seed = int.from_bytes(os.urandom(8), "little") >> 1
    gen.manual_seed(seed)
    pr = Proof()
    bl, bm = pk.bases_lagrange, pk.bases_monomial
    t_last = [time.perf_counter()]

    def phase(name):
        if "on_phase" in hooks:
            hooks["on_phase"](name)
        if timings is not None:
            eng.sync()
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now

    eng2, split_cols = hooks.get("commit_engine"), int(hooks.get("commit_split", 512))

    def commit(bases, t, count, stride_u64):
        out = _zeros(count, 12)
        if eng2 is None or count < 2 * split_cols:
            eng.msm_dev(bases, t.data_ptr(), count, n, stride_u64, out.data_ptr())
            return out
        # EXPERIMENT (hooks["commit_engine"]; measured and not adopted): column groups alternate between two contexts, so
        # that one group's memory-bound sort / latency-bound tree can run beside the other's multiplier-bound accumulation
        eng2.wait_for(eng)
        for i, c0 in enumerate(range(0, count, split_cols)):
            (eng if i % 2 == 0 else eng2).msm_dev(bases, t[c0].data_ptr(), min(split_cols, count - c0), n, stride_u64, out[c0].data_ptr())
        eng.wait_for(eng2)
        return out

    def to_coeff(t, cnt):       # lagrange_to_coeff in place, `tile` columns per call (bounds the transform workspace)
        for c0 in range(0, cnt, tile):
            eng.ntt_dev(t[c0].data_ptr(), min(tile, cnt - c0), 4 * n, M(d.omega_inv), k, None, M(d.n_inv))

    def extend(src, cnt, dst, pt):
        eng.ntt_extend_dev(src.data_ptr(), cnt, 4 * n, dst.data_ptr(), 4 * pt["size"], k, pt["log_e"], M(d.omega), pt["gens"], None)

    # ---- 1. advice: blinding rows, commitments -> theta
    cols[:W, u:] = _random_fr(gen, W, n - u)
    cols[W] = pk.const_lagrange
    if "advice" in hooks:
        hooks["advice"](cols)
    c_adv = commit(bl, cols, W, 4 * n)
    if "after_advice_launch" in hooks:      # the caller's chance to queue independent work (the NEXT proof's witness on another context)
        hooks["after_advice_launch"]()      # while this proof's largest commitment batch runs and before the host waits for it
    (a_adv,) = tr.absorb_points(eng, c_adv)
    tr.squeeze("theta")
    phase("advice_commit")
    # ---- 2. lookups: permuted input / table (one expression each side: theta does not enter) -> beta, gamma
    Ap, Sp, Zl, Z = ws.Ap, ws.Sp, ws.Zl, ws.Z
    lk_in = cols[A:W]
    eng.lookup_permute_dev(lk_in.data_ptr(), Lk, 4 * n, pk.table_lagrange.data_ptr(), u, st.lookup_bits, Ap.data_ptr(), Sp.data_ptr(), 4 * n)
    if "permuted" in hooks:
        hooks["permuted"](Ap, Sp)
    Ap[:, u:] = _random_fr(gen, Lk, n - u)
    Sp[:, u:] = _random_fr(gen, Lk, n - u)
    c_ap, c_sp = commit(bl, Ap, Lk, 4 * n), commit(bl, Sp, Lk, 4 * n)
    a_ap, a_sp = tr.absorb_points(eng, c_ap, c_sp)
    beta_i, gamma_i = tr.squeeze("beta"), tr.squeeze("gamma")
    beta, gamma = M(beta_i), M(gamma_i)
    phase("lookup_permute_commit")
    # ---- 3. grand products, the vanishing argument's random polynomial -> y
    eng.permutation_product_sets_dev(cols.data_ptr(), 4 * n, pk.sigma_lagrange.data_ptr(), 4 * n, m, CHUNK, k, u, M(d.omega), beta, gamma,
                                     M(DELTA), Z.data_ptr(), 4 * n)
    Z[:, u + 1:] = _random_fr(gen, S, n - u - 1)
    eng.lookup_product_dev(lk_in.data_ptr(), 4 * n, pk.table_lagrange.data_ptr(), Ap.data_ptr(), 4 * n, Sp.data_ptr(), 4 * n, Lk, n, beta,
                           gamma, M(1), Zl.data_ptr(), 4 * n)
    Zl[:, u + 1:] = _random_fr(gen, Lk, n - u - 1)
    if "products" in hooks:
        hooks["products"](Z, Zl)
    c_z, c_zl = commit(bl, Z, S, 4 * n), commit(bl, Zl, Lk, 4 * n)
    rnd = ws.rnd
    rnd.copy_(_random_fr(gen, 1, n))
    c_rnd = commit(bm, rnd, 1, 4 * n)
    a_z, a_zl, a_rnd = tr.absorb_points(eng, c_z, c_zl, c_rnd)
    y_i = tr.squeeze("y")
    y = M(y_i)
    phase("products_commit")
    # ---- 4. quotient.  Lagrange -> coefficients for everything the proof opens (in place: the Lagrange forms are done with)
    for t, cnt in ((cols, m), (Ap, Lk), (Sp, Lk), (Z, S), (Zl, Lk)):
        to_coeff(t, cnt)
    n_perm_lines = 2 + (S - 1) + S
    lt = ws.lk_ext[0][0].shape[0]
    for pi, pt in enumerate(d.parts):           # the parts of the quotient's domain: one (halo2's 4n coset) or two (three cosets of <w_n>)
        Np, lg, rot = pt["size"], k + pt["log_e"], pt["E"]
        cg, om = M(pt["coset_g"]), M(pt["omega"])
        z_ext = ws.z_ext[pi]                    # all sets: the chaining lines read z_{j-1} beside z_j
        for s0 in range(0, S, tile):
            extend(Z[s0:s0 + tile], min(tile, S - s0), z_ext[s0:s0 + tile], pt)
        ws.hh[pi].zero_()
        hg, hp = ws.hh[pi][0], ws.hh[pi][1]     # gate lines / permutation lines, folded apart and joined below: one pass over the tiles
        ext = ws.ext[pi]
        l0, llast, lact = (pk.l_ext[pi][i].data_ptr() for i in range(3))
        Rf, Rs = pk.ext_resident

        def key_tile(coeff, resident, R_, which, c0, cnt):
            """the extended forms of key columns [c0, c0 + cnt): resident, or (streamed proving key) re-extended into the tile buffer"""
            if not pk.streamed or c0 + cnt <= R_:
                return resident[pi][c0].data_ptr()
            buf = ws.key_ext[pi][which]
            extend(coeff[c0:c0 + cnt], cnt, buf, pt)
            return buf.data_ptr()

        for c0 in range(0, m, tile):
            cnt = min(tile, m - c0)
            extend(cols[c0:c0 + cnt], cnt, ext, pt)
            na = max(0, min(A, c0 + cnt) - c0)  # advice columns of this tile carry the custom gate
            if na:
                eng.quotient_gate_dev(ext.data_ptr(), 4 * Np, key_tile(pk.fixed_coeff, pk.fixed_ext, Rf, 0, c0, na), 4 * Np, na, lg, rot, y, hg.data_ptr())
            set_lo, nsets = c0 // CHUNK, -(-cnt // CHUNK)
            eng.quotient_permutation_part_dev(ext.data_ptr(), 4 * Np, key_tile(pk.sigma_coeff, pk.sigma_ext, Rs, 1, c0, cnt), 4 * Np, z_ext.data_ptr(), 4 * Np, S, set_lo,
                                              nsets, CHUNK, cnt, c0 == 0, lg, rot, bf + 1, l0, llast, lact, beta, gamma, M(DELTA), cg, om, y,
                                              hp.data_ptr())
        # h = hg * y^(permutation lines) + hp, then the lookup lines on top
        hq = ws.hp[pi]
        eng.fr_lincomb_dev(ws.hh[pi].data_ptr(), 2, 4 * Np, Np, M(pow(y_i, n_perm_lines, FR)), hq.data_ptr())
        for l0_ in range(0, Lk, lt):
            cnt = min(lt, Lk - l0_)
            e_in, e_ap, e_sp, e_zl = ws.lk_ext[pi]
            extend(cols[A + l0_:A + l0_ + cnt], cnt, e_in, pt)
            extend(Ap[l0_:l0_ + cnt], cnt, e_ap, pt)
            extend(Sp[l0_:l0_ + cnt], cnt, e_sp, pt)
            extend(Zl[l0_:l0_ + cnt], cnt, e_zl, pt)
            eng.quotient_lookup_dev(e_in.data_ptr(), 4 * Np, pk.table_ext[pi].data_ptr(), e_ap.data_ptr(), 4 * Np, e_sp.data_ptr(), 4 * Np,
                                    e_zl.data_ptr(), 4 * Np, cnt, lg, rot, l0, llast, lact, beta, gamma, y, hq.data_ptr())
        eng.quotient_finish_dev(hq.data_ptr(), k, pt["log_e"], cg, om)
        # back to coefficients on this part: the quotient modulo X^size - coset_g^size
        eng.ntt_dev(hq.data_ptr(), 1, 4 * Np, M(pt["omega_inv"]), lg, None, M(pt["size_inv"]))
        eng.fr_distribute_powers_dev(hq.data_ptr(), 1, 4 * Np, Np, M(pow(pt["coset_g"], -1, FR)))
    h = ws.h
    pieces = h.view(d.E, n, 4)                  # h(X) = sum_i X^(n i) h_i(X); degree <= 3n - 4
    if len(d.parts) == 1:
        h.copy_(ws.hp[0])                       # halo2's domain: the 4n coefficients themselves (piece 3 and the top of piece 2 vanish)
    else:
        # the quotient from three cosets.  With deg h < 3n:  on part A (g <w_2n>) X^2n = g^2n, so the 2n coefficients found there are
        # [U | h_1] with U = h_0 + g^2n h_2;  on part B (the coset of <w_n> by c = g w_4n) X^n = c^n =: lam (lam^2 = -g^2n), so the n
        # coefficients found there are V = h_0 + lam h_1 - g^2n h_2.  Hence h_2 = (U - V + lam h_1) / (2 g^2n), h_0 = U - g^2n h_2.
        g2n = pow(d.coset_g, 2 * n, FR)
        lam = pow(d.parts[1]["coset_g"], n, FR)
        U, h1, V = ws.hp[0][:n], ws.hp[0][n:], ws.hp[1]
        pieces[1].copy_(h1)
        t01 = ws.tmp
        t01[0].copy_(h1); t01[1].copy_(V)
        eng.fr_lincomb_dev(t01.data_ptr(), 2, 4 * n, n, M(-lam % FR), pieces[2].data_ptr())            # T = V - lam h_1
        t01[0].copy_(pieces[2]); t01[1].copy_(U)
        eng.fr_lincomb_dev(t01.data_ptr(), 2, 4 * n, n, M(FR - 1), pieces[2].data_ptr())               # U - T
        eng.fr_distribute_powers_dev(pieces[2].data_ptr(), 1, 4 * n, n, M(1), M(pow(2 * g2n, -1, FR)))     # h_2
        t01[0].copy_(pieces[2]); t01[1].copy_(U)
        eng.fr_lincomb_dev(t01.data_ptr(), 2, 4 * n, n, M(-g2n % FR), pieces[0].data_ptr())            # h_0 = U - g^2n h_2
        pieces[3].zero_()
    c_h = commit(bm, pieces, d.E - 1, 4 * n)
    (a_h,) = tr.absorb_points(eng, c_h)
    x_i = tr.squeeze("x")
    phase("quotient")
    # ---- 5. evaluations -> SHPLONK's y, v
    xs = rotation_points(d, x_i)
    P = lambda idx: np.stack([M(xs[i]) for i in idx])

    def evals(t, cnt, idx):
        out = _zeros(cnt, len(idx), 4)
        eng.poly_eval_multi_dev(t.data_ptr(), cnt, 4 * n, n, P(idx), out.data_ptr())
        return out

    xn = pow(x_i, n, FR)
    hcomb = ws.hcomb                            # h_0 + x^n h_1 + x^2n h_2: what halo2 opens at x
    for i in reversed(range(d.E - 1)):
        eng.fr_lincomb_dev(pieces[i].data_ptr(), 1, 4 * n, n, M(xn), hcomb.data_ptr(), i != d.E - 2)
    F = A + 2
    e_adv = evals(cols, A, [0, 1, 2, 3])
    e_lk = evals(cols[A:], Lk + 1, [0])         # lookup advice + the constants column (a fixed column, opened at x)
    e_fix = evals(pk.fixed_coeff, F, [0])
    e_sig = evals(pk.sigma_coeff, m, [0])
    e_z = evals(Z, S, [0, 1, 4])
    e_zl = evals(Zl, Lk, [0, 1])
    e_ap = evals(Ap, Lk, [0, 5])
    e_sp = evals(Sp, Lk, [0])
    e_rnd = evals(rnd, 1, [0])
    e_h = evals(hcomb.view(1, n, 4), 1, [0])
    eng.sync()
    host = lambda t: t.cpu().numpy().view(np.uint64)
    polys = {"lookup_advice": cols[A:], "fixed": pk.fixed_coeff, "sigma": pk.sigma_coeff, "perm_tables": Sp, "h": hcomb.view(1, n, 4),
             "random": rnd, "advice": cols, "perm_z": Z, "lookup_z": Zl, "perm_inputs": Ap}
    ev = {"lookup_advice": host(e_lk), "fixed": host(e_fix), "sigma": host(e_sig), "perm_tables": host(e_sp), "h": host(e_h),
          "random": host(e_rnd), "advice": host(e_adv), "perm_z": host(e_z), "lookup_z": host(e_zl), "perm_inputs": host(e_ap)}
    tr.absorb_scalars(*[ev[f] for f in ("advice", "lookup_advice", "fixed", "sigma", "perm_z", "lookup_z", "perm_inputs", "perm_tables", "random")])
    shy, shv = tr.squeeze("sh_y"), tr.squeeze("sh_v")
    phase("evaluations")
    # ---- 6. SHPLONK -> u
    # (pointers and evaluation rows by arithmetic / fancy indexing: a tensor view and a numpy row per member cost ~20 ms of host time
    # for the 11 000 members of a c2 proof)
    base_ptr = {f: t.data_ptr() for f, t in polys.items()}
    sets = []
    for idx, members in query_layout(A, Lk, m, S):
        ptrs = [base_ptr[f] + i * (n * 32) for f, i in members]
        rows, start = [], 0
        while start < len(members):                       # runs of one family: one slice each
            f, i0 = members[start]
            end = start
            while end + 1 < len(members) and members[end + 1] == (f, members[end][1] + 1):
                end += 1
            rows.append(ev[f][i0:i0 + end - start + 1, :len(idx)])
            start = end + 1
        sets.append((ptrs, idx, np.concatenate(rows)))
    w1, w2 = ws.w1, ws.w2
    state = eng.shplonk_begin_dev(n, sets, np.stack([M(p) for p in xs]), M(shy), M(shv), w1.data_ptr())
    c_w1 = commit(bm, w1.view(1, n, 4), 1, 4 * n)
    (a_w1,) = tr.absorb_points(eng, c_w1)
    shu = tr.squeeze("sh_u")
    eng.shplonk_finish_dev(state, M(shu), w1.data_ptr(), w2.data_ptr())
    c_w2 = commit(bm, w2.view(1, n, 4), 1, 4 * n)
    eng.sync()
    phase("multiopen")
    # ---- the proof (the affine forms are the ones the transcript absorbed phase by phase: normalised once)
    pr.commitments = {"advice": a_adv[:A], "lookup_advice": a_adv[A:], "perm_inputs": a_ap, "perm_tables": a_sp,
                      "perm_z": a_z, "lookup_z": a_zl, "random": a_rnd, "h": a_h, "w1": a_w1, "w2": eng.g1_normalize(host(c_w2))}
    pr.evals = {"advice": ev["advice"], "lookup_advice": ev["lookup_advice"][:Lk], "constants": ev["lookup_advice"][Lk:], "fixed": ev["fixed"],
                "sigma": ev["sigma"], "perm_z": ev["perm_z"], "lookup_z": ev["lookup_z"], "perm_inputs": ev["perm_inputs"],
                "perm_tables": ev["perm_tables"], "random": ev["random"], "h": ev["h"]}
    # degree <= 3n - 4.  With halo2's domain: every coefficient from 3n - 3 to 4n - 1; from three cosets the interpolation has only 3n
    # coefficients, of which the top three must vanish (an unsatisfied witness fails this with probability 1 - 2^-760, and the
    # identity at x below it in any case)
    top = host(h[3 * n - 3:] if len(d.parts) == 1 else h[3 * n - 3:3 * n])
    pr.h_top = top
    pr.h_degree_ok = not top.any()
    phase("finalise")
    return pr
