#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Paillier-in-Halo2 hot path.

Metric (BASELINE.json): Paillier-encrypt proofs/s (2048-bit n, k=17); MSM achieved HBM GB/s vs peak.

Since round 6 a "step" is ONE CONNECTED PROOF at config c2 (2048-bit n, k=17, lookup_bits=16, limb_bits=64) -- the whole create_proof
dataflow on the proof's own data, checked after the timed loop as the verifier would (connected_line, bench_connected.py; BASELINE.json
config 2: "full KZG proof at k=17").  Rounds 1-5's headline -- the HOT PATH ONLY -- is the `hot_path_only` leg of the same line
(hot_path_line; `--hot-path-headline` makes it `value` again):

A hot-path step is ONE pass of the hot path of ONE encrypt proof at config c2, inputs resident in HBM:
    K3  witness trace   g^m * r^n mod n^2, ~6145 mul_mod steps (pz_paillier_encrypt_dev)
    K1  commitments     A advice-column MSMs (short witness scalars) + Lk lookup-column MSMs
                        + the full-width MSMs of the lookup / permutation / quotient / opening phases
    K2  polynomials     every column: iNTT 2^k (Lagrange -> coeff) and coset NTT 2^(k+2)
    K4  cell expansion  trace -> ~4e8 advice cells (12.7 GB) + lookup cells, the circuit's columns
with the column / MSM / NTT counts of paillier_halo2_amd/layout.py.  Advice and
lookup columns committed in K1 are the REAL cells K4 wrote; the scalars of the later-phase MSMs and the
NTT inputs are uniformly random field elements (what grand products / quotient pieces look like) from a
resident pool.  All arithmetic work of the listed kernels is performed every step, nothing is cached
between steps.  What a full prover does OUTSIDE this hot path (transcript hashing,
quotient evaluation, permutation/lookup product construction -- SURVEY.md section 8f "next") stays in the
reference's Rust and is NOT in `value`; DESIGN.md section 6 says so next to the number.

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run,
one rank per GPU, independent proofs per rank (weak scaling, no data-path collective).
`--workload msm22` instead times config c4: one 2^22-point MSM with Pippenger windows sharded over
the ranks and an RCCL all-gather + fixed-order fold of the partial points.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# This process drives more than four HIP streams (three library contexts, their copy streams, torch's): with ROCm's default
# of 4 hardware queues per process some of them share a queue and serialise (measured: the uploads of pz_msm_g1_batch then run
# AFTER the kernels they should run beside).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
MULT_PER_MADD = 6 * 180.0 + 2 * 135.0 + 252.0   # v_mad_u64_u32 + v_mul_lo_u32 per XYZZ mixed addition on the 29-bit field: 6 products, 2 squares,
                                                # and Y3 = R t - Y1 PPP as two products under ONE reduction (f29_mul2: 243 + 9); 1710 before round 3


def synth_inputs(enc_bits: int, seed: int):
    """seeded (n, g, m, r): n = p*q exactly enc_bits bits (p, q random odd), g = n+1, m in [0,n), r in [1,n)
    -- SURVEY.md section 8d; same recipe as oracle/pyref.synth_paillier_inputs, restated so the product path
    never imports oracle/."""
    import random

    rng = random.Random(seed)
    half = enc_bits // 2
    while True:
        p = rng.getrandbits(half) | (1 << (half - 1)) | 1
        q = rng.getrandbits(half) | (1 << (half - 1)) | 1
        n = p * q
        if n.bit_length() == enc_bits:
            break
    return n, n + 1, rng.randrange(0, n), rng.randrange(1, n)


def _same_shape_message(m: int, rng) -> int:
    """another message with m's bit length and popcount (so the circuit of paillier.rs:50-55 keeps its shape): a few
    (set bit, clear bit) pairs below the top bit swapped"""
    top = m.bit_length() - 1
    ones = [i for i in range(top) if (m >> i) & 1]
    zeros = [i for i in range(top) if not (m >> i) & 1]
    if not ones or not zeros:
        return m
    for _ in range(min(64, len(ones), len(zeros))):
        i, j = rng.choice(ones), rng.choice(zeros)
        if (m >> i) & 1 and not (m >> j) & 1:
            m ^= (1 << i) | (1 << j)
    return m


def fnv1a64(data: bytes) -> int:
    """FNV-1a over bytes, eight at a time as little-endian words (the same walk as host/prove_c2.cpp's)"""
    h = 0xCBF29CE484222325
    a = np.frombuffer(data, dtype="<u8")
    for w in a.tolist():
        h = ((h ^ w) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


class ProofWorkload:
    """device-resident state of the c2 hot path on one GPU"""

    def __init__(self, eng, torch, enc_bits: int, k: int, seed: int, scale: float, pool: int = 256, circuit: str = "encrypt",
                 lookup_bits=None, shard=(0, 1), dist=None, minimum_rows: int = 20):
        from paillier_halo2_amd import consts, layout

        self.eng, self.torch = eng, torch
        self.shard, self.dist = shard, dist   # (rank, world) when ONE proof's columns are split over the ranks
        self.enc_bits, self.k = enc_bits, k
        self.n = 1 << k
        self.Ln = enc_bits // 64
        self.L = 2 * self.Ln
        dev = "cuda"
        nn, g, m, r = synth_inputs(enc_bits, seed)
        self.circuit = circuit
        self.inputs = tuple(consts.int_to_limbs(x, self.Ln) for x in (nn, g, m, r))
        self._ints = (nn, g, m, r)
        if circuit == "encrypt":
            n_steps = m.bit_length() + bin(m).count("1") + nn.bit_length() + bin(nn).count("1") + 1
        elif circuit == "encrypt_uniform":   # g^m over enc_bits in-circuit bits, two mul_mods per bit
            n_steps = 2 * enc_bits + nn.bit_length() + bin(nn).count("1") + 1
        else:  # PaillierChip::add (paillier.rs:62-85): one mul_mod of two ciphertexts assigned at enc_bits (bench.rs:98-103)
            n_steps = 1
            self.add_ops = tuple(consts.int_to_limbs(x, self.L) for x in (m, r, nn * nn))
        self.n_steps = n_steps
        self.ng = (m.bit_length() + bin(m).count("1")) if circuit == "encrypt" else 2 * enc_bits if circuit == "encrypt_uniform" else 0
        self.nr = (nn.bit_length() + bin(nn).count("1")) if circuit != "add" else 0
        self.shape = layout.encrypt_proof_shape(enc_bits, k, n_steps, lookup_bits=lookup_bits, kind=circuit, n_steps_g=self.ng, minimum_rows=minimum_rows)
        self.kind = {"encrypt": 0, "add": 1, "encrypt_uniform": 2}[circuit]
        # the driver's inputs as the C ABI takes them (n | g | m | r | res words).  `res` is the expected ciphertext the
        # reference's driver receives from paillier_enc_native / paillier_add_native (bench.rs:149,193): here the library's own
        # native entry points (pz_paillier_encrypt without a trace, pz_mul_mod) -- no host big-integer arithmetic
        if circuit != "add":
            res_limbs = eng.paillier_encrypt(self.Ln, *self.inputs, want_steps=False)[0][0]
        else:
            res_limbs = eng.mul_mod(self.L, consts.int_to_limbs(m, self.L), consts.int_to_limbs(r, self.L), consts.int_to_limbs(nn * nn, self.L))[1]
        self.circ_inputs = np.concatenate([consts.int_to_limbs(x, self.Ln) for x in (nn, g, m, r)] + [np.asarray(res_limbs, dtype=np.uint64)])
        # THREE (message, randomness) pairs of the same key go round the TWO witness slots, step i proving pair i % 3: a slot
        # overwritten too early or read too early then holds another message's cells and the check after the timed loop
        # (verify_pipelined) sees it -- with one message every slot would hold the same values whatever the order of the streams.
        # The pairs share the circuit's SHAPE (paillier.rs:50-55 bakes the message's bits into it): same bit length and
        # popcount of m, hence the same step count, columns and proving key
        self.variants = [dict(inputs=self.inputs, circ_inputs=self.circ_inputs, add_ops=getattr(self, "add_ops", None), ints=(nn, g, m, r))]
        import random as _rnd

        vr = _rnd.Random(seed ^ 0x766172)
        for _ in range(2):
            if circuit == "add":
                m2, r2 = vr.randrange(0, nn), vr.randrange(1, nn)
            else:
                m2, r2 = _same_shape_message(m, vr), vr.randrange(1, nn)
            inp = tuple(consts.int_to_limbs(x, self.Ln) for x in (nn, g, m2, r2))
            if circuit != "add":
                res2 = eng.paillier_encrypt(self.Ln, *inp, want_steps=False)[0][0]
                ops = None
            else:
                ops = tuple(consts.int_to_limbs(x, self.L) for x in (m2, r2, nn * nn))
                res2 = eng.mul_mod(self.L, ops[0], ops[1], ops[2])[1]
            ci = np.concatenate(list(inp) + [np.asarray(res2, dtype=np.uint64)])
            self.variants.append(dict(inputs=inp, circ_inputs=ci, add_ops=ops, ints=(nn, g, m2, r2)))
        self._steps_done = 0
        self.drop_edge = os.environ.get("PZ_BENCH_DROP_EDGE", "")   # negative test of verify_pipelined only: "ready" / "ntt_ready" (a consumer does not wait for K4) / "free" (the producer does not wait for its slot's readers)
        sh = self.shape
        sc = lambda x: max(1, int(round(x * scale)))
        self.counts = dict(msm_full=sc(sh.msm_full), polys=sc(sh.polys))
        self.scale = scale
        # K3 output buffer (steps stay in HBM for K4) and the K4 cell streams: the circuit's advice columns are
        # consecutive runs of `rows` cells (2^k minus the blinding rows), lookup columns likewise
        self.d_mod = torch.from_numpy(consts.int_to_limbs(nn * nn, self.L).astype(np.int64)).to(dev)
        self.cells, self.lookups = eng.witness_cells_per_step(self.L, 64, self.shape.lookup_bits)
        self.row_budget = rb = layout.row_budget(k, minimum_rows)     # the tester's: calculate_params(Some(20)) on the bench path (layout.RowBudget)
        self.rows = rb.max_rows
        # the WHOLE circuit's stream (assignments, square + refresh, every mul_mod, assert_equal_fresh): row a6
        self.circ_adv, self.circ_lk = eng.circuit_cells(self.kind, self.Ln, 64, self.shape.lookup_bits, self.ng, self.nr)
        assert (self.circ_adv, self.circ_lk) == (self.shape.advice_cells, self.shape.lookup_cells)
        self.adv_cols = rb.columns_for(self.circ_adv)     # configured columns: >= the ones a cut at `rows` fills (an empty one is committed all the same)
        self.lk_cols = rb.columns_for(self.circ_lk)
        # proofs are independent, so the witness of proof i+1 (K3 trace: 4 wavefronts busy for 50 ms, then K4) is
        # produced on a second context / stream while proof i's commitments and NTTs run: two witness slots
        self.pipeline = os.environ.get("PZ_BENCH_PIPELINE", "1") == "1"
        nslots = 2 if self.pipeline else 1
        self.d_steps = [torch.zeros((n_steps, 4, self.L), dtype=torch.int64, device=dev) for _ in range(nslots)]
        # the circuit's columns as 2^k-row Lagrange vectors: `rows` usable rows of cells, the blinding rows left zero (a
        # prover fills them from its rng; their values change no kernel's work)
        self.d_adv = [torch.zeros((self.adv_cols * self.n, 4), dtype=torch.int64, device=dev) for _ in range(nslots)]
        self.d_lk = [torch.zeros((self.lk_cols * self.n, 4), dtype=torch.int64, device=dev) for _ in range(nslots)]
        self.d_out_adv = torch.zeros((self.adv_cols, 12), dtype=torch.int64, device=dev)
        # optional: the NTTs (multiplier-bound) on a third context / stream, beside the MSMs' memory-bound sort
        self.engn, self.stream_n = eng, None
        if os.environ.get("PZ_BENCH_NTT_STREAM", "1") == "1":
            import paillier_halo2_amd as pz

            self.engn = pz.Engine(eng.device)
            self.stream_n = torch.cuda.Stream()
            self.engn.set_stream(self.stream_n.cuda_stream)
        self.engw, self.stream_w = eng, None
        if self.pipeline:
            import paillier_halo2_amd as pz

            self.engw = pz.Engine(eng.device)
            self.stream_w = torch.cuda.Stream()
            self.engw.set_stream(self.stream_w.cuda_stream)
            self.ready_ev = [torch.cuda.Event() for _ in range(2)]
            self.free_ev = [torch.cuda.Event() for _ in range(2)]
            self.ntt_free_ev = [torch.cuda.Event() for _ in range(2)]
        self.own_ntt = os.environ.get("PZ_BENCH_OWN_NTT", "1") == "1"   # K2 on the proof's own K4 columns (0: pool only, as in round 2)
        self.fused_ntt = os.environ.get("PZ_BENCH_FUSED_NTT", "0") == "1"   # measured: not faster
        self.digit_adds = 0  # filled by count_digit_adds() after a warm-up step
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        self.gen = gen
        # The SRS as the reference gets it (bench.rs:161-171 -> gen_srs): a `params/kzg_bn254_{k}.srs` file.  The file is made here
        # the way gen_srs makes it (ParamsKZG::setup with a seeded scalar -> monomial bases g[i] = [s^i]G, on the device), then
        # READ BACK through paillier_halo2_amd/srs.py, validated on the device (pz_g1_check_dev = read_raw's is_on_curve) and its
        # Lagrange bases derived from the monomial ones by the G1 inverse FFT (pz_srs_lagrange_from_monomial_dev): the toxic
        # scalar is not used after the file exists.
        import random as _random
        import tempfile

        from paillier_halo2_amd import srs as pzsrs

        t_srs = time.perf_counter()
        s_toxic = _random.Random(seed ^ 0x535253).randrange(2, consts.FR_R)
        d_g = torch.zeros((self.n, 8), dtype=torch.int64, device=dev)
        eng.srs_setup_g1_dev(k, consts.fr_mont_limbs(s_toxic), consts.fr_mont_limbs(consts.fr_omega(k)), d_g.data_ptr(), 0)
        eng.sync()
        del s_toxic
        pdir = os.path.join(tempfile.gettempdir(), "pz_params_%d" % os.getpid())
        os.makedirs(pdir, exist_ok=True)
        ppath = os.path.join(pdir, "kzg_bn254_%d.srs" % k)
        g_host = d_g.cpu().numpy().view(np.uint64)
        pzsrs.write_params_kzg(ppath, k, g_host, np.zeros_like(g_host))   # the Lagrange section is re-derived below, not trusted
        del d_g, g_host
        t_file = time.perf_counter()
        params = pzsrs.read_params_kzg(ppath, expect_k=k)
        d_g = torch.from_numpy(np.array(params.g, dtype=np.uint64).view(np.int64)).to(dev)
        assert eng.g1_check_dev(d_g.data_ptr(), self.n) == 0, "SRS point not on the curve"
        d_b = torch.zeros((self.n, 8), dtype=torch.int64, device=dev)
        eng.srs_lagrange_from_monomial_dev(k, consts.fr_mont_limbs(pow(consts.fr_omega(k), -1, consts.FR_R)),
                                           consts.fr_mont_limbs(pow(self.n, -1, consts.FR_R)), d_g.data_ptr(), d_b.data_ptr())
        eng.sync()
        self.bases = eng.load_bases_dev(d_b.data_ptr(), self.n)
        self.lagrange_host = d_b.cpu().numpy().astype(np.uint64)   # 64 B x 2^k: what the checker commits the sampled columns against
        self.srs_ms = {"setup_and_write_file": (t_file - t_srs) * 1e3, "read_check_lagrange_table": (time.perf_counter() - t_file) * 1e3}
        del d_b, d_g, params
        try:
            os.unlink(ppath)
            os.rmdir(pdir)
        except OSError:
            pass
        # column pools (values synthetic, Montgomery form): witness-like / lookup digits / full width
        self.pool = pool
        self.col_f = self._rand_fr(pool * self.n).view(pool, self.n, 4)
        self.d_out = torch.zeros((pool, 12), dtype=torch.int64, device=dev)
        self.d_out_full = torch.zeros((self.counts["msm_full"], 12), dtype=torch.int64, device=dev)
        if self.stream_n is not None:
            self.col_n = self._rand_fr(pool * self.n).view(pool, self.n, 4)   # its own pool: NTTs run in place
        # NTT buffers
        self.ext_n = 1 << sh.ext_k
        self.ntt_batch = min(pool, int(os.environ.get("PZ_BENCH_NTT_BATCH", "40")))
        # 40 polynomials per call: the size at which the NTT stream and the commitment stream of a proof finish together
        # (64: the transforms end ~130 ms early and the commitments run 10 ms longer; 32: the other way round).  The extended
        # tile keeps 64 columns: the steps after the hot path read it 64 columns at a time
        self.d_ext = torch.zeros((max(64, self.ntt_batch), self.ext_n, 4), dtype=torch.int64, device=dev)
        self.omega_inv = consts.fr_mont_limbs(pow(consts.fr_omega(k), -1, consts.FR_R))
        self.n_inv = consts.fr_mont_limbs(pow(self.n, -1, consts.FR_R))
        w_ext = consts.fr_omega(sh.ext_k)
        self.omega_n = consts.fr_mont_limbs(consts.fr_omega(k))
        self.coset_gens = np.stack([consts.fr_mont_limbs(consts.FR_GENERATOR * pow(w_ext, r, consts.FR_R) % consts.FR_R)
                                    for r in range(1 << (sh.ext_k - k))])
        torch.cuda.synchronize()

    # ---- synthetic column values (generated on the GPU; converted to Montgomery by the library)
    def _rand_fr(self, count):
        t = self.torch
        x = t.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=t.int64, device="cuda", generator=self.gen)
        x[:, 3] &= 0x0FFFFFFFFFFFFFFF  # < 2^252 < r: a valid representative; uniform enough for digit statistics
        return x

    def count_digit_adds(self):
        """non-zero signed 16-bit digits of everything K1 accumulates in one step (torch arithmetic on the canonical
        values; measurement support only).  Full-width uniform scalars: 16 digits minus the 2^-16 zero chance each."""
        t = self.torch
        total = 0
        full = self.counts["msm_full"] * self.n * 16 * (1.0 - 2.0 ** -16)
        for buf, ncols in ((self.d_adv[0], self.adv_cols), (self.d_lk[0], self.lk_cols)):
            if self.scale != 1.0:
                ncols = max(1, int(round(ncols * self.scale)))
            cells = ncols * self.n
            CH = 1 << 22
            for c0 in range(0, cells, CH):
                x = buf[c0:min(cells, c0 + CH)].clone()
                self.eng.fr_convert_dev(x.data_ptr(), x.shape[0], False)  # canonical
                self.eng.sync()
                # negative field values (r - small) are negated by the kernel: count digits of min(k, r-k); those
                # are the cells whose top limb is non-zero here (honest witness values are < 2^192 otherwise)
                neg = x[:, 3] != 0
                # 16-bit pieces of the three low limbs; signed recoding adds at most one carry digit per scalar
                nz = t.zeros(x.shape[0], dtype=t.int64, device=x.device)
                for limb in range(3):
                    v = x[:, limb]
                    for sh in (0, 16, 32, 48):
                        nz += ((v >> sh) & 0xFFFF) != 0
                total += int(nz[~neg].sum().item()) + int(neg.sum().item()) * 9
                del x
        self.digit_adds = int(total + full)
        return self.digit_adds

    # ---- one pass of the hot path = produce(slot) [K3 + K4] then consume(slot) [K1 + K2]
    def produce(self, slot, variant=0, eng=None, events=True):
        """K3 + K4 of message `variant` into witness slot `slot` (on the witness context / stream; `eng` + events=False: on
        that context with no event traffic -- the serial reference of verify_pipelined)"""
        t = self.torch
        eng = eng or self.engw
        events = events and self.pipeline
        sh = self.shape
        if events:
            if self.drop_edge != "free":
                self.stream_w.wait_event(self.free_ev[slot])  # the previous consumer of this slot has finished reading it
                if self.stream_n is not None:
                    self.stream_w.wait_event(self.ntt_free_ev[slot])   # ... and so has the transform stream
        var = self.variants[variant]
        nn, g, m, r = var["inputs"]
        if events and self.drop_edge in ("ready", "ntt_ready"):
            # negative test only: with the consumer's wait removed the check must see STALE cells; that must not depend on K3 happening
            # to take longer than the consumer's first reads -- the witness stream idles ~50 ms first
            with t.cuda.stream(self.stream_w):
                t.cuda._sleep(100_000_000)
        skip = os.environ.get("PZ_BENCH_SKIP", "")   # debug only ("k3" / "k4"): see consume()
        self._produced = getattr(self, "_produced", 0) + 1
        if self._produced <= 2:
            skip = ""    # both witness slots hold a real witness before anything is skipped
        if "k3" in skip:
            pass
        elif self.circuit == "encrypt":
            eng.paillier_encrypt_dev(self.Ln, nn, g, m, r, self.d_steps[slot].data_ptr(), self.n_steps)
        elif self.circuit == "encrypt_uniform":
            eng.paillier_encrypt_uniform_dev(self.Ln, self.enc_bits, nn, g, m, r, self.d_steps[slot].data_ptr(), self.n_steps)
        else:
            a, b, mod = var["add_ops"]
            q, rem = eng.mul_mod(self.L, a, b, mod)
            with (t.cuda.stream(self.stream_w) if (self.pipeline and eng is self.engw) else _null()):
                self.d_steps[slot].copy_(t.from_numpy(np.stack([a, b, q, rem]).astype(np.int64)).view(1, 4, self.L))
        # K4: expand the whole operation tape into the advice / lookup cell streams (the circuit's columns)
        if "k4" not in skip:
            eng.circuit_expand_dev(self.kind, self.Ln, 64, sh.lookup_bits, var["circ_inputs"], self.d_steps[slot].data_ptr(), self.ng, self.nr,
                                   self.d_mod.data_ptr(), self.d_adv[slot].data_ptr(), self.d_lk[slot].data_ptr(), self.rows, self.n)
        if events:
            self.ready_ev[slot].record(self.stream_w)

    def consume(self, slot, msm_only=False, tail=False, keep=None):
        """keep: a dict from new_keep() -- the commitments go to its buffers instead of the shared ones and the transforms of
        its sampled columns are copied out (verify_pipelined); the launches and the event edges are the timed loop's"""
        eng, t = self.eng, self.torch
        n, k, sh = self.n, self.k, self.shape
        if self.pipeline and self.drop_edge != "ready":
            t.cuda.current_stream().wait_event(self.ready_ev[slot])
        from paillier_halo2_amd import dist as pzd

        rk, ws = self.shard
        skip = os.environ.get("PZ_BENCH_SKIP", "")   # debug only ("msm" / "ntt"): time one half of the hot path alone
        # K1: commitments -- every advice and lookup-advice column (real witness cells) ...
        for which, (buf, ncols) in enumerate(() if ("msm" in skip or "wit" in skip) else ((self.d_adv[slot], self.adv_cols), (self.d_lk[slot], self.lk_cols))):
            if self.scale != 1.0:
                ncols = max(1, int(round(ncols * self.scale)))
            lo, hi = pzd.column_range(ncols, rk, ws)   # column-parallel mode: this rank's columns of the shared proof
            d_o = self.d_out_adv if keep is None else keep["lk" if which else "adv"]
            if hi > lo:
                eng.msm_dev(self.bases, buf.data_ptr() + lo * self.n * 32, hi - lo, self.n, 4 * self.n,
                            d_o.data_ptr())
            if self.dist is not None:   # column-parallel mode (a process group of one rank runs the collective too: --force-dist)
                pzd.gather_commitments(t, self.dist, d_o[: hi - lo], ncols, rk, ws)
        # ... and the full-width MSMs of the later prover phases (permuted lookup columns, grand products,
        # quotient pieces, openings): uniformly random scalars
        lo, hi = pzd.column_range(self.counts["msm_full"], rk, ws)
        done = hi if "msm" in skip else lo
        while done < hi:
            nc = min(self.pool, hi - done)
            e_ = self.engn if (os.environ.get("PZ_BENCH_MSM2") == "1" and (done // self.pool) & 1) else eng   # experiment
            e_.msm_dev(self.bases, self.col_f.data_ptr(), nc, n, 4 * n, (self.d_out_full if keep is None else keep["full"])[done - lo].data_ptr())
            done += nc
        if self.dist is not None:
            pzd.gather_commitments(t, self.dist, (self.d_out_full if keep is None else keep["full"])[: hi - lo], self.counts["msm_full"], rk, ws)
        if not msm_only and "ntt" not in skip:
            self.consume_ntt(slot, keep=keep)
        if tail:
            self.tail_run(slot)
        if self.pipeline:
            self.free_ev[slot].record(t.cuda.current_stream())

    def consume_ntt(self, slot=0, keep=None):
        eng, n, k, sh = self.engn, self.n, self.k, self.shape
        t = self.torch
        # K2: Lagrange -> coeff (iNTT 2^k) -> extended coset: 4 interleaved coset NTTs with the 1/n divisor folded into
        # their pre-scale tables (pz_ntt_fr_extend_dev == zero-extend, distribute_powers, best_fft(omega_ext)).
        # The proof's OWN advice and lookup columns (K4's output, which the commitment stream reads at the same time) are
        # transformed batch by batch OUT OF PLACE into the transform buffer (pz_ntt_fr_to_dev) -- the coefficient form is a
        # separate allocation in a prover too; the remaining polynomials of a proof (permuted lookup columns, lookup and permutation
        # products: A + 4 Lk + P in all) come from the resident pool of uniformly random field elements.
        from paillier_halo2_amd import dist as pzd

        p_lo, p_hi = pzd.column_range(self.counts["polys"], *self.shard)   # column-parallel mode: this rank's polynomials
        done = 0
        nb = self.ntt_batch
        pool_n = self.col_f if self.stream_n is None else self.col_n
        own = [(self.d_adv[slot], self.adv_cols), (self.d_lk[slot], self.lk_cols)] if (self.own_ntt and self.scale == 1.0 and self.shard == (0, 1)) else []
        if self.stream_n is not None and self.pipeline and self.drop_edge != "ntt_ready":
            self.stream_n.wait_event(self.ready_ev[slot])     # the columns K4 wrote
        for which, (buf, ncols) in enumerate(own):
            cols = buf.view(ncols, n, 4)
            c0 = 0
            while c0 < ncols and done < p_hi - p_lo:
                nc = min(nb, ncols - c0, p_hi - p_lo - done)
                work = pool_n[:nc]
                # lagrange_to_coeff out of place: the Lagrange values stay where the commitment stream reads them
                # (with its 1/n, as halo2's ifft has it: the coefficients are the real ones; the scale rides on an inter-pass product)
                eng.ntt_to_dev(cols[c0].data_ptr(), 4 * n, work.data_ptr(), 4 * n, nc, self.omega_inv, k, None, self.n_inv)
                eng.ntt_extend_dev(work.data_ptr(), nc, 4 * n, self.d_ext.data_ptr(), 4 * self.ext_n, k, sh.ext_k - k,
                                   self.omega_n, self.coset_gens, None)
                if keep is not None:   # sampled columns of this batch: coefficient and extended forms copied out on the transform stream
                    with (t.cuda.stream(self.stream_n) if self.stream_n is not None else _null()):
                        for j, (w_, c_) in enumerate(keep["samples"]):
                            if w_ == which and c0 <= c_ < c0 + nc:
                                keep["coef"][j].copy_(work[c_ - c0])
                                keep["ext"][j].copy_(self.d_ext[c_ - c0])
                c0 += nc
                done += nc
        if own and self.stream_n is not None and self.pipeline:
            self.ntt_free_ev[slot].record(self.stream_n)      # the witness slot has been read by this stream
        if own:
            # the pool rows used as work space hold coefficients now: refill is not needed, the transforms' cost does not depend
            # on the values (any canonical field elements)
            pass
        while done < p_hi - p_lo:
            nc = min(nb, p_hi - p_lo - done)
            # in place on the resident pool columns (as a prover consumes its own columns): they stay uniformly
            # random field elements from step to step, which is all the NTT's cost depends on
            off = (done % self.pool)
            if off + nc > self.pool:
                off = 0
            src = pool_n[off:off + nc]
            if self.fused_ntt:   # lagrange_to_coeff + coeff_to_extended in one call (fused passes; identical results)
                eng.ntt_coeff_extend_dev(src.data_ptr(), nc, 4 * n, self.d_ext.data_ptr(), 4 * self.ext_n, k, sh.ext_k - k,
                                         self.omega_n, self.omega_inv, self.n_inv, self.coset_gens)
            else:
                eng.ntt_dev(src.data_ptr(), nc, 4 * n, self.omega_inv, k, None, self.n_inv)
                eng.ntt_extend_dev(src.data_ptr(), nc, 4 * n, self.d_ext.data_ptr(), 4 * self.ext_n, k, sh.ext_k - k,
                                   self.omega_n, self.coset_gens, None)
            done += nc

    # ---- keygen (bench.rs:174-175 prints vk / pk time): commitments, coefficient and extended-coset forms of the fixed columns
    # (one selector per advice column, the lookup table, a constants column) and of the permutation polynomials, 256 columns at
    # a time through pz_permutation_sigma_dev + pz_keygen_columns_dev.  Column CONTENTS are stand-ins (0/1 selector patterns, the
    # identity permutation: the copy-constraint cycles are the dependency's circuit structure); the work does not depend on them
    # beyond selectors being short scalars and sigma values full-width ones.  Run once per key, outside the per-proof loop.
    def keygen_vk_pk(self):
        eng, t, n, k, sh = self.eng, self.torch, self.n, self.k, self.shape
        dev = self.d_ext.device
        nb = 256
        log_e = sh.ext_k - k
        d_cols = t.zeros((nb, n, 4), dtype=t.int64, device=dev)
        d_ext = t.zeros((nb, n << log_e, 4), dtype=t.int64, device=dev)
        d_com = t.zeros((nb, 12), dtype=t.int64, device=dev)
        sel = t.zeros((n, 4), dtype=t.int64, device=dev)
        sel[::3, 0] = 1
        eng.fr_convert_dev(sel.data_ptr(), n, True)
        map_col = t.arange(nb, dtype=t.int32, device=dev).repeat_interleave(n).contiguous()
        map_row = t.arange(n, dtype=t.int32, device=dev).repeat(nb).contiguous()
        delta = consts_mod().fr_mont_limbs(pow(consts_mod().FR_GENERATOR, 1 << 28, consts_mod().FR_R))
        n_fixed, n_sigma = sh.advice_cols + 2, sh.advice_cols + sh.lookup_cols + 1
        t.cuda.synchronize()
        t0 = time.perf_counter()
        for total, is_sigma in ((n_fixed, False), (n_sigma, True)):
            done = 0
            while done < total:
                nc = min(nb, total - done)
                if is_sigma:
                    eng.permutation_sigma_dev(map_col.data_ptr(), map_row.data_ptr(), nc, k, self.omega_n, delta, d_cols.data_ptr(), 4 * n)
                else:
                    d_cols[:nc] = sel
                eng.keygen_columns_dev(self.bases, d_cols.data_ptr(), nc, 4 * n, k, log_e, self.omega_n, self.omega_inv, self.n_inv,
                                       self.coset_gens, d_com.data_ptr(), d_ext.data_ptr(), 4 * (n << log_e))
                done += nc
        eng.sync()
        t.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        return {"vk_pk_ms": ms, "fixed_columns": n_fixed, "permutation_columns": n_sigma,
                "note": "keygen_vk + keygen_pk on the device for this circuit's column counts (selector-like fixed columns, identity permutation): "
                        "commitment + coefficient form + extended-coset form of every column, 256 columns per call; once per key, not in `value`"}

    # ---- the prover steps that follow the hot path: permutation and lookup grand
    # products, evaluate_h, evaluations at the challenge point and opening quotients.  Inputs: the permutation products
    # and the lookup argument run on the proof's OWN advice / lookup columns (slot's K4 output, 2^k-row columns); sigma
    # polynomials, selectors, l_0 / l_last / l_active and the challenges are synthetic (keygen and the transcript are
    # not on this path); evaluate_h reads the extended-coset tile the NTT stream writes (d_ext, 64 columns at a time).
    def tail_setup(self):
        from paillier_halo2_amd import consts

        t = self.torch
        n, k, N, sh = self.n, self.k, self.ext_n, self.shape
        T = {}
        T["one"] = consts.fr_mont_limbs(1)
        T["ch"] = [consts.fr_mont_limbs(pow(consts.FR_GENERATOR, e, consts.FR_R)) for e in (11, 13, 17, 19)]   # stand-ins for beta, gamma, y, x
        T["delta"] = consts.fr_mont_limbs(pow(consts.FR_GENERATOR, 1 << 28, consts.FR_R))
        zeta_i = pow(consts.FR_GENERATOR, (consts.FR_R - 1) // 3, consts.FR_R)
        T["zeta"] = consts.fr_mont_limbs(zeta_i)
        T["zeta_inv"] = consts.fr_mont_limbs(pow(zeta_i, -1, consts.FR_R))
        T["w_n"] = consts.fr_mont_limbs(consts.fr_omega(k))
        T["w_ext"] = consts.fr_mont_limbs(consts.fr_omega(sh.ext_k))
        T["w_ext_inv"] = consts.fr_mont_limbs(pow(consts.fr_omega(sh.ext_k), -1, consts.FR_R))
        T["n_inv_ext"] = consts.fr_mont_limbs(pow(N, -1, consts.FR_R))
        dev = self.d_ext.device
        T["sigma"] = self._rand_fr(self.pool * n).view(self.pool, n, 4)          # sigma polynomial values (keygen output)
        T["sel_ext"] = self._rand_fr(64 * N).view(64, N, 4)
        T["z_ext"] = self._rand_fr(32 * N).view(32, N, 4)
        T["lpoly"] = self._rand_fr(3 * N).view(3, N, 4)
        T["d_h"] = t.zeros((N, 4), dtype=t.int64, device=dev)
        T["d_z"] = t.zeros((self.pool // 2, n, 4), dtype=t.int64, device=dev)
        rows, lkc, lb = self.rows, self.lk_cols, sh.lookup_bits
        tab = t.zeros((rows, 4), dtype=t.int64, device=dev)
        tab[: 1 << lb, 0] = t.arange(1 << lb, device=dev)
        self.eng.fr_convert_dev(tab.data_ptr(), rows, True)
        T["tab"] = tab
        T["d_pi"] = t.zeros((lkc, n, 4), dtype=t.int64, device=dev)
        T["d_pt"] = t.zeros_like(T["d_pi"])
        T["d_zl"] = t.zeros_like(T["d_pi"])
        T["d_ev"] = t.zeros((self.pool, 4), dtype=t.int64, device=dev)
        T["d_q"] = t.zeros((8, n, 4), dtype=t.int64, device=dev)
        # SHPLONK rotation sets over pool polynomials (pointers into col_f, cycled)
        sh_pts = [pow(consts.FR_GENERATOR, 19, consts.FR_R)]
        w = consts.fr_omega(k)
        sh_pts += [sh_pts[0] * w % consts.FR_R, sh_pts[0] * pow(w, 2, consts.FR_R) % consts.FR_R, sh_pts[0] * pow(w, 3, consts.FR_R) % consts.FR_R,
                   sh_pts[0] * pow(w, -1, consts.FR_R) % consts.FR_R, sh_pts[0] * pow(w, n - 11, consts.FR_R) % consts.FR_R]
        T["sh_points"] = np.stack([consts.fr_mont_limbs(p_) for p_ in sh_pts])
        base_ptr = self.col_f.data_ptr()
        ptr = lambda j: base_ptr + (j % self.pool) * n * 32
        counts = [(sh.advice_cols + sh.advice_cols + sh.lookup_cols + 1 + 1, [0]),        # selectors + sigma + table
                  (sh.advice_cols, [0, 1, 2, 3]), (5 * sh.lookup_cols, [0, 1, 4]), (sh.perm_cols, [0, 1, 5])]
        rngs = np.random.default_rng(7)
        sets, j0 = [], 0
        for cnt, idx in counts:
            ev = rngs.integers(0, 1 << 60, size=(cnt, len(idx), 4), dtype=np.uint64)   # canonical (< r) Montgomery words
            sets.append(([ptr(j0 + j) for j in range(cnt)], idx, ev))
            j0 += cnt
        T["sh_sets"] = sets
        T["d_sh"] = t.zeros((2, n, 4), dtype=t.int64, device=dev)
        T["d_sh_out"] = t.zeros((2, 12), dtype=t.int64, device=dev)
        T["m_perm"] = sh.advice_cols + sh.lookup_cols + 1
        # advice at 4 rotations, its selector at x; per lookup: the permuted input at {x, w^-1 x}, the permuted table at x, the product at
        # {x, wx}; permutation products at {x, wx, w^last x}; sigma polynomials at x
        T["eval_classes"] = [(sh.advice_cols, [0, 1, 2, 3]), (sh.advice_cols, [0]), (sh.lookup_cols, [0, 4]), (sh.lookup_cols, [0]),
                             (sh.lookup_cols, [0, 1]), (sh.perm_cols, [0, 1, 5]), (T["m_perm"], [0])]
        T["n_evals"] = sum(c_ * len(p_) for c_, p_ in T["eval_classes"])
        assert T["n_evals"] == 5 * sh.advice_cols + 5 * sh.lookup_cols + 3 * sh.perm_cols + T["m_perm"]
        T["d_ev"] = t.zeros((self.pool, 4, 4), dtype=t.int64, device=dev)
        t.cuda.synchronize()
        self._tail = T
        return T

    def tail_products(self, slot=0):
        eng, T = self.eng, self._tail
        n, k, rows, lkc, lb = self.n, self.k, self.rows, self.lk_cols, self.shape.lookup_bits
        ch = T["ch"]
        # permutation grand products over the proof's own advice columns (sets of 2 columns), a pool's worth per call
        adv = self.d_adv[slot]
        done = 0
        while done < self.adv_cols:
            mc = min(self.pool, self.adv_cols - done)
            eng.permutation_product_sets_dev(adv.data_ptr() + done * n * 32, 4 * n, T["sigma"].data_ptr(), 4 * n, mc, 2, k, rows - 1,
                                             T["w_n"], ch[0], ch[1], T["delta"], T["d_z"].data_ptr(), 4 * n)
            done += mc
        # ... and over the lookup-advice columns (+ the one fixed table column, here the first sigma column)
        eng.permutation_product_sets_dev(self.d_lk[slot].data_ptr(), 4 * n, T["sigma"].data_ptr(), 4 * n, lkc, 2, k, rows - 1, T["w_n"], ch[0],
                                         ch[1], T["delta"], T["d_z"].data_ptr(), 4 * n)
        # lookup argument on the real digit columns: permute_expression_pair, then the grand products
        eng.lookup_permute_dev(self.d_lk[slot].data_ptr(), lkc, 4 * n, T["tab"].data_ptr(), rows, lb, T["d_pi"].data_ptr(),
                               T["d_pt"].data_ptr(), 4 * n)
        eng.lookup_product_dev(self.d_lk[slot].data_ptr(), 4 * n, T["tab"].data_ptr(), T["d_pi"].data_ptr(), 4 * n, T["d_pt"].data_ptr(),
                               4 * n, lkc, rows, ch[0], ch[1], T["one"], T["d_zl"].data_ptr(), 4 * n)

    def tail_quotient(self):
        eng, T, sh = self.eng, self._tail, self.shape
        N, k = self.ext_n, self.k
        log_e = sh.ext_k - k
        E = 1 << log_e
        ch, lpoly, sel_ext, z_ext, d_h = T["ch"], T["lpoly"], T["sel_ext"], T["z_ext"], T["d_h"]
        done = 0
        while done < sh.advice_cols:
            nc = min(64, sh.advice_cols - done)
            eng.quotient_gate_dev(self.d_ext.data_ptr(), 4 * N, sel_ext.data_ptr(), 4 * N, nc, sh.ext_k, E, ch[2], d_h.data_ptr())
            done += nc
        done = 0
        while done < T["m_perm"]:
            mc = min(64, T["m_perm"] - done)
            eng.quotient_permutation_dev(self.d_ext.data_ptr(), 4 * N, sel_ext.data_ptr(), 4 * N, z_ext.data_ptr(), 4 * N,
                                         -(-mc // 2), 2, mc, sh.ext_k, E, 10, lpoly[0].data_ptr(), lpoly[1].data_ptr(),
                                         lpoly[2].data_ptr(), ch[0], ch[1], T["delta"], T["zeta"], T["w_ext"], ch[2], d_h.data_ptr())
            done += mc
        done = 0
        while done < self.lk_cols:
            nl = min(16, self.lk_cols - done)
            eng.quotient_lookup_dev(self.d_ext.data_ptr(), 4 * N, sel_ext.data_ptr(), self.d_ext[16].data_ptr(), 4 * N,
                                    self.d_ext[32].data_ptr(), 4 * N, self.d_ext[48].data_ptr(), 4 * N, nl, sh.ext_k, E,
                                    lpoly[0].data_ptr(), lpoly[1].data_ptr(), lpoly[2].data_ptr(), ch[0], ch[1], ch[2],
                                    d_h.data_ptr())
            done += nl
        eng.quotient_finish_dev(d_h.data_ptr(), k, log_e, T["zeta"], T["w_ext"])
        eng.ntt_dev(d_h.data_ptr(), 1, 4 * N, T["w_ext_inv"], sh.ext_k, None, T["n_inv_ext"])
        eng.fr_distribute_powers_dev(d_h.data_ptr(), 1, 4 * N, N, T["zeta_inv"])

    def tail_evals(self):
        eng, T, n = self.eng, self._tail, self.n
        # a prover draws a fresh challenge x per proof: new points, hence new power tables, every call (the library's table cache
        # does not help and must not stall: pz_get_pow_table reuses the evicted table's buffer)
        from paillier_halo2_amd import consts

        self._eval_round = getattr(self, "_eval_round", 0) + 1
        x0 = pow(consts.FR_GENERATOR, 19 + 7 * self._eval_round, consts.FR_R)
        w = consts.fr_omega(self.k)
        pts = [x0, x0 * w % consts.FR_R, x0 * pow(w, 2, consts.FR_R) % consts.FR_R, x0 * pow(w, 3, consts.FR_R) % consts.FR_R,
               x0 * pow(w, -1, consts.FR_R) % consts.FR_R, x0 * pow(w, n - 11, consts.FR_R) % consts.FR_R]
        T["sh_points"] = np.stack([consts.fr_mont_limbs(p_) for p_ in pts])
        # every committed polynomial at its rotation set, one pass over the coefficients per polynomial (pz_poly_eval_multi_dev):
        # (count, points) classes as in halo2's evals phase; together T["n_evals"] point evaluations
        for cnt, pts in T["eval_classes"]:
            done = 0
            while done < cnt:
                nc = min(self.pool, cnt - done)
                eng.poly_eval_multi_dev(self.col_f.data_ptr(), nc, 4 * n, n, T["sh_points"][pts], T["d_ev"].data_ptr())
                done += nc
        # SHPLONK (halo2 multiopen): every polynomial of the proof in its rotation set -- fixed / selector / sigma columns at
        # {x}; advice at {x, wx, w^2 x, w^3 x} (the vertical gate's rotations); lookup polynomials at {x, wx, w^-1 x};
        # permutation products at {x, wx, w^last x} -- folded into the two final polynomials, each committed (pool
        # polynomials stand in for the coefficient forms; evaluations are synthetic: the work does not depend on them)
        st = eng.shplonk_begin_dev(n, T["sh_sets"], T["sh_points"], T["ch"][0], T["ch"][1], T["d_sh"][0].data_ptr())
        eng.msm_dev(self.bases, T["d_sh"][0].data_ptr(), 1, n, 4 * n, T["d_sh_out"][0].data_ptr())
        eng.shplonk_finish_dev(st, T["ch"][2], T["d_sh"][0].data_ptr(), T["d_sh"][1].data_ptr())
        eng.msm_dev(self.bases, T["d_sh"][1].data_ptr(), 1, n, 4 * n, T["d_sh_out"][1].data_ptr())

    def tail_run(self, slot=0):
        """everything after the hot path for the proof in `slot`, queued on the main stream"""
        self.tail_products(slot)
        if self.stream_n is not None:   # evaluate_h reads the extended tile the NTT stream wrote
            ev = self.torch.cuda.Event()
            ev.record(self.stream_n)
            self.torch.cuda.current_stream().wait_event(ev)
        self.tail_quotient()
        self.tail_evals()

    def tail(self):
        """per-phase wall time of the steps after the hot path, once, with nothing else on the GPU"""
        t = self.torch
        T = self._tail if hasattr(self, "_tail") else self.tail_setup()
        out = {}
        for name, fn in (("products", lambda: self.tail_products(0)), ("quotient", self.tail_quotient),
                         ("evaluations_and_openings", self.tail_evals)):
            fn()            # warm (pow tables, workspaces)
            t.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t.cuda.synchronize()
            out[name] = (time.perf_counter() - t0) * 1e3
        out["total"] = sum(out.values())
        out["counts"] = {"permutation_sets": -(-T["m_perm"] // 2), "permuted_columns": T["m_perm"], "lookups": self.lk_cols,
                         "gate_columns": self.shape.advice_cols, "point_evaluations": T["n_evals"]}
        out["note"] = ("ms per proof of the prover steps after the hot path, each phase alone on the GPU: permutation products over the "
                       "proof's own advice / lookup columns, permute_expression_pair + lookup products on its digit columns, "
                       "evaluate_h (gate + permutation + lookup terms, division, extended iNTT), evaluations at a point and "
                       "SHPLONK's multi-point opening (four rotation sets over every polynomial, both final commitments); transcript "
                       "and blinding randomness are not included")
        return out

    def run(self, steps, with_tail=False, keep=None):
        """exactly `steps` passes of the hot path (with_tail: each followed by the prover steps after it); with
        PZ_BENCH_PIPELINE the witness of pass i+1 overlaps the commitments / NTTs of pass i (every pass still does all of
        its work inside the timed region)"""
        if steps <= 0:
            return
        base, nv = self._steps_done, len(self.variants)
        self._steps_done += steps
        if not self.pipeline:
            for i in range(steps):
                self.produce(0, (base + i) % nv)
                self.consume(0, tail=with_tail, keep=keep[i] if keep else None)
            return
        self.produce(0, base % nv)
        for i in range(steps):
            self.consume(i & 1, tail=with_tail, keep=keep[i] if keep else None)   # asynchronous: returns once the launches are queued
            if i + 1 < steps:
                self.produce((i + 1) & 1, (base + i + 1) % nv)    # its trace call blocks the host while the GPU works on both streams

    def step(self):
        self.run(1)

    def release(self):
        """give the device memory back (the witness slots alone are 2 x 12.7 GB at c2) and close the extra contexts"""
        self.torch.cuda.synchronize()
        for e_ in (self.engw, self.engn):
            if e_ is not self.eng:
                e_.close()
        try:
            self.bases.free()
        except Exception:
            pass
        for name in list(self.__dict__):
            if name not in ("eng", "torch"):
                delattr(self, name)

    # ---- what the timed loop computes, checked: VERDICT r03 item 1 / ADVICE r03 #2.  The timed loop is a two-slot, three-stream,
    # three-context pipeline ordered by events; the at-size parity tests run the same kernels serially.  After the timed loop,
    # untimed: `steps` more PIPELINED steps (same launches, same event edges; only the output pointers differ and sampled
    # transforms are copied out) whose every advice / lookup commitment and sampled coefficient / extended columns are compared
    # with a SERIAL recomputation of the same message on one context and one stream, synchronised between stages.
    def new_keep(self):
        t = self.torch
        z = lambda *shape: t.zeros(shape, dtype=t.int64, device="cuda")
        a, l = self.adv_cols, self.lk_cols
        af, lf = -(-self.circ_adv // self.rows), -(-self.circ_lk // self.rows)     # the columns the cells fill (the last of them ragged)
        samples = sorted({(0, 0), (0, af // 2), (0, af - 1)} | ({(1, 0), (1, lf - 1)} if l else set()))
        return dict(adv=z(a, 12), lk=z(max(1, l), 12), full=z(self.counts["msm_full"], 12), samples=samples,
                    coef=z(len(samples), self.n, 4), ext=z(len(samples), self.ext_n, 4))

    def serial_reference(self, variant):
        """the same message on ONE context / ONE stream (the commitment context), a synchronisation after every stage; also
        keeps the cells of the sampled columns (for the oracle comparison of the cpu_baseline leg)"""
        eng, t, n, k, sh = self.eng, self.torch, self.n, self.k, self.shape
        t.cuda.synchronize()
        self.produce(0, variant, eng=eng, events=False)
        eng.sync()
        ref = self.new_keep()
        eng.msm_dev(self.bases, self.d_adv[0].data_ptr(), self.adv_cols, n, 4 * n, ref["adv"].data_ptr())
        eng.sync()
        if self.lk_cols:
            eng.msm_dev(self.bases, self.d_lk[0].data_ptr(), self.lk_cols, n, 4 * n, ref["lk"].data_ptr())
            eng.sync()
        done = 0
        while done < self.counts["msm_full"]:
            nc = min(self.pool, self.counts["msm_full"] - done)
            eng.msm_dev(self.bases, self.col_f.data_ptr(), nc, n, 4 * n, ref["full"][done].data_ptr())
            eng.sync()
            done += nc
        ref["cells"] = []
        for j, (w_, c_) in enumerate(ref["samples"]):
            col = (self.d_lk[0] if w_ else self.d_adv[0]).view(-1, n, 4)[c_]
            ref["cells"].append(col.clone())
            eng.ntt_to_dev(col.data_ptr(), 4 * n, ref["coef"][j].data_ptr(), 4 * n, 1, self.omega_inv, k, None, self.n_inv)
            eng.sync()
            eng.ntt_extend_dev(ref["coef"][j].data_ptr(), 1, 4 * n, ref["ext"][j].data_ptr(), 4 * self.ext_n, k, sh.ext_k - k,
                               self.omega_n, self.coset_gens, None)
            eng.sync()
        return ref

    def verify_pipelined(self, steps=4):
        t, eng = self.torch, self.eng
        if not (self.scale == 1.0 and self.shard == (0, 1) and self.own_ntt):
            return {"verified": None, "note": "not run: scaled / column-parallel / pool-only transforms"}
        nv = len(self.variants)
        keeps = [self.new_keep() for _ in range(steps)]
        base = self._steps_done
        err = None
        self.run(steps, keep=keeps)
        t.cuda.synchronize()
        for e_ in {id(x): x for x in (self.eng, self.engw, self.engn)}.values():
            try:
                e_.sync()      # PZ_ERR_ASYNC: a sort kernel saw its scalars change under it
            except Exception as ex:
                err = repr(ex)
        aff = lambda x: eng.g1_normalize(x.cpu().numpy().astype(np.uint64))
        refs, bad, n_com, n_ntt, hashes = {}, [], 0, 0, {}
        for i in range(steps):
            v = (base + i) % nv
            if v not in refs:
                refs[v] = self.serial_reference(v)
                refs[v]["aff"] = {kk: aff(refs[v][kk]) for kk in ("adv", "lk", "full")}
            ref, kp = refs[v], keeps[i]
            got = {kk: aff(kp[kk]) for kk in ("adv", "lk", "full")}
            for kk in ("adv", "lk", "full"):
                ne = np.nonzero((got[kk] != ref["aff"][kk]).any(axis=1))[0]
                n_com += got[kk].shape[0]
                if ne.size:
                    bad.append({"step": i, "message": v, "what": kk + " commitments", "columns_differing": int(ne.size), "first": int(ne[0])})
            for j, (w_, c_) in enumerate(kp["samples"]):
                for form in ("coef", "ext"):
                    n_ntt += 1
                    if not t.equal(kp[form][j], ref[form][j]):
                        bad.append({"step": i, "message": v, "what": "%s form of %s column %d" % (form, "lookup" if w_ else "advice", c_)})
            hashes[v] = "%016x" % fnv1a64(np.ascontiguousarray(np.concatenate([got["adv"], got["lk"][: self.lk_cols]])).tobytes())
        self._verify_refs = refs
        ok = not bad and err is None
        return {"verified": ok, "pipelined_steps_checked": steps, "messages": nv, "commitments_compared": n_com, "transforms_compared": n_ntt,
                "mismatches": bad[:8], "async_error": err, "commitment_hash_by_message": {str(k_): h_ for k_, h_ in sorted(hashes.items())},
                "dropped_edge": self.drop_edge or None,
                "note": "untimed, after the timed loop: pipelined steps (the timed loop's launches and event edges; three messages of one shape "
                        "through the two witness slots) against a serial one-context one-stream recomputation of each message: every advice / lookup / "
                        "full-width commitment (affine form) and the coefficient + extended forms of sampled columns, bit for bit"}


def env_switches():
    """every PZ_* / GPU_MAX_HW_QUEUES environment switch in effect, for the JSON line"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("PZ_") or k == "GPU_MAX_HW_QUEUES"}


def consts_mod():
    from paillier_halo2_amd import consts

    return consts


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _stdout_to_stderr:
    """file descriptor 1 points at stderr inside the block: RCCL prints a version banner to the C stdout when its first communicator
    comes up, and this program's stdout carries ONE JSON line"""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def connected_cpu_counts(c):
    """what halo2's CPU create_proof does for ONE connected proof of the shape `c` (ConnectedWorkload.counts()), in the units the port is
    timed in: MSMs, 2^k and 2^(k+2) transforms (halo2 extends to the 4n-point coset), and the field multiplications of the phases that are
    plain field arithmetic (counted per row from the formulas of halo2's permutation / lookup / evaluation / multiopen code; an estimate of
    the multiplication COUNT, priced with the measured multiplication rate)"""
    A, Lk, m, S, n = c["advice_cols"], c["lookup_cols"], c["permutation_cols"], c["permutation_sets"], c["n"]
    F = A + 2
    evaluations = 4 * A + (Lk + 1) + F + m + 3 * S + 2 * Lk + 2 * Lk + Lk + 2
    mults = {
        "grand_products": n * (5 * m + 4 * S) + n * 12 * Lk,            # numerator / denominator terms, batch inversion, prefix products
        "evaluate_h": 4 * n * (3 * A + 4 * m + 6 * S + 15 * Lk),         # gate, permutation, chaining / boundary and lookup lines on the 4n coset
        "evaluations": evaluations * n,                                  # Horner at x and its rotations
        "multiopen": (c["polys_opened"] + 24) * n,                       # SHPLONK's folds and the divisions by the sets' vanishing polynomials
    }
    return {"msm_witness": A + Lk, "msm_full": c["msm_full"], "ntt_n": A + 4 * Lk + S, "ntt_4n": A + 4 * Lk + S + 1, "evaluations": evaluations,
            "field_mults": mults}


def cpu_baseline(shape, n_steps, enc_bits, k, log, check_wl=None, connected=None):
    """The C restatement (oracle/pz_oracle.c, kind 'port') timed on this host's cores on a bounded
    sample of the same workload, extrapolated with the per-proof counts of `shape`.  This leg is the ONE place of bench.py that
    touches oracle/: besides the timing it runs the checker (`oracle_check`) on the serial reference verify_pipelined kept in
    `check_wl` -- as the checker, never as the thing measured."""
    import random

    from oracle import cref
    from oracle import pyref as P

    cref.build()
    cores = cref.lib().ora_num_threads()
    rng = random.Random(5)
    n = 1 << k
    t0 = time.time()
    bases = cref.walk_bases(n, rng.randrange(1, P.FR_R), rng.randrange(1, P.FR_R))
    full = np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    wit = cref.fr_ints_to_mont(P.witness_like_scalars(n, 3))
    log("cpu baseline setup %.1fs" % (time.time() - t0))

    def timeit(fn, reps):
        fn()
        t = time.time()
        for _ in range(reps):
            fn()
        return (time.time() - t) / reps

    # a bounded sample of ~10-20 s of CPU work (round 5: 0.5 s before -- too few calls to call it a measurement): repetitions
    # sized from a first call of each kernel
    def reps_for(fn, seconds, lo=2, hi=400):
        t = time.time()
        fn()
        one = max(1e-4, time.time() - t)
        return max(lo, min(hi, int(seconds / one)))

    f_full, f_wit = (lambda: cref.msm_g1(full, bases)), (lambda: cref.msm_g1(wit, bases))
    r_full, r_wit = reps_for(f_full, 4.0), reps_for(f_wit, 3.0)
    t_msm_full = timeit(f_full, r_full)
    t_msm_wit = timeit(f_wit, r_wit)
    omega = cref.fr_ints_to_mont([P.fr_omega(k)])[0]
    f_ntt = lambda: cref.ntt_fr(full, omega, k)
    r_ntt = reps_for(f_ntt, 2.0)
    t_ntt = timeit(f_ntt, r_ntt)
    ext = np.concatenate([full] * 4)
    omega_e = cref.fr_ints_to_mont([P.fr_omega(k + 2)])[0]
    f_ext = lambda: cref.ntt_fr(ext, omega_e, k + 2)
    r_ext = reps_for(f_ext, 3.0)
    t_ntt_ext = timeit(f_ext, r_ext)
    nn, g, m, r = P.synth_paillier_inputs(enc_bits, 0x5043)
    L = 2 * (enc_bits // 64)
    eb = min(1024, enc_bits)                       # a bounded sample of the chain: 1024 exponent bits (the whole exponent of a smaller key)
    e = m & ((1 << eb) - 1) | (1 << (eb - 1))
    t = time.time()
    rc, res, steps = cref.pow_mod_trace(L, nn * nn, g, e, L // 2)
    t_step = (time.time() - t) / max(1, len(steps))
    tail = None
    if connected is None:
        per_proof = (shape.msm_full * t_msm_full + (shape.msm_witness + shape.msm_lookup) * t_msm_wit
                     + shape.polys * (t_ntt + t_ntt_ext) + n_steps * t_step)
    else:
        # ONE CONNECTED PROOF on the CPU: the same kernels at the connected proof's counts + the field-arithmetic phases priced with this
        # host's measured multiplication rate (all threads, the way halo2's parallelize() splits them)
        cc = connected_cpu_counts(connected)
        rate = cref.fr_mul_rate()
        tail = {"field_mults_per_proof": cc["field_mults"], "field_mults_per_s_all_threads": rate,
                "seconds_per_proof": {k_: v_ / rate for k_, v_ in cc["field_mults"].items()}}
        per_proof = (cc["msm_full"] * t_msm_full + cc["msm_witness"] * t_msm_wit + cc["ntt_n"] * t_ntt + cc["ntt_4n"] * t_ntt_ext
                     + n_steps * t_step + sum(cc["field_mults"].values()) / rate)
        tail["counts"] = {k_: v_ for k_, v_ in cc.items() if k_ != "field_mults"}
    checker = None
    if check_wl is not None:
        try:
            checker = oracle_check(check_wl, log)
        except Exception as ex:
            checker = {"ok": None, "error": repr(ex)}
    return {
        "checker": checker,
        "value": 1.0 / per_proof, "unit": "proofs/s", "cores": int(cores), "kind": "port",
        "per_kernel_ms": {"msm_2pow%d_full_width" % k: t_msm_full * 1e3, "msm_2pow%d_witness_like" % k: t_msm_wit * 1e3,
                          "ntt_2pow%d" % k: t_ntt * 1e3, "ntt_2pow%d" % (k + 2): t_ntt_ext * 1e3, "mul_mod_step_us_single_thread": t_step * 1e6},
        "threads_note": "OpenMP threads = the cgroup's CPU quota of this box (oracle/pz_oracle.c::ora_num_threads), not the visible CPU count",
        "sample": ("oracle/pz_oracle.c (C restatement of best_multiexp / best_fft / mul_mod, OpenMP): %dx MSM 2^%d full-width "
                   "(%.3fs each), %dx MSM 2^%d witness-like (%.3fs), %dx NTT 2^%d (%.4fs), %dx NTT 2^%d (%.4fs), %d mul_mod "
                   "steps single-thread (%.1f us each); extrapolated with the per-proof counts in config"
                   % (r_full, k, t_msm_full, r_wit, k, t_msm_wit, r_ntt, k, t_ntt, r_ext, k + 2, t_ntt_ext, len(steps), t_step * 1e6)),
        "sample_cpu_seconds": r_full * t_msm_full + r_wit * t_msm_wit + r_ntt * t_ntt + r_ext * t_ntt_ext + len(steps) * t_step,
        "scope": ("one connected proof as halo2's CPU create_proof runs it: commitments, 2^k and 4n-coset transforms, the witness trace, and the "
                  "grand-product / evaluate_h / evaluation / multiopen phases as field-multiplication counts priced with this host's measured rate "
                  "(keygen, the transcript and permute_expression_pair's sort omitted)") if connected is not None else "the hot path's kernels only",
        "connected_tail": tail,
    }


def oracle_check(wl, log):
    """checker only (part of the cpu_baseline leg, the one place bench.py may touch oracle/): the sampled columns of the serial
    reference verify_pipelined compared the pipelined steps with -- K4 cells, K1 commitment, K2 coefficient and extended forms of
    one message -- against the oracle chain (Python trace -> Python cells -> C best_multiexp / best_fft)."""
    from oracle import cref
    from oracle import pyref as P

    t0 = time.time()
    refs = getattr(wl, "_verify_refs", None)
    if not refs:
        return {"ok": None, "note": "no serial reference kept"}
    v = sorted(refs)[-1]
    ref = refs[v]
    nn, g, m, r = wl.variants[v]["ints"]
    res = P.paillier_enc_native(nn, g, m, r)
    n, rows, k, sh = wl.n, wl.rows, wl.k, wl.shape
    assert wl.variants[v]["circ_inputs"][4 * wl.Ln:].tolist() == cref.int_to_limbs(res, wl.L).tolist(), "ciphertext of the witness vs paillier_enc_native"
    win = lambda j, tot: (j * rows, min((j + 1) * rows, tot))
    sa = [c_ for w_, c_ in ref["samples"] if w_ == 0]
    sl = [c_ for w_, c_ in ref["samples"] if w_ == 1]
    tot_a, tot_l, cells_a, cells_l = P.encrypt_circuit_cells_windows(nn, g, m, r, res, wl.enc_bits, 64, sh.lookup_bits,
                                                                     [win(j, wl.circ_adv) for j in sa], [win(j, wl.circ_lk) for j in sl])
    assert (tot_a, tot_l) == (wl.circ_adv, wl.circ_lk)
    bases = wl.lagrange_host
    mont = lambda x: cref.fr_ints_to_mont([x % P.FR_R])[0]
    w_inv, n_inv = pow(P.fr_omega(k), -1, P.FR_R), pow(n, -1, P.FR_R)
    log_e = sh.ext_k - k
    bad = []
    want_cells = {(0, c_): col for c_, col in zip(sa, cells_a)}
    want_cells.update({(1, c_): col for c_, col in zip(sl, cells_l)})
    for j, key in enumerate(ref["samples"]):
        col = want_cells[key]
        col_m = cref.fr_ints_to_mont(col + [0] * (n - len(col)))
        if not np.array_equal(ref["cells"][j].cpu().numpy().astype(np.uint64), col_m):
            bad.append(("cells", key))
        got_c = wl.eng.g1_normalize(ref["lk" if key[0] else "adv"][key[1]].cpu().numpy().astype(np.uint64).reshape(1, 12))[0]
        if not np.array_equal(got_c, cref.g1_normalize(cref.msm_g1(col_m, bases))):
            bad.append(("commitment", key))
        coeff = cref.fr_scale(cref.ntt_fr(col_m, mont(w_inv), k), mont(n_inv))
        if not np.array_equal(ref["coef"][j].cpu().numpy().astype(np.uint64), coeff):
            bad.append(("coefficients", key))
        ext_in = np.zeros((n << log_e, 4), dtype=np.uint64)
        ext_in[:n] = cref.fr_distribute_powers(coeff, mont(P.FR_GENERATOR))
        if not np.array_equal(ref["ext"][j].cpu().numpy().astype(np.uint64), cref.ntt_fr(ext_in, mont(P.fr_omega(sh.ext_k)), sh.ext_k)):
            bad.append(("extended", key))
    return {"ok": not bad, "message": v, "columns": [list(x) for x in ref["samples"]], "mismatches": [list(map(str, b)) for b in bad],
            "seconds": time.time() - t0,
            "note": "serial reference of message %d vs the oracle chain: cells, commitment, coefficient form and extended-coset form of %d sampled columns" % (v, len(ref["samples"]))}


def dropin_host_pointer_path(wl, torch, log, sample=256):
    """What a reference prover patched as INTEGRATION.md sections 2-3 describe would get WITHOUT restructuring its data flow:
    best_multiexp -> pz_msm_g1_batch and best_fft -> pz_ntt_fr_batch with HOST pointers (pinned memory here), i.e. every
    column crosses PCIe (4 MB up per commitment, 4 MB up + 4 MB down per transform of 2^17, 16 + 16 MB at 2^19).  Timed on
    a sample of `sample` columns of each class and extrapolated with the per-proof counts."""
    eng, n, k, sh = wl.eng, wl.n, wl.k, wl.shape
    torch.cuda.synchronize()
    host_full = torch.empty((sample, n, 4), dtype=torch.int64).pin_memory()
    host_full.copy_(wl.col_f[:sample].cpu())
    host_wit = torch.empty((sample, n, 4), dtype=torch.int64).pin_memory()
    host_wit.copy_(wl.d_adv[0].view(-1, n, 4)[:sample].cpu())
    host_ext = torch.empty((sample // 4, 4 * n, 4), dtype=torch.int64).pin_memory()
    host_ext.copy_(wl.d_ext[: sample // 4].cpu())
    as_cols = lambda tt: [tt[i].numpy().view(np.uint64) for i in range(tt.shape[0])]

    def timeit(fn):
        fn()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    def dev_ref():   # the same columns, device resident, one call (what the upload is overlapped with)
        eng.msm_dev(wl.bases, wl.col_f.data_ptr(), sample, n, 4 * n, wl.d_out_full.data_ptr())
        eng.sync()

    t_dev = timeit(dev_ref) / sample
    t_full = timeit(lambda: eng.msm_batch(wl.bases, as_cols(host_full))) / sample
    t_wit = timeit(lambda: eng.msm_batch(wl.bases, as_cols(host_wit))) / sample
    w_inv = wl.omega_inv
    t_ntt = timeit(lambda: eng.ntt_batch_inplace(as_cols(host_full), w_inv, k)) / sample
    from paillier_halo2_amd import consts
    w_ext = consts.fr_mont_limbs(consts.fr_omega(sh.ext_k))
    t_ntt_ext = timeit(lambda: eng.ntt_batch_inplace(as_cols(host_ext), w_ext, sh.ext_k)) / (sample // 4)
    n_wit = wl.adv_cols + wl.lk_cols
    per_proof = n_wit * t_wit + wl.counts["msm_full"] * t_full + wl.counts["polys"] * (t_ntt + t_ntt_ext)
    return {"ms_per_proof_extrapolated": per_proof * 1e3, "proofs_per_s_extrapolated": 1.0 / per_proof,
            "msm_full_ms_per_col": t_full * 1e3, "msm_full_device_resident_ms_per_col": t_dev * 1e3, "msm_witness_ms_per_col": t_wit * 1e3, "ntt_2pow%d_ms_per_col" % k: t_ntt * 1e3,
            "ntt_2pow%d_ms_per_col" % sh.ext_k: t_ntt_ext * 1e3, "sample_columns": sample,
            "note": "host-pointer entry points (pz_msm_g1_batch / pz_ntt_fr_batch) from pinned host memory, one proof's counts: "
                    "%d witness + %d full-width MSMs, %d x (NTT 2^%d + NTT 2^%d); the K3 / K4 witness stays on the host in this "
                    "binding and is not counted" % (n_wit, wl.counts["msm_full"], wl.counts["polys"], k, sh.ext_k)}


def dropin_device_resident(wl, args, log, env_extra=None, replicas=1):
    """The SAME hot path driven from plain C++ through the C ABI alone -- paillier_halo2_amd/host/prove_c2.cpp: pz_dev_alloc /
    pz_upload, three contexts ordered by pz_ctx_wait, no torch and no HIP call in the host -- i.e. what the reference's Rust
    prover patched at points C and D of INTEGRATION.md reaches.  Run as a child process after this process's own timed loops
    (the GPU is idle then); same counts, call sizes and pools as ProofWorkload."""
    import struct
    import subprocess
    import tempfile

    from paillier_halo2_amd import consts

    exe = os.path.join(ROOT, "tests", "cpp", "prove_c2")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "prove_c2"], stdout=subprocess.DEVNULL)
    import random as _random

    s_toxic = _random.Random(args.seed ^ 0x535253).randrange(2, consts.FR_R)
    sh = wl.shape
    nn = consts.limbs_to_int(wl.inputs[0]) if hasattr(consts, "limbs_to_int") else sum(int(v) << (64 * i) for i, v in enumerate(wl.inputs[0]))
    words = [0x335A50, wl.enc_bits, wl.k, sh.lookup_bits, wl.n_steps, wl.counts["msm_full"], wl.counts["polys"], wl.pool, wl.ntt_batch,
             args.steps, args.warmup, sh.ext_k - wl.k, args.seed, wl.rows, wl.row_budget.minimum_rows]
    arrs = [np.asarray(a, dtype=np.uint64) for a in wl.inputs] + [wl.circ_inputs[4 * wl.Ln:].astype(np.uint64),
                                                                   consts.int_to_limbs(nn * nn, wl.L)]
    arrs += [consts.fr_mont_limbs(s_toxic), wl.omega_n, wl.omega_inv, wl.n_inv, wl.coset_gens.reshape(-1)]
    # the other messages of the same shape (step i proves message i % 3, as in ProofWorkload.run): m | r | res each
    arrs.append(np.array([len(wl.variants) - 1], dtype=np.uint64))
    for var in wl.variants[1:]:
        arrs += [np.asarray(var["inputs"][2], dtype=np.uint64), np.asarray(var["inputs"][3], dtype=np.uint64), var["circ_inputs"][4 * wl.Ln:].astype(np.uint64)]
    blob = struct.pack("<%dQ" % len(words), *words) + b"".join(np.ascontiguousarray(a, dtype="<u8").tobytes() for a in arrs)
    with tempfile.NamedTemporaryFile(suffix=".job", delete=False) as f:
        f.write(blob)
        path = f.name
    try:
        env = dict(os.environ)
        env.pop("LD_PRELOAD", None)
        env.update(env_extra or {})
        p = subprocess.run([exe, path] + ([str(replicas)] if replicas > 1 else []), capture_output=True, text=True, timeout=900, env=env)
    finally:
        os.unlink(path)
    if p.returncode != 0:
        return {"error": (p.stderr or p.stdout)[-400:]}
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    out["note"] = ("paillier_halo2_amd/host/prove_c2.cpp: the hot path of one proof per step from plain C++ over include/pz.h only (device "
                   "memory through pz_dev_alloc / pz_upload, contexts ordered by pz_ctx_wait; no torch, no HIP in the host), same work "
                   "per step as `value`; the binding INTEGRATION.md section 5a shows for the reference's Rust side")
    return out


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_gpus: int, argv) -> int:
    """`python bench.py --gpus N` typed as it stands (no WORLD_SIZE in the environment): this parent makes NO GPU call
    (it never imports torch or the library) and starts `python -m torch.distributed.run` as a CHILD process -- one rank
    per GPU -- with the same arguments; the ranks' stdout / stderr are inherited, so rank 0's JSON line is this process's
    output; returns the launcher's exit code (non-zero if any rank failed)."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def stub_workload(args):
    """launcher / timing-contract rehearsal with NO GPU and no library call (tests/test_bench_launcher.py, gloo on the
    CPU): every rank runs the same barrier + max-over-ranks timing as the real workloads around a host-only loop and
    rank 0 prints the JSON line.  The line is marked not comparable: it measures nothing of the hot path."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend or "gloo", rank=rank, world_size=world)
    x = np.arange(1 << 16, dtype=np.uint64)

    def step():
        return int((x * np.uint64(rank + 3)).sum() & np.uint64(0xFFFF))

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "stub steps/s (launcher rehearsal, no GPU work)", "value": args.steps * world / float(tt.item()), "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(tt.item()) / max(1, args.steps) * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "comparable": False,
                          "config": {"workload": "stub"}, "rccl_ranks": dist.get_world_size() if use_dist else 1,
                          "backend": dist.get_backend() if use_dist else None}), flush=True)
    if use_dist:
        dist.destroy_process_group()


def hot_path_line(args, R):
    """the HOT PATH ONLY (K3 -> K4 -> K1 over the proof's own columns + K2, the later phases' commitments / transforms over
    pool scalars): rounds 1-5's headline, now reported under `hot_path_only` beside the connected proof (and still the whole line of
    --workload c3 and of --hot-path-headline).  -> the JSON dict on rank 0, None elsewhere"""
    import torch
    import torch.distributed as dist

    from paillier_halo2_amd import engine as E

    rank, world, use_dist, log, eng, mad_peak, barrier = R.rank, R.world, R.use_dist, R.log, R.eng, R.mad_peak, R.barrier
    t0 = time.time()
    if args.workload == "c3" and args.k == 17:
        args.k = 15
    if args.workload == "c5":   # BASELINE config c5: 3072-bit n, k = 19, 64 proofs over 8 GPUs = 8 independent proofs per GPU (replicas)
        args.enc_bits, args.k = 3072, 19
        if "--steps" not in sys.argv:
            args.steps = 8
        args.workload = "c2"
    colpar = args.parallel == "columns"
    wl = ProofWorkload(eng, torch, args.enc_bits, args.k, seed=args.seed + (0 if colpar else rank), scale=args.scale,
                       pool=int(os.environ.get("PZ_BENCH_POOL", "256")),   # columns per full-width commitment call (tuning only)
                       lookup_bits=args.lookup_bits, shard=(rank, world) if colpar else (0, 1), dist=dist if (colpar and use_dist) else None,
                       circuit="add" if args.workload == "c3" else "encrypt_uniform" if args.workload == "c2u" else "encrypt",
                       minimum_rows=getattr(args, "minimum_rows", 20))
    log("setup %.1fs: %s ; per-step counts %s" % (time.time() - t0, wl.shape, wl.counts))
    wl.run(args.warmup)
    if args.warmup:
        torch.cuda.synchronize()
        wl.count_digit_adds()
    barrier()
    engines = [eng] + ([wl.engw] if wl.engw is not eng else []) + ([wl.engn] if wl.engn is not eng else [])
    for e_ in engines:
        e_.timing_enable(True)
        e_.timing_reset()
    t0 = time.perf_counter()
    wl.run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    def tsum(which):
        a = [e_.timing_get(which) for e_ in engines]
        return sum(x[0] for x in a), sum(x[1] for x in a)

    acc_ms, acc_n = tsum(E.T_MSM_ACC)
    ntt_ms, ntt_n = tsum(E.T_NTT)
    trace_ms, trace_n = tsum(E.T_TRACE)
    msm_ms, msm_n = tsum(E.T_MSM_ALL)
    exp_ms, exp_n = tsum(E.T_EXPAND)
    # the dominant kernel alone (outside the timed region): in the timed region it shares the CUs with the NTT
    # kernels of the other stream, so its launch durations there include that time-slicing
    iso_ms = iso_n = 0
    if wl.stream_n is not None or wl.pipeline:
        eng.timing_reset()
        pl, wl.pipeline = wl.pipeline, False
        wl.consume(0, msm_only=True)
        wl.pipeline = pl
        barrier()
        iso_ms, iso_n = eng.timing_get(E.T_MSM_ACC)
    for e_ in engines:
        e_.timing_enable(False)
    # what the timed loop computed, checked (untimed): pipelined steps against a serial recomputation
    verification = None
    if not args.no_verify and rank == 0:
        try:
            verification = wl.verify_pipelined()
            log("verify_pipelined: %s" % {k_: v_ for k_, v_ in verification.items() if k_ not in ("note",)})
        except Exception as ex:
            verification = {"verified": False, "error": repr(ex)}
    tail = None
    body = None
    if args.scale == 1.0 and not args.no_tail:
        try:
            wl.tail_setup()
            if rank == 0 and world == 1:
                tail = wl.tail()
        except Exception as ex:   # never take the bench line down
            tail = {"error": repr(ex)}
    # second timed loop: the hot path AND the prover steps after it inside the timed region (same barrier / max-over-ranks
    # timing), fewer steps to keep the default run short
    if args.scale == 1.0 and not args.no_tail and not args.no_body and hasattr(wl, "_tail"):
        try:
            steps2 = max(1, args.steps // 2)
            wl.run(1, with_tail=True)
            barrier()
            t1 = time.perf_counter()
            wl.run(steps2, with_tail=True)
            barrier()
            dt2 = time.perf_counter() - t1
            tt2 = torch.tensor([dt2], dtype=torch.float64, device="cuda")
            if use_dist:
                dist.all_reduce(tt2, op=dist.ReduceOp.MAX)
            dt2 = float(tt2.item())
            body = {"value": steps2 * (1 if colpar else world) / dt2, "unit": "proofs/s", "steps": steps2, "ms_per_step": dt2 / steps2 * 1e3,
                    "connected": False,
                    "note": "WORK-EQUIVALENT STAND-IN, not a prover: the pipelined hot path + the later phases' kernels run for their WORK (the "
                            "quotient kernels over one stale 64-column tile with random selector / product inputs, the full-width commitments "
                            "over pool scalars, evaluations and SHPLONK over pool polynomials) -- instruction counts of a proof, not its dataflow. "
                            "The connected flow is the headline `value` (connected_line)"}
        except Exception as ex:
            body = {"error": repr(ex)}
    dropin = None
    if rank == 0 and world == 1 and args.scale == 1.0 and not args.no_dropin:
        try:
            dropin = dropin_host_pointer_path(wl, torch, log)
        except Exception as ex:
            dropin = {"error": repr(ex)}
    dropin_dev = None
    if rank == 0 and world == 1 and args.scale == 1.0 and not args.no_dropin and args.workload == "c2" and wl.pipeline and wl.stream_n is not None:
        try:
            torch.cuda.synchronize()
            dropin_dev = dropin_device_resident(wl, args, log)
        except Exception as ex:
            dropin_dev = {"error": repr(ex)}
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    if rank != 0:
        return None
    sh, cnt = wl.shape, wl.counts
    keygen = None
    fresh = None
    if args.scale == 1.0 and world == 1 and not args.no_tail:
        try:
            keygen = wl.keygen_vk_pk()
        except Exception as ex:
            keygen = {"error": repr(ex)}
        # What the reference's circuit costs a user who encrypts DISTINCT messages: paillier.rs:50-55 bakes the message's bits into
        # the circuit (pow_mod_fixed_exp), so every new m has its own shape, verifying key and proving key, and bench.rs:161-171
        # pays keygen -> prove once each.  Third timed loop: a new message per step, keygen_vk + keygen_pk of the circuit's
        # columns INSIDE the timed region, then the hot path of that proof (nothing of the previous proof to hide the witness under).
        if not args.no_fresh_key and "error" not in (keygen or {}):
            try:
                f_steps = 3
                wl.keygen_vk_pk()
                wl.run(1)
                barrier()
                tf = time.perf_counter()
                for _ in range(f_steps):
                    wl.keygen_vk_pk()
                    wl.run(1)
                barrier()
                dtf = time.perf_counter() - tf
                fresh = {"value": f_steps / dtf, "unit": "proofs/s", "steps": f_steps, "ms_per_step": dtf / f_steps * 1e3,
                         "note": "a new message per proof with the reference's circuit: its exponent bits are circuit structure (paillier.rs:50-55), so "
                                 "keygen_vk + keygen_pk (commitment, coefficient and extended forms of the fixed and permutation columns) run per "
                                 "proof, inside the timed region, before the hot path; `value` above amortises the key over proofs of one message "
                                 "shape, `c2u` (uniform-shape circuit, one key for all messages) is the amortisable design"}
            except Exception as ex:
                fresh = {"error": repr(ex)}
    proofs = args.steps * (1 if colpar else world)
    value = proofs / dt
    # roofline of the dominant kernel (k_msm_accumulate): algorithmic bytes per launch / avg launch time.
    # one launch accumulates nc columns against the shared bases: 64 B per base + 32 B per scalar
    n = 1 << args.k
    n_adv = wl.adv_cols + wl.lk_cols if args.scale == 1.0 else max(1, int(round(wl.adv_cols * args.scale))) + max(1, int(round(wl.lk_cols * args.scale)))
    total_cols = (n_adv + cnt["msm_full"]) * args.steps
    alg_bytes_total = total_cols * n * 32.0 + acc_n * n * 64.0
    ach = alg_bytes_total / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
    # measured HBM-side traffic of the dominant kernel: PMC counters cannot be collected from inside this process,
    # so the per-launch figure of the committed rocprofv3 --pmc passes of this same command is reported
    traffic, traffic_src = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")))
        if args.scale == 1.0 and (args.enc_bits, args.k) == (2048, 17):
            traffic = pj["k_msm_accumulate"]["fetch_bytes_per_launch_raw"] + pj["k_msm_accumulate"]["write_bytes_per_launch"]
            traffic_src = pj["source"]
    except Exception:
        pass
    out = {
        "metric": ("Paillier-encrypt proofs/s (2048-bit n, k=17); MSM achieved HBM GB/s vs peak" if (args.workload == "c2" and (args.enc_bits, args.k) == (2048, 17))
                   else "Paillier-%s proofs/s (%d-bit n, k=%d) -- NOT the headline configuration; MSM achieved HBM GB/s vs peak"
                   % ("add" if args.workload == "c3" else "encrypt (uniform-shape circuit)" if args.workload == "c2u" else "encrypt", args.enc_bits, args.k)),
        "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if colpar else "weak", "vs_baseline": None,
        "dtype": "u32 limbs (29-bit reduced radix, 254-bit modular integers), u64 limbs (4096-bit integers)", "data": "synthetic",
        "config": {
            "workload": "%s: %d-bit n, KZG prover hot path at k=%d (K3 trace + K4 cell expansion + K1 commitments + K2 NTTs), 1 proof per GPU per step"
                        % ("c3 homomorphic add" if args.workload == "c3" else "c2u uniform-shape encrypt" if args.workload == "c2u" else "c2 encrypt" if (args.enc_bits, args.k) == (2048, 17) else "c5-shape encrypt" if (args.enc_bits, args.k) == (3072, 19) else "custom encrypt", args.enc_bits, args.k),
            "enc_bits": args.enc_bits, "k": args.k, "lookup_bits": sh.lookup_bits, "limb_bits": 64,
            "minimum_rows": wl.row_budget.minimum_rows, "max_rows": wl.rows,
            "mul_mod_steps": wl.n_steps, "advice_cols": sh.advice_cols, "lookup_cols": sh.lookup_cols,
            "perm_cols": sh.perm_cols, "advice_cols_committed": wl.adv_cols, "lookup_cols_committed": wl.lk_cols,
            "cells_per_mul_mod": wl.cells, "advice_cells": wl.n_steps * wl.cells, "msm_per_proof": n_adv + cnt["msm_full"],
            "ntt_polys_per_proof": cnt["polys"], "scale": args.scale,
            "scope": "hot path only (SURVEY section 8a): excludes the prover steps after it (products, evaluate_h, evaluations, openings -- all inside the connected proof, the headline `value`) and the transcript",
            "layout_parity": "unpinned: cell patterns and column counts restate the biguint-halo2 / halo2-lib dependencies (SURVEY tag [D]); emitted: assign_integer x5, square, refresh, load_zero, pow_mod constants, every mul_mod, assert_equal_fresh (the whole driver, row a6); omitted: nothing of the driver; blinding rows are left zero",
            "advice_cells_whole_circuit": wl.circ_adv, "lookup_cells_whole_circuit": wl.circ_lk,
            "parallelism": ("one proof, columns split over the ranks, all-gather of the commitments" if colpar
                            else "proof replicas, one per GPU, no collective"), "pipeline_witness_of_next_proof": wl.pipeline,
            "ntt_on_second_stream": wl.stream_n is not None,
        },
        "roofline": {
            "bound": "hbm", "kernel": "k_msm_accumulate", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes_total / max(1, acc_n),
            "launches": int(acc_n), "avg_launch_ms": acc_ms / max(1, acc_n),
            "alone": ({"avg_launch_ms": iso_ms / iso_n, "achieved": alg_bytes_total / args.steps / (iso_ms * 1e-3) / 1e9,
                       "frac": alg_bytes_total / args.steps / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "note": "same launches of one proof re-run after the timed region with nothing else on the GPU; "
                               "in the timed region the kernel shares the CUs with the NTT stream"} if iso_n else None),
            "note": "integer-multiply-issue bound by construction (v_mad_u64_u32); HBM fraction is the metric's definition, see DESIGN.md section 5",
        },
        # the roofline that actually binds this kernel: multiplier-instruction issue (v_mad_u64_u32 / v_mul_lo_u32).  One mixed
        # addition = 8 products x 180 + 2 squares x 135 multiplier instructions of the 29-bit field (fp29.cuh); peak = the
        # independent-multiplicand mad issue rate measured live by libpz_probe.so's pzp_ubench_mad_indep
        "roofline_int": {
            "bound": "v_mad_u64_u32 issue", "kernel": "k_msm_accumulate",
            "achieved": (wl.digit_adds * args.steps * MULT_PER_MADD / (acc_ms * 1e-3) / 1e12) if (acc_ms > 0 and wl.digit_adds) else None,
            "achieved_alone": (wl.digit_adds * MULT_PER_MADD / (iso_ms * 1e-3) / 1e12) if (iso_n and wl.digit_adds) else None,
            "peak": mad_peak, "unit": "T multiplier instructions/s", "digit_adds_per_proof": wl.digit_adds,
            "multiplier_instructions_per_mixed_addition": MULT_PER_MADD,
            "note": "digit_adds = non-zero 16-bit digits accumulated per proof, estimated on the device from the canonical cell values (signed-recoding carries and negated cells approximated); peak measured in this run",
        },
        # per-class HIP-event sums; with the witness / NTT streams on, the classes overlap in time (sum > ms_per_step)
        "breakdown_ms_per_proof": {"trace": trace_ms / args.steps, "expand": exp_ms / args.steps, "msm_all": msm_ms / args.steps,
                                   "msm_accumulate": acc_ms / args.steps, "ntt": ntt_ms / args.steps},
    }
    # what RCCL itself reports (not the launcher's environment): the rank count of the process group the timing used
    out["rccl_ranks"] = dist.get_world_size() if use_dist else 1
    out["backend"] = dist.get_backend() if use_dist else None
    # every PZ_* switch set for this run is echoed; anything that removes or resizes work inside the timed region
    # (PZ_BENCH_SKIP, --scale) marks the line as not comparable
    out["config"]["env_switches"] = env_switches()
    out["comparable"] = bool(args.scale == 1.0 and not os.environ.get("PZ_BENCH_SKIP") and not os.environ.get("PZ_BENCH_DROP_EDGE"))
    # `verified`: the pipelined step's outputs equal a serial recomputation's (verify_pipelined); a failed check voids the line
    out["verified"] = verification.get("verified") if verification else None
    if verification is not None:
        out["verification"] = verification
        if verification.get("verified") is False:
            out["comparable"] = False
    out["srs_ms"] = wl.srs_ms
    out["config"]["srs"] = "params file written as gen_srs does (ParamsKZG::setup), read back by paillier_halo2_amd/srs.py, points checked on the device, Lagrange bases by the G1 inverse FFT (no toxic scalar)"
    out["config"]["ntt_inputs"] = ("the proof's own advice and lookup columns (transformed out of place into the coefficient buffer) + pool polynomials for the rest" if wl.own_ntt else "pool polynomials")
    if keygen is not None:
        out["keygen"] = keygen
    if fresh is not None:
        out["fresh_key"] = fresh
    if tail is not None:
        out["next_rows_ms_per_proof"] = tail
    if body is not None:
        out["with_next_rows_emulated"] = body
    if dropin is not None:
        out["dropin_host_pointer"] = dropin
    if dropin_dev is not None:
        out["dropin_device_resident"] = dropin_dev
        if dropin_dev.get("value"):
            out["dropin_device_resident"]["ratio_to_value"] = dropin_dev["value"] / value
        # the C++ caller's commitments are the Python path's: the per-message hashes of the two verification runs agree
        if verification and verification.get("commitment_hash_by_message") and dropin_dev.get("commitment_hash_by_message"):
            mine, theirs = verification["commitment_hash_by_message"], dropin_dev["commitment_hash_by_message"]
            common = sorted(set(mine) & set(theirs))
            dropin_dev["commitments_equal_python_path"] = bool(common) and all(mine[c_] == theirs[c_] for c_ in common)
            if dropin_dev.get("verified") is False or not dropin_dev["commitments_equal_python_path"]:
                out["comparable"] = False
    if not args.no_cpu_baseline and args.scale == 1.0 and world == 1:   # rank 0 at N = 1 only
        want_check = bool(verification and verification.get("verified") and wl.circuit == "encrypt")
        try:
            out["cpu_baseline"] = cpu_baseline(sh, wl.n_steps, args.enc_bits, args.k, log, check_wl=wl if want_check else None)
        except Exception as ex:  # the checker must never take the bench line down
            out["cpu_baseline"] = {"value": None, "error": repr(ex)}
        # the leg's checker result belongs to the verification: the serial reference of verify_pipelined against the oracle chain
        vo = out["cpu_baseline"].pop("checker", None)
        if vo is not None:
            out["verification"]["vs_oracle"] = vo
            if vo.get("ok") is False:
                out["verified"] = out["verification"]["verified"] = False
                out["comparable"] = False
    if out["roofline_int"]["achieved"]:
        out["roofline_int"]["frac"] = out["roofline_int"]["achieved"] / out["roofline_int"]["peak"]
        if out["roofline_int"]["achieved_alone"]:
            out["roofline_int"]["frac_alone"] = out["roofline_int"]["achieved_alone"] / out["roofline_int"]["peak"]
    # the amortisable design beside it (default c2 run only): the same key and message through the UNIFORM-shape circuit (SURVEY 8f
    # rank 4: g^m over all message bits in circuit) -- one verifying / proving key serves every message.  A second workload, built
    # after the first one's memory is released; same timing rules, its own verification.
    if (args.workload == "c2" and (args.enc_bits, args.k) == (2048, 17) and args.scale == 1.0 and world == 1 and not args.no_c2u
            and not os.environ.get("PZ_BENCH_SKIP")):
        try:
            seed_u, pool_u, lb_u = args.seed, wl.pool, args.lookup_bits
            wl.release()
            del wl
            import gc

            gc.collect()
            torch.cuda.empty_cache()
            t_u = time.time()
            wu = ProofWorkload(eng, torch, args.enc_bits, args.k, seed=seed_u, scale=1.0, pool=pool_u, lookup_bits=lb_u, circuit="encrypt_uniform")
            wu.run(1)
            barrier()
            u_steps = max(2, args.steps // 2)
            tu = time.perf_counter()
            wu.run(u_steps)
            barrier()
            dtu = time.perf_counter() - tu
            vu = None if args.no_verify else wu.verify_pipelined()
            out["c2u"] = {"value": u_steps / dtu, "unit": "proofs/s", "steps": u_steps, "ms_per_step": dtu / u_steps * 1e3,
                          "mul_mod_steps": wu.n_steps, "advice_cols_committed": wu.adv_cols, "lookup_cols_committed": wu.lk_cols,
                          "msm_per_proof": wu.adv_cols + wu.lk_cols + wu.counts["msm_full"], "ntt_polys_per_proof": wu.counts["polys"],
                          "verified": vu.get("verified") if vu else None,
                          "verification": {k_: v_ for k_, v_ in (vu or {}).items() if k_ in ("pipelined_steps_checked", "commitments_compared", "transforms_compared", "mismatches", "async_error")},
                          "note": "uniform-shape encrypt circuit at the c2 key size (pz_paillier_encrypt_uniform_dev + circuit kind 2): hot path only, "
                                  "same definition as `value`; one key for all messages, so this is what a stream of DISTINCT messages gets once the key exists"}
            log("c2u %.1fs: %.3f proofs/s" % (time.time() - t_u, u_steps / dtu))
            wu.release()
        except Exception as ex:
            out["c2u"] = {"error": repr(ex)}
    return out


def count_digit_adds_cols(eng, torch, cols, n_cols, n, full_cols):
    """non-zero signed 16-bit digits K1 accumulates for one proof: the witness columns `cols` [n_cols][n][4] (Montgomery, as K4 wrote them;
    counted on the device from the canonical values) + full-width columns at 16 digits each (measurement support for roofline_int)"""
    total = 0
    flat = cols[:n_cols].reshape(-1, 4)
    CH = 1 << 22
    for c0 in range(0, flat.shape[0], CH):
        x = flat[c0:c0 + CH].clone()
        eng.fr_convert_dev(x.data_ptr(), x.shape[0], False)
        eng.sync()
        neg = x[:, 3] != 0
        nz = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device)
        for limb in range(3):
            v = x[:, limb]
            for sh in (0, 16, 32, 48):
                nz += ((v >> sh) & 0xFFFF) != 0
        total += int(nz[~neg].sum().item()) + int(neg.sum().item()) * 9
        del x
    return int(total + full_cols * n * 16 * (1.0 - 2.0 ** -16))


def connected_line(args, R):
    """THE HEADLINE: one CONNECTED proof per step (bench_connected.ConnectedWorkload: K3 -> K4 in halo2-lib's break-point columns -> advice
    commitments -> permuted lookup columns -> grand products -> quotient -> evaluations -> SHPLONK, five transcript round trips, every phase
    on the proof's own data; reference: /root/reference/src/bench.rs:161-171 gen_proof after keygen), checked after the timed loop as the
    verifier would.  BASELINE.json config 2: "full KZG proof at k=17".  -> the JSON dict on rank 0, None elsewhere"""
    import gc

    import torch
    import torch.distributed as dist

    import bench_connected
    from paillier_halo2_amd import engine as E

    rank, world, use_dist, log, eng, mad_peak, barrier = R.rank, R.world, R.use_dist, R.log, R.eng, R.mad_peak, R.barrier
    t_all = time.time()
    preset = args.workload
    streamed, pipeline, lookup_tile = None, None, None
    if preset == "c5":      # BASELINE config c5: 3072-bit n, k = 19, 64 independent proofs over 8 GPUs = replicas; its extended proving key (239 GB)
        args.enc_bits, args.k = 3072, 19          # does not fit beside the rest: the STREAMED key (coefficient forms resident, tiles re-extended)
        if "--steps" not in sys.argv:
            args.steps = 3
        streamed, pipeline, lookup_tile = "auto", False, 16        # memory_plan: R = 0 there (its extended key alone would be 239 GB)
    if preset == "c3":      # BASELINE config c3: the homomorphic-add circuit (PaillierChip::add, paillier.rs:62-85; bench.rs:77-117) at k = 15
        if args.k == 17:
            args.k = 15
    if os.environ.get("PZ_CONNECTED_STREAMED_KEY", "") != "":
        sk_ = os.environ["PZ_CONNECTED_STREAMED_KEY"]
        streamed = sk_ if sk_ == "auto" else int(sk_)
    elif args.k >= 18 and streamed is None:
        streamed = "auto"       # shapes beyond c2: keep what fits of the extended key, stream the rest (ConnectedWorkload.memory_plan)
    circuit = "encrypt_uniform" if preset == "c2u" else "add" if preset == "c3" else "encrypt"
    headline_cfg = preset == "c2" and (args.enc_bits, args.k) == (2048, 17)
    cw = bench_connected.ConnectedWorkload(eng, torch, args.enc_bits, args.k, args.seed + rank, lookup_bits=args.lookup_bits, log=log, circuit=circuit,
                                           minimum_rows=args.minimum_rows, streamed_key=streamed, pipeline=pipeline, lookup_tile=lookup_tile)
    cnt = cw.counts()
    log("connected setup %.1fs: structure %s ms, keygen %.0f ms, %s" % (time.time() - t_all, {k_: round(v_) for k_, v_ in cw.structure_ms.items()}, cw.keygen_ms, cnt))
    # digits K1 accumulates per proof (roofline_int): counted on the first witness, which the first proof then consumes
    cw.produce()
    torch.cuda.synchronize()
    digit_adds = count_digit_adds_cols(eng, torch, cw.slots[0], cw.A + cw.Lk, cw.n, cnt["msm_full"])
    cw.run(max(1, args.warmup), timed=False)      # (the first proof also grows the library's workspaces)
    barrier()
    engines = [eng] + ([cw.engw] if cw.engw is not eng else [])
    for e_ in engines:
        e_.timing_enable(True)
        e_.timing_reset()
    t0 = time.perf_counter()
    cw.run(args.steps, timed=False)
    barrier()
    dt = time.perf_counter() - t0
    tsum = lambda which: tuple(sum(x) for x in zip(*[e_.timing_get(which) for e_ in engines]))
    acc_ms, acc_n = tsum(E.T_MSM_ACC)
    ntt_ms, ntt_n = tsum(E.T_NTT)
    trace_ms, _ = tsum(E.T_TRACE)
    msm_ms, _ = tsum(E.T_MSM_ALL)
    exp_ms, _ = tsum(E.T_EXPAND)
    for e_ in engines:
        e_.timing_enable(False)
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    if rank != 0:
        cw.release()
        return None
    cw.run(1, timed=True)       # one more proof, synchronised phase by phase, OUTSIDE the timed region: where the time goes
    ver, _cref = None, None
    if not args.no_verify and not args.no_cpu_baseline:
        from oracle import cref as _cref     # the checker leg (the one place bench.py may touch oracle/)

        _cref.build()
        ver = cw.verify(_cref)
    n = 1 << args.k
    value = args.steps * world / dt
    cols_per_proof = cnt["msm_witness"] + cnt["msm_full"]
    alg_bytes_total = cols_per_proof * args.steps * n * 32.0 + acc_n * n * 64.0
    ach = alg_bytes_total / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
    traffic, traffic_src = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic_connected.json")))
        if headline_cfg:
            traffic = pj["k_msm_accumulate"]["fetch_bytes_per_launch_raw"] + pj["k_msm_accumulate"]["write_bytes_per_launch"]
            traffic_src = pj["source"]
    except Exception:
        pass
    shape_name = ("c2 encrypt" if headline_cfg else "c2u uniform-shape encrypt" if preset == "c2u" else "c3 homomorphic add" if preset == "c3"
                  else "c5-shape encrypt" if (args.enc_bits, args.k) == (3072, 19) else "custom encrypt")
    out = {
        "metric": ("Paillier-encrypt proofs/s (2048-bit n, k=17); MSM achieved HBM GB/s vs peak" if headline_cfg
                   else "Paillier-%s proofs/s (%d-bit n, k=%d) -- NOT the headline configuration; MSM achieved HBM GB/s vs peak"
                   % ("add" if preset == "c3" else "encrypt (uniform-shape circuit)" if preset == "c2u" else "encrypt", args.enc_bits, args.k)),
        "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 limbs (29-bit reduced radix, 254-bit modular integers), u64 limbs (4096-bit integers)", "data": "synthetic",
        "config": {
            "workload": "%s: %d-bit n, ONE CONNECTED KZG proof per GPU per step at k=%d (K3 trace -> K4 break-point columns -> create_proof's phases in order, "
                        "a transcript round trip per phase)" % (shape_name, args.enc_bits, args.k),
            "scope": "one connected proof (keygen once per key and message shape, outside: keygen_ms; the transcript is halo2's Blake2b transcript in its "
                     "primitives and round trips -- BLAKE2b-512, its personalisation, domain bytes and wide-reduced challenges -- over Montgomery words, "
                     "not its byte format; the checker re-derives every challenge from the proof)",
            "enc_bits": args.enc_bits, "k": args.k, "lookup_bits": cw.lb, "limb_bits": 64, "mul_mod_steps": cw.n_steps,
            "minimum_rows": cw.minimum_rows, "max_rows": cw.cs.max_rows, "advice_cols": cw.A, "advice_cols_filled": cw.cs.n_adv_used,
            "lookup_cols": cw.Lk, "permutation_cols": cw.m, "permutation_sets": cw.pk.n_sets,
            "row_budget": "halo2-lib's tester [D]: calculate_params(Some(minimum_rows)) fixes the column COUNT (20 on the reference's bench path, bench.rs:161-171); "
                          "columns are filled to 2^k - cs.minimum_rows() = 2^k - 9 (paillier_halo2_amd/layout.py RowBudget)",
            "msm_per_proof": cols_per_proof, "ntt_polys_per_proof": cnt["ntt_polys"], "polys_opened": cnt["polys_opened"],
            "quotient_domain_cosets": cw.pk.dom.cosets, "proving_key_streamed": bool(cw.pk.streamed),
            "pipeline_witness_of_next_proof": bool(cw.pipeline),
            "prover": "paillier_halo2_amd/prover.py (Python driver over the C ABI; the compiled driver's figure: compiled_prover)",
            "layout_parity": "unpinned: cell patterns, break points and column counts restate the biguint-halo2 / halo2-lib dependencies (SURVEY tag [D])",
            "parallelism": "proof replicas, one per GPU, no collective", "scale": 1.0,
        },
        "verified": ver.get("verified") if ver else None, "verification": ver,
        "comparable": bool(not os.environ.get("PZ_BENCH_SKIP")),
        "roofline": {
            "bound": "hbm", "kernel": "k_msm_accumulate", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes_total / max(1, acc_n), "launches": int(acc_n), "avg_launch_ms": acc_ms / max(1, acc_n),
            "launches_per_proof": acc_n / max(1, args.steps),
            "note": "the connected loop's own k_msm_accumulate launches (HIP events on the library's stream): 32 B per scalar of every committed column + "
                    "64 B per base per launch; integer-multiply-issue bound by construction (v_mad_u64_u32) -- the HBM fraction is the metric's definition",
        },
        "roofline_int": {
            "bound": "v_mad_u64_u32 issue", "kernel": "k_msm_accumulate",
            "achieved": (digit_adds * args.steps * MULT_PER_MADD / (acc_ms * 1e-3) / 1e12) if acc_ms > 0 else None,
            "peak": mad_peak, "unit": "T multiplier instructions/s", "digit_adds_per_proof": digit_adds,
            "multiplier_instructions_per_mixed_addition": MULT_PER_MADD,
        },
        "breakdown_ms_per_proof": {"trace": trace_ms / args.steps, "expand": exp_ms / args.steps, "msm_all": msm_ms / args.steps,
                                   "msm_accumulate": acc_ms / args.steps, "ntt": ntt_ms / args.steps, "ntt_calls": ntt_n / args.steps},
        "phases_ms_per_proof": cw.phase_ms(1), "counts": cnt, "memory_gb": cw.memory_gb,
        "keygen_ms": cw.keygen_ms, "circuit_structure_ms": cw.structure_ms,
        "rccl_ranks": dist.get_world_size() if use_dist else 1, "backend": dist.get_backend() if use_dist else None,
    }
    out["memory_gb"]["torch_allocated_peak"] = torch.cuda.max_memory_allocated() / 1e9
    try:
        free_b, total_b = torch.cuda.mem_get_info()
        out["memory_gb"]["device_total"], out["memory_gb"]["device_free_after_timed_loop"] = total_b / 1e9, free_b / 1e9
    except Exception:
        pass
    if out["roofline_int"]["achieved"]:
        out["roofline_int"]["frac"] = out["roofline_int"]["achieved"] / mad_peak
    out["config"]["env_switches"] = env_switches()
    if ver is not None and ver.get("verified") is False:
        out["comparable"] = False
    log("connected %.1fs: %.3f proofs/s (%.1f ms), verified %s" % (time.time() - t_all, value, dt / args.steps * 1e3, out["verified"]))
    if not args.no_cpu_baseline and world == 1:
        try:
            cc = dict(cnt)
            cc["n"] = n
            out["cpu_baseline"] = cpu_baseline(None, cw.n_steps, args.enc_bits, args.k, log, connected=cc)
            out["cpu_baseline"].pop("checker", None)
        except Exception as ex:
            out["cpu_baseline"] = {"value": None, "error": repr(ex)}
    extras = world == 1 and not args.headline_only and not os.environ.get("PZ_BENCH_SKIP")
    # ---- a NEW MESSAGE per proof with the reference's circuit: its bits are circuit structure (paillier.rs:50-55), so every step generates
    # the structure, runs keygen_vk + keygen_pk on its real selectors / sigma and then the connected proof.  The SRS is shared.
    if extras and headline_cfg and not args.no_fresh_key:
        try:
            srs = (cw.bl, cw.bm, cw.s_tox)
            for nm in ("pk", "ws", "slots", "cols", "d_steps"):
                setattr(cw, nm, None)           # the first key leaves the device; its SRS stays
            gc.collect()
            fm_steps, t_f, verf, last = 3, time.perf_counter(), None, None
            parts = {"structure_ms": 0.0, "keygen_ms": 0.0}
            for i in range(fm_steps):
                if last is not None:      # the previous key's blocks go back to torch's allocator, not to the driver: the next key reuses them
                    last.release(trim=False)
                    del last
                    gc.collect()
                last = bench_connected.ConnectedWorkload(eng, torch, args.enc_bits, args.k, args.seed + 7001 + i, lookup_bits=args.lookup_bits,
                                                         log=log, pipeline=False, srs=srs, trim=False, minimum_rows=args.minimum_rows)
                last.run(1, timed=False)
                parts["structure_ms"] += sum(last.structure_ms.values()) / fm_steps
                parts["keygen_ms"] += last.keygen_ms / fm_steps
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t_f
            if ver is not None:
                verf = last.verify(_cref)
            out["fresh_message"] = {
                "value": fm_steps / dtf, "unit": "proofs/s", "steps": fm_steps, "s_per_step": dtf / fm_steps, "connected": True,
                "verified": verf.get("verified") if verf else None, "verification": verf, "of_which": parts,
                "note": "every step: a new key pair and message -> circuit structure generated on the device (circuit_structure.py) -> "
                        "keygen_vk + keygen_pk of its real selectors and sigma (all three forms resident) -> one connected proof; "
                        "the whole step is inside the timed region, incl. the release of the previous key; SRS shared"}
            log("fresh message: %.2f s per step (structure %.0f + keygen %.0f ms), verified %s" % (
                dtf / fm_steps, parts["structure_ms"], parts["keygen_ms"], verf.get("verified") if verf else None))
            if verf is not None and verf.get("verified") is False:
                out["comparable"] = False
            last.release()
            del last
        except Exception as ex:
            import traceback

            out["fresh_message"] = {"error": repr(ex)[:400], "trace": traceback.format_exc()[-600:]}
    cw.release()
    gc.collect()
    torch.cuda.empty_cache()
    # ---- the same proof from the COMPILED prover: plain C++ over the C ABI, a child process with its own contexts
    if extras and not args.no_dropin and preset not in ("c5", "c3"):
        try:
            t_p = time.time()
            out["compiled_prover"] = bench_connected.cpp_connected(cw, proofs=5, verify_with=_cref if ver is not None else None, log=log)
            log("connected, compiled prover %.1fs: %.1f ms per proof, verified %s" % (
                time.time() - t_p, out["compiled_prover"]["ms_per_step"], out["compiled_prover"].get("verified")))
            if out["compiled_prover"].get("verified") is False:
                out["comparable"] = False
            # ... and through the LIBRARY'S stepper (pz_pk_create + pz_proof_*) with the next witness on a second thread and context
            out["compiled_stepper"] = bench_connected.cpp_connected(cw, proofs=5, verify_with=_cref if ver is not None else None, log=log, via_stepper=True)
            log("connected, library stepper from compiled code: %.1f ms per proof (pz_pk_create %.0f ms), verified %s" % (
                out["compiled_stepper"]["ms_per_step"], out["compiled_stepper"]["keygen_ms"], out["compiled_stepper"].get("verified")))
            if out["compiled_stepper"].get("verified") is False:
                out["comparable"] = False
        except Exception as ex:
            out["compiled_prover"] = dict(out.get("compiled_prover") or {}, error=repr(ex)[:600])
    del cw
    gc.collect()
    # ---- ... and a NEW KEY AND MESSAGE per proof from compiled code alone: structure generated by the library on the device
    if extras and headline_cfg and not args.no_dropin and not args.no_fresh_key:
        try:
            t_p = time.time()
            out["fresh_message_cpp"] = bench_connected.cpp_fresh_message(args.enc_bits, args.k, args.lookup_bits, args.seed, steps=4, minimum_rows=args.minimum_rows,
                                                                         verify_with=_cref if ver is not None else None, log=log)
            log("fresh message, compiled prover %.1fs: %.2f s per step %s, verified %s" % (
                time.time() - t_p, out["fresh_message_cpp"]["s_per_step"], out["fresh_message_cpp"]["of_which"], out["fresh_message_cpp"].get("verified")))
            if out["fresh_message_cpp"].get("verified") is False:
                out["comparable"] = False
        except Exception as ex:
            out["fresh_message_cpp"] = {"error": repr(ex)[:600]}
    # ---- config c5: bench.rs:161-171's own flow (keygen + proof per run) from compiled code -- a step fills the GPU (250 GB of key, workspace and
    # witness), so this process first gives back the library's workspaces; the binary carves its memory out of one arena (pz_dev_arena)
    if extras and preset == "c5" and not args.no_dropin and not args.no_fresh_key:
        try:
            t_p = time.time()
            eng.close()
            gc.collect()
            torch.cuda.empty_cache()
            out["fresh_message_cpp"] = bench_connected.cpp_fresh_message(args.enc_bits, args.k, args.lookup_bits, args.seed, steps=3, minimum_rows=args.minimum_rows,
                                                                         verify_with=_cref if ver is not None else None, log=log, streamed_key=0)
            log("fresh message, compiled prover %.1fs: %.2f s per step %s, verified %s, %s" % (
                time.time() - t_p, out["fresh_message_cpp"]["s_per_step"], out["fresh_message_cpp"]["of_which"], out["fresh_message_cpp"].get("verified"),
                out["fresh_message_cpp"].get("device_memory")))
            if out["fresh_message_cpp"].get("verified") is False:
                out["comparable"] = False
        except Exception as ex:
            out["fresh_message_cpp"] = {"error": repr(ex)[:600]}
    # ---- the same through the UNIFORM-shape circuit (row f4): ONE proving key for every message of a key -- the proofs of this loop are
    # of DISTINCT messages (the reference's circuit needs a new structure + keygen per message: `fresh_message`)
    if extras and headline_cfg and not args.no_c2u:
        try:
            t_u = time.time()
            c_steps = max(2, args.steps // 2)
            cu = bench_connected.ConnectedWorkload(eng, torch, args.enc_bits, args.k, args.seed, lookup_bits=args.lookup_bits, log=log,
                                                   circuit="encrypt_uniform", minimum_rows=args.minimum_rows)
            cu.run(1, timed=False)
            barrier()
            tu0 = time.perf_counter()
            cu.run(c_steps, timed=False)
            barrier()
            dtu = time.perf_counter() - tu0
            cu.run(1, timed=True)
            veru = cu.verify(_cref) if ver is not None else None
            out["uniform_circuit_distinct_messages"] = {
                "value": c_steps / dtu, "unit": "proofs/s", "steps": c_steps, "ms_per_step": dtu / c_steps * 1e3, "connected": True,
                "distinct_messages_one_key": True, "verified": veru.get("verified") if veru else None, "verification": veru,
                "phases_ms_per_proof": cu.phase_ms(1), "counts": cu.counts(), "memory_gb": cu.memory_gb, "keygen_ms": cu.keygen_ms,
                "circuit_structure_ms": cu.structure_ms,
                "note": "the connected proof through the uniform-shape circuit (g^m over all message bits in circuit): every proof of the loop is of "
                        "a different message, all under one proving key"}
            log("connected c2u %.1fs: %.3f proofs/s, verified %s" % (time.time() - t_u, c_steps / dtu, veru.get("verified") if veru else None))
            if veru is not None and veru.get("verified") is False:
                out["comparable"] = False
            cu.release()
            del cu
            gc.collect()
            torch.cuda.empty_cache()
        except Exception as ex:
            import traceback

            out["uniform_circuit_distinct_messages"] = {"error": repr(ex)[:400], "trace": traceback.format_exc()[-600:]}
    # ---- the HOT PATH ONLY (rounds 1-5's headline): K3 -> K4 -> K1 + K2 with the later phases' commitments / transforms over pool scalars,
    # three-stream pipeline, checked against a serial recomputation; kept beside the connected proof for continuity
    if extras and headline_cfg and not args.no_hot_path:
        try:
            t_h = time.time()
            a2 = argparse.Namespace(**vars(args))
            a2.workload, a2.steps, a2.warmup = "c2", min(args.steps, 6), 1
            a2.no_tail = a2.no_body = a2.no_fresh_key = a2.no_c2u = a2.no_cpu_baseline = True
            a2.no_dropin = not args.full_hot_path
            hp = hot_path_line(a2, R)
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "verified", "verification", "comparable", "roofline", "roofline_int",
                    "breakdown_ms_per_proof", "srs_ms", "dropin_device_resident", "dropin_host_pointer")
            out["hot_path_only"] = {k_: hp[k_] for k_ in keep if k_ in hp}
            out["hot_path_only"]["config"] = {k_: hp["config"][k_] for k_ in ("workload", "scope", "advice_cols_committed", "lookup_cols_committed", "msm_per_proof",
                                                                                "ntt_polys_per_proof", "pipeline_witness_of_next_proof", "ntt_on_second_stream") if k_ in hp["config"]}
            log("hot path only %.1fs: %.3f proofs/s, verified %s" % (time.time() - t_h, hp["value"], hp.get("verified")))
            if hp.get("verified") is False:
                out["comparable"] = False
        except Exception as ex:
            import traceback

            out["hot_path_only"] = {"error": repr(ex)[:400], "trace": traceback.format_exc()[-600:]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c2u", "c3", "c5", "msm22", "stub"],
                    help="c2: ONE CONNECTED encrypt proof per step (headline); c2u: the same key size through the uniform-shape circuit (g^m over all message bits in circuit, SURVEY 8f rank 4); "
                         "c3: homomorphic-add circuit at --k 15; c5: 3072-bit n, k = 19 (streamed proving key); msm22: one sharded MSM; each of c2 / c2u / c3 / c5 as a connected proof "
                         "(--hot-path-headline: the hot path only)")
    ap.add_argument("--k", type=int, default=17)
    ap.add_argument("--enc-bits", type=int, default=2048)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the per-proof MSM/NTT counts (debug only; "
                    "any value != 1 marks the line as not comparable)")
    ap.add_argument("--log-n", type=int, default=22, help="msm22 workload: log2 of the MSM size")
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0x5043,
                    help="base seed of the synthetic key / message (rank r uses seed + r); default = SURVEY section 8d's c2 seed")
    ap.add_argument("--lookup-bits", type=int, default=None, help="RangeChip lookup bits (default k - 1, the reference's pattern)")
    ap.add_argument("--msm-scalars", default="uniform", choices=["uniform", "witness"],
                    help="msm22 workload: uniform scalars, or SURVEY section 8d's witness-like mix (60%% < 2^16, 30%% < 2^64, 10%% < 2^135)")
    ap.add_argument("--parallel", default="replicas", choices=["replicas", "columns"],
                    help="N > 1: independent proofs per GPU (weak scaling, the default and the headline) or ONE proof whose "
                         "columns are split over the ranks with an all-gather of the commitments (strong scaling)")
    ap.add_argument("--msm-split", default="points", choices=["windows", "points"],
                    help="msm22 workload: shard point ranges across the ranks (default: measured / emulated to scale better, "
                         "DESIGN.md section 8) or Pippenger windows (north_star's split)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-pointer (drop-in binding) measurement")
    ap.add_argument("--no-body", action="store_true", help="skip the second timed loop (hot path + the prover steps after it)")
    ap.add_argument("--emulate-world", type=int, default=0, help="msm22 workload on ONE GPU: run each of W ranks' shares in turn, "
                    "print per-share stage times and the predicted W-GPU efficiency for both splits")
    ap.add_argument("--emulate-ranks", type=int, default=0, help="msm22 workload (and, with --parallel columns, a c2 column batch) on ONE GPU: this "
                    "process plays every rank 0..W-1 of a world of W in turn -- rank r's exact code path (its point / window / column range, "
                    "offsets, padded gather slices) -- through the real process group of one rank (use with --force-dist: backend nccl = RCCL); "
                    "every rank's folded / gathered result must equal the single-call result")
    ap.add_argument("--no-verify", action="store_true", help="skip the output check of the pipelined step after the timed loop (the line then says verified: null)")
    ap.add_argument("--no-fresh-key", action="store_true", help="skip the third timed loop (keygen per message inside the timed region)")
    ap.add_argument("--no-c2u", action="store_true", help="default c2 run only: skip the uniform-shape circuit's line (a second workload after the main one)")
    ap.add_argument("--no-tail", action="store_true", help="skip the (untimed) measurement of the prover steps after the hot path")
    ap.add_argument("--force-dist", action="store_true", help="with --gpus 1: still start through torch.distributed.run, create the process group "
                    "(backend nccl = RCCL) and run every collective of the N > 1 path on the one rank (all_gather_into_tensor + device fold of msm22, "
                    "the commitment all-gather of --parallel columns, the barriers and the MAX all-reduce of the timing)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo only with --workload stub)")
    ap.add_argument("--minimum-rows", type=int, default=20, help="the argument of halo2-lib's calculate_params: 20 on the reference's bench path "
                    "(bench.rs:161-171 -> bench_builder), 9 under MockProver (paillier.rs:167-171); fixes the column COUNT (layout.RowBudget)")
    ap.add_argument("--hot-path-headline", action="store_true", help="rounds 1-5's line: `value` = the hot path only (K3 + K4 + K1 + K2), not the connected proof")
    ap.add_argument("--headline-only", action="store_true", help="the connected proof's timed loop, its check and the CPU baseline only (no fresh_message / "
                    "compiled_prover / uniform circuit / hot_path_only legs)")
    ap.add_argument("--no-hot-path", action="store_true", help="skip the hot_path_only leg")
    ap.add_argument("--full-hot-path", action="store_true", help="hot_path_only leg: also the C++ hot-path caller and the host-pointer binding")
    args = ap.parse_args()

    # typed as `python bench.py --gpus N` (not pre-launched by torch.distributed.run): become the launcher.  Nothing above
    # this line has touched the GPU (numpy only), and the ranks are CHILD processes -- never an exec of this one
    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.workload == "stub":
        return stub_workload(args)

    import torch
    import torch.distributed as dist

    import paillier_halo2_amd as pz
    from paillier_halo2_amd import engine as E

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)  # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _stdout_to_stderr():
            dist.init_process_group(args.backend or "nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            dist.barrier()   # the communicator comes up here (and says so), not inside a timed region
    log = (lambda s: print("[bench] " + s, file=sys.stderr, flush=True)) if rank == 0 else (lambda s: None)

    eng = pz.Engine(local)
    eng.bind_torch_stream()  # a real stream, current for torch too: the event waits of the pipeline order against it
    # multiplier issue peak of this device, measured live (8 independent mads per lane and iteration)
    from paillier_halo2_amd import probe   # libpz_probe.so: measurement only, outside the product ABI

    mad_peak = max(8192 * 256 * 1024 * 8 / (probe.ubench_mad_indep(local, 8192, 1024) * 1e-3) / 1e12 for _ in range(3))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "msm22":
        from paillier_halo2_amd import dist as pzd

        if args.emulate_ranks > 1:
            em = pzd.emulate_ranks_msm(eng, torch, dist if use_dist else None, args.emulate_ranks, args.log_n, args.msm_split, log)
            em_cols = None
            if args.parallel == "columns":     # the commitment all-gather of column-parallel proving, 37 full-width columns of 2^14 rows
                kk, nc = 14, 37
                g_ = torch.Generator(device="cuda")
                g_.manual_seed(7)
                d_c = torch.randint(-(1 << 63), (1 << 63) - 1, (nc, 1 << kk, 4), dtype=torch.int64, device="cuda", generator=g_)
                d_c[:, :, 3] &= 0x0FFFFFFFFFFFFFFF
                d_bb = torch.zeros((1 << kk, 8), dtype=torch.int64, device="cuda")
                eng.g1_fixed_base_mul_dev(d_c[0].data_ptr(), 1 << kk, d_bb.data_ptr())
                tb_ = eng.load_bases_dev(d_bb.data_ptr(), 1 << kk)
                em_cols = pzd.emulate_ranks_columns(eng, torch, dist if use_dist else None, args.emulate_ranks, tb_, d_c, nc, 1 << kk, log)
                tb_.free()
            res = {"metric": "rank emulation of the sharded MSM: every rank's code path on one GPU", "value": 1.0 if em["all_equal"] and (em_cols is None or em_cols["all_equal"]) else 0.0,
                   "unit": "all ranks equal the single-call result", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": None, "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "u32 limbs (254-bit modular integers)", "data": "synthetic",
                   "config": {"workload": "c4 rank emulation: 2^%d-point MSM, world %d, %s split" % (args.log_n, args.emulate_ranks, args.msm_split)},
                   "emulate_ranks": em, "emulate_ranks_columns": em_cols}
        elif args.emulate_world > 1:
            res = pzd.emulate_sharded_msm(eng, torch, args.emulate_world, args.log_n, args.steps, args.warmup, log, scalars=args.msm_scalars,
                                          share_window_bits=int(os.environ.get("PZ_SHARE_WINDOW_BITS", "0")),
                                          window_split_bits=int(os.environ.get("PZ_WINDOW_SPLIT_BITS", "0")))
        else:
            res = pzd.bench_sharded_msm(eng, torch, dist if use_dist else None, rank, world, args.log_n, args.steps,
                                        args.warmup, barrier, log, split=args.msm_split, scalars=args.msm_scalars)
        if rank == 0:
            res["rccl_ranks"] = dist.get_world_size() if use_dist else 1
            res["backend"] = dist.get_backend() if use_dist else None
            res["config"]["env_switches"] = env_switches()
            print(json.dumps(res))
        if use_dist:
            dist.destroy_process_group()
        return

    R = argparse.Namespace(rank=rank, world=world, local=local, use_dist=use_dist, log=log, eng=eng, mad_peak=mad_peak, barrier=barrier)
    # (--parallel columns splits ONE proof's hot-path columns over the ranks: a mode of the hot-path workload)
    if args.workload in ("c2", "c2u", "c3", "c5") and not args.hot_path_headline and args.parallel != "columns":
        out = connected_line(args, R)
    else:
        out = hot_path_line(args, R)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
