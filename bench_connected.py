"""bench.py's HEADLINE workload (round 5: its `with_next_rows` leg): ONE connected proof per step at the benchmark's configuration -- the whole create_proof dataflow
of paillier_halo2_amd/prover.py on the proof's own data (reference: /root/reference/src/bench.rs:161-171, `gen_proof` after keygen):

    K3 trace -> K4 cells in halo2-lib's break-point column layout -> advice commitments -> permuted lookup columns -> grand products
    -> (challenge y) -> coefficient forms -> 64-column tiles extended and folded into the quotient as they are produced, against the
    RESIDENT extended forms of the proving key (selectors + sigma: 77 GB at c2 on the quotient's three cosets) -> h pieces -> evaluations -> SHPLONK,

every phase closed by a synchronising download of its commitments into a hashing transcript (the host round trip a real transcript
forces).  The circuit structure (selectors, copy constraints, break points) comes from paillier_halo2_amd/circuit_structure.py; the
proving key is built once per (key, message-shape) by prover.keygen and timed separately.

`verify()` runs in bench.py's cpu_baseline / checker leg (the one place that may touch oracle/): the quotient's degree bound, the
verifier's identity h(x)(x^n - 1) == the constraint expression of the EVALUATIONS (oracle/verifier.py), and the multi-point opening's
identity in the exponent against the proof's commitments (the bench's SRS is synthetic: its scalar is known to the checker).
"""
from __future__ import annotations

import os
import time

import numpy as np


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def cpp_connected(cw, proofs=4, verify_with=None, log=lambda s: None, via_stepper=False):
    """the same connected proof from the COMPILED prover (paillier_halo2_amd/host/prove_connected.cpp over include/pz.h only): the
    workload's structure and inputs written to a job file, the binary run as a child process (its own contexts: call after
    cw.release()), its last proof checked by verify_file_proof when verify_with (oracle.cref) is given.  -> dict for the bench line"""
    import tempfile

    from paillier_halo2_amd import consts, prover_job

    L = consts.limbs_to_int
    nn, g, m, r = (L(x) for x in cw.variants[0])
    msgs = [(L(v[2]), L(v[3])) for v in cw.variants]
    with tempfile.TemporaryDirectory(prefix="pz_job_") as td:
        job, proof = os.path.join(td, "job.bin"), os.path.join(td, "proof.bin")
        t0 = time.perf_counter()
        prover_job.write_job(job, cw.cs, cw.starts_host, cw.enc_bits, cw.kind, cw.ng, cw.nr, nn, g, msgs, cw.s_tox, seed=11, proofs=proofs)
        t1 = time.perf_counter()
        env = dict(os.environ, PZ_PROVE_VIA_STEPPER="1") if via_stepper else None
        line = prover_job.run(job, proof, timeout=600, env=env)
        t2 = time.perf_counter()
        out = {"value": 1e3 / line["mean_proof_ms"], "unit": "proofs/s", "ms_per_step": line["mean_proof_ms"], "ms_per_proof_best_of": line["best_proof_ms"],
               "proofs": proofs,
               "of_which_witness_ms": line["of_which_witness_ms"], "keygen_ms": line["keygen_ms"], "connected": True,
               "job_file_gb": os.path.getsize(job) / 1e9, "job_write_s": t1 - t0, "binary_wall_s": t2 - t1,
               "quotient_degree_ok": line["quotient_degree_ok"],
               "pipelined_witness": line.get("pipelined_witness"), "via": line.get("via", "host/create_proof.hpp composed by the driver"),
               "note": ("tests/cpp/prove_connected with PZ_PROVE_VIA_STEPPER=1: pz_pk_create (the structure's HOST arrays, selectors as bytes) + "
                        "pz_proof_begin ... pz_proof_open_finish, one call per transcript round, the next proof's K3 + K4 written by a second host "
                        "thread on a second context (rust/pz-rt prove_pipelined's recipe); keygen_ms = pz_pk_create incl. the 3.7 GB upload"
                        if via_stepper else
                        "tests/cpp/prove_connected: keygen + create_proof (paillier_halo2_amd/host/create_proof.hpp) from plain C++ over the C "
                        "ABI only -- no torch, no HIP call in the host, no oracle; the next proof's K3 + K4 on a second context under this proof's "
                        "advice commitments (pipelined_witness); the circuit structure and inputs arrive in a job file")
                       + "; value = mean over the proofs after the first (which grows the library's workspaces), each incl. its K3 + K4"}
        if verify_with is not None:
            rec = prover_job.read_proofs(proof)
            last = "p%d/" % (proofs - 1)
            ver = verify_file_proof(verify_with, rec, last, cw.cs, cw.k, cw.s_tox)
            ver.pop("ciphertext", None)
            out["verified"], out["verification"] = ver["verified"], ver
    return out


def cpp_fresh_message(enc_bits, k, lookup_bits, seed, steps=4, minimum_rows=20, verify_with=None, log=lambda s: None, streamed_key=None, arena_gb=None):
    """a NEW key pair and message per proof from the COMPILED prover alone (`prove_connected --fresh`): per step the circuit structure
    generated on the device by the library (pz_circuit_structure_dev), keygen on its device arrays, K3 + K4, create_proof -- the
    reference's per-message cost (paillier.rs:50-55 makes every message its own circuit; bench.rs:161-171) with no Python in the loop.
    The last proof is checked by verify_file_proof when verify_with (oracle.cref) is given.  -> dict for the bench line"""
    import random
    import tempfile
    from types import SimpleNamespace

    import bench
    from paillier_halo2_amd import consts, prover_job

    s_tox = random.Random(seed ^ 0x535253).randrange(2, consts.FR_R)
    inputs = [bench.synth_inputs(enc_bits, seed + 9001 + i) for i in range(steps)]
    with tempfile.TemporaryDirectory(prefix="pz_fresh_") as td:
        params, proof = os.path.join(td, "params.bin"), os.path.join(td, "proof.bin")
        prover_job.write_fresh_params(params, enc_bits, k, lookup_bits if lookup_bits is not None else k - 1, inputs, s_tox, minimum_rows=minimum_rows,
                                      seed=13)
        t0 = time.perf_counter()
        env = dict(os.environ)
        if streamed_key is not None:
            env["PZ_PROVE_STREAMED_KEY"] = str(streamed_key)
        if arena_gb is not None:             # (default: one arena over nearly all free device memory; 0 = driver allocations + block cache)
            env["PZ_PROVE_ARENA_GB"] = str(arena_gb)
        line = prover_job.run_fresh(params, proof, timeout=1100, env=env)
        wall = time.perf_counter() - t0
        out = {"value": 1e3 / line["mean_step_ms"], "unit": "proofs/s", "steps": steps, "s_per_step": line["mean_step_ms"] / 1e3, "of_which": line["of_which"],
               "connected": True, "binary_wall_s": wall, "quotient_degree_ok": line["quotient_degree_ok"], "per_step_log": line.get("stderr_tail"),
               "device_memory": line.get("arena"),
               "note": "tests/cpp/prove_connected --fresh: every step a new key pair and message -> circuit structure on the device (pz_circuit_structure_dev) "
                       "-> keygen on its device arrays (host/create_proof.hpp) -> K3 + K4 -> create_proof, from plain C++ over the C ABI only; device "
                       "memory from one arena (pz_dev_arena; arena gib 0 = driver allocations recycled through the block cache); mean over the steps after the first"}
        if verify_with is not None:
            rec = prover_job.read_proofs(proof)
            last = "p%d/" % (steps - 1)
            A, Lk, m_ = (int(x) for x in rec[last + "shape"][0][:3])
            st = SimpleNamespace(n_adv=A, n_lk=Lk, m=m_, blinding_factors=6)
            ver = verify_file_proof(verify_with, rec, last, st, k, s_tox)
            nn, g, m, r = inputs[-1]
            ver["ciphertext_is_g_m_r_n"] = bool(ver.pop("ciphertext") == pow(g, m, nn * nn) * pow(r, nn, nn * nn) % (nn * nn))
            out["verified"], out["verification"] = bool(ver["verified"] and ver["ciphertext_is_g_m_r_n"]), ver
    return out


class ConnectedWorkload:
    def __init__(self, eng, torch, enc_bits: int, k: int, seed: int, lookup_bits=None, log=lambda s: None, tile: int = 64,
                 circuit: str = "encrypt", pipeline=None, cosets=None, srs=None, trim: bool = True, minimum_rows: int = 20,
                 streamed_key=None, lookup_tile=None):
        import random

        import bench
        from paillier_halo2_amd import circuit_structure as CS
        from paillier_halo2_amd import consts, prover

        self.eng, self.torch, self.log = eng, torch, log
        self.enc_bits, self.k, self.n = enc_bits, k, 1 << k
        self.lb = lookup_bits if lookup_bits is not None else k - 1
        self.Ln = enc_bits // 64
        self.L = 2 * self.Ln
        nn, g, m, r = bench.synth_inputs(enc_bits, seed)
        self.ints = (nn, g, m, r)
        self.circuit = circuit
        self.uniform = circuit == "encrypt_uniform"
        self.add = circuit == "add"            # PaillierChip::add (paillier.rs:62-85): ONE mul_mod of two ciphertexts assigned at enc_bits (bench.rs:98-103)
        self.kind = 2 if self.uniform else 1 if self.add else 0
        self.ng = 0 if self.add else 2 * enc_bits if self.uniform else m.bit_length() + bin(m).count("1")
        self.nr = 0 if self.add else nn.bit_length() + bin(nn).count("1")
        self.n_steps = self.ng + self.nr + 1
        vr = random.Random(seed ^ 0x636F6E)
        lim = lambda x: consts.int_to_limbs(x, self.Ln)
        if self.add:
            # the add circuit's shape does not depend on its inputs: one key, the proofs of a run are of DISTINCT ciphertext pairs
            # (variant = n | unused | c1 | c2, each Ln words: c1, c2 uniform below 2^enc_bits as the reference's test assigns them)
            self.variants = [tuple(lim(x) for x in (nn, 0, vr.getrandbits(enc_bits), vr.getrandbits(enc_bits))) for _ in range(3)]
        elif self.uniform:
            # the uniform-shape circuit: the message's bits are WITNESS cells -- one key serves every message: the
            # proofs of a run are of DISTINCT messages
            self.variants = [tuple(lim(x) for x in (nn, g, mm, rr)) for mm, rr in ((m, r), (vr.randrange(0, nn), vr.randrange(1, nn)),
                                                                                    (vr.randrange(0, nn), vr.randrange(1, nn)))]
        else:
            # one key serves the proofs of ONE message under different randomness: the message's bits are circuit structure
            # (paillier.rs:50-55), r is not (r^n's exponent is the public n)
            self.variants = [tuple(lim(x) for x in (nn, g, m, rr)) for rr in (r, vr.randrange(1, nn), vr.randrange(1, nn))]
        # ---- circuit structure (what halo2's keygen extracts by synthesising the circuit once)
        t0 = time.perf_counter()
        sa = CS.stream_structure(circuit, enc_bits, 64, self.lb, m, nn, device="cuda")        # the template tiled on the device ...
        assert (sa.n_steps_g, sa.n_steps_r) == (self.ng, self.nr)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        # the tester's row budget (layout.RowBudget): calculate_params(Some(20)) on the reference's bench path (bench.rs:161-171)
        self.minimum_rows = minimum_rows
        self.cs, starts = CS.columns(sa, k, self.lb, minimum_rows=minimum_rows, device="cuda", keep_on_device=True)   # ... and the structure kept there for keygen
        torch.cuda.synchronize()
        self.n_cells, self.n_lookups = sa.n_cells, int(sa.lookup_src.shape[0])
        del sa
        self.structure_ms = {"stream_walk_and_tiling": (t1 - t0) * 1e3, "columns_cycles_selectors": (time.perf_counter() - t1) * 1e3}
        self.d_starts = torch.from_numpy(starts.astype(np.int64)).cuda()
        self.starts_host = np.asarray(starts, dtype=np.int64)
        self.A, self.Lk, self.m = self.cs.n_adv, self.cs.n_lk, self.cs.m
        # ---- SRS: monomial and Lagrange bases from a seeded scalar (ParamsKZG::setup, as gen_srs does)
        # (srs: (bases_lagrange, bases_monomial, scalar) of another workload of the same k -- the parameters do not depend on the message)
        M = consts.fr_mont_limbs
        self.own_srs = srs is None
        if srs is None:
            self.s_tox = random.Random(seed ^ 0x535253).randrange(2, consts.FR_R)
            d_g = torch.zeros((self.n, 8), dtype=torch.int64, device="cuda")
            d_gl = torch.zeros((self.n, 8), dtype=torch.int64, device="cuda")
            eng.srs_setup_g1_dev(k, M(self.s_tox), M(consts.fr_omega(k)), d_g.data_ptr(), d_gl.data_ptr())
            eng.sync()
            self.bl, self.bm = eng.load_bases_dev(d_gl.data_ptr(), self.n), eng.load_bases_dev(d_g.data_ptr(), self.n)
            del d_g, d_gl
        else:
            self.bl, self.bm, self.s_tox = srs
        # ---- keygen_vk + keygen_pk: all three forms of the fixed and permutation polynomials, resident
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # the quotient from THREE cosets (its degree is below 3n) instead of halo2's 4n-point coset: PZ_CONNECTED_COSETS=4 for the A/B
        self.cosets = int(os.environ.get("PZ_CONNECTED_COSETS", "3")) if cosets is None else int(cosets)
        # streamed_key: None = the extended proving key resident (c2: 77 GB); an integer R = only the first R permuted columns' extended forms
        # stay, the rest is re-extended per tile from the coefficient forms (R = 0 at BASELINE config c5: its extended key would be 239 GB)
        if streamed_key is None and os.environ.get("PZ_CONNECTED_STREAMED_KEY", "") != "":
            streamed_key = os.environ["PZ_CONNECTED_STREAMED_KEY"]
            streamed_key = streamed_key if streamed_key == "auto" else int(streamed_key)
        if streamed_key == "auto":
            streamed_key = self.memory_plan(tile, lookup_tile, 2 if (pipeline is None or pipeline) else 1)
        self.streamed_key = streamed_key
        self.pk = prover.keygen(eng, self.cs, self.bl, self.bm, cosets=self.cosets, ext_resident_cols=streamed_key)
        torch.cuda.synchronize()
        self.keygen_ms = (time.perf_counter() - t2) * 1e3
        if trim:                                   # hand the allocator's free blocks back (trim=False: a following key reuses them)
            torch.cuda.empty_cache()
        self.ws = prover.Workspace(self.pk, tile, lookup_tile=lookup_tile)
        # two witness slots: proof i+1's K3 + K4 run on a second context / stream under proof i's advice commitments
        self.pipeline = (os.environ.get("PZ_CONNECTED_PIPELINE", "1") == "1") if pipeline is None else bool(pipeline)
        self.slots = [prover._zeros_cap(self.m, self.n, 4) for _ in range(2 if self.pipeline else 1)]
        self.cols = self.slots[0]
        self.engw, self.stream_w = eng, None
        if self.pipeline:
            import paillier_halo2_amd as pz

            self.engw = pz.Engine(eng.device)
            self.stream_w = torch.cuda.Stream()
            self.engw.set_stream(self.stream_w.cuda_stream)
            self.ready_ev = [torch.cuda.Event() for _ in range(2)]
            self.free_ev = [torch.cuda.Event() for _ in range(2)]
        self.produced = 0
        self.d_steps = torch.zeros((self.n_steps, 4, self.L), dtype=torch.int64, device="cuda")
        self._n2_limbs = consts.int_to_limbs(nn * nn, self.L)
        self.d_mod = torch.from_numpy(self._n2_limbs.astype(np.int64)).cuda()
        self.timings = {}
        self.last = None
        self.done = 0
        ext_bytes = sum(t.numel() * 8 for t in self.pk.fixed_ext + self.pk.sigma_ext)
        pk_bytes = ext_bytes + sum(t.numel() * 8 for t in (self.pk.fixed_coeff, self.pk.sigma_coeff, self.pk.sigma_lagrange))
        self.memory_gb = {"plan": getattr(self, "memory_plan_gb", None), "proving_key_resident": pk_bytes / 1e9, "of_which_extended_forms": ext_bytes / 1e9,
                          "proving_key_streamed": bool(self.pk.streamed), "extended_forms_resident_columns": list(self.pk.ext_resident),
                          "quotient_domain_cosets": self.pk.dom.cosets,
                          "grand_products_extended": self.ws._z_ext_flat.numel() * 8 / 1e9, "witness_columns": sum(t.numel() for t in self.slots) * 8 / 1e9,
                          "torch_allocated_after_setup": torch.cuda.memory_allocated() / 1e9}

    def memory_plan(self, tile, lookup_tile, n_slots, library_reserve_gb: float = 64.0):
        """how many permuted columns' EXTENDED key forms can stay resident on this device (prover.ProvingKey ext_resident_cols): everything the
        proof must hold is counted first -- coefficient forms of the key and sigma's Lagrange values, the witness slot(s), the grand products
        and their extended form, the tiles -- then a reserve for the library's own workspaces (K1's sort / partial sums take up to 48 GiB, the
        transforms and the product scans a few GB); what is left of the device's free memory goes to the extended key, whole tiles of 64
        columns at a time.  -> None (all resident: config c2) or R (0 at config c5)"""
        torch = self.torch
        n, A, Lk, m = self.n, self.cs.n_adv, self.cs.n_lk, self.cs.m
        F, S, E = A + 2, -(-m // 2), 32
        parts = (2 * n, n)                       # the quotient's three cosets: a 2n-point and an n-point part
        cap = lambda c, q=64: -(-c // q) * q
        lt = min(tile, Lk, lookup_tile or tile)
        fixed_need = (cap(F) + 2 * cap(m)) * n * E                                            # fixed + sigma coefficient forms, sigma's Lagrange values
        fixed_need += n_slots * cap(m) * n * E                                                # witness
        fixed_need += cap(S) * n * E + cap(S) * max(parts) * E + 3 * cap(Lk, 8) * n * E       # Z, its extended form (one part at a time), A' / S' / Z_lookup
        fixed_need += sum(tile * p * E + 4 * lt * p * E + 3 * p * E for p in parts)          # advice tile, lookup tiles, accumulators
        fixed_need += 2 * tile * max(parts) * E                                               # (streamed) the two key tiles
        free_b, _total = torch.cuda.mem_get_info()
        free_b += torch.cuda.memory_reserved() - torch.cuda.memory_allocated()       # blocks torch's allocator holds but nothing uses are available too
        budget = free_b - fixed_need - int(library_reserve_gb * 1e9)
        per_col = 2 * sum(parts) * E                                                          # one selector + one sigma column on all three cosets
        R = int(budget // per_col) if budget > 0 else 0
        self.memory_plan_gb = {"device_free_before_keygen": free_b / 1e9, "needed_without_extended_key": fixed_need / 1e9,
                               "library_reserve": library_reserve_gb, "extended_key_if_all_resident": m * per_col / 1e9,
                               "resident_columns_planned": None if R >= m else max(0, R // tile * tile)}
        return None if R >= m else max(0, R // tile * tile)

    def produce(self, eng=None):
        """K3 + K4 of the next proof into the next witness slot (on the witness context when pipelining)"""
        torch = self.torch
        eng = eng or self.engw
        i = self.produced
        slot = i % len(self.slots)
        nn, g, m, r = self.variants[i % len(self.variants)]
        cols = self.slots[slot]
        ctx = torch.cuda.stream(self.stream_w) if self.stream_w is not None else _Null()
        with ctx:
            if self.stream_w is not None:
                self.stream_w.wait_event(self.free_ev[slot])        # the proof that used this slot has finished with it
            cols.zero_()
            if self.add:
                # K3: the single step (c1, c2, q, r) of c1 c2 mod n^2 (pz_mul_mod; host-pointer form: four 4096-bit integers)
                ext = lambda a: np.concatenate([a, np.zeros(self.L - a.shape[0], dtype=np.uint64)])
                a_, b_ = ext(m), ext(r)
                q, rem = eng.mul_mod(self.L, a_, b_, self._n2_limbs)
                self.d_steps.copy_(torch.from_numpy(np.stack([a_, b_, np.asarray(q, dtype=np.uint64), np.asarray(rem, dtype=np.uint64)]).astype(np.int64)).view(1, 4, self.L),
                                   non_blocking=False)
                c = [rem]
            elif self.uniform:
                c, _, _ = eng.paillier_encrypt_uniform_dev(self.Ln, self.enc_bits, nn, g, m, r, self.d_steps.data_ptr(), self.n_steps)
            else:
                c, _, _ = eng.paillier_encrypt_dev(self.Ln, nn, g, m, r, self.d_steps.data_ptr(), self.n_steps)      # K3 (returns the ciphertext)
            inputs = np.concatenate([nn, g, m, r, np.asarray(c[0], dtype=np.uint64)])
            eng.circuit_expand_cols_dev(self.kind, self.Ln, 64, self.lb, inputs, self.d_steps.data_ptr(), self.ng, self.nr, self.d_mod.data_ptr(),
                                        cols.data_ptr(), cols[self.A].data_ptr(), self.d_starts.data_ptr(), self.A, self.cs.max_rows,
                                        self.cs.max_rows, self.n)                                               # K4, break-point layout
            if self.stream_w is not None:
                self.ready_ev[slot].record(self.stream_w)
        self.produced += 1

    def step(self, timed=True, last=False):
        """one proof: its witness (already under way when pipelining), then create_proof with a fresh hashing transcript"""
        from paillier_halo2_amd import prover

        torch, eng = self.torch, self.eng
        t0 = time.perf_counter()
        if self.produced <= self.done:
            self.produce()
        slot = self.done % len(self.slots)
        cols = self.slots[slot]
        if self.stream_w is not None:
            torch.cuda.current_stream().wait_event(self.ready_ev[slot])
        tr = prover.HashTranscript(b"pz-bench-%d" % self.done)
        if timed:
            eng.sync()
            self.timings["witness"] = self.timings.get("witness", 0.0) + (time.perf_counter() - t0) * 1e3
        hooks = {}
        if self.pipeline and not last:
            hooks["after_advice_launch"] = self.produce
        if os.environ.get("PZ_CONNECTED_SPLIT_COMMIT"):      # experiment: commitments' column groups alternate between two contexts
            if getattr(self, "eng2", None) is None:
                import paillier_halo2_amd as pz

                self.eng2 = pz.Engine(eng.device)
            hooks["commit_engine"], hooks["commit_split"] = self.eng2, int(os.environ["PZ_CONNECTED_SPLIT_COMMIT"])
        pr = prover.create_proof(self.pk, cols, tr, seed=1000 + self.done, ws=self.ws, timings=self.timings if timed else None, hooks=hooks)
        if self.stream_w is not None:
            self.free_ev[slot].record(torch.cuda.current_stream())
        self.last = (pr, tr.challenges(), self.done % len(self.variants))
        self.done += 1
        return pr

    def run(self, steps, timed=True):
        for i in range(steps):
            self.step(timed, last=(i == steps - 1))
        self.torch.cuda.synchronize()

    def _lanes(self, n_lanes):
        """contexts for proofs in flight beside each other: lane 0 is the workload's own context / stream / workspace; every further lane
        gets its own library context (stream, library workspaces) and its own prover Workspace; lane i proves out of witness slot i"""
        from paillier_halo2_amd import prover

        import paillier_halo2_amd as pz

        torch = self.torch
        if getattr(self, "lanes", None) and len(self.lanes) >= n_lanes:
            return self.lanes[:n_lanes]
        assert n_lanes <= len(self.slots)
        lanes = [dict(eng=self.eng, stream=torch.cuda.current_stream(), ws=self.ws, cols=self.slots[0], d_steps=self.d_steps, own=False)]
        for i in range(1, n_lanes):
            e = pz.Engine(self.eng.device)
            st = torch.cuda.Stream()
            e.set_stream(st.cuda_stream)
            lanes.append(dict(eng=e, stream=st, ws=prover.Workspace(self.pk, self.ws.tile), cols=self.slots[i],
                              d_steps=torch.zeros_like(self.d_steps), own=True))
        self.lanes = lanes
        return lanes

    def run_in_flight(self, steps, n_lanes=2):
        """`steps` proofs with `n_lanes` in flight: proofs are independent, so while one is in its commitment-heavy phases (advice,
        lookups, grand products: K1, whose sort / fold / reduction kernels leave the multiplier idle a third of the time) the other is
        in its transform-heavy ones (quotient, evaluations, opening: K2 and the line kernels).  Two stage locks keep the lanes in that
        alternation and the proofs in order; each lane is a host thread with its own context, stream and workspace; the witness of a
        lane's next proof (K3: four workgroups) is produced before the lane asks for stage 1.  Per-proof LATENCY is not improved
        (it grows: the lanes share the GPU); throughput is.  -> seconds of wall time for the `steps` proofs"""
        import threading

        from paillier_halo2_amd import prover

        torch = self.torch
        lanes = self._lanes(n_lanes)
        torch.cuda.synchronize()
        stage1, stage2 = threading.Lock(), threading.Lock()
        turn = threading.Condition()
        state = {"next": 0, "err": None}
        results = [None] * steps
        lat = [0.0] * steps

        def lane_main(li):
            ln = lanes[li]
            eng = ln["eng"]
            try:
                with torch.cuda.stream(ln["stream"]):
                    for idx in range(li, steps, n_lanes):
                        nn, g, m, r = self.variants[idx % len(self.variants)]
                        cols = ln["cols"]
                        cols.zero_()
                        if self.uniform:
                            c, _, _ = eng.paillier_encrypt_uniform_dev(self.Ln, self.enc_bits, nn, g, m, r, ln["d_steps"].data_ptr(), self.n_steps)
                        else:
                            c, _, _ = eng.paillier_encrypt_dev(self.Ln, nn, g, m, r, ln["d_steps"].data_ptr(), self.n_steps)
                        inputs = np.concatenate([nn, g, m, r, np.asarray(c[0], dtype=np.uint64)])
                        eng.circuit_expand_cols_dev(self.kind, self.Ln, 64, self.lb, inputs, ln["d_steps"].data_ptr(), self.ng, self.nr,
                                                    self.d_mod.data_ptr(), cols.data_ptr(), cols[self.A].data_ptr(), self.d_starts.data_ptr(), self.A,
                                                    self.cs.max_rows, self.cs.max_rows, self.n)
                        with turn:                      # proofs enter stage 1 in order
                            turn.wait_for(lambda: state["next"] == idx or state["err"])
                        if state["err"]:
                            return
                        stage1.acquire()
                        t0 = time.perf_counter()
                        with turn:
                            state["next"] = idx + 1
                            turn.notify_all()
                        held = {"s1": True, "s2": False}

                        def on_phase(name):
                            if name == "products_commit":
                                stage2.acquire()
                                held["s2"] = True
                                stage1.release()
                                held["s1"] = False

                        try:
                            tr = prover.HashTranscript(b"pz-bench-%d" % (self.done + idx))
                            pr = prover.create_proof(self.pk, cols, tr, seed=1000 + self.done + idx, ws=ln["ws"], hooks={"on_phase": on_phase}, eng=eng)
                            eng.sync()
                        finally:
                            if held["s1"]:
                                stage1.release()
                            if held["s2"]:
                                stage2.release()
                        lat[idx] = time.perf_counter() - t0
                        results[idx] = (pr, tr.challenges(), idx % len(self.variants))
            except BaseException as ex:   # noqa: BLE001 -- reported by the caller
                with turn:
                    state["err"] = ex
                    turn.notify_all()

        t0 = time.perf_counter()
        th = [threading.Thread(target=lane_main, args=(i,)) for i in range(n_lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if state["err"] is not None:
            raise state["err"]
        self.done += steps
        self.produced = self.done
        self.last = results[-1]
        self.in_flight_latency_ms = [x * 1e3 for x in lat]
        return dt

    def phase_ms(self, steps):
        return {k_: v_ / max(1, steps) for k_, v_ in self.timings.items()}

    def counts(self):
        S = self.pk.n_sets
        return {"advice_cols": self.A, "advice_cols_filled": self.cs.n_adv_used, "lookup_cols": self.Lk, "minimum_rows": self.minimum_rows,
                "max_rows": self.cs.max_rows, "permutation_cols": self.m, "permutation_sets": S,
                "advice_cells": self.n_cells, "lookup_cells": self.n_lookups,
                "msm_witness": self.A + self.Lk, "msm_full": 3 * self.Lk + S + 1 + 3 + 2,
                "ntt_polys": self.m + 3 * self.Lk + S, "polys_opened": 2 * self.m + self.A + 2 + 3 * self.Lk + S + 2 - 1}

    def verify(self, cref):
        """the checker leg (oracle/): the LAST timed proof against the verifier's arithmetic"""
        from oracle import pyref as P
        from oracle import verifier as V
        from paillier_halo2_amd import prover

        pr, ch, variant = self.last
        R = P.FR_R
        t0 = time.perf_counter()

        def ints(a):
            a = np.asarray(a, dtype=np.uint64)
            flat = cref.fr_mont_to_ints(a.reshape(-1, 4))
            p = a.shape[1]
            return [flat[i * p:(i + 1) * p] for i in range(a.shape[0])]

        # the verifier's side of Fiat-Shamir: every challenge re-derived from the proof (must be the ones the prover's transcript drew)
        drawn = V.replay_challenges(ch.transcript_seed, pr.commitments, pr.evals)
        replayed = all(getattr(ch, nm) == v_ for nm, v_ in drawn.items()) and len(drawn) == 8
        ev = {k_: ints(v_) for k_, v_ in pr.evals.items()}
        want = V.expected_h(self.k, self.cs.blinding_factors, self.A, self.Lk, prover.CHUNK, ev, ch.beta, ch.gamma, ch.y, ch.x, prover.DELTA)
        ident = bool(want == ev["h"][0][0])
        xn = pow(ch.x, self.n, R)
        hc = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont([pow(xn, i, R) for i in range(3)]), pr.commitments["h"]))
        vk = self.pk.vk_commitments()
        com = dict(pr.commitments)
        com.update(fixed=vk["fixed"], sigma=vk["sigma"], h=[hc])
        lay = prover.query_layout(self.A, self.Lk, self.m, self.pk.n_sets)
        pts = prover.rotation_points(self.pk.dom, ch.x)
        opening = bool(V.shplonk_check(cref, lay, pts, com, ev, ch.sh_y, ch.sh_v, ch.sh_u, pr.commitments["w1"][0], pr.commitments["w2"][0], self.s_tox))
        return {"verified": bool(pr.h_degree_ok and ident and opening and replayed), "quotient_degree_le_3n_minus_4": bool(pr.h_degree_ok),
                "challenges_rederived_from_the_proof": bool(replayed), "h_x_times_xn_minus_1_equals_expression_of_evaluations": ident, "shplonk_identity_on_the_proofs_commitments": opening,
                "commitments": int(sum(v.shape[0] for v in pr.commitments.values())), "evaluations": int(sum(v.shape[0] * v.shape[1] for v in pr.evals.values())),
                "checker_s": time.perf_counter() - t0}

    def release(self, trim: bool = True):
        if self.own_srs:
            for b in (self.bl, self.bm):
                b.free()
        self.torch.cuda.synchronize()
        if self.engw is not self.eng:
            self.engw.close()
        if getattr(self, "eng2", None) is not None:
            self.eng2.close()
            self.eng2 = None
        for ln in getattr(self, "lanes", None) or []:
            if ln["own"]:
                ln["eng"].close()
        self.lanes = None
        for name in ("pk", "ws", "cols", "slots", "d_steps", "last"):
            setattr(self, name, None)
        if trim:
            self.torch.cuda.empty_cache()


def verify_file_proof(cref, rec, prefix, st, k, s_tox):
    """the checker leg for a proof written by the compiled prover (paillier_halo2_amd/host/prove_connected.cpp; `rec` =
    prover_job.read_proofs): the identity h(x)(x^n - 1) == the constraint expression of the evaluations, and SHPLONK's identity in the
    exponent over the proof's commitments and the key's, with the challenges the binary's transcript drew.  st: prover.CircuitStructure"""
    from oracle import pyref as P
    from oracle import verifier as V
    from paillier_halo2_amd import prover

    R = P.FR_R
    A, Lk, m, n = st.n_adv, st.n_lk, st.m, 1 << k
    S = -(-m // prover.CHUNK)
    L = lambda x: sum(int(v) << (64 * i) for i, v in enumerate(x))

    def ints(a):
        flat = cref.fr_mont_to_ints(np.ascontiguousarray(a).reshape(-1, 4))
        p = a.shape[1] // 4
        return [flat[i * p:(i + 1) * p] for i in range(a.shape[0])]

    ch = {nm: L(rec[prefix + "ch/" + nm][0]) for nm in ("theta", "beta", "gamma", "y", "x", "sh_y", "sh_v", "sh_u")}
    # the verifier's side of Fiat-Shamir: the binary seeds a proof's transcript with its index (8 bytes); every challenge it recorded
    # must be the one re-derived here from the proof's own commitments and evaluations
    seed = int(prefix[1:-1]).to_bytes(8, "little")
    drawn = V.replay_challenges(seed, {k_[len(prefix) + 2:]: v_ for k_, v_ in rec.items() if k_.startswith(prefix + "c/")},
                                {k_[len(prefix) + 2:]: v_ for k_, v_ in rec.items() if k_.startswith(prefix + "e/")})
    replayed = drawn == ch
    ev = {k_[len(prefix) + 2:]: ints(v_) for k_, v_ in rec.items() if k_.startswith(prefix + "e/")}
    ev["constants"] = ev["lookup_advice"][Lk:]
    ev["lookup_advice"] = ev["lookup_advice"][:Lk]
    want = V.expected_h(k, st.blinding_factors, A, Lk, prover.CHUNK, ev, ch["beta"], ch["gamma"], ch["y"], ch["x"], prover.DELTA)
    ident = bool(want == ev["h"][0][0])
    com = {k_[len(prefix) + 2:]: v_ for k_, v_ in rec.items() if k_.startswith(prefix + "c/")}
    shapes = com["advice"].shape == (A, 8) and com["perm_z"].shape == (S, 8) and com["h"].shape == (3, 8)
    xn = pow(ch["x"], n, R)
    hc = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont([pow(xn, i, R) for i in range(3)]), com["h"]))
    vkp = prefix if (prefix + "vk/fixed") in rec else ""          # (--fresh: every proof has its own key)
    com.update(fixed=rec[vkp + "vk/fixed"], sigma=rec[vkp + "vk/sigma"], h=[hc])
    lay = prover.query_layout(A, Lk, m, S)
    pts = prover.rotation_points(prover.Domain(k, st.blinding_factors), ch["x"])
    opening = bool(V.shplonk_check(cref, lay, pts, com, ev, ch["sh_y"], ch["sh_v"], ch["sh_u"], com["w1"][0], com["w2"][0], s_tox))
    degree = bool(int(rec[prefix + "flags"][0][0]) == 1)
    return {"verified": bool(degree and ident and opening and shapes and replayed), "quotient_degree_le_3n_minus_4": degree,
            "challenges_rederived_from_the_proof": bool(replayed), "h_x_times_xn_minus_1_equals_expression_of_evaluations": ident, "shplonk_identity_on_the_proofs_commitments": opening,
            "ciphertext": L(rec[prefix + "ciphertext"][0])}
