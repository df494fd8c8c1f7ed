"""ONE connected proof of the Paillier encrypt circuit at the reference's own bench shape (128-bit n, 64-bit limbs, k = 14,
lookup_bits = 13: /root/reference/src/bench.rs:139-140,161-171), every phase on the device in create_proof's order
(paillier_halo2_amd/prover.py), checked the way halo2's verifier would (oracle/verifier.py):

  (i)   the quotient has degree <= 3n - 4 (the constraint numerator vanishes on the domain);
  (ii)  h(x) (x^n - 1) equals the gate / permutation / lookup expression recomputed on the host from the EVALUATIONS only;
  (iii) SHPLONK's final identity holds in the exponent against the proof's actual commitments (the test knows the toxic scalar);
  (iv)  a tampered witness cell, a broken copy constraint, an out-of-table lookup value and a wrong claimed result each break (i) / (ii).

The witness is K3 -> K4's own output in halo2-lib's break-point column layout; the circuit structure (selectors, copy constraints,
constants, break points) is the keygen INPUT, built by oracle/circuit.py from the restated dependency patterns."""
import random

import numpy as np
import pytest

from oracle import circuit as CQ
from oracle import pyref as P
from oracle import verifier as V
from tests.util import challenges_replay

pytestmark = pytest.mark.gpu

K, LB, BITS, W = 14, 13, 128, 64
R = P.FR_R


@pytest.fixture(scope="module")
def eng():
    import paillier_halo2_amd as pz

    e = pz.Engine(0)
    e.bind_torch_stream()
    yield e
    e.close()


@pytest.fixture(scope="module")
def world(eng, cref):
    """inputs, structure, SRS (known toxic scalar), proving key, and a function that re-creates the K3 -> K4 witness columns"""
    import torch
    from paillier_halo2_amd import layout, prover

    nn, g, m, r = P.synth_paillier_inputs(BITS, 0x5042, standard_g=False)
    res = P.paillier_enc_native(nn, g, m, r)
    st = CQ.build("encrypt", nn, g, m, r, res, BITS, W, LB, K)
    assert CQ.mock_prover(st) == []
    n = 1 << K
    Ln = BITS // W
    ng, nr = m.bit_length() + bin(m).count("1"), nn.bit_length() + bin(nn).count("1")
    mask, total = P.gate_mask_circuit("encrypt", BITS, W, LB, ng, nr)
    starts = layout.break_points(mask, st.max_rows)
    assert starts.tolist() == st.starts[: st.info["n_adv_used"] + 1]     # the library's break rule == the oracle's restatement
    d_starts = torch.from_numpy(np.asarray(st.starts, dtype=np.int64)).cuda()     # n_adv + 1 entries (configured columns past the filled ones are empty)
    arr = lambda v, l: cref.int_to_limbs(v, l)

    def witness(res_claimed=res):
        steps_cap = ng + nr + 1
        d_steps = torch.zeros((steps_cap, 4, 2 * Ln), dtype=torch.int64, device="cuda")
        c, g_, r_ = eng.paillier_encrypt_dev(Ln, arr(nn, Ln), arr(g, Ln), arr(m, Ln), arr(r, Ln), d_steps.data_ptr(), steps_cap)
        assert cref.limbs_to_int(c[0]) == res and (int(g_[0]), int(r_[0])) == (ng, nr)
        d_mod = torch.from_numpy(arr(nn * nn, 2 * Ln).astype(np.int64)).cuda()
        cols = torch.zeros((st.m, n, 4), dtype=torch.int64, device="cuda")
        inputs = np.concatenate([arr(nn, Ln), arr(g, Ln), arr(m, Ln), arr(r, Ln), arr(res_claimed, 2 * Ln)])
        eng.circuit_expand_cols_dev(0, Ln, W, LB, inputs, d_steps.data_ptr(), ng, nr, d_mod.data_ptr(), cols.data_ptr(),
                                    cols[st.n_adv].data_ptr(), d_starts.data_ptr(), st.n_adv, st.max_rows, st.max_rows, n)
        eng.sync()
        return cols

    rng = random.Random(0x5eed)
    s_tox = rng.randrange(2, R)
    F = lambda v: cref.fr_ints_to_mont([v % R])[0]
    d_g = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    d_gl = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.srs_setup_g1_dev(K, F(s_tox), F(P.fr_omega(K)), d_g.data_ptr(), d_gl.data_ptr())
    eng.sync()
    bl, bm = eng.load_bases_dev(d_gl.data_ptr(), n), eng.load_bases_dev(d_g.data_ptr(), n)
    # the keygen input comes from the PRODUCT's structure generator (what bench.py uses at config c2); it must describe the same
    # circuit as the oracle's restatement: same selectors, same break points, the same sigma up to the order of the constants column
    from paillier_halo2_amd import circuit_structure as CS

    sa = CS.stream_structure("encrypt", BITS, W, LB, m, nn)
    cs, cs_starts = CS.columns(sa, K, LB, blinding_factors=st.blinding_factors)
    assert cs_starts.tolist() == st.starts and np.array_equal(cs.selectors, st.selectors) and cs.n_lk == st.n_lk
    assert sorted(cs.constants) == sorted(st.constants)
    W_ = st.n_adv + st.n_lk
    keep = st.map_col[:W_] < W_
    assert np.array_equal(cs.map_col[:W_][keep], st.map_col[:W_][keep]) and np.array_equal(cs.map_row[:W_][keep], st.map_row[:W_][keep])
    pk = prover.keygen(eng, cs, bl, bm)
    pk4 = prover.keygen(eng, cs, bl, bm, cosets=4)       # the same key over halo2's own 4n-point quotient domain
    ch = prover.Challenges(*(rng.randrange(2, R) for _ in range(8)))
    yield dict(st=st, pk=pk, pk4=pk4, ch=ch, witness=witness, s_tox=s_tox, inputs=(nn, g, m, r, res))
    bl.free()
    bm.free()


def _ints(cref, a):
    """(count, points, 4) Montgomery words -> [[int]]"""
    a = np.asarray(a, dtype=np.uint64)
    flat = cref.fr_mont_to_ints(a.reshape(-1, 4))
    p = a.shape[1]
    return [flat[i * p:(i + 1) * p] for i in range(a.shape[0])]


def _verify(cref, world, pr):
    """-> (degree bound holds, identity at x holds, opening identity holds)"""
    from paillier_halo2_amd import prover

    st, pk, ch = world["st"], world["pk"], world["ch"]
    ev = {k: _ints(cref, v) for k, v in pr.evals.items()}
    want = V.expected_h(K, st.blinding_factors, st.n_adv, st.n_lk, prover.CHUNK, ev, ch.beta, ch.gamma, ch.y, ch.x, prover.DELTA)
    ident = want == ev["h"][0][0]
    # the verifier's h commitment: sum_i x^(n i) [h_i]
    xn = pow(ch.x, 1 << K, R)
    hc = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont([pow(xn, i, R) for i in range(3)]), pr.commitments["h"]))
    vk = pk.vk_commitments()
    com = dict(pr.commitments)
    com.update(fixed=vk["fixed"], sigma=vk["sigma"], h=[hc])
    lay = prover.query_layout(st.n_adv, st.n_lk, st.m, pk.n_sets)
    pts = prover.rotation_points(pk.dom, ch.x)
    opening = V.shplonk_check(cref, lay, pts, com, ev, ch.sh_y, ch.sh_v, ch.sh_u, pr.commitments["w1"][0], pr.commitments["w2"][0], world["s_tox"])
    return pr.h_degree_ok, ident, opening


def test_break_point_columns_equal_the_oracle_layout(eng, cref, world):
    """K3 -> K4 in the break-point layout: every advice and lookup-advice column cell for cell against oracle/circuit.py"""
    st = world["st"]
    cols = world["witness"]().cpu().numpy().view(np.uint64)
    for j in range(st.n_adv):
        assert cref.fr_mont_to_ints(cols[j]) == st.adv_cols[j], j
    for j in range(st.n_lk):
        assert cref.fr_mont_to_ints(cols[st.n_adv + j]) == st.lk_cols[j], j
    assert not cols[st.m - 1].any()


def test_connected_proof_satisfies_the_verifier(eng, cref, world):
    from paillier_halo2_amd import prover

    pr = prover.create_proof(world["pk"], world["witness"](), world["ch"], seed=1, tile=8)     # 34 columns in five tiles
    assert pr.h_top is not None and pr.h_top.shape[0] == 3
    deg, ident, opening = _verify(cref, world, pr)
    assert deg, "quotient degree > 3n - 4"
    assert ident, "h(x) (x^n - 1) != the constraint expression of the evaluations"
    assert opening, "SHPLONK identity"
    assert pr.commitments["h"].any() and pr.commitments["perm_z"].shape[0] == world["pk"].n_sets
    # the tiling does not change the proof: one 64-column tile gives the same h and the same openings
    pr2 = prover.create_proof(world["pk"], world["witness"](), world["ch"], seed=1, tile=64)
    for kname in pr.commitments:
        assert np.array_equal(pr.commitments[kname], pr2.commitments[kname]), kname
    for kname in pr.evals:
        assert np.array_equal(pr.evals[kname], pr2.evals[kname]), kname
    # the quotient from THREE cosets of <w_n> (the default: deg h < 3n) is the quotient halo2 computes on its 4n-point coset: the same
    # pieces, the same commitments, the same openings -- byte for byte
    pr4 = prover.create_proof(world["pk4"], world["witness"](), world["ch"], seed=1, tile=8)
    assert pr4.h_top.shape[0] == (1 << K) + 3 and pr4.h_degree_ok
    for kname in pr.commitments:
        assert np.array_equal(pr.commitments[kname], pr4.commitments[kname]), kname
    for kname in pr.evals:
        assert np.array_equal(pr.evals[kname], pr4.evals[kname]), kname


def test_connected_proof_with_a_hashing_transcript(eng, cref, world):
    """the same flow with every challenge DERIVED from the commitments of the phase before it (prover.HashTranscript: halo2's
    Blake2b transcript in its primitives, a synchronising download per phase) -- what bench.py's with_next_rows times; verified with the challenges it drew"""
    from paillier_halo2_amd import prover

    tr = prover.HashTranscript(b"test")
    tm = {}
    pr = prover.create_proof(world["pk"], world["witness"](), tr, seed=7, tile=16, timings=tm)
    w2 = dict(world)
    w2["ch"] = tr.challenges()
    assert w2["ch"] != world["ch"] and set(tm) >= {"advice_commit", "quotient", "multiopen"}
    assert _verify(cref, w2, pr) == (True, True, True)
    # ... and they are the challenges a verifier re-derives from the proof alone; a proof with one commitment changed replays to others
    assert challenges_replay(pr, w2["ch"])
    bad = prover.Proof(commitments={k_: v_.copy() for k_, v_ in pr.commitments.items()}, evals=pr.evals)
    bad.commitments["perm_z"][0, 0] ^= np.uint64(1)
    assert not challenges_replay(bad, w2["ch"])
    # a different witness (another randomness r is not available here, so: another blinding seed) changes every challenge
    tr2 = prover.HashTranscript(b"test")
    prover.create_proof(world["pk"], world["witness"](), tr2, seed=8, tile=16)
    assert tr2.challenges().beta != tr.challenges().beta


def _mont1(cref, v):
    import torch

    return torch.from_numpy(cref.fr_ints_to_mont([v % R])[0].astype(np.int64)).cuda()


def test_tampered_witness_cell_breaks_the_identity(eng, cref, world):
    from paillier_halo2_amd import prover

    st = world["st"]
    j = 3
    r0 = int(np.nonzero(st.selectors[j])[0][100])

    def tamper(cols):
        cols[j, r0 + 3] = _mont1(cref, st.adv_cols[j][r0 + 3] + 1)      # the output cell of an enabled gate

    pr = prover.create_proof(world["pk"], world["witness"](), world["ch"], seed=2, tile=8, hooks={"advice": tamper})
    deg, ident, opening = _verify(cref, world, pr)
    assert not deg and not ident
    assert opening          # the openings themselves are still honest openings of what was committed


def test_broken_copy_constraint_breaks_the_identity(eng, cref, world):
    """a lookup-advice cell changed to ANOTHER value of the table: no gate touches it and the lookup still holds -- only the copy
    constraint to the digit it copies is broken, and the permutation product no longer closes"""
    from paillier_halo2_amd import prover

    st = world["st"]
    col, row = st.n_adv + 1, 777
    old = st.lk_cols[1][row]

    def tamper(cols):
        cols[col, row] = _mont1(cref, (old + 1) % (1 << LB))

    pr = prover.create_proof(world["pk"], world["witness"](), world["ch"], seed=3, tile=8, hooks={"advice": tamper})
    deg, ident, _ = _verify(cref, world, pr)
    assert not deg and not ident


def test_out_of_table_lookup_value(eng, cref, world):
    """(a) an out-of-range digit in a lookup-advice column: the prover stops where the reference's does (permute_expression_pair finds
    no table row: PZ_ERR_RANGE); (b) past that point -- a permuted column that is not a permutation of its input -- the lookup
    product does not telescope and the identity fails"""
    import paillier_halo2_amd as pz
    from paillier_halo2_amd import _lib, prover

    st = world["st"]

    def digit(cols):
        cols[st.n_adv, 5] = _mont1(cref, 1 << LB)

    with pytest.raises(pz.PzError) as e:
        prover.create_proof(world["pk"], world["witness"](), world["ch"], seed=4, tile=8, hooks={"advice": digit})
    assert e.value.status == _lib.PZ_ERR_RANGE

    def permuted(Ap, Sp):
        Ap[0, 9] = _mont1(cref, (1 << LB) + 5)
        Sp[0, 9] = _mont1(cref, (1 << LB) + 5)

    pr = prover.create_proof(world["pk"], world["witness"](), world["ch"], seed=4, tile=8, hooks={"permuted": permuted})
    deg, ident, _ = _verify(cref, world, pr)
    assert not deg and not ident


def test_wrong_claimed_result_breaks_the_identity(eng, cref, world):
    """bench.rs:68-74: the driver assigns the claimed ciphertext and asserts equality in-circuit; a wrong `res` expands (K4) to a
    satisfied-gates witness whose assert_equal_fresh bit is 0 -- the copy constraint tying that bit to the constant 1 fails"""
    from paillier_halo2_amd import prover

    res = world["inputs"][4]
    pr = prover.create_proof(world["pk"], world["witness"](res ^ 2), world["ch"], seed=5, tile=8)
    deg, ident, _ = _verify(cref, world, pr)
    assert not deg and not ident


def test_c3_add_circuit_connected_proof_at_size(eng, cref):
    """BASELINE config c3 as ONE connected proof: c1 * c2 mod n^2 at a 2048-bit key (operands assigned at enc_bits, bench.rs:98-103), k = 15,
    lookup_bits 14 -- pz_mul_mod -> K4 (kind 1, break-point columns) -> keygen from the product's own structure generator ->
    create_proof with a hashing transcript -> the verifier's three checks.  (64 limbs: the mul_mod block template at the size the c2
    circuit tiles 6171 times.)"""
    import random

    import torch
    from paillier_halo2_amd import circuit_structure as CS
    from paillier_halo2_amd import prover

    enc_bits, k, lb = 2048, 15, 14
    Ln, L, n = enc_bits // 64, 2 * (enc_bits // 64), 1 << 15
    rng = random.Random(0x5044)
    nn = P.synth_paillier_inputs(enc_bits, 0x5044)[0]
    c1, c2 = rng.getrandbits(enc_bits), rng.getrandbits(enc_bits)
    res = P.paillier_add_native(nn, c1, c2)
    lim = lambda v, l: cref.int_to_limbs(v, l)
    q, rem = eng.mul_mod(L, lim(c1, L), lim(c2, L), lim(nn * nn, L))
    assert cref.limbs_to_int(rem) == res
    sa = CS.stream_structure("add", enc_bits, 64, lb)
    cs, starts = CS.columns(sa, k, lb)
    assert sa.n_cells == eng.circuit_cells(1, Ln, 64, lb)[0] and cs.n_adv >= 2
    d_starts = torch.from_numpy(starts.astype(np.int64)).cuda()
    d_steps = torch.from_numpy(np.stack([lim(c1, L), lim(c2, L), q, rem]).astype(np.int64)).cuda().view(1, 4, L)
    d_mod = torch.from_numpy(lim(nn * nn, L).astype(np.int64)).cuda()
    cols = torch.zeros((cs.m, n, 4), dtype=torch.int64, device="cuda")
    inputs = np.concatenate([lim(nn, Ln), lim(0, Ln), lim(c1, Ln), lim(c2, Ln), lim(res, L)])     # n | g (unused by add) | c1 | c2 | res
    eng.circuit_expand_cols_dev(1, Ln, 64, lb, inputs, d_steps.data_ptr(), 0, 0, d_mod.data_ptr(), cols.data_ptr(), cols[cs.n_adv].data_ptr(),
                                d_starts.data_ptr(), cs.n_adv, cs.max_rows, cs.max_rows, n)
    s_tox = rng.randrange(2, R)
    F = lambda v: cref.fr_ints_to_mont([v % R])[0]
    d_g = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    d_gl = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.srs_setup_g1_dev(k, F(s_tox), F(P.fr_omega(k)), d_g.data_ptr(), d_gl.data_ptr())
    eng.sync()
    bl, bm = eng.load_bases_dev(d_gl.data_ptr(), n), eng.load_bases_dev(d_g.data_ptr(), n)
    pk = prover.keygen(eng, cs, bl, bm)
    tr = prover.HashTranscript(b"c3")
    pr = prover.create_proof(pk, cols, tr, seed=11)
    ch = tr.challenges()
    assert challenges_replay(pr, ch)
    ev = {k_: _ints(cref, v_) for k_, v_ in pr.evals.items()}
    want = V.expected_h(k, cs.blinding_factors, cs.n_adv, cs.n_lk, prover.CHUNK, ev, ch.beta, ch.gamma, ch.y, ch.x, prover.DELTA)
    assert pr.h_degree_ok and want == ev["h"][0][0]
    xn = pow(ch.x, n, R)
    hc = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont([pow(xn, i, R) for i in range(3)]), pr.commitments["h"]))
    vk = pk.vk_commitments()
    com = dict(pr.commitments)
    com.update(fixed=vk["fixed"], sigma=vk["sigma"], h=[hc])
    assert V.shplonk_check(cref, prover.query_layout(cs.n_adv, cs.n_lk, cs.m, pk.n_sets), prover.rotation_points(pk.dom, ch.x), com, ev, ch.sh_y,
                           ch.sh_v, ch.sh_u, pr.commitments["w1"][0], pr.commitments["w2"][0], s_tox)
    bl.free()
    bm.free()


def _connected_workload(eng, bits, k, seed=0x5043):
    import torch

    import bench_connected

    return bench_connected.ConnectedWorkload(eng, torch, bits, k, seed)


def test_connected_proof_many_tiles_and_sets(eng, cref):
    """the bench's connected workload (bench_connected.ConnectedWorkload: the product's structure generator, K3 -> K4 break-point
    columns, keygen, create_proof with a hashing transcript, the next witness produced on a second context) at a 1024-bit key, k = 16:
    967 + 54 + 1 permuted columns = 16 tiles of 64, 511 grand products, 23 GB of proving key -- three proofs, the last one checked as
    the verifier would.  (A batching bug in keygen -- sigma built 256 columns at a time against a 256-entry table of delta powers --
    passed every smaller test and failed exactly here.)"""
    wl = _connected_workload(eng, 1024, 16)
    try:
        assert wl.A > 900 and wl.pk.n_sets > 500
        wl.run(3, timed=False)
        v = wl.verify(cref)
        assert v["verified"] is True, v
        assert v["commitments"] == wl.A + 4 * wl.Lk + wl.pk.n_sets + 1 + 3 + 2
    finally:
        wl.release()


def test_streamed_proving_key_proves_byte_for_byte_the_same(eng, cref):
    """the STREAMED proving key (prover.ProvingKey(ext_resident_cols = R): only the coefficient forms of the fixed / sigma columns stay in
    HBM beyond the first R columns, create_proof re-extends them per 64-column tile beside the advice tile -- the memory plan BASELINE
    config c5 needs) at a 1024-bit key, k = 16 (16 tiles, 511 grand products): verifying key, every commitment
    and every evaluation of the proof are BYTE FOR BYTE those of the resident key, fully streamed (R = 0), with a resident prefix on a
    tile boundary (R = 128) and off it (R = 100: the straddling tile is re-extended whole); and the streamed proof verifies."""
    import torch

    import bench_connected

    srs, ref, ref_vk, key_gb = None, None, None, {}
    for R_ in (None, 0, 128, 100):
        wl = bench_connected.ConnectedWorkload(eng, torch, 1024, 16, 0x5043, pipeline=False, srs=srs, streamed_key=R_)
        try:
            if srs is None:
                srs = (wl.bl, wl.bm, wl.s_tox)
                wl.own_srs = False                   # the bases outlive this workload: freed at the end of the test
            assert wl.pk.streamed == (R_ is not None) and wl.memory_gb["proving_key_streamed"] == (R_ is not None)
            key_gb[R_] = wl.memory_gb["of_which_extended_forms"]
            pr = wl.step(timed=False)
            torch.cuda.synchronize()
            vk = wl.pk.vk_commitments()
            if ref is None:
                ref, ref_vk = pr, vk
                assert pr.h_degree_ok
            else:
                assert all(np.array_equal(vk[f], ref_vk[f]) for f in ("fixed", "sigma"))
                assert sorted(pr.commitments) == sorted(ref.commitments) and sorted(pr.evals) == sorted(ref.evals)
                for f in ref.commitments:
                    assert np.array_equal(pr.commitments[f], ref.commitments[f]), (R_, "commitment", f)
                for f in ref.evals:
                    assert np.array_equal(pr.evals[f], ref.evals[f]), (R_, "evaluation", f)
                assert pr.h_degree_ok
                if R_ == 0:
                    v = wl.verify(cref)
                    assert v["verified"] is True, v
        finally:
            wl.release()
    assert key_gb[0] == 0.0 and 0 < key_gb[100] < key_gb[128] < key_gb[None]
    srs[0].free()
    srs[1].free()


def test_memory_plan_of_the_extended_key(eng):
    """ConnectedWorkload(streamed_key="auto"): what fits of the extended key stays resident -- all of it at a small shape, none when the
    reserve exceeds the device (what config c5 comes to: bench.py --workload c5), whole tiles in between"""
    import torch

    import bench_connected

    wl = bench_connected.ConnectedWorkload(eng, torch, 128, 14, 0x5042, streamed_key="auto", pipeline=False)
    try:
        assert wl.streamed_key is None and wl.pk.streamed is False and wl.memory_gb["plan"]["resident_columns_planned"] is None
        assert wl.memory_plan(8, None, 1, library_reserve_gb=1e6) == 0
        free_gb = wl.memory_plan_gb["device_free_before_keygen"]
        need = wl.memory_plan_gb["needed_without_extended_key"]
        per_col_gb = 2 * 3 * wl.n * 32 / 1e9
        R = wl.memory_plan(8, None, 1, library_reserve_gb=free_gb - need - 10.5 * per_col_gb)     # room for ten and a half columns
        assert R == 8, (R, wl.memory_plan_gb)
    finally:
        wl.release()


def test_c2_structure_and_witness_agree_at_size(eng, cref):
    """config c2 itself (2048-bit n, k = 17: 3.97 x 10^8 advice cells in 3033 break-point columns, 1.1 x 10^7 lookup cells in 84): the
    circuit structure the product generates and the witness K3 -> K4 writes describe the SAME circuit -- every one of sigma's
    4 x 10^8 cells maps to a cell holding the same value (all copy constraints, constants and lookup copies included), and every
    enabled gate holds inside its column (the custom-gate kernel over the 2^k domain with rotation 1 returns zero everywhere)"""
    import torch
    from paillier_halo2_amd import circuit_structure as CS
    from paillier_halo2_amd import consts

    import bench

    bits, k, lb = 2048, 17, 16
    n, Ln = 1 << k, bits // 64
    nn, g, m, r = bench.synth_inputs(bits, 0x5043)
    sa = CS.stream_structure("encrypt", bits, 64, lb, m, nn)
    cs, starts = CS.columns(sa, k, lb)
    A, Lk, M_ = cs.n_adv, cs.n_lk, cs.m
    assert (A, cs.n_adv_used, Lk) == (3034, 3033, 84) and sa.n_cells == eng.circuit_cells(0, Ln, 64, lb, sa.n_steps_g, sa.n_steps_r)[0]
    lim = lambda x, l: consts.int_to_limbs(x, l)
    n_steps = sa.n_steps_g + sa.n_steps_r + 1
    d_steps = torch.zeros((n_steps, 4, 2 * Ln), dtype=torch.int64, device="cuda")
    c, _, _ = eng.paillier_encrypt_dev(Ln, lim(nn, Ln), lim(g, Ln), lim(m, Ln), lim(r, Ln), d_steps.data_ptr(), n_steps)
    d_mod = torch.from_numpy(lim(nn * nn, 2 * Ln).astype(np.int64)).cuda()
    d_starts = torch.from_numpy(starts.astype(np.int64)).cuda()
    cols = torch.zeros((M_, n, 4), dtype=torch.int64, device="cuda")
    inputs = np.concatenate([lim(nn, Ln), lim(g, Ln), lim(m, Ln), lim(r, Ln), np.asarray(c[0], dtype=np.uint64)])
    eng.circuit_expand_cols_dev(0, Ln, 64, lb, inputs, d_steps.data_ptr(), sa.n_steps_g, sa.n_steps_r, d_mod.data_ptr(), cols.data_ptr(),
                                cols[A].data_ptr(), d_starts.data_ptr(), A, cs.max_rows, cs.max_rows, n)
    consts_col = np.zeros((n, 4), dtype=np.uint64)
    consts_col[: len(cs.constants)] = cref.fr_ints_to_mont(list(cs.constants))
    cols[A + Lk] = torch.from_numpy(consts_col.view(np.int64)).cuda()
    eng.sync()
    flat = cols.view(M_ * n, 4)
    bad = 0
    for c0 in range(0, M_, 256):
        c1 = min(M_, c0 + 256)
        img = torch.from_numpy(cs.map_col[c0:c1].astype(np.int64) * n + cs.map_row[c0:c1].astype(np.int64)).cuda().view(-1)
        bad += int((flat[c0 * n:c1 * n] != flat[img]).any(dim=1).sum().item())
    assert bad == 0, "%d cells differ from the cell sigma maps them to" % bad
    one = torch.from_numpy(consts.fr_mont_limbs(1).view(np.int64)).cuda()
    h = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    y = consts.fr_mont_limbs(0x1234567)
    for c0 in range(0, A, 64):
        c1 = min(A, c0 + 64)
        sel = torch.from_numpy(np.ascontiguousarray(cs.selectors[c0:c1])).cuda()
        selm = torch.where(sel.bool().unsqueeze(-1), one, torch.zeros_like(one)).contiguous()
        eng.quotient_gate_dev(cols[c0].data_ptr(), 4 * n, selm.data_ptr(), 4 * n, c1 - c0, k, 1, y, h.data_ptr())
    eng.sync()
    assert not h.any().item(), "an enabled gate does not hold inside its column"
    assert int(cs.selectors.sum()) > 10 ** 8


def test_uniform_shape_circuit_connected_proof(eng, cref):
    """the uniform-shape circuit (row f4: one key for every message) as a connected proof: K3's uniform schedule -> K4 kind 2 in
    break-point columns (cell for cell against the oracle's columns) -> keygen from the product's structure -> THREE DISTINCT
    messages proved under the one key, the last checked as the verifier would"""
    import torch

    import bench_connected

    wl = bench_connected.ConnectedWorkload(eng, torch, 128, 14, 0x51, circuit="encrypt_uniform")
    try:
        nn, g, m, r = wl.ints
        res = P.paillier_enc_native(nn, g, m, r)
        st = CQ.build("encrypt_uniform", nn, g, m, r, res, 128, 64, 13, 14)
        assert (wl.A, wl.Lk) == (st.n_adv, st.n_lk)
        wl.produce()                       # the first proof's witness (on the witness context); run() below proves it first
        torch.cuda.synchronize()
        cols = wl.slots[0].cpu().numpy().view(np.uint64)
        for j in range(st.n_adv):
            assert cref.fr_mont_to_ints(cols[j]) == st.adv_cols[j], j
        for j in range(st.n_lk):
            assert cref.fr_mont_to_ints(cols[st.n_adv + j]) == st.lk_cols[j], j
        msgs = {tuple(v[2].tolist()) for v in wl.variants}
        assert len(msgs) == 3
        wl.run(3, timed=False)
        assert wl.verify(cref)["verified"] is True
    finally:
        wl.release()


def test_library_stepper_is_the_same_prover(eng, cref, world):
    """include/pz.h "patch point D as entry points" (pz_pk_* / pz_proof_*: one call per transcript round, the composition of
    host/create_proof.hpp inside the library) through ctypes: the key it builds has the Python key's commitments, its proof satisfies the
    verifier's three checks (fixed challenges and a hashing transcript, library and caller-supplied blinding), a tampered cell fails
    the degree check and the identity, and the misuse paths return PZ_ERR_INVALID"""
    import ctypes as C

    import paillier_halo2_amd as pz
    from paillier_halo2_amd import _lib, consts, prover, prover_native

    pk = world["pk"]
    key = prover_native.NativeKey(eng, pk.st, pk.bases_lagrange, pk.bases_monomial, tile=8)
    try:
        vk, vk_py = key.vk_commitments(), pk.vk_commitments()
        assert np.array_equal(vk["fixed"], vk_py["fixed"]) and np.array_equal(vk["sigma"], vk_py["sigma"])
        assert key.n_sets == pk.n_sets and key.n_fixed == pk.st.n_adv + 2
        cols = world["witness"]()
        pr = prover_native.create_proof(key, cols.data_ptr(), world["ch"], seed=3)
        assert _verify(cref, world, pr) == (True, True, True)
        # caller-supplied randomness, a hashing transcript: every challenge derived from what the phases handed back
        rng = np.random.default_rng(5)
        words = rng.integers(0, 1 << 63, size=key.blinding_words, dtype=np.uint64)
        tr = prover.HashTranscript(b"native")
        cols = world["witness"]()
        pr2 = prover_native.create_proof(key, cols.data_ptr(), tr, blinding=words)
        w2 = dict(world)
        w2["ch"] = tr.challenges()
        assert _verify(cref, w2, pr2) == (True, True, True) and challenges_replay(pr2, w2["ch"])
        assert not np.array_equal(pr.commitments["advice"], pr2.commitments["advice"])          # other blinding rows
        # too few random words
        cols = world["witness"]()
        with pytest.raises(pz.PzError):
            prover_native.create_proof(key, cols.data_ptr(), world["ch"], blinding=words[:-1])
        # a tampered gate output: degree check and identity fail, the openings stay honest openings
        cols = world["witness"]()
        cols[0, 3, 0] ^= 1
        bad = prover_native.create_proof(key, cols.data_ptr(), world["ch"], seed=3)
        deg, ident, opening = _verify(cref, world, bad)
        assert (deg, ident, opening) == (False, False, True)
        # misuse: a phase out of order, a challenge that is not below r, freeing the key under an open proof
        L_, Mf = eng.L, consts.fr_mont_limbs
        VP = C.c_void_p
        ptr = lambda a: VP(a.ctypes.data)
        cols = world["witness"]()
        h = VP()
        adv = np.zeros((pk.st.n_adv + pk.st.n_lk, 8), dtype=np.uint64)
        SEEDED = C.c_size_t((1 << 64) - 1)             # pz.h PZ_BLINDING_SEEDED_TEST_STREAM: the deterministic stream is asked for by name
        assert L_.pz_proof_begin(key.handle, VP(cols.data_ptr()), 1, None, 0, C.byref(h), ptr(adv)) == _lib.PZ_ERR_INVALID   # NULL blinding without it
        assert L_.pz_proof_begin(key.handle, VP(cols.data_ptr()), 1, None, SEEDED, C.byref(h), ptr(adv)) == 0
        h2 = VP()
        assert L_.pz_proof_begin(key.handle, VP(cols.data_ptr()), 1, None, SEEDED, C.byref(h2), ptr(adv)) != 0      # one proof per key at a time
        assert L_.pz_pk_free(key.handle) != 0
        out3 = np.zeros((3, 8), dtype=np.uint64)
        five = Mf(5)
        assert L_.pz_proof_quotient(h, ptr(five), ptr(out3)) != 0                                              # lookups and products come first
        a_, b_ = np.zeros((pk.st.n_lk, 8), dtype=np.uint64), np.zeros((pk.st.n_lk, 8), dtype=np.uint64)
        not_canonical = np.full(4, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
        assert L_.pz_proof_lookups(h, ptr(not_canonical), ptr(a_), ptr(b_)) != 0
        # sticky failure (pz.h: after any error only pz_proof_free is valid): the out-of-order phase above has voided the proof, so even
        # the phase that WAS next is refused now instead of running on half-built state
        assert L_.pz_proof_lookups(h, ptr(five), ptr(a_), ptr(b_)) == _lib.PZ_ERR_INVALID
        assert L_.pz_proof_free(h) == 0
        # two threads race pz_proof_begin on ONE key (pz.h: entry points may be called from any thread): exactly one claims it
        import threading

        for _round in range(4):
            cols_t = [world["witness"](), world["witness"]()]
            hs, rcs = [VP(), VP()], [None, None]
            advs = [np.zeros_like(adv), np.zeros_like(adv)]
            gate = threading.Barrier(2)

            def racer(i):
                gate.wait()
                rcs[i] = L_.pz_proof_begin(key.handle, VP(cols_t[i].data_ptr()), 1, None, SEEDED, C.byref(hs[i]), ptr(advs[i]))

            th = [threading.Thread(target=racer, args=(i,)) for i in range(2)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert sorted(rcs) == sorted([0, _lib.PZ_ERR_INVALID]), rcs
            assert L_.pz_proof_free(hs[rcs.index(0)]) == 0
        # ... and the key serves the next proof
        pr3 = prover_native.create_proof(key, world["witness"]().data_ptr(), world["ch"], seed=9)
        assert _verify(cref, world, pr3) == (True, True, True)
    finally:
        key.free()


def test_stepper_keys_from_device_arrays_and_streamed(eng, cref, world):
    """pz_pk_create_dev (the structure's arrays already on the device: no host copy, no PCIe crossing) builds the SAME key as pz_pk_create
    (equal verifying-key commitments); and the library's streamed proving key (ext_resident_cols = 0 / 10: extended forms re-derived per
    tile inside pz_proof_quotient) gives the resident key's proof byte for byte -- same seed, same challenges"""
    import dataclasses

    import torch

    from paillier_halo2_amd import prover_native

    pk = world["pk"]
    st_dev = dataclasses.replace(pk.st, selectors=torch.from_numpy(np.ascontiguousarray(pk.st.selectors)).cuda(),
                                 map_col=torch.from_numpy(np.ascontiguousarray(pk.st.map_col).view(np.int32)).cuda(),
                                 map_row=torch.from_numpy(np.ascontiguousarray(pk.st.map_row).view(np.int32)).cuda())
    vk_py = pk.vk_commitments()
    ref = None
    for st_, R_ in ((pk.st, None), (st_dev, None), (st_dev, 0), (pk.st, 10)):
        key = prover_native.NativeKey(eng, st_, pk.bases_lagrange, pk.bases_monomial, tile=8, ext_resident_cols=R_)
        try:
            vk = key.vk_commitments()
            assert np.array_equal(vk["fixed"], vk_py["fixed"]) and np.array_equal(vk["sigma"], vk_py["sigma"])
            pr = prover_native.create_proof(key, world["witness"]().data_ptr(), world["ch"], seed=21)
            if ref is None:
                ref = pr
                assert _verify(cref, world, pr) == (True, True, True)
            else:
                for f in ref.commitments:
                    assert np.array_equal(pr.commitments[f], ref.commitments[f]), (R_, f)
                for f in ref.evals:
                    assert np.array_equal(pr.evals[f], ref.evals[f]), (R_, f)
                assert pr.h_degree_ok
        finally:
            key.free()


def test_stepper_shape_with_many_lookup_columns(eng, cref):
    """the stepper's commitment staging buffer was sized (m + S + 8) points while the lookups phase writes 2 n_lk --
    more once n_lk > 3 n_adv + 19.  pz_pk_create accepts any n_adv, n_lk >= 1, so such a shape must prove without touching memory it
    does not own: one gate-free advice column, 30 lookup columns of in-table values (k = 6), identity permutation."""
    import torch

    from paillier_halo2_amd import prover, prover_native

    k, lb, A, Lk = 6, 3, 1, 30
    n, m = 1 << k, A + Lk + 1
    rng = random.Random(0x6c6b)
    F = lambda v: cref.fr_ints_to_mont([v % R])[0]
    d_g = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    d_gl = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.srs_setup_g1_dev(k, F(rng.randrange(2, R)), F(P.fr_omega(k)), d_g.data_ptr(), d_gl.data_ptr())
    eng.sync()
    bl, bm = eng.load_bases_dev(d_gl.data_ptr(), n), eng.load_bases_dev(d_g.data_ptr(), n)
    ident_c = np.repeat(np.arange(m, dtype=np.uint32), n).reshape(m, n)
    ident_r = np.tile(np.arange(n, dtype=np.uint32), m).reshape(m, n)
    cs = prover.CircuitStructure(k=k, lookup_bits=lb, max_rows=n - 9, blinding_factors=6, selectors=np.zeros((A, n), dtype=np.uint8), n_lk=Lk,
                                 constants=[0, 1], map_col=ident_c, map_row=ident_r)
    key = prover_native.NativeKey(eng, cs, bl, bm, tile=2)
    try:
        assert 2 * Lk > (m - 1) + key.n_sets + 8 - 1            # the shape the old sizing could not hold
        vals = np.zeros((m, n, 4), dtype=np.uint64)
        lk = [[rng.randrange(1 << lb) for _ in range(n - 9)] + [0] * 9 for _ in range(Lk)]
        for j in range(Lk):
            vals[A + j] = cref.fr_ints_to_mont(lk[j])
        for seed in (1, 2):
            cols = torch.from_numpy(vals.view(np.int64)).cuda()
            ch = prover.Challenges(*(rng.randrange(2, R) for _ in range(8)))
            pr = prover_native.create_proof(key, cols.data_ptr(), ch, seed=seed)
            eng.sync()                                            # PZ_ERR_ASYNC would surface here
            assert pr.h_degree_ok and pr.commitments["perm_inputs"].shape == (Lk, 8) and pr.commitments["perm_tables"].shape == (Lk, 8)
            assert not np.array_equal(pr.commitments["perm_inputs"], pr.commitments["perm_tables"])
    finally:
        key.free()
        bl.free()
        bm.free()
