"""The library's own circuit-structure generator (csrc/pz_structure.hip: pz_circuit_structure_dev, the compiled counterpart of
paillier_halo2_amd/circuit_structure.py) -- what halo2's keygen extracts by synthesising the reference's drivers
(/root/reference/src/bench.rs:33-117; PaillierChip::{encrypt, add}, src/paillier.rs:32-85): selectors, the copy-constraint permutation,
constants, break points.  Held array for array against the Python generator (itself held against the oracle's independent walk WITH
values: tests/test_circuit_structure.py) on the reference's encrypt shape, a larger lookup width, a 3-limb key, the add circuit on
88-bit limbs (paillier.rs:186-187), 48-bit limbs, the uniform-shape circuit, and both row budgets of the tester; then used: structure ->
pz_pk_create_dev -> a connected proof that verifies, all from the library."""
import random

import numpy as np
import pytest

from oracle import pyref as P
from tests.util import challenges_replay

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import paillier_halo2_amd as pz

    e = pz.Engine(0)
    e.bind_torch_stream()
    yield e
    e.close()


SHAPES = ((128, 64, 13, 14, 0x5042, "encrypt", 20), (128, 64, 15, 16, 0x77, "encrypt", 20), (264, 88, 12, 13, 0x99, "add", 20),
          (192, 64, 11, 14, 0x31, "encrypt", 9), (96, 48, 9, 13, 0x62, "encrypt", 20), (176, 88, 12, 16, 0x63, "encrypt_uniform", 20),
          (128, 64, 10, 11, 0x5042, "encrypt", 20),      # 249 configured advice columns, 248 filled (calculate_params(Some(20)))
          (128, 64, 13, 14, 0x51, "encrypt_uniform", 9))


@pytest.mark.parametrize("bits,W,lb,k,seed,kind,mr", SHAPES)
def test_native_structure_equals_the_python_generator(eng, bits, W, lb, k, seed, kind, mr):
    from paillier_halo2_amd import circuit_structure as CS
    from paillier_halo2_amd import prover_native

    n, g, m, r = P.synth_paillier_inputs(bits, seed, standard_g=False)
    sa = CS.stream_structure(kind, bits, W, lb, m, n)
    cs, starts = CS.columns(sa, k, lb, minimum_rows=mr, device="cpu")
    ns = prover_native.NativeStructure(eng, kind, bits, W, lb, k, exp_g=m, exp_r=n, minimum_rows=mr)
    try:
        assert (ns.n_cells, ns.n_lookups, ns.n_steps_g, ns.n_steps_r) == (sa.n_cells, sa.lookup_src.shape[0], sa.n_steps_g, sa.n_steps_r)
        assert (ns.n_adv, ns.n_adv_used, ns.n_lk, ns.max_rows) == (cs.n_adv, cs.n_adv_used, cs.n_lk, cs.max_rows)
        assert ns.constants() == [int(c) for c in cs.constants]          # same constants in the same rows of the constants column
        assert ns.starts().tolist() == starts.tolist()
        sel, mc, mr_ = ns.download()
        assert np.array_equal(sel, cs.selectors)
        assert np.array_equal(mc, cs.map_col) and np.array_equal(mr_, cs.map_row)
    finally:
        ns.free()


def test_native_structure_against_the_oracle_walk(eng):
    """the library's generator held directly against the ORACLE's restatement (oracle/circuit.py: a walk of the whole circuit WITH values, no code
    shared with either product generator): break points, selectors, lookup column count, and sigma wherever the image is not in the constants column
    (whose row order is each generator's own); and the native sigma is satisfied by the oracle's witness (a cell only maps to a cell of equal value)"""
    from oracle import circuit as CQ
    from paillier_halo2_amd import prover_native

    for bits, W, lb, k, seed, kind in ((128, 64, 13, 14, 0x5042, "encrypt"), (264, 88, 12, 13, 0x99, "add")):
        n, g, m, r = P.synth_paillier_inputs(bits, seed, standard_g=False)
        res = P.paillier_add_native(n, m, r) if kind == "add" else P.paillier_enc_native(n, g, m, r)
        st = CQ.build(kind, n, g, m, r, res, bits, W, lb, k)
        assert CQ.mock_prover(st) == []
        ns = prover_native.NativeStructure(eng, kind, bits, W, lb, k, exp_g=m, exp_r=n)
        try:
            assert (ns.n_adv, ns.n_lk, ns.max_rows) == (st.n_adv, st.n_lk, st.max_rows) and ns.starts().tolist() == st.starts
            sel, mc, mr = ns.download()
            assert np.array_equal(sel, st.selectors)
            Wd = st.n_adv + st.n_lk
            keep = st.map_col[:Wd] < Wd
            assert np.array_equal(mc[:Wd] < Wd, keep)
            assert np.array_equal(mc[:Wd][keep], st.map_col[:Wd][keep]) and np.array_equal(mr[:Wd][keep], st.map_row[:Wd][keep])
            st.constants = ns.constants()
            cols = CQ.perm_columns(st)
            flat = mc.astype(np.int64) * st.n + mr
            assert np.unique(flat).size == flat.size                      # a permutation
            moved = np.argwhere(flat != np.arange(flat.size).reshape(flat.shape))
            for c, rr in moved[:: max(1, moved.shape[0] // 4000)].tolist():
                assert cols[c][rr] == cols[int(mc[c, rr])][int(mr[c, rr])], (kind, c, rr)
        finally:
            ns.free()


def test_structure_to_key_to_proof_from_the_library_alone(eng, cref):
    """pz_circuit_structure_dev -> pz_pk_create_dev -> pz_structure_free -> K3 -> K4 with the structure's break points -> the stepper's
    proof, checked as the verifier would (oracle/verifier.py); the key equals the Python prover's key on the Python structure"""
    import torch

    from oracle import verifier as V
    from paillier_halo2_amd import circuit_structure as CS
    from paillier_halo2_amd import prover, prover_native

    bits, W, lb, k = 128, 64, 13, 14
    n = 1 << k
    Ln = bits // W
    nn, g, m, r = P.synth_paillier_inputs(bits, 0x5042, standard_g=False)
    res = P.paillier_enc_native(nn, g, m, r)
    rng = random.Random(0x57)
    R = P.FR_R
    F = lambda v: cref.fr_ints_to_mont([v % R])[0]
    s_tox = rng.randrange(2, R)
    d_g = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    d_gl = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.srs_setup_g1_dev(k, F(s_tox), F(P.fr_omega(k)), d_g.data_ptr(), d_gl.data_ptr())
    eng.sync()
    bl, bm = eng.load_bases_dev(d_gl.data_ptr(), n), eng.load_bases_dev(d_g.data_ptr(), n)
    ns = prover_native.NativeStructure(eng, "encrypt", bits, W, lb, k, exp_g=m, exp_r=nn)
    key = ns.key(bl, bm, tile=8)
    # the witness in the structure's own break-point layout (its device table of column starts)
    arr = lambda v, l: cref.int_to_limbs(v, l)
    cap = ns.n_steps_g + ns.n_steps_r + 1
    d_steps = torch.zeros((cap, 4, 2 * Ln), dtype=torch.int64, device="cuda")
    c, g_, r_ = eng.paillier_encrypt_dev(Ln, arr(nn, Ln), arr(g, Ln), arr(m, Ln), arr(r, Ln), d_steps.data_ptr(), cap)
    assert cref.limbs_to_int(c[0]) == res and (int(g_[0]), int(r_[0])) == (ns.n_steps_g, ns.n_steps_r)
    d_mod = torch.from_numpy(arr(nn * nn, 2 * Ln).astype(np.int64)).cuda()
    cols = torch.zeros((ns.m, n, 4), dtype=torch.int64, device="cuda")
    inputs = np.concatenate([arr(nn, Ln), arr(g, Ln), arr(m, Ln), arr(r, Ln), arr(res, 2 * Ln)])
    eng.circuit_expand_cols_dev(0, Ln, W, lb, inputs, d_steps.data_ptr(), ns.n_steps_g, ns.n_steps_r, d_mod.data_ptr(), cols.data_ptr(),
                                cols[ns.n_adv].data_ptr(), ns.d_starts, ns.n_adv, ns.max_rows, ns.max_rows, n)
    eng.sync()
    n_adv, n_lk, m_ = ns.n_adv, ns.n_lk, ns.m
    ns.free()                                            # the key holds its own forms: the structure is done with
    try:
        sa = CS.stream_structure("encrypt", bits, W, lb, m, nn)
        cs, _ = CS.columns(sa, k, lb, device="cpu")
        pk = prover.keygen(eng, cs, bl, bm)
        vk, vk_py = key.vk_commitments(), pk.vk_commitments()
        assert np.array_equal(vk["fixed"], vk_py["fixed"]) and np.array_equal(vk["sigma"], vk_py["sigma"])
        tr = prover.HashTranscript(b"native-structure")
        pr = prover_native.create_proof(key, cols.data_ptr(), tr, seed=5)
        ch = tr.challenges()
        assert challenges_replay(pr, ch)

        def ints(a):
            a = np.asarray(a, dtype=np.uint64)
            flat = cref.fr_mont_to_ints(a.reshape(-1, 4))
            p_ = a.shape[1]
            return [flat[i * p_:(i + 1) * p_] for i in range(a.shape[0])]

        ev = {k_: ints(v_) for k_, v_ in pr.evals.items()}
        assert pr.h_degree_ok
        assert V.expected_h(k, 6, n_adv, n_lk, prover.CHUNK, ev, ch.beta, ch.gamma, ch.y, ch.x, prover.DELTA) == ev["h"][0][0]
        xn = pow(ch.x, n, R)
        hc = cref.g1_normalize(cref.msm_g1(cref.fr_ints_to_mont([pow(xn, i, R) for i in range(3)]), pr.commitments["h"]))
        com = dict(pr.commitments)
        com.update(fixed=vk["fixed"], sigma=vk["sigma"], h=[hc])
        assert V.shplonk_check(cref, prover.query_layout(n_adv, n_lk, m_, key.n_sets), prover.rotation_points(pk.dom, ch.x), com, ev, ch.sh_y,
                               ch.sh_v, ch.sh_u, pr.commitments["w1"][0], pr.commitments["w2"][0], s_tox)
    finally:
        key.free()
        bl.free()
        bm.free()


def _library_proof(e, cref, s_tox, tile=8):
    """structure -> key -> witness -> the stepper's proof on context `e`, every device array allocated by the LIBRARY (pz_dev_alloc) -- what a
    compiled caller does.  Fixed challenges and seed: the result is a function of the inputs alone.  -> (commitments, evals, vk)"""
    from paillier_halo2_amd import prover, prover_native

    bits, W, lb, k = 128, 64, 13, 14
    n = 1 << k
    Ln = bits // W
    R = P.FR_R
    nn, g, m, r = P.synth_paillier_inputs(bits, 0x5043, standard_g=False)
    res = P.paillier_enc_native(nn, g, m, r)
    F = lambda v: cref.fr_ints_to_mont([v % R])[0]
    arr = lambda v, l: cref.int_to_limbs(v, l)
    d_g, d_gl = e.dev_alloc(n * 64), e.dev_alloc(n * 64)
    e.srs_setup_g1_dev(k, F(s_tox), F(P.fr_omega(k)), d_g, d_gl)
    e.sync()
    bl, bm = e.load_bases_dev(d_gl, n), e.load_bases_dev(d_g, n)
    e.dev_free(d_g)
    e.dev_free(d_gl)
    ns = prover_native.NativeStructure(e, "encrypt", bits, W, lb, k, exp_g=m, exp_r=nn)
    key = ns.key(bl, bm, tile=tile)
    cap = ns.n_steps_g + ns.n_steps_r + 1
    d_steps, d_mod, cols = e.dev_alloc(cap * 4 * 2 * Ln * 8), e.dev_alloc(2 * Ln * 8), e.dev_alloc(ns.m * n * 32)
    try:
        c, g_, r_ = e.paillier_encrypt_dev(Ln, arr(nn, Ln), arr(g, Ln), arr(m, Ln), arr(r, Ln), d_steps, cap)
        assert cref.limbs_to_int(c[0]) == res
        e.upload(d_mod, arr(nn * nn, 2 * Ln))
        e.dev_memset(cols, 0, ns.m * n * 32)
        inputs = np.concatenate([arr(nn, Ln), arr(g, Ln), arr(m, Ln), arr(r, Ln), arr(res, 2 * Ln)])
        e.circuit_expand_cols_dev(0, Ln, W, lb, inputs, d_steps, ns.n_steps_g, ns.n_steps_r, d_mod, cols, cols + ns.n_adv * n * 32, ns.d_starts, ns.n_adv,
                                  ns.max_rows, ns.max_rows, n)
        e.sync()
        ns.free()
        rng = random.Random(9)
        ch = prover.Challenges(*(rng.randrange(2, R) for _ in range(8)))
        pr = prover_native.create_proof(key, cols, ch, seed=5)
        assert pr.h_degree_ok
        return pr.commitments, pr.evals, key.vk_commitments()
    finally:
        for d in (d_steps, d_mod, cols):
            e.dev_free(d)
        key.free()
        bl.free()
        bm.free()


def test_library_path_from_a_poisoned_arena_is_byte_identical(eng, cref):
    """pz_dev_arena: the whole library path (structure generator, keygen, K3, K4, the stepper's six phases, every workspace and table)
    with its memory carved out of ONE block that is filled with 0xA5 at creation and refilled on every free gives the same key and the same
    proof, byte for byte, as the same path on driver allocations -- nothing depends on fresh memory being zero; a second proof from the
    same arena leaves no more memory in use than the first (no leak), and nothing had to be passed to the driver"""
    import os

    import paillier_halo2_amd as pz

    s_tox = random.Random(0x58).randrange(2, P.FR_R)
    com0, ev0, vk0 = _library_proof(eng, cref, s_tox)
    e = pz.Engine(0)
    os.environ["PZ_DEV_ARENA_POISON"] = "1"
    try:
        e.dev_arena(6 << 30)
        com1, ev1, vk1 = _library_proof(e, cref, s_tox)
        used1 = e.dev_arena_info()
        com2, ev2, vk2 = _library_proof(e, cref, s_tox)
        used2 = e.dev_arena_info()
        for got_c, got_e, got_vk in ((com1, ev1, vk1), (com2, ev2, vk2)):
            assert set(got_c) == set(com0) and set(got_e) == set(ev0)
            for nm in com0:
                assert np.array_equal(got_c[nm], com0[nm]), nm
            for nm in ev0:
                assert np.array_equal(got_e[nm], ev0[nm]), nm
            assert np.array_equal(got_vk["fixed"], vk0["fixed"]) and np.array_equal(got_vk["sigma"], vk0["sigma"])
        assert used1["served"] > 50 and used1["missed"] == 0 and used2["missed"] == 0
        assert used2["used"] == used1["used"]            # what stays in use is the context's grown workspaces and tables, not a key's arrays
    finally:
        os.environ.pop("PZ_DEV_ARENA_POISON", None)
        e.close()
