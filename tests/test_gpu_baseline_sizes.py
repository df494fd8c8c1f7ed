"""GPU parity at the sizes BASELINE.json's configs name (round-1 verdict: c3 / c4 / c5 were only ever run below size).

  c2  2048-bit encrypt, k = 17, lookup_bits 16 (the headline config): the WHOLE circuit's 3.97e8 advice and 1.1e7 lookup
      cells written by K3 -> K4 as 3033 + 84 columns; sampled columns (first, two inside the chains, last = ragged) cell
      for cell and commitment for commitment vs the oracle chain; every other column through linearity:
      commit(sum_j v^j col_j) == sum_j v^j commit(col_j) over all columns at once
  c3  2048-bit homomorphic add, k = 15: K3 (one mul_mod) -> K4 -> K1 over the whole circuit vs the oracle chain,
      MSM 2^15 vs the C restatement of best_multiexp, NTT 2^15 / 2^16 (see also test_gpu_kernels' parametrisations)
  c4  one 2^22-point MSM: full, as 8 disjoint window ranges and as 8 disjoint point ranges (the two multi-GPU
      splits, each rank's share run in turn on this one GPU, folded on the device in rank order) -- all equal, and
      equal to the closed form of the walk bases' discrete logs; uniform and witness-like scalars
  c5  3072-bit encrypt, k = 19: MSM 2^19 (closed form), and a sample of the circuit's columns through
      K3 -> K4 -> K1 vs the oracle chain (Python trace -> Python cells -> C best_multiexp)
Everything goes through the C ABI; the oracle is only the checker.
"""
import numpy as np
import pytest

from oracle import pyref as P
from tests.util import column_rows, canon_rand_scalars, ints_to_u64x4, walk_dlog_sum, witness_like_canon

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import paillier_halo2_amd as pz

    e = pz.Engine(0)
    e.bind_torch_stream()
    yield e
    e.close()


def _walk_bases_dev(eng, torch, n, s, t):
    """P_i = [s + i t] G on the device (fixed-base multiplication), s + n t < r so no reduction is involved"""
    assert s + n * t < P.FR_R
    ks = ints_to_u64x4([s + i * t for i in range(n)])
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    eng.fr_convert_dev(d_k.data_ptr(), n, True)
    d_b = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.g1_fixed_base_mul_dev(d_k.data_ptr(), n, d_b.data_ptr())
    eng.sync()
    return d_b


def _aff(cref, eng, jac):
    return cref.affine_mont_to_ints(eng.g1_normalize(np.asarray(jac, dtype=np.uint64).reshape(-1, 12)))[0]


@pytest.mark.parametrize("mix", ["uniform", "witness"])
def test_c4_msm_2pow22_full_and_both_8way_splits(eng, cref, mix):
    import torch

    from paillier_halo2_amd import dist as pzd

    log_n, world = 22, 8
    n = 1 << log_n
    s, t = (0x2F3A << 230) + 0x1234567890ABCDEF, 0xFEDCBA9876543211
    d_b = _walk_bases_dev(eng, torch, n, s, t)
    sc = canon_rand_scalars(n, 2201) if mix == "uniform" else witness_like_canon(n, 2202)
    want = P.g1_mul(P.G1_GEN, walk_dlog_sum(sc, s, t))
    d_s = torch.from_numpy(sc.view(np.int64)).cuda()
    eng.fr_convert_dev(d_s.data_ptr(), n, True)   # the ABI takes Montgomery form
    d_parts = torch.zeros((world, 12), dtype=torch.int64, device="cuda")
    d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    # (1) the whole MSM
    tb = eng.load_bases_dev(d_b.data_ptr(), n)
    assert (tb.window_bits, tb.n_windows) == (16, 16)
    eng.msm_dev(tb, d_s.data_ptr(), 1, n, 4 * n, d_out.data_ptr())
    eng.sync()
    full = d_out.cpu().numpy().astype(np.uint64)[0]
    assert _aff(cref, eng, full) == want, "full MSM vs closed form"
    # (2) north_star's split: 8 disjoint Pippenger window ranges, folded on the device in rank order
    for r in range(world):
        lo, hi = pzd.window_range(tb.n_windows, r, world)
        eng.msm_dev(tb, d_s.data_ptr(), 1, n, 4 * n, d_parts[r].data_ptr(), lo, hi)
    eng.g1_sum_dev(d_parts.data_ptr(), world, d_out.data_ptr())
    eng.sync()
    by_windows = d_out.cpu().numpy().astype(np.uint64)[0]
    assert _aff(cref, eng, by_windows) == want, "8 window ranges"
    # the host-pointer fold gives the same point
    assert _aff(cref, eng, eng.g1_sum(d_parts.cpu().numpy().astype(np.uint64))) == want
    tb.free()
    # (3) SURVEY 8e's alternative: 8 disjoint point ranges, each with its own table of n/8 bases
    for r in range(world):
        lo, hi = pzd.point_range(n, r, world)
        tr = eng.load_bases_dev(d_b.data_ptr() + lo * 64, hi - lo)
        eng.msm_dev(tr, d_s.data_ptr() + lo * 32, 1, hi - lo, 4 * (hi - lo), d_parts[r].data_ptr())
        eng.sync()
        tr.free()
    eng.g1_sum_dev(d_parts.data_ptr(), world, d_out.data_ptr())
    eng.sync()
    assert _aff(cref, eng, d_out.cpu().numpy().astype(np.uint64)[0]) == want, "8 point ranges"


def test_c5_msm_2pow19(eng, cref):
    import torch

    n = 1 << 19
    s, t = (0x1B7 << 240) + 0xA5A5A5A5, 0x9E3779B97F4A7C15
    d_b = _walk_bases_dev(eng, torch, n, s, t)
    tb = eng.load_bases_dev(d_b.data_ptr(), n)
    d_out = torch.zeros((2, 12), dtype=torch.int64, device="cuda")
    cols = np.stack([canon_rand_scalars(n, 1901), witness_like_canon(n, 1902)])
    d_s = torch.from_numpy(cols.view(np.int64)).cuda()
    eng.fr_convert_dev(d_s.data_ptr(), 2 * n, True)
    eng.msm_dev(tb, d_s.data_ptr(), 2, n, 4 * n, d_out.data_ptr())
    eng.sync()
    got = d_out.cpu().numpy().astype(np.uint64)
    for j in range(2):
        assert _aff(cref, eng, got[j]) == P.g1_mul(P.G1_GEN, walk_dlog_sum(cols[j], s, t)), j
    tb.free()


def _lagrange_srs_dev(eng, torch, cref, k, s_toxic):
    d_l = torch.zeros((1 << k, 8), dtype=torch.int64, device="cuda")
    eng.srs_setup_g1_dev(k, cref.fr_ints_to_mont([s_toxic])[0], cref.fr_ints_to_mont([P.fr_omega(k)])[0], 0, d_l.data_ptr())
    eng.sync()
    return d_l


def test_c5_3072bit_k19_column_sample(eng, cref):
    """config c5's shape: 3072-bit key (96 limbs of n^2), k = 19, lookup_bits 18.  The whole K3 trace, then three
    of the circuit's ~2.2k advice columns (first, middle, last = ragged) through K4 and K1 against the oracle chain."""
    import torch

    enc_bits, k, lb = 3072, 19, 18
    Ln, L = enc_bits // 64, 2 * (enc_bits // 64)
    rows = column_rows(k)
    nn, g, m, r = P.synth_paillier_inputs(enc_bits, 0x5046)
    arr = lambda x: cref.int_to_limbs(x, Ln)
    c, steps, ng, nr = eng.paillier_encrypt(Ln, arr(nn), arr(g), arr(m), arr(r))
    tot = int(ng[0]) + int(nr[0]) + 1
    assert cref.limbs_to_int(c[0]) == P.paillier_enc_native(nn, g, m, r)
    _, sg, sr, fin = P.encrypt_trace(nn, g, m, r)
    osteps = sg + sr + [fin]
    assert len(osteps) == tot
    cps, lps = eng.witness_cells_per_step(L, 64, lb)
    ncols = -(-(tot * cps) // rows)
    d_steps = torch.from_numpy(steps[0, :tot].astype(np.int64)).cuda()
    d_mod = torch.from_numpy(cref.int_to_limbs(nn * nn, L).astype(np.int64)).cuda()
    d_l = _lagrange_srs_dev(eng, torch, cref, k, 0x7777777 * 0x1111111 + 3)
    tb = eng.load_bases_dev(d_l.data_ptr(), 1 << k)
    bases = d_l.cpu().numpy().astype(np.uint64)
    d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    for j in (0, ncols // 2, ncols - 1):
        c0, c1 = j * rows, min((j + 1) * rows, tot * cps)
        s0, s1 = c0 // cps, -(-c1 // cps)
        # the trace records of these steps are the oracle's
        for si in (s0, s1 - 1):
            got = tuple(cref.limbs_to_int(steps[0, si, q]) for q in range(4))
            assert got == osteps[si], ("trace step", si)
        d_cells = torch.zeros(((s1 - s0) * cps, 4), dtype=torch.int64, device="cuda")
        eng.witness_expand_dev(L, 64, lb, d_steps[s0].data_ptr(), s1 - s0, d_mod.data_ptr(), d_cells.data_ptr(), 0)
        d_col = torch.zeros((rows, 4), dtype=torch.int64, device="cuda")
        d_col[: c1 - c0] = d_cells[c0 - s0 * cps: c1 - s0 * cps]
        eng.msm_dev(tb, d_col.data_ptr(), 1, rows, 4 * rows, d_out.data_ptr())
        eng.sync()
        cells = []
        for (a, b, q, rr) in osteps[s0:s1]:
            cells += P.expand_mul_mod_cells(a, b, q, rr, nn * nn, L, lb)[0]
        col = cells[c0 - s0 * cps: c1 - s0 * cps]
        col += [0] * (rows - len(col))
        col_m = cref.fr_ints_to_mont(col)
        assert np.array_equal(d_col.cpu().numpy().astype(np.uint64), col_m), ("cells of column", j)
        want = cref.g1_normalize(cref.msm_g1(col_m, bases[:rows]))
        assert np.array_equal(eng.g1_normalize(d_out.cpu().numpy().astype(np.uint64))[0], want), ("commitment of column", j)
    tb.free()


def test_c2_encrypt_circuit_k17_at_size(eng, cref):
    """BASELINE config c2 at its full size (bench.rs:120-173 with ENC_BIT_LEN 2048, k 17, lookup_bits 16): the K3 trace on the
    device, the whole circuit's cell stream cut into 2^17-row columns by K4 (12.7 GB), every column committed by K1."""
    import torch

    enc_bits, k, lb = 2048, 17, 16
    Ln, L = enc_bits // 64, 2 * (enc_bits // 64)
    n = 1 << k
    rows = column_rows(k)
    nn, g, m, r = P.synth_paillier_inputs(enc_bits, 0x5043)
    res = P.paillier_enc_native(nn, g, m, r)
    arr = lambda x: cref.int_to_limbs(x, Ln)
    cap = m.bit_length() + bin(m).count("1") + nn.bit_length() + bin(nn).count("1") + 1
    d_steps = torch.zeros((cap, 4, L), dtype=torch.int64, device="cuda")
    c, ng, nr = eng.paillier_encrypt_dev(Ln, arr(nn), arr(g), arr(m), arr(r), d_steps.data_ptr(), cap)
    ng, nr = int(ng[0]), int(nr[0])
    assert cref.limbs_to_int(c[0]) == res and ng + nr + 1 == cap
    adv_n, lk_n = eng.circuit_cells(0, Ln, 64, lb, ng, nr)
    ncol_a, ncol_l = -(-adv_n // rows), -(-lk_n // rows)
    assert (ncol_a, ncol_l) == (3033, 84)
    d_adv = torch.zeros((ncol_a * n, 4), dtype=torch.int64, device="cuda")
    d_lk = torch.zeros((ncol_l * n, 4), dtype=torch.int64, device="cuda")
    d_mod = torch.from_numpy(cref.int_to_limbs(nn * nn, L).astype(np.int64)).cuda()
    inputs = np.concatenate([arr(nn), arr(g), arr(m), arr(r), cref.int_to_limbs(res, L)])
    eng.circuit_expand_dev(0, Ln, 64, lb, inputs, d_steps.data_ptr(), ng, nr, d_mod.data_ptr(), d_adv.data_ptr(), d_lk.data_ptr(), rows, n)
    d_l = _lagrange_srs_dev(eng, torch, cref, k, 0x2222222 * 0x3333333 + 5)
    tb = eng.load_bases_dev(d_l.data_ptr(), n)
    d_ca = torch.zeros((ncol_a, 12), dtype=torch.int64, device="cuda")
    d_cl = torch.zeros((ncol_l, 12), dtype=torch.int64, device="cuda")
    eng.msm_dev(tb, d_adv.data_ptr(), ncol_a, n, 4 * n, d_ca.data_ptr())
    eng.msm_dev(tb, d_lk.data_ptr(), ncol_l, n, 4 * n, d_cl.data_ptr())
    eng.sync()
    com_a = eng.g1_normalize(d_ca.cpu().numpy().astype(np.uint64))
    com_l = eng.g1_normalize(d_cl.cpu().numpy().astype(np.uint64))
    bases = d_l.cpu().numpy().astype(np.uint64)
    # sampled columns against the oracle chain (Python trace -> Python cells -> C best_multiexp)
    sample_a = (0, ncol_a // 3, (2 * ncol_a) // 3, ncol_a - 1)
    sample_l = (0, ncol_l - 1)
    win = lambda j, tot: (j * rows, min((j + 1) * rows, tot))
    tot_a, tot_l, cells_a, cells_l = P.encrypt_circuit_cells_windows(nn, g, m, r, res, enc_bits, 64, lb, [win(j, adv_n) for j in sample_a],
                                                                     [win(j, lk_n) for j in sample_l])
    assert (tot_a, tot_l) == (adv_n, lk_n)
    for buf, com, sample, cells in ((d_adv, com_a, sample_a, cells_a), (d_lk, com_l, sample_l, cells_l)):
        for j, col in zip(sample, cells):
            col_m = cref.fr_ints_to_mont(col + [0] * (n - len(col)))
            got = buf[j * n:(j + 1) * n].cpu().numpy().astype(np.uint64)
            assert np.array_equal(got, col_m), ("cells of column", j)
            want = cref.g1_normalize(cref.msm_g1(col_m, bases))
            assert np.array_equal(com[j], want), ("commitment of column", j)
    assert cells_a[-1][-1] == 1          # assert_equal_fresh's final bit: the circuit is satisfied
    # all columns at once: K1 is linear, sum_j v^(N-1-j) col_j committed == the same combination of the commitments (the
    # combination of the columns by pz_fr_lincomb_dev, that of the 3033 commitments by the oracle's multiexp)
    v = pow(P.FR_GENERATOR, 0x1234567, P.FR_R)
    d_fold = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    d_cf = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    for buf, com, nc in ((d_adv, com_a, ncol_a), (d_lk, com_l, ncol_l)):
        eng.fr_lincomb_dev(buf.data_ptr(), nc, 4 * n, n, cref.fr_ints_to_mont([v])[0], d_fold.data_ptr())
        eng.msm_dev(tb, d_fold.data_ptr(), 1, n, 4 * n, d_cf.data_ptr())
        eng.sync()
        got = eng.g1_normalize(d_cf.cpu().numpy().astype(np.uint64))[0]
        pw = cref.fr_ints_to_mont([pow(v, nc - 1 - j, P.FR_R) for j in range(nc)])
        want = cref.g1_normalize(cref.msm_g1(pw, com))
        assert np.array_equal(got, want), "linearity over all columns"
    tb.free()


def test_c2_uniform_circuit_k17_at_size(eng, cref):
    """The uniform-shape circuit at the c2 key size: 2 * 2048 + ~3070 mul_mod steps, num_to_bits and
    limb-wise selects, 4.6e8 advice cells in 3544 columns -- K3 (uniform schedule) -> K4 (kind 2) -> K1 on the device; sampled
    columns vs the oracle chain, every column through linearity of the commitment."""
    import torch

    enc_bits, k, lb = 2048, 17, 16
    Ln, L = enc_bits // 64, 2 * (enc_bits // 64)
    n = 1 << k
    rows = column_rows(k)
    nn, g, m, r = P.synth_paillier_inputs(enc_bits, 0x5047)
    res = P.paillier_enc_native(nn, g, m, r)
    arr = lambda x: cref.int_to_limbs(x, Ln)
    cap = 2 * enc_bits + nn.bit_length() + bin(nn).count("1") + 1
    d_steps = torch.zeros((cap, 4, L), dtype=torch.int64, device="cuda")
    c, ng, nr = eng.paillier_encrypt_uniform_dev(Ln, enc_bits, arr(nn), arr(g), arr(m), arr(r), d_steps.data_ptr(), cap)
    ng, nr = int(ng[0]), int(nr[0])
    assert cref.limbs_to_int(c[0]) == res and ng == 2 * enc_bits and ng + nr + 1 == cap
    adv_n, lk_n = eng.circuit_cells(2, Ln, 64, lb, ng, nr)
    ncol_a, ncol_l = -(-adv_n // rows), -(-lk_n // rows)
    d_adv = torch.zeros((ncol_a * n, 4), dtype=torch.int64, device="cuda")
    d_lk = torch.zeros((ncol_l * n, 4), dtype=torch.int64, device="cuda")
    d_mod = torch.from_numpy(cref.int_to_limbs(nn * nn, L).astype(np.int64)).cuda()
    inputs = np.concatenate([arr(nn), arr(g), arr(m), arr(r), cref.int_to_limbs(res, L)])
    eng.circuit_expand_dev(2, Ln, 64, lb, inputs, d_steps.data_ptr(), ng, nr, d_mod.data_ptr(), d_adv.data_ptr(), d_lk.data_ptr(), rows, n)
    d_l = _lagrange_srs_dev(eng, torch, cref, k, 0x4444444 * 0x5555555 + 7)
    tb = eng.load_bases_dev(d_l.data_ptr(), n)
    d_ca = torch.zeros((ncol_a, 12), dtype=torch.int64, device="cuda")
    eng.msm_dev(tb, d_adv.data_ptr(), ncol_a, n, 4 * n, d_ca.data_ptr())
    eng.sync()
    com_a = eng.g1_normalize(d_ca.cpu().numpy().astype(np.uint64))
    bases = d_l.cpu().numpy().astype(np.uint64)
    # columns inside num_to_bits / select territory (the first of the g^m chain), in the middle of it, in the r^n chain, the last
    sample_a = (0, 1, ncol_a // 2, ncol_a - 1)
    win = lambda j, tot: (j * rows, min((j + 1) * rows, tot))
    tot_a, tot_l, cells_a, cells_l = P.uniform_circuit_cells_windows(nn, g, m, r, res, enc_bits, 64, lb, [win(j, adv_n) for j in sample_a],
                                                                     [win(0, lk_n), win(ncol_l - 1, lk_n)])
    assert (tot_a, tot_l) == (adv_n, lk_n)
    for j, col in zip(sample_a, cells_a):
        col_m = cref.fr_ints_to_mont(col + [0] * (n - len(col)))
        assert np.array_equal(d_adv[j * n:(j + 1) * n].cpu().numpy().astype(np.uint64), col_m), ("cells of column", j)
        assert np.array_equal(com_a[j], cref.g1_normalize(cref.msm_g1(col_m, bases))), ("commitment of column", j)
    for j, col in zip((0, ncol_l - 1), cells_l):
        col_m = cref.fr_ints_to_mont(col + [0] * (n - len(col)))
        assert np.array_equal(d_lk[j * n:(j + 1) * n].cpu().numpy().astype(np.uint64), col_m), ("lookup cells of column", j)
    assert cells_a[-1][-1] == 1
    v = pow(P.FR_GENERATOR, 0x7654321, P.FR_R)
    d_fold = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    d_cf = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    eng.fr_lincomb_dev(d_adv.data_ptr(), ncol_a, 4 * n, n, cref.fr_ints_to_mont([v])[0], d_fold.data_ptr())
    eng.msm_dev(tb, d_fold.data_ptr(), 1, n, 4 * n, d_cf.data_ptr())
    eng.sync()
    pw = cref.fr_ints_to_mont([pow(v, ncol_a - 1 - j, P.FR_R) for j in range(ncol_a)])
    assert np.array_equal(eng.g1_normalize(d_cf.cpu().numpy().astype(np.uint64))[0], cref.g1_normalize(cref.msm_g1(pw, com_a))), "linearity"
    tb.free()


def test_c3_add_circuit_k15_whole(eng, cref):
    """config c3: c1 * c2 mod n^2 at a 2048-bit key (operands assigned at enc_bits, bench.rs:98-103), k = 15,
    lookup_bits 14 -- K3 (pz_mul_mod) -> K4 -> K1 over ALL of the circuit's columns vs the oracle chain."""
    import random

    import torch

    enc_bits, k, lb = 2048, 15, 14
    L = 2 * (enc_bits // 64)
    rows = column_rows(k)
    rng = random.Random(0x5044)
    nn = P.synth_paillier_inputs(enc_bits, 0x5044)[0]
    c1, c2 = rng.getrandbits(enc_bits), rng.getrandbits(enc_bits)
    n2 = nn * nn
    q, rem = eng.mul_mod(L, cref.int_to_limbs(c1, L), cref.int_to_limbs(c2, L), cref.int_to_limbs(n2, L))
    res, st = P.add_trace(nn, c1, c2)
    assert (cref.limbs_to_int(q), cref.limbs_to_int(rem)) == (st[2], st[3]) and res == P.paillier_add_native(nn, c1, c2)
    cps, lps = eng.witness_cells_per_step(L, 64, lb)
    ncols, nlk = -(-cps // rows), -(-lps // rows)
    step = np.stack([cref.int_to_limbs(x, L) for x in st]).reshape(1, 4, L)
    d_steps = torch.from_numpy(step.astype(np.int64)).cuda()
    d_mod = torch.from_numpy(cref.int_to_limbs(n2, L).astype(np.int64)).cuda()
    d_adv = torch.zeros((ncols * rows, 4), dtype=torch.int64, device="cuda")
    d_lk = torch.zeros((nlk * rows, 4), dtype=torch.int64, device="cuda")
    eng.witness_expand_dev(L, 64, lb, d_steps.data_ptr(), 1, d_mod.data_ptr(), d_adv.data_ptr(), d_lk.data_ptr())
    d_l = _lagrange_srs_dev(eng, torch, cref, k, 0x5151515 * 0x3333333 + 7)
    tb = eng.load_bases_dev(d_l.data_ptr(), 1 << k)
    bases = d_l.cpu().numpy().astype(np.uint64)
    adv, lk = P.expand_mul_mod_cells(*st, n2, L, lb)
    for d_buf, cells, nc in ((d_adv, adv, ncols), (d_lk, lk, nlk)):
        cells = cells + [0] * (nc * rows - len(cells))
        d_out = torch.zeros((nc, 12), dtype=torch.int64, device="cuda")
        eng.msm_dev(tb, d_buf.data_ptr(), nc, rows, 4 * rows, d_out.data_ptr())
        eng.sync()
        got = eng.g1_normalize(d_out.cpu().numpy().astype(np.uint64))
        for j in range(nc):
            col = cref.fr_ints_to_mont(cells[j * rows:(j + 1) * rows])
            assert np.array_equal(got[j], cref.g1_normalize(cref.msm_g1(col, bases[:rows]))), j
    tb.free()


def test_c3_msm_2pow15_vs_oracle(eng, cref):
    import random

    n = 1 << 15
    rng = random.Random(1501)
    bases = cref.walk_bases(n, rng.randrange(1, P.FR_R), rng.randrange(1, P.FR_R))
    tb = eng.load_bases(bases)
    cols = [cref.fr_ints_to_mont([rng.randrange(P.FR_R) for _ in range(n)]), cref.fr_ints_to_mont(P.witness_like_scalars(n, 15))]
    out = eng.msm_batch(tb, cols)
    for j, col in enumerate(cols):
        assert np.array_equal(eng.g1_normalize(out[j])[0], cref.g1_normalize(cref.msm_g1(col, bases))), j
    tb.free()


@pytest.mark.parametrize("world", [1, 3, 8])
def test_msm_multi_contexts_one_call(cref, world):
    """pz_msm_g1_multi: one 2^20-point MSM over `world` contexts (all on this one device: the entry point does not care) in ONE call
    of the C ABI -- window split and point split -- equals the single-context MSM and the closed form; empty shares (more contexts
    than windows would give) are identities"""
    import torch

    import paillier_halo2_amd as pz

    log_n = 20
    n = 1 << log_n
    s, t = (0x19D << 236) + 0x55AA, 0x1234567890ABCDEF1
    engs = [pz.Engine(0) for _ in range(world)]
    engs[0].bind_torch_stream()
    try:
        d_b = _walk_bases_dev(engs[0], torch, n, s, t)
        sc = canon_rand_scalars(n, 2020 + world)
        want = P.g1_mul(P.G1_GEN, walk_dlog_sum(sc, s, t))
        d_s = torch.from_numpy(sc.view(np.int64)).cuda()
        engs[0].fr_convert_dev(d_s.data_ptr(), n, True)
        engs[0].sync()
        torch.cuda.synchronize()
        # north_star's split: every context a table of all bases, its window range
        tabs = [e.load_bases_dev(d_b.data_ptr(), n) for e in engs]
        got = pz.Engine.msm_multi(engs, tabs, [d_s.data_ptr()] * world, [n] * world, split_points=False)
        assert _aff(cref, engs[0], got) == want, ("window split", world)
        for tb in tabs:
            tb.free()
        # point ranges: every context a table of its own bases
        from paillier_halo2_amd import dist as pzd

        rng = [pzd.point_range(n, r, world) for r in range(world)]
        tabs = [engs[r].load_bases_dev(d_b.data_ptr() + lo * 64, hi - lo) for r, (lo, hi) in enumerate(rng)]
        got = pz.Engine.msm_multi(engs, tabs, [d_s.data_ptr() + lo * 32 for lo, _ in rng], [hi - lo for lo, hi in rng], split_points=True)
        assert _aff(cref, engs[0], got) == want, ("point split", world)
        for tb in tabs:
            tb.free()
    finally:
        for e in engs:
            e.close()
