"""GPU parity tests of the "next" rows: grand products, the custom-gate part of
the quotient, division by the vanishing polynomial, kate_division -- through the C ABI against oracle/pyref.py, plus
size-independent properties (the quotient of a REAL K3 -> K4 witness column is a polynomial of the right degree and
satisfies h(x) (x^n - 1) = gate(x) at a random point)."""
import random

import numpy as np
import pytest

from oracle import pyref as P
from tests.util import column_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import paillier_halo2_amd as pz

    e = pz.Engine(0)
    e.bind_torch_stream()  # torch fills / copies and the library's kernels in one order
    yield e
    e.close()


def _dev(cref, ints):
    """list (or list of lists) of field ints -> int64 CUDA tensor of Montgomery limbs"""
    import torch

    a = np.asarray(cref.fr_ints_to_mont([x for row in ints for x in row] if ints and isinstance(ints[0], list) else ints))
    if ints and isinstance(ints[0], list):
        a = a.reshape(len(ints), -1, 4)
    return torch.from_numpy(a.astype(np.int64)).cuda()


def _ints(cref, t):
    return cref.fr_mont_to_ints(t.cpu().numpy().astype(np.uint64).reshape(-1, 4))


def _m(cref, x):
    return cref.fr_ints_to_mont([x % P.FR_R])[0]


@pytest.mark.parametrize("n", [1, 5, 64, 1000, (1 << 17) + 3])
def test_batch_invert_vs_oracle(eng, cref, n):
    rng = random.Random(400 + n)
    a = [rng.randrange(P.FR_R) for _ in range(n)]
    for i in range(0, n, 7):
        a[i] = 0  # zeros stay zero (halo2 BatchInvert)
    d = _dev(cref, a)
    eng.fr_batch_invert_dev(d.data_ptr(), n)
    eng.sync()
    got = _ints(cref, d)
    if n <= 1000:
        assert got == P.batch_invert(a)
    else:  # property at size: a * a^-1 == 1 on the non-zeros, a sample against the oracle
        idx = rng.sample(range(n), 200)
        assert all((got[i] * a[i] % P.FR_R == 1) if a[i] else got[i] == 0 for i in range(n))
        assert [got[i] for i in idx] == P.batch_invert([a[i] for i in idx])


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 4096, 4097, 70001])
def test_prefix_product_vs_oracle(eng, cref, n):
    import torch

    rng = random.Random(410 + n)
    a = [rng.randrange(P.FR_R) for _ in range(n)]
    z0 = rng.randrange(1, P.FR_R)
    d = _dev(cref, a)
    d_z = torch.zeros_like(d)
    eng.fr_prefix_product_dev(d.data_ptr(), n, _m(cref, z0), d_z.data_ptr())
    eng.sync()
    want = P.prefix_product(a, z0)
    assert _ints(cref, d_z) == want
    eng.fr_prefix_product_dev(d.data_ptr(), n, _m(cref, z0), d.data_ptr())  # in place
    eng.sync()
    assert _ints(cref, d) == want


@pytest.mark.parametrize("log_n,m", [(4, 1), (6, 3), (10, 4)])
def test_permutation_product_vs_oracle(eng, cref, log_n, m):
    import torch

    rng = random.Random(420 + log_n)
    n = 1 << log_n
    omega, delta = P.fr_omega(log_n), pow(P.FR_GENERATOR, 1 << P.FR_S, P.FR_R)
    beta, gamma, z0 = (rng.randrange(1, P.FR_R) for _ in range(3))
    dstart = pow(delta, 5, P.FR_R)
    # a real permutation of the m*n cells with values constant on its cycles: the product telescopes to 1
    labels = [[dstart * pow(delta, j, P.FR_R) * pow(omega, i, P.FR_R) % P.FR_R for i in range(n)] for j in range(m)]
    cells = [(j, i) for j in range(m) for i in range(n)]
    perm = list(cells)
    rng.shuffle(perm)
    sigma = [[0] * n for _ in range(m)]
    val = [[None] * n for _ in range(m)]
    for (j, i), (pj, pi) in zip(cells, perm):
        sigma[j][i] = labels[pj][pi]
    for (j, i) in cells:  # fill cycle by cycle
        if val[j][i] is None:
            v = rng.randrange(P.FR_R)
            cj, ci = j, i
            while val[cj][ci] is None:
                val[cj][ci] = v
                cj, ci = perm[cj * n + ci]
    stride = 4 * n + 8
    d_cols = torch.zeros((m, stride), dtype=torch.int64, device="cuda")
    d_sig = torch.zeros((m, stride), dtype=torch.int64, device="cuda")
    d_cols[:, : 4 * n] = _dev(cref, val).reshape(m, 4 * n)
    d_sig[:, : 4 * n] = _dev(cref, sigma).reshape(m, 4 * n)
    d_z = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    eng.permutation_product_dev(d_cols.data_ptr(), stride, d_sig.data_ptr(), stride, m, log_n, _m(cref, omega),
                                _m(cref, beta), _m(cref, gamma), _m(cref, dstart), _m(cref, delta), _m(cref, z0),
                                d_z.data_ptr())
    eng.sync()
    got = _ints(cref, d_z)
    want = P.permutation_product(val, sigma, omega, beta, gamma, dstart, delta, z0)
    assert got == want
    # telescoping: one more factor returns to z0
    i = n - 1
    num = den = 1
    for j in range(m):
        num = num * (val[j][i] + beta * labels[j][i] + gamma) % P.FR_R
        den = den * (val[j][i] + beta * sigma[j][i] + gamma) % P.FR_R
    assert got[-1] * num % P.FR_R == z0 * den % P.FR_R


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 1 << 13])
def test_kate_division_vs_oracle(eng, cref, n):
    import torch

    rng = random.Random(430 + n)
    cols = [[rng.randrange(P.FR_R) for _ in range(n)] for _ in range(3)]
    x = rng.randrange(P.FR_R)
    d = _dev(cref, cols)
    d_q = torch.zeros_like(d)
    eng.poly_div_linear_dev(d.data_ptr(), 3, 4 * n, n, _m(cref, x), d_q.data_ptr(), 4 * n)
    eng.sync()
    got = _ints(cref, d_q)
    for j in range(3):
        assert got[j * n:(j + 1) * n] == P.kate_division(cols[j], x), (n, j)
    eng.poly_div_linear_dev(d.data_ptr(), 3, 4 * n, n, _m(cref, x), d.data_ptr(), 4 * n)  # in place
    eng.sync()
    assert _ints(cref, d) == got


def test_kate_division_identity_at_scale(eng, cref):
    """2^17 coefficients (the c2 column size): p(t) - p(x) == (t - x) q(t) at a random t, via pz_poly_eval_dev"""
    import torch

    n, ncols = 1 << 17, 4
    gen = torch.Generator(device="cuda")
    gen.manual_seed(440)
    d = torch.randint(-(1 << 63), (1 << 63) - 1, (ncols, n, 4), dtype=torch.int64, device="cuda", generator=gen)
    d[:, :, 3] &= 0x0FFFFFFFFFFFFFFF
    rng = random.Random(441)
    x, t = rng.randrange(P.FR_R), rng.randrange(P.FR_R)
    d_q = torch.zeros_like(d)
    eng.poly_div_linear_dev(d.data_ptr(), ncols, 4 * n, n, _m(cref, x), d_q.data_ptr(), 4 * n)
    ev = torch.zeros((3, ncols, 4), dtype=torch.int64, device="cuda")
    eng.poly_eval_dev(d.data_ptr(), ncols, 4 * n, n, _m(cref, t), ev[0].data_ptr())
    eng.poly_eval_dev(d.data_ptr(), ncols, 4 * n, n, _m(cref, x), ev[1].data_ptr())
    eng.poly_eval_dev(d_q.data_ptr(), ncols, 4 * n, n, _m(cref, t), ev[2].data_ptr())
    eng.sync()
    pt, px, qt = (_ints(cref, ev[k]) for k in range(3))
    for j in range(ncols):
        assert (pt[j] - px[j]) % P.FR_R == (t - x) * qt[j] % P.FR_R, j


@pytest.mark.parametrize("n,npts", [(1, 1), (5, 2), (300, 3), (4097, 4), (1 << 17, 4)])
def test_poly_eval_multi_point(eng, cref, n, npts):
    """pz_poly_eval_multi_dev: a batch of polynomials at their whole rotation set in one pass == the oracle's Horner value
    (small n) and == pz_poly_eval_dev point by point (every n)"""
    import torch

    rng = random.Random(450 + n)
    ncols = 3
    w = P.fr_omega(17)
    x0 = rng.randrange(P.FR_R)
    pts = [x0 * pow(w, k, P.FR_R) % P.FR_R for k in (0, 1, P.FR_R - 2, 3)][:npts]     # x, wx, w^-1 x, w^3 x
    if n <= 5000:
        cols = [[rng.randrange(P.FR_R) for _ in range(n)] for _ in range(ncols)]
        d = _dev(cref, cols)
    else:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(451)
        d = torch.randint(-(1 << 63), (1 << 63) - 1, (ncols * n, 4), dtype=torch.int64, device="cuda", generator=gen)
        d[:, 3] &= 0x0FFFFFFFFFFFFFFF
        cols = None
    d_out = torch.zeros((ncols, npts, 4), dtype=torch.int64, device="cuda")
    eng.poly_eval_multi_dev(d.data_ptr(), ncols, 4 * n, n, np.stack([_m(cref, x) for x in pts]), d_out.data_ptr())
    d_one = torch.zeros((npts, ncols, 4), dtype=torch.int64, device="cuda")
    for q, x in enumerate(pts):
        eng.poly_eval_dev(d.data_ptr(), ncols, 4 * n, n, _m(cref, x), d_one[q].data_ptr())
    eng.sync()
    got = _ints(cref, d_out)
    one = _ints(cref, d_one)
    for j in range(ncols):
        for q in range(npts):
            assert got[j * npts + q] == one[q * ncols + j], (j, q)
            if cols is not None:
                assert got[j * npts + q] == P.poly_eval(cols[j], pts[q]), (j, q)


@pytest.mark.parametrize("log_n,log_e,ncols", [(3, 2, 1), (5, 2, 3), (6, 1, 2)])
def test_quotient_gate_finish_distribute_vs_oracle(eng, cref, log_n, log_e, ncols):
    import torch

    rng = random.Random(450 + log_n)
    N, step = 1 << (log_n + log_e), 1 << log_e
    adv = [[rng.randrange(P.FR_R) for _ in range(N)] for _ in range(ncols)]
    sel = [[rng.randrange(P.FR_R) for _ in range(N)] for _ in range(ncols)]
    h0 = [rng.randrange(P.FR_R) for _ in range(N)]
    y, g, c = (rng.randrange(1, P.FR_R) for _ in range(3))
    w_ext = P.fr_omega(log_n + log_e)
    d_a, d_s, d_h = _dev(cref, adv), _dev(cref, sel), _dev(cref, h0)
    eng.quotient_gate_dev(d_a.data_ptr(), 4 * N, d_s.data_ptr(), 4 * N, ncols, log_n + log_e, step, _m(cref, y), d_h.data_ptr())
    eng.sync()
    want = P.quotient_gate(adv, sel, step, y, h0)
    assert _ints(cref, d_h) == want
    eng.quotient_finish_dev(d_h.data_ptr(), log_n, log_e, _m(cref, g), _m(cref, w_ext))
    eng.sync()
    want = P.quotient_finish(want, log_n, log_e, g, w_ext)
    assert _ints(cref, d_h) == want
    eng.fr_distribute_powers_dev(d_h.data_ptr(), 1, 4 * N, N, _m(cref, g), _m(cref, c))
    eng.sync()
    assert _ints(cref, d_h) == P.distribute_powers(want, g, c)
    eng.fr_distribute_powers_dev(d_a.data_ptr(), ncols, 4 * N, N, _m(cref, g))  # c = NULL -> 1, batched
    eng.sync()
    got = _ints(cref, d_a)
    for j in range(ncols):
        assert got[j * N:(j + 1) * N] == P.distribute_powers(adv[j], g), j


def test_quotient_of_real_witness_columns(eng, cref):
    """K3 -> K4 -> K2 -> quotient on the device with a 128-bit key: the advice columns built from the real cell stream
    satisfy the gate on the whole domain, so sum_j y^(..) q_j (a_j + a_j(wX) a_j(w^2 X) - a_j(w^3 X)) is divisible by
    X^n - 1: the quotient the library returns has degree <= 2n - 3 and satisfies the identity at a random point."""
    import torch

    nn, g, m, r = P.synth_paillier_inputs(128, 0x5042, standard_g=False)
    Ln, L, k = 2, 4, 12
    lb, n, E, log_e = k - 1, 1 << k, 4, 2
    arr = lambda x: cref.int_to_limbs(x, Ln)
    _, steps, ng, nr = eng.paillier_encrypt(Ln, arr(nn), arr(g), arr(m), arr(r))
    tot = int(ng[0]) + int(nr[0]) + 1
    adv_n, lk_n = eng.witness_cells_per_step(L, 64, lb)
    per_col = column_rows(k) // adv_n   # whole mul_mod blocks per column: halo2-lib never splits a gate across columns
    assert per_col >= 1
    ncols = 5
    use = ncols * per_col
    assert use <= tot
    d_steps = torch.from_numpy(steps[0, :use].astype(np.int64)).cuda()
    d_mod = torch.from_numpy(cref.int_to_limbs(nn * nn, L).astype(np.int64)).cuda()
    d_cells = torch.zeros((use, adv_n, 4), dtype=torch.int64, device="cuda")
    eng.witness_expand_dev(L, 64, lb, d_steps.data_ptr(), use, d_mod.data_ptr(), d_cells.data_ptr(), 0)
    gates, end = P.gate_offsets_mul_mod(L, lb)
    assert end == adv_n
    d_cols = torch.zeros((ncols, n, 4), dtype=torch.int64, device="cuda")
    sel = np.zeros((ncols, n), dtype=np.uint64)
    for c in range(ncols):
        d_cols[c, : per_col * adv_n] = d_cells[c * per_col:(c + 1) * per_col].reshape(-1, 4)
        for b in range(per_col):
            sel[c, [b * adv_n + o for o in gates]] = 1
    one = np.asarray(_m(cref, 1), dtype=np.uint64)
    d_sel = torch.from_numpy((sel[:, :, None] * one[None, None, :]).astype(np.int64)).cuda()
    # Lagrange -> coefficients -> extended coset, both column sets
    w_n, w_ext = P.fr_omega(k), P.fr_omega(k + log_e)
    cg = P.FR_GENERATOR
    gens = np.stack([_m(cref, cg * pow(w_ext, rr, P.FR_R)) for rr in range(E)])
    ext = []
    for d in (d_cols, d_sel):
        eng.ntt_dev(d.data_ptr(), ncols, 4 * n, _m(cref, pow(w_n, -1, P.FR_R)), k, None, _m(cref, pow(n, -1, P.FR_R)))
        e = torch.zeros((ncols, n * E, 4), dtype=torch.int64, device="cuda")
        eng.ntt_extend_dev(d.data_ptr(), ncols, 4 * n, e.data_ptr(), 4 * n * E, k, log_e, _m(cref, w_n), gens, None)
        ext.append(e)
    rng = random.Random(460)
    y, x = rng.randrange(1, P.FR_R), rng.randrange(2, P.FR_R)
    d_h = torch.zeros((n * E, 4), dtype=torch.int64, device="cuda")
    eng.quotient_gate_dev(ext[0].data_ptr(), 4 * n * E, ext[1].data_ptr(), 4 * n * E, ncols, k + log_e, E, _m(cref, y),
                          d_h.data_ptr())
    eng.quotient_finish_dev(d_h.data_ptr(), k, log_e, _m(cref, cg), _m(cref, w_ext))
    # extended_to_coeff: inverse NTT over the extended domain, then undo the coset shift
    eng.ntt_dev(d_h.data_ptr(), 1, 4 * n * E, _m(cref, pow(w_ext, -1, P.FR_R)), k + log_e, None,
                _m(cref, pow(n * E, -1, P.FR_R)))
    eng.fr_distribute_powers_dev(d_h.data_ptr(), 1, 4 * n * E, n * E, _m(cref, pow(cg, -1, P.FR_R)))
    eng.sync()
    hc = d_h.cpu().numpy()
    assert hc[: 2 * n - 2].any()                 # a non-trivial quotient ...
    assert not hc[2 * n - 2:].any()              # ... of degree <= 2n - 3: the gate expression vanishes on the domain
    # identity at a random point, all evaluations by the library from the coefficient forms
    ev = torch.zeros((6, ncols, 4), dtype=torch.int64, device="cuda")
    for t in range(4):
        eng.poly_eval_dev(d_cols.data_ptr(), ncols, 4 * n, n, _m(cref, x * pow(w_n, t, P.FR_R)), ev[t].data_ptr())
    eng.poly_eval_dev(d_sel.data_ptr(), ncols, 4 * n, n, _m(cref, x), ev[4].data_ptr())
    eng.poly_eval_dev(d_h.data_ptr(), 1, 4 * n * E, n * E, _m(cref, x), ev[5].data_ptr())
    eng.sync()
    a0, a1, a2, a3, q = (_ints(cref, ev[t]) for t in range(5))
    hx = _ints(cref, ev[5])[0]
    lhs = 0
    for j in range(ncols):
        lhs = (lhs * y + q[j] * (a0[j] + a1[j] * a2[j] - a3[j])) % P.FR_R
    assert lhs == hx * (pow(x, n, P.FR_R) - 1) % P.FR_R
    assert lhs != 0


def test_params_kzg_file_to_commitments(eng, cref, tmp_path):
    """SRS through its file format: derive g / g_lagrange on the device (ParamsKZG::setup), write the RawBytes file,
    read it back (memory-mapped), validate on-curve on the device, and commit a column against the file's
    g_lagrange: same point as the commitment against the device-resident SRS.  A corrupted point is detected."""
    import torch

    from paillier_halo2_amd import srs

    k = 10
    n = 1 << k
    rng = random.Random(470)
    s_toxic = rng.randrange(2, P.FR_R)
    d_g = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    d_gl = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.srs_setup_g1_dev(k, _m(cref, s_toxic), _m(cref, P.fr_omega(k)), d_g.data_ptr(), d_gl.data_ptr())
    eng.sync()
    path = str(tmp_path / ("kzg_bn254_%d.srs" % k))
    srs.write_params_kzg(path, k, d_g.cpu().numpy().astype(np.uint64), d_gl.cpu().numpy().astype(np.uint64))
    prm = srs.read_params_kzg(path, expect_k=k)
    d_file = torch.from_numpy(np.ascontiguousarray(prm.g_lagrange).astype(np.int64)).cuda()
    assert eng.g1_check_dev(d_file.data_ptr(), n) == 0
    assert eng.g1_check_dev(torch.from_numpy(np.ascontiguousarray(prm.g).astype(np.int64)).cuda().data_ptr(), n) == 0
    col = [rng.randrange(P.FR_R) for _ in range(n)]
    d_col = _dev(cref, col)
    outs = []
    for src in (d_file, d_gl):
        bases = eng.load_bases_dev(src.data_ptr(), n)
        d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
        eng.msm_dev(bases, d_col.data_ptr(), 1, n, 4 * n, d_out.data_ptr())
        eng.sync()
        outs.append(eng.g1_normalize(d_out.cpu().numpy().astype(np.uint64))[0])
        bases.free()
    assert np.array_equal(outs[0], outs[1])
    # f(s) * G with f the interpolant of the column: commit_lagrange == [f(s)] G
    coeffs = P.intt(col, P.fr_omega(k))
    want = P.g1_mul(P.G1_GEN, P.poly_eval(coeffs, s_toxic))
    assert cref.affine_mont_to_ints(outs[0].reshape(1, 8))[0] == want
    # corruption: flip one limb of one y coordinate, and a non-canonical coordinate
    bad = d_file.clone()
    bad[7, 4] ^= 1
    bad[9, 0:4] = -1
    assert eng.g1_check_dev(bad.data_ptr(), n) == 2
    bad[11] = 0  # the identity is a valid element
    assert eng.g1_check_dev(bad.data_ptr(), n) == 2


@pytest.mark.parametrize("rows,bits,ncols", [(1, 3, 1), (50, 4, 2), (1000, 8, 3), (4086, 11, 2), (column_rows(17), 16, 2)])
def test_lookup_permute_and_product_vs_oracle(eng, cref, rows, bits, ncols):
    """permute_expression_pair (counting sort) and the lookup product vs the oracle's sort + BTreeMap restatement;
    table = {0 .. 2^bits - 1} zero-padded, as halo2-lib's RangeChip lays it out"""
    import torch

    rng = random.Random(480 + rows)
    M = 1 << bits
    table = [i if i < M else 0 for i in range(rows)]
    present = set(table)
    cols = []
    for j in range(ncols):
        if j == 0:   # skewed like range-check digits: many zeros and small values
            c = [rng.choice((0, 0, 1, rng.randrange(M))) for _ in range(rows)]
        else:
            c = [rng.randrange(M) for _ in range(rows)]
        cols.append([v if v in present else 0 for v in c])
    stride = 4 * rows + 4
    d_in = torch.zeros((ncols, stride), dtype=torch.int64, device="cuda")
    d_in[:, : 4 * rows] = _dev(cref, cols).reshape(ncols, 4 * rows)
    d_tab = _dev(cref, table)
    d_pi = torch.zeros((ncols, stride), dtype=torch.int64, device="cuda")
    d_pt = torch.zeros((ncols, stride), dtype=torch.int64, device="cuda")
    eng.lookup_permute_dev(d_in.data_ptr(), ncols, stride, d_tab.data_ptr(), rows, bits, d_pi.data_ptr(), d_pt.data_ptr(), stride)
    eng.sync()
    beta, gamma, z0 = (rng.randrange(1, P.FR_R) for _ in range(3))
    d_z = torch.zeros((ncols, stride), dtype=torch.int64, device="cuda")
    eng.lookup_product_dev(d_in.data_ptr(), stride, d_tab.data_ptr(), d_pi.data_ptr(), stride, d_pt.data_ptr(), stride, ncols, rows,
                           _m(cref, beta), _m(cref, gamma), _m(cref, z0), d_z.data_ptr(), stride)
    eng.sync()
    for j in range(ncols):
        Ap, Sp = P.permute_expression_pair(cols[j], table)
        got_a, got_s = _ints(cref, d_pi[j, : 4 * rows]), _ints(cref, d_pt[j, : 4 * rows])
        assert got_a == Ap, j
        assert got_s == Sp, j
        assert sorted(got_s) == sorted(table)          # S' is a permutation of the table ...
        assert all(got_a[i] == got_s[i] or got_a[i] == got_a[i - 1] for i in range(rows))   # ... and halo2's row rule holds
        z = _ints(cref, d_z[j, : 4 * rows])
        if rows <= 5000:
            assert z == P.lookup_product(cols[j], table, Ap, Sp, beta, gamma, z0), j
        # telescoping: both sides are permutations of each other, so one more factor returns to z0
        i = rows - 1
        num = (cols[j][i] + beta) * (table[i] + gamma) % P.FR_R
        den = (Ap[i] + beta) * (Sp[i] + gamma) % P.FR_R
        assert z[-1] * num % P.FR_R == z0 * den % P.FR_R, j


def test_zero_denominator_and_noncanonical_challenge(eng, cref):
    """(a) a planted zero denominator in the fused num / den path of the grand products: (A'[i] + beta) = 0 at one row -- halo2's
    BatchInvert leaves the inverse of zero at zero, so the factor of that row is 0 (as the oracle's batch_invert gives), not the bare
    numerator; (b) a challenge that is not canonical (>= r) is refused with PZ_ERR_INVALID by the entry points that convert their
    challenges on the host (pz.h), not silently mis-reduced"""
    import torch
    import paillier_halo2_amd as pz
    from paillier_halo2_amd import _lib

    R = P.FR_R
    rng = random.Random(4100)
    n = 64
    beta, gamma = rng.randrange(1, R), rng.randrange(1, R)
    A = [rng.randrange(R) for _ in range(n)]
    S = [rng.randrange(R) for _ in range(n)]
    Ap = [rng.randrange(R) for _ in range(n)]
    Sp = [rng.randrange(R) for _ in range(n)]
    Ap[5] = (-beta) % R
    d_A, d_S, d_Ap, d_Sp = (_dev(cref, x) for x in (A, S, Ap, Sp))
    d_z = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    eng.lookup_product_dev(d_A.data_ptr(), 4 * n, d_S.data_ptr(), d_Ap.data_ptr(), 4 * n, d_Sp.data_ptr(), 4 * n, 1, n, _m(cref, beta),
                           _m(cref, gamma), _m(cref, 1), d_z.data_ptr(), 4 * n)
    eng.sync()
    z = _ints(cref, d_z)
    assert z == P.lookup_product(A, S, Ap, Sp, beta, gamma, 1)
    assert z[5] != 0 and all(v == 0 for v in z[6:])
    # (b)
    bad = np.array([R & 0xFFFFFFFFFFFFFFFF, (R >> 64) & 0xFFFFFFFFFFFFFFFF, (R >> 128) & 0xFFFFFFFFFFFFFFFF, R >> 192], dtype=np.uint64)
    N = 256
    d_a = torch.zeros((N, 4), dtype=torch.int64, device="cuda")
    d_h = torch.zeros((N, 4), dtype=torch.int64, device="cuda")
    with pytest.raises(pz.PzError) as e:
        eng.quotient_gate_dev(d_a.data_ptr(), 4 * N, d_a.data_ptr(), 4 * N, 1, 8, 4, bad, d_h.data_ptr())
    assert e.value.status == _lib.PZ_ERR_INVALID
    with pytest.raises(pz.PzError) as e:
        eng.permutation_product_sets_dev(d_a.data_ptr(), 4 * N, d_a.data_ptr(), 4 * N, 1, 1, 8, N - 6, _m(cref, P.fr_omega(8)), bad,
                                         _m(cref, gamma), _m(cref, 7), d_h.data_ptr(), 4 * N)
    assert e.value.status == _lib.PZ_ERR_INVALID


def test_lookup_permute_rejects_unsatisfiable_inputs(eng, cref):
    import torch

    import paillier_halo2_amd as pz
    from paillier_halo2_amd import _lib

    rows, bits = 64, 4
    table = [i % 16 for i in range(rows)]
    d_tab = _dev(cref, table)
    out = torch.zeros((2, rows, 4), dtype=torch.int64, device="cuda")
    for bad in ([16] + [0] * (rows - 1),               # not below 2^value_bits
                [P.FR_R - 1] + [0] * (rows - 1)):      # a "negative" cell
        d_bad = _dev(cref, bad)
        with pytest.raises(pz.PzError) as e:
            eng.lookup_permute_dev(d_bad.data_ptr(), 1, 4 * rows, d_tab.data_ptr(), rows, bits, out[0].data_ptr(),
                                   out[1].data_ptr(), 4 * rows)
        assert e.value.status == _lib.PZ_ERR_RANGE
    # value in range but absent from the table
    d_nine, d_tab2 = _dev(cref, [9] * rows), _dev(cref, [i % 8 for i in range(rows)])
    with pytest.raises(pz.PzError) as e:
        eng.lookup_permute_dev(d_nine.data_ptr(), 1, 4 * rows, d_tab2.data_ptr(), rows, bits, out[0].data_ptr(),
                               out[1].data_ptr(), 4 * rows)
    assert e.value.status == _lib.PZ_ERR_RANGE


def test_lookup_argument_on_real_witness_digits(eng, cref):
    """K3 -> K4 lookup cell stream (the range-check digits of a real encrypt trace) as lookup columns at k = 12:
    every digit is in the table, the permuted pair obeys halo2's row rule and the product telescopes"""
    import torch

    nn, g, m, r = P.synth_paillier_inputs(128, 0x5042, standard_g=False)
    Ln, L, k = 2, 4, 12
    lb, n = k - 1, 1 << k
    rows = column_rows(k)
    arr = lambda x: cref.int_to_limbs(x, Ln)
    _, steps, ng, nr = eng.paillier_encrypt(Ln, arr(nn), arr(g), arr(m), arr(r))
    tot = int(ng[0]) + int(nr[0]) + 1
    adv_n, lk_n = eng.witness_cells_per_step(L, 64, lb)
    ncols = (tot * lk_n) // rows
    assert ncols >= 2
    ncols = min(ncols, 6)
    d_steps = torch.from_numpy(steps[0, :tot].astype(np.int64)).cuda()
    d_mod = torch.from_numpy(cref.int_to_limbs(nn * nn, L).astype(np.int64)).cuda()
    d_lk = torch.zeros((tot * lk_n, 4), dtype=torch.int64, device="cuda")
    eng.witness_expand_dev(L, 64, lb, d_steps.data_ptr(), tot, d_mod.data_ptr(), 0, d_lk.data_ptr())
    table = [i if i < (1 << lb) else 0 for i in range(rows)]
    d_tab = _dev(cref, table)
    d_pi = torch.zeros((ncols, rows, 4), dtype=torch.int64, device="cuda")
    d_pt = torch.zeros((ncols, rows, 4), dtype=torch.int64, device="cuda")
    eng.lookup_permute_dev(d_lk.data_ptr(), ncols, 4 * rows, d_tab.data_ptr(), rows, lb, d_pi.data_ptr(), d_pt.data_ptr(), 4 * rows)
    rng = random.Random(490)
    beta, gamma = rng.randrange(1, P.FR_R), rng.randrange(1, P.FR_R)
    cells = _ints(cref, d_lk[: ncols * rows])
    d_z = torch.zeros((ncols, rows, 4), dtype=torch.int64, device="cuda")
    eng.lookup_product_dev(d_lk.data_ptr(), 4 * rows, d_tab.data_ptr(), d_pi.data_ptr(), 4 * rows, d_pt.data_ptr(), 4 * rows, ncols,
                           rows, _m(cref, beta), _m(cref, gamma), _m(cref, 1), d_z.data_ptr(), 4 * rows)
    eng.sync()
    for j in range(ncols):
        A = cells[j * rows:(j + 1) * rows]
        Ap, Sp = _ints(cref, d_pi[j]), _ints(cref, d_pt[j])
        assert Ap == sorted(A) and sorted(Sp) == sorted(table)
        assert all(Ap[i] == Sp[i] or Ap[i] == Ap[i - 1] for i in range(rows))
        z = _ints(cref, d_z[j])
        i = rows - 1
        assert z[-1] * (A[i] + beta) % P.FR_R * (table[i] + gamma) % P.FR_R == (Ap[i] + beta) * (Sp[i] + gamma) % P.FR_R


def test_full_quotient_of_a_small_circuit(eng, cref):
    """A complete evaluate_h on the device for a small halo2-shaped instance (k = 6, 5 blinding rows): 5 permuted
    columns in 3 sets, 2 range-check lookups, blinded tails -- products, permuted columns and every constraint term
    computed by the library.  (i) each term array equals the oracle's restatement of halo2's formulas; (ii) the
    numerator vanishes on the domain: after the division the quotient has degree <= 3n - 4."""
    import torch

    R = P.FR_R
    k, log_e, bf = 6, 2, 5
    n, E = 1 << k, 1 << log_e
    N = n * E
    u = n - (bf + 1)                                   # usable rows; row u is the "last" row
    rng = random.Random(500)
    w_n, w_ext, cg = P.fr_omega(k), P.fr_omega(k + log_e), pow(P.FR_GENERATOR, (R - 1) // 3, R)   # coset gen = ZETA
    delta = pow(P.FR_GENERATOR, 1 << P.FR_S, R)
    beta, gamma, y = (rng.randrange(1, R) for _ in range(3))
    m, chunk = 5, 2
    nsets = -(-m // chunk)
    # permutation over the usable rows of all columns; identity elsewhere
    labels = [[pow(delta, c, R) * pow(w_n, i, R) % R for i in range(n)] for c in range(m)]
    cells = [(c, i) for c in range(m) for i in range(u)]
    image = list(cells)
    rng.shuffle(image)
    to = dict(zip(cells, image))
    sigma = [list(labels[c]) for c in range(m)]
    val = [[rng.randrange(R) for _ in range(n)] for _ in range(m)]     # blinded rows keep their random values
    seen = set()
    for cell in cells:
        if cell in seen:
            continue
        v, cur = rng.randrange(R), cell
        while cur not in seen:
            seen.add(cur)
            val[cur[0]][cur[1]] = v
            cur = to[cur]
    for (c, i), (pc, pi) in to.items():
        sigma[c][i] = labels[pc][pi]
    d_val, d_sig = _dev(cref, val), _dev(cref, sigma)
    # permutation products: all sets in one call (z_j[0] = z_{j-1}[u]); cross-checked set by set against the single-chunk
    # entry point and the oracle; blinding rows randomised afterwards
    d_z = torch.zeros((nsets, n, 4), dtype=torch.int64, device="cuda")
    eng.permutation_product_sets_dev(d_val.data_ptr(), 4 * n, d_sig.data_ptr(), 4 * n, m, chunk, k, u, _m(cref, w_n), _m(cref, beta),
                                     _m(cref, gamma), _m(cref, delta), d_z.data_ptr(), 4 * n)
    d_z1 = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    z0 = 1
    z_sets = []
    for j in range(nsets):
        c0, mc = j * chunk, min(chunk, m - j * chunk)
        eng.permutation_product_dev(d_val[c0].data_ptr(), 4 * n, d_sig[c0].data_ptr(), 4 * n, mc, k, _m(cref, w_n), _m(cref, beta),
                                    _m(cref, gamma), _m(cref, pow(delta, c0, R)), _m(cref, delta), _m(cref, z0), d_z1.data_ptr())
        eng.sync()
        zj = _ints(cref, d_z[j])
        want = P.permutation_product(val[c0:c0 + mc], sigma[c0:c0 + mc], w_n, beta, gamma, pow(delta, c0, R), delta, z0)
        assert zj == want, j
        assert _ints(cref, d_z1) == want, j
        zj[u + 1:] = [rng.randrange(R) for _ in range(n - u - 1)]
        z_sets.append(zj)
        z0 = zj[u]
    assert z_sets[-1][u] == 1          # the grand product closes
    d_z = _dev(cref, z_sets)
    # lookups: 2 columns of 4-bit digits on the usable rows, table {0..15} zero-padded
    lbits, nl = 4, 2
    table = [i if i < (1 << lbits) else 0 for i in range(n)]
    A = [[rng.randrange(1 << lbits) if i < u else rng.randrange(R) for i in range(n)] for _ in range(nl)]
    d_A, d_S = _dev(cref, A), _dev(cref, table)
    d_Ap = torch.zeros((nl, n, 4), dtype=torch.int64, device="cuda")
    d_Sp = torch.zeros((nl, n, 4), dtype=torch.int64, device="cuda")
    eng.lookup_permute_dev(d_A.data_ptr(), nl, 4 * n, d_S.data_ptr(), u, lbits, d_Ap.data_ptr(), d_Sp.data_ptr(), 4 * n)
    eng.sync()
    Ap = [_ints(cref, d_Ap[j]) for j in range(nl)]
    Sp = [_ints(cref, d_Sp[j]) for j in range(nl)]
    zl = []
    for j in range(nl):
        assert (Ap[j][:u], Sp[j][:u]) == P.permute_expression_pair(A[j][:u], table[:u])
        for arr in (Ap[j], Sp[j]):
            arr[u:] = [rng.randrange(R) for _ in range(n - u)]
    d_Ap, d_Sp = _dev(cref, Ap), _dev(cref, Sp)
    d_zl = torch.zeros((nl, n, 4), dtype=torch.int64, device="cuda")
    eng.lookup_product_dev(d_A.data_ptr(), 4 * n, d_S.data_ptr(), d_Ap.data_ptr(), 4 * n, d_Sp.data_ptr(), 4 * n, nl, n,
                           _m(cref, beta), _m(cref, gamma), _m(cref, 1), d_zl.data_ptr(), 4 * n)
    eng.sync()
    for j in range(nl):
        z = _ints(cref, d_zl[j])
        assert z[u] == 1, j
        z[u + 1:] = [rng.randrange(R) for _ in range(n - u - 1)]
        zl.append(z)
    d_zl = _dev(cref, zl)
    l0 = [1] + [0] * (n - 1)
    l_last = [1 if i == u else 0 for i in range(n)]
    l_active = [1 if i < u else 0 for i in range(n)]
    d_l = _dev(cref, [l0, l_last, l_active])

    gens = np.stack([_m(cref, cg * pow(w_ext, rr, R)) for rr in range(E)])

    def extend(d, ncols):
        """Lagrange values -> extended coset values, on the device (in place iNTT, then coeff_to_extended)"""
        eng.ntt_dev(d.data_ptr(), ncols, 4 * n, _m(cref, pow(w_n, -1, R)), k, None, _m(cref, pow(n, -1, R)))
        e = torch.zeros((ncols, N, 4), dtype=torch.int64, device="cuda")
        eng.ntt_extend_dev(d.data_ptr(), ncols, 4 * n, e.data_ptr(), 4 * N, k, log_e, _m(cref, w_n), gens, None)
        return e

    e_val, e_sig, e_z = extend(d_val, m), extend(d_sig, m), extend(d_z, nsets)
    e_A, e_S, e_Ap, e_Sp, e_zl, e_l = extend(d_A, nl), extend(d_S.reshape(1, n, 4), 1), extend(d_Ap, nl), extend(d_Sp, nl), extend(d_zl, nl), extend(d_l, 3)
    d_h = torch.zeros((N, 4), dtype=torch.int64, device="cuda")
    eng.quotient_permutation_dev(e_val.data_ptr(), 4 * N, e_sig.data_ptr(), 4 * N, e_z.data_ptr(), 4 * N, nsets, chunk, m,
                                 k + log_e, E, bf + 1, e_l[0].data_ptr(), e_l[1].data_ptr(), e_l[2].data_ptr(), _m(cref, beta),
                                 _m(cref, gamma), _m(cref, delta), _m(cref, cg), _m(cref, w_ext), _m(cref, y), d_h.data_ptr())
    eng.sync()
    I = lambda t: [_ints(cref, t[j]) for j in range(t.shape[0])]
    lv = I(e_l)
    want = P.quotient_permutation(I(e_val), I(e_sig), I(e_z), chunk, E, bf + 1, lv[0], lv[1], lv[2], beta, gamma, delta, cg,
                                  w_ext, y, [0] * N)
    assert _ints(cref, d_h) == want
    eng.quotient_lookup_dev(e_A.data_ptr(), 4 * N, e_S.data_ptr(), e_Ap.data_ptr(), 4 * N, e_Sp.data_ptr(), 4 * N,
                            e_zl.data_ptr(), 4 * N, nl, k + log_e, E, e_l[0].data_ptr(), e_l[1].data_ptr(), e_l[2].data_ptr(),
                            _m(cref, beta), _m(cref, gamma), _m(cref, y), d_h.data_ptr())
    eng.sync()
    want = P.quotient_lookup(I(e_A), I(e_S)[0], I(e_Ap), I(e_Sp), I(e_zl), E, lv[0], lv[1], lv[2], beta, gamma, y, want)
    assert _ints(cref, d_h) == want
    # the same lines added set range by set range (pz_quotient_permutation_part_dev: how the prover streams tiles of extended
    # columns): head + sets [0, 2), then set [2, 3) with its own column pointers -- bit for bit the single call's h
    d_hp = torch.zeros((N, 4), dtype=torch.int64, device="cuda")
    for set_lo, ns in ((0, 2), (2, 1)):
        c0 = set_lo * chunk
        cnt = min(m - c0, ns * chunk)
        eng.quotient_permutation_part_dev(e_val[c0].data_ptr(), 4 * N, e_sig[c0].data_ptr(), 4 * N, e_z.data_ptr(), 4 * N, nsets, set_lo, ns,
                                          chunk, cnt, set_lo == 0, k + log_e, E, bf + 1, e_l[0].data_ptr(), e_l[1].data_ptr(), e_l[2].data_ptr(),
                                          _m(cref, beta), _m(cref, gamma), _m(cref, delta), _m(cref, cg), _m(cref, w_ext), _m(cref, y), d_hp.data_ptr())
    eng.sync()
    assert _ints(cref, d_hp) == P.quotient_permutation(I(e_val), I(e_sig), I(e_z), chunk, E, bf + 1, lv[0], lv[1], lv[2], beta, gamma, delta, cg,
                                                       w_ext, y, [0] * N)
    # divide by X^n - 1 and return to coefficients: the quotient is a polynomial of degree <= 3n - 4
    eng.quotient_finish_dev(d_h.data_ptr(), k, log_e, _m(cref, cg), _m(cref, w_ext))
    eng.ntt_dev(d_h.data_ptr(), 1, 4 * N, _m(cref, pow(w_ext, -1, R)), k + log_e, None, _m(cref, pow(N, -1, R)))
    eng.fr_distribute_powers_dev(d_h.data_ptr(), 1, 4 * N, N, _m(cref, pow(cg, -1, R)))
    eng.sync()
    hc = _ints(cref, d_h)
    assert any(hc[: 3 * n - 3]) and not any(hc[3 * n - 3:])
    # negative control: break one copy constraint and the numerator stops vanishing on the domain
    val[0][3] = (val[0][3] + 1) % R
    e_bad = extend(_dev(cref, val), m)
    d_h2 = torch.zeros((N, 4), dtype=torch.int64, device="cuda")
    eng.quotient_permutation_dev(e_bad.data_ptr(), 4 * N, e_sig.data_ptr(), 4 * N, e_z.data_ptr(), 4 * N, nsets, chunk, m,
                                 k + log_e, E, bf + 1, e_l[0].data_ptr(), e_l[1].data_ptr(), e_l[2].data_ptr(), _m(cref, beta),
                                 _m(cref, gamma), _m(cref, delta), _m(cref, cg), _m(cref, w_ext), _m(cref, y), d_h2.data_ptr())
    eng.quotient_finish_dev(d_h2.data_ptr(), k, log_e, _m(cref, cg), _m(cref, w_ext))
    eng.ntt_dev(d_h2.data_ptr(), 1, 4 * N, _m(cref, pow(w_ext, -1, R)), k + log_e, None, _m(cref, pow(N, -1, R)))
    eng.fr_distribute_powers_dev(d_h2.data_ptr(), 1, 4 * N, N, _m(cref, pow(cg, -1, R)))
    eng.sync()
    assert any(_ints(cref, d_h2)[3 * n - 3:])


def test_multiopen_fold_and_division(eng, cref):
    """multiopen building blocks: w(X) = sum_j v^(m-1-j) p_j(X) folded on the device (continued across calls), divided
    by (X - x): the fold equals the oracle's, the quotient equals kate_division's and (X - x) q(X) = w(X) - w(x)."""
    import torch

    rng = random.Random(510)
    R = P.FR_R
    n, m = 1 << 10, 5
    polys = [[rng.randrange(R) for _ in range(n)] for _ in range(m)]
    v, x, t = (rng.randrange(1, R) for _ in range(3))
    d_p = _dev(cref, polys)
    d_w = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    eng.fr_lincomb_dev(d_p.data_ptr(), 2, 4 * n, n, _m(cref, v), d_w.data_ptr())                    # first two columns ...
    eng.fr_lincomb_dev(d_p[2].data_ptr(), m - 2, 4 * n, n, _m(cref, v), d_w.data_ptr(), True)      # ... continued across calls
    eng.sync()
    want = [0] * n
    for j in range(m):
        want = [(a * v + b) % R for a, b in zip(want, polys[j])]
    assert _ints(cref, d_w) == want
    d_q = torch.zeros_like(d_w)
    eng.poly_div_linear_dev(d_w.data_ptr(), 1, 4 * n, n, _m(cref, x), d_q.data_ptr(), 4 * n)
    eng.sync()
    q = _ints(cref, d_q)
    assert q == P.kate_division(want, x)
    assert (P.poly_eval(want, t) - P.poly_eval(want, x)) % R == (t - x) * P.poly_eval(q, t) % R


def test_keygen_sigma_and_resident_columns(eng, cref):
    """SURVEY 8f rank 2, keygen on the device: the permutation polynomials of a given copy-constraint structure
    (sigma_j[i] = delta^col' omega^row') and keygen_vk / keygen_pk of a batch of fixed columns -- commitments, coefficient
    forms and extended-coset forms left resident -- against the oracle restatements."""
    import torch

    rng = random.Random(5150)
    k, m, log_e = 6, 3, 2
    n, E = 1 << k, 1 << log_e
    w, delta = P.fr_omega(k), pow(P.FR_GENERATOR, 1 << 28, P.FR_R)
    # identity mapping with one 3-cycle and one 2-cycle of copy constraints
    mc = np.repeat(np.arange(m, dtype=np.uint32), n).reshape(m, n)
    mr = np.tile(np.arange(n, dtype=np.uint32), m).reshape(m, n)
    for cyc in ([(0, 1), (2, 5), (1, 7)], [(1, 0), (1, 63)]):
        for a, b in zip(cyc, cyc[1:] + cyc[:1]):
            mc[a], mr[a] = b[0], b[1]
    d_mc, d_mr = torch.from_numpy(mc.astype(np.int32)).cuda(), torch.from_numpy(mr.astype(np.int32)).cuda()
    d_sig = torch.zeros((m, n, 4), dtype=torch.int64, device="cuda")
    eng.permutation_sigma_dev(d_mc.data_ptr(), d_mr.data_ptr(), m, k, cref.fr_ints_to_mont([w])[0], cref.fr_ints_to_mont([delta])[0],
                              d_sig.data_ptr(), 4 * n)
    eng.sync()
    got = d_sig.cpu().numpy().astype(np.uint64)
    for j in range(m):
        want = [pow(delta, int(mc[j, i]), P.FR_R) * pow(w, int(mr[j, i]), P.FR_R) % P.FR_R for i in range(n)]
        assert cref.fr_mont_to_ints(got[j]) == want, j
    # an image outside the call's cells (here: column m) is clamped, never dereferenced, and reported at the next synchronisation
    import paillier_halo2_amd as pz_
    from paillier_halo2_amd import _lib as lib_

    bad = mc.copy()
    bad[1, 5] = m
    d_bad = torch.from_numpy(bad.astype(np.int32)).cuda()
    d_tmp = torch.zeros((m, n, 4), dtype=torch.int64, device="cuda")
    eng.permutation_sigma_dev(d_bad.data_ptr(), d_mr.data_ptr(), m, k, cref.fr_ints_to_mont([w])[0], cref.fr_ints_to_mont([delta])[0],
                              d_tmp.data_ptr(), 4 * n)
    with pytest.raises(pz_.PzError) as ei:
        eng.sync()
    assert ei.value.status == lib_.PZ_ERR_ASYNC
    eng.sync()       # the flag is cleared by the report
    # keygen of these sigma columns + a selector column: commit, coefficient form, extended coset
    sel = [rng.getrandbits(1) for _ in range(n)]
    cols = np.concatenate([got, cref.fr_ints_to_mont(sel).reshape(1, n, 4)])
    nc = cols.shape[0]
    s_tox = rng.randrange(2, P.FR_R)
    d_l = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.srs_setup_g1_dev(k, cref.fr_ints_to_mont([s_tox])[0], cref.fr_ints_to_mont([w])[0], 0, d_l.data_ptr())
    eng.sync()
    tb = eng.load_bases_dev(d_l.data_ptr(), n)
    bases = d_l.cpu().numpy().astype(np.uint64)
    d_cols = torch.from_numpy(cols.astype(np.int64)).cuda()
    d_com = torch.zeros((nc, 12), dtype=torch.int64, device="cuda")
    d_ext = torch.zeros((nc, n * E, 4), dtype=torch.int64, device="cuda")
    w_ext = P.fr_omega(k + log_e)
    gens = np.stack([cref.fr_ints_to_mont([7 * pow(w_ext, r, P.FR_R) % P.FR_R])[0] for r in range(E)])
    eng.keygen_columns_dev(tb, d_cols.data_ptr(), nc, 4 * n, k, log_e, cref.fr_ints_to_mont([w])[0], cref.fr_ints_to_mont([pow(w, -1, P.FR_R)])[0],
                           cref.fr_ints_to_mont([pow(n, -1, P.FR_R)])[0], gens, d_com.data_ptr(), d_ext.data_ptr(), 4 * n * E)
    eng.sync()
    com = eng.g1_normalize(d_com.cpu().numpy().astype(np.uint64))
    coeff = d_cols.cpu().numpy().astype(np.uint64)
    ext = d_ext.cpu().numpy().astype(np.uint64)
    for j in range(nc):
        vals = cref.fr_mont_to_ints(cols[j])
        assert np.array_equal(com[j], cref.g1_normalize(cref.msm_g1(cols[j], bases))), j
        cf = P.intt(vals, w)
        assert cref.fr_mont_to_ints(coeff[j]) == cf, j
        want_ext = P.ntt(P.coset_scale(cf + [0] * (n * E - n), 7), w_ext)
        assert cref.fr_mont_to_ints(ext[j]) == want_ext, j
    tb.free()


@pytest.mark.parametrize("k", [5, 8])
def test_shplonk_two_commitments(eng, cref, k):
    """SHPLONK's multi-point batching on the device -- both output polynomials vs the oracle restatement,
    their commitments with a known-scalar SRS, and the opening identity in the exponent:
    sum_k v^k z_k (sum_j y^j [p_kj(s)] - R_k(u)) G - Z_T(u) H == z_0 (s - u) H'."""
    import torch

    rng = random.Random(600 + k)
    n = 1 << k
    w = P.fr_omega(k)
    x = rng.randrange(P.FR_R)
    points = [x, x * w % P.FR_R, x * pow(w, -1, P.FR_R) % P.FR_R, x * pow(w, n - 11, P.FR_R) % P.FR_R]
    npoly = 7
    polys = [[rng.randrange(P.FR_R) for _ in range(n)] for _ in range(npoly)]
    groups = [([0, 1, 2], [0]), ([3, 4], [0, 1]), ([5], [0, 1, 2]), ([6], [0, 3])]   # (polynomial ids, point indices)
    y, v, u, s_tox = (rng.randrange(2, P.FR_R) for _ in range(4))
    d_p = torch.from_numpy(np.stack([cref.fr_ints_to_mont(p) for p in polys]).astype(np.int64)).cuda()
    sets_dev, sets_ref = [], []
    for ids, idx in groups:
        ev = np.stack([cref.fr_ints_to_mont([P.poly_eval(polys[i], points[t]) for t in idx]) for i in ids])
        sets_dev.append(([d_p[i].data_ptr() for i in ids], idx, ev))
        sets_ref.append(([polys[i] for i in ids], idx))
    d_h = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    d_h2 = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    F = lambda val: cref.fr_ints_to_mont([val])[0]
    st = eng.shplonk_begin_dev(n, sets_dev, cref.fr_ints_to_mont(points), F(y), F(v), d_h.data_ptr())
    eng.sync()
    want_h, want_h2, z0 = P.shplonk_h2(sets_ref, points, y, v, u, n)
    assert cref.fr_mont_to_ints(d_h.cpu().numpy().astype(np.uint64)) == want_h
    eng.shplonk_finish_dev(st, F(u), d_h.data_ptr(), d_h2.data_ptr())
    eng.sync()
    assert cref.fr_mont_to_ints(d_h2.cpu().numpy().astype(np.uint64)) == want_h2
    # the two commitments (monomial SRS from a known scalar) and the verifier's identity in the exponent
    d_g = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.srs_setup_g1_dev(k, F(s_tox), F(w), d_g.data_ptr(), 0)
    eng.sync()
    tb = eng.load_bases_dev(d_g.data_ptr(), n)
    d_out = torch.zeros((2, 12), dtype=torch.int64, device="cuda")
    eng.msm_dev(tb, d_h.data_ptr(), 1, n, 4 * n, d_out[0].data_ptr())
    eng.msm_dev(tb, d_h2.data_ptr(), 1, n, 4 * n, d_out[1].data_ptr())
    eng.sync()
    H, H2 = cref.affine_mont_to_ints(eng.g1_normalize(d_out.cpu().numpy().astype(np.uint64)))
    assert H == P.g1_mul(P.G1_GEN, P.poly_eval(want_h, s_tox)) and H2 == P.g1_mul(P.G1_GEN, P.poly_eval(want_h2, s_tox))
    zt = 1
    for t in points:
        zt = zt * (u - t) % P.FR_R
    acc = 0
    for kk, (ps, idx) in enumerate(sets_ref):
        zk = 1
        for t, pt in enumerate(points):
            if t not in idx:
                zk = zk * (u - pt) % P.FR_R
        C_s = sum(pow(y, j, P.FR_R) * P.poly_eval(p, s_tox) for j, p in enumerate(ps)) % P.FR_R
        xs = [points[i] for i in idx]
        R = P.interpolate(xs, [sum(pow(y, j, P.FR_R) * P.poly_eval(p, xx) for j, p in enumerate(ps)) % P.FR_R for xx in xs])
        acc = (acc + pow(v, kk, P.FR_R) * zk * (C_s - P.poly_eval(R, u))) % P.FR_R
    lhs = P.g1_add_aff(P.g1_mul(P.G1_GEN, acc), P.aff_neg(P.g1_mul(H, zt)))
    assert lhs == P.g1_mul(H2, z0 * (s_tox - u) % P.FR_R)
    tb.free()


def test_shplonk_identities_at_scale(eng, cref):
    """SHPLONK at the bench's scale (k = 17, 460 polynomials in four rotation sets: the chunked parallel folds): both output
    polynomials satisfy their defining identities at a random point t,
        h(t)  = sum_k v^k (C_k(t) - R_k(t)) / Z_{S_k}(t),   C_k = sum_j y^j p_kj,  R_k = the interpolation of C_k on S_k
        h'(t) z_0 (t - u) = sum_k v^k z_k (C_k(t) - R_k(u)) - Z_T(u) h(t)
    with every polynomial evaluated on the device (pz_poly_eval_multi_dev, itself checked against the oracle)."""
    import torch

    k = 17
    n = 1 << k
    rng = random.Random(700)
    w = P.fr_omega(k)
    x = rng.randrange(P.FR_R)
    points = [x, x * w % P.FR_R, x * pow(w, 2, P.FR_R) % P.FR_R, x * pow(w, 3, P.FR_R) % P.FR_R, x * pow(w, -1, P.FR_R) % P.FR_R,
              x * pow(w, n - 11, P.FR_R) % P.FR_R]
    groups = [(300, [0]), (100, [0, 1, 2, 3]), (40, [0, 1, 4]), (20, [0, 1, 5])]
    npoly = sum(c for c, _ in groups)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(701)
    d_p = torch.randint(-(1 << 63), (1 << 63) - 1, (npoly, n, 4), dtype=torch.int64, device="cuda", generator=gen)
    d_p[:, :, 3] &= 0x0FFFFFFFFFFFFFFF
    y, v, u, t = (rng.randrange(2, P.FR_R) for _ in range(4))
    F = lambda val: cref.fr_ints_to_mont([val])[0]
    pts_m = cref.fr_ints_to_mont(points)

    def evals(d_polys, count, xs):   # (count, len(xs)) integers
        d_o = torch.zeros((count, len(xs), 4), dtype=torch.int64, device="cuda")
        eng.poly_eval_multi_dev(d_polys.data_ptr(), count, 4 * n, n, cref.fr_ints_to_mont(xs), d_o.data_ptr())
        eng.sync()
        flat = cref.fr_mont_to_ints(d_o.cpu().numpy().astype(np.uint64).reshape(-1, 4))
        return [flat[i * len(xs):(i + 1) * len(xs)] for i in range(count)]

    sets_dev, ev_sets, off = [], [], 0
    for cnt, idx in groups:
        ev = evals(d_p[off:off + cnt], cnt, [points[i] for i in idx])
        ev_sets.append(ev)
        ev_m = np.stack([cref.fr_ints_to_mont(row) for row in ev])
        sets_dev.append(([d_p[off + i].data_ptr() for i in range(cnt)], idx, ev_m))
        off += cnt
    d_h = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    d_h2 = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    st = eng.shplonk_begin_dev(n, sets_dev, pts_m, F(y), F(v), d_h.data_ptr())
    d_hc = d_h.clone()
    eng.shplonk_finish_dev(st, F(u), d_h.data_ptr(), d_h2.data_ptr())
    eng.sync()
    h_t = evals(d_hc.view(1, n, 4), 1, [t])[0][0]
    h2_t = evals(d_h2.view(1, n, 4), 1, [t])[0][0]
    zt_u = 1
    for pt in points:
        zt_u = zt_u * (u - pt) % P.FR_R
    rhs_h, rhs_l, z0, off = 0, (-zt_u * h_t) % P.FR_R, None, 0
    for kk, ((cnt, idx), ev) in enumerate(zip(groups, ev_sets)):
        p_t = [e[0] for e in evals(d_p[off:off + cnt], cnt, [t])]
        off += cnt
        C_t = sum(pow(y, j, P.FR_R) * p_t[j] for j in range(cnt)) % P.FR_R
        xs = [points[i] for i in idx]
        R = P.interpolate(xs, [sum(pow(y, j, P.FR_R) * ev[j][q] for j in range(cnt)) % P.FR_R for q in range(len(xs))])
        zs_t, zk_u = 1, 1
        for q, pt in enumerate(points):
            if q in idx:
                zs_t = zs_t * (t - pt) % P.FR_R
            else:
                zk_u = zk_u * (u - pt) % P.FR_R
        if kk == 0:
            z0 = zk_u
        vk = pow(v, kk, P.FR_R)
        rhs_h = (rhs_h + vk * (C_t - P.poly_eval(R, t)) * pow(zs_t, -1, P.FR_R)) % P.FR_R
        rhs_l = (rhs_l + vk * zk_u * (C_t - P.poly_eval(R, u))) % P.FR_R
    assert h_t == rhs_h, "h(t)"
    assert h2_t * z0 % P.FR_R * ((t - u) % P.FR_R) % P.FR_R == rhs_l, "h'(t)"


def test_quotient_kernels_at_scale_on_sampled_rows(eng, cref):
    """evaluate_h at the c2 sizes (extended domain 2^19, 64 columns / 32 permutation sets / 16 lookups per call, the calls
    chained through h as bench.py chains them): the three device kernels against the oracle's formulas on sampled rows --
    the first and last rows (where the rotations wrap around), and random ones."""
    import torch

    k, log_e = 17, 2
    N = 1 << (k + log_e)
    step = 1 << log_e
    ncol, nsets, nlk, last_rot = 64, 32, 16, 10
    gen = torch.Generator(device="cuda")
    gen.manual_seed(800)

    def rnd(count):
        x = torch.randint(-(1 << 63), (1 << 63) - 1, (count, N, 4), dtype=torch.int64, device="cuda", generator=gen)
        x[:, :, 3] &= 0x0FFFFFFFFFFFFFFF
        return x

    cols, sigma, z, sel, lp = rnd(ncol), rnd(ncol), rnd(nsets), rnd(ncol), rnd(3)
    lk_a, lk_ap, lk_sp, lk_z, lk_s = rnd(nlk), rnd(nlk), rnd(nlk), rnd(nlk), rnd(1)
    d_h = rnd(1)[0].contiguous()
    rng = random.Random(801)
    beta, gamma, delta, y, x0 = (rng.randrange(2, P.FR_R) for _ in range(5))
    w_ext = P.fr_omega(k + log_e)
    rows = sorted(set([0, 1, 2, step, N - 1, N - step, N - 3 * step - 1, N - last_rot * step] + [rng.randrange(N) for _ in range(12)]))
    need = sorted(set((i + d) % N for i in rows for d in (0, step, 2 * step, 3 * step, -step, -last_rot * step)))
    idx = torch.tensor(need, device="cuda")

    def sparse(t):   # (count, N, 4) -> list of {row: int}
        vals = cref.fr_mont_to_ints(t[:, idx, :].cpu().numpy().astype(np.uint64).reshape(-1, 4))
        return [dict(zip(need, vals[j * len(need):(j + 1) * len(need)])) for j in range(t.shape[0])]

    S = {name: sparse(t) for name, t in dict(cols=cols, sigma=sigma, z=z, sel=sel, lp=lp, a=lk_a, ap=lk_ap, sp=lk_sp, lz=lk_z, s=lk_s).items()}
    h0 = sparse(d_h.view(1, N, 4))[0]
    F = lambda v_: _m(cref, v_)
    # device: gate, then permutation, then lookups, all accumulating into h
    eng.quotient_gate_dev(cols.data_ptr(), 4 * N, sel.data_ptr(), 4 * N, ncol, k + log_e, step, F(y), d_h.data_ptr())
    eng.quotient_permutation_dev(cols.data_ptr(), 4 * N, sigma.data_ptr(), 4 * N, z.data_ptr(), 4 * N, nsets, 2, ncol, k + log_e, step,
                                 last_rot, lp[0].data_ptr(), lp[1].data_ptr(), lp[2].data_ptr(), F(beta), F(gamma), F(delta), F(x0),
                                 F(w_ext), F(y), d_h.data_ptr())
    eng.quotient_lookup_dev(lk_a.data_ptr(), 4 * N, lk_s.data_ptr(), lk_ap.data_ptr(), 4 * N, lk_sp.data_ptr(), 4 * N, lk_z.data_ptr(),
                            4 * N, nlk, k + log_e, step, lp[0].data_ptr(), lp[1].data_ptr(), lp[2].data_ptr(), F(beta), F(gamma), F(y),
                            d_h.data_ptr())
    eng.sync()
    got = sparse(d_h.view(1, N, 4))[0]
    # oracle on the sampled rows
    want = P.quotient_gate(S["cols"], S["sel"], step, y, h0, rows=rows, N=N)
    want = P.quotient_permutation(S["cols"], S["sigma"], S["z"], 2, step, last_rot, S["lp"][0], S["lp"][1], S["lp"][2], beta, gamma, delta,
                                  x0, w_ext, y, want, rows=rows, N=N)
    want = P.quotient_lookup(S["a"], S["s"][0], S["ap"], S["sp"], S["lz"], step, S["lp"][0], S["lp"][1], S["lp"][2], beta, gamma, y, want,
                             rows=rows, N=N)
    for i in rows:
        assert got[i] == want[i], ("row", i)


def test_permutation_product_sets_at_scale(eng, cref):
    """permutation::Argument::commit at the c2 column size (2^17 rows, 64 columns in 32 sets of 2, as bench.py calls it): the
    recurrence z_j[i+1] * den_j(i) == z_j[i] * num_j(i) on sampled rows of every set, z_0[0] = 1 and the chaining
    z_j[0] = z_(j-1)[usable_rows]."""
    import torch

    k, m, chunk = 17, 64, 2
    n = 1 << k
    usable = n - 11
    nsets = m // chunk
    gen = torch.Generator(device="cuda")
    gen.manual_seed(810)

    def rnd(count):
        x = torch.randint(-(1 << 63), (1 << 63) - 1, (count, n, 4), dtype=torch.int64, device="cuda", generator=gen)
        x[:, :, 3] &= 0x0FFFFFFFFFFFFFFF
        return x

    cols, sigma = rnd(m), rnd(m)
    d_z = torch.zeros((nsets, n, 4), dtype=torch.int64, device="cuda")
    rng = random.Random(811)
    beta, gamma = rng.randrange(2, P.FR_R), rng.randrange(2, P.FR_R)
    w, delta = P.fr_omega(k), pow(P.FR_GENERATOR, 1 << P.FR_S, P.FR_R)
    eng.permutation_product_sets_dev(cols.data_ptr(), 4 * n, sigma.data_ptr(), 4 * n, m, chunk, k, usable, _m(cref, w), _m(cref, beta),
                                     _m(cref, gamma), _m(cref, delta), d_z.data_ptr(), 4 * n)
    eng.sync()
    rows = sorted(set([0, 1, usable - 1, usable - 2] + [rng.randrange(usable) for _ in range(10)]))
    need = sorted(set(rows + [i + 1 for i in rows] + [0, usable]))
    idx = torch.tensor(need, device="cuda")

    def sparse(t):
        vals = cref.fr_mont_to_ints(t[:, idx, :].cpu().numpy().astype(np.uint64).reshape(-1, 4))
        return [dict(zip(need, vals[j * len(need):(j + 1) * len(need)])) for j in range(t.shape[0])]

    C, S, Z = sparse(cols), sparse(sigma), sparse(d_z)
    assert Z[0][0] == 1
    for j in range(nsets):
        if j:
            assert Z[j][0] == Z[j - 1][usable], ("chain", j)
        for i in rows:
            num = den = 1
            for c in range(j * chunk, (j + 1) * chunk):
                num = num * ((C[c][i] + beta * pow(delta, c, P.FR_R) % P.FR_R * pow(w, i, P.FR_R) + gamma) % P.FR_R) % P.FR_R
                den = den * ((C[c][i] + beta * S[c][i] + gamma) % P.FR_R) % P.FR_R
            assert Z[j][i + 1] * den % P.FR_R == Z[j][i] * num % P.FR_R, ("set", j, "row", i)
