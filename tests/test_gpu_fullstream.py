"""GPU parity at size, whole streams:
  (a) c5: pz_circuit_expand_dev at 48 limbs / lookup_bits 18 / k = 19 -- the first column (assign / square / refresh cells),
      a middle and the last column cell for cell vs the oracle, their commitments vs the C best_multiexp, and the gate
      identity + lookup range over the WHOLE 1.16e9-cell stream;
  (b) c2: gate identity q (a + b c - d) = 0 on every enabled window and the lookup range of every digit over the WHOLE
      GPU-written stream (3.97e8 + 1.1e7 cells), by the C checker (oracle/pz_oracle.c::ora_check_gates);
  (c) copy constraints of the MockProver analogue on GPU-written streams at the reference's own shapes;
  (d) c2: 64 of the proof's real K4 columns through iNTT -> coset extension (K2), sampled columns vs the C best_fft chain.
Everything goes through the C ABI; the oracle is only the checker."""
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
    e.bind_torch_stream()
    yield e
    e.close()


def _stream_to_host(torch, d_buf, ncols, n, rows, total):
    """column-cut device stream (ncols columns of n rows, `rows` used) -> dense (total, 4) uint64 host array; asserts that
    everything outside the stream (blinding rows, the ragged tail of the last column) is zero"""
    out = np.empty((ncols * rows, 4), dtype=np.uint64)
    v = d_buf.view(ncols, n, 4)
    step = 128
    for c0 in range(0, ncols, step):
        c1 = min(ncols, c0 + step)
        blk = v[c0:c1].cpu().numpy().view(np.uint64)
        assert not blk[:, rows:].any(), "rows above the usable ones stay untouched"
        out[c0 * rows:c1 * rows] = blk[:, :rows].reshape(-1, 4)
    assert not out[total:].any(), "cells past the end of the stream"
    return out[:total]


def _circuit_on_device(eng, cref, torch, enc_bits, k, lb, seed):
    Ln, L = enc_bits // 64, 2 * (enc_bits // 64)
    n = 1 << k
    rows = column_rows(k)
    nn, g, m, r = P.synth_paillier_inputs(enc_bits, seed)
    res = P.paillier_enc_native(nn, g, m, r)
    arr = lambda x: cref.int_to_limbs(x, Ln)
    cap = m.bit_length() + bin(m).count("1") + nn.bit_length() + bin(nn).count("1") + 1
    d_steps = torch.zeros((cap, 4, L), dtype=torch.int64, device="cuda")
    c, ng, nr = eng.paillier_encrypt_dev(Ln, arr(nn), arr(g), arr(m), arr(r), d_steps.data_ptr(), cap)
    ng, nr = int(ng[0]), int(nr[0])
    assert cref.limbs_to_int(c[0]) == res and ng + nr + 1 == cap
    adv_n, lk_n = eng.circuit_cells(0, Ln, 64, lb, ng, nr)
    ncol_a, ncol_l = -(-adv_n // rows), -(-lk_n // rows)
    d_adv = torch.zeros((ncol_a * n, 4), dtype=torch.int64, device="cuda")
    d_lk = torch.zeros((ncol_l * n, 4), dtype=torch.int64, device="cuda")
    d_mod = torch.from_numpy(cref.int_to_limbs(nn * nn, L).astype(np.int64)).cuda()
    inputs = np.concatenate([arr(nn), arr(g), arr(m), arr(r), cref.int_to_limbs(res, L)])
    eng.circuit_expand_dev(0, Ln, 64, lb, inputs, d_steps.data_ptr(), ng, nr, d_mod.data_ptr(), d_adv.data_ptr(), d_lk.data_ptr(), rows, n)
    eng.sync()
    return dict(nn=nn, g=g, m=m, r=r, res=res, ng=ng, nr=nr, adv_n=adv_n, lk_n=lk_n, ncol_a=ncol_a, ncol_l=ncol_l, d_adv=d_adv, d_lk=d_lk,
                rows=rows, n=n)


def _check_whole_stream(cref, torch, st, enc_bits, lb):
    mask, tot = P.gate_mask_circuit("encrypt", enc_bits, 64, lb, st["ng"], st["nr"])
    assert tot == st["adv_n"]
    cells = _stream_to_host(torch, st["d_adv"], st["ncol_a"], st["n"], st["rows"], st["adv_n"])
    bad, first = cref.check_gates(cells, mask)
    assert (bad, first) == (0, tot), "gate identity fails on %d windows of the GPU-written stream, first at cell %d" % (bad, first)
    # the assert_equal_fresh bit, the last cell of the stream, is 1 (Montgomery one): the circuit is satisfied
    assert cref.fr_mont_to_ints(cells[-1:]) == [1]
    n_gates = int(mask.sum())
    del cells, mask
    lk = _stream_to_host(torch, st["d_lk"], st["ncol_l"], st["n"], st["rows"], st["lk_n"])
    assert cref.check_range(lk, lb) == (0, st["lk_n"]), "a lookup digit outside [0, 2^lookup_bits)"
    return n_gates


def test_c2_whole_stream_gates_lookups_and_ntt_chain(eng, cref):
    """(b) + (d): BASELINE config c2 at full size (2048-bit n, k = 17, lookup_bits 16; bench.rs:120-173 shapes)"""
    import torch

    enc_bits, k, lb = 2048, 17, 16
    st = _circuit_on_device(eng, cref, torch, enc_bits, k, lb, 0x5043)
    assert (st["ncol_a"], st["ncol_l"]) == (3033, 84)
    n_gates = _check_whole_stream(cref, torch, st, enc_bits, lb)
    assert n_gates > 10 ** 8          # 1.0e8 enabled windows were checked, not a sample
    # (d) the proof's own columns through K2: Lagrange values -> coefficients (iNTT, 1/n) -> the extended coset (2^19)
    n, log_e, ncols = st["n"], 2, 64
    c0 = 1500                          # 64 columns from the middle of the g^m chain
    d_cols = st["d_adv"].view(st["ncol_a"], n, 4)[c0:c0 + ncols].clone()
    want_in = d_cols.cpu().numpy().view(np.uint64)
    d_ext = torch.zeros((ncols, n << log_e, 4), dtype=torch.int64, device="cuda")
    mont = lambda v: cref.fr_ints_to_mont([v % P.FR_R])[0]
    w_n, w_ext = P.fr_omega(k), P.fr_omega(k + log_e)
    w_inv, n_inv = pow(w_n, -1, P.FR_R), pow(n, -1, P.FR_R)
    gens = np.stack([mont(P.FR_GENERATOR * pow(w_ext, r, P.FR_R)) for r in range(1 << log_e)])
    eng.ntt_dev(d_cols.data_ptr(), ncols, 4 * n, mont(w_inv), k, None, mont(n_inv))
    eng.ntt_extend_dev(d_cols.data_ptr(), ncols, 4 * n, d_ext.data_ptr(), 4 * (n << log_e), k, log_e, mont(w_n), gens, None)
    eng.sync()
    got_coeff = d_cols.cpu().numpy().view(np.uint64)
    got_ext = d_ext.cpu().numpy().view(np.uint64)
    for j in (0, 31, 63):
        coeff = cref.fr_scale(cref.ntt_fr(want_in[j], mont(w_inv), k), mont(n_inv))
        assert np.array_equal(got_coeff[j], coeff), ("coefficients of column", c0 + j)
        ext_in = np.zeros((n << log_e, 4), dtype=np.uint64)
        ext_in[:n] = cref.fr_distribute_powers(coeff, mont(P.FR_GENERATOR))
        assert np.array_equal(got_ext[j], cref.ntt_fr(ext_in, mont(w_ext), k + log_e)), ("extended values of column", c0 + j)


def test_c5_whole_circuit_k19(eng, cref):
    """(a): BASELINE config c5's shape -- 3072-bit key (48 / 96 limbs), k = 19, lookup_bits 18 -- the whole driver on the
    device (K3 -> pz_circuit_expand_dev kind 0), not only its mul_mod blocks"""
    import torch

    enc_bits, k, lb = 3072, 19, 18
    st = _circuit_on_device(eng, cref, torch, enc_bits, k, lb, 0x5046)
    n, rows, adv_n, lk_n, ncol_a, ncol_l = st["n"], st["rows"], st["adv_n"], st["lk_n"], st["ncol_a"], st["ncol_l"]
    # sampled columns cell for cell: the first one holds assign_integer x4, square, refresh, load_zero and the first steps
    sample_a, sample_l = (0, ncol_a // 2, ncol_a - 1), (0, ncol_l - 1)
    win = lambda j, tot: (j * rows, min((j + 1) * rows, tot))
    tot_a, tot_l, cells_a, cells_l = P.encrypt_circuit_cells_windows(st["nn"], st["g"], st["m"], st["r"], st["res"], enc_bits, 64, lb,
                                                                     [win(j, adv_n) for j in sample_a], [win(j, lk_n) for j in sample_l])
    assert (tot_a, tot_l) == (adv_n, lk_n)
    # Lagrange SRS for the commitments of the sampled columns
    d_l = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    eng.srs_setup_g1_dev(k, cref.fr_ints_to_mont([0x1357911 * 0x2468ACE + 9])[0], cref.fr_ints_to_mont([P.fr_omega(k)])[0], 0, d_l.data_ptr())
    eng.sync()
    tb = eng.load_bases_dev(d_l.data_ptr(), n)
    bases = d_l.cpu().numpy().astype(np.uint64)
    d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    for d_buf, sample, cells, commit in ((st["d_adv"], sample_a, cells_a, True), (st["d_lk"], sample_l, cells_l, False)):
        for j, col in zip(sample, cells):
            col_m = cref.fr_ints_to_mont(col + [0] * (n - len(col)))
            assert np.array_equal(d_buf[j * n:(j + 1) * n].cpu().numpy().astype(np.uint64), col_m), ("cells of column", j)
            if commit:
                eng.msm_dev(tb, d_buf[j * n:(j + 1) * n].data_ptr(), 1, n, 4 * n, d_out.data_ptr())
                eng.sync()
                want = cref.g1_normalize(cref.msm_g1(col_m, bases))
                assert np.array_equal(eng.g1_normalize(d_out.cpu().numpy().astype(np.uint64))[0], want), ("commitment of column", j)
    assert cells_a[-1][-1] == 1
    tb.free()
    del d_l
    # and the whole 1.16e9-cell stream through the gate / lookup-range checker
    n_gates = _check_whole_stream(cref, torch, st, enc_bits, lb)
    assert n_gates > 3 * 10 ** 8


@pytest.mark.parametrize("kind,bits,W,lb", [("encrypt", 128, 64, 15),     # paillier.rs:113-182
                                            ("add", 264, 88, 15),         # paillier.rs:184-259
                                            ("encrypt", 128, 64, 13),     # bench.rs:137-179
                                            ("add", 128, 64, 13)])        # bench.rs:181-222
def test_copy_constraints_on_gpu_stream(eng, cref, kind, bits, W, lb):
    """(c): the MockProver analogue's copy constraints (oracle/pyref.py::expand_circuit_cells_wired) on the GPU-written
    stream at the reference's own shapes: assert_equal_fresh's operands, the n re-assigned by every mul_mod, extend_limbs'
    zero cells, every operand cell of the limb convolutions, range_check accumulators"""
    import torch

    rng = random.Random(bits * 977 + W + lb)
    Ln = bits // W
    L = 2 * Ln
    wn, wr = -(-Ln * W // 64), -(-L * W // 64)
    n, g, x, y = (rng.getrandbits(bits) for _ in range(4))
    n |= 1
    if kind == "encrypt":
        x &= (1 << 40) - 1
        res = P.paillier_enc_native(n, g, x, y)
        _, sg, sr, fin = P.encrypt_trace(n, g, x, y)
        ng, nr = len(sg), len(sr)
        arr = lambda v: cref.int_to_limbs(v, wn)
        cc, steps, ngd, nrd = eng.paillier_encrypt(Ln, arr(n), arr(g), arr(x), arr(y))
        assert (int(ngd[0]), int(nrd[0])) == (ng, nr) and cref.limbs_to_int(cc[0]) == res
        steps = steps[0, : ng + nr + 1]
    else:
        res = P.paillier_add_native(n, x, y)
        ng = nr = 0
        q, rem = eng.mul_mod(wr, cref.int_to_limbs(x, wr), cref.int_to_limbs(y, wr), cref.int_to_limbs(n * n, wr))
        steps = np.stack([cref.int_to_limbs(v, wr) for v in (x, y, cref.limbs_to_int(q), cref.limbs_to_int(rem))]).reshape(1, 4, wr)
    want_adv, pairs, eq = P.expand_circuit_cells_wired(kind, n, g, x, y, res, bits, W, lb)
    adv_n, lk_n = eng.circuit_cells(0 if kind == "encrypt" else 1, Ln, W, lb, ng, nr)
    assert adv_n == len(want_adv) and eq == 1
    d_steps = torch.from_numpy(np.ascontiguousarray(steps).astype(np.int64)).cuda()
    d_mod = torch.from_numpy(cref.int_to_limbs(n * n, wr).astype(np.int64)).cuda()
    d_adv = torch.zeros((adv_n, 4), dtype=torch.int64, device="cuda")
    d_lk = torch.zeros((lk_n, 4), dtype=torch.int64, device="cuda")
    inputs = np.concatenate([cref.int_to_limbs(v, wn) for v in (n, g, x, y)] + [cref.int_to_limbs(res, wr)])
    eng.circuit_expand_dev(0 if kind == "encrypt" else 1, Ln, W, lb, inputs, d_steps.data_ptr(), ng, nr, d_mod.data_ptr(), d_adv.data_ptr(),
                           d_lk.data_ptr())
    eng.sync()
    cells = d_adv.cpu().numpy().view(np.uint64)
    src = np.array([p_[0] for p_ in pairs], dtype=np.int64)
    dst = np.array([p_[1] for p_ in pairs], dtype=np.int64)
    neq = np.nonzero((cells[src] != cells[dst]).any(axis=1))[0]
    assert neq.size == 0, "copy constraint broken between cells %d and %d" % (src[neq[0]], dst[neq[0]])
    mask, tot = P.gate_mask_circuit(kind, bits, W, lb, ng, nr)
    assert cref.check_gates(cells, mask) == (0, tot)
    assert cref.check_range(d_lk.cpu().numpy().view(np.uint64), lb) == (0, lk_n)
