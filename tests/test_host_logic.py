"""CPU: host-side logic of the package (no GPU): circuit-shape arithmetic, constants, window partition."""
import os
import random

import numpy as np
import pytest

from oracle import pyref as P
from paillier_halo2_amd import consts, layout
from paillier_halo2_amd.dist import window_range


def test_layout_counts_match_the_cell_stream():
    rng = random.Random(21)
    for L, lb in ((2, 15), (4, 16), (4, 14), (4, 8), (6, 13), (64, 16), (64, 14), (96, 18)):
        bits = 64 * L
        n = rng.getrandbits(bits) | (1 << (bits - 1))
        a, b = rng.randrange(n), rng.randrange(n)
        q, r = divmod(a * b, n)
        adv, lk = P.expand_mul_mod_cells(a, b, q, r, n, L, lb)
        sc = layout.mul_mod_cells(L, 64, lb)
        assert (sc.advice, sc.lookup) == (len(adv), len(lk)), (L, lb)
        gates, end = P.gate_offsets_mul_mod(L, lb)
        assert end == sc.advice
        # segment offsets are increasing and inside the block
        offs = [sc.seg[k] for k in ("assign", "mul_ab", "mul_qn", "add_r", "eq", "lt", "end")]
        assert offs == sorted(offs) and offs[-1] == sc.advice


def test_proof_shape_c2_c3_c5():
    c2 = layout.encrypt_proof_shape(2048, 17, 6145)
    assert c2.limbs == 64 and c2.lookup_bits == 16 and c2.ext_k == 19
    assert 2800 <= c2.advice_cols <= 3200          # "about 2.8k advice columns at k=17"
    assert c2.perm_cols == -(-(c2.advice_cols + c2.lookup_cols + 1) // 2)
    assert c2.polys == c2.advice_cols + 4 * c2.lookup_cols + c2.perm_cols
    c3 = layout.encrypt_proof_shape(2048, 15, 1, lookup_bits=14)
    assert c3.advice_cols <= 8
    c5 = layout.encrypt_proof_shape(3072, 19, 9217)
    assert c5.limbs == 96 and 2000 <= c5.advice_cols <= 2600


def test_consts_against_oracle():
    assert consts.FR_R == P.FR_R and consts.FQ_P == P.FQ_P
    assert consts.FR_ROOT_OF_UNITY == P.FR_ROOT_OF_UNITY
    for k in (0, 1, 5, 17, 19, 28):
        assert consts.fr_omega(k) == P.fr_omega(k)
        assert pow(consts.fr_omega(k), 1 << k, consts.FR_R) == 1
    x = 0x1234567890ABCDEF1234567890ABCDEF
    limbs = consts.fr_mont_limbs(x)
    assert consts.limbs_to_int(limbs) == P.to_mont(x, P.FR_R)
    assert consts.limbs_to_int(consts.int_to_limbs(x, 4)) == x


def test_window_ranges_cover_exactly():
    for W in range(1, 40):
        for world in (1, 2, 3, 4, 5, 8, 16):
            seen = []
            for r in range(world):
                lo, hi = window_range(W, r, world)
                seen += list(range(lo, hi))
            assert seen == list(range(W))


def test_params_kzg_file_roundtrip_and_malformed(tmp_path):
    """ParamsKZG RawBytes file (params/kzg_bn254_{k}.srs): write -> read is the identity; wrong sizes / k are refused"""
    from paillier_halo2_amd import srs

    k = 4
    rng = np.random.default_rng(5)
    g = rng.integers(0, 1 << 63, size=(1 << k, 8), dtype=np.uint64)
    gl = rng.integers(0, 1 << 63, size=(1 << k, 8), dtype=np.uint64)
    g2, sg2 = bytes(range(128)), bytes(reversed(range(128)))
    p = str(tmp_path / "kzg_bn254_4.srs")
    srs.write_params_kzg(p, k, g, gl, g2, sg2)
    assert os.path.getsize(p) == srs.file_size(k) == 4 + 2 * 16 * 64 + 256
    prm = srs.read_params_kzg(p, expect_k=k)
    assert prm.k == k and prm.n == 16 and prm.g2 == g2 and prm.s_g2 == sg2
    assert np.array_equal(prm.g, g) and np.array_equal(prm.g_lagrange, gl)
    raw = open(p, "rb").read()
    assert raw[:4] == (4).to_bytes(4, "little") and raw[4:12] == int(g[0, 0]).to_bytes(8, "little")
    with pytest.raises(ValueError):
        srs.read_params_kzg(p, expect_k=5)
    open(p, "wb").write(raw[:-1])
    with pytest.raises(ValueError):
        srs.read_params_kzg(p)
    open(p, "wb").write(raw[:3])
    with pytest.raises(ValueError):
        srs.read_params_kzg(p)
    with pytest.raises(ValueError):
        srs.write_params_kzg(p, k, g[:3], gl)


def test_host_field_helpers_of_the_row_kernels():
    """CPU: the host-side Fr arithmetic that turns challenges into the SGPR-resident limbs of the quotient kernels
    (pz_quotient.hip: host_fr_mul / host_fr_pow / host_fr_shl, exported for tests as pzx_*) against Python integers"""
    import ctypes
    import random

    import numpy as np

    import paillier_halo2_amd as pz
    from paillier_halo2_amd import consts

    pz.build()
    L = ctypes.CDLL(pz.SO_PATH)
    assert L.pzx_host_selftest() == 0
    r, R = consts.FR_R, 1 << 256
    rng = random.Random(41)
    u64x4 = lambda x: np.array([(x >> (64 * i)) & (2 ** 64 - 1) for i in range(4)], dtype=np.uint64)
    val = lambda a: sum(int(v) << (64 * i) for i, v in enumerate(a))
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for _ in range(50):
        x, y = rng.randrange(r), rng.randrange(r)
        a, b, out = u64x4(x * R % r), u64x4(y * R % r), np.zeros(4, dtype=np.uint64)
        L.pzx_host_fr_mul(P(a), P(b), P(out))
        assert val(out) == x * y * R % r                       # Montgomery product, canonical
        e = rng.randrange(0, 5000)
        L.pzx_host_fr_pow(P(a), ctypes.c_uint(e), P(out))
        assert val(out) == pow(x, e, r) * R % r
        k = rng.choice([0, 1, 5, 10, 29, 64])
        limbs = np.zeros(9, dtype=np.uint32)
        L.pzx_host_fr_shl(P(a), ctypes.c_uint(k), P(limbs))
        assert sum(int(v) << (29 * i) for i, v in enumerate(limbs)) == (x * R << k) % r
        assert all(int(v) < (1 << 29) for v in limbs[:8])
    # edge values: 0, 1, r - 1
    for x in (0, 1, r - 1):
        a, out = u64x4(x * R % r), np.zeros(4, dtype=np.uint64)
        L.pzx_host_fr_mul(P(a), P(a), P(out))
        assert val(out) == x * x * R % r
