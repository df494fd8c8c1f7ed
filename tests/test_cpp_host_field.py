"""the host field arithmetic of the C++ prover (paillier_halo2_amd/host/fr_host.hpp) against Python integers: domain generators, the
coset generator, DELTA, inverse, a mixed chain.  CPU only."""
import os
import subprocess

from paillier_halo2_amd import consts

HERE = os.path.dirname(os.path.abspath(__file__))
R = consts.FR_R


def test_fr_host_constants(tmp_path):
    exe = str(tmp_path / "fr_host_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(HERE, "cpp", "fr_host_check.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = {ln.split()[0]: int(ln.split()[1], 16) for ln in out.splitlines()}
    for k in range(1, 29):
        assert got["omega%d" % k] == consts.fr_omega(k), k
    assert got["zeta"] == pow(consts.FR_GENERATOR, (R - 1) // 3, R)
    assert got["delta"] == pow(consts.FR_GENERATOR, 1 << consts.FR_S, R)
    assert got["one"] == 1
    assert got["inv7"] == pow(7, -1, R)
    assert got["neg7"] == R - 7
    assert got["pow7_1000003"] == pow(7, 1000003, R)
    x, acc = 0x123456789ABCDEF, 1
    for i in range(100):
        acc = (acc * x - i) % R
        x = x * x % R
    assert got["chain"] == acc


def test_transcript_is_blake2b_as_hashlib_has_it(tmp_path):
    """host/blake2b.hpp against RFC 7693's vector and hashlib (personalised, every block boundary, uneven pieces); host/transcript.hpp's
    challenges against prover.HashTranscript's and the oracle's replay on the same items"""
    import hashlib

    import numpy as np

    from oracle import verifier as V
    from paillier_halo2_amd import prover

    exe = str(tmp_path / "transcript_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(HERE, "cpp", "transcript_check.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = {ln.split()[0]: ln.split()[1] for ln in out.splitlines()}
    # RFC 7693 appendix A
    assert got["abc"] == ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d17d87c5392aab792dc252d5de4533cc95"
                          "18d38aa8dbf1925ab92386edd4009923")
    assert got["empty"] == hashlib.blake2b(b"").hexdigest()
    for ln in (1, 127, 128, 129, 255, 256, 257, 1000, 4099):
        msg = bytes((i * 131 + 7) & 0xFF for i in range(ln))
        assert got["pers%d" % ln] == hashlib.blake2b(msg, person=b"Halo2-Transcript").hexdigest(), ln
    M64 = (1 << 64) - 1
    pts = np.array([(0x9E3779B97F4A7C15 * (i + 1)) & M64 for i in range(24)], dtype=np.uint64).reshape(3, 8)
    sc = np.array([(0xBF58476D1CE4E5B9 * (i + 3)) & M64 for i in range(8)], dtype=np.uint64).reshape(2, 4)
    tr = prover.HashTranscript((5).to_bytes(8, "little"))
    tr.absorb_affine(pts)
    a = tr.squeeze("a")
    tr.absorb_scalars(sc)
    b, c = tr.squeeze("b"), tr.squeeze("c")
    for name, v in (("a", a), ("b", b), ("c", c)):
        assert int.from_bytes(bytes.fromhex(got["ch_" + name]), "little") == v, name
        assert 0 <= v < R
    # by hand: the definition in one place
    h = hashlib.blake2b((5).to_bytes(8, "little"), digest_size=64, person=b"Halo2-Transcript")
    for row in pts:
        h.update(b"\x01" + row.tobytes())
    h.update(b"\x00")
    assert int.from_bytes(h.digest(), "little") % R == a


def test_verifier_replay_equals_the_provers_transcript():
    """oracle/verifier.py::replay_challenges (item by item) == prover.HashTranscript (arrays) over a proof-shaped set of families, and
    every item matters: one flipped bit anywhere moves every later challenge"""
    import numpy as np

    from oracle import verifier as V
    from paillier_halo2_amd import prover

    rng = np.random.default_rng(11)
    A, Lk, m, S, F = 5, 2, 8, 4, 7
    u = lambda *shape: rng.integers(0, 1 << 63, size=shape, dtype=np.uint64)
    com = {"advice": u(A, 8), "lookup_advice": u(Lk, 8), "perm_inputs": u(Lk, 8), "perm_tables": u(Lk, 8), "perm_z": u(S, 8), "lookup_z": u(Lk, 8),
           "random": u(1, 8), "h": u(3, 8), "w1": u(1, 8), "w2": u(1, 8)}
    ev = {"advice": u(A, 4, 4), "lookup_advice": u(Lk, 1, 4), "constants": u(1, 1, 4), "fixed": u(F, 1, 4), "sigma": u(m, 1, 4), "perm_z": u(S, 3, 4),
          "lookup_z": u(Lk, 2, 4), "perm_inputs": u(Lk, 2, 4), "perm_tables": u(Lk, 1, 4), "random": u(1, 1, 4), "h": u(1, 1, 4)}

    def prove_side(com, ev):
        tr = prover.HashTranscript(b"seed")
        tr.absorb_affine(np.concatenate([com["advice"], com["lookup_advice"]]))
        tr.squeeze("theta")
        tr.absorb_affine(com["perm_inputs"], com["perm_tables"])
        tr.squeeze("beta"), tr.squeeze("gamma")
        tr.absorb_affine(com["perm_z"], com["lookup_z"], com["random"])
        tr.squeeze("y")
        tr.absorb_affine(com["h"])
        tr.squeeze("x")
        tr.absorb_scalars(ev["advice"], np.concatenate([ev["lookup_advice"], ev["constants"]]), *[ev[f] for f in V.EVAL_FAMILIES[2:]])
        tr.squeeze("sh_y"), tr.squeeze("sh_v")
        tr.absorb_affine(com["w1"])
        tr.squeeze("sh_u")
        return tr.drawn

    base = V.replay_challenges(b"seed", com, ev)
    assert base == prove_side(com, ev) and len(base) == 8
    assert V.replay_challenges(b"seeD", com, ev)["theta"] != base["theta"]
    order = ["theta", "beta", "gamma", "y", "x", "sh_y", "sh_v", "sh_u"]
    first_moved = {"advice": "theta", "perm_tables": "beta", "random": "y", "h": "x", "w1": "sh_u"}
    for fam, first in first_moved.items():
        c2 = {k_: v_.copy() for k_, v_ in com.items()}
        c2[fam][-1, 3] ^= np.uint64(1)
        got = V.replay_challenges(b"seed", c2, ev)
        i0 = order.index(first)
        assert all(got[nm] == base[nm] for nm in order[:i0]) and all(got[nm] != base[nm] for nm in order[i0:]), fam
    for fam in ("advice", "constants", "sigma", "random"):
        e2 = {k_: v_.copy() for k_, v_ in ev.items()}
        e2[fam][0, 0, 0] ^= np.uint64(1)
        got = V.replay_challenges(b"seed", com, e2)
        assert got["x"] == base["x"] and got["sh_y"] != base["sh_y"] and got["sh_u"] != base["sh_u"], fam
    e2 = {k_: v_.copy() for k_, v_ in ev.items()}
    e2["h"][0, 0, 0] ^= np.uint64(1)            # h's value at x is not in the transcript: the verifier computes it itself
    assert V.replay_challenges(b"seed", com, e2) == base
