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
