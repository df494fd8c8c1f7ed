"""GPU: the C++ host mirror of the reference's chip interface (paillier_halo2_amd/host/) driven by
tests/cpp/test_paillier.cpp, which restates the reference's four #[test] functions with seeded inputs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_mirror_of_reference_tests():
    exe = os.path.join(ROOT, "tests", "cpp", "test_paillier")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(p.stdout[-3000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "ALL OK" in p.stdout


def test_cpp_mirror_builds():
    """CPU: the header-only host mirror compiles and links against the C ABI"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-B"], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "test_paillier"))
