"""CPU: the C-ABI library loads and exports every symbol include/pz.h declares; without a GPU the
product path fails loudly (no fallback)."""
import os
import re

import pytest

import paillier_halo2_amd as pz
from paillier_halo2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "pz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pz_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    pz.build()
    L = pz.lib()
    names = declared_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/pz.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    # the version is explicit (not a count of exports) and matches the header; the measurement probes are not in this ABI
    hdr = open(os.path.join(ROOT, "include", "pz.h")).read()
    assert L.pz_abi_version() == int(re.search(r"#define PZ_ABI_VERSION (\d+)", hdr).group(1)) == 5
    assert not [n for n in names if "ubench" in n or n == "pz_fq_mul29"]
    assert L.pz_strerror(0) == b"ok" and L.pz_strerror(-6).startswith(b"quotient")


def test_integration_doc_binds_every_entry_point():
    """INTEGRATION.md's `extern "C"` block (the reference-side binding) names every function include/pz.h declares"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in declared_functions() if ("pub fn %s(" % n) not in text]
    assert not missing, missing


def test_product_never_imports_oracle():
    """the product path must not import, link, dlopen or execute anything under oracle/"""
    pkg = os.path.join(ROOT, "paillier_halo2_amd")
    bad = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|libpz_oracle|pz_oracle\.c|ora_[a-z_]+\(", re.M)
    for base in (pkg, os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".hip", ".cuh", ".h", ".hpp", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert not bad.search(text), os.path.join(dirpath, f)


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pz.PzError):
        pz.Engine(0)
