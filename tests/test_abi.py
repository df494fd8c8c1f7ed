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
    assert L.pz_abi_version() == int(re.search(r"#define PZ_ABI_VERSION (\d+)", hdr).group(1)) == 7
    assert not [n for n in names if "ubench" in n or n == "pz_fq_mul29"]
    assert L.pz_strerror(0) == b"ok" and L.pz_strerror(-6).startswith(b"quotient")


def test_integration_doc_binds_every_entry_point():
    """INTEGRATION.md's `extern "C"` block (the reference-side binding) names every function include/pz.h declares"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in declared_functions() if ("pub fn %s(" % n) not in text]
    assert not missing, missing


def _c_prototypes():
    """include/pz.h -> {name: (return class, [parameter classes])}; classes: ptr, int, u32, u64, usize, f64"""
    src = open(os.path.join(ROOT, "include", "pz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    src = re.sub(r"enum\s+pz_status\s*\{.*?\};", "", src, flags=re.S)
    src = src.replace('extern "C" {', "")
    out = {}
    for stmt in src.split(";"):
        st = stmt.strip()
        m = re.search(r"((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*(pz_[a-z0-9_]+)\s*\((.*)\)\s*$", st, flags=re.S)
        if not m or st.startswith("typedef"):
            continue
        ret, name, params = m.group(1), m.group(2), m.group(3)

        def cls(decl):
            d = " ".join(decl.split())
            if "*" in d or "[" in d:
                return "ptr"
            base = re.sub(r"\bconst\b", "", d).split()
            ty = base[0] if len(base) == 1 else " ".join(base[:-1])
            return {"int": "int", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "double": "f64", "void": "void"}[ty]

        ps = [] if params.strip() in ("", "void") else [cls(x) for x in params.split(",")]
        out[name] = (cls(ret + " x") if "*" in ret else cls(ret + " x"), ps)
    return out


def _rust_bindings(path="INTEGRATION.md"):
    """the extern "C" declarations of INTEGRATION.md (or of the pz-sys crate source) -> the same shape"""
    text = open(os.path.join(ROOT, path)).read()
    out = {}
    for m in re.finditer(r"pub fn (pz_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", text, flags=re.S):
        name, params, ret = m.group(1), m.group(2), (m.group(3) or "void").strip()

        def cls(ty):
            ty = " ".join(ty.split())
            if ty.startswith("*") or ty.startswith("Option<") or ty.startswith("&"):
                return "ptr"
            return {"c_int": "int", "i32": "int", "u32": "u32", "u64": "u64", "usize": "usize", "f64": "f64", "void": "void"}[ty]

        ps = [cls(x.split(":", 1)[1]) for x in params.split(",") if x.strip()]
        out.setdefault(name, (cls(ret), ps))
    return out


def test_bindings_match_the_header_parameter_by_parameter():
    """the reference-side binding and the ctypes table against include/pz.h: parameter COUNT and the pointer / c_int /
    u32 / u64 / usize / f64 class of every parameter and of the return value, for every entry point (names alone were checked before)"""
    import ctypes as C

    c = _c_prototypes()
    names = declared_functions()
    assert sorted(c) == names, sorted(set(names) ^ set(c))
    rust = _rust_bindings()
    mism = []
    for n in names:
        if n not in rust:
            mism.append((n, "not bound"))
        elif rust[n] != c[n]:
            mism.append((n, c[n], rust[n]))
    assert not mism, mism
    # the crate source (rust/pz-sys/src/lib.rs) is the same binding, file for file
    crate = _rust_bindings(os.path.join("rust", "pz-sys", "src", "lib.rs"))
    assert crate == rust and len(crate) == len(names)
    # the ctypes signatures
    def ct(t):
        if t is None:
            return "void"
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        return {C.c_int: "int", C.c_uint32: "u32", C.c_uint64: "u64", C.c_size_t: "usize", C.c_double: "f64"}[t]

    for n in names:
        res, args = _lib.SIGNATURES[n]
        got = (ct(res), [ct(a) for a in args])
        want = c[n]
        if C.sizeof(C.c_size_t) == C.sizeof(C.c_uint64):   # ctypes aliases c_size_t and c_uint64 on LP64
            norm = lambda sig: (sig[0], ["u64" if x == "usize" else x for x in sig[1]])
            got, want = norm(got), norm(want)
        assert got == want, (n, want, got)


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
