"""Files of the compiled prover (paillier_halo2_amd/host/prove_connected.cpp): the job it reads -- inputs + the circuit STRUCTURE keygen
takes (prover.CircuitStructure; the format is stated at the top of prove_connected.cpp) -- and the proof file it writes."""
from __future__ import annotations

import os
import subprocess
from typing import Dict, Sequence

import numpy as np

from . import consts

MAGIC = 0x435A50
HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(os.path.dirname(HERE), "tests", "cpp", "prove_connected")


def write_job(path: str, st, starts: Sequence[int], enc_bits: int, kind: int, ng: int, nr: int, nn: int, g: int, messages, s_toxic: int,
              seed: int = 1, proofs: int = 1, tile: int = 64):
    """st: prover.CircuitStructure; starts: the advice break points (n_adv + 1); messages: [(m, r)] integers of the SAME circuit shape"""
    Ln = enc_bits // 64
    n = 1 << st.k
    A, m = st.n_adv, st.m
    lim = lambda x, l: consts.int_to_limbs(x, l).astype("<u8")
    hdr = np.array([MAGIC, enc_bits, st.k, st.lookup_bits, st.max_rows, st.blinding_factors, A, st.n_lk, len(st.constants), kind, ng, nr, seed,
                    proofs, tile, len(messages)], dtype="<u8")
    host = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else a          # the structure may live on the device
    selectors, map_col, map_row = host(st.selectors), host(st.map_col), host(st.map_row)
    assert len(starts) == A + 1 and selectors.shape == (A, n) and map_col.shape == (m, n) == map_row.shape
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        f.write(lim(nn, Ln).tobytes())
        f.write(lim(g, Ln).tobytes())
        f.write(lim(nn * nn, 2 * Ln).tobytes())
        f.write(consts.fr_mont_limbs(s_toxic).astype("<u8").tobytes())
        f.write(np.asarray(starts, dtype="<u8").tobytes())
        for c in st.constants:
            f.write(lim(int(c) % consts.FR_R, 4).tobytes())
        for mm, rr in messages:
            f.write(lim(mm, Ln).tobytes())
            f.write(lim(rr, Ln).tobytes())
        for a in (np.ascontiguousarray(selectors, dtype=np.uint8), np.ascontiguousarray(map_col).view("<u4"), np.ascontiguousarray(map_row).view("<u4")):
            a.tofile(f)
            f.write(b"\0" * (-a.nbytes % 8))


FRESH_MAGIC = 0x465A50


def write_fresh_params(path: str, enc_bits: int, k: int, lookup_bits: int, inputs, s_toxic: int, minimum_rows: int = 20, blinding_factors: int = 6,
                       seed: int = 1, tile: int = 64):
    """the params file of `prove_connected --fresh`: inputs = [(n, g, m, r)] integers, one NEW key pair and message per step (the format is
    stated at the top of host/prove_connected.cpp).  No circuit structure: the binary generates it on the device per message."""
    Ln = enc_bits // 64
    lim = lambda x: consts.int_to_limbs(x, Ln).astype("<u8")
    hdr = np.array([FRESH_MAGIC, enc_bits, k, lookup_bits, minimum_rows, blinding_factors, seed, len(inputs), tile], dtype="<u8")
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        f.write(consts.fr_mont_limbs(s_toxic).astype("<u8").tobytes())
        for nn, g, m, r in inputs:
            for x in (nn, g, m, r):
                f.write(lim(x).tobytes())


def run_fresh(params_path: str, proof_path: str, timeout: float = 600.0, env=None) -> dict:
    import json

    if not os.path.exists(BINARY):
        raise RuntimeError("tests/cpp/prove_connected is not built (make -C tests/cpp prove_connected)")
    r = subprocess.run([BINARY, "--fresh", params_path, proof_path], capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0:
        raise RuntimeError("prove_connected --fresh exit %d: %s" % (r.returncode, r.stderr[-2000:]))
    line = json.loads(r.stdout.strip().splitlines()[-1])
    line["stderr_tail"] = r.stderr.strip().splitlines()[-8:]
    return line


def read_proofs(path: str) -> Dict[str, np.ndarray]:
    """-> {record name: uint64 array (count, words per item)}"""
    w = np.fromfile(path, dtype="<u8")
    out, p = {}, 0
    while p < len(w):
        ln = int(w[p]); p += 1
        nw = (ln + 7) // 8
        name = w[p:p + nw].tobytes()[:ln].decode(); p += nw
        _kind, count, per = (int(x) for x in w[p:p + 3]); p += 3
        out[name] = w[p:p + count * per].reshape(count, per).copy(); p += count * per
    assert p == len(w)
    return out


def run(job_path: str, proof_path: str, timeout: float = 600.0, env=None, allow_unsatisfied: bool = False) -> dict:
    """runs the binary (built by tests/cpp/Makefile); raises with its stderr if it fails.  -> its JSON line.  Exit code 1 = the proofs were
    written but a quotient's degree check failed (an unsatisfied witness): raised unless allow_unsatisfied"""
    import json

    if not os.path.exists(BINARY):
        raise RuntimeError("tests/cpp/prove_connected is not built (make -C tests/cpp prove_connected)")
    r = subprocess.run([BINARY, job_path, proof_path], capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0 and not (allow_unsatisfied and r.returncode == 1):
        raise RuntimeError("prove_connected exit %d: %s" % (r.returncode, r.stderr[-2000:]))
    return json.loads(r.stdout.strip().splitlines()[-1])
