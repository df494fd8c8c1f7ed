"""Shared helpers for the test-suite: golden fixture loading and representation conversion."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def column_rows(k):
    """rows a column of the circuit is filled to -- the row budget's max_rows (paillier_halo2_amd/layout.py RowBudget: 2^k - 9)"""
    from paillier_halo2_amd import layout

    return layout.row_budget(k).max_rows


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def H(s):
    return int(s, 16)


def steps_digest_arr(steps, L):
    """steps: ndarray (n_steps, 4, L) u64 -> sha256 hex, same byte order as make_golden.steps_digest"""
    return hashlib.sha256(np.ascontiguousarray(steps, dtype="<u8").tobytes()).hexdigest()


def steps_digest_ints(steps, L):
    h = hashlib.sha256()
    for st in steps:
        for v in st:
            h.update(int(v).to_bytes(8 * L, "little"))
    return h.hexdigest()


# ---- bulk helpers for the at-size GPU tests (2^19 .. 2^22 scalars: Python-int loops would take minutes) ----
def ints_to_u64x4(xs):
    """list of ints below 2^256 -> (n, 4) uint64 little-endian limbs"""
    buf = b"".join(int(x).to_bytes(32, "little") for x in xs)
    return np.frombuffer(buf, dtype="<u8").reshape(-1, 4).copy()


def canon_rand_scalars(n, seed):
    """(n, 4) uint64: uniformly random canonical integers below 2^252 (< r)"""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    return a


def witness_like_canon(n, seed):
    """SURVEY.md section 8d mix (ii): 60 % below 2^16, 30 % below 2^64, 10 % below 2^135 -- canonical (n, 4) uint64"""
    rng = np.random.default_rng(seed)
    a = canon_rand_scalars(n, seed + 1)
    u = rng.random(n)
    a[:, 3] = 0
    big = u >= 0.9
    a[:, 2] = np.where(big, a[:, 2] & np.uint64(0x7F), np.uint64(0))
    a[:, 1] = np.where(big, a[:, 1], np.uint64(0))
    a[:, 0] = np.where(u < 0.6, a[:, 0] & np.uint64(0xFFFF), a[:, 0])
    return a


def walk_dlog_sum(sc, s, t, r=0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001):
    """sum_i c_i * (s + i t) mod r for canonical scalars sc (n, 4) uint64, n <= 2^22: exact, by 16-bit pieces
    (every partial sum stays below 2^64)"""
    n = sc.shape[0]
    assert n <= 1 << 22
    q = np.ascontiguousarray(sc).view(np.uint16).reshape(n, 16).astype(np.uint64)
    idx = np.arange(n, dtype=np.uint64)[:, None]
    s0 = q.sum(axis=0)            # < 2^38
    s1 = (q * idx).sum(axis=0)    # < 2^60
    c_sum = sum(int(v) << (16 * j) for j, v in enumerate(s0.tolist()))
    ic_sum = sum(int(v) << (16 * j) for j, v in enumerate(s1.tolist()))
    return (s * c_sum + t * ic_sum) % r


def challenges_replay(pr, ch):
    """the verifier's side of Fiat-Shamir (oracle/verifier.py::replay_challenges): True iff every challenge the prover's HashTranscript drew is
    the one re-derived from the proof's own commitments and evaluations"""
    from oracle import verifier as V

    d = V.replay_challenges(ch.transcript_seed, pr.commitments, pr.evals)
    return len(d) == 8 and all(getattr(ch, nm) == v for nm, v in d.items())
