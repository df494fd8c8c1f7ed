"""Shared helpers for the test-suite: golden fixture loading and representation conversion."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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
