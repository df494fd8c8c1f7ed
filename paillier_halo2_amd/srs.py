"""ParamsKZG file format (the `./params/kzg_bn254_{k}.srs` files halo2-lib's `gen_srs` caches; the reference
git-ignores that directory, `/root/reference/.gitignore:4`, and reaches `gen_srs` through `bench.rs:161-171`).

Layout restated from halo2-axiom `ParamsKZG::write` == `write_custom(.., SerdeFormat::RawBytes)` (dependency
behaviour, SURVEY tag [D]; the reference ships no sample file, so the format is unpinned here):

    u32 LE   k
    2^k x 64 B   g[i]           G1Affine, RawBytes: x then y, each the 4 x u64 LE Montgomery limbs
    2^k x 64 B   g_lagrange[i]  same
    128 B        g2             G2Affine RawBytes (x.c0, x.c1, y.c0, y.c1)
    128 B        s_g2

RawBytes is exactly the C ABI's in-memory point layout (include/pz.h), so the point sections are handed to
`pz_bases_load_g1` without conversion; `read` memory-maps them (a k = 26 file is 8.6 GB).  The reader checks
sizes; on-curve validation of what it returns is `Engine.g1_check` (on the device), the analogue of the
`is_on_curve` assertion inside halo2curves' `read_raw`.  G2 elements are opaque here (verifier side).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass

import numpy as np

G2_BYTES = 128


@dataclass
class ParamsKZG:
    k: int
    g: np.ndarray            # (2^k, 8) uint64
    g_lagrange: np.ndarray   # (2^k, 8) uint64
    g2: bytes
    s_g2: bytes

    @property
    def n(self) -> int:
        return 1 << self.k


def file_size(k: int) -> int:
    return 4 + 2 * (64 << k) + 2 * G2_BYTES


def write_params_kzg(path: str, k: int, g, g_lagrange, g2: bytes = bytes(G2_BYTES), s_g2: bytes = bytes(G2_BYTES)) -> None:
    n = 1 << k
    g = np.ascontiguousarray(g, dtype=np.uint64).reshape(-1, 8)
    gl = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(-1, 8)
    if g.shape[0] != n or gl.shape[0] != n:
        raise ValueError("g and g_lagrange must hold 2^k points")
    if len(g2) != G2_BYTES or len(s_g2) != G2_BYTES:
        raise ValueError("g2 / s_g2 are 128 raw bytes each")
    with open(path, "wb") as f:
        f.write(struct.pack("<I", k))
        f.write(g.astype("<u8", copy=False).tobytes())
        f.write(gl.astype("<u8", copy=False).tobytes())
        f.write(g2)
        f.write(s_g2)


def read_params_kzg(path: str, expect_k: int | None = None) -> ParamsKZG:
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(4)
        if len(head) != 4:
            raise ValueError("truncated ParamsKZG file (no header)")
        (k,) = struct.unpack("<I", head)
        if k > 28:
            raise ValueError("implausible k = %d in ParamsKZG header" % k)
        if expect_k is not None and k != expect_k:
            raise ValueError("ParamsKZG file is for k = %d, expected %d" % (k, expect_k))
        if size != file_size(k):
            raise ValueError("ParamsKZG file for k = %d must be %d bytes, found %d (not the RawBytes format?)" % (k, file_size(k), size))
        n = 1 << k
        f.seek(4 + 2 * 64 * n)
        g2 = f.read(G2_BYTES)
        s_g2 = f.read(G2_BYTES)
    g = np.memmap(path, dtype="<u8", mode="r", offset=4, shape=(n, 8))
    gl = np.memmap(path, dtype="<u8", mode="r", offset=4 + 64 * n, shape=(n, 8))
    return ParamsKZG(k, g, gl, g2, s_g2)
