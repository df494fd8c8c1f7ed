"""BN254 constants and tiny Python-int helpers the host side needs to PARAMETRISE kernel calls
(domain generators, 1/n, Montgomery encodings of a handful of scalars).  Not a compute path: bulk
arithmetic lives in libpz_hip.so.  Values as in SURVEY.md section 8c (checked in tests/test_oracle.py)."""
import numpy as np

FQ_P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
FR_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FR_S = 28
FR_GENERATOR = 7  # multiplicative generator; also the coset shift halo2 uses (ZETA-independent part)
FR_ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
MONT_R = 1 << 256


def fr_omega(log_n: int) -> int:
    """generator of the 2^log_n domain: ROOT_OF_UNITY^(2^(28-log_n))"""
    assert 0 <= log_n <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - log_n), FR_R)


def fr_mont_limbs(x: int) -> np.ndarray:
    """canonical integer -> 4 x u64 Montgomery limbs (one scalar; for kernel parameters only)"""
    v = (x % FR_R) * MONT_R % FR_R
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def int_to_limbs(x: int, n: int) -> np.ndarray:
    assert 0 <= x < 1 << (64 * n)
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def limbs_to_int(a) -> int:
    acc = 0
    for i, l in enumerate(np.asarray(a).reshape(-1).tolist()):
        acc |= int(l) << (64 * i)
    return acc
