"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

Closed-form, library-independent tensor generator used by the golden-fixture
script, the oracle and the parity tests so that 136 M parameters never have to
be stored: every element is a pure function of (tensor name, flat index).

    seed(name)   = FNV-1a 64 of the UTF-8 name
    u64(i)       = splitmix64(seed + i)
    uniform(i)   = (u64(i) >> 40) / 2**24                 in [0, 1)
    normal(i)    = sqrt(3) * (U0 + U1 + U2 + U3 - 2)      (Irwin-Hall, var 1)
                   with Uj = uniform(4*i + j)

All arithmetic is uint64 wrap-around / exact float32 conversions, so numpy,
C or a GPU kernel reproduce it bit for bit.
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _u01(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = _splitmix64(idx.astype(np.uint64) + np.uint64(seed))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))


def uniform(name: str, shape, lo=0.0, hi=1.0, chunk=1 << 24) -> np.ndarray:
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    seed = fnv1a64(name)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        out[s:e] = _u01(seed, np.arange(s, e, dtype=np.uint64))
    out = out * np.float32(hi - lo) + np.float32(lo)
    return out.reshape(shape)


def normal(name: str, shape, std=1.0, mean=0.0, chunk=1 << 22) -> np.ndarray:
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    seed = fnv1a64(name)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        i4 = np.arange(s, e, dtype=np.uint64) * np.uint64(4)
        acc = _u01(seed, i4)
        for j in (1, 2, 3):
            acc = acc + _u01(seed, i4 + np.uint64(j))
        out[s:e] = (acc - np.float32(2.0)) * np.float32(np.sqrt(3.0))
    out = out * np.float32(std) + np.float32(mean)
    return out.reshape(shape)


def randint(name: str, shape, lo, hi) -> np.ndarray:
    """integers in [lo, hi)"""
    u = uniform(name, shape)
    return (lo + np.floor(u.astype(np.float64) * (hi - lo))).astype(np.int64)


def checksum(a: np.ndarray) -> str:
    """order-sensitive 64-bit checksum of the raw bytes (FNV-1a over 8-byte words)."""
    b = np.ascontiguousarray(a).view(np.uint8).ravel()
    pad = (-len(b)) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    w = b.view(np.uint64)
    # vectorised polynomial hash: sum_i w_i * P^(i mod 2^k) folded, cheap + order sensitive
    with np.errstate(over="ignore"):
        idx = np.arange(len(w), dtype=np.uint64)
        mix = _splitmix64(idx)
        return "%016x" % int(np.bitwise_xor.reduce(_splitmix64(w ^ mix)) if len(w) else 0)
