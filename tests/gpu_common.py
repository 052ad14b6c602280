"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np

import pymodel as pm


def rand_fe_array(rng, p, n, edge=True):
    vals = [rng.below(p) for _ in range(n)]
    if edge and n >= 4:
        vals[0], vals[1], vals[2], vals[3] = 0, 1, p - 1, (1 << 256) % p
    return np.array([[(v >> (64 * i)) & (2 ** 64 - 1) for i in range(4)] for v in vals], dtype=np.uint64)


def np_scalars(seed, n, kind="uniform"):
    """n x 32 B scalars, standard form, deterministic. kinds mirror SURVEY 8(d)."""
    rs = np.random.RandomState(seed & 0x7FFFFFFF)
    if kind == "uniform":
        s = rs.randint(0, 256, size=(n, 32), dtype=np.uint8)
        s[:, 31] &= 0x1F  # < 2^253 < r : uniform enough for a schedule test, always canonical
    elif kind == "full256":
        s = rs.randint(0, 256, size=(n, 32), dtype=np.uint8)  # scalars are not range-checked
    elif kind == "ones":
        s = np.zeros((n, 32), dtype=np.uint8)
        s[:, 0] = 1
    elif kind == "zeros":
        s = np.zeros((n, 32), dtype=np.uint8)
    elif kind == "witness":  # 90 % bits, 8 % bytes, 2 % full width
        s = np.zeros((n, 32), dtype=np.uint8)
        u = rs.rand(n)
        bits = u < 0.90
        byts = (u >= 0.90) & (u < 0.98)
        full = u >= 0.98
        s[bits, 0] = rs.randint(0, 2, size=bits.sum())
        s[byts, 0] = rs.randint(0, 256, size=byts.sum())
        f = rs.randint(0, 256, size=(full.sum(), 32), dtype=np.uint8)
        f[:, 31] &= 0x1F
        s[full] = f
    elif kind == "topwindow":
        s = np.zeros((n, 32), dtype=np.uint8)
        s[:, 30:32] = rs.randint(0, 256, size=(n, 2), dtype=np.uint8)
        s[:, 31] &= 0x1F
    elif kind == "same":
        one = rs.randint(0, 256, size=(1, 32), dtype=np.uint8)
        one[:, 31] &= 0x1F
        s = np.repeat(one, n, axis=0)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(s)
