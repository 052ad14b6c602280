"""Pure-Python big-integer model of BN254 (alt_bn128) used ONLY by tests.

Independent of both the C oracle (oracle/) and the HIP path: plain Python
ints, textbook affine formulas, naive O(n^2) DFT.  Small cases only.

Curve/field constants: /root/reference/rust-rapidsnark/rapidsnark/src/
fq_raw_generic.cpp:6, fr_raw_generic.cpp:5, alt_bn128.hpp:41-54.
"""
import struct

Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
MONT = 1 << 256

G1 = (1, 2)
G2 = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


def to_mont(x, p):
    return (x * MONT) % p


def from_mont(x, p):
    return (x * pow(MONT, -1, p)) % p


def mont_mul(a, b, p):
    return (a * b * pow(MONT, -1, p)) % p


def limbs(x):
    """256-bit int -> 32 little-endian bytes."""
    return x.to_bytes(32, "little")


def unlimbs(b):
    return int.from_bytes(bytes(b), "little")


# ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1)
def f2_add(x, y):
    return ((x[0] + y[0]) % Q, (x[1] + y[1]) % Q)


def f2_sub(x, y):
    return ((x[0] - y[0]) % Q, (x[1] - y[1]) % Q)


def f2_mul(x, y):
    return ((x[0] * y[0] - x[1] * y[1]) % Q, (x[0] * y[1] + x[1] * y[0]) % Q)


def f2_inv(x):
    d = pow(x[0] * x[0] + x[1] * x[1], -1, Q)
    return (x[0] * d % Q, (-x[1]) * d % Q)


def f2_scalar(x, k):
    return (x[0] * k % Q, x[1] * k % Q)


# ---------------------------------------------------------------- affine curve arithmetic (None = infinity)
class Fq1Ops:
    zero = 0
    add = staticmethod(lambda a, b: (a + b) % Q)
    sub = staticmethod(lambda a, b: (a - b) % Q)
    mul = staticmethod(lambda a, b: (a * b) % Q)
    inv = staticmethod(lambda a: pow(a, -1, Q))
    small = staticmethod(lambda a, k: (a * k) % Q)


class Fq2Ops:
    zero = (0, 0)
    add = staticmethod(f2_add)
    sub = staticmethod(f2_sub)
    mul = staticmethod(f2_mul)
    inv = staticmethod(f2_inv)
    small = staticmethod(f2_scalar)


def ec_add(F, p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if y1 == y2:
            if y1 == F.zero:
                return None
            lam = F.mul(F.small(F.mul(x1, x1), 3), F.inv(F.small(y1, 2)))
        else:
            return None
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_neg(F, p):
    if p is None:
        return None
    return (p[0], F.sub(F.zero, p[1]))


def ec_mul(F, p, k):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(F, acc, p)
        p = ec_add(F, p, p)
        k >>= 1
    return acc


def ec_msm(F, points, scalars):
    acc = None
    for p, k in zip(points, scalars):
        acc = ec_add(F, acc, ec_mul(F, p, k))
    return acc


# ---------------------------------------------------------------- byte encodings used by zkey / the C-ABI
def g1_aff_bytes(p):
    """Affine G1 point (standard ints, or None) -> 64 B Montgomery LE (zkey format)."""
    if p is None:
        return b"\0" * 64
    return limbs(to_mont(p[0], Q)) + limbs(to_mont(p[1], Q))


def g1_aff_from_bytes(b):
    x = from_mont(unlimbs(b[0:32]), Q)
    y = from_mont(unlimbs(b[32:64]), Q)
    if unlimbs(b[0:32]) == 0 and unlimbs(b[32:64]) == 0:
        return None
    return (x, y)


def g2_aff_bytes(p):
    if p is None:
        return b"\0" * 128
    (xa, xb), (ya, yb) = p
    return b"".join(limbs(to_mont(v, Q)) for v in (xa, xb, ya, yb))


def g2_aff_from_bytes(b):
    v = [unlimbs(b[32 * i:32 * i + 32]) for i in range(4)]
    if not any(v):
        return None
    v = [from_mont(x, Q) for x in v]
    return ((v[0], v[1]), (v[2], v[3]))


def xyzz_to_affine_g1(b):
    """128 B XYZZ Montgomery -> affine ints / None."""
    x, y, zz, zzz = (from_mont(unlimbs(b[32 * i:32 * i + 32]), Q) for i in range(4))
    if zz == 0:
        return None
    return (x * pow(zz, -1, Q) % Q, y * pow(zzz, -1, Q) % Q)


# ---------------------------------------------------------------- NTT model (fft.cpp semantics)
def root_of_unity(log2n):
    """g = 5^((r-1)/2^log2n): the primitive 2^log2n-th root the reference's table is built from
    (fft.cpp:60-97; quadratic non-residue search finds 5)."""
    return pow(5, (R - 1) >> log2n, R)


def ntt_naive(a, log2n):
    """Forward transform exactly as FFT::fft computes it: A[k] = sum_j a[j] * w^(jk), w = root(log2n)."""
    n = 1 << log2n
    w = root_of_unity(log2n)
    return [sum(a[j] * pow(w, (j * k) % n, R) for j in range(n)) % R for k in range(n)]


def intt_naive(a, log2n):
    n = 1 << log2n
    w = pow(root_of_unity(log2n), -1, R)
    ninv = pow(n, -1, R)
    return [sum(a[j] * pow(w, (j * k) % n, R) for j in range(n)) * ninv % R for k in range(n)]


# ---------------------------------------------------------------- splitmix64 (deterministic test inputs)
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def below(self, p):
        """Uniform in [0,p) by rejection on 254-bit draws."""
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << 254) - 1
            if v < p:
                return v


def pack_u64s(vals):
    return struct.pack("<%dQ" % len(vals), *vals)
