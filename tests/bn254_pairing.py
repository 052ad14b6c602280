"""Minimal pure-Python BN254 optimal-ate pairing check, used ONLY by tests to apply the
reference's own acceptance criterion for a proof: it must verify under the circuit's
verification key (prover-service/src/tests/prover_handler.rs:279-290 verifies every toy
proof with public input 2; prover_handler.rs:329-336 re-verifies every production proof).

Textbook construction: Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), G2 points untwisted into
E(Fq12), affine Miller loop, one shared final exponentiation.  Slow (seconds), tiny inputs.
"""
import json

from pymodel import Q, R

ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE = 63
FQ12_MOD = (82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0)  # w^12 = 18 w^6 - 82


class F12:
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [x % Q for x in c]

    @staticmethod
    def one():
        return F12([1] + [0] * 11)

    @staticmethod
    def zero():
        return F12([0] * 12)

    def __add__(self, o):
        return F12([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return F12([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return F12([-a for a in self.c])

    def __eq__(self, o):
        return self.c == o.c

    def scale(self, k):
        return F12([a * k for a in self.c])

    def __mul__(self, o):
        if isinstance(o, int):
            return self.scale(o)
        b = [0] * 23
        for i, x in enumerate(self.c):
            if x:
                for j, y in enumerate(o.c):
                    b[i + j] += x * y
        for i in range(22, 11, -1):
            t = b[i]
            if t:
                b[i - 6] += 18 * t
                b[i - 12] -= 82 * t
        return F12(b[:12])

    def inv(self):
        # extended Euclid over Fq[w]
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = self.c + [0], [x % Q for x in FQ12_MOD] + [1]

        def deg(p):
            d = len(p) - 1
            while d and p[d] == 0:
                d -= 1
            return d

        def poly_div(a, b):
            dega, degb = deg(a), deg(b)
            temp, o = list(a), [0] * len(a)
            lead_inv = pow(b[degb], -1, Q)
            for i in range(dega - degb, -1, -1):
                o[i] = temp[degb + i] * lead_inv % Q
                for c in range(degb + 1):
                    temp[c + i] = (temp[c + i] - o[i] * b[c]) % Q
            return o[: deg(o) + 1]

        while deg(low):
            r = poly_div(high, low)
            r += [0] * (13 - len(r))
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % Q
                    new[i + j] = (new[i + j] - low[i] * r[j]) % Q
            lm, low, hm, high = nm, new, lm, low
        inv0 = pow(low[0], -1, Q)
        return F12([x * inv0 for x in lm[:12]])

    def __truediv__(self, o):
        return self * o.inv()

    def __pow__(self, e):
        r, b = F12.one(), self
        while e:
            if e & 1:
                r = r * b
            b = b * b
            e >>= 1
        return r


W = F12([0, 1] + [0] * 10)
W2 = W * W
W3 = W2 * W


def _f12_from_int(x):
    return F12([x] + [0] * 11)


def twist(pt):
    """E'(Fq2) -> E(Fq12).  Fq2 here is a + b*u with u^2 = -1; w^6 = 9 + u."""
    (xa, xb), (ya, yb) = pt
    nx = F12([xa - 9 * xb] + [0] * 5 + [xb] + [0] * 5)
    ny = F12([ya - 9 * yb] + [0] * 5 + [yb] + [0] * 5)
    return (nx * W2, ny * W3)


def _double(p):
    x, y = p
    m = (x * x).scale(3) / y.scale(2)
    nx = m * m - x.scale(2)
    return (nx, m * (x - nx) - y)


def _add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        return _double(p1) if y1 == y2 else None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def _line(p1, p2, t):
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if not (x1 == x2):
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1).scale(3) / y1.scale(2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(q_g2, p_g1):
    """Miller loop WITHOUT final exponentiation. q_g2: ((xa,xb),(ya,yb)) ints; p_g1: (x,y) ints."""
    if q_g2 is None or p_g1 is None:
        return F12.one()
    Qt = twist(q_g2)
    P = (_f12_from_int(p_g1[0]), _f12_from_int(p_g1[1]))
    Rr, f = Qt, F12.one()
    for i in range(LOG_ATE, -1, -1):
        f = f * f * _line(Rr, Rr, P)
        Rr = _double(Rr)
        if ATE_LOOP_COUNT & (1 << i):
            f = f * _line(Rr, Qt, P)
            Rr = _add(Rr, Qt)
    Q1 = (Qt[0] ** Q, Qt[1] ** Q)
    nQ2 = (Q1[0] ** Q, -(Q1[1] ** Q))
    f = f * _line(Rr, Q1, P)
    Rr = _add(Rr, Q1)
    f = f * _line(Rr, nQ2, P)
    return f


def final_exp(f):
    return f ** ((Q ** 12 - 1) // R)


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % Q)


def groth16_verify(vk, proof, public_inputs):
    """vk: snarkjs verification_key.json dict; proof: snarkjs proof dict (decimal strings)."""
    from pymodel import Fq1Ops, ec_add, ec_mul

    def g1(v):
        return (int(v[0]), int(v[1]))

    def g2(v):
        return ((int(v[0][0]), int(v[0][1])), (int(v[1][0]), int(v[1][1])))

    A, B, Cc = g1(proof["pi_a"]), g2(proof["pi_b"]), g1(proof["pi_c"])
    alpha, beta = g1(vk["vk_alpha_1"]), g2(vk["vk_beta_2"])
    gamma, delta = g2(vk["vk_gamma_2"]), g2(vk["vk_delta_2"])
    ic = [g1(v) for v in vk["IC"]]
    assert len(ic) == len(public_inputs) + 1
    vkx = ic[0]
    for w, pt in zip(public_inputs, ic[1:]):
        vkx = ec_add(Fq1Ops, vkx, ec_mul(Fq1Ops, pt, w % R))
    # e(A,B) * e(-alpha,beta) * e(-vkx,gamma) * e(-C,delta) == 1
    f = miller_loop(B, A) * miller_loop(beta, g1_neg(alpha)) * miller_loop(gamma, g1_neg(vkx)) * miller_loop(
        delta, g1_neg(Cc))
    return final_exp(f) == F12.one()


def verify_json(vk_path, proof_json, public_inputs):
    vk = json.load(open(vk_path))
    return groth16_verify(vk, json.loads(proof_json), public_inputs)
