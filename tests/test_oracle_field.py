"""Pins the oracle's Fq / Fr / Fq2 arithmetic.

1. the raw Montgomery known-answer vectors the reference's own tests declare
   (test_prover.cpp Fr_Rw_* / Fq_Rw_*, extracted by tests/golden/make_field_kats.py);
2. an independent Python big-int model on seeded random inputs;
3. (build container only) the reference's field sources compiled as they lie (oracle/_ref).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import pymodel as pm

OPS = {"add": ol.OP_ADD, "sub": ol.OP_SUB, "neg": ol.OP_NEG, "mul": ol.OP_MUL, "sqr": ol.OP_SQR,
       "tomont": ol.OP_TOMONT, "frommont": ol.OP_FROMMONT}
MOD = {"Fq": pm.Q, "Fr": pm.R}
FID = {"Fq": ol.FQ, "Fr": ol.FR}


def _limbs_to_bytes(l):
    return b"".join(int(x, 16).to_bytes(8, "little") for x in l)


def _kats(golden_dir):
    return json.load(open(os.path.join(golden_dir, "field_kats.json")))


def test_reference_kats_all(golden_dir):
    """Every raw KAT the reference declares (canonical and non-canonical operands) matches bit-for-bit."""
    kats = _kats(golden_dir)
    assert len(kats) >= 60
    for k in kats:
        a = _limbs_to_bytes(k["a"])
        b = _limbs_to_bytes(k["b"]) if k["b"] else None
        got = ol.field_op(FID[k["field"]], OPS[k["op"]], a, b)
        assert got == _limbs_to_bytes(k["r"]), (k["field"], k["op"], k["case"])


@pytest.mark.parametrize("fname", ["Fq", "Fr"])
def test_against_python_model(fname):
    p, fid = MOD[fname], FID[fname]
    rng = pm.SplitMix64(0xC0FFEE + fid)
    vals = [0, 1, p - 1, p - 2, (1 << 256) % p, 2 ** 64 - 1, 2 ** 128, 2 ** 192 - 1] + [rng.below(p) for _ in range(200)]
    for i in range(len(vals) - 1):
        a, b = vals[i], vals[i + 1]
        A, B = pm.limbs(a), pm.limbs(b)
        assert pm.unlimbs(ol.field_op(fid, ol.OP_ADD, A, B)) == (a + b) % p
        assert pm.unlimbs(ol.field_op(fid, ol.OP_SUB, A, B)) == (a - b) % p
        assert pm.unlimbs(ol.field_op(fid, ol.OP_NEG, A)) == (-a) % p
        assert pm.unlimbs(ol.field_op(fid, ol.OP_MUL, A, B)) == pm.mont_mul(a, b, p)
        assert pm.unlimbs(ol.field_op(fid, ol.OP_SQR, A)) == pm.mont_mul(a, a, p)
        assert pm.unlimbs(ol.field_op(fid, ol.OP_TOMONT, A)) == pm.to_mont(a, p)
        assert pm.unlimbs(ol.field_op(fid, ol.OP_FROMMONT, A)) == pm.from_mont(a, p)
    for a in vals[1:40]:
        inv = pm.unlimbs(ol.field_op(fid, ol.OP_INV, pm.limbs(pm.to_mont(a, p))))
        assert pm.from_mont(inv, p) == pow(a, -1, p)
    assert ol.field_op(fid, ol.OP_INV, pm.limbs(0)) == pm.limbs(0)


def test_decimal_io():
    rng = pm.SplitMix64(7)
    for p, fid in ((pm.Q, ol.FQ), (pm.R, ol.FR)):
        for v in [0, 1, 10 ** 19, 10 ** 19 - 1, 10 ** 38, p - 1] + [rng.below(p) for _ in range(50)]:
            m = pm.limbs(pm.to_mont(v, p))
            assert ol.fe_to_dec(fid, m) == str(v)
            assert ol.fe_from_dec(fid, str(v)) == m


def test_fq2_against_model():
    """Includes the reference's own F2 KAT: (2,2)*(3,3) == (0,12)  (alt_bn128_test.cpp:12-30)."""
    def enc(x):
        return pm.limbs(pm.to_mont(x[0], pm.Q)) + pm.limbs(pm.to_mont(x[1], pm.Q))

    def dec(b):
        return (pm.from_mont(pm.unlimbs(b[:32]), pm.Q), pm.from_mont(pm.unlimbs(b[32:]), pm.Q))

    assert dec(ol.fq2_op(ol.OP_MUL, enc((2, 2)), enc((3, 3)))) == (0, 12)
    rng = pm.SplitMix64(99)
    for _ in range(100):
        x = (rng.below(pm.Q), rng.below(pm.Q))
        y = (rng.below(pm.Q), rng.below(pm.Q))
        assert dec(ol.fq2_op(ol.OP_MUL, enc(x), enc(y))) == pm.f2_mul(x, y)
        assert dec(ol.fq2_op(ol.OP_SQR, enc(x))) == pm.f2_mul(x, x)
        assert dec(ol.fq2_op(ol.OP_ADD, enc(x), enc(y))) == pm.f2_add(x, y)
        assert dec(ol.fq2_op(ol.OP_SUB, enc(x), enc(y))) == pm.f2_sub(x, y)
        assert dec(ol.fq2_op(ol.OP_INV, enc(x))) == pm.f2_inv(x)


REF_SO = os.path.join(ol.ORACLE_DIR, "_ref", "libref_field.so")


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_against_compiled_reference_field():
    """oracle/_ref = the reference's fq/fr/f2field sources compiled unmodified (oracle/build_ref.sh)."""
    ref = C.CDLL(REF_SO)
    ref.ref_field_op_vec.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    ref.ref_fq2_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = pm.SplitMix64(2024)
    n = 5000
    for fid, p in ((ol.FQ, pm.Q), (ol.FR, pm.R)):
        def arr():
            return np.array([[(v >> (64 * i)) & (2 ** 64 - 1) for i in range(4)]
                             for v in (rng.below(p) for _ in range(n))], dtype=np.uint64)
        a, b = arr(), arr()
        a[0] = 0
        b[0] = 0
        a[1] = [((p - 1) >> (64 * i)) & (2 ** 64 - 1) for i in range(4)]
        b[1] = a[1]
        for op in range(8):
            nn = n if op != ol.OP_INV else 100
            got = ol.field_op_vec(fid, op, a[:nn], b[:nn])
            want = np.zeros((nn, 4), dtype=np.uint64)
            ref.ref_field_op_vec(fid, op, a[:nn].ctypes.data, b[:nn].ctypes.data, want.ctypes.data, nn)
            assert np.array_equal(got, want), (fid, op)
    # Fq2
    for _ in range(500):
        x = np.array([(v >> (64 * i)) & (2 ** 64 - 1) for v in (rng.below(pm.Q), rng.below(pm.Q)) for i in range(4)],
                     dtype=np.uint64)
        y = np.array([(v >> (64 * i)) & (2 ** 64 - 1) for v in (rng.below(pm.Q), rng.below(pm.Q)) for i in range(4)],
                     dtype=np.uint64)
        for op in (0, 1, 2, 3, 4, 7):
            want = np.zeros(8, dtype=np.uint64)
            ref.ref_fq2_op(op, x.ctypes.data, y.ctypes.data, want.ctypes.data)
            assert ol.fq2_op(op, x.tobytes(), y.tobytes()) == want.tobytes(), op
