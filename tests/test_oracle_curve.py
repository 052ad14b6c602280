"""Pins the oracle's curve / MSM / NTT layer against the reference's own unit-test vectors
(rust-rapidsnark/rapidsnark/src/alt_bn128_test.cpp) and an independent Python model."""
import numpy as np
import pytest

import oracle_lib as ol
import pymodel as pm

G1, G2 = ol.G1, ol.G2
ORDER = pm.R


def sc(k):
    return pm.limbs(k)


def g1_xyzz_of_aff(aff):
    one = pm.limbs(pm.to_mont(1, pm.Q))
    return bytes(aff) + one + one


def g1_aff_std(aff_bytes):
    return pm.g1_aff_from_bytes(aff_bytes)


def test_g1_identities():
    """alt_bn128_test.cpp:32-124 : P+0, P-P, 4P two ways, 3P, 5P."""
    g = ol.generator(G1)
    one = g1_xyzz_of_aff(g)
    zero = ol.mul_scalar(G1, g, sc(0))
    assert ol.pt_to_affine(G1, zero) == b"\0" * 64
    assert ol.pt_eq(G1, ol.pt_op(G1, ol.PT_ADD, one, zero), one)
    neg = ol.pt_op(G1, ol.PT_NEG, one)
    assert ol.pt_to_affine(G1, ol.pt_op(G1, ol.PT_ADD, one, neg)) == b"\0" * 64
    p = ol.pt_op(G1, ol.PT_MADD, one, g)  # 2G via the P==Q branch of madd
    p = ol.pt_op(G1, ol.PT_MADD, p, g)
    p4 = ol.pt_op(G1, ol.PT_MADD, p, g)
    d = ol.pt_op(G1, ol.PT_DBL, ol.pt_op(G1, ol.PT_DBL, one))
    assert ol.pt_eq(G1, p4, d)
    assert ol.pt_eq(G1, p, ol.pt_op(G1, ol.PT_ADD, d, neg))
    assert ol.pt_eq(G1, ol.mul_scalar(G1, g, sc(3)), p)
    p5 = ol.pt_op(G1, ol.PT_ADD, d, one)
    assert g1_aff_std(ol.pt_to_affine(G1, p5)) == pm.ec_mul(pm.Fq1Ops, pm.G1, 5)


def test_scalar_mul_65_and_order():
    """alt_bn128_test.cpp:104-170 : 65*G; r*G == infinity on G1 and G2."""
    g = ol.generator(G1)
    assert g1_aff_std(ol.pt_to_affine(G1, ol.mul_scalar(G1, g, sc(65)))) == pm.ec_mul(pm.Fq1Ops, pm.G1, 65)
    assert ol.pt_to_affine(G1, ol.mul_scalar(G1, g, sc(ORDER))) == b"\0" * 64
    g2 = ol.generator(G2)
    assert ol.pt_to_affine(G2, ol.mul_scalar(G2, g2, sc(ORDER))) == b"\0" * 128
    assert pm.g2_aff_from_bytes(ol.pt_to_affine(G2, ol.mul_scalar(G2, g2, sc(7)))) == pm.ec_mul(pm.Fq2Ops, pm.G2, 7)


def test_msm_two_point_kat():
    """alt_bn128_test.cpp:215-248 (multiExp2): explicit expected affine output."""
    bases = [
        (1626275109576878988287730541908027724405348106427831594181487487855202143055,
         18706364085805828895917702468512381358405767972162700276238017959231481018884),
        (17245156998235704504461341147511350131061011207199931581281143511105381019978,
         3858908536032228066651712470282632925312300188207189106507111128103204506804),
    ]
    scalars = [1, 20187316456970436521602619671088988952475789765726813868033071292105413408473]
    want = (9163953212624378696742080269971059027061360176019470242548968584908855004282,
            20922060990592511838374895951081914567856345629513259026540392951012456141360)
    B = np.frombuffer(b"".join(pm.g1_aff_bytes(p) for p in bases), dtype=np.uint8).reshape(2, 64)
    S = np.frombuffer(b"".join(sc(s) for s in scalars), dtype=np.uint8).reshape(2, 32)
    _, aff = ol.msm(G1, B, S)
    assert g1_aff_std(aff) == want


def test_msm_closed_form_40000():
    """alt_bn128_test.cpp:172-212 (multiExp): sum (i+1)*((i+1)G) == (sum (i+1)^2) G, n = 40000."""
    n = 40000
    B = ol.gen_points(G1, 0, n)
    S = np.zeros((n, 32), dtype=np.uint8)
    S[:, :4] = np.arange(1, n + 1, dtype=np.uint32).view(np.uint8).reshape(n, 4)
    acc = sum((i + 1) * (i + 1) for i in range(n))
    x, _ = ol.msm(G1, B, S, nthreads=4)
    want = ol.mul_scalar(G1, ol.generator(G1), sc(acc))
    assert ol.pt_eq(G1, x, want)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 16, 33, 100])
@pytest.mark.parametrize("group", [G1, G2])
def test_msm_small_vs_python(group, n):
    rng = pm.SplitMix64(1000 + n + 17 * group)
    F, gen = (pm.Fq1Ops, pm.G1) if group == G1 else (pm.Fq2Ops, pm.G2)
    enc = pm.g1_aff_bytes if group == G1 else pm.g2_aff_bytes
    dec = pm.g1_aff_from_bytes if group == G1 else pm.g2_aff_from_bytes
    pts = [pm.ec_mul(F, gen, (rng.next() % (1 << 40)) + 1) for _ in range(n)]
    if n >= 5:
        pts[3] = None          # (0,0) base must be skipped (multiexp.cpp:59)
        pts[4] = pts[2]        # duplicate base
    scal = [rng.below(pm.R) for _ in range(n)]
    if n >= 3:
        scal[0] = 0
        scal[1] = (1 << 256) - 1   # scalars are not range-checked (Appendix B)
    B = np.frombuffer(b"".join(enc(p) for p in pts), dtype=np.uint8).reshape(n, ol.AFF_BYTES[group])
    S = np.frombuffer(b"".join(sc(s) for s in scal), dtype=np.uint8).reshape(n, 32)
    _, aff = ol.msm(group, B, S)
    assert dec(aff) == pm.ec_msm(F, pts, [s % pm.R for s in scal])


def test_msm_threads_and_windows_agree():
    """c = clamp(log2(n/2),2,16) takes different values across these n; threaded == serial."""
    for n in (7, 300, 5000):
        rng = pm.SplitMix64(n)
        B = ol.gen_points(G1, 5, n)
        S = np.frombuffer(b"".join(sc(rng.below(pm.R)) for _ in range(n)), dtype=np.uint8).reshape(n, 32)
        a1 = ol.msm(G1, B, S, nthreads=1)[1]
        a4 = ol.msm(G1, B, S, nthreads=4)[1]
        assert a1 == a4
        if n <= 300:
            pts = [g1_aff_std(bytes(B[i])) for i in range(n)]
            want = pm.ec_msm(pm.Fq1Ops, pts, [pm.unlimbs(bytes(S[i])) for i in range(n)])
            assert g1_aff_std(a1) == want


def test_gen_points_matches_model():
    B = ol.gen_points(G1, 10, 6)
    for i in range(6):
        assert g1_aff_std(bytes(B[i])) == pm.ec_mul(pm.Fq1Ops, pm.G1, 11 + i)
    B2 = ol.gen_points(G2, 0, 4)
    for i in range(4):
        assert pm.g2_aff_from_bytes(bytes(B2[i])) == pm.ec_mul(pm.Fq2Ops, pm.G2, 1 + i)


def _fr_arr(vals):
    return np.array([[(pm.to_mont(v, pm.R) >> (64 * i)) & (2 ** 64 - 1) for i in range(4)] for v in vals],
                    dtype=np.uint64)


def _fr_vals(arr):
    return [pm.from_mont(sum(int(arr[k, i]) << (64 * i) for i in range(4)), pm.R) for k in range(arr.shape[0])]


def test_ntt_roundtrip_1024():
    """alt_bn128_test.cpp:250-271 : fft then ifft of 1..1024 is the identity."""
    n = 1 << 10
    a = _fr_arr(range(1, n + 1))
    f = ol.ntt(a)
    back = ol.ntt(f, inverse=True)
    assert np.array_equal(back, a)
    assert not np.array_equal(f, a)


@pytest.mark.parametrize("log2n", [1, 2, 3, 5, 6])
def test_ntt_vs_naive_dft(log2n):
    n = 1 << log2n
    rng = pm.SplitMix64(log2n)
    vals = [rng.below(pm.R) for _ in range(n)]
    a = _fr_arr(vals)
    assert _fr_vals(ol.ntt(a)) == pm.ntt_naive(vals, log2n)
    assert _fr_vals(ol.ntt(a, inverse=True)) == pm.intt_naive(vals, log2n)
    # a table built for a larger domain gives the same transform (groth16.hpp:96 uses 2*domainSize)
    assert np.array_equal(ol.ntt(a, max_domain=4 * n), ol.ntt(a))


def test_ntt_root_table():
    # root(k, j) = g_S^(j << (S-k)) with g_S = 5^((r-1)/2^S)   (fft.hpp:40-43, fft.cpp:93-97)
    S = 6
    g = pm.root_of_unity(S)
    for k, j in ((6, 1), (6, 5), (3, 1), (4, 3), (1, 1)):
        got = pm.from_mont(pm.unlimbs(ol.ntt_root(1 << S, k, j)), pm.R)
        assert got == pow(g, j << (S - k), pm.R)
    assert pm.from_mont(pm.unlimbs(ol.ntt_root(1 << S, 1, 1)), pm.R) == pm.R - 1
