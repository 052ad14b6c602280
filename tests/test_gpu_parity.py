"""-m gpu : the HIP path, called through the C ABI (libk16.so), against the CPU oracle.
Bit-exact: integer arithmetic only."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import pymodel as pm
from gpu_common import np_scalars, rand_fe_array

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import k16
    c = k16.Context(0)   # raises if libk16.so is missing or there is no GPU: no fallback
    yield c
    c.close()


# ---------------------------------------------------------------- field / curve primitives
@pytest.mark.parametrize("field", [0, 1])
def test_field_ops(ctx, field):
    p = pm.Q if field == 0 else pm.R
    rng = pm.SplitMix64(77 + field)
    n = 4096
    a, b = rand_fe_array(rng, p, n), rand_fe_array(rng, p, n)
    b[:8] = a[:8][::-1]
    for op in range(7):
        got = ctx.field_op_vec(field, op, a, b)
        want = ol.field_op_vec(field, op, a, b)
        assert np.array_equal(got, want), (field, op)


@pytest.mark.parametrize("group", [0, 1])
def test_point_ops_incl_exceptional_cases(ctx, group):
    """curve.cpp:91-250 branches: inf+P, P+inf, P+P (-> dbl), P+(-P) (-> inf), (0,0) affine operand."""
    import k16
    n = 64
    aff = ol.gen_points(group, 3, n)
    xb, ab = k16.XYZZ_BYTES[group], k16.AFF_BYTES[group]
    g = ol.generator(group)
    # projective operands: k*G in non-trivial XYZZ form (zz != 1)
    proj = np.zeros((n, xb), dtype=np.uint8)
    for i in range(n):
        proj[i] = np.frombuffer(ol.mul_scalar(group, g, pm.limbs(i + 4)), dtype=np.uint8)   # == aff[i] as a point
    other = np.zeros((n, xb), dtype=np.uint8)
    for i in range(n):
        other[i] = np.frombuffer(ol.mul_scalar(group, g, pm.limbs(1000 + 7 * i)), dtype=np.uint8)
    inf = np.frombuffer(ol.mul_scalar(group, g, pm.limbs(0)), dtype=np.uint8)
    p1 = other.copy()
    p2x = proj.copy()
    p2a = aff.copy()
    p1[0] = inf                       # inf + P
    p2x[1] = inf                      # P + inf
    p2a[1] = 0                        # P + (0,0)
    p1[2] = proj[2]                   # P + P  -> dbl branch
    p1[3] = np.frombuffer(ol.pt_op(group, ol.PT_NEG, bytes(proj[3])), dtype=np.uint8)  # P + (-P) -> inf
    p1[4] = inf
    p2x[4] = inf
    p2a[4] = 0                        # inf + inf
    for op, p2 in ((k16.PT_ADD, p2x), (k16.PT_MADD, p2a), (k16.PT_DBL, None)):
        got = ctx.point_op_vec(group, op, p1, p2)
        for i in range(n):
            want = ol.pt_op(group, {k16.PT_ADD: ol.PT_ADD, k16.PT_MADD: ol.PT_MADD, k16.PT_DBL: ol.PT_DBL}[op],
                            bytes(p1[i]), bytes(p2[i]) if p2 is not None else None)
            # same formulas, same branch order => identical XYZZ representation, not just the same point
            assert bytes(got[i]) == want, (group, op, i)


# ---------------------------------------------------------------- MSM
def _check_msm(ctx, group, bases, scalars, threads=4):
    _, got = ctx.msm(group, bases, scalars)
    _, want = ol.msm(group, bases, scalars, nthreads=threads)
    assert got == want


@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 64, 255, 1000])
@pytest.mark.parametrize("group", [0, 1])
def test_msm_small(ctx, group, n):
    bases = ol.gen_points(group, 0, max(n, 1))[:n]
    scalars = np_scalars(100 + n, n, "full256")
    if n >= 3:
        bases[1] = 0                      # (0,0) base is skipped (multiexp.cpp:59)
        bases[2] = bases[0]               # duplicate base
        scalars[0] = 0
    _check_msm(ctx, group, bases, scalars)


@pytest.mark.parametrize("kind", ["uniform", "full256", "ones", "zeros", "witness", "topwindow", "same"])
def test_msm_g1_distributions(ctx, kind):
    n = 1 << 13
    bases = ol.gen_points(0, 0, n)
    _check_msm(ctx, 0, bases, np_scalars(5, n, kind))


def test_msm_g1_two_point_kat(ctx):
    """alt_bn128_test.cpp:215-248 through the HIP path."""
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
    S = np.frombuffer(b"".join(pm.limbs(s) for s in scalars), dtype=np.uint8).reshape(2, 32)
    _, aff = ctx.msm(0, B, S)
    assert pm.g1_aff_from_bytes(aff) == want


def test_msm_g1_closed_form_40000(ctx):
    """alt_bn128_test.cpp:172-212: sum (i+1)*((i+1)G) == (sum (i+1)^2) G."""
    n = 40000
    B = ol.gen_points(0, 0, n)
    S = np.zeros((n, 32), dtype=np.uint8)
    S[:, :4] = np.arange(1, n + 1, dtype=np.uint32).view(np.uint8).reshape(n, 4)
    x, _ = ctx.msm(0, B, S)
    want = ol.mul_scalar(0, ol.generator(0), pm.limbs(sum((i + 1) ** 2 for i in range(n))))
    assert ol.pt_eq(0, x, want)


@pytest.mark.parametrize("c", [4, 7, 11, 13, 16])
def test_msm_window_sizes_agree(ctx, c):
    n = 3000
    bases = ol.gen_points(0, 11, n)
    scalars = np_scalars(9, n, "full256")
    ctx.set_window_bits(c)
    try:
        _check_msm(ctx, 0, bases, scalars)
    finally:
        ctx.set_window_bits(0)


def test_msm_g2_medium(ctx):
    n = 1 << 11
    bases = ol.gen_points(1, 0, n)
    _check_msm(ctx, 1, bases, np_scalars(21, n, "uniform"))
    _check_msm(ctx, 1, bases, np_scalars(22, n, "witness"))


def test_msm_g1_2p16_vs_oracle(ctx):
    n = 1 << 16
    bases = ol.gen_points(0, 0, n)
    _check_msm(ctx, 0, bases, np_scalars(31, n, "uniform"), threads=8)


def test_msm_g1_full_size_properties(ctx):
    """BASELINE config 2 size (2^20): closed form and linearity instead of the (slow) oracle."""
    n = 1 << 20
    bases = ol.gen_points(0, 0, n)
    d_b = ctx.to_device(bases)
    # (1) closed form with scalars i+1
    S = np.zeros((n, 32), dtype=np.uint8)
    S[:, :4] = np.arange(1, n + 1, dtype=np.uint32).view(np.uint8).reshape(n, 4)
    d_s = ctx.to_device(S)
    x, _ = ctx.msm_device(0, d_b, d_s, n)
    total = n * (n + 1) * (2 * n + 1) // 6
    assert ol.pt_eq(0, x, ol.mul_scalar(0, ol.generator(0), pm.limbs(total % pm.R)))
    # (2) linearity: MSM(s) + MSM(t) == MSM(s + t) for uniform s, t (no modular wrap: both < 2^252)
    s = np_scalars(1, n, "uniform")
    t = np_scalars(2, n, "uniform")
    s[:, 31] &= 0x0F
    t[:, 31] &= 0x0F
    s64, t64 = s.view(np.uint64).astype(object), t.view(np.uint64).astype(object)
    carry = np.zeros(n, dtype=object)
    u = np.zeros((n, 4), dtype=np.uint64)
    for k in range(4):
        tot = s64[:, k] + t64[:, k] + carry
        u[:, k] = np.array([int(v) & (2 ** 64 - 1) for v in tot], dtype=np.uint64)
        carry = np.array([int(v) >> 64 for v in tot], dtype=object)
    xs, _ = ctx.msm_device(0, d_b, d_s.upload(s), n)
    xt, _ = ctx.msm_device(0, d_b, d_s.upload(t), n)
    xu, au = ctx.msm_device(0, d_b, d_s.upload(u.view(np.uint8).reshape(n, 32)), n)
    import k16
    _, a_sum = k16.points_sum(0, np.frombuffer(xs + xt, dtype=np.uint8).reshape(2, 128))
    assert a_sum == au
    d_b.free()
    d_s.free()


# ---------------------------------------------------------------- NTT
@pytest.mark.parametrize("log2n", [0, 1, 2, 3, 8, 12, 16])
def test_ntt_vs_oracle(ctx, log2n):
    n = 1 << log2n
    rng = pm.SplitMix64(log2n)
    a = rand_fe_array(rng, pm.R, n, edge=n >= 4)
    for inverse in (False, True):
        for md in (n, 2 * n):
            got = ctx.ntt(a, max_domain=md, inverse=inverse)
            want = ol.ntt(a, max_domain=md, inverse=inverse)
            assert np.array_equal(got, want), (log2n, inverse, md)


def test_ntt_roundtrip_2p21(ctx):
    """Keyless domain size: iNTT(NTT(a)) == a (alt_bn128_test.cpp:250-271 at full size)."""
    n = 1 << 21
    rs = np.random.RandomState(3)
    a = rs.randint(0, 2 ** 63, size=(n, 4)).astype(np.uint64)
    a[:, 3] &= (1 << 60) - 1     # < r
    f = ctx.ntt(a, max_domain=2 * n)
    back = ctx.ntt(f, max_domain=2 * n, inverse=True)
    assert np.array_equal(back, a)
    assert not np.array_equal(f, a)


# ---------------------------------------------------------------- full prove
def test_toy_proof_bit_exact(ctx, toy_paths):
    import k16
    zkey, wtns, vk = toy_paths
    p = k16.Prover(ctx, zkey)
    assert p.info() == dict(n_vars=3, n_public=1, domain_size=4, n_coefs=4)
    for r, s in ((0, 0), (12345, pm.R - 1), (pm.SplitMix64(5).below(pm.R), pm.SplitMix64(6).below(pm.R))):
        got = p.prove_file(wtns, pm.limbs(r), pm.limbs(s))
        want = ol.prove_files(zkey, wtns, pm.limbs(r), pm.limbs(s))
        assert got == want
    # production path: CSPRNG blinding; proof must verify (reference criterion, prover_handler.rs:279-290)
    import bn254_pairing as bp
    js = p.prove_file(wtns)
    assert bp.verify_json(vk, js, [2])
    assert js != p.prove_file(wtns)
    p.close()


@pytest.mark.parametrize("shape", [(6, 1, 1, 4), (8, 1, 2, 8), (30, 2, 8, 40), (300, 1, 1024, 900), (700, 1, 2048, 2000),
                                   (3000, 2, 4096, 9000), (20000, 1, 1 << 15, 60000), (40000, 1, 1 << 16, 150000),
                                   (70000, 1, 1 << 17, 200000), (140000, 1, 1 << 18, 400000)])
def test_synthetic_circuit_proof_bit_exact(ctx, tmp_path, shape):
    """Non-toy prove(): random circuit of the given (nVars, nPublic, domainSize, nCoefs); the HIP proof JSON and
    the H scalars must equal the oracle's byte for byte (same witness, same injected r, s)."""
    import k16
    import zkey_builder as zb
    n_vars, n_pub, N, n_coefs = shape
    zk, wt = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    zb.build_zkey(zk, n_vars, n_pub, N, n_coefs, seed=11)
    w = zb.build_wtns(wt, n_vars, seed=12)
    r, s = pm.limbs(pm.SplitMix64(77).below(pm.R)), pm.limbs(pm.SplitMix64(78).below(pm.R))
    p = k16.Prover(ctx, zk)
    assert p.info() == dict(n_vars=n_vars, n_public=n_pub, domain_size=N, n_coefs=n_coefs)
    got = p.prove_file(wt, r, s)
    h_gpu = p.last_h()
    want, h_ref = ol.prove_files(zk, wt, r, s, nthreads=8, want_h=True)
    assert np.array_equal(h_gpu, h_ref)
    assert got == want
    # witness handed over in memory gives the same proof
    assert p.prove_mem(w, r, s) == want
    p.close()


def test_proof_with_long_constraint_rows_bit_exact(ctx, tmp_path):
    """Constraint rows far from the average length (RS/groth16.cpp:137-156 walks the coefficient list, so the reference
    does not care): rows of 65, 200, 1000 and 5000 entries next to ~3 per row, rows of exactly 64 and 63, and an A / B
    matrix whose other rows are mostly empty.  The device keeps rows of <= 64 entries in length-sorted slices and gives a
    wave to each longer row (k_spmv)."""
    import k16
    import zkey_builder as zb
    n_vars, N, n_coefs = 9000, 1 << 13, 30000
    zk, wt = str(tmp_path / "l.zkey"), str(tmp_path / "l.wtns")
    zb.build_zkey(zk, n_vars, 1, N, n_coefs, seed=21, long_rows=(65, 200, 1000, 5000, 64, 63, 129))
    zb.build_wtns(wt, n_vars, seed=22)
    r, s = pm.limbs(pm.SplitMix64(5).below(pm.R)), pm.limbs(pm.SplitMix64(6).below(pm.R))
    p = k16.Prover(ctx, zk)
    got = p.prove_file(wt, r, s)
    h_gpu = p.last_h()
    want, h_ref = ol.prove_files(zk, wt, r, s, nthreads=8, want_h=True)
    assert np.array_equal(h_gpu, h_ref)
    assert got == want
    p.close()


def test_msm_g1_2p23_shard_closed_form(ctx):
    """BASELINE config 5 shard size (2^26 points over 8 GPUs = 2^23 per GPU): bases (off+i+1)*G generated on the
    device, scalars i+1, checked against the closed form sum (i+1)(off+i+1) * G; and the per-shard partials of two
    such shards folded with k16_points_sum equal the closed form of the union (the multi-GPU combine)."""
    import k16
    n = 1 << 23
    S = np.zeros((n, 32), dtype=np.uint8)
    S[:, :4] = np.arange(1, n + 1, dtype=np.uint32).view(np.uint8).reshape(n, 4)
    d_s = ctx.to_device(S)
    parts = []
    total = 0
    for shard in range(2):
        off = shard * n
        d_b = ctx.synth_points(k16.G1, off, n)
        x, _ = ctx.msm_device(k16.G1, d_b, d_s, n)
        d_b.free()
        want = sum_closed = (n * (n + 1) * (2 * n + 1) // 6 + off * n * (n + 1) // 2) % pm.R
        assert ol.pt_eq(0, x, ol.mul_scalar(0, ol.generator(0), pm.limbs(want)))
        parts.append(np.frombuffer(x, dtype=np.uint8))
        total = (total + sum_closed) % pm.R
    xs, _ = k16.points_sum(k16.G1, np.stack(parts))
    assert ol.pt_eq(0, xs, ol.mul_scalar(0, ol.generator(0), pm.limbs(total)))
    d_s.free()


@pytest.mark.parametrize("group", [0, 1])
def test_msm_witness_like_large_forced_c13(ctx, group):
    """The prover runs its witness MSMs with c = 13 once nVars >= 2^17; this exercises that configuration
    (skewed scalars: one bucket holds ~45 % of the points -> sliced sort, giant-bucket fold) against the oracle."""
    n = 150000
    bases = ol.gen_points(group, 17, n)
    bases[::3] = 0 if group == 1 else bases[::3]      # B2-like sparsity for G2
    scalars = np_scalars(41 + group, n, "witness")
    ctx.set_window_bits(13)
    try:
        _check_msm(ctx, group, bases, scalars, threads=8)
    finally:
        ctx.set_window_bits(0)


def test_msm_repeatability_stress(ctx):
    """Regression for a lost-update race (hipcc dropped the LDS wait of a barrier after a loop of no-return LDS
    atomics; see k16_lds_sync in msm_kernels.inc): 300 launches on skewed input must all equal the oracle."""
    n = 150000
    bases = ol.gen_points(0, 17, n)
    scalars = np_scalars(41, n, "witness")
    _, want = ol.msm(0, bases, scalars, nthreads=8)
    d_b, d_s = ctx.to_device(bases), ctx.to_device(scalars)
    for c in (13, 0):
        ctx.set_window_bits(c)
        try:
            bad = [i for i in range(150) if ctx.msm_device(0, d_b, d_s, n)[1] != want]
        finally:
            ctx.set_window_bits(0)
        assert bad == []
    d_b.free()
    d_s.free()


def test_generic_atomic_sort_path_agrees(monkeypatch):
    """n > 2^24 uses the generic global-atomic bucket sort instead of the LDS partition sort; force it on a small
    input and compare with the oracle (K16_ATOMIC_SORT is read when a context is created)."""
    import k16
    n = 5000
    bases = ol.gen_points(0, 3, n)
    monkeypatch.setenv("K16_ATOMIC_SORT", "1")
    c2 = k16.Context(0)
    try:
        for kind in ("full256", "witness"):
            _check_msm(c2, 0, bases, np_scalars(77, n, kind))
        _check_msm(c2, 1, ol.gen_points(1, 3, 600), np_scalars(78, 600, "full256"))
        # a zero-row mask on this sort is dropped, not refused: same sum
        d_b = c2.to_device(bases)
        prep = c2.bases_prepare(k16.G1, d_b, n)
        mask = c2.alloc(((n + 63) // 64) * 8)
        c2._chk(c2.L.k16_msm_zero_row_mask(c2.h, k16.G1, prep.ptr, n, mask.ptr))
        sc = np_scalars(79, n, "witness")
        d_s = c2.to_device(sc)
        c2._chk(c2.L.k16_msm_set_zero_row_mask(c2.h, mask.ptr))
        c2.msm_enqueue_prepared(k16.G1, prep, d_s, n)
        _, got = c2.msm_finish(k16.G1)
        assert got == ol.msm(0, bases, sc, nthreads=4)[1]
        for d in (d_b, prep, mask, d_s):
            d.free()
    finally:
        c2.close()


def test_msm_pipelined_enqueue_finish_fifo(ctx):
    """Several MSMs in flight (k16_msm_enqueue x k, then k16_msm_finish x k) return their results in order."""
    import k16
    n = 3000
    bases = ol.gen_points(0, 5, n)
    d_b = ctx.to_device(bases)
    sc = [np_scalars(500 + i, n, "full256") for i in range(5)]
    d_s = [ctx.to_device(s) for s in sc]
    for i, d in enumerate(d_s):
        ctx.set_lane(i % 3)             # MSMs on different lanes overlap; results still come back in enqueue order
        ctx.msm_enqueue(k16.G1, d_b, d, n)
    ctx.set_lane(0)
    got = [ctx.msm_finish(k16.G1)[1] for _ in d_s]
    want = [ol.msm(0, bases, s, nthreads=4)[1] for s in sc]
    assert got == want
    with pytest.raises(k16.K16Error):
        ctx.msm_finish(k16.G1)          # nothing pending


def _hash_style_points(n, seed=0x9E3779B97F4A7C15):
    """SURVEY 8(d) config 2, second point set: x from a splitmix64 stream, y = (x^3+3)^((p+1)/4), kept if square."""
    rng = pm.SplitMix64(seed)
    out = np.zeros((n, 64), dtype=np.uint8)
    k = 0
    while k < n:
        x = (rng.next() | (rng.next() << 64) | (rng.next() << 128) | (rng.next() << 192)) % pm.Q
        rhs = (x * x * x + 3) % pm.Q
        y = pow(rhs, (pm.Q + 1) // 4, pm.Q)
        if y * y % pm.Q != rhs:
            continue
        if rng.next() & 1:
            y = pm.Q - y
        out[k] = np.frombuffer(pm.g1_aff_bytes((x, y)), dtype=np.uint8)
        k += 1
    return out


def test_msm_g1_random_curve_points(ctx):
    # points with no relation to each other (not consecutive multiples of G), uniform and witness-like scalars
    n = 6000
    bases = _hash_style_points(n)
    _check_msm(ctx, 0, bases, np_scalars(301, n, "uniform"))
    _check_msm(ctx, 0, bases, np_scalars(302, n, "witness"))
    # and against the affine big-int model on a prefix (independent of the C oracle)
    m = 40
    sc = np_scalars(303, m, "full256")
    _, got = ctx.msm(0, bases[:m], sc)
    pts = [pm.g1_aff_from_bytes(bytes(bases[i])) for i in range(m)]
    ks = [int.from_bytes(bytes(sc[i]), "little") for i in range(m)]
    assert got == pm.g1_aff_bytes(pm.ec_msm(pm.Fq1Ops, pts, ks))


@pytest.mark.parametrize("pattern", ["dup_first", "neg_first", "dup_run", "zero_first", "mixed"])
def test_msm_buckets_with_duplicates_and_negations(ctx, pattern):
    # every point gets the SAME scalar, so each window has one bucket holding all points in index order: the first add of
    # the bucket segment is affine + affine (P + P -> doubling, P + (-P) -> infinity, (0,0) rows), later ones mixed adds
    g = ol.gen_points(0, 0, 8)
    neg = g.copy()
    for i in range(8):
        x, y = pm.g1_aff_from_bytes(bytes(g[i]))
        neg[i] = np.frombuffer(pm.g1_aff_bytes((x, pm.Q - y)), dtype=np.uint8)
    zero = np.zeros(64, dtype=np.uint8)
    rows = {
        "dup_first": [g[0], g[0], g[1], g[2]],
        "neg_first": [g[0], neg[0], g[1], neg[1], g[3]],
        "dup_run": [g[2]] * 7 + [neg[2]] * 3,
        "zero_first": [zero, g[1], zero, zero, g[1], neg[1], g[1]],
        "mixed": [g[0], neg[0], g[0], g[0], zero, g[1], g[1], neg[3], g[3], g[3], g[3], g[5]],
    }[pattern]
    bases = np.stack(rows * 9)               # > 32 entries: more than one segment per bucket as well
    n = bases.shape[0]
    for kind in ("same", "ones"):
        _check_msm(ctx, 0, bases, np_scalars(77, n, kind))


# ---------------------------------------------------------------- fixed-base MSM (precomputed window tables)
def _fixed_base_msm(ctx, d_table, scalars, n):
    import k16
    d_s = ctx.to_device(scalars)
    ctx.msm_enqueue_fixed_base(k16.G1, d_table, d_s, n)
    xyzz, aff = ctx.msm_finish(k16.G1)
    d_s.free()
    return xyzz, aff


@pytest.mark.parametrize("n,want_c", [((1 << 13) + 37, 12), ((1 << 15) + 1, 14), (1 << 17, 16), ((1 << 19) + 3, 18)])
def test_msm_fixed_base_tables_agree_with_oracle(ctx, n, want_c):
    import k16
    bases = ol.gen_points(0, 5, n)
    bases[3] = 0                                  # a (0,0) row stays (0,0) in every window table
    bases[9] = bases[8]                           # duplicates
    d_b = ctx.to_device(bases)
    d_t, c = ctx.fixed_base_prepare(k16.G1, d_b, n)
    assert c == want_c
    kinds = ["uniform", "full256", "witness", "ones", "topwindow", "same"] if n < (1 << 18) else ["uniform", "full256"]
    for kind in kinds:
        sc = np_scalars(400 + want_c, n, kind)
        _, want = ol.msm(0, bases, sc, nthreads=8)
        assert _fixed_base_msm(ctx, d_t, sc, n)[1] == want, kind
    # the ordinary path on the same inputs, then the fixed-base one again: the two modes share the lane's workspace
    sc = np_scalars(17, n, "uniform")
    _, want = ol.msm(0, bases, sc, nthreads=8)
    assert ctx.msm(0, bases, sc)[1] == want
    assert _fixed_base_msm(ctx, d_t, sc, n)[1] == want
    d_t.free()
    d_b.free()


def test_msm_fixed_base_not_available_for_small_tables(ctx):
    import k16
    bases = ol.gen_points(0, 0, 1000)
    d_b = ctx.to_device(bases)
    d_t, c = ctx.fixed_base_prepare(k16.G1, d_b, 1000)
    assert d_t is None and c == 0
    with pytest.raises(k16.K16Error):
        ctx.msm_enqueue_fixed_base(k16.G1, d_b, d_b, 1000)
    d_b.free()


def test_msm_fixed_base_2p21_closed_form(ctx):
    """The Keyless H-MSM shape: n = 2^21, c = 20 (13 tables, 1.7 GB): sum_i s_i * (i+1)G against (sum_i s_i (i+1)) * G."""
    import k16
    n = 1 << 21
    d_b = ctx.synth_points(k16.G1, 0, n)
    d_t, c = ctx.fixed_base_prepare(k16.G1, d_b, n)
    assert c == 20
    sc = np_scalars(99, n, "uniform")
    xyzz, _ = _fixed_base_msm(ctx, d_t, sc, n)
    vals = sc.view("<u8").reshape(n, 4).astype(object)
    ints = vals[:, 0] + (vals[:, 1] << 64) + (vals[:, 2] << 128) + (vals[:, 3] << 192)
    k = int(sum(int(s) * (i + 1) for i, s in enumerate(ints)) % pm.R)
    want = ol.mul_scalar(0, ol.generator(0), pm.limbs(k))
    assert ol.pt_eq(0, xyzz, want)
    d_t.free()
    d_b.free()


# ---------------------------------------------------------------- C-ABI misuse: errors, never crashes or silent results
def test_c_abi_error_paths(ctx, tmp_path, toy_paths):
    import ctypes as C
    import k16
    L = ctx.L
    h = ctx.h
    # finish with nothing in flight
    out = (C.c_uint8 * 128)()
    assert L.k16_msm_finish(h, out, None) == -3                                                   # K16_ERR_ARG
    # more MSMs in flight than staging slots: the 9th enqueue is refused, the first eight still complete correctly
    n = 64
    bases = ol.gen_points(0, 0, n)
    sc = np_scalars(5, n, "uniform")
    d_b, d_s = ctx.to_device(bases), ctx.to_device(sc)
    for _ in range(8):
        ctx.msm_enqueue(k16.G1, d_b, d_s, n)
    with pytest.raises(k16.K16Error):
        ctx.msm_enqueue(k16.G1, d_b, d_s, n)
    _, want = ol.msm(0, bases, sc)
    for _ in range(8):
        assert ctx.msm_finish(k16.G1)[1] == want
    # bad group / lane / window size
    assert L.k16_msm_enqueue(h, 7, d_b.ptr, d_s.ptr, n) < 0
    assert L.k16_msm_set_lane(h, 99) < 0
    assert L.k16_msm_set_window_bits(h, 40) < 0
    # NTT: size not a power of two, size above the table, domain above the 2-adicity of r
    a = np.zeros((8, 32), dtype=np.uint8)
    d_a = ctx.to_device(a)
    assert L.k16_ntt(h, d_a.ptr, 6, 8, 0) < 0
    assert L.k16_ntt(h, d_a.ptr, 8, 4, 0) < 0
    assert L.k16_ntt(h, d_a.ptr, 8, 1 << 29, 0) < 0
    # prover: missing file, truncated zkey, witness shorter than the circuit, blinding scalar >= r
    zkey, wtns, _ = toy_paths
    pp = C.c_void_p()
    assert L.k16_prover_create(h, str(tmp_path / "missing.zkey").encode(), C.byref(pp)) == -4     # K16_ERR_IO
    trunc = tmp_path / "trunc.zkey"
    trunc.write_bytes(open(zkey, "rb").read()[:200])
    assert L.k16_prover_create(h, str(trunc).encode(), C.byref(pp)) in (-5, -4)                   # K16_ERR_FORMAT
    pr = k16.Prover(ctx, zkey)
    nv = pr.info()["n_vars"]
    short = np.zeros((nv - 1, 32), dtype=np.uint8)
    with pytest.raises(k16.K16Error):
        pr.prove_mem(short)
    w = np.frombuffer(open(wtns, "rb").read()[-nv * 32:], dtype=np.uint8).reshape(nv, 32).copy()
    big = (pm.R).to_bytes(32, "little")                                                           # r itself: not < r
    with pytest.raises(k16.K16Error):
        pr.prove_mem(w, r=big, s=bytes(32))
    js = pr.prove_mem(w, r=bytes(32), s=bytes(32))                                                 # and the good call still works
    assert json.loads(js)["protocol"] == "groth16"
    pr.close()
    for b in (d_b, d_s, d_a):
        b.free()


def test_msm_g1_above_2p24_is_chunked(ctx):
    """n > 2^24 through k16_msm: chunks of 2^24 on two lanes + EC-add fold (the single-GPU leg of BASELINE config 5).
    Bases (i+1)G from the device generator, scalars i+1 -> closed form sum (i+1)^2 * G."""
    import k16
    n = (1 << 24) + (1 << 20) + 5
    S = np.zeros((n, 32), dtype=np.uint8)
    S[:, :4] = np.arange(1, n + 1, dtype=np.uint32).view(np.uint8).reshape(n, 4)
    d_s = ctx.to_device(S)
    del S
    d_b = ctx.synth_points(k16.G1, 0, n)
    x, _ = ctx.msm_device(k16.G1, d_b, d_s, n)
    total = n * (n + 1) * (2 * n + 1) // 6
    want = ol.mul_scalar(0, ol.generator(0), pm.limbs(total % pm.R))
    assert ol.pt_eq(0, x, want)
    d_b.free()
    d_s.free()


# ---------------------------------------------------------------- the arithmetic the HOT kernels run (radix-2^29 field, Eng9 / Eng2n)
@pytest.mark.parametrize("field,sel", [(0, "FQ9"), (1, "FR9")])
def test_field_ops_hot_path_representation(ctx, field, sel):
    """k16_field_op_vec with the K16_FQ9 / K16_FR9 selectors: the same canonical inputs and outputs as test_field_ops, but
    the operation runs on the unsaturated radix-2^29 field the MSM / NTT kernels use (fmul9, fsqr9, fadd9, fsub9 and the
    conversions); operands are also moved up to the bounds the point formulas document (value + k*p)."""
    import k16
    p = pm.Q if field == 0 else pm.R
    rng = pm.SplitMix64(177 + field)
    n = 4096
    a, b = rand_fe_array(rng, p, n), rand_fe_array(rng, p, n)
    b[:8] = a[:8][::-1]
    selv = getattr(k16, sel)
    for op in range(7):
        want = ol.field_op_vec(field, op, a, b)
        # (ka, kb): multiples of the modulus added to the operands; products must stay within (2+ka)(2+kb) <= 128
        for ka, kb in ((0, 0), (6, 0), (0, 6), (6, 6), (2, 5)):
            got = ctx.field_op_vec(selv, k16.op_bound(op, ka, kb), a, b)
            assert np.array_equal(got, want), (sel, op, ka, kb)


@pytest.mark.parametrize("sel", ["FQ2N", "FQ2H"])
def test_fq2_ops_hot_path_representation(ctx, sel):
    """Fq2n (Fq2 over the radix-2^29 field: fmul9_sum2 products, fred9 partial reductions) against the oracle's Fq2
    (f2field.cpp:94-176), including the reference's own KAT (2,2)*(3,3) = (0,12) (alt_bn128_test.cpp:12-30)."""
    import k16
    rng = pm.SplitMix64(909)
    n = 2048
    a = np.concatenate([rand_fe_array(rng, pm.Q, n), rand_fe_array(rng, pm.Q, n)], axis=1)   # (n, 8): a | b
    b = np.concatenate([rand_fe_array(rng, pm.Q, n), rand_fe_array(rng, pm.Q, n)], axis=1)
    mont = lambda v: np.frombuffer(pm.limbs(v * pm.MONT % pm.Q), dtype=np.uint64)
    a[0] = np.concatenate([mont(2), mont(2)])
    b[0] = np.concatenate([mont(3), mont(3)])
    a[1], b[1] = 0, b[5]                   # 0 * y, and x - x, x + (-x) below
    b[2] = a[2]
    selv = getattr(k16, sel)     # FQ2H: the same values with the two components on a lane pair (bn254_fq2pair.h, round 5)
    for op in (k16.OP_ADD, k16.OP_SUB, k16.OP_NEG, k16.OP_MUL, k16.OP_SQR):
        got = ctx.field_op_vec(selv, op, a, b)
        for i in range(n):
            want = ol.fq2_op(op, a[i].tobytes(), b[i].tobytes())
            assert got[i].tobytes() == want, (op, i)
    prod = ctx.field_op_vec(selv, k16.OP_MUL, a[:1], b[:1])[0]
    assert prod[:4].tobytes() == bytes(32) and prod[4:].tobytes() == mont(12).tobytes()


@pytest.mark.parametrize("sel,group", [("G1_ENG9", 0), ("G2_ENG2N", 1), ("G2_PAIR", 1)])
def test_point_ops_hot_path_formulas(ctx, sel, group):
    """padd9 / padd_mixed9 / pdbl9 (G1) and the Fq2n instantiation of the XYZZ templates (G2) -- the formulas the bucket
    accumulation, fold and reduction kernels execute -- against the oracle, XYZZ representation for every branch of
    curve.cpp:91-250 (inf + P, P + inf, P + P -> dbl, P + (-P) -> inf, (0,0) rows), with G1 operands also at the documented
    bounds X < 8p, Y < 4p."""
    import k16
    selv = getattr(k16, sel)
    n = 64
    aff = ol.gen_points(group, 3, n)
    xb = k16.XYZZ_BYTES[group]
    g = ol.generator(group)
    proj = np.zeros((n, xb), dtype=np.uint8)
    other = np.zeros((n, xb), dtype=np.uint8)
    for i in range(n):
        proj[i] = np.frombuffer(ol.mul_scalar(group, g, pm.limbs(i + 4)), dtype=np.uint8)
        other[i] = np.frombuffer(ol.mul_scalar(group, g, pm.limbs(1000 + 7 * i)), dtype=np.uint8)
    inf = np.frombuffer(ol.mul_scalar(group, g, pm.limbs(0)), dtype=np.uint8)
    p1, p2x, p2a = other.copy(), proj.copy(), aff.copy()
    p1[0] = inf
    p2x[1] = inf
    p2a[1] = 0
    p1[2] = proj[2]
    p1[3] = np.frombuffer(ol.pt_op(group, ol.PT_NEG, bytes(proj[3])), dtype=np.uint8)
    p1[4] = inf
    p2x[4] = inf
    p2a[4] = 0
    omap = {k16.PT_ADD: ol.PT_ADD, k16.PT_MADD: ol.PT_MADD, k16.PT_DBL: ol.PT_DBL}
    for op, p2 in ((k16.PT_ADD, p2x), (k16.PT_MADD, p2a), (k16.PT_DBL, None)):
        want = [ol.pt_op(group, omap[op], bytes(p1[i]), bytes(p2[i]) if p2 is not None else None) for i in range(n)]
        bounds = [(0, 0)]
        if group == 0:
            bounds += [(2, 0), (1, 0)] + ([(2, 2), (0, 2)] if op == k16.PT_ADD else [])
        for ka, kb in bounds:
            got = ctx.point_op_vec(selv, k16.op_bound(op, ka, kb), p1, p2)
            for i in range(n):
                if i == 3 and op != k16.PT_DBL:
                    # P + (-P): zz3 = 0 by the general formula; x3, y3 are whatever it leaves (equal mod p)
                    assert ol.pt_eq(group, bytes(got[i]), want[i]) and bytes(got[i][xb // 2:]) == bytes(xb // 2), (sel, op, i)
                assert bytes(got[i]) == want[i], (sel, op, ka, kb, i)
    if group == 0:
        # the bucket accumulation's own addition (acc9_madd, round 5): W = +-Y with a flag, the entry's sign inside the
        # addition, lazy subtractions -- every branch again, both signs, both flag values, accumulator at X < 8p, Y < 4p
        def neg_aff(row):
            y = int.from_bytes(bytes(row[32:]), "little")
            out = row.copy()
            if y:
                out[32:] = np.frombuffer((pm.Q - y).to_bytes(32, "little"), dtype=np.uint8)
            return out
        p2n = np.stack([neg_aff(r) for r in p2a])
        for sign in (0, 1):
            # result = p1 + (sign ? -row : row); rows 2 / 3 must stay P + P / P + (-P): hand the kernel the row whose signed
            # value is the affine point of the table above
            rows = p2n if sign else p2a
            want = [ol.pt_op(group, ol.PT_MADD, bytes(p1[i]), bytes(p2a[i])) for i in range(n)]
            for flag in (0, 1):
                for ka in (0, 1, 2):
                    got = ctx.point_op_vec(selv, k16.op_bound(k16.PT_MADD_ACC, ka, sign | (flag << 1)), p1, rows)
                    for i in range(n):
                        if i == 3:
                            assert ol.pt_eq(group, bytes(got[i]), want[i]) and bytes(got[i][xb // 2:]) == bytes(xb // 2), (sign, flag, ka)
                        assert bytes(got[i]) == want[i], ("acc9_madd", sign, flag, ka, i)


# ---------------------------------------------------------------- full-size parity (BASELINE config 3 at scale 1.0)
def test_keyless_shape_proof_full_size_bit_exact(ctx, tmp_path):
    """The Keyless SHAPE at its stated size -- nVars 1,343,588, nPublic 1, N = 2^21, 8.3 M coefficients, B1/B2 half (0,0),
    witness 90 % bits / 8 % bytes / 2 % full width -- on a synthetic key: proof JSON and all 2^21 H scalars byte-equal to
    the CPU oracle's (RS/groth16.cpp:41-360).  This is the configuration whose code paths no smaller test reaches: NTT
    passes 1-10 / 11-16 / 17-21, the c = 20 fixed-base H MSM with 13 window tables, a ~600 k-point witness bucket
    through the giant fold, B2 with 1.34 M G2 points."""
    import bench
    import k16
    n_vars, N, n_coefs = bench.KEYLESS["n_vars"], bench.KEYLESS["domain"], bench.KEYLESS["n_coefs"]
    zk = str(tmp_path / "keyless_shape.zkey")
    wt = str(tmp_path / "keyless_shape.wtns")
    with open(zk, "wb") as f:
        f.write(bench.synth_zkey_bytes(ctx, k16, n_vars, 1, N, n_coefs))
    w = bench.synth_witness(n_vars, 100)
    bench.write_wtns(wt, w)
    r, s = pm.limbs(pm.SplitMix64(177).below(pm.R)), pm.limbs(pm.SplitMix64(178).below(pm.R))
    p = k16.Prover(ctx, zk)
    assert p.info() == dict(n_vars=n_vars, n_public=1, domain_size=N, n_coefs=n_coefs)
    got = p.prove_mem(w, r, s)
    h_gpu = p.last_h()
    got_file = p.prove_file(wt, r, s)
    want, h_ref = ol.prove_files(zk, wt, r, s, nthreads=os.cpu_count() or 8, want_h=True)
    assert np.array_equal(h_gpu, h_ref)
    assert got == want
    assert got_file == want
    # and a second witness on the warm prover (workspaces, sort reuse across proofs)
    w2 = bench.synth_witness(n_vars, 101)
    bench.write_wtns(wt, w2)
    assert p.prove_mem(w2, r, s) == ol.prove_files(zk, wt, r, s, nthreads=os.cpu_count() or 8)
    p.close()


def test_ntt_forward_and_inverse_2p21_vs_oracle(ctx):
    """The Keyless domain, both directions, compared with the oracle directly (not a round trip): FFT::fft / ::ifft
    (RS/fft.cpp:192-246) with the 2^22 root table the prover builds."""
    n = 1 << 21
    rs = np.random.RandomState(13)
    a = rs.randint(0, 2 ** 63, size=(n, 4)).astype(np.uint64)
    a[:, 3] &= (1 << 60) - 1     # < r
    for inverse in (False, True):
        got = ctx.ntt(a, max_domain=2 * n, inverse=inverse)
        want = ol.ntt(a, max_domain=2 * n, inverse=inverse)
        assert np.array_equal(got, want), inverse


def test_ntt_first_stage_pair_with_representatives_above_r(ctx):
    """Regression: the twiddle-free first two stages of a transform subtract a2 = x2 + x3, which is not fresh from a
    multiplication.  The kernels keep values as representatives in [0, 2r); when x2 and x3 are both >= r and x0 + x1 is
    small, a0 - a2 + 2r went negative (the offset has to be 4r there).  Rare with random data (about one proof in 10^3 at
    2^21), so the inputs are built for it: words whose radix-2^29 Montgomery representative (bn254_fq9.h fmul9_t:
    (X * K_IN + m r) / 2^261) is >= r at the third and fourth element of a quad, zeros at the first two."""
    logn = 12
    n = 1 << logn
    R9, k_in = 1 << 261, pow(2, 266, pm.R)
    rinv = pow(pm.R, -1, R9)

    def rep(x):
        t = x * k_in
        return (t + ((-t * rinv) % R9) * pm.R) >> 261

    rng = pm.SplitMix64(4242)
    above = []
    while len(above) < 6:
        x = rng.below(pm.R)
        if rep(x) >= pm.R:
            above.append(x)
    rs = np.random.RandomState(17)
    a = rs.randint(0, 2 ** 63, size=(n, 4)).astype(np.uint64)
    a[:, 3] &= (1 << 60) - 1
    brev = lambda v: int(format(v, "0%db" % logn)[::-1], 2)
    for k, q in enumerate((0, 3, 700)):
        pos = [brev(4 * q + j) for j in range(4)]      # the quad's elements in natural order
        a[pos[0]] = 0
        a[pos[1]] = 0
        for j in (2, 3):
            a[pos[j]] = np.frombuffer(pm.limbs(above[2 * k + j - 2]), dtype=np.uint64)
    for inverse in (False, True):
        assert np.array_equal(ctx.ntt(a, max_domain=2 * n, inverse=inverse), ol.ntt(a, max_domain=2 * n, inverse=inverse)), inverse


@pytest.mark.parametrize("kind", ["uniform", "witness"])
def test_msm_g2_2p20_vs_oracle(ctx, kind):
    """G2 at the size of the prover's B2 MSM (RS/groth16.cpp:100-102), half of the rows (0,0) as in a real key."""
    n = 1 << 20
    bases = ol.gen_points(1, 7, n)
    rs = np.random.RandomState(5)
    bases[rs.rand(n) < 0.5] = 0
    scalars = np_scalars(61, n, kind)
    if kind == "witness":
        ctx.set_window_bits(13)          # what the prover uses for its witness MSMs
    try:
        _check_msm(ctx, 1, bases, scalars, threads=os.cpu_count() or 8)
    finally:
        ctx.set_window_bits(0)


def test_msm_more_giant_buckets_than_the_giant_list_holds(monkeypatch):
    """300 buckets with > 2048 segments each (K16_SEG=1: one entry per segment): the giant list holds 256, the rest must
    be folded through the medium path (k_classify) -- before the fix they kept only their first partial."""
    vals, copies = 300, 2100
    n = vals * copies
    bases = ol.gen_points(0, 0, 4096)[np.arange(n) % 4096]
    scalars = np.zeros((n, 32), dtype=np.uint8)
    v = (np.arange(n) % vals + 1).astype(np.uint16)
    scalars[:, :2] = v.view(np.uint8).reshape(n, 2)
    _, want = ol.msm(0, bases, scalars, nthreads=os.cpu_count() or 8)
    monkeypatch.setenv("K16_SEG", "1")
    import k16
    c2 = k16.Context(0)          # the switch is read when a context is created
    try:
        _, got = c2.msm(0, bases, scalars)
    finally:
        c2.close()
    assert got == want


def test_prover_failure_leaves_no_stale_msms(tmp_path):
    """A prove that fails after its witness MSMs were enqueued (K16_FAULT_INJECT, a device fault or a std::bad_alloc) must
    drain them: the queue is empty afterwards and the next proof on the same prover is the right one (before the fix it
    was built from the failed proof's MSMs, or overflowed a G1 buffer with a G2 result).  The fault hooks exist only in the
    TESTING build of the library (libk16_testing.so), so the body runs in a child process that loads that build
    (tests/fault_inject_child.py); the same child checks that the production library ignores the variable."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(here), "keyless-zk-proofs_amd")
    for lib, mode in (("libk16_testing.so", "testing"), ("libk16.so", "production")):
        env = dict(os.environ, K16_LIB_PATH=os.path.join(pkg, lib))
        env.pop("K16_FAULT_INJECT", None)
        out = subprocess.run([sys.executable, os.path.join(here, "fault_inject_child.py"), mode, str(tmp_path)],
                             capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0 and "fault child OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_msm_finish_refuses_a_result_of_the_other_group(ctx):
    import k16
    n = 64
    d_b, d_s = ctx.to_device(ol.gen_points(1, 0, n)), ctx.to_device(np_scalars(5, n, "uniform"))
    ctx.msm_enqueue(k16.G2, d_b, d_s, n)
    with pytest.raises(k16.K16Error):
        ctx.msm_finish(k16.G1)
    assert ctx.msm_pending() == 1
    assert ctx.msm_finish(k16.G2)[1] == ol.msm(1, ol.gen_points(1, 0, n), np_scalars(5, n, "uniform"))[1]
    d_b.free()
    d_s.free()


def test_msm_with_hip_graphs_option(ctx):
    """K16_OPT_GRAPHS: sort and fold + reduction replayed as HIP graphs after their second use (per lane, shape and
    staging slot).  Results must not change, across repeated calls (eager, capture, replay) and lanes."""
    import k16
    n = 20000
    bases = ol.gen_points(0, 9, n)
    sc = [np_scalars(700 + i, n, "uniform") for i in range(2)]
    want = [ol.msm(0, bases, s, nthreads=8)[1] for s in sc]
    d_b = ctx.to_device(bases)
    d_s = [ctx.to_device(s) for s in sc]
    ctx.set_option(k16.OPT_GRAPHS, 1)
    try:
        for rep in range(5):
            for lane in (0, 1):
                ctx.set_lane(lane)
                for i in range(2):
                    ctx.msm_enqueue(k16.G1, d_b, d_s[i], n)
                for i in range(2):
                    assert ctx.msm_finish(k16.G1)[1] == want[i], (rep, lane, i)
    finally:
        ctx.set_option(k16.OPT_GRAPHS, 0)
        ctx.set_lane(0)
    for d in d_s + [d_b]:
        d.free()


def test_masked_sorts_never_replay_another_masks_graph(ctx):
    """a captured sort has its kernels' arguments baked in and its key names neither the zero-row mask nor the
    row indirection.  With K16_OPT_GRAPHS on, the SAME scalar array enqueued with mask A, with mask B, and with no mask
    must each give ITS sum (masked sorts are launched eagerly); a mask that hides rows whose points are not (0,0) makes the
    difference visible: every expected value below differs from the others."""
    import k16
    n = 1 << 16                      # the staged / partition sort
    bases = ol.gen_points(0, 21, n)
    sc = np_scalars(811, n, "uniform")
    d_b = ctx.to_device(bases)
    prep = ctx.bases_prepare(k16.G1, d_b, n)
    d_s = ctx.to_device(sc)

    def mask_of(hidden):
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        for i in hidden:
            bits[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
        return ctx.to_device(bits), hidden

    def want(hidden):
        b = bases.copy()
        b[list(hidden)] = 0          # a hidden row contributes nothing
        return ol.msm(0, b, sc, nthreads=8)[1]

    ma, mb = mask_of(range(0, 4000, 3)), mask_of(range(1, 9000, 7))
    w_a, w_b, w_none = want(ma[1]), want(mb[1]), want([])
    assert len({w_a, w_b, w_none}) == 3
    ctx.set_option(k16.OPT_GRAPHS, 1)
    try:
        for rep in range(4):         # eager, capture, replay, replay
            for m, w in ((None, w_none), (ma[0], w_a), (mb[0], w_b), (None, w_none), (mb[0], w_b), (ma[0], w_a)):
                if m is not None:
                    ctx.msm_set_zero_row_mask(m)
                ctx.msm_enqueue_prepared(k16.G1, prep, d_s, n)
                assert ctx.msm_finish(k16.G1)[1] == w, rep
    finally:
        ctx.set_option(k16.OPT_GRAPHS, 0)
    for d in (d_b, prep, d_s, ma[0], mb[0]):
        d.free()


def test_one_shot_requests_do_not_outlive_an_enqueue_that_returns_early(ctx):
    """a zero-row mask (or a sort-reuse request) set for the NEXT enqueue covers exactly one enqueue -- also one
    that returns early (n = 0, an argument error): the MSM after it must see every row."""
    import k16
    n = 1 << 16
    bases = ol.gen_points(0, 33, n)
    sc = np_scalars(812, n, "uniform")
    d_b = ctx.to_device(bases)
    prep = ctx.bases_prepare(k16.G1, d_b, n)
    d_s = ctx.to_device(sc)
    full = ol.msm(0, bases, sc, nthreads=8)[1]
    bits = np.zeros((n + 63) // 64, dtype=np.uint64)
    bits[:200] = np.uint64(0xFFFFFFFFFFFFFFFF)          # hides 12 800 real points
    d_m = ctx.to_device(bits)
    # (1) n = 0 consumes the mask
    ctx.msm_set_zero_row_mask(d_m)
    ctx.msm_enqueue_prepared(k16.G1, prep, d_s, 0)
    x0, _ = ctx.msm_finish(k16.G1)
    assert ol.pt_eq(0, x0, ol.mul_scalar(0, ol.generator(0), pm.limbs(0)))
    ctx.msm_enqueue_prepared(k16.G1, prep, d_s, n)
    assert ctx.msm_finish(k16.G1)[1] == full
    # (2) an enqueue refused for its arguments consumes it too
    ctx.msm_set_zero_row_mask(d_m)
    ctx.msm_sort_from_lane(3)                            # no sort on lane 3: the enqueue fails
    with pytest.raises(k16.K16Error):
        ctx.msm_enqueue_prepared(k16.G1, prep, d_s, n)
    ctx.msm_enqueue_prepared(k16.G1, prep, d_s, n)
    assert ctx.msm_finish(k16.G1)[1] == full
    for d in (d_b, prep, d_s, d_m):
        d.free()


@pytest.mark.parametrize("kind", ["uniform", "full256", "witness"])
def test_msm_g1_2p20_vs_oracle(ctx, kind):
    """BASELINE config 2 at its stated size against the oracle directly (not only through closed forms): 2^20 points,
    uniform scalars below r, arbitrary 256-bit scalars (top window + carry window in use), witness-like scalars."""
    n = 1 << 20
    bases = ol.gen_points(0, 3, n)
    bases[5] = 0
    bases[7] = bases[6]
    _check_msm(ctx, 0, bases, np_scalars(801, n, kind), threads=os.cpu_count() or 8)


def test_msm_fixed_base_2p21_vs_oracle(ctx):
    """The H MSM's configuration (n = 2^21, c = 20, 13 window tables) against the oracle itself."""
    import k16
    n = 1 << 21
    bases = ol.gen_points(0, 11, n)
    d_b = ctx.to_device(bases)
    d_t, c = ctx.fixed_base_prepare(k16.G1, d_b, n)
    assert c == 20
    sc = np_scalars(902, n, "uniform")
    _, want = ol.msm(0, bases, sc, nthreads=os.cpu_count() or 8)
    assert _fixed_base_msm(ctx, d_t, sc, n)[1] == want
    d_t.free()
    d_b.free()


@pytest.mark.parametrize("log2n", [17, 19, 20])
def test_ntt_large_sizes_vs_oracle(ctx, log2n):
    """Sizes between the small-size sweep (<= 2^16) and the Keyless domain (2^21): every pass split (10 + 7, 10 + 6 + 3,
    10 + 6 + 4 stages) against the oracle, forward and inverse, with the table of twice the size."""
    n = 1 << log2n
    rs = np.random.RandomState(log2n)
    a = rs.randint(0, 2 ** 63, size=(n, 4)).astype(np.uint64)
    a[:, 3] &= (1 << 60) - 1
    for inverse in (False, True):
        assert np.array_equal(ctx.ntt(a, max_domain=2 * n, inverse=inverse), ol.ntt(a, max_domain=2 * n, inverse=inverse))


@pytest.mark.parametrize("n_vars", [70000, 65539])
def test_compact_witness_upload_edge_cases(tmp_path, monkeypatch, n_vars):
    """The witness crosses PCIe in compact form for circuits of >= 2^16 wires (prover.hip WitnessPacker / k_wtns_expand_*: one
    byte per wire + per-host-thread lists of the wide values; a list that overflows makes the proof fall back to the plain
    copy).  What the Keyless-shape witnesses never reach: values exactly at the byte boundary and with only a high word set,
    wide values at the first / last wire and across the host threads' range boundaries, a range holding exactly its list
    capacity, one more than that (fallback), and a witness of nothing but wide values.  Every proof must equal the oracle's
    (RS/groth16.cpp:41-360 reads the same 32-byte values whatever their size).  65539 wires: the byte array is read a dword per
    lane, 256 wires per wave step (k_wtns_expand_narrow) -- neither divides the wire count, the last dword is partly padding."""
    import k16
    import zkey_builder as zb
    monkeypatch.setenv("K16_HOST_THREADS", "4")      # four ranges: the capacities below are exact for that split
    c = k16.Context(0)
    try:
        N, n_coefs, T = 1 << 17, 200000, 4
        zk, wt = str(tmp_path / "c.zkey"), str(tmp_path / "c.wtns")
        zb.build_zkey(zk, n_vars, 1, N, n_coefs, seed=31)
        p = k16.Prover(c, zk)
        r, s = pm.limbs(pm.SplitMix64(91).below(pm.R)), pm.limbs(pm.SplitMix64(92).below(pm.R))
        rs = np.random.RandomState(5)
        cap = (n_vars // T) // 4 + 64

        def wide(k):
            return np.frombuffer(pm.limbs(pm.SplitMix64(1000 + k).below(pm.R)), dtype=np.uint8)

        def narrow_base():
            w = np.zeros((n_vars, 32), dtype=np.uint8)
            w[:, 0] = rs.randint(0, 256, size=n_vars)
            return w

        cases = {}
        w = narrow_base()
        w[100:400, 0] = 255
        cases["narrow_only"] = w
        w = narrow_base()
        edges = [1, 2, n_vars - 1] + [n_vars * t // T + d for t in range(1, T) for d in (-1, 0, 1)]
        vals = [1 << 8, 1 << 32, 1 << 64, 1 << 128, 1 << 192, pm.R - 1, 255 + (1 << 200), (1 << 8) + 7]
        for k, i in enumerate(edges):
            w[i] = np.frombuffer(pm.limbs(vals[k % len(vals)]), dtype=np.uint8)
        cases["boundary_values"] = w
        for name, count in (("range_exactly_full", cap), ("range_overflows", cap + 1)):
            w = narrow_base()
            for k in range(count):
                w[1 + k] = wide(k)
            cases[name] = w
        w = np.zeros((n_vars, 32), dtype=np.uint8)
        f = rs.randint(0, 256, size=(n_vars, 32), dtype=np.uint8)
        f[:, 31] &= 0x1F
        f[:, 1] |= 1                                   # every value >= 256
        cases["all_wide"] = f
        for name, w in cases.items():
            w[0] = 0
            w[0, 0] = 1
            with open(wt, "wb") as fh:
                import struct
                sec1 = struct.pack("<I", 32) + pm.limbs(pm.R) + struct.pack("<I", n_vars)
                fh.write(b"wtns" + struct.pack("<II", 2, 2) + zb._section(1, sec1) + zb._section(2, w.tobytes()))
            want, h_ref = ol.prove_files(zk, wt, r, s, nthreads=8, want_h=True)
            got = p.prove_mem(w, r, s)
            assert np.array_equal(p.last_h(), h_ref), name
            assert got == want, name
            assert p.prove_file(wt, r, s) == want, name
        p.close()
    finally:
        c.close()


def _fill_compact(prover, w):
    """what a witness calculator would write: one byte per wire (0 for values >= 256) + the list of the wide wires"""
    narrow, idx, val = prover.compact_buffers()
    wide = np.flatnonzero(w[:, 1:].any(axis=1))
    narrow[:] = w[:, 0]
    narrow[wide] = 0
    assert len(wide) <= len(idx)
    idx[:len(wide)] = wide[::-1]                      # any order
    val[:len(wide)] = w[wide[::-1]]
    return len(wide)


def test_compact_witness_hand_off(tmp_path):
    """k16_prover_compact_buffers / k16_prover_prove_compact (include/k16.h): the caller writes the witness in the
    form the device expands -- one byte per wire + the list of the wide values -- into the prover's pinned buffers, and the
    proof skips the host scan.  Same proof bytes and H scalars as k16_prover_prove_mem and as the oracle (RS/groth16.cpp:41-360)
    for the same witness; a bad list entry is skipped on the device and fails the proof; a prover that uploads plainly refuses."""
    import k16
    import zkey_builder as zb
    c = k16.Context(0)
    try:
        n_vars, N, n_coefs = 70001, 1 << 17, 200000
        zk, wt = str(tmp_path / "c.zkey"), str(tmp_path / "c.wtns")
        zb.build_zkey(zk, n_vars, 1, N, n_coefs, seed=37)
        p = k16.Prover(c, zk)
        r, s = pm.limbs(pm.SplitMix64(191).below(pm.R)), pm.limbs(pm.SplitMix64(192).below(pm.R))
        rs = np.random.RandomState(9)
        for n_wide_target in (0, 1, 1500):
            w = np.zeros((n_vars, 32), dtype=np.uint8)
            w[:, 0] = rs.randint(0, 256, size=n_vars)
            for k, i in enumerate(rs.choice(np.arange(1, n_vars), size=n_wide_target, replace=False)):
                w[i] = np.frombuffer(pm.limbs(pm.SplitMix64(5000 + k).below(pm.R) | 256), dtype=np.uint8) if k % 7 else \
                    np.frombuffer(pm.limbs(256 + k), dtype=np.uint8)
            w[0] = 0
            w[0, 0] = 1
            import struct
            with open(wt, "wb") as fh:
                sec1 = struct.pack("<I", 32) + pm.limbs(pm.R) + struct.pack("<I", n_vars)
                fh.write(b"wtns" + struct.pack("<II", 2, 2) + zb._section(1, sec1) + zb._section(2, w.tobytes()))
            want, h_ref = ol.prove_files(zk, wt, r, s, nthreads=8, want_h=True)
            n_wide = _fill_compact(p, w)
            assert n_wide == n_wide_target
            got = p.prove_compact(n_wide, r, s)
            assert np.array_equal(p.last_h(), h_ref), n_wide_target
            assert got == want, n_wide_target
            assert p.prove_mem(w, r, s) == want          # (overwrites the buffers: refill before the next compact proof)
        # a bad list entry never reaches the witness array: the proof fails when its device work has been joined
        narrow, idx, val = p.compact_buffers()
        n_wide = _fill_compact(p, w)
        idx[3] = n_vars                                  # wire number out of range
        with pytest.raises(k16.K16Error) as e:
            p.prove_compact(n_wide, r, s)
        assert e.value.rc == -5
        n_wide = _fill_compact(p, w)
        narrow[idx[5]] = 7                               # a listed wire with a byte of its own
        with pytest.raises(k16.K16Error) as e:
            p.prove_compact(n_wide, r, s)
        assert e.value.rc == -5
        n_wide = _fill_compact(p, w)
        idx[n_wide] = idx[7]                             # a wire listed twice (round 6: claimed atomically on the device)
        val[n_wide] = val[7]
        with pytest.raises(k16.K16Error) as e:
            p.prove_compact(n_wide + 1, r, s)
        assert e.value.rc == -5 and "twice" in str(e.value)
        with pytest.raises(k16.K16Error) as e:
            p.prove_compact(len(idx) + 1, r, s)          # more than the list holds
        assert e.value.rc == -3
        n_wide = _fill_compact(p, w)
        assert p.prove_compact(n_wide, r, s) == want     # and the prover is fine afterwards
        # a bad compact call directly followed by a witness whose wide values overflow the packer's lists (the
        # plain-copy branch, which never runs the list kernel): the flag of the failed call must not fail this proof
        n_wide = _fill_compact(p, w)
        idx[2] = n_vars + 9
        with pytest.raises(k16.K16Error):
            p.prove_compact(n_wide, r, s)
        wfull = np.zeros((n_vars, 32), dtype=np.uint8)
        wfull[:, :31] = rs.randint(0, 256, size=(n_vars, 31))
        wfull[:, 1] |= 1                                 # every wire >= 256: more wide values than the lists hold
        wfull[0] = 0
        wfull[0, 0] = 1
        with open(wt, "wb") as fh:
            fh.write(b"wtns" + struct.pack("<II", 2, 2) + zb._section(1, sec1) + zb._section(2, wfull.tobytes()))
        assert p.prove_mem(wfull, r, s) == ol.prove_files(zk, wt, r, s, nthreads=8)
        p.close()
        # a small circuit uploads plainly: no compact buffers
        zb.build_zkey(zk, 300, 1, 512, 900, seed=38)
        p2 = k16.Prover(c, zk)
        with pytest.raises(k16.K16Error) as e:
            p2.compact_buffers()
        assert e.value.rc == -3
        p2.close()
    finally:
        c.close()


def test_context_behind_placeholder_streams_gives_the_same_results(toy_paths):
    """k16_ctx_create_ex (include/k16.h, round 6): a context created behind 1 .. 7 placeholder streams lands on other hardware queues --
    a placement choice: MSM results and proof bytes equal the oracle's (RS/multiexp.cpp:183-245, RS/groth16.cpp:41-360); an offset
    outside 0 .. 7 is refused."""
    import k16
    zkey, wtns, _ = toy_paths
    n = 3000
    bases = ol.gen_points(0, 4, n)
    scalars = np_scalars(616, n, "full256")
    want = ol.msm(0, bases, scalars, nthreads=4)[1]
    z = bytes(32)
    proof = ol.prove_files(zkey, wtns, z, z)
    for off in (1, 3, 7):
        c = k16.Context(0, stream_offset=off)
        try:
            assert c.msm(0, bases, scalars)[1] == want
            p = k16.Prover(c, zkey)
            assert p.prove_file(wtns, z, z) == proof
            p.close()
        finally:
            c.close()
    with pytest.raises(k16.K16Error) as e:
        k16.Context(0, stream_offset=8)
    assert e.value.rc == -3


def test_yielding_waits_give_the_same_results(tmp_path, toy_paths):
    """K16_OPT_YIELDING_WAITS (include/k16.h, round 6): the host waits of k16_msm_finish* and of the prove calls poll + sleep
    instead of spinning inside the runtime -- a scheduling choice of the HOST: MSM results and proof bytes equal the oracle's
    (RS/multiexp.cpp:183-245, RS/groth16.cpp:41-360) with it on, off again, and with four MSMs in flight."""
    import k16
    zkey, wtns, _ = toy_paths
    c = k16.Context(0)
    try:
        n = 5000
        bases = ol.gen_points(0, 2, n)
        scalars = np_scalars(515, n, "full256")
        want = ol.msm(0, bases, scalars, nthreads=4)[1]
        p = k16.Prover(c, zkey)
        z = bytes(32)
        proof = ol.prove_files(zkey, wtns, z, z)
        for on in (1, 0, 1):
            c.set_option(k16.OPT_YIELDING_WAITS, on)
            assert c.msm(0, bases, scalars)[1] == want
            assert p.prove_file(wtns, z, z) == proof
            d_b, d_s = c.to_device(bases), c.to_device(scalars)
            for lane in range(4):
                c.set_lane(lane)
                c.msm_enqueue(k16.G1, d_b, d_s, n)
            c.set_lane(0)
            for _ in range(4):
                assert c.msm_finish(k16.G1)[1] == want
            d_b.free()
            d_s.free()
        with pytest.raises(k16.K16Error):
            c.set_option(99, 1)
        p.close()
    finally:
        c.close()


@pytest.mark.parametrize("field,sel", [(0, "FQ9"), (1, "FR9")])
def test_lazy_limb_butterfly_forms(ctx, field, sel):
    """fadd9_lazy / fsub9_lazy4_t (bn254_fq9.h): the NTT double stage keeps the sums and differences of its first stage
    without carry propagation and feeds them straight into the next multiplication (ntt.hip).  (a + b) * b and
    (a - b) * b with a moved up to a + 28 p (the passes reach 32 r) against the oracle's canonical arithmetic."""
    import k16
    p = pm.Q if field == 0 else pm.R
    rng = pm.SplitMix64(4242 + field)
    n = 8192
    a, b = rand_fe_array(rng, p, n), rand_fe_array(rng, p, n)
    edge = [0, 1, p - 1, p - 2, (1 << 29) - 1, 1 << 29, (1 << 232) - 1, 1 << 232, (p >> 1), (p >> 1) + 1]
    for i, v in enumerate(edge):          # raw stored values (Montgomery form is just another field element here)
        a[i] = np.frombuffer(pm.limbs(v), dtype=np.uint64)
        b[len(edge) + i] = np.frombuffer(pm.limbs(v), dtype=np.uint64)
        a[2 * len(edge) + i] = b[2 * len(edge) + i] = np.frombuffer(pm.limbs(v), dtype=np.uint64)
    selv = getattr(k16, sel)
    add = ol.field_op_vec(field, k16.OP_ADD, a, b)
    sub = ol.field_op_vec(field, k16.OP_SUB, a, b)
    want_add = ol.field_op_vec(field, k16.OP_MUL, add, b)
    want_sub = ol.field_op_vec(field, k16.OP_MUL, sub, b)
    for ka in (0, 1, 7, 13, 14):
        assert np.array_equal(ctx.field_op_vec(selv, k16.op_bound(k16.OP_LAZY_ADDMUL, ka), a, b), want_add), (sel, ka)
        assert np.array_equal(ctx.field_op_vec(selv, k16.op_bound(k16.OP_LAZY_SUBMUL, ka), a, b), want_sub), (sel, ka)
