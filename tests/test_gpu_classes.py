"""-m gpu : the scalar-class MSM (k16_scalar_classes_*, k16_msm_enqueue_classified; csrc/msm_classes.hip and the
"Scalar-class MSM" part of csrc/msm_kernels.inc) through the C ABI against the CPU oracle's multiexp on the SAME bases and
scalars.  It replaces what the reference's zero-digit skip (multiexp.cpp:59-65) does for the four witness MSMs of
groth16.cpp:88-112, so the cases are the witness's: zeros, ones, bytes, field elements, class boundaries, (0,0) rows."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import pymodel as pm
from gpu_common import np_scalars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import k16
    c = k16.Context(0)
    yield c
    c.close()


def _ints_to_scalars(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32).copy()


def _classified(ctx, group, bases, scalars, use_mask=True, bound=-1, sets=1):
    """bases: (n, AFF) zkey format; scalars (n, 32).  Returns the affine result bytes of the classified MSM."""
    n = scalars.shape[0]
    d_b = ctx.to_device(bases)
    d_p = ctx.bases_prepare(group, d_b, n)
    d_s = ctx.to_device(scalars)
    cls = ctx.classes_create(max(n, 1), sets)
    try:
        mask = ctx.zero_row_mask(group, d_p, n) if (use_mask and n) else None
        ctx.classes_build(cls, d_s, n, [mask] + [None] * (sets - 1), bound)
        ctx.msm_enqueue_classified(group, d_p, cls, 0)
        _, aff = ctx.msm_finish(group)
        ctx.sync()
    finally:
        ctx.classes_destroy(cls)
        for d in (d_b, d_p, d_s):
            d.free()
        if mask is not None:
            mask.free()
    return aff


def _check(ctx, group, bases, scalars, **kw):
    got = _classified(ctx, group, bases, scalars, **kw)
    _, want = ol.msm(group, bases, scalars, nthreads=4)
    assert got == want


@pytest.mark.parametrize("group", [0, 1])
@pytest.mark.parametrize("kind", ["zeros", "ones", "witness", "uniform", "full256"])
def test_classified_msm_distributions(ctx, group, kind):
    n = 5000 if group == 0 else 1500
    _check(ctx, group, ol.gen_points(group, 5, n), np_scalars(11, n, kind))


@pytest.mark.parametrize("group", [0, 1])
def test_classified_msm_all_bytes(ctx, group):
    n = 6000 if group == 0 else 1200
    rs = np.random.RandomState(5)
    s = np.zeros((n, 32), dtype=np.uint8)
    s[:, 0] = rs.randint(2, 256, size=n)
    _check(ctx, group, ol.gen_points(group, 9, n), s)
    s[:, 0] = 255            # every wire in all eight lists
    _check(ctx, group, ol.gen_points(group, 9, n), s)


@pytest.mark.parametrize("group", [0, 1])
def test_classified_msm_class_boundaries(ctx, group):
    """values 0, 1, 2, 127, 128, 255 (narrow) and 256, 257, 2^16, 2^253, r-1, r, 2^256-1 (wide) side by side"""
    vals = [0, 1, 2, 3, 127, 128, 129, 254, 255, 256, 257, 511, 512, 65535, 65536, (1 << 64) - 1, 1 << 64, 1 << 128,
            1 << 253, pm.R - 1, pm.R, pm.R + 1, (1 << 256) - 1]
    vals = vals * 9
    n = len(vals)
    _check(ctx, group, ol.gen_points(group, 2, n), _ints_to_scalars(vals))


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4097])
def test_classified_msm_sizes_around_the_tiles(ctx, n):
    _check(ctx, 0, ol.gen_points(0, 1, n) if n else np.zeros((0, 64), np.uint8), np_scalars(n + 3, n, "witness"))


@pytest.mark.parametrize("group", [0, 1])
@pytest.mark.parametrize("use_mask", [True, False])
def test_classified_msm_zero_rows_and_duplicates(ctx, group, use_mask):
    """(0,0) rows under every class (skipped through the mask, or met by the additions without it), duplicate and negated
    rows next to each other in every class (P + P and P - P inside the chains and the trees)"""
    n = 3000 if group == 0 else 900
    bases = ol.gen_points(group, 21, n).copy()
    ab = bases.shape[1]
    rs = np.random.RandomState(8)
    s = np_scalars(31, n, "witness")
    zero = rs.rand(n) < 0.4
    bases[zero] = 0
    # duplicates: runs of the same point, and a point next to its negation
    for start in (10, 500, 501, 502, 777):
        bases[start:start + 8] = bases[start]
    for at in (200, 640):     # a point next to its negation, under a one and under a byte
        k = at % n
        bases[k] = ol.gen_points(group, 5000 + at, 1)[0]
        x = ol.mul_scalar(group, bytes(bases[k]), pm.limbs(1))
        bases[k + 1] = np.frombuffer(ol.pt_to_affine(group, ol.pt_op(group, ol.PT_NEG, x)), dtype=np.uint8)
        s[k:k + 2, :] = 0
        s[k:k + 2, 0] = 1 if at == 200 else 6
    s[5:40, :] = 0
    s[5:40, 0] = 1            # ones on top of duplicates
    s[500:520, :] = 0
    s[500:520, 0] = 7
    s[770:790] = np_scalars(2, 20, "uniform")
    assert ab in (64, 128)
    _check(ctx, group, bases, s, use_mask=use_mask)


def test_classified_msm_all_rows_zero(ctx):
    n = 1000
    _check(ctx, 0, np.zeros((n, 64), dtype=np.uint8), np_scalars(1, n, "witness"))


def test_classified_msm_shares_one_classification_between_tables_and_groups(ctx):
    """The prover's use: one build, four tables with three different (0,0) patterns on three lanes, results in enqueue
    order; C-style table with a zero prefix."""
    import k16
    n = 20000
    s = np_scalars(77, n, "witness")
    rs = np.random.RandomState(3)
    tabs = []
    for t, group in enumerate([0, 0, 1, 0]):
        b = ol.gen_points(group, 40 + t, n if group == 0 else 3000).copy()
        if group == 1:   # G2 table over the first 3000 wires only: the rest are (0,0) rows
            full = np.zeros((n, 128), dtype=np.uint8)
            full[:3000] = b
            b = full
        if t in (1, 2):
            z = rs.rand(n) < 0.5
            b[z] = 0
        if t == 3:
            b[:2] = 0
        tabs.append((group, b))
    d_s = ctx.to_device(s)
    d_tabs, masks = [], []
    for group, b in tabs:
        d_b = ctx.to_device(b)
        d_p = ctx.bases_prepare(group, d_b, n)
        d_b.free()
        d_tabs.append(d_p)
        masks.append(ctx.zero_row_mask(group, d_p, n))
    cls = ctx.classes_create(n, 4)
    n_wide = int((s[:, 1:].any(axis=1)).sum())
    ctx.classes_build(cls, d_s, n, masks, n_wide)          # exact bound from the host, no read-back
    counts = ctx.classes_counts(cls, 4)
    assert counts[32] == n_wide
    narrow = ~s[:, 1:].any(axis=1)
    for t, (group, b) in enumerate(tabs):
        nz = b.any(axis=1)
        for bit in range(8):
            assert counts[t * 8 + bit] == int((narrow & nz & (((s[:, 0] >> bit) & 1) == 1)).sum()), (t, bit)
    for lane, t in ((0, 0), (1, 3), (0, 1), (2, 2)):
        ctx.set_lane(lane)
        ctx.msm_enqueue_classified(tabs[t][0], d_tabs[t], cls, t)
    ctx.set_lane(0)
    for t in (0, 3, 1, 2):
        _, got = ctx.msm_finish(tabs[t][0])
        _, want = ol.msm(tabs[t][0], tabs[t][1], s, nthreads=4)
        assert got == want, t
    ctx.sync()
    ctx.classes_destroy(cls)
    d_s.free()
    for d in d_tabs + masks:
        d.free()


def test_classified_msm_bound_below_the_actual_count_fails_loudly(ctx):
    import k16
    n = 4000
    s = np_scalars(5, n, "witness")
    n_wide = int((s[:, 1:].any(axis=1)).sum())
    assert n_wide > 10
    bases = ol.gen_points(0, 6, n)
    with pytest.raises(k16.K16Error) as e:
        _classified(ctx, 0, bases, s, bound=n_wide - 3)
    assert e.value.rc == -3
    ctx.msm_abort_all()
    # a generous bound is fine (unused rows hold zero scalars)
    got = _classified(ctx, 0, bases, s, bound=n_wide + 100)
    assert got == ol.msm(0, bases, s, nthreads=4)[1]
    got = _classified(ctx, 0, bases, s, bound=n_wide)
    assert got == ol.msm(0, bases, s, nthreads=4)[1]


@pytest.mark.parametrize("group", [0, 1])
def test_classified_msm_witness_2p20_vs_oracle(ctx, group):
    """BASELINE config 3's witness MSM shape at full table size."""
    n = 1 << 20
    s = np_scalars(123 + group, n, "witness")
    d_b = ctx.synth_points(group, 0, n)
    bases = d_b.download(np.uint8, (n, 64 if group == 0 else 128)).copy()
    if group == 1:
        bases[::2] = 0                # B2-like: half of the rows are (0,0)
        d_b.upload(bases)
    d_p = ctx.bases_prepare(group, d_b, n)
    mask = ctx.zero_row_mask(group, d_p, n)
    d_s = ctx.to_device(s)
    cls = ctx.classes_create(n, 1)
    ctx.classes_build(cls, d_s, n, [mask], -1)
    ctx.msm_enqueue_classified(group, d_p, cls, 0)
    _, got = ctx.msm_finish(group)
    _, want = ol.msm(group, bases, s, nthreads=8)
    assert got == want
    ctx.sync()
    ctx.classes_destroy(cls)
    for d in (d_b, d_p, d_s, mask):
        d.free()


@pytest.mark.parametrize("group", [0, 1])
def test_bucket_sort_leaves_out_zero_rows_through_the_mask(ctx, group):
    """k16_msm_set_zero_row_mask: the ordinary (bucket) MSM with the table's (0,0) rows left out of its sort gives the same
    sum (curve.cpp:185-250: adding (0,0) returns the other operand); the mask covers ONE enqueue."""
    n = 1 << 17 if group == 0 else 1 << 15
    s = np_scalars(19 + group, n, "witness")
    d_b = ctx.synth_points(group, 3, n)
    bases = d_b.download(np.uint8, (n, 64 if group == 0 else 128)).copy()
    rs = np.random.RandomState(4)
    bases[rs.rand(n) < 0.5] = 0
    bases[:3] = 0
    d_b.upload(bases)
    d_p = ctx.bases_prepare(group, d_b, n)
    mask = ctx.zero_row_mask(group, d_p, n)
    m = mask.download(np.uint64)[: (n + 63) // 64]
    bits = np.unpackbits(m.view(np.uint8), bitorder="little")[:n].astype(bool)
    assert np.array_equal(bits, ~bases.any(axis=1))
    d_s = ctx.to_device(s)
    _, want = ol.msm(group, bases, s, nthreads=8)
    ctx.set_window_bits(13)
    try:
        ctx.msm_set_zero_row_mask(mask)
        ctx.msm_enqueue_prepared(group, d_p, d_s, n)
        _, got = ctx.msm_finish(group)
        assert got == want
        ctx.msm_enqueue_prepared(group, d_p, d_s, n)      # the mask was consumed: a plain sort again
        _, got = ctx.msm_finish(group)
        assert got == want
    finally:
        ctx.set_window_bits(0)
    for d in (d_b, d_p, d_s, mask):
        d.free()


@pytest.mark.parametrize("group", [0, 1])
@pytest.mark.parametrize("window_bits", [13, 0])
def test_tables_that_share_their_scalars_share_or_derive_the_bucket_sort(ctx, group, window_bits):
    """k16_msm_sort_from_lane: three tables indexed by ONE scalar array (groth16.cpp:88-112: A, B1, B2, C over the witness).
    Lane 0 sorts for table 0; lane 2 DERIVES lists of its own from lane 0's partition without the (0,0) rows of tables 1 and
    2 (half of them); lane 1 reads lane 2's lists as they are.  Every sum equals the oracle's multiexp of that table.
    window_bits 0 = automatic (c = 16 at these sizes: lane 0's sort is the staged five-byte one, whose partition cannot be
    read again -- the derived request then reads lane 0's lists and steps over the rows in the accumulation)."""
    import k16
    n = 1 << 17 if group == 0 else 1 << 16
    s = np_scalars(23 + group, n, "witness")
    s[n // 2: n // 2 + 4000] = np_scalars(5, 4000, "uniform")
    aff = k16.AFF_BYTES[group]
    rs = np.random.RandomState(8 + group)
    dead = rs.rand(n) < 0.5
    dead[:70] = True
    tabs = []
    for t in range(3):
        d = ctx.synth_points(group, 31 + t, n)
        b = d.download(np.uint8, (n, aff)).copy()
        d.free()
        if t:
            b[dead] = 0
        else:
            b[rs.rand(n) < 0.01] = 0
        tabs.append(b)
    d_s = ctx.to_device(s)
    d_tabs, masks = [], []
    for b in tabs:
        d_b = ctx.to_device(b)
        d_tabs.append(ctx.bases_prepare(group, d_b, n))
        d_b.free()
        masks.append(ctx.zero_row_mask(group, d_tabs[-1], n))
    want = [ol.msm(group, b, s, nthreads=8)[1] for b in tabs]
    ctx.set_window_bits(window_bits)
    try:
        for rep in range(2):                       # the second round replays captured graphs where they are on
            ctx.set_lane(0)
            ctx.msm_enqueue_prepared(group, d_tabs[0], d_s, n)
            ctx.set_lane(2)
            ctx.msm_sort_from_lane(0, derive=True)
            ctx.msm_set_zero_row_mask(masks[1])
            ctx.msm_enqueue_prepared(group, d_tabs[1], d_s, n)
            ctx.set_lane(1)
            ctx.msm_sort_from_lane(2 if window_bits else 0)
            ctx.msm_set_zero_row_mask(masks[1] if window_bits else None)
            ctx.msm_enqueue_prepared(group, d_tabs[2], d_s, n)
            ctx.set_lane(0)
            for t in range(3):
                _, got = ctx.msm_finish(group)
                assert got == want[t], (rep, t)
        # no matching sort on the named lane, the lane itself, a derived sort without a mask: refused, nothing left pending
        ctx.set_lane(2)
        ctx.msm_sort_from_lane(3, derive=True)
        ctx.msm_set_zero_row_mask(masks[1])
        with pytest.raises(k16.K16Error):
            ctx.msm_enqueue_prepared(group, d_tabs[1], d_s, n)
        ctx.msm_sort_from_lane(2, derive=True)
        ctx.msm_set_zero_row_mask(masks[1])
        with pytest.raises(k16.K16Error):
            ctx.msm_enqueue_prepared(group, d_tabs[1], d_s, n)
        ctx.msm_sort_from_lane(0, derive=True)
        with pytest.raises(k16.K16Error):
            ctx.msm_enqueue_prepared(group, d_tabs[1], d_s, n)
        assert ctx.msm_pending() == 0
        ctx.set_lane(0)
        ctx.msm_enqueue_prepared(group, d_tabs[1], d_s, n)     # and the context is as usable as before
        _, got = ctx.msm_finish(group)
        assert got == want[1]
    finally:
        ctx.set_lane(0)
        ctx.set_window_bits(0)
        ctx.sync()
    for d in d_tabs + masks + [d_s]:
        d.free()


OPTIONAL_PATHS = {"classes": {"K16_CLASSES": "1"}, "b_sort": {"K16_B_SORT": "1"}, "b_derive": {"K16_B_DERIVE": "1"},
                  "fused_h_scalars": {"K16_FUSED_HSCALARS": "1"}, "b2_first": {"K16_B2_FIRST": "1"},
                  # round 6: the scheduling switches of DESIGN.md 7b and the split G2 accumulation (k_accumulate_split)
                  "g2_split2": {"K16_G2_ACC_SPLIT": "2"}, "g2_split4_seg16": {"K16_G2_ACC_SPLIT": "4", "K16_WITNESS_SEG": "16"},
                  "h_lane3_wait_first": {"K16_H_LANE": "3", "K16_H_WAIT_FIRST": "1"}, "b1_lane3_seg48": {"K16_B1_LANE": "3", "K16_WITNESS_SEG": "48"}}


@pytest.mark.parametrize("mode", list(OPTIONAL_PATHS))
def test_keyless_shape_proof_through_the_optional_witness_paths(ctx, tmp_path, monkeypatch, mode):
    """BASELINE config 3 at its stated size (nVars 1,343,588, N = 2^21, B1 / B2 half (0,0)) with the witness MSMs taking
    the paths that are off by default -- scalar classes (K16_CLASSES=1), a bucket sort of B's own without its (0,0)
    rows (K16_B_SORT=1), bucket lists for B1 / B2 derived from A's partition without them (K16_B_DERIVE=1), the H scalars formed by the H MSM's
    counting pass (K16_FUSED_HSCALARS=1): proof JSON byte-equal to the CPU oracle's (RS/groth16.cpp:41-360), for two witnesses on one
    prover."""
    import bench
    import k16
    for k, v in OPTIONAL_PATHS[mode].items():
        monkeypatch.setenv(k, v)
    round6 = mode in ("g2_split2", "g2_split4_seg16", "h_lane3_wait_first", "b1_lane3_seg48")
    n_vars, N, n_coefs = bench.KEYLESS["n_vars"], bench.KEYLESS["domain"], bench.KEYLESS["n_coefs"]
    zk = str(tmp_path / "keyless_shape.zkey")
    wt = str(tmp_path / "keyless_shape.wtns")
    with open(zk, "wb") as f:
        f.write(bench.synth_zkey_bytes(ctx, k16, n_vars, 1, N, n_coefs))
    r, s = pm.limbs(pm.SplitMix64(277).below(pm.R)), pm.limbs(pm.SplitMix64(278).below(pm.R))
    ctx2 = k16.Context(0)            # the switches are read when a CONTEXT is created (round 5: no getenv on any hot path)
    try:
        p = k16.Prover(ctx2, zk)
        for seed in ((100,) if round6 else (100, 103)):
            w = bench.synth_witness(n_vars, seed)
            bench.write_wtns(wt, w)
            assert p.prove_mem(w, r, s) == ol.prove_files(zk, wt, r, s, nthreads=os.cpu_count() or 8)
        p.close()
    finally:
        ctx2.close()
