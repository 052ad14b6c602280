"""Differential fuzz of the round-4 MSM paths against the CPU oracle: the scalar-class MSM (k16_scalar_classes_* /
k16_msm_enqueue_classified: 1-4 tables with their own (0,0) rows sharing one classification, exact / generous / read-back
wide bounds), the bucket sort with a zero-row mask (k16_msm_set_zero_row_mask), and -- at sizes of 2^16 and above with the
automatic window size (c = 16) -- the five-byte staged sort, and bucket lists shared between lanes or derived from another
lane's partition (k16_msm_sort_from_lane).  Random sizes, both groups, scalar kinds mixed in stretches,
random (0,0) rows, duplicates.      python tools/classes_fuzz.py [cases] [seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "keyless-zk-proofs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import k16  # noqa: E402
import oracle_lib as ol  # noqa: E402
from gpu_common import np_scalars  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
ctx = k16.Context(0)
KINDS = ["uniform", "full256", "ones", "zeros", "witness", "witness", "witness", "topwindow", "same"]
NMAX = {0: 300000, 1: 60000}
pool = {}
for g in (0, 1):
    d = ctx.synth_points(g, 11, NMAX[g])
    pool[g] = d.download(np.uint8, (NMAX[g], k16.AFF_BYTES[g])).copy()
    d.free()
bad, t0, counts = [], time.time(), {"classified": 0, "masked_bucket": 0, "plain": 0, "shared_or_derived": 0}
for c in range(cases):
    g = int(rs.rand() < 0.3)
    n = int(rs.choice([0, 1, 63, 64, 65, 2047, 2048, 2049, 65535, 65536, 65537, 131072])) if rs.rand() < 0.4 else int(rs.randint(0, NMAX[g] + 1))
    n = min(n, NMAX[g])
    sc = np.zeros((n, 32), dtype=np.uint8)
    for lo in range(0, n, max(1, n // 5)):
        hi = min(n, lo + max(1, n // 5))
        sc[lo:hi] = np_scalars(int(rs.randint(1 << 30)), hi - lo, KINDS[int(rs.randint(len(KINDS)))])
    n_tabs = int(rs.randint(1, 5))
    tabs = []
    for t in range(n_tabs):
        b = pool[g][rs.permutation(NMAX[g])[:n]].copy() if n else pool[g][:0].copy()
        if n:
            b[rs.rand(n) < rs.choice([0.0, 0.02, 0.5, 0.97])] = 0
            if n >= 4:
                i = rs.permutation(n)[:2]
                b[i[0]] = b[i[1]]
        tabs.append(b)
    d_s = ctx.to_device(sc) if n else None
    d_tabs, masks = [], []
    for b in tabs:
        if n:
            d_b = ctx.to_device(b)
            d_tabs.append(ctx.bases_prepare(g, d_b, n))
            d_b.free()
            masks.append(ctx.zero_row_mask(g, d_tabs[-1], n))
        else:
            d_tabs.append(None)
            masks.append(None)
    want = [ol.msm(g, b, sc, nthreads=8)[1] for b in tabs]
    n_wide = int(sc[:, 1:].any(axis=1).sum()) if n else 0
    cls = ctx.classes_create(max(n, 1), n_tabs)
    try:
        bound = [-1, n_wide, min(n, n_wide + int(rs.randint(0, 50)))][int(rs.randint(3))]
        use_masks = rs.rand() < 0.8
        ctx.classes_build(cls, d_s, n, masks if use_masks else [None] * n_tabs, bound)
        for t in range(n_tabs):
            ctx.set_lane(t % 3)
            ctx.msm_enqueue_classified(g, d_tabs[t], cls, t)
        ctx.set_lane(0)
        for t in range(n_tabs):
            _, got = ctx.msm_finish(g)
            counts["classified"] += 1
            if got != want[t]:
                bad.append((c, "classified", g, n, t))
        if n:
            # bucket path, automatic window size (n >= 2^16: c = 16 without the mask -> the staged sort), with and without the mask
            ctx.msm_enqueue_prepared(g, d_tabs[0], d_s, n)
            _, got = ctx.msm_finish(g)
            counts["plain"] += 1
            if got != want[0]:
                bad.append((c, "plain", g, n, 0))
            ctx.msm_set_zero_row_mask(masks[0])
            ctx.msm_enqueue_prepared(g, d_tabs[0], d_s, n)
            _, got = ctx.msm_finish(g)
            counts["masked_bucket"] += 1
            if got != want[0]:
                bad.append((c, "masked_bucket", g, n, 0))
            if n_tabs >= 2:
                # k16_msm_sort_from_lane: lane 0 sorts for table 0 (forced window size on every other case: the plain
                # partition, else the staged one at n >= 2^16), lane 2 derives lists without table 1's (0,0) rows from lane 0's
                # partition (or falls back to stepping over them), lane 1 reads lane 2's lists for table 1 again
                wb = int(rs.choice([0, 0, 9, 13]))
                ctx.set_window_bits(wb)
                ctx.set_lane(0)
                ctx.msm_enqueue_prepared(g, d_tabs[0], d_s, n)
                ctx.set_lane(2)
                ctx.msm_sort_from_lane(0, derive=True)
                ctx.msm_set_zero_row_mask(masks[1])
                ctx.msm_enqueue_prepared(g, d_tabs[1], d_s, n)
                ctx.set_lane(1)
                ctx.msm_sort_from_lane(0)
                ctx.msm_enqueue_prepared(g, d_tabs[0], d_s, n)
                ctx.set_lane(0)
                ctx.set_window_bits(0)
                for t in (0, 1, 0):
                    _, got = ctx.msm_finish(g)
                    counts["shared_or_derived"] += 1
                    if got != want[t]:
                        bad.append((c, "shared_or_derived", g, n, t, wb))
    finally:
        ctx.sync()
        ctx.classes_destroy(cls)
        for d in d_tabs + masks + [d_s]:
            if d is not None:
                d.free()
print(json.dumps({"fuzz": "scalar-class MSM, masked bucket sort, staged sort, shared / derived bucket lists vs oracle", "cases": cases, "seed": seed, "checks": counts,
                  "mismatches": bad, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
