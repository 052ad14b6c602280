"""Differential fuzz of the MSM entry points against the CPU oracle: random sizes (0 ... 40 000), both groups, every scalar
pattern of tests/gpu_common.np_scalars mixed row by row, bases with (0,0) rows, duplicates and negations, forced window
sizes, the host, device, prepared and fixed-base paths.  python tools/msm_fuzz.py [cases] [seed]"""
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
import pymodel as pm  # noqa: E402
from gpu_common import np_scalars  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
ctx = k16.Context(0)
KINDS = ["uniform", "full256", "ones", "zeros", "witness", "topwindow", "same"]
pool = {g: ol.gen_points(g, 3, 40000) for g in (0, 1)}
bad, t0, paths = [], time.time(), {}
for c in range(cases):
    g = int(rs.rand() < 0.3)
    n = int(rs.choice([0, 1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, 4096, 8191, 8192, 20000, 32768, 40000])) if rs.rand() < 0.5 \
        else int(rs.randint(0, 40001 if g == 0 else 12001))
    n = min(n, 40000 if g == 0 else 12000)
    bases = pool[g][rs.permutation(40000)[:n]].copy() if n else pool[g][:0].copy()
    sc = np.zeros((n, 32), dtype=np.uint8)
    if n:
        for lo in range(0, n, max(1, n // 4)):       # four stretches of different scalar kinds
            hi = min(n, lo + max(1, n // 4))
            sc[lo:hi] = np_scalars(int(rs.randint(1 << 30)), hi - lo, KINDS[int(rs.randint(len(KINDS)))])
        k = int(rs.randint(0, 4))
        if n >= 8 and k:
            idx = rs.permutation(n)[:8]
            bases[idx[0]] = 0                                 # (0,0) rows are skipped (multiexp.cpp:59)
            bases[idx[1]] = bases[idx[2]]                     # duplicate base
            bases[idx[3]] = np.frombuffer(ol.pt_to_affine(g, ol.mul_scalar(g, bytes(bases[idx[4]]), pm.limbs(pm.R - 1))), dtype=np.uint8)  # -P
            sc[idx[5]] = sc[idx[6]]
    cbits = int(rs.choice([0, 0, 0, 4, 8, 11, 13, 16]))
    ctx.set_window_bits(cbits)
    try:
        _, want = ol.msm(g, bases, sc, nthreads=4)
        got = {}
        _, got["host"] = ctx.msm(g, bases, sc)
        if n:
            d_b, d_s = ctx.to_device(bases), ctx.to_device(sc)
            _, got["device"] = ctx.msm_device(g, d_b, d_s, n)
            d_p = ctx.bases_prepare(g, d_b, n)
            ctx.msm_enqueue_prepared(g, d_p, d_s, n)
            _, got["prepared"] = ctx.msm_finish(g)
            if g == 0 and cbits == 0:
                tab, fc = ctx.fixed_base_prepare(g, d_b, n)
                if tab is not None:
                    ctx.msm_enqueue_fixed_base(g, tab, d_s, n)
                    _, got["fixed_base"] = ctx.msm_finish(g)
                    tab.free()
            for b in (d_b, d_s, d_p):
                b.free()
        for k2, v in got.items():
            paths[k2] = paths.get(k2, 0) + 1
            if bytes(v) != bytes(want):
                bad.append((c, g, n, cbits, k2))
    finally:
        ctx.set_window_bits(0)
print(json.dumps({"fuzz": "MSM entry points vs oracle", "cases": cases, "seed": seed, "checks_per_path": paths, "mismatches": bad,
                  "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
