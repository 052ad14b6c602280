"""Soak: many G1 MSMs (generic, prepared and fixed-base paths) with fresh uniform scalars, each checked against the closed
form  MSM(s, (i+1)G) = (sum s_i (i+1) mod r) G  computed exactly on the host -- looks for rare data-dependent faults
(lazy-reduction bounds, races) that a handful of parity vectors cannot hit.
    python tools/msm_soak.py [log2n] [rounds] [group: 0 = G1 (default), 1 = G2]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import k16  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
GRP = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = 1 << logn
ctx = k16.Context(0)
d_b = ctx.synth_points(GRP, 0, n)
d_prep = ctx.bases_prepare(GRP, d_b, n)
tab, c = ctx.fixed_base_prepare(GRP, d_b, n) if GRP == 0 else (None, 0)
d_g1 = ctx.synth_points(GRP, 0, 1)


def k_times_g(k):
    d_k = ctx.to_device(np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32))
    _, aff = ctx.msm_device(GRP, d_g1, d_k, 1)
    d_k.free()
    return aff


bad, t0 = [], time.time()
for r in range(rounds):
    s = bench.fast_scalars(n, 1000 + r)
    if r % 3 == 0:                       # every seventh scalar with bit 253 set: up to 2^254 > r, as raw bits (multiexp.cpp:26-41)
        s[:: 7, 31] |= 0x20
    want = k_times_g(bench.weighted_sum_mod_r(s, 0))
    d_s = ctx.to_device(s)
    got = {}
    _, got["generic"] = ctx.msm_device(GRP, d_b, d_s, n)
    ctx.msm_enqueue_prepared(GRP, d_prep, d_s, n)
    _, got["prepared"] = ctx.msm_finish(GRP)
    if tab is not None:
        ctx.msm_enqueue_fixed_base(k16.G1, tab, d_s, n)
        _, got["fixed_base"] = ctx.msm_finish(k16.G1)
    d_s.free()
    for k, v in got.items():
        if bytes(v) != bytes(want):
            bad.append((r, k))
print(json.dumps({"soak": "G%d MSM 2^%d vs closed form" % (GRP + 1, logn), "rounds": rounds, "paths": ["generic", "prepared"] + (["fixed_base c=%d" % c] if tab is not None else []),
                  "mismatches": bad, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
