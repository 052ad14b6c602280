"""Stage times of ONE witness-like MSM (prepared table, forced c, 32-entry segments as the prover uses), G1 and G2."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import k16
from gpu_common import np_scalars
os.environ["K16_SEG"] = os.environ.get("K16_SEG", "32")
ctx = k16.Context(0)
n = 1343588
s = np_scalars(3, n, "witness")
d_s = ctx.to_device(s)
for group in (0, 1):
    d_b = ctx.synth_points(group, 7, n)
    prep = ctx.bases_prepare(group, d_b, n)
    for c in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "13").split(",")]:
        ctx.set_window_bits(c)
        for _ in range(2):
            ctx.msm_enqueue_prepared(group, prep, d_s, n); ctx.msm_finish(group)
        ctx.stats_enable(1); ctx.stats_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.msm_enqueue_prepared(group, prep, d_s, n); ctx.msm_finish(group)
        wall = (time.perf_counter() - t0) / 5 * 1e3
        st = {k: round(ctx.stats_get(k)[1] / max(ctx.stats_get(k)[0], 1), 3) for k in ("msm_sort", "msm_accumulate", "msm_fold", "msm_reduce")}
        print("G%d witness MSM n=%d c=%d: %.3f ms wall; stages %s" % (group + 1, n, c, wall, st), flush=True)
        ctx.stats_enable(0)
    d_b.free(); prep.free()
ctx.close()
