"""Quick device-side timing of the MSM / NTT stages (development aid; bench.py is the contract)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import k16  # noqa: E402
import oracle_lib as ol  # noqa: E402
from gpu_common import np_scalars  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["uniform", "witness"]
ctx = k16.Context(0)
n = 1 << logn
t0 = time.time()
bases = ol.gen_points(0, 0, n)
print("gen_points %.1fs" % (time.time() - t0), flush=True)
d_b = ctx.to_device(bases)
for kind in kinds:
    s = np_scalars(1, n, kind)
    d_s = ctx.to_device(s)
    for cbits in ([0] if len(sys.argv) <= 3 else [int(x) for x in sys.argv[3].split(",")]):
        ctx.set_window_bits(cbits)
        ctx.msm_device(0, d_b, d_s, n)  # warm-up (workspace allocation)
        ctx.stats_enable(False)
        ctx.timer_start()
        iters = 5
        for _ in range(iters):
            ctx.msm_enqueue(0, d_b, d_s, n)
        for _ in range(iters):
            ctx.msm_finish(0)
        ms = ctx.timer_stop()
        print("G1 MSM 2^%d %-8s c=%2d : %.3f ms/msm  (%.1f Mpts/s)" % (logn, kind, cbits, ms / iters, n / (ms / iters) / 1e3), flush=True)
        ctx.stats_enable(True)
        ctx.stats_reset()
        ctx.msm_device(0, d_b, d_s, n)
        print("   stages:", {k: round(ctx.stats_get(k)[1], 3) for k in ("msm_sort", "msm_accumulate", "msm_fold", "msm_reduce")}, flush=True)
        ctx.stats_enable(False)
    d_s.free()
ctx.set_window_bits(0)
# NTT
for ln in (16, 21):
    m = 1 << ln
    a = np.random.RandomState(0).randint(0, 2 ** 60, size=(m, 4)).astype(np.uint64)
    d_a = ctx.to_device(a)
    ctx.ntt_device(d_a, m, 2 * m)
    ctx.timer_start()
    for _ in range(5):
        ctx.ntt_device(d_a, m, 2 * m)
    print("NTT 2^%d : %.3f ms" % (ln, ctx.timer_stop() / 5), flush=True)
    d_a.free()
