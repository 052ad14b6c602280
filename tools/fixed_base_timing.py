"""One fixed-base G1 MSM at a time (the prover's H MSM shape: n = 2^21, c = 20), for rocprofv3 --kernel-trace --stats (dev aid).
    python tools/fixed_base_timing.py [log2n] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
import k16  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 21
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << logn
ctx = k16.Context(0)
d_b = ctx.synth_points(0, 1, n)
tab, _c = ctx.fixed_base_prepare(0, d_b, n)
rng = np.random.default_rng(3)
s = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
s[:, 3] &= (1 << 60) - 1
d_s = ctx.to_device(s.view(np.uint8).reshape(n, 32))
for _ in range(2):
    ctx.msm_enqueue_fixed_base(0, tab, d_s, n)
    ctx.msm_finish(0)
ctx.stats_enable(True)
ctx.stats_reset()
for _ in range(reps):
    ctx.msm_enqueue_fixed_base(0, tab, d_s, n)
    ctx.msm_finish(0)
print("fixed-base 2^%d:" % logn, {k: round(ctx.stats_get(k)[1] / reps, 3) for k in ("msm_sort", "msm_accumulate", "msm_fold", "msm_reduce")})
