"""Single-GPU leg of BASELINE config 5: one 2^26-point G1 MSM through k16_msm (chunks of 2^24 on two lanes + fold).
Scalars: one uniform 2^24 block repeated (generating 2 GB of host randomness is not the point); bases (i+1)G on device."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
import k16  # noqa: E402
from bench import uniform_scalars  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log2n
blk = min(n, 1 << 24)
ctx = k16.Context(0)
d_b = ctx.synth_points(k16.G1, 0, n)
sc = uniform_scalars(blk, 7)
d_s = ctx.alloc(n * 32)
for k in range(n // blk):
    ctx._chk(ctx.L.k16_h2d(ctx.h, (d_s.ptr.value if hasattr(d_s.ptr, "value") else d_s.ptr) + k * blk * 32, sc.ctypes.data, blk * 32))
ctx.sync()
ctx.msm_device(k16.G1, d_b, d_s, n)  # warm-up: workspace allocation
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    ctx.msm_device(k16.G1, d_b, d_s, n)
    best = min(best, time.perf_counter() - t0)
print('{"workload": "BN254 G1 MSM 2^%d on one MI355X (k16_msm, chunked)", "ms": %.2f, "points_per_s": %.4g}' % (log2n, best * 1e3, n / best))
