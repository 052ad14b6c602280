"""Device time of the public NTT entry point (k16_ntt: bit reversal + fused passes) at one size, HIP-event statistics (dev aid).
    python tools/ntt_timing.py [log2n] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
import k16  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 21
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << logn
ctx = k16.Context(0)
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64).view(np.uint8).reshape(n, 32)
a[:, 31] &= 0x1F
d = ctx.to_device(a)
for _ in range(3):
    ctx.ntt_device(d, n)
ctx.sync()
ctx.stats_enable(True)
ctx.stats_reset()
for _ in range(reps):
    ctx.ntt_device(d, n)
ctx.sync()
print("log2n %d: %.1f us per transform (bit reversal + passes)" % (logn, ctx.stats_get("ntt")[1] * 1e3 / reps))
