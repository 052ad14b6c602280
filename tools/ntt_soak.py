"""Soak: NTT then iNTT of fresh random vectors must return the input (canonical Montgomery words), many times -- the
twiddle-free first stage pair, the lazy reductions and the conversions see ~10^9 fresh butterflies (the 2r
bound).  python tools/ntt_soak.py [log2n] [rounds]"""
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

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 21
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = 1 << logn
ctx = k16.Context(0)
bad, t0 = [], time.time()
for r in range(rounds):
    x = bench.fast_scalars(n, 5000 + r)            # < 2^253 < r: canonical words
    d = ctx.to_device(x)
    ctx.ntt_device(d, n, 2 * n, inverse=False)
    ctx.ntt_device(d, n, 2 * n, inverse=True)
    y = d.download(np.uint8, (n, 32))
    d.free()
    if not np.array_equal(x, y):
        bad.append((r, int((x != y).any(axis=1).sum())))
print(json.dumps({"soak": "NTT / iNTT round trip 2^%d" % logn, "rounds": rounds, "mismatches": bad, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
