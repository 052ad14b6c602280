"""Differential fuzz of k16_pairing_vec against the CPU oracle (ark-ec 0.4.2 Bn::pairing restated in oracle/pairing_ref.h):
random multiples of the generators, GT VALUES (384 bytes) byte-equal; plus bilinearity on the device alone:
e(aP, bQ) == e(abP, Q).  python tools/pairing_fuzz.py [pairs] [seed]"""
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

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = k16.Context(0)
rng = pm.SplitMix64(seed)
a = [rng.below(pm.R - 1) + 1 for _ in range(pairs)]
b = [rng.below(pm.R - 1) + 1 for _ in range(pairs)]
a[0], b[0] = 1, 1
a[1], b[1] = pm.R - 1, 1
t0 = time.time()
g1 = ctx.synth_points_scalars(0, a)
g2 = ctx.synth_points_scalars(1, b)
got = k16.pairing_vec(ctx, g1, g2)
bad = [i for i in range(pairs) if bytes(got[i]) != ol.pairing(bytes(g1[i]), bytes(g2[i]))]
ab = ctx.synth_points_scalars(0, [x * y % pm.R for x, y in zip(a, b)])
q = ctx.synth_points_scalars(1, [1] * pairs)
bil = k16.pairing_vec(ctx, ab, q)
bad_bil = [i for i in range(pairs) if bytes(bil[i]) != bytes(got[i])]
print(json.dumps({"fuzz": "k16_pairing_vec vs oracle + bilinearity", "pairs": pairs, "seed": seed, "value_mismatches": bad,
                  "bilinearity_mismatches": bad_bil, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad or bad_bil else 0)
