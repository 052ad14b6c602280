"""Differential fuzz of k16_ntt against the CPU oracle (FFT::fft / ::ifft, RS/fft.cpp:192-246): random sizes 2^0 ... 2^17,
table sizes n ... 8n, both directions, inputs that mix uniform elements with 0, 1, r - 1 and Montgomery one.
python tools/ntt_fuzz.py [cases] [seed]"""
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

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
ctx = k16.Context(0)
special = [np.frombuffer(pm.limbs(v), dtype=np.uint64) for v in (0, 1, pm.R - 1, pm.MONT % pm.R, (pm.R - 1) // 2)]
bad, t0, sizes = [], time.time(), set()
for c in range(cases):
    logn = int(rs.randint(0, 18))
    n = 1 << logn
    a = rs.randint(0, 2 ** 63, size=(n, 4)).astype(np.uint64)
    a[:, 3] &= (1 << 60) - 1
    for i in rs.randint(0, n, size=min(n, 6)):
        a[i] = special[int(rs.randint(len(special)))]
    dom = n << int(rs.randint(0, 4))
    inv = bool(rs.randint(2))
    sizes.add(logn)
    if not np.array_equal(ctx.ntt(a, max_domain=dom, inverse=inv), ol.ntt(a, max_domain=dom, inverse=inv)):
        bad.append((logn, dom, inv))
print(json.dumps({"fuzz": "k16_ntt vs oracle", "cases": cases, "seed": seed, "log2_sizes_seen": sorted(sizes), "mismatches": bad,
                  "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
