"""Differential fuzz of the whole prove() path against the CPU oracle: random circuit shapes (wires, public inputs, domain,
coefficient count, a few long constraint rows), random witnesses of the Keyless mix, injected (r, s); proof JSON and H
scalars must be byte-equal.  python tools/prove_fuzz.py [cases] [seed]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "keyless-zk-proofs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import k16  # noqa: E402
import oracle_lib as ol  # noqa: E402
import pymodel as pm  # noqa: E402
import zkey_builder as zb  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
ctx = k16.Context(0)
d = tempfile.mkdtemp()
zk, wt = d + "/f.zkey", d + "/f.wtns"
bad, shapes, t0 = [], [], time.time()
for c in range(cases):
    logn = int(rs.randint(0, 15))
    N = 1 << logn
    n_pub = int(rs.randint(1, 4))
    n_vars = n_pub + 2 + int(rs.randint(1, max(2, min(3 * N, 6000))))
    n_coefs = int(rs.randint(1, max(2, min(4 * N, 12000)) + 1))
    longs = tuple(int(x) for x in rs.randint(65, 700, size=rs.randint(0, 3)) if x <= n_coefs // 3)
    try:
        zb.build_zkey(zk, n_vars, n_pub, N, n_coefs, seed=seed * 1000 + c, long_rows=longs)
    except AssertionError:
        longs = ()
        zb.build_zkey(zk, n_vars, n_pub, N, n_coefs, seed=seed * 1000 + c)
    zb.build_wtns(wt, n_vars, seed=seed * 2000 + c)
    r, s = pm.limbs(pm.SplitMix64(c).below(pm.R)), pm.limbs(pm.SplitMix64(c + 77).below(pm.R))
    p = k16.Prover(ctx, zk)
    got = p.prove_file(wt, r, s)
    h = p.last_h()
    p.close()
    want, h_ref = ol.prove_files(zk, wt, r, s, nthreads=4, want_h=True)
    shapes.append((n_vars, n_pub, N, n_coefs, longs))
    if got != want or not np.array_equal(h, h_ref):
        bad.append(shapes[-1])
print(json.dumps({"fuzz": "prove() vs oracle, random small circuits", "cases": cases, "seed": seed, "mismatches": bad,
                  "domains_seen": sorted({s[2] for s in shapes}), "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
