import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, k16, oracle_lib as ol
from gpu_common import np_scalars
ctx = k16.Context(0)
n = 150000
bases = ol.gen_points(0, 17, n)
sc = np_scalars(41, n, "witness")
_, want = ol.msm(0, bases, sc, nthreads=8)
ctx.set_window_bits(13)
d_b = ctx.to_device(bases); d_s = ctx.to_device(sc)
bad = [i for i in range(int(os.environ.get("ITERS", "300"))) if ctx.msm_device(0, d_b, d_s, n)[1] != want]
print("bad iterations:", bad, file=sys.stderr, flush=True)
