import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, k16, oracle_lib as ol
from gpu_common import np_scalars
ctx = k16.Context(0)
for group, n in ((0, 1), (1, 1), (1, 64), (1, 2048)):
    print("group", group, "n", n, flush=True)
    bases = ol.gen_points(group, 0, n)
    sc = np_scalars(3, n, "full256")
    _, got = ctx.msm(group, bases, sc)
    print("  gpu done", flush=True)
    _, want = ol.msm(group, bases, sc, nthreads=4)
    print("  match", got == want, flush=True)
