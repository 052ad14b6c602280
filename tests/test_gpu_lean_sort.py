"""-m gpu : the <= 32-VGPR variants of the bucket-sort kernels (K16_LEAN_SORT=1: k_convert_bases_lean, k_part_scatter_lean,
k_part_bins<..., LEAN>) against the CPU oracle -- a child process, because the switch is read when a context is created.
RS/multiexp.cpp:26-71 (getChunk + the scatter of processChunk) is what the sort replaces."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r'''
import sys, os
import numpy as np
sys.path.insert(0, %(here)r); sys.path.insert(0, os.path.join(os.path.dirname(%(here)r), "keyless-zk-proofs_amd"))
import k16, oracle_lib as ol
from gpu_common import np_scalars
ctx = k16.Context(0)
for group, n, kind in ((0, 5000, "uniform"), (0, 70000, "witness"), (0, 1 << 17, "full256"), (1, 20000, "uniform"),
                       (0, 3, "ones"), (0, 40000, "same")):
    bases = ol.gen_points(group, 3, n)
    if n > 100:
        bases[7] = 0                       # a (0,0) row
        bases[11] = bases[12]              # a duplicate
    sc = np_scalars(17 + n, n, kind)
    _, want = ol.msm(group, bases, sc, nthreads=os.cpu_count() or 8)
    _, got = ctx.msm(group, bases, sc)                      # unprepared table: k_convert_bases_lean for G1
    assert got == want, (group, n, kind)
    d_b, d_s = ctx.to_device(bases), ctx.to_device(sc)
    prep = ctx.bases_prepare(group, d_b, n)
    ctx.msm_enqueue_prepared(group, prep, d_s, n)
    assert ctx.msm_finish(group)[1] == want, ("prepared", group, n, kind)
    d_b.free(); d_s.free(); prep.free()
# the fixed-base (flat) path
n = 1 << 16
bases = ol.gen_points(0, 5, n)
sc = np_scalars(99, n, "uniform")
d_b, d_s = ctx.to_device(bases), ctx.to_device(sc)
tab, c = ctx.fixed_base_prepare(k16.G1, d_b, n)
ctx.msm_enqueue_fixed_base(k16.G1, tab, d_s, n)
assert ctx.msm_finish(k16.G1)[1] == ol.msm(0, bases, sc, nthreads=os.cpu_count() or 8)[1]
ctx.close()
print("lean sort OK")
'''


def test_msm_with_the_lean_sort_kernels():
    env = dict(os.environ, K16_LEAN_SORT="1")
    out = subprocess.run([sys.executable, "-c", CHILD % {"here": HERE}], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "lean sort OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
