"""Host-side tail of a proof: time of the MSM finish calls' combine step (per proof) with and without the host pool."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd")); sys.path.insert(0, ROOT)
import k16
import bench
ctx = k16.Context(0)
n_vars, N, n_coefs = bench.KEYLESS["n_vars"], bench.KEYLESS["domain"], bench.KEYLESS["n_coefs"]
zk = bench.synth_zkey_bytes(ctx, k16, n_vars, 1, N, n_coefs)
zp = "/tmp/k16_tail_%d.zkey" % os.getpid()
open(zp, "wb").write(zk); del zk
p = k16.Prover(ctx, zp)
wits = [bench.synth_witness(n_vars, 100 + i) for i in range(3)]
for w in wits: p.prove_mem(w)
ctx.stats_enable(2); ctx.stats_reset()
lat = []
for i in range(12):
    t = time.perf_counter(); p.prove_mem(wits[i % 3]); lat.append((time.perf_counter() - t) * 1e3)
print("p50 %.3f ms; per proof: finish_wait %.3f ms, finish_combine %.3f ms (sum over the 5 MSMs), combine max %.3f" % (
    float(np.median(lat)), ctx.stats_get("host_finish_wait")[1] / 12, ctx.stats_get("host_finish_combine")[1] / 12,
    ctx.stats_get("host_finish_combine_max")[1]))
p.close(); ctx.close(); os.unlink(zp)
