"""Round 6 host-code stress (dev aid, run on a GPU box from the repository root): persistent shard workers through create / run /
destroy cycles, and the rank communicator's ring of four exchanges in flight through real RCCL at world size 1."""
import sys, time
sys.path.insert(0, 'keyless-zk-proofs_amd'); sys.path.insert(0, 'tests')
import numpy as np
import k16, oracle_lib as ol
from gpu_common import np_scalars
n = 2000
bases = ol.gen_points(0, 3, n)
sc = np_scalars(9, n, "full256")
want = ol.msm(0, bases, sc, nthreads=4)[1]
t0 = time.time()
for it in range(60):
    sm = k16.ShardedMsm([0] * (1 + it % 5), k16.G1, n)
    sm.set_bases(bases)
    if it % 3 == 0:
        sm.set_piece_rows(128, 128)
    for _ in range(1 + it % 4):
        assert sm.run(sc)[1] == want
    sm.close()
print("shard worker stress ok: 60 create / run / destroy cycles, 1-5 shards, %.1f s" % (time.time() - t0))
# pool lease stress through ctypes is covered by the harness tests; rank comm ring: world 1 through real RCCL, 200 exchanges, 4 in flight
ctx = k16.Context(0)
rc = k16.RankComm(ctx, 0, 1, k16.RankComm.unique_id())
g = ol.generator(0)
import pymodel as pm
parts = [ol.mul_scalar(0, g, pm.limbs(1000 + i)) for i in range(8)]
inflight = []
for i in range(200):
    rc.allgather_start(k16.G1, parts[i % 8]); inflight.append(i % 8)
    if len(inflight) == 4:
        j = inflight.pop(0)
        x, _ = rc.allgather_finish()
        assert ol.pt_eq(0, x, parts[j])
while inflight:
    j = inflight.pop(0)
    assert ol.pt_eq(0, rc.allgather_finish()[0], parts[j])
try:
    for _ in range(5):
        rc.allgather_start(k16.G1, parts[0])
    raise SystemExit("a fifth exchange in flight was accepted")
except k16.K16Error as e:
    assert e.rc == -3
for _ in range(4):
    rc.allgather_finish()
rc.close(); ctx.close()
print("rank comm ring stress ok (real RCCL, world 1)")
