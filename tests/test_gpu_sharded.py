"""ONE MSM sharded over several devices through the C entry points (include/k16.h: k16_msm_sharded_*, k16_rank_comm_*) --
SURVEY 8(e) / BASELINE config 5 from C / C++, not only from Python.  A one-GPU box runs the shards as several contexts on
device 0 (the code path is the one several devices take: a context, a slice of the table and a host thread per shard, the
host-side EC-add fold); with >= 2 devices the same tests spread over them.  The per-shard pipeline is the product's ordinary
MSM, so the results are compared with the oracle's multiexp (RS/multiexp.cpp:183-245) and with the closed form of
alt_bn128_test.cpp:172-212."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
import pymodel as pm
from gpu_common import np_scalars

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import k16
    c = k16.Context(0)   # raises if libk16.so is missing or there is no GPU: no fallback
    yield c
    c.close()


def _devices(k16, shards):
    n = max(1, k16.load().k16_device_count())
    return [r % n for r in range(shards)]


@pytest.mark.parametrize("group", [0, 1])
@pytest.mark.parametrize("shards,n", [(2, (1 << 12) + 5), (3, 1000), (2, 1), (3, 2), (2, 0), (1, 777)])
def test_sharded_msm_equals_the_oracle(group, shards, n):
    import k16
    bases = ol.gen_points(group, 3, max(n, 1))[:n]
    scalars = np_scalars(41 + n, n, "full256")
    if n >= 8:
        bases[1] = 0                     # (0,0) row
        bases[5] = bases[4]              # duplicate
        scalars[2] = 0
    sm = k16.ShardedMsm(_devices(k16, shards), group, n)
    try:
        assert sm.count() == shards
        cover = [sm.shard_range(r) for r in range(shards)]
        assert cover[0][0] == 0 and cover[-1][1] == n and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        sm.set_bases(bases)
        _, got = sm.run(scalars)
        _, want = ol.msm(group, bases, scalars, nthreads=4)
        assert got == want
        # a second run with other scalars on the resident table
        s2 = np_scalars(99, n, "witness")
        assert sm.run(s2)[1] == ol.msm(group, bases, s2, nthreads=4)[1]
    finally:
        sm.close()


def test_sharded_msm_2p22_closed_form_with_tables_made_on_the_shards():
    """Two shards of 2^21 rows: every shard fills ITS slice on its own device ((lo + i + 1) * G, k16_synth_points on the
    shard's context), scalars arrive from the host; result == (sum s_i (i + 1)) * G."""
    import bench
    import k16
    n = 1 << 22
    sm = k16.ShardedMsm(_devices(k16, 2), k16.G1, n)
    try:
        for r in range(2):
            lo, hi = sm.shard_range(r)
            c = sm.shard_ctx(r)
            d = c.synth_points(k16.G1, lo, hi - lo)
            sm.set_bases_device(r, d)
            d.free()
        scalars = bench.fast_scalars(n, seed=7)
        xyzz, aff = sm.run(scalars)
        k = bench.weighted_sum_mod_r(scalars, 0)
        ctx0 = sm.shard_ctx(0)
        assert aff == bench.scalar_times_g(ctx0, k16, k)
        ms = sm.last_ms()
        assert ms["total_ms"] > 0 and ms["fold_ms"] < ms["total_ms"]
        # scalars already resident on the shards
        ds = []
        for r in range(2):
            lo, hi = sm.shard_range(r)
            ds.append(sm.shard_ctx(r).to_device(scalars[lo:hi]))
        assert sm.run_device(ds)[1] == aff
        for d in ds:
            d.free()
    finally:
        sm.close()


def test_sharded_msm_error_paths():
    import k16
    L = k16.load()
    with pytest.raises(k16.K16Error):
        k16.ShardedMsm([], k16.G1, 10)
    with pytest.raises(k16.K16Error):
        k16.ShardedMsm([0], 7, 10)
    with pytest.raises(k16.K16Error):
        k16.ShardedMsm([L.k16_device_count() + 3], k16.G1, 10)      # no such device
    sm = k16.ShardedMsm([0, 0], k16.G1, 64)
    try:
        with pytest.raises(k16.K16Error) as ei:
            sm.run(np_scalars(1, 64, "uniform"))                   # bases were never set
        assert ei.value.rc == -3
    finally:
        sm.close()


def test_rank_comm_world_of_one_through_rccl(ctx):
    """The one-process-per-GPU exchange (ncclCommInitRank + ncclAllGather, RCCL dlopen'ed by the library) at world size 1:
    the gathered result is the rank's own partial; with a second device the two-rank leg runs in
    test_gpu_multirank.py."""
    import k16
    uid = k16.RankComm.unique_id()
    assert len(uid) == 128
    rc = k16.RankComm(ctx, 0, 1, uid)
    try:
        g = ol.generator(0)
        part = ol.mul_scalar(0, g, pm.limbs(123456789))
        x, a = rc.allgather_fold(k16.G1, part)
        assert ol.pt_eq(0, x, part)
        part2 = ol.mul_scalar(1, ol.generator(1), pm.limbs(987654321))
        x2, _ = rc.allgather_fold(k16.G2, part2)
        assert ol.pt_eq(1, x2, part2)
    finally:
        rc.close()
