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


def test_page_locked_scalars_give_the_same_msm(ctx):
    """k16_host_register / k16_host_unregister (include/k16.h): the caller's scalar array page-locked for DMA uploads -- same
    results through k16_msm_host and k16_msm_sharded_run as from pageable memory; a null pointer is an error code."""
    import k16
    n = (1 << 16) + 77
    bases = ol.gen_points(0, 9, n)
    scalars = np_scalars(404, n, "full256")
    want = ol.msm(0, bases, scalars, nthreads=4)[1]
    assert ctx.msm(0, bases, scalars)[1] == want
    ctx.host_register(scalars)
    try:
        assert ctx.msm(0, bases, scalars)[1] == want
        sm = k16.ShardedMsm(_devices(k16, 2), k16.G1, n)
        try:
            sm.set_bases(bases)
            sm.set_piece_rows(4096, 0)
            assert sm.run(scalars)[1] == want
        finally:
            sm.close()
    finally:
        ctx.host_unregister(scalars)
    assert ctx.msm(0, bases, scalars)[1] == want             # pageable again
    with pytest.raises(k16.K16Error):
        ctx._chk(ctx.L.k16_host_register(ctx.h, None, 16))   # null pointer: K16_ERR_ARG


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


@pytest.mark.parametrize("group", [0, 1])
def test_pieced_pipeline_many_small_pieces_equals_the_oracle(group):
    """the pieced host-scalar pipeline of k16_msm_sharded_run (msm_sharded.hip sharded_run: pieces uploaded through
    lane 0 while up to two earlier pieces' MSMs run on lanes 1 / 2, offset slices of the prepared table, the per-piece fold)
    only starts above 2^22 rows per shard at the default piece size -- k16_msm_sharded_set_piece_rows makes it run at test
    sizes: 5+ pieces per shard, the three-in-flight branch included, host and resident scalars, against the oracle's multiexp."""
    import k16
    n = 11003 if group == 0 else 3001
    bases = ol.gen_points(group, 3, n)
    bases[7] = 0
    sm = k16.ShardedMsm(_devices(k16, 2), group, n)
    try:
        sm.set_bases(bases)
        with pytest.raises(k16.K16Error):
            sm.set_piece_rows(5, 0)                           # below 64 rows
        with pytest.raises(k16.K16Error):
            sm.set_piece_rows(0, (1 << 24) + 1)
        for kind, host_rows, dev_rows in (("full256", 1024, 2048), ("witness", 700, 64), ("uniform", 5501, 5500)):
            sm.set_piece_rows(host_rows if group == 0 else max(64, host_rows // 4), dev_rows if group == 0 else max(64, dev_rows // 4))
            sc = np_scalars(31 + len(kind), n, kind)
            want = ol.msm(group, bases, sc, nthreads=4)[1]
            assert sm.run(sc)[1] == want, (kind, "host")
            ds = []
            for r in range(2):
                lo, hi = sm.shard_range(r)
                ds.append(sm.shard_ctx(r).to_device(sc[lo:hi]))
            assert sm.run_device(ds)[1] == want, (kind, "device")
            for d in ds:
                d.free()
        sm.set_piece_rows(0, 0)                               # defaults again: one piece per shard
        sc = np_scalars(77, n, "full256")
        assert sm.run(sc)[1] == ol.msm(group, bases, sc, nthreads=4)[1]
    finally:
        sm.close()


def test_a_failing_piece_drains_the_pieces_in_flight(tmp_path):
    """A piece that fails with two earlier pieces' MSMs still in flight (K16_FAULT_INJECT=shard_piece:k, testing build of the
    library only): the run returns the error, k16_msm_abort_all has drained the shard's context (nothing pending), and the
    next run on the same object is correct.  The production library ignores the variable."""
    import subprocess
    code = r'''
import os, sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(pkg)r)
import numpy as np
import k16, oracle_lib as ol
from gpu_common import np_scalars
testing = os.path.basename(k16.LIB_PATH) == "libk16_testing.so"
n = 9000
bases = ol.gen_points(0, 3, n)
sc = np_scalars(5, n, "full256")
want = ol.msm(0, bases, sc, nthreads=4)[1]
sm = k16.ShardedMsm([0, 0], k16.G1, n)
sm.set_bases(bases)
sm.set_piece_rows(512, 512)
assert sm.run(sc)[1] == want
for piece in (0, 2, 5):
    os.environ["K16_FAULT_INJECT"] = "shard_piece:%%d" %% piece
    if testing:
        try:
            sm.run(sc)
            raise AssertionError("the injected fault did not surface")
        except k16.K16Error as e:
            assert e.rc == -2 and "fault injected" in str(e), (e.rc, str(e))
    else:
        assert sm.run(sc)[1] == want
    del os.environ["K16_FAULT_INJECT"]
    for r in range(2):
        assert sm.shard_ctx(r).msm_pending() == 0
    assert sm.run(sc)[1] == want
sm.close()
print("shard fault child OK", "testing" if testing else "production")
''' % {"tests": os.path.join(ROOT, "tests"), "pkg": os.path.join(ROOT, "keyless-zk-proofs_amd")}
    for lib in ("libk16_testing.so", "libk16.so"):
        env = dict(os.environ, K16_LIB_PATH=os.path.join(ROOT, "keyless-zk-proofs_amd", lib))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0 and "shard fault child OK" in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]


def test_config5_2p26_eight_shards_closed_form():
    """BASELINE config 5 at its stated size in the form one box allows: ONE 2^26-point G1 MSM cut into EIGHT shards
    (k16_msm_sharded_*: eight contexts, on device 0 when the box has one GPU, spread over the devices otherwise), every shard
    makes ITS slice of the table on its own device ((lo + i + 1) * G), the 2 GiB of scalars arrive in host memory and are
    uploaded in pieces inside the run, host-side EC-add fold; result == (sum_i s_i (i + 1)) * G -- the closed form of the
    reference's own MSM test (alt_bn128_test.cpp:172-212) at the size of ParallelMultiexp's stress case
    (RS/multiexp.cpp:183-245)."""
    import bench
    import k16
    n = 1 << 26
    sm = k16.ShardedMsm(_devices(k16, 8), k16.G1, n)
    try:
        assert sm.count() == 8
        for r in range(8):
            lo, hi = sm.shard_range(r)
            assert hi - lo == 1 << 23
            d = sm.shard_ctx(r).synth_points(k16.G1, lo, hi - lo)
            sm.set_bases_device(r, d)
            d.free()
        scalars = bench.fast_scalars(n, seed=26)
        xyzz, aff = sm.run(scalars)
        ms = sm.last_ms()
        k = bench.weighted_sum_mod_r(scalars, 0)
        assert aff == bench.scalar_times_g(sm.shard_ctx(0), k16, k)
        assert 0 < ms["fold_ms"] < ms["total_ms"] < 60000
    finally:
        sm.close()
