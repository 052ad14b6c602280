"""-m gpu : libk16.so's one-process-per-GPU exchange (include/k16.h k16_rank_comm_*: ncclCommInitRank + ONE ncclAllGather per
MSM + the EC-add fold in rank order; SURVEY 8(e), BASELINE config 5) at WORLD SIZE 2 AND 3 ON A ONE-GPU BOX.

Real RCCL refuses two ranks on one device, and the pool has one-GPU boxes, so until round 6 nothing had ever executed the
library's gather + world-strided fold with more than one rank (missing #2).  Here the ranks are separate processes
(tests/rank_comm_child.py, no torch) that share GPU 0 and load a TEST DOUBLE of RCCL (tests/cpp/fake_rccl.cpp: shared memory +
hipMemcpy) through K16_RCCL_LIB -- the library's code (id hand-off, communicator, gather, bounded wait, abort, fold) is the
production code; only the fabric is faked.  The folded results are compared with the oracle's MSM over ALL rows
(RS/multiexp.cpp:183-245).  The real-RCCL two-GPU tests stay in test_gpu_multirank.py."""
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_lib as ol
from gpu_common import np_scalars

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "rank_comm_child.py")


def build_fake_rccl(dirpath):
    """tests/cpp/fake_rccl.cpp -> <dir>/librccl.so.1 (never part of the product: tests/test_boundary.py)"""
    out = os.path.join(str(dirpath), "librccl.so.1")
    if not os.path.exists(out):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lrt",
                               "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


@pytest.fixture(scope="module")
def fake(tmp_path_factory):
    return build_fake_rccl(tmp_path_factory.mktemp("fake_rccl"))


def _spawn(fake, tmp_path, world, mode, extra_env=None, args=()):
    idfile = str(tmp_path / ("id_%d_%s_%d" % (world, mode, time.time_ns())))
    env = dict(os.environ, K16_RCCL_LIB=fake, FAKE_RCCL_TIMEOUT_MS="30000")
    env.update(extra_env or {})
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, CHILD, str(r), str(world), idfile, mode] + [str(a) for a in args],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=600)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in glob.glob("/dev/shm/k16_fake_rccl_*"):     # (a rank that died before the segment was unlinked)
            try:
                os.unlink(f)
            except OSError:
                pass
    return outs


@pytest.mark.parametrize("world,async_mode", [(2, "0"), (3, "0"), (2, "1")])
def test_rank_comm_fold_at_world_2_and_3_equals_the_oracle(fake, tmp_path, world, async_mode):
    n, seed = 3000, 5
    outs = _spawn(fake, tmp_path, world, "msm", {"FAKE_RCCL_ASYNC": async_mode}, (n, seed))
    for rc, o, e in outs:
        assert rc == 0, e[-3000:]
    res = [json.loads([l for l in o.splitlines() if l.startswith("{")][-1]) for _, o, _ in outs]
    assert sorted(r["rank"] for r in res) == list(range(world))
    want = []
    for group in (0, 1):
        nn = n if group == 0 else max(n // 4, 8)
        bases = ol.gen_points(group, 3, nn)
        for rnd in range(3):
            sc = np_scalars(seed + 17 * rnd + group, nn, "full256" if rnd != 1 else "witness")
            want.append(ol.msm(group, bases, sc, nthreads=4)[1].hex())
    for r in res:                       # EVERY rank folded the same, complete result
        assert r["results"] == want, r["rank"]


@pytest.mark.parametrize("async_mode", ["0", "1"])
def test_a_rank_that_dies_does_not_hang_the_others(fake, tmp_path, async_mode):
    """rank 1 leaves after ncclCommInitRank; rank 0's gather must come back with K16_ERR_HIP after the bounded wait (the
    double's own in blocking mode; the LIBRARY's K16_RANK_COMM_TIMEOUT_MS + ncclCommAbort when the collective is enqueued
    like the real one), and a second call on the dead communicator fails at once."""
    t0 = time.time()
    outs = _spawn(fake, tmp_path, 2, "die", {"FAKE_RCCL_ASYNC": async_mode, "FAKE_RCCL_TIMEOUT_MS": "1500",
                                            "K16_RANK_COMM_TIMEOUT_MS": "1500"})
    assert outs[0][0] == 0, outs[0][2][-3000:]
    d = json.loads([l for l in outs[0][1].splitlines() if l.startswith("{")][-1])
    rc1, t1, rc2, t2 = d["errs"]
    assert rc1 == -2 and rc2 == -2, d
    assert 1.0 < t1 < 20.0 and t2 - t1 < 0.5, d           # bounded, and the second call does not wait again
    assert "rank" in d["msg"] or "arrive" in d["msg"], d
    assert time.time() - t0 < 120


def test_loader_reports_a_missing_library_instead_of_crashing(tmp_path):
    """with no loadable RCCL, k16_rank_comm_unique_id / _create must return K16_ERR_NO_DEVICE with a message
    (the old loader passed a NULL from a second dlerror() call to std::string and crashed the process)."""
    code = ("import sys; sys.path.insert(0, %r); import k16\n"
            "try:\n    k16.RankComm.unique_id()\n    print('LOADED')\n"
            "except k16.K16Error as e:\n    print('RC', e.rc, (k16.load().k16_rank_comm_load_error() or b'').decode())\n"
            % os.path.join(ROOT, "keyless-zk-proofs_amd"))
    env = dict(os.environ, K16_RCCL_LIB=str(tmp_path / "no_such_librccl.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("RC -1 dlopen librccl:") and "no_such_librccl" in out.stdout, out.stdout


def test_bench_strong_mode_two_ranks_through_the_c_exchange_over_the_double(fake):
    """bench.py --mode strong --gpus 2 with the library's own exchange (k16_rank_comm_*) carrying the partials between two
    REAL processes: both ranks on GPU 0 (K16_BENCH_SHARE_GPU=1; torch.distributed over gloo is only the launcher's store
    here), RCCL = the double.  Closed form checked over both shards."""
    env = dict(os.environ, K16_BENCH_SHARE_GPU="1", K16_BENCH_PREWARM="1", K16_BENCH_EXCHANGE="c", K16_RCCL_LIB=fake,
               K16_BENCH_NO_CONFIG_LEGS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--mode",
                          "strong", "--total-log2n", "20", "--proofs", "0", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ranks_seen"] == 2 and d["result_checked"] is True and d["scaling"] == "strong"
    assert "k16_rank_comm" in d["config"]["sharding"], d["config"]


@pytest.mark.parametrize("async_mode", ["0", "1"])
def test_bench_weak_mode_two_ranks_overlap_their_exchanges_through_the_c_leg(fake, async_mode):
    """bench.py --gpus 2 in its DEFAULT (weak) mode: every rank owns a 2^16-point shard, each step's partial goes out with
    k16_rank_comm_allgather_start while the next steps' MSMs run and comes back folded with _finish (up to four exchanges in
    flight) -- the path the driver's scaling runs take at N > 1, here over the test double with both ranks on GPU 0.  The last
    step's folded result equals the closed form over BOTH shards."""
    env = dict(os.environ, K16_BENCH_SHARE_GPU="1", K16_BENCH_PREWARM="2", K16_RCCL_LIB=fake, K16_BENCH_NO_CONFIG_LEGS="1",
               FAKE_RCCL_ASYNC=async_mode)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "K16_BENCH_EXCHANGE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--log2n", "16",
                          "--proofs", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ranks_seen"] == 2 and d["result_checked"] is True and d["scaling"] == "weak"
    assert "k16_rank_comm" in d["config"]["sharding"] and d["config"]["exchange_note"] is None, d["config"]


def test_bench_eight_ranks_rehearsal_over_the_double(fake):
    """The shape of the driver's scaling run at N = 8 -- `bench.py --gpus 8`, default mode, default legs -- with all eight ranks on
    GPU 0 and the RCCL double as the fabric, at reduced sizes (a 5 % circuit, a 10 % wave key, a 2^22-point sharded MSM) so that it
    fits the suite's time budget: weak-mode exchange through k16_rank_comm_allgather_start / _finish at world 8, replica proofs on
    eight ranks, the 64-proof wave spread over them and verified in one batch on rank 0, ONE MSM as eight one-shard ranks + the
    library's gather and fold.  (The full-size rehearsal: profiles/r06/bench_8ranks_sharing_one_gpu_default_command.json.)"""
    env = dict(os.environ, K16_BENCH_SHARE_GPU="1", K16_BENCH_PREWARM="1", K16_RCCL_LIB=fake, K16_BENCH_WAVE_SCALE="0.1",
               K16_BENCH_2P26_LOG2N="22", K16_BENCH_PROVERS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "K16_BENCH_EXCHANGE", "K16_BENCH_NO_CONFIG_LEGS"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--log2n", "16",
                          "--proofs", "2", "--proof-scale", "0.05", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["result_checked"] is True
    assert "k16_rank_comm" in d["config"]["sharding"] and d["config"]["exchange_note"] is None
    assert d["proof"]["proofs"] == 16 and d["proof"]["parallelism"].startswith("replicas")
    w, s5 = d["config4_wave"], d["strong_2p22"]
    assert w["ranks_seen"] == 8 and w["proofs"] == 64 and w["all_accepted"] and w["wrong_inputs_rejected"] and w["distinct"]
    assert s5["ranks_seen"] == 8 and s5["shards"] == 8 and s5["result_checked"] is True and "k16_rank_comm_allgather_fold" in s5["entry"]
