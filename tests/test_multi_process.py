"""N > 1 path on CPU: world_size-2 gloo run of the shard -> exchange -> fold logic that bench.py and a
multi-GPU deployment use (keyless-zk-proofs_amd/sharding.py).  The per-shard partial MSMs come from the
oracle here (no GPU in this container); the exchange and the EC-add fold are the product code."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, group, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (HERE, os.path.join(ROOT, "keyless-zk-proofs_amd")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_lib as ol
    import sharding
    from gpu_common import np_scalars
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bases = ol.gen_points(group, 0, n)
    scalars = np_scalars(1234, n, "full256")
    lo, hi = sharding.shard_range(n, world, rank)
    part, _ = ol.msm(group, bases[lo:hi], scalars[lo:hi])
    xyzz, aff = sharding.exchange_and_fold(dist, group, part)
    _, want = ol.msm(group, bases, scalars)
    q.put((rank, aff == want, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("group,n", [(0, 1001), (1, 130), (0, 1)])
def test_sharded_msm_gloo_world2(group, n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, group, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    ranges = sorted((lo, hi) for _, _, lo, hi in res)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == n


def test_shard_range_partition():
    sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
    import sharding
    for n in (0, 1, 7, 8, 1 << 20, (1 << 26) + 3):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = sharding.shard_range(n, world, r)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == n


def _worker_pipelined(rank, world, port, q):
    """Three MSMs whose exchanges are in flight together (bench.py overlaps step k's exchange with step k+1)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (HERE, os.path.join(ROOT, "keyless-zk-proofs_amd")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_lib as ol
    import sharding
    from gpu_common import np_scalars
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 300
    bases = ol.gen_points(0, 0, n)
    lo, hi = sharding.shard_range(n, world, rank)
    handles, wants = [], []
    for step in range(3):
        scalars = np_scalars(900 + step, n, "uniform")
        part, _ = ol.msm(0, bases[lo:hi], scalars[lo:hi])
        handles.append(sharding.exchange_start(dist, 0, part))
        wants.append(ol.msm(0, bases, scalars)[1])
    ok = all(sharding.exchange_finish(h)[1] == w for h, w in zip(handles, wants))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_exchanges_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _worker_exchange_fails_everywhere(rank, world, port, q):
    """a RankExchange whose id cannot be made on rank 0 (no loadable RCCL) must fail on EVERY rank, after the
    broadcast, so that the fallback agreement that follows (bench.py: all_reduce(MIN)) is reached by all of them."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      K16_RCCL_LIB="/nonexistent/librccl.so.1")
    for p in (HERE, os.path.join(ROOT, "keyless-zk-proofs_amd")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    msg = None
    try:
        sharding.RankExchange(dist, None)
    except RuntimeError as e:
        msg = str(e)
    ok = torch.tensor([0 if msg else 1])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)       # the agreement bench.py makes next: must not hang or mismatch
    q.put((rank, msg, int(ok.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_exchange_fails_on_all_ranks_or_none_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange_fails_everywhere, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, msg, agreed in res:
        assert msg and "rank 0 could not create the communicator id" in msg and "dlopen librccl" in msg, (rank, msg)
        assert agreed == 0
