"""Multi-GPU sharding of one MSM: one process per GPU, contiguous index ranges of the
(static) point table and of the scalar array per rank, one exchange of the tiny per-shard partial
results, then an EC-add fold.  No bucket-level data ever crosses xGMI.

RCCL has no elliptic-curve reduction op, so the exchange is an all_gather of raw XYZZ bytes
(128 B for G1, 256 B for G2 per rank) followed by k16_points_sum on every rank.

Two carriers of the same exchange:
  * `RankExchange` -- the library's own C entry points (include/k16.h: k16_rank_comm_*: ncclCommInitRank + ONE ncclAllGather
    per MSM, RCCL dlopen'ed by libk16.so), what a C++ service with one prover process per GPU calls; this module is a ctypes
    caller of it.  The 128-byte communicator id travels through the launcher's store (torch.distributed here).
  * `exchange_start` / `exchange_finish` -- torch.distributed's all_gather_into_tensor, asynchronous, which bench.py's weak
    mode overlaps with the next step's GPU work; the same code runs over gloo on CPU tensors (tests/test_multi_process.py).
One process driving several devices needs neither: k16.ShardedMsm (k16_msm_sharded_*) folds on the host.
"""
import numpy as np

import k16


def shard_range(n, world, rank):
    """Contiguous [lo, hi) of rank's shard; the first n % world ranks get one extra element."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


_BUFS = {}
_SLOTS = 4  # exchanges that may be in flight at once (bench.py overlaps one with the next step's GPU work)


def exchange_start(dist, group, partial_xyzz, device=None):
    """Begin the exchange of this rank's partial result: copy to the device and issue an asynchronous
    all_gather_into_tensor.  Returns a handle for exchange_finish.  Buffers are cached per (group, world, device)
    in a small ring: one pinned staging tensor, one send and one receive tensor per slot."""
    import torch

    nbytes = k16.XYZZ_BYTES[group]
    assert len(partial_xyzz) == nbytes
    world = dist.get_world_size()
    key = (group, world, str(device))
    if key not in _BUFS:
        dev = torch.device(device) if device is not None else torch.device("cpu")
        ring = []
        for _ in range(_SLOTS):
            stage = torch.empty(nbytes, dtype=torch.uint8)
            back = torch.empty(world * nbytes, dtype=torch.uint8)
            if dev.type == "cuda":
                stage = stage.pin_memory()
                back = back.pin_memory()
            ring.append((stage, torch.empty(nbytes, dtype=torch.uint8, device=dev),
                         torch.empty(world * nbytes, dtype=torch.uint8, device=dev), back))
        _BUFS[key] = [ring, 0]
    ring, nxt = _BUFS[key]
    _BUFS[key][1] = (nxt + 1) % _SLOTS
    stage, send, recv, back = ring[nxt]
    stage.copy_(torch.frombuffer(bytearray(partial_xyzz), dtype=torch.uint8))
    send.copy_(stage, non_blocking=True)
    work = dist.all_gather_into_tensor(recv, send, async_op=True)
    done = None
    if recv.is_cuda:
        # the download is queued right behind the collective (stream-ordered): exchange_finish only waits for an event
        work.wait()                       # the current stream waits for the collective; the host does not
        back.copy_(recv, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
    return (work, recv, back, done, group, world, nbytes)


def exchange_finish(handle):
    """Wait for the all_gather and fold the shards' partial results with EC adds. Returns (xyzz_bytes, affine_bytes)."""
    work, recv, back, done, group, world, nbytes = handle
    if done is not None:
        done.synchronize()
        parts = back.numpy().reshape(world, nbytes)
    else:
        work.wait()
        parts = recv.numpy().reshape(world, nbytes)
    return k16.points_sum(group, np.ascontiguousarray(parts))


def exchange_and_fold(dist, group, partial_xyzz, device=None):
    """all_gather every rank's partial MSM result and fold them. Returns (xyzz_bytes, affine_bytes)."""
    return exchange_finish(exchange_start(dist, group, partial_xyzz, device))


class RankExchange:
    """The exchange through libk16.so's RCCL leg (k16_rank_comm_create / _allgather_fold).  `dist` is only the launcher's
    store here: it carries rank 0's communicator id to the other ranks; the collective itself is the library's."""

    def __init__(self, dist, ctx):
        """Collective, and it fails on ALL ranks or on none: rank 0's failure to make the id travels in the
        broadcast itself, and the outcome of ncclCommInitRank is agreed on with an all_gather before anybody returns -- a rank
        that raised on its own would leave the others inside the next collective."""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [None]
        if rank == 0:
            try:
                box = [("id", k16.RankComm.unique_id())]
            except Exception as e:               # noqa: BLE001 -- whatever it is, the other ranks must hear of it
                box = [("error", repr(e))]
        dist.broadcast_object_list(box, src=0)
        kind, payload = box[0]
        if kind != "id":
            raise RuntimeError("rank 0 could not create the communicator id: %s" % payload)
        self.comm, err = None, None
        try:
            self.comm = k16.RankComm(ctx, rank, world, payload)
        except Exception as e:                   # noqa: BLE001
            err = repr(e)
        errs = [None] * world
        dist.all_gather_object(errs, err)
        if any(errs):
            if self.comm is not None:
                self.comm.close()
                self.comm = None
            raise RuntimeError("k16_rank_comm_create failed on rank(s) %s: %s"
                               % ([r for r, e in enumerate(errs) if e], next(e for e in errs if e)))
        self.world = world

    def exchange_and_fold(self, group, partial_xyzz):
        """all ranks' partial results gathered with ONE ncclAllGather and folded in rank order: (xyzz_bytes, affine_bytes)"""
        return self.comm.allgather_fold(group, partial_xyzz)

    def start(self, group, partial_xyzz):
        """enqueue one exchange (k16_rank_comm_allgather_start; up to 4 in flight) -- nothing waits"""
        self.comm.allgather_start(group, partial_xyzz)

    def finish(self):
        """complete the oldest exchange in flight and fold it: (xyzz_bytes, affine_bytes)"""
        return self.comm.allgather_finish()

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
