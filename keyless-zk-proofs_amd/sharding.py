"""Multi-GPU sharding of one MSM: one process per GPU, contiguous index ranges of the
(static) point table and of the scalar array per rank, one exchange of the tiny per-shard partial
results, then an EC-add fold.  No bucket-level data ever crosses xGMI.

RCCL has no elliptic-curve reduction op, so the exchange is an all_gather of raw XYZZ bytes
(128 B for G1, 256 B for G2 per rank) followed by k16_points_sum on every rank.
The same code runs over gloo on CPU tensors (tests/test_multi_process.py).
"""
import numpy as np

import k16


def shard_range(n, world, rank):
    """Contiguous [lo, hi) of rank's shard; the first n % world ranks get one extra element."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


_BUFS = {}


def exchange_and_fold(dist, group, partial_xyzz, device=None):
    """all_gather every rank's partial MSM result and fold them. Returns (xyzz_bytes, affine_bytes).
    Buffers are cached per (group, world, device): one pinned staging tensor, one send and one receive
    tensor, so a step costs two small copies and one all_gather_into_tensor."""
    import torch

    nbytes = k16.XYZZ_BYTES[group]
    assert len(partial_xyzz) == nbytes
    world = dist.get_world_size()
    key = (group, world, str(device))
    if key not in _BUFS:
        dev = torch.device(device) if device is not None else torch.device("cpu")
        stage = torch.empty(nbytes, dtype=torch.uint8)
        if dev.type == "cuda":
            stage = stage.pin_memory()
        _BUFS[key] = (stage, torch.empty(nbytes, dtype=torch.uint8, device=dev),
                      torch.empty(world * nbytes, dtype=torch.uint8, device=dev))
    stage, send, recv = _BUFS[key]
    stage.copy_(torch.frombuffer(bytearray(partial_xyzz), dtype=torch.uint8))
    send.copy_(stage, non_blocking=True)
    dist.all_gather_into_tensor(recv, send)
    parts = recv.cpu().numpy().reshape(world, nbytes)
    return k16.points_sum(group, np.ascontiguousarray(parts))
