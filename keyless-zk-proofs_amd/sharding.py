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


def exchange_and_fold(dist, group, partial_xyzz, device=None):
    """all_gather every rank's partial MSM result and fold them. Returns (xyzz_bytes, affine_bytes)."""
    import torch

    nbytes = k16.XYZZ_BYTES[group]
    assert len(partial_xyzz) == nbytes
    mine = torch.frombuffer(bytearray(partial_xyzz), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    world = dist.get_world_size()
    bufs = [torch.empty(nbytes, dtype=torch.uint8, device=mine.device) for _ in range(world)]
    dist.all_gather(bufs, mine)
    parts = torch.stack(bufs).cpu().numpy()
    return k16.points_sum(group, np.ascontiguousarray(parts))
