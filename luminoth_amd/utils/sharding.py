"""Data-parallel sharding helpers (per-image independent units, one process per GPU).

The reference's workers each pull from their own input queue (train.py:46-59, asynchronous parameter servers); under
synchronous data parallelism every rank must see DIFFERENT records and the SAME number of steps (a rank that ran one
step more would wait forever in its all-reduce)."""
import numpy as np


def rank_world():
    """(rank, world_size) of the initialised torch.distributed process group, else (0, 1)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def shared_seed(seed):
    """A seed every rank agrees on: `seed` itself when given, else rank 0's random draw broadcast to all."""
    rank, world = rank_world()
    if seed is not None or world == 1:
        return seed
    import torch
    import torch.distributed as dist
    box = [int(np.random.SeedSequence().entropy % (2 ** 31)) if rank == 0 else None]
    if dist.get_backend() == 'nccl':
        t = torch.tensor([box[0] or 0], dtype=torch.int64, device='cuda')
        dist.broadcast(t, 0)
        return int(t[0])
    dist.broadcast_object_list(box, 0)
    return int(box[0])


def shard_order(order, rank, world):
    """Records of one epoch for `rank`: every world-th element of the (identically permuted) order, trimmed to the
    common length so that all ranks run the same number of steps."""
    order = list(order)
    per = len(order) // world
    return order[rank::world][:per]
