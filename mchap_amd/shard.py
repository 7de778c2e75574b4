"""Sharding of a (locus x sample) unit list over the GPUs of a node: one process per GPU.

Units are independent (the reference parallelises loci over processes, application/baseclass.py:360-388) and every
unit's RNG stream is keyed by its GLOBAL index, so results are identical for any world size.  There is no data-path
collective; the only exchange is the gather of fixed-size result records (RCCL `all_gather` over xGMI, gloo on CPU).
"""
import numpy as np


def shard_range(n_units, rank, world):
    """Contiguous, balanced block partition: [start, stop) of `rank` (numpy.array_split boundaries)."""
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_records(local, n_units, dist=None):
    """Gather per-unit fixed-size records (torch tensor [n_local, ...]) from every rank into global unit order.

    Ranks may own different numbers of units; shards are padded to the largest before `all_gather`."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_range(n_units, r, world) for r in range(world)]
    biggest = max(b - a for a, b in sizes)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)], dim=0)
