"""world_size-2 gloo tests of the multi-GPU host logic (sharding + record gather).  CPU only."""
import os
import socket

import numpy as np
import pytest

from mchap_amd.shard import shard_range


def test_shard_range_is_a_contiguous_partition():
    for n in (0, 1, 7, 10000, 100003):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_units, out):
    import torch
    import torch.distributed as dist
    from mchap_amd.shard import gather_records, shard_range
    from mchap_amd.synth import synth_units

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_range(n_units, rank, world)
    # every rank builds its own shard; generation is keyed by the global unit id
    reads, _, truth = synth_units(b - a, n_pos=4, n_reads=6, first_unit=a)
    rec = torch.from_numpy(np.concatenate([truth.reshape(b - a, -1).astype(np.float64), reads.reshape(b - a, -1)[:, :3]], axis=1))
    full = gather_records(rec, n_units, dist)
    if rank == 0:
        np.save(out, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    from mchap_amd.synth import synth_units

    n_units = 7  # uneven split: 4 + 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, port, n_units, out), nprocs=2, join=True)
    got = np.load(out)
    reads, _, truth = synth_units(n_units, n_pos=4, n_reads=6, first_unit=0)
    expect = np.concatenate([truth.reshape(n_units, -1).astype(np.float64), reads.reshape(n_units, -1)[:, :3]], axis=1)
    assert np.array_equal(got, expect, equal_nan=True)
