"""world_size-2 tests of the data-parallel plumbing on CPU (gloo): ray sharding, frame assembly,
max-over-ranks timing and the gradient all-reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smpl_nerf_amd import dist as sd


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 16384, 16385):
        for world in (1, 2, 3, 8):
            parts = [sd.shard_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rays = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)
        z = torch.arange(n * 4, dtype=torch.float32).reshape(n, 4)
        mine = sd.shard_rays([rays, z])
        b, e = sd.shard_range(n, world, rank)
        assert mine[0].shape[0] == e - b and torch.equal(mine[1], z[b:e])
        # "render" = a per-ray function; assembling the shards must reproduce the single-process result
        local = mine[0] * 2 + 1
        full = sd.gather_rows(local, n)
        assert torch.equal(full, rays * 2 + 1)
        assert sd.max_over_ranks(float(rank + 1)) == float(world)
        sd.barrier()
        # gradient all-reduce: mean over ranks of rank-dependent flat buffers
        g = torch.full((1220872,), float(rank), dtype=torch.float32)
        sd.allreduce_mean_(g)
        assert torch.allclose(g, torch.full_like(g, (world - 1) / 2.0))
        assert sd.shard_frames(world, rank) == rank
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_allreduce():
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 1001), nprocs=2, join=True)
