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


def _train_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smpl_nerf_amd.trainer import DataParallelTrainer
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))

        class Pipe(torch.nn.Module):          # stands in for NerfPipeline: (rgb, rgb_fine, ...) from a batch list
            def forward(self, data):
                y = torch.sigmoid(net(data[0]))
                return y, y * 0.5

        g = torch.Generator().manual_seed(1)
        x, gt = torch.rand(64, 6, generator=g), torch.rand(64, 3, generator=g)
        b, e = sd.shard_range(64, world, rank)
        tr = DataParallelTrainer(Pipe(), [net], lr=1e-2, fused=False)
        for _ in range(3):
            tr.step([x[b:e], gt[b:e]])
        if rank == 0:
            q.put([p.detach().numpy().copy() for p in net.parameters()])   # by value
    finally:
        dist.destroy_process_group()


def test_data_parallel_steps_equal_single_process_on_the_concatenated_batch():
    from smpl_nerf_amd.trainer import DataParallelTrainer
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join()
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))

    class Pipe(torch.nn.Module):
        def forward(self, data):
            y = torch.sigmoid(net(data[0]))
            return y, y * 0.5

    g = torch.Generator().manual_seed(1)
    x, gt = torch.rand(64, 6, generator=g), torch.rand(64, 3, generator=g)
    tr = DataParallelTrainer(Pipe(), [net], lr=1e-2, fused=False)
    for _ in range(3):
        tr.step([x, gt])
    for a, b in zip(got, net.parameters()):
        assert torch.allclose(torch.from_numpy(a), b.detach(), rtol=1e-5, atol=1e-6)
