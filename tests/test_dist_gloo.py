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


def _ragged_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smpl_nerf_amd.trainer import DataParallelTrainer
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))

        class Pipe(torch.nn.Module):
            def forward(self, data):
                y = torch.sigmoid(net(data[0]))
                return y, y * 0.5

        g = torch.Generator().manual_seed(1)
        x, gt = torch.rand(53, 6, generator=g), torch.rand(53, 3, generator=g)
        b, e = (0, 48) if rank == 0 else (48, 53)          # unequal shards: RayBatchLoader's short last batch
        tr = DataParallelTrainer(Pipe(), [net], lr=1e-2, fused=False)
        assert tr._comm is False                            # (gloo: the ranks agreed on the three-call form at construction)
        for _ in range(3):
            tr.step([x[b:e], gt[b:e]])
        q.put((rank, [p.detach().numpy().copy() for p in net.parameters()]))
    finally:
        dist.destroy_process_group()


def test_data_parallel_steps_with_ragged_batches_per_rank():
    """RayBatchLoader allows unequal shards and a short last batch, so the ranks of a step may hold different ray
    counts.  The collective schedule must not depend on them (the GPU suite checks the in-library RCCL form with a recording
    communicator, tests/test_gpu_round6.py); here the torch.distributed form at world size 2 with 48 and 5 rays: no rank waits for
    a collective the other never issues, the replicas stay equal, and the step equals Adam on the mean of the ranks' gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get() for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for a, b in zip(got[0], got[1]):
        assert np.array_equal(a, b)                        # replicas in lock-step
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    g = torch.Generator().manual_seed(1)
    x, gt = torch.rand(53, 6, generator=g), torch.rand(53, 3, generator=g)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    mse = torch.nn.MSELoss()
    for _ in range(3):
        grads = []
        for b, e in ((0, 48), (48, 53)):
            net.zero_grad()
            y = torch.sigmoid(net(x[b:e]))
            (mse(y, gt[b:e]) + mse(y * 0.5, gt[b:e])).backward()
            grads.append([p.grad.clone() for p in net.parameters()])
        for p, g0, g1 in zip(net.parameters(), *grads):
            p.grad = (g0 + g1) / 2
        opt.step()
    for a, b in zip(got[0], net.parameters()):
        assert torch.allclose(torch.from_numpy(a), b.detach(), rtol=1e-5, atol=1e-6)


def _replica_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smpl_nerf_amd.trainer import DataParallelTrainer
        torch.manual_seed(100 + rank)                      # ranks build DIFFERENT replicas ...
        net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
        dead = torch.nn.Linear(4, 4)                        # ... and only rank 1 ever produces a gradient for `dead`

        class Pipe(torch.nn.Module):
            def forward(self, data):
                y = torch.sigmoid(net(data[0]))
                if rank == 1:
                    y = y + 0.0 * dead(data[0][:, :4]).sum()
                return y, y * 0.5

        tr = DataParallelTrainer(Pipe(), [net, dead], lr=1e-2, fused=False)
        start = [p.detach().clone() for p in tr.params]    # after the constructor's broadcast
        g = torch.Generator().manual_seed(1)
        x, gt = torch.rand(64, 6, generator=g), torch.rand(64, 3, generator=g)
        b, e = sd.shard_range(64, world, rank)
        for _ in range(2):
            tr.step([x[b:e], gt[b:e]])                     # must not hang although rank 0 has p.grad None for `dead`
        q.put((rank, [s.numpy() for s in start], [p.detach().numpy().copy() for p in tr.params]))
    finally:
        dist.destroy_process_group()


def test_trainer_broadcasts_the_replica_and_survives_rank_dependent_missing_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (s, e)) for r, s, e in (q.get(), q.get()))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)                        # same replica after construction (rank 0's)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)                        # and still in lock-step after two steps


def test_flat_parameter_vector_is_zero_copy_after_flattening():
    from smpl_nerf_amd.nets import RenderRayNet, flat_parameter_vector
    from smpl_nerf_amd.trainer import flatten_parameters_
    net = RenderRayNet(4, 128, 60, 24, skips=[2])
    before = [p.detach().clone() for p in net._ordered_params()]
    v0 = flat_parameter_vector(net._ordered_params())
    assert v0.data_ptr() != net._ordered_params()[0].data_ptr()          # separate tensors: a torch.cat copy
    flat, gflat, segs, order = flatten_parameters_([net])
    assert [id(p) for p in order] == [id(p) for p in net._ordered_params()]
    v1 = flat_parameter_vector(net._ordered_params())
    assert v1.data_ptr() == flat.data_ptr() == net._ordered_params()[0].data_ptr() and torch.equal(v0, v1)
    assert all(torch.equal(a, b) for a, b in zip(before, net._ordered_params()))
    assert segs == [(net, 0, v1.numel())] and gflat.shape == flat.shape
    assert list(net.state_dict().keys())[0] == "positions_pose_input.weight"   # the checkpoint contract is untouched


def test_flattened_buffer_order_drives_the_optimiser_and_the_gradient_views():
    """the gradient views must follow the buffer order (kernel-ordered parameters first), not
    m.parameters() order; a net listed twice gets one segment; a model whose kernel order differs from its registration
    order still lines up."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.trainer import DataParallelTrainer, flatten_parameters_

    class Reordered(torch.nn.Module):          # registration order (b, a) != kernel order (a, b)
        def __init__(self):
            super().__init__()
            self.b = torch.nn.Linear(3, 2)
            self.a = torch.nn.Linear(2, 3)

        def _ordered_params(self):
            return [self.a.weight, self.a.bias, self.b.weight, self.b.bias]

    m = Reordered()
    flat, gflat, segs, order = flatten_parameters_([m])
    assert [id(p) for p in order] == [id(p) for p in m._ordered_params()] and segs == [(m, 0, flat.numel())]
    tr = DataParallelTrainer(None, [m], fused=False)
    assert [id(p) for p in tr.params] == [id(p) for p in m._ordered_params()]
    off = 0
    for p, v in zip(tr.params, tr._views):     # view i is exactly parameter i's slice of the flat gradient buffer
        assert v.shape == p.shape and v.data_ptr() == tr._flat_g.data_ptr() + 4 * off
        assert p.data_ptr() == tr._flat_p.data_ptr() + 4 * off
        off += p.numel()
    net = RenderRayNet(4, 128, 60, 24, skips=[2])
    flat, gflat, segs, order = flatten_parameters_([net, net])       # coarse is fine
    assert len(segs) == 1 and segs[0][2] == flat.numel() == sum(p.numel() for p in net.parameters())


def test_frames_are_sharded_by_image_across_ranks():
    """the data set is sharded by image - the union over ranks covers every frame exactly once, no frame is
    held twice, counts differ by at most one; RayGenerator.for_rank moves only this rank's frames."""
    from smpl_nerf_amd import dist as sdist
    from smpl_nerf_amd.raygen import RayGenerator
    for n_frames, world in ((1200, 8), (10, 4), (7, 8), (1, 2)):
        shards = [sdist.shard_frame_indices(n_frames, world, r) for r in range(world)]
        assert sorted(i for s in shards for i in s) == list(range(n_frames))
        assert max(map(len, shards)) - min(map(len, shards)) <= 1
    assert len(sdist.shard_frame_indices(1200, 8, 3)) == 150
    poses = np.tile(np.eye(4)[None], (10, 1, 1))
    poses[:, 0, 3] = np.arange(10)
    images = np.zeros((10, 4, 4, 3), np.float32) + np.arange(10, dtype=np.float32)[:, None, None, None]
    seen = []
    for r in range(4):
        g = RayGenerator.for_rank(poses, 4, 4, np.pi / 3, 1.0, 4.0, 8, "cpu", images=images, world=4, rank=r)
        assert g.n_frames == len(g.frame_ids) and g.images.shape == (g.n_frames * 16, 3)
        assert [float(v) for v in g.poses[:, 0, 3]] == [float(i) for i in g.frame_ids]      # the right poses ...
        assert [float(v) for v in g.images.view(g.n_frames, 16, 3)[:, 0, 0]] == [float(i) for i in g.frame_ids]  # ... and images
        seen += g.frame_ids
    assert sorted(seen) == list(range(10))


def _loader_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smpl_nerf_amd.trainer import RayBatchLoader

        class Gen:      # what the loader reads of a RayGenerator: this rank's ray count and device
            n_rays = 1000 if rank == 0 else 1300
            device = torch.device("cpu")
        loader = RayBatchLoader(Gen(), 256, iterations=None, seed=1, shuffle=True)
        q.put((rank, len(loader), sd.min_over_ranks(float(rank + 5))))
    finally:
        dist.destroy_process_group()


def test_shuffled_loader_runs_the_same_number_of_steps_on_every_rank():
    """image shards of different sizes (1000 / 1300 rays -> 4 / 6 batches of 256) would leave the ranks with different
    numbers of steps and the per-step gradient all-reduce would hang: the loader takes the minimum over the ranks."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_loader_worker, args=(2, port, q), nprocs=2, join=True)
    got = sorted(q.get() for _ in range(2))
    assert [g[1] for g in got] == [4, 4] and [g[2] for g in got] == [5.0, 5.0]
