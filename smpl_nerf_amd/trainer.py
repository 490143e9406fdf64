"""Data-parallel training step for the ray-march path - what NerfSolver.train does per batch
(solver/nerf_solver.py:76-87) with the pieces the reference lacks for multi-GPU:

    rgb, rgb_fine, ... = pipeline(batch)                       # HIP forward (activations saved)
    loss = MSE(rgb, gt) + MSE(rgb_fine, gt)                     # solver/nerf_solver.py:48-52
    loss.backward()                                             # HIP backward kernels
    all-reduce(mean) of ONE flat gradient buffer over RCCL      # the only collective of the path
    Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay)        # solver/nerf_solver.py:11-14, 31-33

Every rank holds a replica of the weights and draws its own rays (rays are independent); the loss of
the global batch is the mean of the per-rank losses, so averaging the gradients reproduces the
single-process gradient of the concatenated batch.
"""
from __future__ import annotations

import os
import time

import torch

from . import dist as sdist
from . import io as sio


def flatten_parameters_(models):
    """Moves every parameter of `models` into ONE contiguous fp32 buffer (values kept; `p.data` becomes a view of it)
    and returns (flat_params, flat_grads, segments, order) with segments = [(model, offset, numel of its kernel-ordered
    parameters)] and order = the parameters in buffer order.  Per model the kernel-ordered parameters
    (`_ordered_params()` / `_params()`: the C-ABI's params_flat order) come first, so a net packs its weight streams from
    its segment without a torch.cat (nets.flat_parameter_vector) and the backward kernels can write its gradients into the
    matching segment of `flat_grads` (set_grad_sink).  A model listed twice, or one whose kernel-ordered parameters are
    shared with an earlier model, gets no segment (its gradients then take the copying path of sync_gradients)."""
    order, segments, seen, with_segment = [], [], set(), set()
    for m in models:
        first = m._ordered_params() if hasattr(m, "_ordered_params") else (m._params() if hasattr(m, "_params") else [])
        off = sum(p.numel() for p in order)
        own = [p for p in list(first) + list(m.parameters()) if not (id(p) in seen or seen.add(id(p)))]
        order += own
        if first and id(m) not in with_segment and len(own) >= len(first) and all(a is b for a, b in zip(first, own)):
            segments.append((m, off, sum(p.numel() for p in first)))
            with_segment.add(id(m))
    if not order:
        return None, None, [], []
    dev = order[0].device
    if any(p.device != dev or p.dtype != torch.float32 for p in order):
        return None, None, [], order   # mixed devices / dtypes: keep the parameters where they are
    flat = torch.empty(sum(p.numel() for p in order), dtype=torch.float32, device=dev)
    off = 0
    with torch.no_grad():
        for p in order:
            v = flat[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            off += p.numel()
    return flat, torch.zeros_like(flat), segments, order


class DataParallelTrainer:
    default_adam_args = {"lr": 1e-4, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0}  # nerf_solver.py:11-14

    def __init__(self, pipeline, models, lr: float = 5e-4, weight_decay: float = 0.0, loss_func=None, fused=None):
        self.pipeline = pipeline
        self.models = list(models)
        self.world, self.rank = sdist.world_rank()
        # one flat parameter buffer and one flat gradient buffer for all nets (what is all-reduced is the flat
        # gradient the backward kernels wrote)
        self._flat_p, self._flat_g, self._segments, order = flatten_parameters_(self.models)
        # the optimiser's parameter list IS the buffer order (kernel-ordered parameters of a model first), so that the
        # gradient views below line up with the sinks whatever a model's registration order is
        self.params = order
        self._views = None
        if self._flat_g is not None:
            off, self._views = 0, []
            for p in self.params:
                self._views.append(self._flat_g[off:off + p.numel()].view(p.shape))
                off += p.numel()
        # every rank must start from the same replica (DDP broadcasts at construction; so does this)
        if self.world > 1:
            if self._flat_p is not None:
                sdist.broadcast_(self._flat_p, 0)
            else:
                for p in self.params:
                    sdist.broadcast_(p.data, 0)
            for m in self.models:
                for buf in m.buffers():
                    sdist.broadcast_(buf, 0)
                if hasattr(m, "mark_weights_changed"):
                    m.mark_weights_changed()
        args = dict(self.default_adam_args)
        args.update({"lr": lr, "weight_decay": weight_decay})
        if fused is None:
            fused = all(p.is_cuda for p in self.params)
        self.optim = torch.optim.Adam(self.params, fused=fused, **args) if fused else torch.optim.Adam(self.params, **args)
        self.loss_func = loss_func or torch.nn.MSELoss()

    def loss(self, rgb, rgb_fine, rgb_truth):
        return self.loss_func(rgb, rgb_truth) + self.loss_func(rgb_fine, rgb_truth)  # nerf_solver.py:48-52

    def _arm_grad_sinks(self):
        if self._flat_g is None:
            return
        for m, off, n in self._segments:
            if hasattr(m, "set_grad_sink"):
                m.set_grad_sink(self._flat_g[off:off + n])

    def sync_gradients(self):
        """Mean over ranks of all parameter gradients through ONE flat fp32 buffer (one collective).  The buffer covers
        every parameter on every rank - a parameter without a gradient on this rank contributes zeros - so the
        collective has the same size everywhere.  Gradients the backward kernels already wrote into the buffer (the
        grad sinks) are not copied; afterwards every p.grad is a view of the buffer.

        Like torch's DistributedDataParallel, a parameter that took part in no rank's step (the fine net with
        run_fine = 0, estimator heads) comes out with a ZERO gradient rather than None when world > 1: Adam then still
        decays its moments (and applies weight_decay), whereas a single process skips it.  The reference has no multi-GPU
        path to compare with; the multi-rank path keeps the collective's shape fixed instead of exchanging a has-grad
        mask (which would put a device -> host read into every step)."""
        if self.world == 1:
            return
        if self._flat_g is None:       # parameters could not be flattened: per-tensor fallback, fixed list, zeros for None
            for p in self.params:
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                sdist.allreduce_mean_(g)
                p.grad = g
            return
        for p, v in zip(self.params, self._views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
        sdist.allreduce_mean_(self._flat_g)
        for p, v in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def step(self, batch):
        """One optimisation step on this rank's batch (list of tensors, rgb_truth last). Returns the
        local loss tensor (not synchronised with the host)."""
        self.optim.zero_grad(set_to_none=True)
        self._arm_grad_sinks()
        out = self.pipeline(batch)
        loss = self.loss(out[0], out[1], batch[-1])
        loss.backward()
        self.sync_gradients()
        self.optim.step()
        # fused optimisers update the parameters without bumping autograd's version counters, which the nets' packed
        # weight caches key on: tell them
        for m in self.models:
            if hasattr(m, "mark_weights_changed"):
                m.mark_weights_changed()
        return loss.detach()

    # ------------------------------------------------------------------ the Solver loop (solver/nerf_solver.py:54-163)
    @torch.no_grad()
    def validate(self, val_loader, h: int = 0, w: int = 0):
        """Validation pass of NerfSolver.train (:107-150) under no_grad: mean loss over the loader, the re-rendered
        frames (rgb_fine reshaped to [-1, h, w, 3] when whole frames were covered, :143-145) and their PSNR
        (util/scores.py:47-48).  Returns (val_loss, psnr or None, frames or None)."""
        for m in self.models:
            m.eval()
        total, count, renders, truths = 0.0, 0, [], []
        for data in val_loader:
            out = self.pipeline(data)
            total += float(self.loss(out[0], out[1], data[-1]))
            count += 1
            renders.append(out[1].detach())
            truths.append(data[-1].detach())
        val_loss = total / (count or 1)                                   # :152 (guards the empty loader the same way)
        frames = psnr = None
        if count and h and w:
            r, t = torch.cat(renders).cpu().numpy(), torch.cat(truths).cpu().numpy()
            if r.shape[0] % (h * w) == 0:
                frames = r.reshape(-1, h, w, 3)
                psnr = sio.img2psnr(frames, t.reshape(-1, h, w, 3))
        return val_loss, psnr, frames

    def save_checkpoint(self, save_dir: str, model_names, epoch: int, history=None):
        """utils.save_run's per-model state_dict files (utils.py:282-283), plus what the reference does not keep: the
        optimiser state and the epoch counter, so that a run can be resumed.  Rank 0 writes."""
        if self.rank != 0:
            return
        sio.save_run(save_dir, self.models, model_names)
        torch.save({"optim": self.optim.state_dict(), "epoch": int(epoch), "history": history or {}},
                   os.path.join(save_dir, "trainer_state.pt"))

    def load_checkpoint(self, load_dir: str, model_names) -> int:
        """Restores models and optimiser; returns the number of finished epochs."""
        dev = self.params[0].device
        sio.load_run(load_dir, self.models, model_names, map_location=dev)
        state = torch.load(os.path.join(load_dir, "trainer_state.pt"), map_location=dev)
        self.optim.load_state_dict(state["optim"])
        return int(state["epoch"])

    def fit(self, train_loader, val_loader=(), num_epochs: int = 1, h: int = 0, w: int = 0, save_dir: str = None,
            model_names=("model_coarse.pt", "model_fine.pt"), log_iterations: int = 0, start_epoch: int = 0, log=print):
        """NerfSolver.train (solver/nerf_solver.py:54-163) on this trainer: per epoch - the training batches of
        `train_loader` (any iterable of pipeline input lists, rgb_truth last; tensors are moved to the models' device
        like :78-79), the average training loss (:105), a no_grad validation pass with re-rendered frames and PSNR, a
        checkpoint per epoch (:160-161).  Adds throughput (rays/s over the epoch's training steps, all ranks) to the
        log.  Returns the history dict."""
        dev = self.params[0].device
        hist = {"train_loss": [], "val_loss": [], "val_psnr": [], "rays_per_s": []}
        for epoch in range(start_epoch, num_epochs):
            for m in self.models:
                m.train()
            losses, rays = [], 0
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i, data in enumerate(train_loader):
                data = [e.to(dev) for e in data]
                losses.append(self.step(data))
                rays += int(data[-1].shape[0])
                if log_iterations and i % log_iterations == log_iterations - 1 and self.rank == 0:
                    log("[Epoch %d, Iteration %5d] TRAIN loss: %.7f" % (epoch + 1, i + 1, float(losses[-1])))
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            dt = max(time.perf_counter() - t0, 1e-9)
            train_loss = float(torch.stack(losses).mean()) if losses else 0.0
            val_loss, psnr, _ = self.validate([[e.to(dev) for e in d] for d in val_loader], h, w)
            hist["train_loss"].append(train_loss)
            hist["val_loss"].append(val_loss)
            hist["val_psnr"].append(psnr)
            hist["rays_per_s"].append(self.world * rays / dt)
            if self.rank == 0:
                log("[Epoch %d] Average loss of Epoch: %.7f | VAL loss: %.7f%s | %.3e rays/s" %
                    (epoch + 1, train_loss, val_loss, "" if psnr is None else " PSNR %.2f dB" % psnr, hist["rays_per_s"][-1]))
            if save_dir:
                self.save_checkpoint(save_dir, model_names, epoch + 1, hist)
        return hist


class RayBatchLoader:
    """The shuffled DataLoader over RaysFromImagesDataset (train.py:96-100) with the rays generated on the device
    (raygen.RayGenerator): `iterations` batches of `batch_size` uniformly drawn rays per epoch; each rank draws from its
    own generator seed (base + rank).  Data-parallel runs shard the data set BY IMAGE (1200 images -> 150 per
    GPU): build this rank's generator with RayGenerator.for_rank(...), which keeps only frames rank, rank + world, ...
    on the device - the union over ranks covers every frame exactly once and no image is replicated."""

    def __init__(self, ray_generator, batch_size: int, iterations: int, seed: int = 0):
        self.gen, self.batch_size, self.iterations = ray_generator, int(batch_size), int(iterations)
        _, rank = sdist.world_rank()
        self.rng = torch.Generator(device=ray_generator.device)
        self.rng.manual_seed(int(seed) + rank)

    def __len__(self):
        return self.iterations

    def __iter__(self):
        for _ in range(self.iterations):
            yield self.gen.random_batch(self.batch_size, generator=self.rng)


class FrameLoader:
    """Validation loader: whole frames of a RayGenerator in row-major ray order, `batch_size` rays at a time, jitter
    0.5 (the deterministic mid-bin samples render.py uses)."""

    def __init__(self, ray_generator, frames, batch_size: int):
        self.gen, self.frames, self.batch_size = ray_generator, list(frames), int(batch_size)

    def __iter__(self):
        hw = self.gen.h * self.gen.w
        for f in self.frames:
            for r0 in range(0, hw, self.batch_size):
                idx = torch.arange(f * hw + r0, f * hw + min(r0 + self.batch_size, hw), device=self.gen.device)
                yield self.gen.batch(idx, torch.full((idx.shape[0],), 0.5, dtype=torch.float64, device=self.gen.device))
