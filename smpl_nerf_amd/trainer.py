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

On the GPU the whole body above is ONE call into the HIP library for a plain NerfPipeline
(snerf_nerf_train_step_f32; with more than one rank its two halves, snerf_nerf_train_grads_f32 and
snerf_adam_step_f32, around the all-reduce), and the optimiser of every pipeline is the library's Adam over the flat
parameter buffer (HipAdam): no torch.optim, no autograd graph, no re-pack of the weight streams.
"""
from __future__ import annotations

import ctypes
import os
import time

import torch

from . import _lib
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


class HipAdam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) (solver/nerf_solver.py:11-14, 31-33) over the trainer's flat
    parameter / gradient buffers, the update being ONE launch of the library's Adam kernel (snerf_adam_step_f32: the
    statements of torch's single-tensor update in their order).  Same surface as the torch optimiser where the trainer and
    checkpoints touch it: param_groups (hyper-parameters are read at every step), zero_grad, step, state_dict /
    load_state_dict in torch.optim.Adam's own format (a checkpoint moves between the two).  Like torch, a parameter whose
    .grad is None is skipped and its step counter does not advance (one device counter per parameter tensor)."""

    def __init__(self, params, flat_p, flat_g, views, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params, self.flat_p, self.flat_g, self.views = list(params), flat_p, flat_g, views
        dev = flat_p.device
        self.exp_avg = torch.zeros_like(flat_p)
        self.exp_avg_sq = torch.zeros_like(flat_p)
        self.steps = torch.zeros(len(self.params), dtype=torch.int64, device=dev)     # torch: state[p]["step"]
        self._host_steps = [0] * len(self.params)      # mirror of `steps` (grouping only; the device counters are what the kernel reads)
        self._pending = None
        self.scratch = torch.zeros(64, dtype=torch.float32, device=dev)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.offsets.append(off)
        self.param_groups = [{"params": self.params, "lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay,
                              "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                              "differentiable": False, "fused": None, "decoupled_weight_decay": False}]
        self._range_cache = (None, None, 0)

    # -- C structures -------------------------------------------------------------------------------------------
    def c_state(self):
        g = self.param_groups[0]
        return _lib.AdamState(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                              self.exp_avg_sq.data_ptr(), self.flat_p.numel(), self.scratch.data_ptr(), float(g["lr"]),
                              float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))

    def c_ranges(self, has_grad):
        """(snerf_adam_range array, count) for the parameters flagged in `has_grad` (one bool per parameter tensor): runs
        of adjacent tensors with equal step counts.  commit_step() advances the host mirror of the counters once the call went through."""
        hs = self._host_steps
        key = (tuple(has_grad), tuple(hs[i] == hs[i - 1] for i in range(1, len(hs))))
        if self._range_cache[0] != key:
            runs, i, n = [], 0, len(self.params)
            while i < n:
                if not has_grad[i]:
                    i += 1
                    continue
                j = i
                while j + 1 < n and has_grad[j + 1] and self._host_steps[j + 1] == self._host_steps[i]:
                    j += 1
                runs.append((i, j + 1))
                i = j + 1
            if len(runs) > 32:
                raise RuntimeError("HipAdam: more than 32 runs of parameters with different gradient / step patterns")
            arr = (_lib.AdamRange * max(len(runs), 1))()
            for k, (a, b) in enumerate(runs):
                arr[k] = _lib.AdamRange(self.offsets[a], self.offsets[b], self.steps.data_ptr() + 8 * a, b - a)
            self._range_cache = (key, arr, len(runs))
        self._pending = tuple(has_grad)
        return self._range_cache[1], self._range_cache[2]

    def commit_step(self):
        """Advance the host mirror of the step counters for the ranges c_ranges handed out last - called once the library call
        that runs the update has returned SNERF_OK (advancing before the call left the mirror ahead of the device
        counters when the call failed)."""
        for i, h in enumerate(self._pending or ()):
            if h:
                self._host_steps[i] += 1
        self._pending = None

    # -- torch.optim.Optimizer surface -------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def bind_gradients(self):
        """Every gradient into its slot of the flat buffer (the backward kernels wrote most of them there already: the grad
        sinks); afterwards p.grad is that view.  Returns the has-grad flags."""
        flags = []
        for p, v in zip(self.params, self.views):
            g = p.grad
            flags.append(g is not None)
            if g is not None and g.data_ptr() != v.data_ptr():
                v.copy_(g)
                p.grad = v
        return flags

    @torch.no_grad()
    def step(self, nets=None, n_nets=0):
        """One update from the gradients in p.grad.  nets: optional snerf_adam_net array (weight streams to keep current)."""
        ranges, n = self.c_ranges(self.bind_gradients())
        if n == 0:
            return
        st = self.c_state()
        lib = _lib.load()
        with torch.cuda.device(self.flat_p.device):
            _lib.check(lib.snerf_adam_step_f32(ctypes.byref(st), ranges, n, nets, n_nets, _lib.current_stream()),
                       "snerf_adam_step_f32")
        self.commit_step()

    def state_dict(self):
        """torch.optim.Adam.state_dict()'s format (loads into a torch.optim.Adam over the same parameter list)."""
        steps = self.steps.cpu().tolist()
        self._host_steps = list(steps)
        state = {}
        for i, t in enumerate(steps):
            if t > 0:
                a, b = self.offsets[i], self.offsets[i + 1]
                shp = self.params[i].shape
                state[i] = {"step": torch.tensor(float(t)), "exp_avg": self.exp_avg[a:b].view(shp).clone(),
                            "exp_avg_sq": self.exp_avg_sq[a:b].view(shp).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise ValueError("HipAdam.load_state_dict: the checkpoint has a different number of parameters")
        for k in ("lr", "betas", "eps", "weight_decay"):
            self.param_groups[0][k] = tuple(group[k]) if k == "betas" else group[k]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = [0] * len(self.params)
        for i, st in sd["state"].items():
            i = int(i)
            a, b = self.offsets[i], self.offsets[i + 1]
            self.exp_avg[a:b].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[a:b].copy_(st["exp_avg_sq"].reshape(-1))
            steps[i] = int(round(float(st["step"])))
        self._host_steps = steps
        self.steps.copy_(torch.tensor(steps, dtype=torch.int64))
        self._range_cache = (None, None, 0)


class DataParallelTrainer:
    default_adam_args = {"lr": 1e-4, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0}  # nerf_solver.py:11-14

    def __init__(self, pipeline, models, lr: float = 5e-4, weight_decay: float = 0.0, loss_func=None, fused=None,
                 one_call=None, sync_at_world_one=False):
        self.pipeline = pipeline
        self.models = list(models)
        self.world, self.rank = sdist.world_rank()
        # the gradient all-reduce runs when there is more than one rank - or, on request, in a process group of ONE rank
        # (bench.py on a 1-GPU box: the collective of the path on RCCL with nothing to exchange; the mean over one rank is the
        # identity)
        self._sync = self.world > 1 or (bool(sync_at_world_one) and sdist.is_dist())
        self._comm = False     # dist.RcclComm of the one-call data-parallel step (_init_comm; False: not usable here)
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
        self.broadcast_ms = None
        if self.world > 1:
            t0 = time.perf_counter()
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
            if self._flat_p is not None and self._flat_p.is_cuda:
                torch.cuda.synchronize(self._flat_p.device)
            self.broadcast_ms = (time.perf_counter() - t0) * 1e3
        args = dict(self.default_adam_args)
        args.update({"lr": lr, "weight_decay": weight_decay})
        # optimiser: the library's Adam over the flat buffer wherever the parameters live on the GPU (`fused` None or
        # "hip"); fused = True / False select torch.optim.Adam(fused=True) / the unfused torch optimiser (CPU tests, A/B runs)
        on_gpu = self._flat_p is not None and self._flat_p.is_cuda
        if fused is None:
            fused = "hip" if on_gpu else False
        if fused == "hip":
            if not on_gpu:
                raise RuntimeError("DataParallelTrainer: the HIP optimiser needs fp32 parameters on one GPU")
            self.optim = HipAdam(self.params, self._flat_p, self._flat_g, self._views, **args)
        elif fused:
            self.optim = torch.optim.Adam(self.params, fused=True, **args)
        else:
            self.optim = torch.optim.Adam(self.params, **args)
        self.loss_func = loss_func or torch.nn.MSELoss()
        # the one-call training step (snerf_nerf_train_step_f32): None = wherever it applies (a plain NerfPipeline trained
        # with the default loss through HipAdam), False = always the autograd path
        self.one_call = one_call
        # rays per chunk of the one-call step: the saved activations are sized by the chunk, not by the batch (include/smplnerf.h)
        self.rays_per_chunk = int(os.environ.get("SNERF_TRAIN_CHUNK_RAYS", "2048"))
        self._oc = None            # state of the one-call path (descriptors, slot tables, workspace)
        self.last_outputs = None   # (rgb, rgb_fine) of the last one-call step
        self.timing = None         # optional dict: HIP-event pairs around the gradient all-reduce (bench.py)
        self._init_comm()

    def _init_comm(self):
        """The RCCL communicator of the one-call data-parallel step (dist.RcclComm), created HERE - its construction is a collective
        (broadcast of the id, ncclCommInitRank), so it must not hang on a per-step, per-rank condition - and only when
        EVERY rank can use it: the ranks agree once (a MIN all-reduce of the local verdict) between the one-call form with RCCL
        inside the call and the three-call form (gradients, torch.distributed all-reduce, optimiser).  False: not usable here (the
        ranks share a GPU - the gloo dry runs -, SNERF_DP_ONE_CALL=0, one_call=False, parameters not flattened / not on a GPU)."""
        self._comm = False
        if not self._sync:
            return
        dev = self._flat_p.device if self._flat_p is not None else None
        ok = (os.environ.get("SNERF_DP_ONE_CALL", "1") != "0" and self.one_call is not False and self._flat_g is not None
              and isinstance(self.optim, HipAdam) and sdist.rccl_usable(dev))
        if sdist.is_dist() and self.world > 1:
            on_dev = dev is not None and dev.type == "cuda" and sdist.dist.get_backend() == "nccl"
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if on_dev else "cpu")
            sdist.dist.all_reduce(flag, op=sdist.dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
        if ok:
            self._comm = sdist.RcclComm(dev)

    def close(self):
        """Destroys the communicator (ncclCommDestroy); the trainer keeps working on the three-call form afterwards."""
        comm, self._comm = getattr(self, "_comm", None), False
        if comm:
            comm.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

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
        if not self._sync:
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
        self._allreduce_flat()
        for p, v in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def _rccl_comm(self, dev):
        """The communicator _init_comm created, or None: the three-call form (gradients, torch.distributed all-reduce, optimiser)."""
        return self._comm or None

    def _allreduce_flat(self):
        """The one collective of a step, optionally between two HIP events (bench.py reads self.timing)."""
        if self.timing is not None and self._flat_g.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sdist.allreduce_mean_(self._flat_g)
            e1.record()
            self.timing.setdefault("allreduce_events", []).append((e0, e1))
        else:
            sdist.allreduce_mean_(self._flat_g)

    # ------------------------------------------------------------------ the one-call step
    def _one_call_state(self):
        """Descriptors, segment offsets and slot tables of the one-call path, or None when it does not apply: a plain
        NerfPipeline (snerf_nerf_train_step_f32) or a SmplNerfPipeline with the encoded pose (snerf_smpl_nerf_train_step_aux_f32)
        over RenderRayNets without additional inputs (and the WarpFieldNet) that are exactly this trainer's models, the
        default MSE loss, the library's optimiser, every parameter trainable."""
        # what the decision below depends on and a caller can change between steps (a later requires_grad_(False) or a
        # swapped loss_func used to be ignored): decided again when it changes
        sig = (tuple(p.requires_grad for p in self.params), type(self.loss_func), getattr(self.loss_func, "reduction", None), self.one_call)
        if getattr(self, "_oc_sig", None) != sig:
            self._oc, self._oc_sig = None, sig
        if self._oc is not None:
            return self._oc or None
        from .nets import AppendVerticesNet, RenderRayNet, WarpFieldNet
        from .pipelines import (AppendSmplParamsPipeline, AppendToNerfPipeline, AppendVerticesPipeline, NerfPipeline,
                                SmplNerfPipeline)
        self._oc = False
        pipe = self.pipeline
        smpl = type(pipe) is SmplNerfPipeline
        posed = type(pipe) in (AppendSmplParamsPipeline, AppendToNerfPipeline)    # per-ray pose columns in front of the encoding
        verts = type(pipe) is AppendVerticesPipeline      # per-ray vertex floats (quirk Q7), estimator / body model not trained here
        if self.one_call is False or not (type(pipe) is NerfPipeline or smpl or posed or verts) or not isinstance(self.optim, HipAdam):
            return None
        mc, mf = pipe.model_coarse, pipe.model_fine
        want = AppendVerticesNet if verts else RenderRayNet
        if type(mc) is not want or type(mf) is not want or mc is mf:
            return None
        if getattr(mc, "_layered", False) or getattr(mf, "_layered", False) or \
                (smpl and getattr(pipe.model_warp_field, "_layered", False)):
            return None        # widths above the fused kernels': the layer-by-layer autograd path (layered.py)
        mine = [mc, mf]
        upstream = []      # trained modules in front of the nets (AppendVerticesSolver's second parameter group: the pose estimator,
        if verts:          # solver/append_vertices_solver.py, lrate_pose): they receive d loss / d vertices from the call (r05)
            upstream = [m for m in (pipe.smpl_estimator, pipe.smpl_model) if isinstance(m, torch.nn.Module)
                        and any(p.requires_grad for p in m.parameters())]
            mine = mine + [m for m in upstream if any(m is q for q in self.models)]
            if len(mine) != 2 + len(upstream):
                return None    # (a trained upstream module this trainer does not optimise: the autograd path leaves its .grad)
        if smpl:        # the fused warp stage with the encoded pose (human_pose_encoding = 1; the raw-pose mode's fine branch fails like
            mw = pipe.model_warp_field       # the reference's, quirk Q5, and stays on the autograd path)
            if type(mw) is not WarpFieldNet or not getattr(pipe.args, "human_pose_encoding", 0):
                return None
            if not (mc.use_directional_input and mf.use_directional_input):
                return None
            # the split-precision dgrad returns input gradients for the default-sized encoders only (identity columns or more
            # frequencies need the fp32 variant and its own transposed stream): nets.py routes that per call, the one call does not
            from .nets import _split_code, _wide_encoders
            for m in (mc, mf):
                if _split_code(m, 0) and _wide_encoders(m.desc_for_encoders(pipe.position_encoder, pipe.direction_encoder)):
                    return None
            mine.append(mw)
        if len(self.models) != len(mine) or {id(m) for m in self.models} != {id(m) for m in mine}:
            return None
        if type(self).loss is not DataParallelTrainer.loss or type(self.loss_func) is not torch.nn.MSELoss or \
                self.loss_func.reduction != "mean":
            return None
        if (not (posed or verts) and (mc.additional_input_dim or mf.additional_input_dim)) or \
                not all(p.requires_grad for p in (self.params if not verts else [q for m in (mc, mf) for q in m._ordered_params()])):
            return None
        if upstream and self._flat_g is None:
            return None
        if posed and (not mc.additional_input_dim or mc.additional_input_dim != mf.additional_input_dim):
            return None
        seg = {id(m): (off, n) for m, off, n in self._segments}
        if any(id(m) not in seg for m in mine if not any(m is u for u in upstream)):
            return None        # (the nets need their segment of the flat buffers; an upstream module's gradients are copied into its views)
        lib = _lib.load()
        oc = {"nets": (mc, mf), "seg": (seg[id(mc)], seg[id(mf)]), "slots": {}, "ws": None, "lib": lib, "warp": None, "posed": posed or verts, "verts": verts,
              "upstream": upstream}
        # second stream for the coarse net's backward of small batches (include/smplnerf.h: aux_stream); SNERF_TRAIN_AUX_STREAM=0: none
        oc["aux"] = torch.cuda.Stream(self._flat_p.device) if os.environ.get("SNERF_TRAIN_AUX_STREAM", "1") != "0" else None
        # parameter tensors of each net (indices into self.params): the optimiser's has-grad flags of a step
        index = {id(p): i for i, p in enumerate(self.params)}
        oc["tensors"] = tuple(frozenset(index[id(p)] for p in m._ordered_params()) for m in (mc, mf))
        oc["upstream_tensors"] = frozenset(index[id(p)] for m in upstream for p in m.parameters() if id(p) in index and p.requires_grad)
        if smpl:
            mw = pipe.model_warp_field
            oc["warp"] = {"net": mw, "seg": seg[id(mw)], "tensors": frozenset(index[id(p)] for p in mw._params()), "packed_t": None}
        self._oc = oc
        return oc

    def _slot_tables(self, oc, net, desc, input_grad=False):
        key = (id(net), net._desc_key(desc), bool(input_grad))
        hit = oc["slots"].get(key)
        if hit is None:
            n = int(oc["lib"].snerf_mlp_param_floats(desc))
            dev = self._flat_p.device
            hit = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
            with torch.cuda.device(dev):
                _lib.check(oc["lib"].snerf_mlp_stream_slots(desc, hit[0].data_ptr(), hit[1].data_ptr(), 1 if input_grad else 0,
                                                            _lib.current_stream()), "snerf_mlp_stream_slots")
            oc["slots"][key] = hit
        return hit

    def _step_one_call(self, oc, batch):
        # the per-ray rows the nets read, under autograd when something in front of them wants the gradient the call returns (a
        # trained pose estimator, a goal_pose with requires_grad); everything else of the step records nothing
        add_graph = None
        with torch.set_grad_enabled(bool(oc["upstream"]) or (oc["posed"] and not oc["verts"] and batch[4].requires_grad)):
            if oc["verts"] or oc["posed"]:
                add_graph = self._additional_rows(oc, batch)
        with torch.no_grad():
            return self._step_one_call_impl(oc, batch, add_graph)

    def _upstream_backward(self, oc, add_graph, d_add):
        """torch autograd from the rows' gradient back into what produced them; the upstream modules' parameter gradients are
        written into their views of the flat gradient buffer."""
        ups = [self.params[i] for i in sorted(oc["upstream_tensors"])]
        for p in ups:
            p.grad = None
        with torch.enable_grad():
            add_graph.backward(d_add)
        for i in sorted(oc["upstream_tensors"]):
            p, v = self.params[i], self._views[i]
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)

    def _additional_rows(self, oc, batch):
        pipe, args = self.pipeline, self.pipeline.args
        if oc["verts"]:      # models/append_vertices_pipeline.py:30-58: estimator -> body model -> the vertex floats the nets read (Q7)
            images, z_vals = batch[4], batch[3]
            goal_poses, betas = pipe.smpl_estimator(images)
            orient = torch.zeros([1, 3], device=z_vals.device).expand(z_vals.shape[0], -1)
            vertices = pipe.smpl_model(betas=betas, return_verts=True, body_pose=goal_poses, global_orient=orient).vertices
            return vertices.reshape(z_vals.shape[0], -1)[:, :oc["nets"][0].positions_dim].contiguous().float()
        # models/append_smpl_params_pipeline.py:29-37 / append_to_nerf_pipeline.py:26: the pose rows the nets read
        add = pipe._select(batch[4]).contiguous()
        if args.human_pose_encoding:
            add = pipe.human_pose_encoder.encode(add)
        add = add.reshape(add.shape[0], -1).contiguous()
        if add.shape[1] != oc["nets"][0].additional_input_dim:
            raise RuntimeError("DataParallelTrainer: the pose rows do not match the nets' additional_input_dim")
        return add.float()

    def _step_one_call_impl(self, oc, batch, add_graph):
        from . import ops
        from .nets import _split_code
        lib = oc["lib"]
        pipe, args = self.pipeline, self.pipeline.args
        W = oc["warp"]
        goal_pose = add = None
        want_d_add = add_graph is not None and add_graph.requires_grad
        if oc["verts"]:
            ray_samples, rays_o, rays_d, z_vals, images, rgb_truth = (t.detach().contiguous() for t in batch)
            add = add_graph.detach()
        elif oc["posed"]:
            ray_samples, rays_o, rays_d, z_vals, goal_pose, rgb_truth = (t.detach().contiguous() for t in batch)
            add = add_graph.detach()
        elif W is not None:
            ray_samples, rays_o, rays_d, z_vals, goal_pose, rgb_truth = (t.contiguous() for t in batch)
        else:
            ray_samples, rays_o, rays_d, z_vals, rgb_truth = (t.contiguous() for t in batch)
        dev = self._flat_p.device
        B, Nc = z_vals.shape
        Nf = int(args.number_fine_samples) if args.run_fine else 0
        mc, mf = oc["nets"]
        ns = _split_code(mc, 0)
        if Nf and _split_code(mf, 0) != ns:
            raise RuntimeError("DataParallelTrainer: both nets must use the same precision mode")
        descs, packed, packed_t, nets_c = [], [], [], (_lib.AdamNet * 2)()
        for k, m in enumerate((mc, mf)):
            d = m.desc_for_rows() if oc["verts"] else m.desc_for_encoders(pipe.position_encoder, pipe.direction_encoder, bool(oc["posed"]))
            descs.append(d)
            if k == 1 and not Nf:          # run_fine = 0: the fine net takes no part (models/nerf_pipeline.py:43-44)
                packed.append(None), packed_t.append(None)
                continue
            ig = W is not None          # the warp stage: the nets back-propagate into their inputs (input_grad = 1 streams)
            if ns:
                packed.append(m.packed_weights_bf16(d, ns, training=True))
                packed_t.append(m.packed_weights_t_bf16(d, ns, ig))
                sf = st = None
            else:
                packed.append(m.packed_weights(d, training=True))
                packed_t.append(m.packed_weights_t(d, ig))
                sf, st = self._slot_tables(oc, m, d, ig)
            nets_c[k] = _lib.AdamNet(ctypes.pointer(d), oc["seg"][k][0], ns, packed[k].data_ptr(), packed_t[k].data_ptr(),
                                     _lib.ptr(sf), _lib.ptr(st))
        n_nets = 2 if Nf else 1
        if W is not None:
            mw, pe_ = W["net"], pipe.position_encoder
            wdesc = _lib.WarpDesc(mw.width, pe_.number_frequencies, 1 if pe_.include_identity else 0, mw.direcions_dim)
            packed_w = mw._packed(wdesc, training=True)
            if W["packed_t"] is None or W["packed_t"][0] is not packed_w:      # (a re-built forward stream = the weights changed behind the step)
                sizes = [ctypes.c_int64() for _ in range(4)]
                _lib.check(lib.snerf_warp_train_sizes(wdesc, 0, *[ctypes.byref(v) for v in sizes]), "snerf_warp_train_sizes")
                pt = torch.empty(sizes[2].value, device=dev, dtype=torch.float32)
                seg_w = self._flat_p[W["seg"][0]:W["seg"][0] + W["seg"][1]]
                with torch.cuda.device(dev):
                    _lib.check(lib.snerf_warp_pack_t_f32(wdesc, seg_w.data_ptr(), pt.data_ptr(), _lib.current_stream()), "snerf_warp_pack_t_f32")
                W["packed_t"] = (packed_w, pt)
            packed_t_w = W["packed_t"][1]
            two = torch.stack([goal_pose[:, 38], goal_pose[:, 41]], axis=-1).contiguous()          # models/smpl_nerf_pipeline.py:28
            pose_enc = pipe.human_pose_encoder.encode(two).contiguous()                              # :30
            need = int(lib.snerf_smpl_nerf_train_workspace_bytes(descs[0], descs[1] if Nf else None, wdesc, B, Nc, Nf, self.rays_per_chunk))
        else:
            need = int(lib.snerf_nerf_train_workspace_bytes(descs[0], descs[1] if Nf else None, B, Nc, Nf, self.rays_per_chunk))
        if need < 0:
            _lib.check(need, "snerf_nerf_train_workspace_bytes")
        if oc["ws"] is None or oc["ws"].numel() < need:
            oc["ws"] = None
            oc["ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty(4 + 6 * B, dtype=torch.float32, device=dev)
        loss, rgb, rgb_fine = out[:3], out[4:4 + 3 * B].view(B, 3), out[4 + 3 * B:].view(B, 3)
        u = ops.uniform_u(Nf, dev) if Nf else None
        nz_c = pipe._noise((B, Nc), dev)
        nz_f = pipe._noise((B, Nc + Nf), dev) if Nf else None
        cb = _lib.NerfBatch(ray_samples.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(),
                            rgb_truth.data_ptr(), _lib.ptr(u), _lib.ptr(nz_c), _lib.ptr(nz_f), _lib.ptr(add), B, Nc, Nf,
                            1 if args.white_background else 0)
        (oc_off, oc_n), (of_off, of_n) = oc["seg"]
        g_c = self._flat_g.data_ptr() + 4 * oc_off
        g_f = self._flat_g.data_ptr() + 4 * of_off
        head = (descs[0], packed[0].data_ptr(), packed_t[0].data_ptr(), descs[1] if Nf else None, _lib.ptr(packed[1]),
                _lib.ptr(packed_t[1]), ns, ctypes.byref(cb), self.rays_per_chunk, oc["ws"].data_ptr(), g_c, g_f if Nf else None,
                loss.data_ptr(), rgb.data_ptr(), rgb_fine.data_ptr())
        opt = self.optim
        aux = oc["aux"].cuda_stream if oc["aux"] is not None else None
        live = oc["tensors"][0] | (oc["tensors"][1] if Nf or self._sync else frozenset())
        if W is not None:
            live = live | W["tensors"]
        flags = [i in live for i in range(len(self.params))]
        # the call(s): one entry when this rank steps alone; its two halves around the all-reduce otherwise
        stream = _lib.current_stream
        if W is not None:
            w_off = W["seg"][0]
            g_w = self._flat_g.data_ptr() + 4 * w_off
            head = head[:6] + (wdesc, packed_w.data_ptr(), packed_t_w.data_ptr(), ns, ctypes.byref(cb), pose_enc.data_ptr(),
                               self.rays_per_chunk, oc["ws"].data_ptr(), g_c, g_f if Nf else None, g_w, loss.data_ptr(), rgb.data_ptr(),
                               rgb_fine.data_ptr())
            # (the _aux_ forms, 0.1.8: small chunks run the coarse chain of the backward on the auxiliary stream; comm may be None)
            name, tail = "snerf_smpl_nerf_train", (aux,)
            step_tail = lambda st, ranges, nr, comm=None: (ctypes.byref(st), ranges, nr, nets_c, n_nets, w_off, comm, stream(), aux)
        else:
            name, tail = "snerf_nerf_train", (aux,)
            step_tail = lambda st, ranges, nr: (ctypes.byref(st), ranges, nr, nets_c, n_nets, stream(), aux)
        ig = d_add = None
        if want_d_add:      # d loss / d additional rows comes back from the call (include/smplnerf.h: snerf_input_grads)
            d_add = torch.empty_like(add)
            pc = self._flat_p.data_ptr() + 4 * oc_off
            pf = self._flat_p.data_ptr() + 4 * of_off
            ig = _lib.InputGrads(d_add.data_ptr(), pc, pf if Nf else None)
        with torch.cuda.device(dev), _lib.timed(f"train_step{'_smpl' if W is not None else ''}[B={B}]"):
            comm = self._rccl_comm(dev) if self._sync else None
            if not self._sync and ig is not None:
                ranges, nr = opt.c_ranges(flags)
                st = opt.c_state()
                _lib.check(lib.snerf_nerf_train_step_ig_f32(*head, ctypes.byref(st), ranges, nr, nets_c, n_nets, ctypes.byref(ig), stream(), aux),
                           "snerf_nerf_train_step_ig_f32")
                opt.commit_step()
            elif not self._sync:
                ranges, nr = opt.c_ranges(flags)
                st = opt.c_state()
                entry = name + ("_step_aux_f32" if W is not None else "_step_f32")
                _lib.check(getattr(lib, entry)(*head, *step_tail(st, ranges, nr)), entry)
                opt.commit_step()
            elif comm is not None:
                # more than one rank, one call all the same: the gradient average is RCCL's ncclAllReduce(ncclAvg) of the flat
                # buffer inside the call, on the compute stream between the backward and Adam (include/smplnerf.h 8(e))
                if not Nf:      # every rank contributes the same shape: zeros for the net that took no part
                    self._flat_g[of_off:of_off + of_n].zero_()
                ranges, nr = opt.c_ranges(flags)
                st = opt.c_state()
                if ig is not None:
                    # ... and with d loss / d additional rows coming back (r06: snerf_nerf_train_step_dp_ig_f32; the rows are this
                    # rank's own rays - the estimator's gradient is averaged below, behind its autograd)
                    _lib.check(lib.snerf_nerf_train_step_dp_ig_f32(*head, ctypes.byref(st), ranges, nr, nets_c, n_nets, ctypes.byref(ig),
                                                                   comm.handle, stream(), aux), "snerf_nerf_train_step_dp_ig_f32")
                else:
                    if W is not None:
                        t, entry = step_tail(st, ranges, nr, comm.handle), name + "_step_aux_f32"
                    else:
                        t = step_tail(st, ranges, nr)
                        t, entry = t[:-2] + (comm.handle,) + t[-2:], name + "_step_dp_f32"
                    _lib.check(getattr(lib, entry)(*head, *t), entry)
                opt.commit_step()
                self.collective_calls = getattr(self, "collective_calls", 0) + 1
            else:
                if ig is not None:
                    _lib.check(lib.snerf_nerf_train_grads_ig_f32(*head, ctypes.byref(ig), stream(), aux), "snerf_nerf_train_grads_ig_f32")
                else:
                    entry = name + ("_grads_aux_f32" if W is not None else "_grads_f32")
                    _lib.check(getattr(lib, entry)(*head, stream(), *tail), entry)
                if ig is not None and oc["upstream"]:      # the upstream modules' gradients must be in the flat buffer before it is averaged
                    self._upstream_backward(oc, add_graph, d_add)
                if not Nf:      # every rank contributes the same shape: zeros for the net that took no part
                    self._flat_g[of_off:of_off + of_n].zero_()
                self._allreduce_flat()
                if ig is not None and oc["upstream"]:
                    flags = [f or (i in oc["upstream_tensors"]) for i, f in enumerate(flags)]
                ranges, nr = opt.c_ranges(flags)
                st = opt.c_state()
                _lib.check(lib.snerf_adam_step_f32(ctypes.byref(st), ranges, nr, nets_c, n_nets, stream()), "snerf_adam_step_f32")
                opt.commit_step()
                if W is not None:
                    _lib.check(lib.snerf_warp_repack_f32(wdesc, self._flat_p.data_ptr(), self._flat_p.numel(), w_off, packed_w.data_ptr(),
                                                         packed_t_w.data_ptr(), stream()), "snerf_warp_repack_f32")
        if self._sync and not Nf and hasattr(mf, "mark_weights_changed"):
            #  run_fine = 0 with more than one rank: the fine net's parameters are live in the ranges (zero gradients:
            # weight decay and resumed moments still move them) but its streams were not among the nets the call refreshed
            mf.mark_weights_changed()
        if ig is not None and not (self._sync and comm is None and oc["upstream"]):
            # back through what produced the rows: a goal_pose that wants its gradient gets it here; a trained estimator's parameter
            # gradients land in the flat buffer and its tensors take their optimiser step (a second, small C-ABI call)
            self._upstream_backward(oc, add_graph, d_add)
            if oc["upstream"]:
                uflags = [i in oc["upstream_tensors"] for i in range(len(self.params))]
                if comm is not None:      # more than one rank: the upstream tensors' stretches of the flat gradient, averaged
                    idx = sorted(oc["upstream_tensors"])
                    k = 0
                    while k < len(idx):
                        j = k
                        while j + 1 < len(idx) and idx[j + 1] == idx[j] + 1:
                            j += 1
                        comm.allreduce_avg_(self._flat_g[opt.offsets[idx[k]]:opt.offsets[idx[j] + 1]])
                        k = j + 1
                with torch.cuda.device(dev):
                    ranges, nr = opt.c_ranges(uflags)
                    if nr:
                        st = opt.c_state()
                        _lib.check(lib.snerf_adam_step_f32(ctypes.byref(st), ranges, nr, None, 0, stream()), "snerf_adam_step_f32")
                        opt.commit_step()
                flags = [a or b for a, b in zip(flags, uflags)]
        elif ig is not None:
            flags = [f or (i in oc["upstream_tensors"]) for i, f in enumerate(flags)]
        # p.grad = what autograd would have left: views of the flat gradient buffer (None for a net that took no part)
        for i, (p, v) in enumerate(zip(self.params, self._views)):
            want = v if flags[i] else None
            if p.grad is not want:
                p.grad = want
        self.last_outputs = (rgb, rgb_fine)
        return loss[0]

    def step(self, batch):
        """One optimisation step on this rank's batch (list of tensors, rgb_truth last). Returns the
        local loss tensor (not synchronised with the host)."""
        oc = self._one_call_state()
        if oc is not None and len(batch) == (6 if (oc["warp"] is not None or oc["posed"]) else 5) and \
                all(t.is_cuda and (not t.requires_grad or (i == 4 and oc["posed"] and not oc["verts"])) and
                    (t.dtype == torch.float32 or (oc["verts"] and i == 4)) for i, t in enumerate(batch)) and \
                not getattr(self.pipeline.args, "strict_cumsum", 0):
            # (a batch tensor that wants a gradient: the autograd path - except the pose rows of the pose-conditioned pipelines,
            # whose gradient the call returns, r05)
            return self._step_one_call(oc, batch)
        self.optim.zero_grad(set_to_none=True)
        self._arm_grad_sinks()
        out = self.pipeline(batch)
        loss = self.loss(out[0], out[1], batch[-1])
        loss.backward()
        self.sync_gradients()
        self.optim.step()
        # the optimisers update the parameters without bumping autograd's version counters, which the nets' packed
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
    (raygen.RayGenerator).  Data-parallel runs shard the data set BY IMAGE (1200 images -> 150 per GPU): build
    this rank's generator with RayGenerator.for_rank(...), which keeps only frames rank, rank + world, ... on the
    device - the union over ranks covers every frame exactly once and no image is replicated.  Each rank draws from its own
    generator seed (base + rank).

    shuffle = True (the reference's loader: DataLoader(shuffle=True, drop_last=False), train.py:100): an epoch is a random
    permutation of this rank's rays cut into batches of `batch_size` - every ray exactly once, the last batch short - with a
    new permutation per epoch; `iterations` then caps the number of batches (None: the whole epoch).  With more than one rank the
    count is the minimum over the ranks (shards may differ by a frame; the gradient all-reduce of a step is collective).
    shuffle = False: `iterations` batches of uniformly drawn rays, with replacement (no epoch structure)."""

    def __init__(self, ray_generator, batch_size: int, iterations: int = None, seed: int = 0, shuffle: bool = False):
        self.gen, self.batch_size = ray_generator, int(batch_size)
        self.shuffle = bool(shuffle)
        if iterations is None and not self.shuffle:
            raise ValueError("RayBatchLoader: draws with replacement need `iterations`")
        full = -(-ray_generator.n_rays // self.batch_size)
        self.iterations = full if iterations is None else (min(int(iterations), full) if self.shuffle else int(iterations))
        world, rank = sdist.world_rank()
        if world > 1 and self.shuffle:
            # ranks with image shards of different sizes would run different numbers of steps and the per-step gradient
            # all-reduce would hang: every rank runs the smallest count
            self.iterations = int(sdist.min_over_ranks(self.iterations, ray_generator.device))
        self.rng = torch.Generator(device=ray_generator.device)
        self.rng.manual_seed(int(seed) + rank)

    def __len__(self):
        return self.iterations

    def __iter__(self):
        if not self.shuffle:
            for _ in range(self.iterations):
                yield self.gen.random_batch(self.batch_size, generator=self.rng)
            return
        dev = self.gen.device
        perm = torch.randperm(self.gen.n_rays, device=dev, generator=self.rng)
        for i in range(self.iterations):
            idx = perm[i * self.batch_size:(i + 1) * self.batch_size]
            jit = torch.rand((idx.shape[0],), device=dev, dtype=torch.float64, generator=self.rng)   # one scalar per ray (Q8)
            yield self.gen.batch(idx, jit)


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
