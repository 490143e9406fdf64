"""RenderRayNet drop-in (models/render_ray_net.py:8-61): same constructor arguments, same
sub-module names and therefore the same state_dict keys / checkpoint files (utils.py:267-289), but
forward() runs the fused HIP kernel instead of 13 nn.Linear calls.

The nn.Linear children only hold the parameters; they are never called.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import MlpDesc, check, current_stream, ptr


def _encoder_shape(dim: int):
    """(L, identity) of a 3-channel encoder producing `dim` features, or None."""
    if dim % 6 == 0:
        return dim // 6, 0
    if dim >= 3 and (dim - 3) % 6 == 0:
        return (dim - 3) // 6, 1
    return None


_ENCODED_ROWS = -1  # marker passed in the `per_sample` slot of _FusedMlpFn
MAX_WIDTH = 512     # --netwidth limit of the fused kernels (csrc/mlp_plan.h: make_plan; above 256: the 512-feature kernels, fp32)
MAX_WARP_WIDTH = 256  # --netwidth_warp limit (csrc/mlp_plan.h: make_warp_plan)


def _need_f32_cuda(what: str, *tensors):
    """The kernels read raw device pointers as fp32: anything else must be rejected here, not reinterpreted."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"{what}: inputs must live on the GPU (got {t.device}); smpl_nerf_amd has no CPU path")
        if t.dtype != torch.float32:
            raise RuntimeError(f"{what}: inputs must be float32 (got {t.dtype})")


def flat_parameter_vector(params) -> torch.Tensor:
    """The parameters as one contiguous fp32 vector in the given order (the C-ABI's params_flat).  Zero-copy when
    they already are adjacent views of one buffer (trainer.flatten_parameters_ lays them out like that, so a training step
    packs its weight streams without a torch.cat of 26 tensors); otherwise a torch.cat."""
    p0 = params[0]
    if all(p.dtype == torch.float32 and p.is_contiguous() for p in params):
        base, addr = p0.untyped_storage().data_ptr(), p0.data_ptr()
        for p in params:
            if p.untyped_storage().data_ptr() != base or p.data_ptr() != addr:
                break
            addr += 4 * p.numel()
        else:
            return torch.as_strided(p0.detach(), (sum(p.numel() for p in params),), (1,))
    return torch.cat([p.detach().reshape(-1).float() for p in params])


def _dy_rows(desc):
    """(first tile-row, n_out) of every forward layer in the `dy` buffer the backward kernels leave behind
    (include/smplnerf.h: snerf_mlp_dy_layout)."""
    lib = _lib.load()
    cnt = ctypes.c_int32()
    rows = (ctypes.c_int32 * 21)()
    nout = (ctypes.c_int32 * 21)()
    check(lib.snerf_mlp_dy_layout(desc, ctypes.byref(cnt), rows, nout, None), "snerf_mlp_dy_layout")
    return [(rows[l], nout[l]) for l in range(cnt.value)]


def _contract(dy, n, first_row, n_feat, weight, col0, ncols, spr, out, out_col0, accumulate):
    """out[s or ray, out_col0 + c] (+)= sum_f d Y[s, f] * weight[f, col0 + c] (summed over the samples of a ray when spr > 0):
    snerf_dy_contract_f32 on the stored tile-rows - no gathered [n, 256] copy, no library GEMM."""
    lib = _lib.load()
    w = weight.detach()
    if not w.is_contiguous() or w.dtype != torch.float32:
        w = w.contiguous().float()
    if spr == 1:
        spr = 0          # one sample per "ray": per-sample rows, no partial sums
    scratch = None
    if spr:
        scratch = torch.empty(int(lib.snerf_dy_contract_scratch_floats(n, ncols, int(spr))), device=dy.device, dtype=torch.float32)
    with torch.cuda.device(dy.device), _lib.timed(f"dy_contract[n={n}]"):
        check(lib.snerf_dy_contract_f32(ptr(dy), n, int(first_row), int(n_feat), ptr(w), w.shape[1], int(col0), int(ncols),
                                        int(spr), ptr(out), out.stride(0), int(out_col0), 1 if accumulate else 0, ptr(scratch),
                                        current_stream()), "snerf_dy_contract_f32")


def _extra_input_grads(net, desc, dy, n, d_rows=None, d_add=None, spr=1):
    """Gradients w.r.t. the non-hidden input columns of the net, as contractions of the stored d Y_l with the weight
    columns that read them (what autograd does in the reference, models/render_ray_net.py:43-56): layer 0 and the skip
    layers read [positions | additional] (in the column order of the weight matrix), directional_input reads the
    direction encoding.  d_rows [n, row_floats] (zero-filled by the caller): gradient of already-encoded input rows
    [positions (+ additional) | ... | directions]; d_add [n / spr, add_dim]: gradient of the per-ray additional inputs,
    summed over the samples of each ray."""
    nh = net.n_layers - 1
    pin = net.positions_pose_input.weight.shape[1]       # positions (+ additional) columns
    layout = _dy_rows(desc)
    readers = [(0, net.positions_pose_input.weight, 0)]
    readers += [(i + 1, net.positional_net[i].weight, net.width) for i in range(nh) if i in net.skips]
    if d_rows is not None:
        for l, w, c0 in readers:
            _contract(dy, n, layout[l][0], layout[l][1], w, c0, pin, 0, d_rows, 0, True)
        if net.use_directional_input:
            l = nh + 3
            ddim = net.directional_input.weight.shape[1] - net.width
            _contract(dy, n, layout[l][0], layout[l][1], net.directional_input.weight, net.width, ddim, 0, d_rows,
                      d_rows.shape[1] - ddim, True)
    if d_add is not None:
        a0 = 0 if desc.add_first else 3 * ((1 if desc.pos_identity else 0) + 2 * desc.pos_freqs)
        for k, (l, w, c0) in enumerate(readers):
            _contract(dy, n, layout[l][0], layout[l][1], w, c0 + a0, desc.add_dim, spr, d_add, 0, k > 0)


def _wide_encoders(desc) -> bool:
    """More than 4 position / 2 direction encoder k-blocks of 16 slots (csrc/mlp_plan.h: pe_nkb, bwd_pe_tiles)."""
    def nkb(L, ident):
        per_lane = (3 * (1 if ident else 0) + 3 * L + 3) // 4
        return (2 * per_lane + 3) // 4
    return nkb(desc.pos_freqs, desc.pos_identity) > 4 or (desc.use_dir and nkb(desc.dir_freqs, desc.dir_identity) > 2)


def _grads_from_flat(flat, shapes):
    grads, off = [], 0
    for shp in shapes:
        k = 1
        for v in shp:
            k *= v
        grads.append(flat[off:off + k].view(shp))
        off += k
    return grads


def _split_code(net, per_sample):
    """nsplit / precision code of the net's matrix-core arithmetic for a fused call (0 = exact fp32 kernels)."""
    if net.width != 256 or per_sample == _ENCODED_ROWS:
        return 0
    return {"bf16x6": 3, "bf16x3": 2, "f16x3": _lib.SPLIT_F16X3}.get(net.precision, 0)


def _fold_workspace(size_fn, desc, n, spr, dev):
    """(tensor or None, bytes) - the caller-allocated table of the per-ray fold of an fp32 inference call (include/smplnerf.h:
    snerf_mlp_fold_workspace_bytes / snerf_warp_fold_workspace_bytes; 0 bytes = the fold does not apply)."""
    nbytes = int(size_fn(desc, n, int(spr)))
    if nbytes < 0:
        check(nbytes, "fold_workspace_bytes")
    return (torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes) if nbytes else (None, 0)


def _train_sizes(desc, n):
    """(act_floats, dy_floats, gpart_floats) of snerf_mlp_train_sizes for n samples."""
    lib = _lib.load()
    sizes = [ctypes.c_int64() for _ in range(4)]
    cnt = ctypes.c_int32()
    check(lib.snerf_mlp_train_sizes(desc, n, *[ctypes.byref(v) for v in sizes], ctypes.byref(cnt)), "snerf_mlp_train_sizes")
    return sizes[0].value, sizes[1].value, sizes[3].value


def _launch_forward(net, desc, ns, x, d, per_sample, spr, add, raw, act=None, like_training=False):
    """One launch of the fused encode + MLP kernel on n = raw.shape[0] samples: inference (act None) or the training
    forward that also saves the layer inputs into `act`.  like_training: an inference launch whose `raw` must equal the
    training forward's bit for bit (the block-wise backward recomputes with the training kernel): no per-ray fold of the
    additional inputs (include/smplnerf.h: SNERF_FWD_NO_RAY_FOLD)."""
    lib = _lib.load()
    n = raw.shape[0]
    train = act is not None
    tag = "mlp_fwd_train" if train else "mlp_fwd"
    with torch.cuda.device(raw.device), _lib.timed(f"{tag}[n={n}]"):
        if per_sample == _ENCODED_ROWS:      # x holds already-encoded rows (RenderRayNet.forward(x))
            packed = net.packed_weights(desc, training=True)
            if train:
                check(lib.snerf_mlp_fwd_encoded_train_f32(desc, ptr(packed), ptr(x), n, x.shape[1], ptr(raw), ptr(act),
                                                          current_stream()), "snerf_mlp_fwd_encoded_train_f32")
            else:
                check(lib.snerf_mlp_fwd_encoded_f32(desc, ptr(packed), ptr(x), n, x.shape[1], ptr(raw), current_stream()),
                      "snerf_mlp_fwd_encoded_f32")
        elif ns:                             # on the 16-bit matrix cores; saved activations are fp32 all the same
            packed = net.packed_weights_bf16(desc, ns, training=True)
            if train:
                check(lib.snerf_mlp_fwd_train_bf16_f32(desc, ptr(packed), ns, ptr(x), ptr(d), per_sample, ptr(add), n,
                                                       int(spr), ptr(raw), ptr(act), current_stream()),
                      "snerf_mlp_fwd_train_bf16_f32")
            else:
                check(lib.snerf_mlp_fwd_bf16_f32(desc, ptr(packed), ns, ptr(x), ptr(d), per_sample, ptr(add), n, int(spr),
                                                 ptr(raw), current_stream()), "snerf_mlp_fwd_bf16_f32")
        else:
            packed = net.packed_weights(desc, training=True)
            if train:
                check(lib.snerf_mlp_fwd_train_f32(desc, ptr(packed), ptr(x), ptr(d), per_sample, ptr(add), n, int(spr),
                                                  ptr(raw), ptr(act), current_stream()), "snerf_mlp_fwd_train_f32")
            elif like_training or add is None:
                check(lib.snerf_mlp_fwd_f32(desc, ptr(packed), ptr(x), ptr(d), per_sample, ptr(add), n, int(spr), ptr(raw),
                                            current_stream()), "snerf_mlp_fwd_f32")
            else:
                ws, nb = _fold_workspace(lib.snerf_mlp_fold_workspace_bytes, desc, n, spr, raw.device)
                check(lib.snerf_mlp_fwd_ws_f32(desc, ptr(packed), ptr(x), ptr(d), per_sample, ptr(add), n, int(spr), ptr(raw),
                                               ptr(ws), nb, current_stream()), "snerf_mlp_fwd_ws_f32")


def _launch_backward(net, desc, ns, act, d_raw, n, sizes, flat, input_grad, x=None, d=None, per_sample=0, spr=1,
                     d_x=None, d_d=None):
    """dgrad + split-K wgrad + reduce on one block of n samples: flat (snerf_mlp_param_floats floats) is overwritten; returns
    the d Y buffer (tile-row-major, snerf_mlp_dy_layout) for the contractions of _extra_input_grads."""
    lib = _lib.load()
    dev = d_raw.device
    if ns and input_grad and _wide_encoders(desc):
        ns = 0       # input gradients through encoders with identity columns / more frequencies: fp32 dgrad variant only
    packed_t = net.packed_weights_t_bf16(desc, ns, input_grad) if ns else net.packed_weights_t(desc, input_grad)
    dy = torch.empty(sizes[1], device=dev, dtype=torch.float32)
    gpart = torch.empty(sizes[2], device=dev, dtype=torch.float32)
    if input_grad:
        with torch.cuda.device(dev), _lib.timed(f"mlp_bwd_inputs[n={n}]"):
            if ns:
                check(lib.snerf_mlp_bwd_inputs_bf16_f32(desc, ptr(packed_t), ns, ptr(act), ptr(d_raw), ptr(x), ptr(d),
                                                        per_sample, spr, n, ptr(dy), ptr(gpart), ptr(flat), ptr(d_x),
                                                        ptr(d_d), current_stream()), "snerf_mlp_bwd_inputs_bf16_f32")
            else:
                check(lib.snerf_mlp_bwd_inputs_f32(desc, ptr(packed_t), ptr(act), ptr(d_raw), ptr(x), ptr(d), per_sample,
                                                   spr, n, ptr(dy), ptr(gpart), ptr(flat), ptr(d_x), ptr(d_d),
                                                   current_stream()), "snerf_mlp_bwd_inputs_f32")
    else:
        with torch.cuda.device(dev), _lib.timed(f"mlp_bwd[n={n}]"):
            if ns:
                check(lib.snerf_mlp_bwd_bf16_f32(desc, ptr(packed_t), ns, ptr(act), ptr(d_raw), n, ptr(dy), ptr(gpart),
                                                 ptr(flat), current_stream()), "snerf_mlp_bwd_bf16_f32")
            else:
                check(lib.snerf_mlp_bwd_f32(desc, ptr(packed_t), ptr(act), ptr(d_raw), n, ptr(dy), ptr(gpart), ptr(flat),
                                            current_stream()), "snerf_mlp_bwd_f32")
    return dy


class _FusedMlpFn(torch.autograd.Function):
    """raw = RenderRayNet(encode(x), encode(normalise(d))) with gradients for every weight and bias and, where the caller's
    graph asks for them, for the inputs: positions / per-sample directions (SmplNerfPipeline: the dgrad kernel's
    input-gradient variant), per-ray additional inputs and already-encoded rows (contractions of the stored d Y_l,
    _extra_input_grads).  Forward saves the layer inputs in the tile-row-major activation buffer; backward = dgrad +
    split-K wgrad + reduce (snerf_mlp_bwd_*).  In NerfPipeline positions and directions are leaves (the hierarchical
    samples are detached, utils.py:260) and only the parameters receive gradients.

    Memory bound ("store only x, d"): the saved layer inputs and the d Y buffer cost ~21 KB per sample
    (256-wide, depth 8) - 89 GB for a 128x128 frame, 357 GB for a 256x256 one.  When that exceeds the net's
    `activation_budget_bytes`, the forward runs the inference kernel and keeps only its inputs; the backward then walks the
    samples in blocks of whole rays that fit the budget: training forward (recomputed: the same arithmetic, so the same
    ReLU masks) -> dgrad -> wgrad -> reduce per block, parameter gradients summed over the blocks in fp32.  One third more
    matrix work than the stored form, any batch size."""

    @staticmethod
    def forward(ctx, net, desc, x, d, per_sample, spr, add, *params):
        n = x.shape[0]
        dev = x.device
        spr = int(spr)
        net._begin_training_forward()
        ns = _split_code(net, per_sample)
        act_floats, dy_floats, gpart_floats = _train_sizes(desc, n)
        raw = torch.empty((n, 4), device=dev, dtype=torch.float32)
        group = 1 if per_sample == _ENCODED_ROWS else spr            # blocks are whole rays (per-ray directions / inputs)
        budget = getattr(net, "activation_budget_bytes", 0)
        if budget is None:
            # default: a quarter of what this process can get on THIS device - free at the driver plus what torch's caching
            # allocator holds without using (driver-free memory alone shrinks as the cache fills, so the same batch
            # could flip between the stored and the block-wise backward) - decided ONCE per net and device, not per forward
            # ... but not forever: a call that wants to STORE its activations under the remembered budget looks at what
            # is obtainable now, and a budget remembered from emptier times is replaced - so a second pipeline, larger batches or
            # validation frames that arrived in between turn the call to the block-wise backward instead of an out-of-memory error
            cache = net.__dict__.setdefault("_activation_budget_cache", {})
            budget = cache.get(dev.index)
            need = 4 * (act_floats + dy_floats + gpart_floats)
            if budget is None or (need <= budget and need > (1 << 28)):
                free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                if budget is None or need > 0.5 * free:
                    budget = cache[dev.index] = int(0.25 * free)
        budget = int(budget or 0)
        ctx.block = 0
        if budget > 0 and 4 * (act_floats + dy_floats + gpart_floats) > budget and n > group:
            per_sample_bytes = 4.0 * (act_floats + dy_floats + gpart_floats) / n
            ctx.block = max(group, int(budget / per_sample_bytes) // group * group)
        if ctx.block and ctx.block < n:
            _launch_forward(net, desc, ns, x, d, per_sample, spr, add, raw, like_training=True)   # inference kernel: nothing saved
            ctx.act = None
            # through save_for_backward: an in-place change of the inputs or of a parameter between forward and backward
            # (the recompute reads both) raises autograd's version-counter error instead of giving silently wrong gradients
            ctx.save_for_backward(x, d, add, *params)
            ctx.blocked_inputs = True
        else:
            ctx.block = 0
            act = torch.empty(act_floats, device=dev, dtype=torch.float32)
            _launch_forward(net, desc, ns, x, d, per_sample, spr, add, raw, act)
            ctx.act = act
        ctx.net, ctx.desc, ctx.n, ctx.ns, ctx.per_sample = net, desc, n, ns, per_sample
        ctx.sizes = (act_floats, dy_floats, gpart_floats)
        ctx.shapes = [p.shape for p in params]
        ctx.input_grad = bool(ctx.needs_input_grad[2] or ctx.needs_input_grad[3]) and per_sample != _ENCODED_ROWS
        if ctx.input_grad:       # SmplNerfPipeline: positions / per-sample directions depend on the warp net
            ctx.xd = (x, d, per_sample, spr)
        # gradients w.r.t. already-encoded rows (RenderRayNet.forward(x) feeding an upstream module) and w.r.t. the per-ray
        # additional inputs (AppendVerticesPipeline: vertices <- smpl_model <- smpl_estimator)
        ctx.rows_grad = bool(ctx.needs_input_grad[2]) and per_sample == _ENCODED_ROWS
        ctx.add_grad = bool(ctx.needs_input_grad[6]) and add is not None
        ctx.spr, ctx.row_floats = spr, (x.shape[1] if per_sample == _ENCODED_ROWS else 0)
        return raw

    @staticmethod
    def _input_grads_from_dy(ctx, dy, n, d_x_rows, d_add):
        """Contractions of the stored d Y_l of a block of n samples: encoded-row gradients into d_x_rows [n, row_floats],
        per-ray additional-input gradients into d_add [n / spr, add_dim]."""
        if ctx.rows_grad:        # encoded rows = [positions (+ additional) | ... | directions]
            d_x_rows.zero_()
            _extra_input_grads(ctx.net, ctx.desc, dy, n, d_rows=d_x_rows)
        elif ctx.add_grad:       # per-ray constants: the per-sample contributions of a ray are summed in the kernel
            _extra_input_grads(ctx.net, ctx.desc, dy, n, d_add=d_add, spr=ctx.spr)

    @staticmethod
    def backward(ctx, d_raw):
        lib = _lib.load()
        net, desc, n, ns = ctx.net, ctx.desc, ctx.n, ctx.ns
        dev = d_raw.device
        d_raw = d_raw.contiguous().float()
        flat = net._take_grad_sink(lib.snerf_mlp_param_floats(desc), dev)
        d_x = d_d = d_add = None
        spr = ctx.spr
        if ctx.input_grad:
            x, d, per_sample, _ = ctx.xd
            d_x = torch.empty((n, 3), device=dev, dtype=torch.float32)
            d_d = torch.zeros((n, 3), device=dev, dtype=torch.float32)
        if ctx.rows_grad:
            d_x = torch.empty((n, ctx.row_floats), device=dev, dtype=torch.float32)
        if ctx.add_grad:
            d_add = torch.empty((n // spr, desc.add_dim), device=dev, dtype=torch.float32)

        if not ctx.block:        # everything was saved by the forward: one pass
            if ctx.input_grad:
                dy = _launch_backward(net, desc, ns, ctx.act, d_raw, n, ctx.sizes, flat, True, x, d, per_sample, spr, d_x, d_d)
            else:
                dy = _launch_backward(net, desc, ns, ctx.act, d_raw, n, ctx.sizes, flat, False)
            ctx.act = None
            _FusedMlpFn._input_grads_from_dy(ctx, dy, n, d_x if ctx.rows_grad else None, d_add)
        else:                    # blocks of whole rays: recompute the layer inputs, back-propagate, accumulate
            xin, din, addin = ctx.saved_tensors[:3]
            per_sample = ctx.per_sample
            tmp = None
            for s0 in range(0, n, ctx.block):
                m = min(ctx.block, n - s0)
                sizes = _train_sizes(desc, m)
                act = torch.empty(sizes[0], device=dev, dtype=torch.float32)
                raw_b = torch.empty((m, 4), device=dev, dtype=torch.float32)
                xb = xin[s0:s0 + m]
                if per_sample == _ENCODED_ROWS:
                    db = ab = None
                else:
                    db = din[s0:s0 + m] if per_sample else din[s0 // spr:(s0 + m) // spr]
                    ab = None if addin is None else addin[s0 // spr:(s0 + m) // spr]
                _launch_forward(net, desc, ns, xb, db, per_sample, spr, ab, raw_b, act)
                if s0 == 0:
                    target = flat
                else:
                    if tmp is None:
                        tmp = torch.empty_like(flat)
                    target = tmp
                if ctx.input_grad:
                    dy = _launch_backward(net, desc, ns, act, d_raw[s0:s0 + m], m, sizes, target, True, xb, db, per_sample,
                                          spr, d_x[s0:s0 + m], d_d[s0:s0 + m])
                else:
                    dy = _launch_backward(net, desc, ns, act, d_raw[s0:s0 + m], m, sizes, target, False)
                if s0:
                    flat.add_(tmp)
                _FusedMlpFn._input_grads_from_dy(ctx, dy, m, d_x[s0:s0 + m] if ctx.rows_grad else None,
                                                 None if d_add is None else d_add[s0 // spr:(s0 + m) // spr])
                del act, dy
        if ctx.input_grad:
            if not per_sample:       # one direction per ray: sum the per-sample contributions
                d_d = d_d.view(-1, spr, 3).sum(1)
            ctx.xd = None
        return (None, None, d_x, d_d, None, None, d_add) + tuple(_grads_from_flat(flat, ctx.shapes))



class _PackedWeightsEpoch:
    """The packed weight streams are cached against (data_ptr, autograd version) of every parameter.  Not every
    in-place update bumps the version counter - torch.optim.Adam(fused=True) does not - so the caches also carry an
    epoch: every training forward starts a new one (an optimiser step is expected to follow), the first inference call
    after a training forward starts another, and whoever updates weights behind autograd's back calls
    mark_weights_changed() (DataParallelTrainer.step does after every optimiser step)."""

    def mark_weights_changed(self):
        """Call after changing parameter values in a way autograd's version counters do not see: fused optimisers,
        `p.data.copy_(...)` / `p.data[...] = ...` (`.data` has its own version counter), raw-pointer writes.  In-place ops
        on the parameter itself (`p.copy_()` under no_grad, `load_state_dict`) and re-assigned parameters are detected
        without it."""
        self._weights_epoch += 1

    def _begin_training_forward(self):
        self._weights_epoch += 1
        self._trained_since_pack = True

    def _begin_inference(self):
        if self._trained_since_pack:
            self._weights_epoch += 1
            self._trained_since_pack = False

    # -- gradient sink: DataParallelTrainer hands every net its segment of ONE flat gradient buffer; the backward kernels
    #    write the parameter gradients straight into it (what is all-reduced is what snerf_mlp_bwd_* wrote - no copy-in /
    #    copy-out).  A segment is handed out once per backward pass: a net that is evaluated twice per step (the warp
    #    field: coarse and fine stage) gets a private buffer the second time and autograd accumulates into the first.
    _grad_sink = None
    _grad_sink_taken = False

    def set_grad_sink(self, flat_segment):
        self._grad_sink = flat_segment
        self._grad_sink_taken = False

    def _take_grad_sink(self, numel: int, dev):
        sink = self._grad_sink
        if sink is not None and not self._grad_sink_taken and sink.numel() == numel and sink.device == dev:
            self._grad_sink_taken = True
            return sink
        return torch.empty(numel, device=dev, dtype=torch.float32)

    def _cached_pack(self, cache, key, params, build, keep_others=False):
        """Weight stream `key`, rebuilt by build(flat fp32 parameter vector) only when a parameter changed: the cache is
        keyed on (data_ptr, autograd version) of every parameter and the weights epoch (mark_weights_changed)."""
        dev = params[0].device
        if not params[0].is_cuda:
            raise RuntimeError(f"{type(self).__name__}: parameters must be on the GPU (smpl_nerf_amd has no CPU path)")
        stamp = (str(dev), self._weights_epoch) + tuple((p.data_ptr(), p._version) for p in params)
        hit = cache.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.cuda.device(dev):
            packed = build(flat_parameter_vector(params), dev)
        if not keep_others:
            cache.clear()
        cache[key] = (stamp, packed)
        return packed


class RenderRayNet(_PackedWeightsEpoch, nn.Module):

    def __init__(self, n_layers=8, width=256, positions_dim=60, directions_dim=24, additional_input_dim=0,
                 skips=[4], use_directional_input=1):
        super(RenderRayNet, self).__init__()
        if int(width) < 2 or int(n_layers) < 1:
            raise ValueError(f"RenderRayNet: width {width} / n_layers {n_layers}: need width >= 2 and n_layers >= 1")
        # the reference's parser accepts any --netwidth / --netdepth (config_parser.py:19-20); every configuration it ships uses
        # 256 / 8.  The fused kernels keep the layer chain in registers up to 512 features and 16 layers; above that the net runs
        # layer by layer (layered.py, csrc/linear.hip: one exact-fp32 MFMA GEMM per nn.Linear, activations in HBM) - r06
        self._layered = int(width) > MAX_WIDTH or int(n_layers) > 16
        self.n_layers = n_layers
        self.width = width
        self.positions_dim = positions_dim
        self.direcions_dim = directions_dim  # (sic) attribute name of the reference, :14
        self.skips = skips
        self.additional_input_dim = additional_input_dim
        self.use_directional_input = use_directional_input

        self.positions_pose_input = torch.nn.Linear(positions_dim + additional_input_dim, width)
        self.positional_net = nn.ModuleList()
        for i in range(self.n_layers - 1):
            if i in skips:
                self.positional_net.append(torch.nn.Linear(width + positions_dim + additional_input_dim, width))
            else:
                self.positional_net.append(torch.nn.Linear(width, width))
        self.additional_linear_layer = torch.nn.Linear(width, width)
        self.sigma_out_layer = torch.nn.Linear(width, 1)
        directional_width = width // 2
        if use_directional_input:
            self.directional_input = torch.nn.Linear(width + directions_dim, directional_width)
        else:
            self.directional_input = torch.nn.Linear(width, directional_width)
        self.directional_net = nn.ModuleList()
        for i in range(1):
            self.directional_net.append(torch.nn.Linear(directional_width, directional_width))
        self.rgb_out_layer = torch.nn.Linear(directional_width, 3)
        self._pack_cache = {}
        self._pack_t_cache = {}
        self._weights_epoch = 0            # see mark_weights_changed()
        self._trained_since_pack = False
        # matrix-core arithmetic of inference and of a training step's forward, dgrad and wide wgrad jobs: "fp32"
        # (v_mfma_f32_16x16x4_f32) or split-bf16 "bf16x6" (3 parts, fp32-class accuracy) / "bf16x3" (2 parts, ~1e-5
        # relative); activations, gradients, the narrow wgrad jobs and the reductions are fp32 in every mode
        self.precision = os.environ.get("SNERF_PRECISION", "fp32")
        # training memory bound (autograd path): saved layer inputs + d Y + wgrad partials of ONE forward call above this many
        # bytes are not kept but recomputed block by block in the backward (_FusedMlpFn).  None (default) = 25 % of the device
        # memory that is free when the forward runs (on an otherwise empty MI355X: a 4096-ray step of 22 GB keeps everything, a
        # 256x256 frame of 357 GB trains in a few blocks); SNERF_TRAIN_ACT_GB sets a fixed figure; 0 = never recompute.
        env = os.environ.get("SNERF_TRAIN_ACT_GB")
        self.activation_budget_bytes = int(float(env) * (1 << 30)) if env else None

    # ------------------------------------------------------------------ parameter plumbing
    def _ordered_params(self):
        """Weights and biases in registration (= state_dict) order, which is the order the C-ABI's
        flat parameter vector uses (include/smplnerf.h: snerf_mlp_param_floats)."""
        # (r06: every inference call asks for this list to validate its weight-stream cache - 70 nn.Module attribute look-ups per
        # net were a third of the host time of a 64-ray render.  The list is kept and re-validated by identity against the
        # layers' own parameter dicts, so a re-assigned `layer.weight = nn.Parameter(...)` is still seen.)
        hit = self.__dict__.get("_op_cache")
        if hit is not None:
            for d, k, q in hit[0]:
                if d.get(k) is not q:
                    break
            else:
                return hit[1]
        mods = [self.positions_pose_input] + list(self.positional_net) + [
            self.additional_linear_layer, self.sigma_out_layer, self.directional_input] + list(
            self.directional_net) + [self.rgb_out_layer]
        out, where = [], []
        for m in mods:
            out += [m.weight, m.bias]
            where += [(m._parameters, "weight", m.weight), (m._parameters, "bias", m.bias)]
        self.__dict__["_op_cache"] = (where, out)
        return list(out)

    def _skip_mask(self) -> int:
        mask = 0
        for i in self.skips:
            if 0 <= i < self.n_layers - 1:
                mask |= 1 << i
        return mask

    def make_desc(self, pos_L, pos_id, dir_L, dir_id, add_dim, add_first=0) -> MlpDesc:
        key = (self.n_layers, self.width, pos_L, pos_id, dir_L, dir_id, add_dim, self._skip_mask(),
               1 if self.use_directional_input else 0, 1 if add_first else 0)
        cache = self.__dict__.setdefault("_desc_cache", {})
        d = cache.get(key)
        if d is None:        # (descriptors are immutable by convention: one ctypes struct per shape, its key computed once)
            d = cache[key] = MlpDesc(*key)
            d._key = tuple(getattr(d, f[0]) for f in d._fields_)
        return d

    def desc_for_encoders(self, position_encoder, direction_encoder, add_first=False) -> MlpDesc:
        """Descriptor of the fused (encode + MLP) path; checks that the encoders produce what this net
        was built for (train.py:102-107: positions_dim = 3 * encoder.output_dim)."""
        pos_L, pos_id = position_encoder.number_frequencies, 1 if position_encoder.include_identity else 0
        dir_L, dir_id = direction_encoder.number_frequencies, 1 if direction_encoder.include_identity else 0
        if 3 * (pos_id + 2 * pos_L) != self.positions_dim or 3 * (dir_id + 2 * dir_L) != self.direcions_dim:
            raise RuntimeError("RenderRayNet: encoder output sizes do not match positions_dim/directions_dim")
        return self.make_desc(pos_L, pos_id, dir_L, dir_id, self.additional_input_dim, add_first)

    def desc_for_encoded(self) -> MlpDesc:
        """Descriptor for forward(x) on already-encoded rows: any slot assignment of the position
        columns is valid there, so an encoder-shaped one is used when positions_dim allows it and
        plain columns otherwise."""
        d = _encoder_shape(self.direcions_dim) if self.use_directional_input else (0, 0)
        if d is None:
            raise RuntimeError(f"RenderRayNet: directions_dim={self.direcions_dim} is not a 3-channel encoding")
        p = _encoder_shape(self.positions_dim)
        if p is None:
            return self.make_desc(0, 0, d[0], d[1], self.positions_dim + self.additional_input_dim)
        return self.make_desc(p[0], p[1], d[0], d[1], self.additional_input_dim)

    @staticmethod
    def _desc_key(desc):
        key = getattr(desc, "_key", None)
        return key if key is not None else tuple(getattr(desc, f[0]) for f in desc._fields_)

    def packed_weights(self, desc: MlpDesc, training: bool = False) -> torch.Tensor:
        """MFMA-ordered fp32 weight stream for `desc` (snerf_mlp_pack_f32), re-packed only when a parameter changed."""
        if not training:
            self._begin_inference()
        lib = _lib.load()

        def build(flat, dev):
            n_param, n_pack = lib.snerf_mlp_param_floats(desc), lib.snerf_mlp_packed_floats(desc)
            if n_param < 0 or n_pack < 0:
                check(int(min(n_param, n_pack)), "snerf_mlp_packed_floats")
            if flat.numel() != n_param:
                raise RuntimeError(f"RenderRayNet: {flat.numel()} parameters but the descriptor expects {n_param}")
            packed = torch.empty(n_pack, device=dev, dtype=torch.float32)
            check(lib.snerf_mlp_pack_f32(desc, ptr(flat), ptr(packed), current_stream()), "snerf_mlp_pack_f32")
            return packed

        return self._cached_pack(self._pack_cache, self._desc_key(desc), self._ordered_params(), build)

    def packed_weights_bf16(self, desc: MlpDesc, nsplit: int, training: bool = False) -> torch.Tensor:
        """Split-bf16 / two-part fp16 weight stream (snerf_mlp_pack_bf16), cached like packed_weights."""
        if not training:
            self._begin_inference()
        lib = _lib.load()

        def build(flat, dev):
            nbytes = lib.snerf_mlp_packed_bf16_bytes(desc, nsplit)
            if nbytes < 0:
                check(int(nbytes), "snerf_mlp_packed_bf16_bytes")
            packed = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            check(lib.snerf_mlp_pack_bf16(desc, ptr(flat), ptr(packed), nsplit, current_stream()), "snerf_mlp_pack_bf16")
            return packed

        return self._cached_pack(self._pack_cache, self._desc_key(desc) + ("bf16", nsplit), self._ordered_params(), build)

    def packed_weights_t(self, desc: MlpDesc, input_grad: bool = False) -> torch.Tensor:
        """Transposed fp32 weight stream for the dgrad kernel (same caching rule as packed_weights)."""
        lib = _lib.load()

        def build(flat, dev):
            n_pack = ctypes.c_int64()
            check(lib.snerf_mlp_train_sizes(desc, 0, None, None, ctypes.byref(n_pack), None, None), "snerf_mlp_train_sizes")
            packed = torch.empty(n_pack.value, device=dev, dtype=torch.float32)
            check(lib.snerf_mlp_pack_t_f32(desc, ptr(flat), ptr(packed), 1 if input_grad else 0, current_stream()),
                  "snerf_mlp_pack_t_f32")
            return packed

        return self._cached_pack(self._pack_t_cache, self._desc_key(desc) + (bool(input_grad),), self._ordered_params(),
                                 build, keep_others=True)

    def packed_weights_t_bf16(self, desc: MlpDesc, nsplit: int, input_grad: bool = False) -> torch.Tensor:
        """Split-bf16 transposed weight stream for the bf16 dgrad kernel (snerf_mlp_pack_t_bf16)."""
        lib = _lib.load()

        def build(flat, dev):
            nbytes = lib.snerf_mlp_packed_t_bf16_bytes(desc, nsplit, 1 if input_grad else 0)
            if nbytes < 0:
                check(int(nbytes), "snerf_mlp_packed_t_bf16_bytes")
            packed = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            check(lib.snerf_mlp_pack_t_bf16(desc, ptr(flat), ptr(packed), nsplit, 1 if input_grad else 0, current_stream()),
                  "snerf_mlp_pack_t_bf16")
            return packed

        return self._cached_pack(self._pack_t_cache, self._desc_key(desc) + (bool(input_grad), "bf16", nsplit),
                                 self._ordered_params(), build, keep_others=True)

    # ------------------------------------------------------------------ forward paths
    def forward(self, x):
        """x [..., positions_dim + additional_input_dim + directions_dim] -> [..., 4] = [rgb | sigma]
        (models/render_ray_net.py:42-61)."""
        if not x.is_cuda:
            raise RuntimeError("RenderRayNet.forward: input must be on the GPU (no CPU path)")
        if self._layered:
            from . import layered
            xf = x.reshape(-1, x.shape[-1]).float()
            pin = self.positions_dim + self.additional_input_dim
            dd = xf[:, xf.shape[1] - self.direcions_dim:] if self.use_directional_input else None      # :43-44
            return layered.render_ray_net(self, xf[:, :pin], dd).reshape(x.shape[:-1] + (4,))
        desc = self.desc_for_encoded()
        xf = x.reshape(-1, x.shape[-1]).contiguous().float()
        n = xf.shape[0]
        if torch.is_grad_enabled() and (xf.requires_grad or any(p.requires_grad for p in self.parameters())):
            # gradients for the parameters and, like the reference's module, for the encoded rows themselves (the
            # reference's smpl_nerf / dynamic / image-wise pipelines train upstream modules through model(inputs))
            raw = _FusedMlpFn.apply(self, desc, xf, None, _ENCODED_ROWS, 1, None, *self._ordered_params())
            return raw.reshape(x.shape[:-1] + (4,))
        packed = self.packed_weights(desc)
        raw = torch.empty((n, 4), device=x.device, dtype=torch.float32)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            check(lib.snerf_mlp_fwd_encoded_f32(desc, ptr(packed), ptr(xf), n, xf.shape[1], ptr(raw),
                                                current_stream()), "snerf_mlp_fwd_encoded_f32")
        return raw.reshape(x.shape[:-1] + (4,))

    def forward_fused(self, positions, directions, samples_per_ray, position_encoder, direction_encoder,
                      additional=None, add_first=False):
        """Encode + MLP in one launch.  positions [n,3] (samples of a ray contiguous), directions
        [n/samples_per_ray, 3] (per ray) or [n, 3] (per sample), un-normalised; additional: optional
        [n/samples_per_ray, additional_input_dim] per-ray constants whose weight columns sit after
        (add_first=False) or before (add_first=True) the position-encoding columns.  Returns raw [n, 4]."""
        desc = self.desc_for_encoders(position_encoder, direction_encoder, add_first) if not self._layered else None
        _need_f32_cuda("RenderRayNet.forward_fused", positions, directions, additional)
        x = positions.reshape(-1, 3).contiguous()
        n = x.shape[0]
        d = directions.reshape(-1, 3).contiguous()
        if d.shape[0] == n and samples_per_ray != 1:
            per_sample = 1
        else:
            per_sample = 0
            if d.shape[0] * samples_per_ray != n:
                raise RuntimeError("forward_fused: directions do not match positions / samples_per_ray")
        add = None
        if self.additional_input_dim:
            if additional is None:
                raise RuntimeError("forward_fused: this net needs `additional` inputs")
            add = additional.reshape(-1, self.additional_input_dim).contiguous()
        if self._layered:
            from . import layered
            if 3 * position_encoder.output_dim != self.positions_dim or 3 * direction_encoder.output_dim != self.direcions_dim:
                raise RuntimeError("RenderRayNet: encoder output sizes do not match positions_dim/directions_dim")
            return layered.render_ray_net_fused(self, x, d, int(samples_per_ray), position_encoder, direction_encoder, add, add_first)
        if torch.is_grad_enabled() and (x.requires_grad or d.requires_grad or (add is not None and add.requires_grad) or
                                        any(p.requires_grad for p in self.parameters())):
            return _FusedMlpFn.apply(self, desc, x, d, per_sample, int(samples_per_ray), add, *self._ordered_params())
        raw = torch.empty((n, 4), device=x.device, dtype=torch.float32)
        lib = _lib.load()
        if self.precision in ("bf16x6", "bf16x3", "f16x3") and self.width == 256:
            # "f16x3": two fp16 parts (inference kernel only)
            ns = {"bf16x6": 3, "bf16x3": 2, "f16x3": _lib.SPLIT_F16X3}[self.precision]
            packed = self.packed_weights_bf16(desc, ns)
            with torch.cuda.device(x.device), _lib.timed(f"mlp_fwd[n={n}]"):
                check(lib.snerf_mlp_fwd_bf16_f32(desc, ptr(packed), ns, ptr(x), ptr(d), per_sample, ptr(add), n,
                                                 int(samples_per_ray), ptr(raw), current_stream()), "snerf_mlp_fwd_bf16_f32")
            return raw
        packed = self.packed_weights(desc)
        ws, nb = _fold_workspace(lib.snerf_mlp_fold_workspace_bytes, desc, n, samples_per_ray, x.device) if add is not None else (None, 0)
        with torch.cuda.device(x.device), _lib.timed(f"mlp_fwd[n={n}]"):
            check(lib.snerf_mlp_fwd_ws_f32(desc, ptr(packed), ptr(x), ptr(d), per_sample, ptr(add), n,
                                           int(samples_per_ray), ptr(raw), ptr(ws), nb, current_stream()), "snerf_mlp_fwd_ws_f32")
        return raw

    @property
    def is_cuda(self):
        return next(self.parameters()).is_cuda


class _WarpFn(torch.autograd.Function):
    """(warp, warped, sdirs) = WarpFieldNet stage with gradients for linear1/linear2 and for the per-ray pose rows.
    warped = x + warp and sdirs = warped - o, so the three incoming gradients add up to d loss / d warp.  With x = None
    (WarpFieldNet.forward(rows): desc.pos_freqs = pos_identity = 0, samples_per_ray = 1) the pose rows ARE the encoded
    input rows of models/warp_field_net.py:17-21 and their gradient is what the reference's autograd leaves in x.grad:
    d h @ linear1.weight, d h being the stored layer-0 d Y tile-rows of snerf_warp_bwd_f32.  The sample positions of the
    fused form are leaves (the pipeline's ray samples, models/smpl_nerf_pipeline.py:27,68)."""

    @staticmethod
    def forward(ctx, net, desc, x, pose_enc, o, spr, *params):
        lib = _lib.load()
        n = x.shape[0] if x is not None else pose_enc.shape[0] * int(spr)
        dev = pose_enc.device if x is None else x.device
        net._begin_training_forward()
        packed = net._packed(desc, training=True)
        sizes = [ctypes.c_int64() for _ in range(4)]
        check(lib.snerf_warp_train_sizes(desc, n, *[ctypes.byref(v) for v in sizes]), "snerf_warp_train_sizes")
        act = torch.empty(sizes[0].value, device=dev, dtype=torch.float32)
        warp = torch.empty((n, 3), device=dev, dtype=torch.float32)
        warped = sdirs = None
        if x is not None:
            warped, sdirs = (torch.empty((n, 3), device=dev, dtype=torch.float32) for _ in range(2))
        with torch.cuda.device(dev), _lib.timed(f"warp_fwd_train[n={n}]"):
            check(lib.snerf_warp_fwd_train_f32(desc, ptr(packed), ptr(x), ptr(pose_enc), ptr(o), n, int(spr), ptr(warp),
                                               ptr(warped), ptr(sdirs), ptr(act), current_stream()),
                  "snerf_warp_fwd_train_f32")
        ctx.net, ctx.desc, ctx.n, ctx.act, ctx.spr = net, desc, n, act, int(spr)
        ctx.sizes = (sizes[1].value, sizes[2].value, sizes[3].value)
        ctx.shapes = [p.shape for p in params]
        ctx.pose_grad = bool(ctx.needs_input_grad[3])
        ctx.set_materialize_grads(False)
        return warp, warped, sdirs

    @staticmethod
    def backward(ctx, d_warp, d_warped, d_sdirs):
        lib = _lib.load()
        net, desc, n = ctx.net, ctx.desc, ctx.n
        parts = [g for g in (d_warp, d_warped, d_sdirs) if g is not None]
        if not parts:
            return (None,) * (6 + len(ctx.shapes))
        total = parts[0]
        for g in parts[1:]:
            total = total + g
        total = total.contiguous().float()
        dev = total.device
        params = net._params()
        flatp = flat_parameter_vector(params)
        packed_t = torch.empty(ctx.sizes[1], device=dev, dtype=torch.float32)
        dy = torch.empty(ctx.sizes[0], device=dev, dtype=torch.float32)
        gpart = torch.empty(ctx.sizes[2], device=dev, dtype=torch.float32)
        flat = net._take_grad_sink(flatp.numel(), dev)
        with torch.cuda.device(dev), _lib.timed(f"warp_bwd[n={n}]"):
            check(lib.snerf_warp_pack_t_f32(desc, ptr(flatp), ptr(packed_t), current_stream()), "snerf_warp_pack_t_f32")
            check(lib.snerf_warp_bwd_f32(desc, ptr(packed_t), ptr(ctx.act), ptr(total), n, ptr(dy), ptr(gpart), ptr(flat),
                                         current_stream()), "snerf_warp_bwd_f32")
        ctx.act = None
        d_pose = None
        if ctx.pose_grad:      # d h (tile-rows 0 .. of dy: layer 0) contracted with linear1's pose columns, summed per ray
            pos_dim = 3 * ((1 if desc.pos_identity else 0) + 2 * desc.pos_freqs)
            d_pose = torch.empty((n // ctx.spr, desc.pose_dim), device=dev, dtype=torch.float32)
            _contract(dy, n, 0, desc.width, net.linear1.weight, pos_dim, desc.pose_dim, ctx.spr, d_pose, 0, False)
        return (None, None, None, d_pose, None, None) + tuple(_grads_from_flat(flat, ctx.shapes))


class WarpFieldNet(_PackedWeightsEpoch, nn.Module):
    """models/warp_field_net.py:6-22 drop-in: linear1 (width, positions_dim+pose_dim) + ReLU + linear2
    (3, width); `n_layers` is accepted and ignored exactly like the reference (:14-15).  forward(x) runs the
    fused HIP kernel on already-encoded rows; forward_fused() also encodes the positions and returns the
    warped samples and per-sample view directions (models/smpl_nerf_pipeline.py:38-53)."""

    def __init__(self, n_layers=8, width=256, positions_dim=60, pose_dim=24):
        super(WarpFieldNet, self).__init__()
        if int(width) < 1:
            raise ValueError(f"WarpFieldNet: width {width}: need width >= 1")
        # (config_parser.py:30 --netwidth_warp is free: above the fused kernels' 256 features the net runs layer by layer, layered.py)
        self._layered = int(width) > MAX_WARP_WIDTH
        self.positions_dim = positions_dim
        self.direcions_dim = pose_dim  # (sic) reference attribute name, :12
        self.width = width
        self.linear1 = torch.nn.Linear(positions_dim + pose_dim, width)
        self.linear2 = torch.nn.Linear(width, 3)
        self._pack_cache = {}
        self._weights_epoch = 0
        self._trained_since_pack = False
        # "fp32" (v_mfma_f32_16x16x4_f32) or split-bf16: any of "bf16x6" / "bf16x3" runs the warp net with three parts
        # (fp32-class accuracy: the warp moves the sample in front of the position encoding's 2^9 band)
        self.precision = os.environ.get("SNERF_PRECISION", "fp32")

    def _params(self):
        return [self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias]

    def _packed(self, desc, training: bool = False):
        if not training:
            self._begin_inference()
        lib = _lib.load()

        def build(flat, dev):
            n_param, n_pack = lib.snerf_warp_param_floats(desc), lib.snerf_warp_packed_floats(desc)
            if n_param < 0 or n_pack < 0:
                check(int(min(n_param, n_pack)), "snerf_warp_packed_floats")
            if flat.numel() != n_param:
                raise RuntimeError(f"WarpFieldNet: {flat.numel()} parameters but the descriptor expects {n_param}")
            packed = torch.empty(n_pack, device=dev, dtype=torch.float32)
            check(lib.snerf_warp_pack_f32(desc, ptr(flat), ptr(packed), current_stream()), "snerf_warp_pack_f32")
            return packed

        return self._cached_pack(self._pack_cache, tuple(getattr(desc, f[0]) for f in desc._fields_), self._params(), build)

    def _packed_bf16(self, desc):
        """Split-bf16 weight stream of the fused forward (snerf_warp_pack_bf16), cached like _packed."""
        self._begin_inference()
        lib = _lib.load()

        def build(flat, dev):
            nbytes = lib.snerf_warp_packed_bf16_bytes(desc)
            if nbytes < 0:
                check(int(nbytes), "snerf_warp_packed_bf16_bytes")
            packed = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            check(lib.snerf_warp_pack_bf16(desc, ptr(flat), ptr(packed), current_stream()), "snerf_warp_pack_bf16")
            return packed

        return self._cached_pack(self._pack_cache, tuple(getattr(desc, f[0]) for f in desc._fields_) + ("bf16",),
                                 self._params(), build)

    def forward(self, x):
        """x [..., positions_dim + pose_dim] (already encoded) -> warp [..., 3] (models/warp_field_net.py:17-21);
        differentiable w.r.t. the rows and the parameters like the reference's module."""
        if not x.is_cuda:
            raise RuntimeError("WarpFieldNet.forward: input must be on the GPU (no CPU path)")
        if x.shape[-1] != self.linear1.weight.shape[1]:
            raise RuntimeError(f"WarpFieldNet.forward: rows of {x.shape[-1]} floats, linear1 expects "
                               f"{self.linear1.weight.shape[1]}")
        rows = x.reshape(-1, x.shape[-1]).contiguous().float()
        if self._layered:
            from . import layered
            return layered.warp_field_net(self, rows).reshape(x.shape[:-1] + (3,))
        desc = _lib.WarpDesc(self.width, 0, 0, rows.shape[1])
        n = rows.shape[0]
        if torch.is_grad_enabled() and (rows.requires_grad or any(p.requires_grad for p in self.parameters())):
            warp, _, _ = _WarpFn.apply(self, desc, None, rows, None, 1, *self._params())
            return warp.reshape(x.shape[:-1] + (3,))
        packed = self._packed(desc)
        warp = torch.empty((n, 3), device=x.device, dtype=torch.float32)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            check(lib.snerf_warp_fwd_f32(desc, ptr(packed), None, ptr(rows), None, n, 1, ptr(warp), None, None,
                                         current_stream()), "snerf_warp_fwd_f32")
        return warp.reshape(x.shape[:-1] + (3,))

    def forward_fused(self, positions, pose_encoding, ray_translation, samples_per_ray, position_encoder):
        """positions [n,3] (samples of a ray contiguous), pose_encoding [n/spr, pose_dim], ray_translation
        [n/spr, 3] -> (warp, warped = positions + warp, sdirs = warped - ray_translation), each [n,3]."""
        pos_L, pos_id = position_encoder.number_frequencies, 1 if position_encoder.include_identity else 0
        if 3 * (pos_id + 2 * pos_L) != self.positions_dim or pose_encoding.shape[-1] != self.direcions_dim:
            raise RuntimeError("WarpFieldNet: encoder output sizes do not match positions_dim/pose_dim")
        desc = _lib.WarpDesc(self.width, pos_L, pos_id, self.direcions_dim)
        _need_f32_cuda("WarpFieldNet.forward_fused", positions, pose_encoding, ray_translation)
        x = positions.reshape(-1, 3).contiguous()
        n = x.shape[0]
        pe = pose_encoding.reshape(-1, self.direcions_dim).contiguous()
        o = ray_translation.reshape(-1, 3).contiguous()
        if pe.shape[0] * samples_per_ray != n or o.shape[0] * samples_per_ray != n:
            raise RuntimeError("forward_fused: per-ray inputs do not match positions / samples_per_ray")
        if self._layered:      # models/smpl_nerf_pipeline.py:38-56: rows = [PE(x) | pose], x' = x + warp, per-sample directions x' - o
            from . import layered
            xs = x.detach()
            rows = torch.cat([position_encoder.encode(xs), pe.repeat_interleave(int(samples_per_ray), dim=0)], -1)
            warp = layered.warp_field_net(self, rows)
            warped = xs + warp
            return warp, warped, warped - o.detach().repeat_interleave(int(samples_per_ray), dim=0)
        if torch.is_grad_enabled() and (pe.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _WarpFn.apply(self, desc, x.detach(), pe, o.detach(), int(samples_per_ray), *self._params())
        warp, warped, sdirs = (torch.empty((n, 3), device=x.device, dtype=torch.float32) for _ in range(3))
        lib = _lib.load()
        if self.precision in ("bf16x6", "bf16x3", "f16x3") and self.width == 256:
            packed = self._packed_bf16(desc)
            with torch.cuda.device(x.device), _lib.timed(f"warp_fwd[n={n}]"):
                check(lib.snerf_warp_fwd_bf16_f32(desc, ptr(packed), ptr(x), ptr(pe), ptr(o), n, int(samples_per_ray),
                                                  ptr(warp), ptr(warped), ptr(sdirs), current_stream()),
                      "snerf_warp_fwd_bf16_f32")
            return warp, warped, sdirs
        packed = self._packed(desc)
        ws, nb = _fold_workspace(lib.snerf_warp_fold_workspace_bytes, desc, n, samples_per_ray, x.device)
        with torch.cuda.device(x.device), _lib.timed(f"warp_fwd[n={n}]"):
            check(lib.snerf_warp_fwd_ws_f32(desc, ptr(packed), ptr(x), ptr(pe), ptr(o), n, int(samples_per_ray), ptr(warp),
                                            ptr(warped), ptr(sdirs), ptr(ws), nb, current_stream()), "snerf_warp_fwd_ws_f32")
        return warp, warped, sdirs

    @property
    def is_cuda(self):
        return next(self.parameters()).is_cuda


class AppendVerticesNet(RenderRayNet):
    """models/append_vertices_net.py:6-66 drop-in.  The reference slices its input as
    positions = x[:, :positions_dim], vertices = x[:, positions_dim:positions_dim+additional_input_dim],
    directions = x[:, -directions_dim:] (:44-47), pushes `vertices` through `vertices_net` and never uses the
    result (:48-50): the output is RenderRayNet(positions, directions) on the first positions_dim input
    columns.  The parameters of `vertices_net` are kept (same state_dict) and, as in the reference, receive
    no gradient; the dead branch is not evaluated."""

    def __init__(self, n_layers=8, width=256, positions_dim=60, directions_dim=24, additional_input_dim=6980,
                 additional_input_layers=1, skips=[4]):
        super(AppendVerticesNet, self).__init__(n_layers, width, positions_dim, directions_dim, 0, skips, 1)
        self.additional_input_dim = additional_input_dim
        self.vertices_net = nn.ModuleList()
        self.vertices_net.append(torch.nn.Linear(additional_input_dim, width))
        for i in range(additional_input_layers):
            self.vertices_net.append(torch.nn.Linear(width, width))

    def desc_for_rows(self) -> MlpDesc:
        """positions are plain per-ray columns (no encoder): pos_freqs = 0, add_dim = positions_dim."""
        d = _encoder_shape(self.direcions_dim)
        if d is None:
            raise RuntimeError(f"AppendVerticesNet: directions_dim={self.direcions_dim} is not a 3-channel encoding")
        return MlpDesc(self.n_layers, self.width, 0, 0, d[0], d[1], self.positions_dim, self._skip_mask(), 1, 0)

    def forward(self, x):
        """x [..., positions_dim + additional_input_dim + directions_dim] -> [..., 4] (:43-66)."""
        keep = self.additional_input_dim
        try:
            self.additional_input_dim = 0        # live columns: x[..., :positions_dim] and x[..., -directions_dim:]
            return RenderRayNet.forward(self, x)
        finally:
            self.additional_input_dim = keep

    def forward_rays(self, ray_inputs, directions, samples_per_ray, n):
        """ray_inputs [B, positions_dim] (the first positions_dim columns of the reference's input rows, a
        per-ray constant), directions [B,3] -> raw [n = B*samples_per_ray, 4]."""
        desc = self.desc_for_rows()
        _need_f32_cuda("AppendVerticesNet.forward_rays", ray_inputs, directions)
        add = ray_inputs.reshape(-1, self.positions_dim).contiguous()
        d = directions.reshape(-1, 3).contiguous()
        if self._layered:
            from . import layered
            from .ops import PositionalEncoder
            L, ident = _encoder_shape(self.direcions_dim)
            dn = d.detach() / torch.norm(d.detach(), dim=-1, keepdim=True)
            dd = PositionalEncoder(L, bool(ident)).encode(dn).repeat_interleave(int(samples_per_ray), dim=0)
            keep = self.additional_input_dim
            try:
                self.additional_input_dim = 0
                return layered.render_ray_net(self, add.repeat_interleave(int(samples_per_ray), dim=0), dd)
            finally:
                self.additional_input_dim = keep
        dummy_x = torch.zeros((n, 3), device=add.device, dtype=torch.float32)   # no position encoder: never read for slots
        if torch.is_grad_enabled() and (add.requires_grad or any(p.requires_grad for p in self.parameters())):
            # `add` keeps its graph: the vertex floats come from smpl_model(smpl_estimator(images)), whose parameters
            # AppendVerticesSolver optimises in their own group (solver/append_vertices_solver.py)
            return _FusedMlpFn.apply(self, desc, dummy_x, d.detach(), 0, int(samples_per_ray), add, *self._ordered_params())
        raw = torch.empty((n, 4), device=add.device, dtype=torch.float32)
        lib = _lib.load()
        if self.precision in ("bf16x6", "bf16x3", "f16x3") and self.width == 256:
            ns = {"bf16x6": 3, "bf16x3": 2, "f16x3": _lib.SPLIT_F16X3}[self.precision]
            packed = self.packed_weights_bf16(desc, ns)
            with torch.cuda.device(add.device), _lib.timed(f"mlp_fwd[n={n}]"):
                check(lib.snerf_mlp_fwd_bf16_f32(desc, ptr(packed), ns, ptr(dummy_x), ptr(d), 0, ptr(add), n,
                                                 int(samples_per_ray), ptr(raw), current_stream()), "snerf_mlp_fwd_bf16_f32")
            return raw
        packed = self.packed_weights(desc)
        ws, nb = _fold_workspace(lib.snerf_mlp_fold_workspace_bytes, desc, n, samples_per_ray, add.device)
        with torch.cuda.device(add.device), _lib.timed(f"mlp_fwd[n={n}]"):
            check(lib.snerf_mlp_fwd_ws_f32(desc, ptr(packed), ptr(dummy_x), ptr(d), 0, ptr(add), n, int(samples_per_ray),
                                           ptr(raw), ptr(ws), nb, current_stream()), "snerf_mlp_fwd_ws_f32")
        return raw
