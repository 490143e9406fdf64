"""The layer-by-layer path for widths the fused kernels do not cover (config_parser.py:20,24,30 accept ANY --netwidth /
--netwidth_fine / --netwidth_warp; the fused kernels keep a layer chain in registers: RenderRayNet up to 512 features,
WarpFieldNet up to 256).

A wider net runs one nn.Linear at a time with its activations in HBM - exactly the structure of models/render_ray_net.py:42-61 and
models/warp_field_net.py:17-21 - each layer an exact-fp32 MFMA GEMM of csrc/linear.hip (snerf_linear_fwd_f32 and its two
gradients, ReLU and bias fused into the epilogue) reading the weights in the reference's own [out, in] layout straight from the
parameters: no packed stream, nothing to refresh after an optimiser step.  Autograd sees one Function per layer, so every gradient
the fused path returns (parameters, encoded rows, warped samples and view directions, per-ray additional inputs) exists here too.
A skip layer reads its two input blocks ([h | encoding], :47-48) from two tensors into one accumulator - the concatenated row is
never built.  The encodings come from the library's encoder (snerf_posenc_f32); normalising, expanding per-ray rows to their
samples and the final concatenation are torch element-wise ops on the device.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, current_stream, ptr


def _rows(t: torch.Tensor) -> torch.Tensor:
    """2-D fp32 view with unit column stride (a column block of a wider tensor keeps its row stride: the kernels take it as ld)."""
    if t.dim() != 2 or t.dtype != torch.float32 or (t.shape[1] > 1 and t.stride(1) != 1) or t.stride(0) < t.shape[1]:
        t = t.reshape(-1, t.shape[-1]).float().contiguous()
    return t


class _LinearFn(torch.autograd.Function):
    """y = act(sum_i x_i w_i^T + bias): nn.Linear over an input that is the concatenation of the x_i (models/render_ray_net.py:47-48,
    :55) without building it.  forward(relu, bias, x_1, w_1, x_2, w_2, ...)."""

    @staticmethod
    def forward(ctx, relu, bias, *xw):
        xs, ws = [_rows(x.detach()) for x in xw[0::2]], [_rows(w.detach()) for w in xw[1::2]]
        n, m = xs[0].shape[0], ws[0].shape[0]
        dev = xs[0].device
        y = torch.empty((n, m), device=dev, dtype=torch.float32)
        lib = _lib.load()
        b = bias.detach().contiguous() if bias is not None else None
        with torch.cuda.device(dev), _lib.timed(f"linear_fwd[n={n}]"):
            for i, (x, w) in enumerate(zip(xs, ws)):
                if x.shape[0] != n or w.shape[0] != m or x.shape[1] != w.shape[1]:
                    raise RuntimeError(f"layered linear: block {i}: x {tuple(x.shape)} does not fit w {tuple(w.shape)}")
                last = i == len(xs) - 1
                check(lib.snerf_linear_fwd_f32(ptr(x), n, x.shape[1], x.stride(0), ptr(w), w.stride(0), m, ptr(b) if last else None,
                                               1 if i else 0, 1 if (relu and last) else 0, ptr(y), m, current_stream()),
                      "snerf_linear_fwd_f32")
        ctx.relu, ctx.nblk, ctx.has_bias = bool(relu), len(xs), bias is not None
        ctx.save_for_backward(y if relu else None, *xs, *ws)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, *rest = ctx.saved_tensors
        xs, ws = rest[:ctx.nblk], rest[ctx.nblk:]
        n, m = xs[0].shape[0], ws[0].shape[0]
        dev = dy.device
        lib = _lib.load()
        dy = dy.contiguous()
        need = ctx.needs_input_grad      # (relu, bias, x_1, w_1, ...)
        grads = [None, None] + [None] * (2 * ctx.nblk)
        with torch.cuda.device(dev), _lib.timed(f"linear_bwd[n={n}]"):
            if ctx.relu:
                dy = dy.clone()          # (autograd's buffer may be somebody else's gradient too)
                check(lib.snerf_relu_bwd_f32(ptr(dy), ptr(y), n, m, m, m, current_stream()), "snerf_relu_bwd_f32")
            db = torch.empty(m, device=dev, dtype=torch.float32) if (ctx.has_bias and need[1]) else None
            db_done = db is None
            for i, (x, w) in enumerate(zip(xs, ws)):
                k = x.shape[1]
                if need[2 + 2 * i]:
                    dx = torch.empty((n, k), device=dev, dtype=torch.float32)
                    check(lib.snerf_linear_bwd_input_f32(ptr(dy), n, m, m, ptr(w), w.stride(0), k, 0, ptr(dx), k, current_stream()),
                          "snerf_linear_bwd_input_f32")
                    grads[2 + 2 * i] = dx
                if need[3 + 2 * i]:
                    dw = torch.empty((m, k), device=dev, dtype=torch.float32)
                    scratch = torch.empty(int(lib.snerf_linear_bwd_weight_scratch_floats(n, m, k)), device=dev, dtype=torch.float32)
                    check(lib.snerf_linear_bwd_weight_f32(ptr(dy), n, m, m, ptr(x), x.stride(0), k, 0, ptr(dw), k,
                                                          None if db_done else ptr(db), ptr(scratch), current_stream()),
                          "snerf_linear_bwd_weight_f32")
                    db_done = True
                    grads[3 + 2 * i] = dw
            if not db_done:              # the bias alone wants its gradient
                scratch = torch.empty(int(lib.snerf_linear_bwd_weight_scratch_floats(n, m, 0)), device=dev, dtype=torch.float32)
                check(lib.snerf_linear_bwd_weight_f32(ptr(dy), n, m, m, None, 0, 0, 0, None, 0, ptr(db), ptr(scratch), current_stream()),
                      "snerf_linear_bwd_weight_f32")
            grads[1] = db
        return tuple(grads)


def linear(blocks, bias, relu=False):
    """blocks: [(x_i [n, k_i], w_i [m, k_i]), ...] -> act(sum x_i w_i^T + bias) [n, m]."""
    flat = []
    for x, w in blocks:
        flat += [x, w]
    return _LinearFn.apply(bool(relu), bias, *flat)


def render_ray_net(net, pp: torch.Tensor, dd: torch.Tensor | None) -> torch.Tensor:
    """models/render_ray_net.py:42-61 on pp [n, positions_dim + additional_input_dim] (the columns positions_pose_input reads) and dd
    [n, directions_dim] (None with use_directional_input = 0): raw [n, 4] = [rgb | sigma]."""
    W = net.width
    o = linear([(pp, net.positions_pose_input.weight)], net.positions_pose_input.bias, relu=True)                 # :45
    for i, layer in enumerate(net.positional_net):                                                               # :46-50
        if i in net.skips:
            o = linear([(o, layer.weight[:, :W]), (pp, layer.weight[:, W:])], layer.bias, relu=True)              # cat([o, pp]), :47-48
        else:
            o = linear([(o, layer.weight)], layer.bias, relu=True)
    o = linear([(o, net.additional_linear_layer.weight)], net.additional_linear_layer.bias)                      # :51
    sigma = linear([(o, net.sigma_out_layer.weight)], net.sigma_out_layer.bias)                                  # :52
    if net.use_directional_input:                                                                                # :54-57
        w = net.directional_input.weight
        o = linear([(o, w[:, :W]), (dd, w[:, W:])], net.directional_input.bias)
    else:
        o = linear([(o, net.directional_input.weight)], net.directional_input.bias)
    for layer in net.directional_net:                                                                            # :58-59
        o = linear([(o, layer.weight)], layer.bias, relu=True)
    rgb = linear([(o, net.rgb_out_layer.weight)], net.rgb_out_layer.bias)                                        # :60
    return torch.cat([rgb, sigma], -1)                                                                           # :61


def _per_sample(t: torch.Tensor, n: int, spr: int) -> torch.Tensor:
    """a per-ray tensor [n / spr, c] as per-sample rows [n, c] (the reference's expand + reshape, models/nerf_pipeline.py:31-36)"""
    return t if t.shape[0] == n else t.repeat_interleave(int(spr), dim=0)


def render_ray_net_fused(net, x, d, samples_per_ray, position_encoder, direction_encoder, additional, add_first):
    """RenderRayNet.forward_fused for a layered net: the encodings of models/nerf_pipeline.py:29-38, then the layers."""
    n = x.shape[0]
    enc_x = position_encoder.encode(x)                                               # [n, positions_dim]
    pp = enc_x
    if additional is not None:
        a = _per_sample(additional, n, samples_per_ray)
        pp = torch.cat([a, enc_x], -1) if add_first else torch.cat([enc_x, a], -1)    # models/append_smpl_params_pipeline.py:49-52
    dd = None
    if net.use_directional_input:
        dn = d / torch.norm(d, dim=-1, keepdim=True)                                  # :33-34
        dd = _per_sample(direction_encoder.encode(dn), n, samples_per_ray)
    return render_ray_net(net, pp.contiguous(), dd)


def warp_field_net(net, rows: torch.Tensor) -> torch.Tensor:
    """models/warp_field_net.py:17-21 on rows [n, positions_dim + pose_dim]: warp [n, 3]."""
    h = linear([(rows, net.linear1.weight)], net.linear1.bias, relu=True)
    return linear([(h, net.linear2.weight)], net.linear2.bias)
