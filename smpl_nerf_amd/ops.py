"""Drop-in operators with the reference's names, signatures and error behaviour, running on the
HIP library (include/smplnerf.h).  Reference counterparts:

    searchsorted        torchsearchsorted/src/torchsearchsorted/searchsorted.py:20-53
    PositionalEncoder   utils.py:114-131
    raw2outputs         utils.py:134-191
    sample_pdf          utils.py:194-228
    fine_sampling       utils.py:231-264

Every function takes CUDA (ROCm) fp32 tensors and launches on PyTorch's current stream.  There is
no CPU implementation here: a CPU tensor is an error, like a missing library.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, current_stream, ptr


def _need_cuda(name: str, t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError(f"smpl_nerf_amd: `{name}` must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"smpl_nerf_amd: `{name}` must be float32 (got {t.dtype})")


# ------------------------------------------------------------------------------------------------
# a6 searchsorted
# ------------------------------------------------------------------------------------------------
def searchsorted(a: torch.Tensor, v: torch.Tensor, out: Optional[torch.LongTensor] = None,
                 side="left") -> torch.LongTensor:
    """Same contract as torchsearchsorted.searchsorted (searchsorted.py:20-53): 2-D `a` (sorted rows)
    and `v`, equal row counts or one of them a single row, int64 result of shape
    (max(rows), v.shape[1]); `out` may be supplied."""
    # the reference's preconditions (searchsorted.py:21-36) as AssertionErrors, in this package's own words
    assert a.dim() == 2, f"searchsorted: `a` must have two dimensions (rows of sorted values), got {a.dim()}"
    assert v.dim() == 2, f"searchsorted: `v` must have two dimensions (rows of queries), got {v.dim()}"
    assert a.shape[0] == v.shape[0] or a.shape[0] == 1 or v.shape[0] == 1, (
        f"searchsorted: `a` has {a.shape[0]} rows and `v` {v.shape[0]}: the counts must match, or one side must be a single row "
        "that is broadcast")
    assert a.device == v.device, f"searchsorted: `a` is on {a.device} but `v` on {v.device}"
    result_shape = (max(a.shape[0], v.shape[0]), v.shape[1])
    if out is not None:
        assert out.device == a.device, f"searchsorted: `out` is on {out.device}, the inputs on {a.device}"
        assert out.dtype == torch.long, f"searchsorted: `out` must be int64 (torch.long), got {out.dtype}"
        assert out.shape == result_shape, f"searchsorted: `out` has shape {tuple(out.shape)}, the result has {result_shape}"
    else:
        out = torch.empty(result_shape, device=v.device, dtype=torch.long)
    for nm, t in (("a", a), ("v", v)):
        if not t.is_cuda:
            raise RuntimeError(f"smpl_nerf_amd: `{nm}` must live on the GPU (got {t.device}); there is no CPU path")
    code = _SEARCHSORTED_DTYPES.get(a.dtype)
    if code is None:     # the reference dispatches AT_DISPATCH_ALL_TYPES (searchsorted_cpu_wrapper.cpp:100): no half / bool
        raise RuntimeError(f"searchsorted: unsupported dtype {a.dtype}")
    if v.dtype != a.dtype:   # the reference reads `v` through a's scalar type and fails on a mismatch (:103)
        raise RuntimeError(f"searchsorted: `a` ({a.dtype}) and `v` ({v.dtype}) must have the same dtype")
    if not a.is_contiguous() or not v.is_contiguous() or not out.is_contiguous():
        # the reference's CUDA wrapper asserts contiguity (searchsorted_cuda_wrapper.cpp:5-7)
        raise RuntimeError("searchsorted: a, v and out must be contiguous")
    lib = _lib.load()
    with torch.cuda.device(a.device):
        check(lib.snerf_searchsorted(code, ptr(a), a.shape[0], a.shape[1], ptr(v), v.shape[0], v.shape[1], ptr(out),
                                     1 if side == "left" else 0, current_stream()), "snerf_searchsorted")
    return out


# torch dtype -> SNERF_DTYPE_* (include/smplnerf.h)
_SEARCHSORTED_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3, torch.int16: 4, torch.int8: 5,
                        torch.uint8: 6}


# ------------------------------------------------------------------------------------------------
# a1 PositionalEncoder
# ------------------------------------------------------------------------------------------------
class PositionalEncoder:
    """utils.py:114-131.  `output_dim` counts embedding functions per input channel exactly like the
    reference (train.py multiplies it by 3)."""

    def __init__(self, number_frequencies, include_identity):
        self.number_frequencies = int(number_frequencies)
        self.include_identity = include_identity
        self.output_dim = (1 if include_identity else 0) + 2 * self.number_frequencies

    def encode(self, coordinate: torch.Tensor) -> torch.Tensor:
        _need_cuda("coordinate", coordinate)
        if torch.is_grad_enabled() and coordinate.requires_grad:     # differentiable like utils.py:123-131
            return _PosEncFn.apply(coordinate, self.number_frequencies, 1 if self.include_identity else 0)
        return self._encode(coordinate)

    def _encode(self, coordinate: torch.Tensor) -> torch.Tensor:
        x = coordinate.contiguous()
        c = x.shape[-1]
        n = x.numel() // c if c else 0
        out = torch.empty(x.shape[:-1] + (c * self.output_dim,), device=x.device, dtype=torch.float32)
        if out.numel() == 0:
            return out
        lib = _lib.load()
        with torch.cuda.device(x.device):
            check(lib.snerf_posenc_f32(ptr(x), n, c, self.number_frequencies, 1 if self.include_identity else 0,
                                       ptr(out), current_stream()), "snerf_posenc_f32")
        return out


class _PosEncFn(torch.autograd.Function):
    """encode() under autograd: snerf_posenc_f32 forward, snerf_posenc_bwd_f32 backward."""

    @staticmethod
    def forward(ctx, x, L, identity):
        x = x.detach().contiguous()
        ctx.save_for_backward(x)
        ctx.cfg = (int(L), int(identity))
        return PositionalEncoder(L, bool(identity))._encode(x)

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        L, identity = ctx.cfg
        c = x.shape[-1]
        n = x.numel() // c if c else 0
        d_x = torch.zeros_like(x)
        if n == 0 or (identity + 2 * L) == 0:
            return d_x, None, None
        d_out = d_out.contiguous().float()
        lib = _lib.load()
        with torch.cuda.device(x.device):
            check(lib.snerf_posenc_bwd_f32(ptr(x), ptr(d_out), n, c, L, identity, ptr(d_x), current_stream()),
                  "snerf_posenc_bwd_f32")
        return d_x, None, None


# ------------------------------------------------------------------------------------------------
# a4 raw2outputs
# ------------------------------------------------------------------------------------------------
def _directions_arg(samples_directions: torch.Tensor, B: int, N: int):
    """(tensor, per_sample flag).  The pipelines pass ray directions as an expanded [B,N,3] view
    (models/nerf_pipeline.py:30-32); a zero stride over the sample axis is consumed as [B,3]."""
    d = samples_directions
    if d.dim() == 2 and d.shape == (B, 3):
        return d.contiguous(), 0
    if d.dim() == 3 and d.shape == (B, N, 3):
        if N == 1 or d.stride(1) == 0:
            return d[:, 0, :].contiguous(), 0
        return d.contiguous(), 1
    if d.dim() == 1 and d.shape[0] == 3:
        return d.expand(B, 3).contiguous(), 0
    raise RuntimeError(f"raw2outputs: samples_directions of shape {tuple(d.shape)} does not match raw [B={B}, N={N}]")


class _CompositeFn(torch.autograd.Function):
    """Differentiable alpha compositing, all of utils.py:134-191 under autograd: gradients arriving at rgb, weights and
    alpha (SmplNerfSolver's density loss reads the returned alpha, solver/smpl_nerf_solver.py:40) flow to raw and, where the
    caller's graph asks, to the directions (per-sample: SmplNerfPipeline's x' - o; per-ray) and to z_vals
    (snerf_composite_bwd_all_f32)."""

    @staticmethod
    def forward(ctx, raw, z_vals, dirs, per_sample, white_background, noise, want_weights, want_alpha):
        rgb, weights, alpha = _composite_launch(raw, z_vals, dirs, per_sample, white_background, noise,
                                                want_weights, want_alpha)
        ctx.save_for_backward(raw, z_vals, dirs, noise)
        ctx.cfg = (per_sample, white_background, bool(ctx.needs_input_grad[1]), bool(ctx.needs_input_grad[2]))
        ctx.set_materialize_grads(False)
        return rgb, weights, alpha

    @staticmethod
    def backward(ctx, d_rgb, d_w, d_a):
        raw, z_vals, dirs, noise = ctx.saved_tensors
        per_sample, wb, want_dz, want_ddirs = ctx.cfg
        if d_rgb is None and d_w is None and d_a is None:
            return (None,) * 8
        B, N = z_vals.shape
        d_rgb, d_w, d_a = (None if g is None else g.contiguous().float() for g in (d_rgb, d_w, d_a))
        d_raw = torch.empty_like(raw)
        d_dirs = torch.empty_like(dirs) if want_ddirs else None
        d_z = torch.empty_like(z_vals) if want_dz else None
        lib = _lib.load()
        with torch.cuda.device(raw.device), _lib.timed(f"composite_bwd[N={N}]"):
            check(lib.snerf_composite_bwd_all_f32(ptr(raw), ptr(z_vals), ptr(dirs), per_sample, ptr(noise), B, N,
                                                  1 if wb else 0, ptr(d_rgb), ptr(d_w), ptr(d_a), ptr(d_raw), ptr(d_dirs),
                                                  ptr(d_z), current_stream()), "snerf_composite_bwd_all_f32")
        return (d_raw, d_z, d_dirs) + (None,) * 5


def composite(raw, z_vals, samples_directions, white_background: bool, noise=None,
              want_weights=True, want_alpha=True):
    B, N = z_vals.shape
    for nm, t in (("raw", raw), ("z_vals", z_vals), ("samples_directions", samples_directions), ("noise", noise)):
        if t is not None:
            _need_cuda(nm, t)
    dirs, per_sample = _directions_arg(samples_directions, B, N)
    raw = raw.contiguous()
    z_vals = z_vals.contiguous()
    if noise is not None:
        noise = noise.contiguous()
    if torch.is_grad_enabled() and (raw.requires_grad or z_vals.requires_grad or dirs.requires_grad):
        return _CompositeFn.apply(raw.view(B, N, 4), z_vals, dirs, per_sample, bool(white_background), noise, want_weights,
                                  want_alpha)
    return _composite_launch(raw, z_vals, dirs, per_sample, white_background, noise, want_weights, want_alpha)


def _composite_launch(raw, z_vals, dirs, per_sample, white_background, noise, want_weights, want_alpha):
    B, N = z_vals.shape
    dev = raw.device
    rgb = torch.empty((B, 3), device=dev, dtype=torch.float32)
    weights = torch.empty((B, N), device=dev, dtype=torch.float32) if want_weights else None
    alpha = torch.empty((B, N), device=dev, dtype=torch.float32) if want_alpha else None
    lib = _lib.load()
    with torch.cuda.device(dev), _lib.timed(f"composite_fwd[N={N}]"):
        check(lib.snerf_composite_fwd_f32(ptr(raw), ptr(z_vals), ptr(dirs), per_sample, ptr(noise), B, N,
                                          1 if white_background else 0, ptr(rgb), ptr(weights), ptr(alpha),
                                          current_stream()), "snerf_composite_fwd_f32")
    return rgb, weights, alpha


def raw2outputs(raw: torch.Tensor, z_vals: torch.Tensor, samples_directions: torch.Tensor,
                args) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """utils.py:134-191: returns (rgb [B,3], weights [B,N], density/alpha [B,N]).  Reads
    args.sigma_noise_std and args.white_background; the Gaussian sigma noise is drawn here with
    torch.normal exactly where the reference draws it (utils.py:171-173) - also in eval mode."""
    for nm, t in (("raw", raw), ("z_vals", z_vals), ("samples_directions", samples_directions)):
        _need_cuda(nm, t)
    noise = None
    if z_vals.shape[-1] > 1 and args.sigma_noise_std > 0.:
        noise = torch.normal(0, args.sigma_noise_std, raw[..., 3].shape, device=raw.device)
    return composite(raw, z_vals, samples_directions, bool(args.white_background), noise)


# ------------------------------------------------------------------------------------------------
# a5 sample_pdf / fine_sampling
# ------------------------------------------------------------------------------------------------
_U_CACHE = {}


def uniform_u(n: int, device) -> torch.Tensor:
    """u = torch.linspace(0., 1., steps=n) (utils.py:206), evaluated by the same torch CPU kernel the
    reference's CPU path uses, then kept on the device."""
    key = (int(n), str(device))
    u = _U_CACHE.get(key)
    if u is None:
        u = torch.linspace(0., 1., steps=int(n)).to(device)
        _U_CACHE[key] = u
    return u


def reference_normalising_sum(interior_weights: torch.Tensor) -> torch.Tensor:
    """torch.sum(weights + 1e-5, -1) of utils.py:200-201 evaluated by torch's own CPU kernel on this host - exactly what
    the reference's CPU path computes here (a vectorised fp32 cascade: its bits depend on the host's SIMD width, which
    is why it cannot be restated portably) - returned on the weights' device.  Costs a device -> host -> device round
    trip; used only in strict mode (args.strict_cumsum)."""
    w = interior_weights.detach().to("cpu", torch.float32)
    return torch.sum(w + 1e-5, -1).to(interior_weights.device).contiguous()


def hierarchical_samples(ray_translation, ray_direction, z_vals, weights, number_fine_samples: int,
                         want_inds=False, want_samples=False, tot=None, strict=False):
    """One launch of snerf_sample_pdf_f32 -> dict(z_fine, pts, [inds], [z_samples]).  strict (or an explicit `tot` [B]):
    the normalising sums come from the reference's own host kernel, so the indices equal the reference's bit for bit
    from the same weights (snerf_sample_pdf_strict_f32)."""
    B, Nc = z_vals.shape
    Nf = int(number_fine_samples)
    dev = z_vals.device
    for nm, t in (("ray_translation", ray_translation), ("ray_direction", ray_direction), ("z_vals", z_vals),
                  ("weights", weights)):
        _need_cuda(nm, t)
    z_vals, weights = z_vals.contiguous(), weights.contiguous()
    o, d = ray_translation.contiguous(), ray_direction.contiguous()
    u = uniform_u(Nf, dev)
    z_fine = torch.empty((B, Nc + Nf), device=dev, dtype=torch.float32)
    pts = torch.empty((B, Nc + Nf, 3), device=dev, dtype=torch.float32)
    inds = torch.empty((B, Nf), device=dev, dtype=torch.long) if want_inds else None
    zs = torch.empty((B, Nf), device=dev, dtype=torch.float32) if want_samples else None
    lib = _lib.load()
    if tot is None and strict:
        tot = reference_normalising_sum(weights[:, 1:-1])
    if tot is not None:
        _need_cuda("tot", tot)
        tot = tot.reshape(-1).contiguous()
        if tot.shape[0] != B:
            raise RuntimeError(f"hierarchical_samples: tot must have one entry per ray ({B}), got {tot.shape[0]}")
        with torch.cuda.device(dev), _lib.timed("sample_pdf"):
            check(lib.snerf_sample_pdf_strict_f32(ptr(z_vals), ptr(weights), ptr(u), ptr(o), ptr(d), ptr(tot), B, Nc, Nf,
                                                  ptr(inds), ptr(zs), ptr(z_fine), ptr(pts), current_stream()),
                  "snerf_sample_pdf_strict_f32")
        return dict(z_fine=z_fine, pts=pts, inds=inds, z_samples=zs)
    with torch.cuda.device(dev), _lib.timed("sample_pdf"):
        check(lib.snerf_sample_pdf_f32(ptr(z_vals), ptr(weights), ptr(u), ptr(o), ptr(d), B, Nc, Nf, ptr(inds),
                                       ptr(zs), ptr(z_fine), ptr(pts), current_stream()), "snerf_sample_pdf_f32")
    return dict(z_fine=z_fine, pts=pts, inds=inds, z_samples=zs)


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, args) -> torch.Tensor:
    """utils.py:194-228 with the reference's calling convention: bins [B,Nb] (the coarse midpoints),
    weights [B,Nb-1] (the interior coarse weights).  Returns the Nf samples [B, Nf]."""
    _need_cuda("bins", bins)
    _need_cuda("weights", weights)
    B, Nb = bins.shape
    assert weights.shape == (B, Nb - 1), "weights must have one entry less than bins"
    Nf = int(args.number_fine_samples)
    want_grad = torch.is_grad_enabled() and (bins.requires_grad or weights.requires_grad)
    bins_in, weights_in = bins, weights
    bins, weights = bins.detach().contiguous(), weights.detach().contiguous()
    u = uniform_u(Nf, bins.device)
    zs = torch.empty((B, Nf), device=bins.device, dtype=torch.float32)
    inds = torch.empty((B, Nf), device=bins.device, dtype=torch.long) if want_grad else None
    lib = _lib.load()
    tot = None
    with torch.cuda.device(bins.device):
        if getattr(args, "strict_cumsum", 0):
            tot = reference_normalising_sum(weights)
            check(lib.snerf_sample_pdf_bins_strict_f32(ptr(bins), ptr(weights), ptr(u), ptr(tot), B, Nb, Nf, ptr(inds), ptr(zs),
                                                       current_stream()), "snerf_sample_pdf_bins_strict_f32")
        else:
            check(lib.snerf_sample_pdf_bins_f32(ptr(bins), ptr(weights), ptr(u), B, Nb, Nf, ptr(inds), ptr(zs),
                                                current_stream()), "snerf_sample_pdf_bins_f32")
    if not want_grad:
        return zs
    return _SamplePdfGrad.apply(bins_in, weights_in, zs, inds, u, tot)


class _SamplePdfGrad(torch.autograd.Function):
    """sample_pdf under autograd (the reference's fine_sampling detaches the result, utils.py:260, but sample_pdf itself is
    differentiable w.r.t. bins and weights, utils.py:200-228).  The values are the forward kernel's; the gradient is the
    reference's with the searchsorted indices held fixed (integers carry no gradient there either) - a piecewise-linear map
    whose pieces the kernel's `inds` select, differentiated by snerf_sample_pdf_bins_bwd_f32."""

    @staticmethod
    def forward(ctx, bins, weights, zs, inds, u, tot):
        # dense copies: the backward kernel walks [B, Nb] / [B, Nb - 1] rows by pointer, and the reference's own caller passes a
        # strided view (`weights[..., 1:-1]`, utils.py:259)
        ctx.save_for_backward(bins.detach().contiguous(), weights.detach().contiguous(), inds, u, tot)
        return zs

    @staticmethod
    def backward(ctx, d_zs):
        bins, weights, inds, u, tot = ctx.saved_tensors
        B, Nb = bins.shape
        d_zs = d_zs.contiguous().float()
        gb = torch.empty(bins.shape, device=bins.device, dtype=torch.float32)
        gw = torch.empty(weights.shape, device=bins.device, dtype=torch.float32)
        lib = _lib.load()
        with torch.cuda.device(bins.device):
            check(lib.snerf_sample_pdf_bins_bwd_f32(ptr(bins), ptr(weights), ptr(u), ptr(inds), ptr(tot), ptr(d_zs), B, Nb, inds.shape[1],
                                                    ptr(gb), ptr(gw), current_stream()), "snerf_sample_pdf_bins_bwd_f32")
        return gb, gw, None, None, None, None


def fine_sampling(ray_translation: torch.Tensor, samples_directions: torch.Tensor, z_vals: torch.Tensor,
                  weights: torch.Tensor, args) -> Tuple[torch.Tensor, torch.Tensor]:
    """utils.py:231-264 -> (z_vals [B,Nc+Nf] ascending, ray_samples_fine [B,Nc+Nf,3]); the samples are
    detached like in the reference (utils.py:260)."""
    for nm, t in (("ray_translation", ray_translation), ("samples_directions", samples_directions),
                  ("z_vals", z_vals), ("weights", weights)):
        _need_cuda(nm, t)
    r = hierarchical_samples(ray_translation.detach(), samples_directions.detach(), z_vals.detach(),
                             weights.detach(), args.number_fine_samples, strict=bool(getattr(args, "strict_cumsum", 0)))
    return r["z_fine"], r["pts"]
