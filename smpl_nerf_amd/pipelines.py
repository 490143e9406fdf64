"""NerfPipeline drop-in (models/nerf_pipeline.py:8-67).

Same constructor, same `forward(data)` contract - `data` is the list
[ray_samples[B,Nc,3], ray_translation[B,3], ray_direction[B,3], z_vals[B,Nc], rgb_truth[B,3]] a Solver
hands over (solver/nerf_solver.py:77-81) and the result is the 4-tuple
(rgb, rgb_fine, ray_samples_fine, densities) - but the whole march is five HIP launches:

    fused encode+MLP (coarse)  ->  composite  ->  inverse-CDF sampler+merge+points
                               ->  fused encode+MLP (fine)  ->  composite

No [B*N, 84] encoded input, no [B*N, 256] activation and no [B, 128, 63] gather operand is ever
materialised in HBM.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _NullCtx()


def _on_device(dev):
    """torch.cuda.device(dev), or nothing when `dev` is already current (the usual case; the context manager costs ~10 us per call,
    a fifth of the host time of a small render)."""
    return _NULL if torch.cuda.current_device() == (dev.index or 0) else torch.cuda.device(dev)


def _render_outputs(B, N, n_sample_tensors, dev):
    """rgb [B,3], rgb_fine [B,3], n_sample_tensors x [B,N,3], densities [B,N] as views of ONE allocation (16-byte aligned each)."""
    sizes = [B * 3, B * 3] + [B * N * 3] * n_sample_tensors + [B * N]
    offs, off = [], 0
    for n in sizes:
        offs.append(off)
        off += (n + 3) & ~3
    buf = torch.empty(off, device=dev, dtype=torch.float32)
    shapes = [(B, 3), (B, 3)] + [(B, N, 3)] * n_sample_tensors + [(B, N)]
    return [buf[o:o + n].view(sh) for o, n, sh in zip(offs, sizes, shapes)]


class PipelineArgs:
    """The fields of the reference's argparse namespace (config_parser.py) that the pipelines read.  Any object with these
    attributes works as `args` (the reference's own namespace included).

    The defaults here are the deterministic, runnable settings - NOT the reference's parser defaults in two fields:
    sigma_noise_std is 0 (config_parser.py:87 has 1: Gaussian noise on sigma, drawn even in eval mode, quirk Q3) and
    human_pose_encoding is 1 (config_parser.py:72 has 0, with which the reference's own smpl_nerf fine branch crashes,
    quirk Q5).  PipelineArgs.reference_defaults() returns the parser's values.

    strict_cumsum (not a reference field): 1 = the hierarchical sampler takes its normalising sums from
    torch's CPU kernel on this host, like the reference's CPU path, so that the sample indices equal the reference's bit
    for bit from the same weights (ops.reference_normalising_sum; costs a host round trip per call)."""

    def __init__(self, **kw):
        self.sigma_noise_std = 0.0
        self.white_background = 0
        self.run_fine = 1
        self.number_fine_samples = 128
        self.human_pose_encoding = 1
        self.strict_cumsum = 0
        self.u = None  # optional explicit linspace(0, 1, number_fine_samples) buffer (ops.uniform_u)
        self.__dict__.update(kw)

    @classmethod
    def reference_defaults(cls, **kw):
        """config_parser.py:27,71,72,87,89."""
        d = dict(sigma_noise_std=1.0, white_background=0, run_fine=1, number_fine_samples=128, human_pose_encoding=0)
        d.update(kw)
        return cls(**d)


class NerfPipeline(nn.Module):

    def __init__(self, model_coarse, model_fine, args, position_encoder, direction_encoder):
        super().__init__()
        self.device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")  # singe_sample_pipeline.py:11
        self.args = args
        self.model_coarse = model_coarse
        self.model_fine = model_fine
        self.position_encoder = position_encoder
        self.direction_encoder = direction_encoder

    def set_precision(self, precision: str):
        """Matrix-core arithmetic of every net of the pipeline: "fp32" (exact fp32 MFMA), "bf16x6" (split-bf16, fp32-class
        accuracy - the parity mode on the bf16 matrix cores) or "bf16x3" (split-bf16, ~2^-16 relative)."""
        if precision not in ("fp32", "bf16x6", "bf16x3", "f16x3"):
            raise ValueError(f"unknown precision {precision!r}")
        for m in self.modules():
            if hasattr(m, "precision") and m is not self:
                m.precision = precision
        return self

    def _noise(self, shape, device):
        # utils.py:171-173: drawn whenever sigma_noise_std > 0, also in eval mode (quirk Q3)
        std = getattr(self.args, "sigma_noise_std", 0.)
        return torch.normal(0, std, shape, device=device) if std > 0. else None

    # keep_fine = True (checkers only: tools/ab/fuzz_*.py): forward() leaves the merged depths and the hierarchical sample points
    # - which are not outputs of the reference's forward - in `last_fine`; off by default (a [B, N, 3] tensor pinned
    # between calls, 38 MB for a 128 x 128 frame)
    keep_fine = False
    last_fine = None

    def _single_call_ok(self, data) -> bool:
        """Inference (autograd off) goes through the single C-ABI call of render_rays() - what inference.py:251-252's
        `pipeline(data)` costs a host then is one call instead of five; same kernels, same results (tools/ab/fuzz_render.py holds
        them bit for bit).  The five-call form stays for autograd, for a _lib.profile() (which brackets every launch with an
        event pair: bench.py's roofline), for the strict sampler (a host round trip) and for the checkers' keep_fine."""
        from . import _lib
        if torch.is_grad_enabled() or _lib._PROFILE is not None or self.keep_fine or getattr(self.args, "strict_cumsum", 0):
            return False
        nets = [self.model_coarse, self.model_fine] + ([self.model_warp_field] if hasattr(self, "model_warp_field") else [])
        if len({getattr(m, "precision", "fp32") for m in nets}) != 1 or any(getattr(m, "_layered", False) for m in nets):
            return False      # (nets above the fused kernels' widths run layer by layer: layered.py)
        return all(torch.is_tensor(t) and t.is_cuda and (t.dtype == torch.float32) for t in data[:4])

    def forward(self, data):
        if self._single_call_ok(data):
            return self.render_rays(data)
        return self._forward_calls(data)

    def _forward_calls(self, data):
        ray_samples, ray_translation, ray_direction, z_vals, _ = data
        args = self.args
        B, Nc = z_vals.shape
        wb = bool(args.white_background)
        dev = ray_samples.device
        # coarse net on the given samples (:29-41)
        raw = self.model_coarse.forward_fused(ray_samples, ray_direction, Nc, self.position_encoder,
                                              self.direction_encoder)
        rgb, weights, densities = ops.composite(raw.view(B, Nc, 4), z_vals, ray_direction, wb,
                                                self._noise((B, Nc), dev))                        # :42
        if not args.run_fine:
            return rgb, rgb, ray_samples, densities                                               # :43-44
        # hierarchical samples (:47) and the fine net on them (:49-60)
        hs = ops.hierarchical_samples(ray_translation, ray_direction, z_vals, weights, args.number_fine_samples,
                                      strict=bool(getattr(args, "strict_cumsum", 0)))
        z_fine, ray_samples_fine = hs["z_fine"], hs["pts"]
        if self.keep_fine:
            self.last_fine = (z_fine, ray_samples_fine)
        N = z_fine.shape[1]
        raw_fine = self.model_fine.forward_fused(ray_samples_fine, ray_direction, N, self.position_encoder,
                                                 self.direction_encoder)
        rgb_fine, _, densities_fine = ops.composite(raw_fine.view(B, N, 4), z_fine, ray_direction, wb,
                                                    self._noise((B, N), dev), want_weights=False)  # :65
        return rgb, rgb_fine, ray_samples_fine, densities_fine                                    # :67


    def render_rays(self, data):
        """forward(data) for inference through the single C-ABI call snerf_render_rays_f32 (include/smplnerf.h):
        same outputs, one host call instead of five - the entry point a non-Python host binds.  No autograd."""
        from . import _lib
        from ._lib import check, ptr, current_stream
        if any(getattr(m, "_layered", False) for m in self.children()):      # (widths above the fused kernels': layered.py)
            with torch.no_grad():
                return self._forward_calls(data)
        ray_samples, ray_translation, ray_direction, z_vals, _ = data
        args = self.args
        B, Nc = z_vals.shape
        Nf = int(args.number_fine_samples) if args.run_fine else 0
        N = Nc + Nf
        dev = ray_samples.device
        mc, mf = self.model_coarse, self.model_fine
        if mc.precision != mf.precision:
            raise RuntimeError("render_rays: both nets must use the same precision mode")
        prec = {"fp32": 0, "bf16x3": 2, "bf16x6": 3, "f16x3": _lib.SPLIT_F16X3}[mc.precision]
        if mc.width != 256 or mf.width != 256:
            prec = 0      # the split-precision kernels exist for width 256; other widths run exact fp32 like forward() does
        descs, packed = [], []
        for m in (mc, mf):
            d = m.desc_for_encoders(self.position_encoder, self.direction_encoder, False)
            descs.append(d)
            packed.append(m.packed_weights(d) if prec == 0 else m.packed_weights_bf16(d, prec))
        lib = _lib.load()
        ws = torch.empty(int(lib.snerf_render_rays_workspace_bytes(B, Nc, Nf)), device=dev, dtype=torch.uint8)
        rgb, rgb_fine, samples_fine, dens = _render_outputs(B, N, 1, dev)
        u = ops.uniform_u(Nf, dev) if Nf else None
        nc, nf = self._noise((B, Nc), dev), (self._noise((B, N), dev) if Nf else None)
        x, o, d, z = (t.contiguous() for t in (ray_samples, ray_translation, ray_direction, z_vals))
        with _on_device(dev), _lib.timed("render_rays"):
            check(lib.snerf_render_rays_f32(descs[0], ptr(packed[0]), descs[1], ptr(packed[1]), prec, ptr(x), ptr(o),
                                            ptr(d), ptr(z), ptr(u), ptr(nc), ptr(nf), B, Nc, Nf,
                                            1 if args.white_background else 0, ptr(ws), ptr(rgb), ptr(rgb_fine),
                                            ptr(samples_fine), ptr(dens), current_stream()), "snerf_render_rays_f32")
        if not Nf:
            return rgb, rgb, ray_samples, dens      # (:43-44: the same tensor twice and the caller's own samples, quirk Q10)
        return rgb, rgb_fine, samples_fine, dens


class SmplNerfPipeline(NerfPipeline):
    """models/smpl_nerf_pipeline.py:7-100 drop-in: NerfPipeline with a pose-conditioned warp of the samples.

    data = [ray_samples, ray_translation, ray_direction, z_vals, goal_pose[B,69], rgb_truth]; returns
    (rgb, rgb_fine, warp_fine, ray_samples_fine, warped_samples_fine, densities_fine) or, with
    run_fine = 0, (rgb, rgb, warp, ray_samples, warped_samples, densities).  Quirks kept: joints 38 and 41
    are hard-coded (:28); the coarse compositing scales distances by |x' - o| per sample (:63) while the
    fine one uses the ray direction (:95-98); hierarchical samples are drawn on the un-warped ray (:68).
    human_pose_encoding = 0 (the parser default, config_parser.py:72): the warp net reads [x | two joint angles] un-encoded
    (:40-45; WarpFieldNet(positions_dim=3, pose_dim=2), train.py:111-114) - the fused warp kernel with an identity-only
    position "encoder".  The fine branch always builds encoded rows (:71-77), so with run_fine = 1 that mode fails at
    the fine warp evaluation exactly like the reference (quirk Q5: a RuntimeError from the 5-column linear1)."""

    class _RawPositions:
        """PositionalEncoder(0, True): the identity columns only (warp-net input of human_pose_encoding = 0)."""
        number_frequencies, include_identity, output_dim = 0, True, 1

    def __init__(self, model_coarse, model_fine, model_warp_field, args, position_encoder, direction_encoder,
                 human_pose_encoder):
        super().__init__(model_coarse, model_fine, args, position_encoder, direction_encoder)
        self.human_pose_encoder = human_pose_encoder
        self.model_warp_field = model_warp_field

    def _stage(self, net, samples, ray_translation, pose_enc, n_per_ray, warp_encoder=None):
        warp, warped, sdirs = self.model_warp_field.forward_fused(samples, pose_enc, ray_translation, n_per_ray,
                                                                  warp_encoder or self.position_encoder)
        raw = net.forward_fused(warped, sdirs, n_per_ray, self.position_encoder, self.direction_encoder)
        return warp, warped, sdirs, raw

    def _single_call_ok(self, data) -> bool:
        # (the one-call entry covers the full coarse + fine march with the encoded pose)
        return bool(self.args.human_pose_encoding and self.args.run_fine) and super()._single_call_ok(data) and \
            torch.is_tensor(data[4]) and data[4].is_cuda and data[4].dtype == torch.float32

    def _forward_calls(self, data):
        ray_samples, ray_translation, ray_direction, z_vals, goal_pose, _ = data
        args = self.args
        B, Nc = z_vals.shape
        wb = bool(args.white_background)
        dev = ray_samples.device
        goal_pose = torch.stack([goal_pose[:, 38], goal_pose[:, 41]], axis=-1).contiguous()         # :28
        pose_enc = self.human_pose_encoder.encode(goal_pose)                                       # :30
        if args.human_pose_encoding:                                                               # :37-39
            warp, warped, sdirs, raw = self._stage(self.model_coarse, ray_samples, ray_translation, pose_enc, Nc)
        else:                                                                                      # :40-45
            warp, warped, sdirs, raw = self._stage(self.model_coarse, ray_samples, ray_translation, goal_pose, Nc,
                                                   self._RawPositions)
        rgb, weights, densities = ops.composite(raw.view(B, Nc, 4), z_vals, sdirs.view(B, Nc, 3), wb,
                                                self._noise((B, Nc), dev))                        # :63
        if not args.run_fine:
            return rgb, rgb, warp.view(B, Nc, 3), ray_samples, warped.view(B, Nc, 3), densities   # :64-65
        hs = ops.hierarchical_samples(ray_translation, ray_direction, z_vals, weights, args.number_fine_samples,
                                      strict=bool(getattr(args, "strict_cumsum", 0)))
        z_fine, ray_samples_fine = hs["z_fine"], hs["pts"]                                         # :68
        if self.keep_fine:
            self.last_fine = (z_fine, ray_samples_fine)
        N = z_fine.shape[1]
        warp_f, warped_f, _, raw_f = self._stage(self.model_fine, ray_samples_fine, ray_translation, pose_enc, N)
        rgb_fine, _, densities_fine = ops.composite(raw_f.view(B, N, 4), z_fine, ray_direction, wb,
                                                    self._noise((B, N), dev), want_weights=False)  # :95-98
        return (rgb, rgb_fine, warp_f.view(B, N, 3), ray_samples_fine, warped_f.view(B, N, 3), densities_fine)  # :100


    def render_rays(self, data):
        """forward(data) for inference through the single C-ABI call snerf_render_rays_smpl_f32.  The one-call entry covers
        the full coarse + fine march with the encoded pose; run_fine = 0 and human_pose_encoding = 0 (three launches instead
        of eight) go through forward() under no_grad - the same kernels, the same results."""
        from . import _lib
        from ._lib import check, ptr, current_stream
        if any(getattr(m, "_layered", False) for m in self.children()):      # (widths above the fused kernels': layered.py)
            with torch.no_grad():
                return self._forward_calls(data)
        ray_samples, ray_translation, ray_direction, z_vals, goal_pose, _ = data
        args = self.args
        if not args.human_pose_encoding or not args.run_fine:
            with torch.no_grad():
                return self._forward_calls(data)
        B, Nc = z_vals.shape
        Nf = int(args.number_fine_samples)
        N = Nc + Nf
        dev = ray_samples.device
        mc, mf, mw = self.model_coarse, self.model_fine, self.model_warp_field
        if not (mc.precision == mf.precision == mw.precision):
            raise RuntimeError("render_rays: all nets must use the same precision mode (set_precision)")
        prec = {"fp32": 0, "bf16x3": 2, "bf16x6": 3, "f16x3": _lib.SPLIT_F16X3}[mc.precision]
        if any(m.width != 256 for m in (mc, mf, self.model_warp_field)):
            prec = 0      # (split precision: width 256 only - other widths run exact fp32, like forward())
        goal_pose = torch.stack([goal_pose[:, 38], goal_pose[:, 41]], axis=-1)
        pose_enc = self.human_pose_encoder.encode(goal_pose.contiguous()).contiguous()
        descs, packed = [], []
        for m in (mc, mf):
            d = m.desc_for_encoders(self.position_encoder, self.direction_encoder, False)
            descs.append(d)
            packed.append(m.packed_weights(d) if prec == 0 else m.packed_weights_bf16(d, prec))
        pe = self.position_encoder
        wdesc = _lib.WarpDesc(mw.width, pe.number_frequencies, 1 if pe.include_identity else 0, mw.direcions_dim)
        wpacked = mw._packed(wdesc) if prec == 0 else mw._packed_bf16(wdesc)
        lib = _lib.load()
        ws = torch.empty(int(lib.snerf_render_rays_smpl_workspace_bytes(B, Nc, Nf)), device=dev, dtype=torch.uint8)
        rgb, rgb_fine, warp_f, samples_f, warped_f, dens = _render_outputs(B, N, 3, dev)
        u = ops.uniform_u(Nf, dev)
        nc, nf = self._noise((B, Nc), dev), self._noise((B, N), dev)
        x, o, d, z = (t.contiguous() for t in (ray_samples, ray_translation, ray_direction, z_vals))
        with _on_device(dev), _lib.timed("render_rays_smpl"):
            check(lib.snerf_render_rays_smpl_f32(descs[0], ptr(packed[0]), descs[1], ptr(packed[1]), wdesc, ptr(wpacked), prec,
                                                 ptr(x), ptr(o), ptr(d), ptr(z), ptr(pose_enc), ptr(u), ptr(nc), ptr(nf), B, Nc,
                                                 Nf, 1 if args.white_background else 0, ptr(ws), ptr(rgb), ptr(rgb_fine),
                                                 ptr(warp_f), ptr(samples_f), ptr(warped_f), ptr(dens), current_stream()),
                  "snerf_render_rays_smpl_f32")
        return rgb, rgb_fine, warp_f, samples_f, warped_f, dens


class AppendVerticesPipeline(NerfPipeline):
    """models/append_vertices_pipeline.py:7-94 drop-in.  data = [ray_samples, ray_translation, ray_direction,
    z_vals, images (estimator input, e.g. image indices), rgb_truth]; `smpl_estimator(images)` returns
    (goal_poses, betas) and `smpl_model(betas=, return_verts=True, body_pose=, global_orient=)` an object
    with `.vertices [B, 6890, 3]` (smplx in the reference; any callable with that contract here).

    The reference concatenates the 20 670 vertex floats IN FRONT of every sample's encoding (:56-58) and the
    net reads its "positions" from the first positions_dim columns (models/append_vertices_net.py:44-47), so
    the effective network input is a per-ray constant: the first positions_dim vertex floats and the encoded
    ray direction.  That is what is evaluated here - without materialising the
    [B*N, 20754] fp32 input rows (83 KB per sample in the reference)."""

    def __init__(self, model_coarse, model_fine, smpl_estimator, smpl_model, args, position_encoder,
                 direction_encoder):
        super().__init__(model_coarse, model_fine, args, position_encoder, direction_encoder)
        self.smpl_estimator = smpl_estimator
        self.smpl_model = smpl_model

    def _single_call_ok(self, data) -> bool:
        return False      # (estimator and body model are torch modules in front of the nets: no single-call entry)

    def render_rays(self, data):
        with torch.no_grad():
            return self._forward_calls(data)

    def _forward_calls(self, data):
        ray_samples, ray_translation, ray_direction, z_vals, images, _ = data
        args = self.args
        B, Nc = z_vals.shape
        wb = bool(args.white_background)
        dev = ray_samples.device
        goal_poses, betas = self.smpl_estimator(images)                                            # :30
        global_orient = torch.zeros([1, 3], device=dev).expand(B, -1)                              # :13, :36
        goal_models = self.smpl_model(betas=betas, return_verts=True, body_pose=goal_poses,
                                      global_orient=global_orient)                                 # :38-39
        vertices_flat = goal_models.vertices.reshape(B, -1)                                        # :41
        pdim = self.model_coarse.positions_dim
        ray_inputs = vertices_flat[:, :pdim]          # the columns AppendVerticesNet.forward actually reads
        raw = self.model_coarse.forward_rays(ray_inputs, ray_direction, Nc, B * Nc)
        rgb, weights, densities = ops.composite(raw.view(B, Nc, 4), z_vals, ray_direction, wb,
                                                self._noise((B, Nc), dev))                        # :65
        if not args.run_fine:
            return rgb, rgb, ray_samples, densities                                               # :66-67
        hs = ops.hierarchical_samples(ray_translation, ray_direction, z_vals, weights, args.number_fine_samples,
                                      strict=bool(getattr(args, "strict_cumsum", 0)))
        z_fine, ray_samples_fine = hs["z_fine"], hs["pts"]                                         # :70
        if self.keep_fine:
            self.last_fine = (z_fine, ray_samples_fine)
        N = z_fine.shape[1]
        raw_f = self.model_fine.forward_rays(ray_inputs, ray_direction, N, B * N)
        rgb_fine, _, densities_fine = ops.composite(raw_f.view(B, N, 4), z_fine, ray_direction, wb,
                                                    self._noise((B, N), dev), want_weights=False)  # :92
        return rgb, rgb_fine, ray_samples_fine, densities_fine                                    # :94


class AppendSmplParamsPipeline(NerfPipeline):
    """models/append_smpl_params_pipeline.py:7-91 drop-in (the paper's headline model): NerfPipeline whose nets
    read [pose | PE(x) | PE(d)] rows, the pose being the 69 SMPL body-pose parameters of the ray's frame,
    optionally encoded (args.human_pose_encoding).  The pose columns are per-ray constants: they are read as
    the fused kernel's `additional` input (weight columns in front of the position encoding) instead of
    being expanded to every sample."""

    def __init__(self, model_coarse, model_fine, args, position_encoder, direction_encoder, human_pose_encoder):
        super().__init__(model_coarse, model_fine, args, position_encoder, direction_encoder)
        self.human_pose_encoder = human_pose_encoder

    def _select(self, goal_pose):
        return goal_pose

    def _single_call_ok(self, data) -> bool:
        return super()._single_call_ok(data) and torch.is_tensor(data[4]) and data[4].is_cuda and data[4].dtype == torch.float32

    def _forward_calls(self, data):
        ray_samples, ray_translation, ray_direction, z_vals, goal_pose, _ = data
        args = self.args
        B, Nc = z_vals.shape
        wb = bool(args.white_background)
        dev = ray_samples.device
        goal_pose = self._select(goal_pose).contiguous()
        pose = self.human_pose_encoder.encode(goal_pose) if args.human_pose_encoding else goal_pose   # :29-37
        raw = self.model_coarse.forward_fused(ray_samples, ray_direction, Nc, self.position_encoder,
                                              self.direction_encoder, additional=pose, add_first=True)   # :49-52
        rgb, weights, densities = ops.composite(raw.view(B, Nc, 4), z_vals, ray_direction, wb,
                                                self._noise((B, Nc), dev))                             # :55
        if not args.run_fine:
            return rgb, rgb, ray_samples, densities                                                    # :56-57
        hs = ops.hierarchical_samples(ray_translation, ray_direction, z_vals, weights, args.number_fine_samples,
                                      strict=bool(getattr(args, "strict_cumsum", 0)))
        z_fine, ray_samples_fine = hs["z_fine"], hs["pts"]                                              # :60
        if self.keep_fine:
            self.last_fine = (z_fine, ray_samples_fine)
        N = z_fine.shape[1]
        raw_f = self.model_fine.forward_fused(ray_samples_fine, ray_direction, N, self.position_encoder,
                                              self.direction_encoder, additional=pose, add_first=True)  # :77-81
        rgb_fine, _, densities_fine = ops.composite(raw_f.view(B, N, 4), z_fine, ray_direction, wb,
                                                    self._noise((B, N), dev), want_weights=False)       # :89
        return rgb, rgb_fine, ray_samples_fine, densities_fine                                         # :91


    def render_rays(self, data):
        """forward(data) for inference through the single C-ABI call snerf_render_rays_add_f32 (include/smplnerf.h): the pose
        rows as the nets' per-ray additional inputs, folded per ray inside the call in fp32.  Same outputs as forward()."""
        from . import _lib
        from ._lib import check, ptr, current_stream
        if any(getattr(m, "_layered", False) for m in self.children()):      # (widths above the fused kernels': layered.py)
            with torch.no_grad():
                return self._forward_calls(data)
        ray_samples, ray_translation, ray_direction, z_vals, goal_pose, _ = data
        args = self.args
        B, Nc = z_vals.shape
        Nf = int(args.number_fine_samples) if args.run_fine else 0
        N = Nc + Nf
        dev = ray_samples.device
        mc, mf = self.model_coarse, self.model_fine
        if mc.precision != mf.precision:
            raise RuntimeError("render_rays: both nets must use the same precision mode")
        prec = {"fp32": 0, "bf16x3": 2, "bf16x6": 3, "f16x3": _lib.SPLIT_F16X3}[mc.precision]
        if mc.width != 256 or mf.width != 256:
            prec = 0
        goal_pose = self._select(goal_pose).contiguous()
        pose = self.human_pose_encoder.encode(goal_pose) if args.human_pose_encoding else goal_pose
        pose = pose.reshape(B, -1).contiguous().float()
        descs, packed = [], []
        for m in (mc, mf):
            d = m.desc_for_encoders(self.position_encoder, self.direction_encoder, True)
            descs.append(d)
            packed.append(m.packed_weights(d) if prec == 0 else m.packed_weights_bf16(d, prec))
        lib = _lib.load()
        need = int(lib.snerf_render_rays_add_workspace_bytes(descs[0], descs[1], B, Nc, Nf))
        if need < 0:
            check(need, "snerf_render_rays_add_workspace_bytes")
        ws = torch.empty(need, device=dev, dtype=torch.uint8)
        rgb, rgb_fine, samples_fine, dens = _render_outputs(B, N, 1, dev)
        u = ops.uniform_u(Nf, dev) if Nf else None
        nc, nf = self._noise((B, Nc), dev), (self._noise((B, N), dev) if Nf else None)
        x, o, d, z = (t.contiguous() for t in (ray_samples, ray_translation, ray_direction, z_vals))
        with _on_device(dev), _lib.timed("render_rays_add"):
            check(lib.snerf_render_rays_add_f32(descs[0], ptr(packed[0]), descs[1], ptr(packed[1]), prec, ptr(x), ptr(o), ptr(d),
                                                ptr(z), ptr(pose), ptr(u), ptr(nc), ptr(nf), B, Nc, Nf,
                                                1 if args.white_background else 0, ptr(ws), ptr(rgb), ptr(rgb_fine),
                                                ptr(samples_fine), ptr(dens), current_stream()), "snerf_render_rays_add_f32")
        if not Nf:
            return rgb, rgb, ray_samples, dens      # (models/append_smpl_params_pipeline.py:56-57)
        return rgb, rgb_fine, samples_fine, dens


class AppendToNerfPipeline(AppendSmplParamsPipeline):
    """models/append_to_nerf_pipeline.py:7-90 drop-in: the same with only joints 38 and 41 of the pose (:26)."""

    def _select(self, goal_pose):
        return torch.stack([goal_pose[:, 38], goal_pose[:, 41]], axis=-1)
