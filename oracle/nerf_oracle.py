"""CPU oracle for the NeRF ray-march hot path of HannesStark/SMPL-NeRF.

TEST INFRASTRUCTURE ONLY.  This module is a plain-numpy restatement of the
reference's algorithm (fp32, same op order as the reference's torch code).  It
is imported only by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg, never by the product path (smpl_nerf_amd/ calls the HIP
library through the C-ABI and fails loudly if it is missing).

Parity status: PINNED.  Every function below is checked in
tests/test_oracle_golden.py against golden vectors captured by importing the
reference itself in the build container (tests/golden/make_golden.py, which is
committed next to the vectors).  The reference's one native unit
(torchsearchsorted) does not build against this image's torch (ATen API break
at torchsearchsorted/src/cpu/searchsorted_cpu_wrapper.cpp:100), so its
algorithm is restated in C in oracle/searchsorted_ref.c and pinned against
numpy.searchsorted, which is the reference's own test oracle
(torchsearchsorted/test/test_searchsorted.py:41-44).

All file:line citations are relative to the reference repository root.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------------
# a1  PositionalEncoder  (utils.py:114-131)
# ----------------------------------------------------------------------------------------------
class PositionalEncoder:
    """utils.py:114-131.  freq_bands = 2**linspace(0, L-1, L); no pi factor;
    frequency-major layout [sin(f*x) | cos(f*x)] per frequency, identity first
    when requested.  `output_dim` counts embedding functions (per input
    channel), exactly like the reference (utils.py:118-128)."""

    def __init__(self, number_frequencies: int, include_identity):
        self.number_frequencies = int(number_frequencies)
        self.include_identity = bool(include_identity)
        self.freq_bands = (F32(2.0) ** np.linspace(0.0, number_frequencies - 1, number_frequencies)
                           ).astype(F32) if number_frequencies > 0 else np.zeros((0,), F32)
        self.output_dim = (1 if include_identity else 0) + 2 * self.number_frequencies

    def encode(self, coordinate: np.ndarray) -> np.ndarray:
        x = np.asarray(coordinate, dtype=F32)
        outs = []
        if self.include_identity:
            outs.append(x)
        for f in self.freq_bands:
            xf = x * f  # utils.py:127 periodic_fn(x * freq)
            outs.append(np.sin(xf))
            outs.append(np.cos(xf))
        return np.concatenate(outs, -1).astype(F32)


# ----------------------------------------------------------------------------------------------
# a2  RenderRayNet  (models/render_ray_net.py:8-61)  /  WarpFieldNet (models/warp_field_net.py:8-22)
# ----------------------------------------------------------------------------------------------
def _linear(x, w, b):
    # torch.nn.Linear: y = x W^T + b, weight stored [out, in]
    return (x @ w.T + b).astype(F32)


def _relu(x):
    return np.maximum(x, F32(0))


def render_ray_net_param_shapes(n_layers=8, width=256, positions_dim=60, directions_dim=24,
                                additional_input_dim=0, skips=(4,), use_directional_input=1):
    """state_dict keys and shapes in registration order (models/render_ray_net.py:19-40)."""
    pin = positions_dim + additional_input_dim
    shapes = [("positions_pose_input", (width, pin))]
    for i in range(n_layers - 1):
        shapes.append((f"positional_net.{i}", (width, width + pin if i in skips else width)))
    shapes.append(("additional_linear_layer", (width, width)))
    shapes.append(("sigma_out_layer", (1, width)))
    dw = width // 2
    shapes.append(("directional_input", (dw, width + directions_dim if use_directional_input else width)))
    shapes.append(("directional_net.0", (dw, dw)))
    shapes.append(("rgb_out_layer", (3, dw)))
    out = []
    for name, shp in shapes:
        out.append((name + ".weight", shp))
        out.append((name + ".bias", (shp[0],)))
    return out


def render_ray_net_forward(params: dict, x: np.ndarray, n_layers=8, positions_dim=60, directions_dim=24,
                           additional_input_dim=0, skips=(4,), use_directional_input=1) -> np.ndarray:
    """models/render_ray_net.py:42-61.  x: [M, positions_dim+add+directions_dim] -> [M, 4] = [rgb | sigma]."""
    x = np.asarray(x, dtype=F32)
    pin = positions_dim + additional_input_dim
    positions_pose, directions = x[..., :pin], x[..., x.shape[-1] - directions_dim:]
    o = _relu(_linear(positions_pose, params["positions_pose_input.weight"], params["positions_pose_input.bias"]))
    for i in range(n_layers - 1):
        w, b = params[f"positional_net.{i}.weight"], params[f"positional_net.{i}.bias"]
        if i in skips:
            o = _relu(_linear(np.concatenate([o, positions_pose], -1), w, b))
        else:
            o = _relu(_linear(o, w, b))
    o = _linear(o, params["additional_linear_layer.weight"], params["additional_linear_layer.bias"])
    sigma = _linear(o, params["sigma_out_layer.weight"], params["sigma_out_layer.bias"])
    if use_directional_input:
        o = _linear(np.concatenate([o, directions], -1), params["directional_input.weight"],
                    params["directional_input.bias"])
    else:
        o = _linear(o, params["directional_input.weight"], params["directional_input.bias"])
    o = _relu(_linear(o, params["directional_net.0.weight"], params["directional_net.0.bias"]))
    rgb = _linear(o, params["rgb_out_layer.weight"], params["rgb_out_layer.bias"])
    return np.concatenate([rgb, sigma], -1).astype(F32)


def warp_field_net_forward(params: dict, x: np.ndarray) -> np.ndarray:
    """models/warp_field_net.py:17-21: linear1 -> relu -> linear2 (n_layers is ignored, :14-15)."""
    h = _relu(_linear(np.asarray(x, F32), params["linear1.weight"], params["linear1.bias"]))
    return _linear(h, params["linear2.weight"], params["linear2.bias"])


# ----------------------------------------------------------------------------------------------
# a4  raw2outputs  (utils.py:134-191)
# ----------------------------------------------------------------------------------------------
def sigmoid(x):
    x = np.asarray(x, F32)
    return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)


def raw2outputs(raw, z_vals, samples_directions, white_background=0, noise=None):
    """utils.py:134-191.  `noise` replaces torch.normal(0, sigma_noise_std) (utils.py:172-174);
    None == sigma_noise_std 0.  Returns (rgb[B,3], weights[B,N], density(alpha)[B,N])."""
    raw = np.asarray(raw, F32)
    z_vals = np.asarray(z_vals, F32)
    samples_directions = np.asarray(samples_directions, F32)
    dists = z_vals[..., 1:] - z_vals[..., :-1]                                           # :161
    dists = np.concatenate([dists, np.full(dists[..., :1].shape, 1e10, F32)], -1)       # :162-163
    norm = np.sqrt(np.sum(samples_directions * samples_directions, -1, dtype=F32)).astype(F32)
    dists = (dists * norm).astype(F32)                                                   # :165
    rgb = sigmoid(raw[..., :3])                                                          # :167
    if z_vals.shape[-1] == 1:                                                            # :168-169
        B = raw.shape[0]
        return rgb.reshape(B, 3), np.ones((B, 1), F32), np.ones((B, 1), F32)
    sig = raw[..., 3] if noise is None else (raw[..., 3] + np.asarray(noise, F32)).astype(F32)
    density = (F32(1) - np.exp(-_relu(sig) * dists)).astype(F32)                         # :159,173
    one_minus = (F32(1) - density + F32(1e-10)).astype(F32)                              # :174
    excl = np.concatenate([np.ones(one_minus[..., :1].shape, F32), one_minus[..., :-1]], -1)  # :177-178
    # torch.cumprod on CPU accumulates in double (at::acc_type<float,false>) and rounds each output.
    trans = np.cumprod(excl.astype(np.float64), -1).astype(F32)
    weights = (density * trans).astype(F32)                                              # :179
    rgb_map = np.sum(weights[..., None] * rgb, -2, dtype=F32)                            # :180
    acc_map = np.sum(weights, -1, dtype=F32)                                             # :185
    if white_background:
        rgb_map = (rgb_map + (F32(1) - acc_map[..., None])).astype(F32)                  # :186-187
    return rgb_map.astype(F32), weights, density


# ----------------------------------------------------------------------------------------------
# a6  searchsorted  (torchsearchsorted/src/torchsearchsorted/searchsorted.py:20-53)
# ----------------------------------------------------------------------------------------------
def searchsorted(a: np.ndarray, v: np.ndarray, side: str = "left") -> np.ndarray:
    """Batched searchsorted with row broadcast, identical to the reference's own numpy
    oracle (torchsearchsorted/src/torchsearchsorted/utils.py:4-15).  int64 result."""
    a = np.asarray(a)
    v = np.asarray(v)
    assert a.ndim == 2 and v.ndim == 2
    assert a.shape[0] == v.shape[0] or a.shape[0] == 1 or v.shape[0] == 1
    nrow = max(a.shape[0], v.shape[0])
    out = np.empty((nrow, v.shape[1]), np.int64)
    for r in range(nrow):
        ar = a[0] if a.shape[0] == 1 else a[r]
        vr = v[0] if v.shape[0] == 1 else v[r]
        out[r] = np.searchsorted(ar, vr, side=side)
    return out


def _searchsorted_rows(a, v, side):
    """Vectorised equivalent of `searchsorted` for equal row counts (counting form:
    #elements < v (left) or <= v (right) of a sorted row)."""
    if side == "left":
        return np.sum(a[:, None, :] < v[:, :, None], -1).astype(np.int64)
    return np.sum(a[:, None, :] <= v[:, :, None], -1).astype(np.int64)


# ----------------------------------------------------------------------------------------------
# a5  sample_pdf / fine_sampling  (utils.py:194-264)
# ----------------------------------------------------------------------------------------------
def linspace01(n: int) -> np.ndarray:
    """u = linspace(0, 1, n) in fp32 (utils.py:206), two-sided form: start + step*i for the first
    half, end - step*(n-1-i) for the second, step = 1/(n-1).  torch's CPU kernel evaluates the same
    two-sided form in SIMD chunks with fused multiply-adds, so a few entries differ from this by
    one ulp depending on the host's vector width; `u` is therefore an explicit INPUT of the sampler
    (here and in the C-ABI): callers that need the reference's exact bits pass torch.linspace's
    output, as the golden tests do."""
    if n == 1:
        return np.zeros((1,), F32)
    step = F32(1.0) / F32(n - 1)
    i = np.arange(n)
    lo = (F32(0) + step * i.astype(F32)).astype(F32)
    hi = (F32(1) - step * (n - 1 - i).astype(F32)).astype(F32)
    return np.where(i < n // 2, lo, hi).astype(F32)


def cdf_from_weights(weights, tot=None):
    """utils.py:200-203: weights+1e-5 -> pdf -> cdf with a leading 0 ([..., len(weights)+1]).

    The normalising sum (utils.py:201) is evaluated in float64 and rounded once; the cumsum
    (utils.py:202) accumulates in float64 and rounds every prefix.  The latter is exactly what
    torch's CPU cumsum does (acc_type<float> = double - verified bit for bit in the build
    container); the former is within 1 ulp of torch's vectorised fp32 sum (whose bits depend on
    the host's SIMD width) and makes the result independent of summation order, which is what lets
    the HIP kernel reproduce this oracle bit for bit.

    Strict mode: `tot` [..., 1] = the sums as torch.sum returned them on the reference's host (recorded in the golden
    fixtures); with it the cdf - and everything downstream - equals the reference's bit for bit."""
    w = (np.asarray(weights, F32) + F32(1e-5)).astype(F32)                               # :200
    if tot is None:
        tot = np.sum(w.astype(np.float64), -1, keepdims=True).astype(F32)
    else:
        tot = np.asarray(tot, F32).reshape(w.shape[:-1] + (1,))
    pdf = (w / tot).astype(F32)                                                          # :201
    cdf = np.cumsum(pdf.astype(np.float64), -1).astype(F32)                              # :202
    return np.concatenate([np.zeros(cdf[..., :1].shape, F32), cdf], -1)                  # :203


def invert_cdf(bins, cdf, u):
    """utils.py:212-226: inds = searchsorted(cdf, u, 'right'), clamp, gather, lerp.
    bins, cdf: [B, Nb]; u: [B, Nf] -> (inds int64 [B, Nf], samples fp32 [B, Nf])."""
    bins = np.asarray(bins, F32)
    cdf = np.asarray(cdf, F32)
    u = np.asarray(u, F32)
    inds = _searchsorted_rows(cdf, u, "right")                                           # :212
    below = np.maximum(0, inds - 1)                                                      # :213
    above = np.minimum(cdf.shape[-1] - 1, inds)                                          # :214
    cdf_g0 = np.take_along_axis(cdf, below, -1)                                          # :219-221
    cdf_g1 = np.take_along_axis(cdf, above, -1)
    bins_g0 = np.take_along_axis(bins, below, -1)
    bins_g1 = np.take_along_axis(bins, above, -1)
    denom = (cdf_g1 - cdf_g0).astype(F32)                                                # :223
    denom = np.where(denom < F32(1e-5), F32(1), denom).astype(F32)                       # :224
    t = ((u - cdf_g0) / denom).astype(F32)                                               # :225
    samples = (bins_g0 + t * (bins_g1 - bins_g0)).astype(F32)                            # :226
    return inds, samples


def sample_pdf_detail(bins, weights, number_fine_samples, u=None, tot=None):
    """utils.py:194-228 with every intermediate returned (cdf, u, inds, samples)."""
    cdf = cdf_from_weights(weights, tot)
    u1 = linspace01(number_fine_samples) if u is None else np.asarray(u, F32).reshape(-1)  # :206
    u = np.broadcast_to(u1, cdf.shape[:-1] + (number_fine_samples,)).copy()              # :207-210
    inds, samples = invert_cdf(bins, cdf, u)
    return dict(cdf=cdf, u=u, inds=inds, samples=samples)


def sample_pdf(bins, weights, number_fine_samples, u=None, tot=None):
    return sample_pdf_detail(bins, weights, number_fine_samples, u, tot)["samples"]


def fine_sampling(ray_translation, samples_directions, z_vals, weights, number_fine_samples, u=None, tot=None):
    """utils.py:231-264 -> (z_vals[B,Nc+Nf] sorted, ray_samples_fine[B,Nc+Nf,3])."""
    z_vals = np.asarray(z_vals, F32)
    weights = np.asarray(weights, F32)
    o = np.asarray(ray_translation, F32)
    d = np.asarray(samples_directions, F32)
    z_mid = (F32(0.5) * (z_vals[..., 1:] + z_vals[..., :-1])).astype(F32)                # :258
    z_samples = sample_pdf(z_mid, weights[..., 1:-1], number_fine_samples, u, tot)      # :259
    z_all = np.sort(np.concatenate([z_vals, z_samples], -1), -1)                        # :261
    pts = (o[..., None, :] + d[..., None, :] * z_all[..., :, None]).astype(F32)          # :262-263
    return z_all.astype(F32), pts


# ----------------------------------------------------------------------------------------------
# a3  NerfPipeline.forward  (models/nerf_pipeline.py:14-67)
# ----------------------------------------------------------------------------------------------
class Args:
    """Duck-typed namespace with the fields the hot path reads (config_parser.py)."""

    def __init__(self, **kw):
        self.sigma_noise_std = 0.0
        self.white_background = 0
        self.run_fine = 1
        self.number_fine_samples = 128
        self.human_pose_encoding = 1
        self.default_device = "cpu"
        self.u = None            # optional explicit linspace buffer (see linspace01)
        self.__dict__.update(kw)


def _normalize(v):
    n = np.sqrt(np.sum(v * v, -1, keepdims=True, dtype=F32)).astype(F32)
    return (v / n).astype(F32)


def nerf_pipeline_forward(params_coarse, params_fine, args, position_encoder, direction_encoder, data,
                          net_kw=None):
    """models/nerf_pipeline.py:14-67.  data = [ray_samples, ray_translation, ray_direction, z_vals, rgb_truth]."""
    net_kw = net_kw or {}
    ray_samples, ray_translation, ray_direction, z_vals = [np.asarray(t, F32) for t in data[:4]]
    B, Nc = ray_samples.shape[:2]
    samples_encoding = position_encoder.encode(ray_samples)                              # :29
    dirs = np.broadcast_to(ray_direction[:, None, :], (B, Nc, 3))                        # :30-32
    dir_norm = _normalize(dirs)                                                          # :33-34
    directions_encoding = direction_encoder.encode(dir_norm)                             # :35
    inputs = np.concatenate([samples_encoding.reshape(B * Nc, -1),
                             directions_encoding.reshape(B * Nc, -1)], -1)               # :37-38
    raw = render_ray_net_forward(params_coarse, inputs, **net_kw).reshape(B, Nc, 4)      # :39-41
    rgb, weights, densities = raw2outputs(raw, z_vals, dirs, args.white_background)      # :42
    if not args.run_fine:
        return rgb, rgb, ray_samples, densities                                          # :43-44
    z_f, pts_f = fine_sampling(ray_translation, ray_direction, z_vals, weights,
                               args.number_fine_samples, getattr(args, "u", None))      # :47
    N = pts_f.shape[1]
    enc_f = position_encoder.encode(pts_f)                                               # :49
    dir_enc_f = np.broadcast_to(directions_encoding[:, :1, :], (B, N, directions_encoding.shape[-1]))  # :51-53
    inputs_f = np.concatenate([enc_f.reshape(B * N, -1), dir_enc_f.reshape(B * N, -1)], -1)
    raw_f = render_ray_net_forward(params_fine, inputs_f, **net_kw).reshape(B, N, 4)     # :56-60
    dirs_f = np.broadcast_to(ray_direction[:, None, :], (B, N, 3))                       # :62-64
    rgb_fine, _, dens_f = raw2outputs(raw_f, z_f, dirs_f, args.white_background)         # :65
    return rgb, rgb_fine, pts_f, dens_f                                                  # :67


# ----------------------------------------------------------------------------------------------
# a7  SmplNerfPipeline.forward  (models/smpl_nerf_pipeline.py:16-100), human_pose_encoding=1
# ----------------------------------------------------------------------------------------------
def smpl_nerf_pipeline_forward(params_coarse, params_fine, params_warp, args, position_encoder,
                               direction_encoder, human_pose_encoder, data, net_kw=None):
    net_kw = net_kw or {}
    ray_samples, ray_translation, ray_direction, z_vals, goal_pose = [np.asarray(t, F32) for t in data[:5]]
    B, Nc = ray_samples.shape[:2]
    goal_pose = np.stack([goal_pose[:, 38], goal_pose[:, 41]], -1)                       # :28
    pose_enc_flat = human_pose_encoder.encode(goal_pose)                                 # :30

    def stage(pts):
        n = pts.shape[1]
        enc = position_encoder.encode(pts)                                               # :35 / :71
        if args.human_pose_encoding:
            pe = np.broadcast_to(pose_enc_flat[:, None, :], (B, n, pose_enc_flat.shape[-1]))
            winp = np.concatenate([enc.reshape(B * n, -1), pe.reshape(B * n, -1)], -1)   # :38-39 / :75-76
        else:
            gp = np.broadcast_to(goal_pose[:, None, :], (B, n, 2))
            winp = np.concatenate([pts.reshape(B * n, 3), gp.reshape(B * n, 2)], -1)     # :41-45
        warp = warp_field_net_forward(params_warp, winp).reshape(pts.shape)              # :48 / :77
        warped = (pts + warp).astype(F32)                                                # :49 / :79
        enc_w = position_encoder.encode(warped)                                          # :50 / :80
        sdirs = (warped - ray_translation[:, None, :]).astype(F32)                       # :52 / :82
        denc = direction_encoder.encode(_normalize(sdirs))                               # :54-56 / :84-86
        inp = np.concatenate([enc_w.reshape(B * n, -1), denc.reshape(B * n, -1)], -1)
        return warp, warped, sdirs, inp

    warp, warped, sdirs, inp = stage(ray_samples)
    raw = render_ray_net_forward(params_coarse, inp, **net_kw).reshape(B, Nc, 4)         # :60-62
    rgb, weights, densities = raw2outputs(raw, z_vals, sdirs, args.white_background)     # :63 (dirs = x'-o)
    if not args.run_fine:
        return rgb, rgb, warp, ray_samples, warped, densities                            # :64-65
    z_f, pts_f = fine_sampling(ray_translation, ray_direction, z_vals, weights,
                               args.number_fine_samples, getattr(args, "u", None))      # :68
    N = pts_f.shape[1]
    warp_f, warped_f, _, inp_f = stage(pts_f)
    raw_f = render_ray_net_forward(params_fine, inp_f, **net_kw).reshape(B, N, 4)        # :89-93
    dirs_f = np.broadcast_to(ray_direction[:, None, :], (B, N, 3))                       # :95-97 (ray dir)
    rgb_fine, _, dens_f = raw2outputs(raw_f, z_f, dirs_f, args.white_background)         # :98
    return rgb, rgb_fine, warp_f, pts_f, warped_f, dens_f                                # :100


# ----------------------------------------------------------------------------------------------
# f-4  AppendSmplParamsPipeline / AppendToNerfPipeline  (models/append_smpl_params_pipeline.py:14-91,
#      models/append_to_nerf_pipeline.py:14-90)
# ----------------------------------------------------------------------------------------------
def append_pose_pipeline_forward(params_coarse, params_fine, args, position_encoder, direction_encoder,
                                 human_pose_encoder, data, two_joints=False, net_kw=None):
    """Rows are [pose | PE(x) | PE(d)] (:49-51) fed to RenderRayNet(additional_input_dim = pose columns)."""
    net_kw = dict(net_kw or {})
    ray_samples, ray_translation, ray_direction, z_vals, goal_pose = [np.asarray(t, F32) for t in data[:5]]
    if two_joints:
        goal_pose = np.stack([goal_pose[:, 38], goal_pose[:, 41]], -1)                   # append_to_nerf :26
    pose = human_pose_encoder.encode(goal_pose) if args.human_pose_encoding else goal_pose   # :29-37
    net_kw["additional_input_dim"] = pose.shape[-1]
    B, Nc = ray_samples.shape[:2]
    dirs = np.broadcast_to(ray_direction[:, None, :], (B, Nc, 3))
    directions_encoding = direction_encoder.encode(_normalize(dirs))

    def rows(pts):
        n = pts.shape[1]
        pz = np.broadcast_to(pose[:, None, :], (B, n, pose.shape[-1]))
        denc = np.broadcast_to(directions_encoding[:, :1, :], (B, n, directions_encoding.shape[-1]))
        return np.concatenate([pz.reshape(B * n, -1), position_encoder.encode(pts).reshape(B * n, -1),
                               denc.reshape(B * n, -1)], -1)

    raw = render_ray_net_forward(params_coarse, rows(ray_samples), **net_kw).reshape(B, Nc, 4)
    rgb, weights, densities = raw2outputs(raw, z_vals, dirs, args.white_background)      # :55
    if not args.run_fine:
        return rgb, rgb, ray_samples, densities
    z_f, pts_f = fine_sampling(ray_translation, ray_direction, z_vals, weights, args.number_fine_samples,
                               getattr(args, "u", None))                                 # :60
    N = pts_f.shape[1]
    raw_f = render_ray_net_forward(params_fine, rows(pts_f), **net_kw).reshape(B, N, 4)
    dirs_f = np.broadcast_to(ray_direction[:, None, :], (B, N, 3))
    rgb_fine, _, dens_f = raw2outputs(raw_f, z_f, dirs_f, args.white_background)         # :89
    return rgb, rgb_fine, pts_f, dens_f


# ----------------------------------------------------------------------------------------------
# a8  AppendVerticesPipeline.forward  (models/append_vertices_pipeline.py:16-94)
# ----------------------------------------------------------------------------------------------
def append_vertices_net_forward(params, x, n_layers=8, positions_dim=60, directions_dim=24, skips=(4,)):
    """models/append_vertices_net.py:43-66.  positions = x[:, :positions_dim] (:44), directions =
    x[:, -directions_dim:] (:47); `vertices_net` is evaluated by the reference and its result dropped
    (:48-50), so it is not restated here (it cannot influence the output)."""
    return render_ray_net_forward(params, x, n_layers=n_layers, positions_dim=positions_dim,
                                  directions_dim=directions_dim, additional_input_dim=0, skips=skips)


def append_vertices_pipeline_forward(params_coarse, params_fine, vertices, args, position_encoder,
                                     direction_encoder, data, net_kw=None):
    """models/append_vertices_pipeline.py:16-94 given the posed vertices [B, 6890, 3] the body model returned
    (:38-40).  Rows are [vertices_flat | PE(x) | PE(d)] (:56-58); only the columns the net reads are built."""
    net_kw = net_kw or {}
    ray_samples, ray_translation, ray_direction, z_vals = [np.asarray(t, F32) for t in data[:4]]
    B, Nc = ray_samples.shape[:2]
    vflat = np.asarray(vertices, F32).reshape(B, -1)                                     # :41
    pdim = net_kw.get("positions_dim", 60)
    dirs = np.broadcast_to(ray_direction[:, None, :], (B, Nc, 3))
    directions_encoding = direction_encoder.encode(_normalize(dirs))                     # :52-55

    def rows(n):
        head = np.broadcast_to(vflat[:, None, :pdim], (B, n, pdim))                      # first columns of the row
        denc = np.broadcast_to(directions_encoding[:, :1, :], (B, n, directions_encoding.shape[-1]))
        return np.concatenate([head.reshape(B * n, -1), denc.reshape(B * n, -1)], -1)

    raw = append_vertices_net_forward(params_coarse, rows(Nc), **net_kw).reshape(B, Nc, 4)     # :59-62
    rgb, weights, densities = raw2outputs(raw, z_vals, dirs, args.white_background)      # :63
    if not args.run_fine:
        return rgb, rgb, ray_samples, densities
    z_f, pts_f = fine_sampling(ray_translation, ray_direction, z_vals, weights, args.number_fine_samples,
                               getattr(args, "u", None))                                 # :68
    N = pts_f.shape[1]
    raw_f = append_vertices_net_forward(params_fine, rows(N), **net_kw).reshape(B, N, 4)  # :84-88
    dirs_f = np.broadcast_to(ray_direction[:, None, :], (B, N, 3))
    rgb_fine, _, dens_f = raw2outputs(raw_f, z_f, dirs_f, args.white_background)         # :92
    return rgb, rgb_fine, pts_f, dens_f


# ----------------------------------------------------------------------------------------------
# adjacent: rays + stratified coarse samples (utils.py:26-54, datasets/transforms.py:58-90)
# ----------------------------------------------------------------------------------------------
def get_rays(H, W, focal, camera_transform):
    """utils.py:50-54 (Q9): integer pixel centres, dirs un-normalised, float64 result."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    dirs = np.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -np.ones_like(i)], -1)
    rays_direction = np.sum(dirs[..., np.newaxis, :] * camera_transform[:3, :3], -1)
    rays_translation = np.broadcast_to(camera_transform[:3, -1], np.shape(rays_direction))
    return rays_translation, rays_direction


def coarse_sampling(ray_translation, ray_direction, near, far, number_samples, jitter):
    """datasets/transforms.py:80-89 (Q8) for a batch of rays: bins linear in disparity, ONE
    jitter scalar per ray (np.random.rand() in the reference), float64 math, cast to fp32 by
    ToTensor (datasets/transforms.py:13-21)."""
    t_vals = np.linspace(0., 1., number_samples)
    z_vals = 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
    mids = .5 * (z_vals[1:] + z_vals[:-1])
    upper = np.concatenate([mids, z_vals[-1:]], -1)
    lower = np.concatenate([z_vals[:1], mids], -1)
    jitter = np.asarray(jitter, np.float64).reshape(-1, 1)
    z = lower[None, :] + (upper - lower)[None, :] * jitter
    o = np.asarray(ray_translation, np.float64)
    d = np.asarray(ray_direction, np.float64)
    samples = o[:, None, :] + d[:, None, :] * z[:, :, None]
    return samples.astype(F32), o.astype(F32), d.astype(F32), z.astype(F32)


def mse2psnr_img(mse):
    """util/scores.py:47-48: -10*ln(mse)/ln(10)."""
    return float(-10.0 * np.log(mse) / np.log(10.0))
