"""TEST / MEASUREMENT INFRASTRUCTURE - not part of the product path.

fp32 PyTorch-CPU restatement of the reference's forward hot path, op for op (same ATen calls in the same
order, so that it is also a fair *timing* stand-in for the reference's CPU path, which cannot travel to the GPU
box).  Used by bench.py's `cpu_baseline` leg and by tests; pinned two ways in the build container:

  * values: tests/test_oracle_golden.py::test_torch_cpu_path_* against the fixtures the reference produced
    (tests/golden/g5_nerf_pipeline.npz, g6_smpl_nerf_pipeline.npz);
  * speed:  oracle/calibrate_cpu_baseline.py imports the reference itself and times
    NerfPipeline.forward beside nerf_pipeline_forward() below (must agree within +-10 %); its
    result is committed as oracle/cpu_baseline_calibration.json.

Reference lines followed:
    PositionalEncoder            utils.py:114-131
    RenderRayNet.forward         models/render_ray_net.py:42-61
    raw2outputs                  utils.py:134-191
    sample_pdf / fine_sampling   utils.py:194-264   (torchsearchsorted bound to torch.searchsorted: the native
                                                    extension does not build against this torch)
    NerfPipeline.forward         models/nerf_pipeline.py:14-67
    WarpFieldNet.forward         models/warp_field_net.py:17-22
    SmplNerfPipeline.forward     models/smpl_nerf_pipeline.py:16-100
    NerfSolver's training step   solver/nerf_solver.py:31-33 (Adam), :48-52 (loss), :76-87 (per-batch body)
    SmplNerfSolver's step        solver/smpl_nerf_solver.py:26-28 (one Adam over the three nets), :35-43, :75-86

The training step (TrainState / train_step) is the same forward under autograd followed by the reference's own sequence
zero_grad -> loss -> backward -> Adam.step -> loss.item(); oracle/calibrate_cpu_baseline.py times it beside the imported
reference's NerfSolver objects (losses bit-identical, speed within +-10 %).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


class Args:
    def __init__(self, **kw):
        self.sigma_noise_std = 0.0
        self.white_background = 0
        self.run_fine = 1
        self.number_fine_samples = 128
        self.human_pose_encoding = 1
        self.__dict__.update(kw)


class PositionalEncoder:
    """utils.py:114-131 (frequency-major cat of sin/cos, optional identity first)."""

    def __init__(self, number_frequencies, include_identity):
        self.freq_bands = torch.pow(2, torch.linspace(0., number_frequencies - 1, number_frequencies))
        self.number_frequencies = number_frequencies
        self.include_identity = include_identity
        self.output_dim = (1 if include_identity else 0) + 2 * number_frequencies

    def encode(self, x):
        outs = [x] if self.include_identity else []
        for freq in self.freq_bands:
            outs.append(torch.sin(x * freq))
            outs.append(torch.cos(x * freq))
        return torch.cat(outs, -1)


def tparams(params):
    """{state_dict key: numpy array} -> {key: fp32 CPU tensor}."""
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in params.items()}


def render_ray_net(P, x, n_layers=8, positions_dim=60, directions_dim=24, additional_input_dim=0, skips=(4,),
                   use_directional_input=1):
    """models/render_ray_net.py:42-61 on a parameter dict."""
    lin = lambda v, n: F.linear(v, P[n + ".weight"], P[n + ".bias"])
    positions_pose, directions = x[..., :positions_dim + additional_input_dim], x[..., -directions_dim:]
    o = F.relu(lin(positions_pose, "positions_pose_input"))
    for i in range(n_layers - 1):
        if i in skips:
            o = F.relu(lin(torch.cat([o, positions_pose], -1), f"positional_net.{i}"))
        else:
            o = F.relu(lin(o, f"positional_net.{i}"))
    o = lin(o, "additional_linear_layer")
    sigma = lin(o, "sigma_out_layer")
    o = lin(torch.cat([o, directions], -1) if use_directional_input else o, "directional_input")
    o = F.relu(lin(o, "directional_net.0"))
    return torch.cat([lin(o, "rgb_out_layer"), sigma], -1)


def warp_field_net(P, x):
    """models/warp_field_net.py:17-22."""
    h = F.relu(F.linear(x, P["linear1.weight"], P["linear1.bias"]))
    return F.linear(h, P["linear2.weight"], P["linear2.bias"])


def raw2outputs(raw, z_vals, samples_directions, args):
    """utils.py:134-191."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.tensor([1e10]).expand(dists[..., :1].shape)], -1)
    dists = dists * torch.norm(samples_directions, dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    if z_vals.shape[-1] == 1:
        return rgb.view(raw.shape[0], 3), torch.ones(raw.shape[0], 1), torch.ones(raw.shape[0], 1)
    noise = 0.
    if args.sigma_noise_std > 0.:
        noise = torch.normal(0, args.sigma_noise_std, raw[..., 3].shape)
    density = 1. - torch.exp(-F.relu(raw[..., 3] + noise) * dists)
    one_minus_density = 1. - density + 1e-10
    ones = torch.ones(one_minus_density.shape[:-1]).unsqueeze(-1)
    exclusive = torch.cat([ones, one_minus_density[..., :-1]], -1)
    weights = density * torch.cumprod(exclusive, -1)
    rgb = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)  # noqa: F841  (computed and dropped by the reference too, :183)
    acc_map = torch.sum(weights, -1)
    if args.white_background:
        rgb = rgb + (1. - acc_map[..., None])
    return rgb, weights, density


LAST = {}      # the searchsorted indices of the most recent sample_pdf call (a reference to the tensor, no copy)


def sample_pdf(bins, weights, args):
    """utils.py:194-228."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = torch.linspace(0., 1., steps=args.number_fine_samples)
    u = u.expand(list(cdf.shape[:-1]) + [args.number_fine_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    LAST["inds"] = inds          # (checker's tap: bench.py compares the HIP sampler's indices with these)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min(cdf.shape[-1] - 1 * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    matched_shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(matched_shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(matched_shape), 2, inds_g)
    denom = (cdf_g[..., 1] - cdf_g[..., 0])
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


FINE_OVERRIDE = None      # checker's tap: (z_vals_fine [B, Nc + Nf], ray_samples_fine [B, Nc + Nf, 3]) to use instead of this path's own
                          # hierarchical samples - the HIP path's, so that the fine pass is compared on equal samples (tools/ab/fuzz_*).
                          # Set it through fine_override() only: a leftover value would make every later run of this path reuse
                          # those samples and mask a sampler divergence; tests/conftest.py asserts it is None.


class fine_override:
    """with fine_override((z_fine, pts)): ...   - the tap above for the duration of the block, always reset."""

    def __init__(self, value):
        self.value = value

    def __enter__(self):
        global FINE_OVERRIDE
        assert FINE_OVERRIDE is None, "fine_override does not nest"
        FINE_OVERRIDE = self.value
        return self

    def __exit__(self, *exc):
        global FINE_OVERRIDE
        FINE_OVERRIDE = None
        return False


def fine_sampling(ray_translation, samples_directions, z_vals, weights, args):
    """utils.py:231-264."""
    if FINE_OVERRIDE is not None:
        return FINE_OVERRIDE
    z_vals_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    z_samples = sample_pdf(z_vals_mid, weights[..., 1:-1], args).detach()
    z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
    pts = ray_translation[..., None, :] + samples_directions[..., None, :] * z_vals[..., :, None]
    return z_vals, pts


def nerf_pipeline_forward(Pc, Pf, args, position_encoder, direction_encoder, data, net_kw=None):
    """models/nerf_pipeline.py:14-67; data = [ray_samples, ray_translation, ray_direction, z_vals, rgb_truth]."""
    net_kw = net_kw or {}
    ray_samples, ray_translation, ray_direction, z_vals, _ = data
    samples_encoding = position_encoder.encode(ray_samples)
    csd = ray_direction[..., None, :].expand(ray_direction.shape[0], ray_samples.shape[1], ray_direction.shape[-1])
    sdn = csd / torch.norm(csd, dim=-1, keepdim=True)
    directions_encoding = direction_encoder.encode(sdn)
    inputs = torch.cat([samples_encoding.view(-1, samples_encoding.shape[-1]),
                        directions_encoding.view(-1, directions_encoding.shape[-1])], -1)
    raw = render_ray_net(Pc, inputs, **net_kw)
    raw = raw.view(samples_encoding.shape[0], samples_encoding.shape[1], raw.shape[-1])
    rgb, weights, densities = raw2outputs(raw, z_vals, csd, args)
    if not args.run_fine:
        return rgb, rgb, ray_samples, densities
    z_vals, ray_samples_fine = fine_sampling(ray_translation, ray_direction, z_vals, weights, args)
    sef = position_encoder.encode(ray_samples_fine)
    def_ = directions_encoding[..., :1, :].expand(directions_encoding.shape[0], ray_samples_fine.shape[1],
                                                  directions_encoding.shape[-1])
    inputs_fine = torch.cat([sef.view(-1, sef.shape[-1]), def_.reshape(-1, def_.shape[-1])], -1)
    raw_f = render_ray_net(Pf, inputs_fine, **net_kw)
    raw_f = raw_f.reshape(sef.shape[0], sef.shape[1], raw_f.shape[-1])
    fsd = ray_direction[..., None, :].expand(ray_direction.shape[0], ray_samples_fine.shape[1], ray_direction.shape[-1])
    rgb_fine, _, densities = raw2outputs(raw_f, z_vals, fsd, args)
    return rgb, rgb_fine, ray_samples_fine, densities


def smpl_nerf_pipeline_forward(Pc, Pf, Pw, args, position_encoder, direction_encoder, human_pose_encoder, data, net_kw=None):
    """models/smpl_nerf_pipeline.py:16-100 (human_pose_encoding = 1);
    data = [ray_samples, ray_translation, ray_direction, z_vals, goal_pose, rgb_truth]."""
    net_kw = net_kw or {}
    ray_samples, ray_translation, ray_direction, z_vals, goal_pose, _ = data
    goal_pose = torch.stack([goal_pose[:, 38], goal_pose[:, 41]], axis=-1)
    pose_enc = human_pose_encoder.encode(goal_pose)

    def stage(P, samples):
        B, N = samples.shape[:2]
        enc = position_encoder.encode(samples)
        pe = pose_enc[:, None, :].expand(B, N, pose_enc.shape[-1])
        warp = warp_field_net(Pw, torch.cat([enc, pe], -1).view(B * N, -1)).view(B, N, 3)
        warped = samples + warp
        enc_w = position_encoder.encode(warped)
        sdirs = warped - ray_translation[:, None, :]
        sdn = sdirs / torch.norm(sdirs, dim=-1, keepdim=True)
        denc = direction_encoder.encode(sdn)
        inputs = torch.cat([enc_w.view(B * N, -1), denc.view(B * N, -1)], -1)
        return warp, warped, sdirs, render_ray_net(P, inputs, **net_kw).view(B, N, 4)

    warp, warped, sdirs, raw = stage(Pc, ray_samples)
    rgb, weights, densities = raw2outputs(raw, z_vals, sdirs, args)
    if not args.run_fine:
        return rgb, rgb, warp, ray_samples, warped, densities
    z_fine, ray_samples_fine = fine_sampling(ray_translation, ray_direction, z_vals, weights, args)
    warp_f, warped_f, _, raw_f = stage(Pf, ray_samples_fine)
    fsd = ray_direction[..., None, :].expand(ray_direction.shape[0], ray_samples_fine.shape[1], ray_direction.shape[-1])
    rgb_fine, _, densities_fine = raw2outputs(raw_f, z_fine, fsd, args)
    return rgb, rgb_fine, warp_f, ray_samples_fine, warped_f, densities_fine


def append_pose_pipeline_forward(Pc, Pf, args, position_encoder, direction_encoder, human_pose_encoder, data, two_joints=False):
    """models/append_smpl_params_pipeline.py:14-91 (two_joints: append_to_nerf_pipeline.py:14-90): the pose row of a ray is
    expanded over its samples and concatenated IN FRONT of the encodings, rows [pose | PE(x) | PE(d)] (:49-51);
    data = [ray_samples, ray_translation, ray_direction, z_vals, goal_pose, rgb_truth]."""
    ray_samples, ray_translation, ray_direction, z_vals, goal_pose, _ = data
    if two_joints:
        goal_pose = torch.stack([goal_pose[:, 38], goal_pose[:, 41]], axis=-1)
    pose = human_pose_encoder.encode(goal_pose) if args.human_pose_encoding else goal_pose          # :29-37
    kw = dict(additional_input_dim=pose.shape[-1])

    def rows(samples, denc):
        B, N = samples.shape[:2]
        pz = pose[..., None, :].expand(B, N, pose.shape[-1])
        enc = position_encoder.encode(samples)
        return torch.cat([pz.reshape(-1, pz.shape[-1]), enc.view(-1, enc.shape[-1]), denc.reshape(-1, denc.shape[-1])], -1)

    csd = ray_direction[..., None, :].expand(ray_direction.shape[0], ray_samples.shape[1], ray_direction.shape[-1])
    directions_encoding = direction_encoder.encode(csd / torch.norm(csd, dim=-1, keepdim=True))
    raw = render_ray_net(Pc, rows(ray_samples, directions_encoding), **kw).view(ray_samples.shape[0], ray_samples.shape[1], 4)
    rgb, weights, densities = raw2outputs(raw, z_vals, csd, args)
    if not args.run_fine:
        return rgb, rgb, ray_samples, densities
    z_vals, ray_samples_fine = fine_sampling(ray_translation, ray_direction, z_vals, weights, args)
    def_ = directions_encoding[..., :1, :].expand(directions_encoding.shape[0], ray_samples_fine.shape[1], directions_encoding.shape[-1])
    raw_f = render_ray_net(Pf, rows(ray_samples_fine, def_), **kw).reshape(ray_samples_fine.shape[0], ray_samples_fine.shape[1], 4)
    fsd = ray_direction[..., None, :].expand(ray_direction.shape[0], ray_samples_fine.shape[1], ray_direction.shape[-1])
    rgb_fine, _, densities = raw2outputs(raw_f, z_vals, fsd, args)
    return rgb, rgb_fine, ray_samples_fine, densities


def append_vertices_net(P, x, n_layers=8, positions_dim=60, directions_dim=24, additional_input_dim=20670, skips=(4,)):
    """models/append_vertices_net.py:43-66 as the reference EXECUTES it: `vertices_net` runs on the vertex columns of every
    row and its result is dropped (:48-50) - part of what the reference's CPU pays, so the timing port pays it too."""
    lin = lambda v, n: torch.nn.functional.linear(v, P[n + ".weight"], P[n + ".bias"])
    positions = x[..., :positions_dim]
    verts = x[..., positions_dim:positions_dim + additional_input_dim]
    i = 0
    while f"vertices_net.{i}.weight" in P:
        verts = torch.relu(lin(verts, f"vertices_net.{i}"))
        i += 1
    directions = x[..., x.shape[-1] - directions_dim:]
    o = torch.relu(lin(positions, "positions_pose_input"))
    for i in range(n_layers - 1):
        o = torch.relu(lin(torch.cat([o, positions], -1) if i in skips else o, f"positional_net.{i}"))
    o = lin(o, "additional_linear_layer")
    sigma = lin(o, "sigma_out_layer")
    o = torch.relu(lin(lin(torch.cat([o, directions], -1), "directional_input"), "directional_net.0"))
    return torch.cat([lin(o, "rgb_out_layer"), sigma], -1)


def append_vertices_pipeline_forward_coarse(Pc, vertices, args, position_encoder, direction_encoder, data):
    """models/append_vertices_pipeline.py:16-63 up to the coarse result - the part of it the reference can run (its fine
    branch raises at :71) - given the posed vertices [B, 6890, 3] of the body model (:38-40).  The rows are materialised
    like the reference's: [vertices_flat (20 670) | PE(x) | PE(d)] per SAMPLE (:56-58), 83 KB each."""
    ray_samples, ray_translation, ray_direction, z_vals = data[:4]
    B, N = ray_samples.shape[:2]
    vflat = vertices.reshape(B, -1)
    vrows = vflat[..., None, :].expand(B, N, vflat.shape[-1])
    enc = position_encoder.encode(ray_samples)
    csd = ray_direction[..., None, :].expand(B, N, 3)
    denc = direction_encoder.encode(csd / torch.norm(csd, dim=-1, keepdim=True))
    inputs = torch.cat([vrows.reshape(-1, vflat.shape[-1]), enc.view(-1, enc.shape[-1]), denc.view(-1, denc.shape[-1])], -1)
    # (the net is built with additional_input_dim = 6890, train.py:206-211: vertices_net reads row columns 60 .. 6950)
    raw = append_vertices_net(Pc, inputs, additional_input_dim=Pc["vertices_net.0.weight"].shape[1]).view(B, N, 4)
    rgb, weights, densities = raw2outputs(raw, z_vals, csd, args)
    return rgb, rgb, ray_samples, densities


# ----------------------------------------------------------------------------------------------------------------
# training step (solver/nerf_solver.py:76-87, solver/smpl_nerf_solver.py:75-86)
# ----------------------------------------------------------------------------------------------------------------
class TrainState:
    """Leaf parameter dicts of the nets + the reference's optimiser: torch.optim.Adam over coarse, fine (, warp field)
    parameters in registration order with NerfSolver.default_adam_args overridden by lr / weight_decay
    (solver/nerf_solver.py:10-14, 31-33; smpl: solver/smpl_nerf_solver.py:26-28)."""

    def __init__(self, params_list, lr=5e-4, weight_decay=0.0, workload="nerf", args=None):
        self.P = [{k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).clone().requires_grad_(True)
                   for k, v in p.items()} for p in params_list]
        self.workload = workload
        self.args = args or Args()
        self.optim = torch.optim.Adam([t for P in self.P for t in P.values()], lr=lr, betas=(0.9, 0.999), eps=1e-8,
                                      weight_decay=weight_decay)
        self.loss_func = torch.nn.MSELoss()
        self.pe, self.de, self.he = PositionalEncoder(10, False), PositionalEncoder(4, False), PositionalEncoder(10, False)

    def forward(self, data):
        if self.workload == "smpl_nerf":
            return smpl_nerf_pipeline_forward(self.P[0], self.P[1], self.P[2], self.args, self.pe, self.de, self.he, data)
        return nerf_pipeline_forward(self.P[0], self.P[1], self.args, self.pe, self.de, data)


def train_step(state: TrainState, data) -> float:
    """One batch of NerfSolver.train (solver/nerf_solver.py:76-87), statement for statement."""
    rgb_truth = data[-1]
    out = state.forward(data)                                                    # :81
    rgb, rgb_fine = out[0], out[1]
    state.optim.zero_grad()                                                      # :83
    loss = state.loss_func(rgb, rgb_truth) + state.loss_func(rgb_fine, rgb_truth)   # :85, :48-52
    loss.backward()                                                              # :86
    state.optim.step()                                                           # :87
    return loss.item()                                                           # :89
