"""Deterministic synthetic inputs for the ray-march path (host side, numpy only).

The reference's datasets need SMPL assets and rendered PNGs that are not shipped, so tests, the
golden-vector generator and bench.py all draw their cameras, rays, coarse samples and network
weights from here.  Formulas follow the reference:

* camera pose      camera.py:86-110 (get_sphere_pose) / camera.py:33-37 (get_pose_matrix)
* focal length     datasets/rays_from_images_dataset.py:44  (.5*w/tan(.5*camera_angle_x)),
                   camera_angle_x = pi/3 (create_dataset.py:141)
* rays             utils.py:50-54 (get_rays; integer pixel centres, un-normalised directions, fp64)
* coarse samples   datasets/transforms.py:80-89 (bins linear in disparity, one jitter per ray, fp64
                   math, cast to fp32 by ToTensor :13-21)
* weights          torch.nn.Linear default init bounds (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for
                   weight and bias) drawn from numpy's PCG64 so that fixtures need not store them.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def pose_matrix(x=0.0, y=0.0, z=0.0, phi=0.0, theta=0.0, psi=0.0) -> np.ndarray:
    """Extrinsic 'xyz' Euler rotation in degrees + translation (camera.py:33-37)."""
    a, b, c = np.radians([phi, theta, psi])
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    pose = np.eye(4)
    pose[:3, :3] = rz @ ry @ rx          # scipy 'xyz' (lower case) = extrinsic x, then y, then z
    pose[:3, 3] = [x, y, z]
    return pose


def sphere_pose(phi: float, theta: float, r: float) -> np.ndarray:
    """camera.py:106-110."""
    z = r * np.cos(np.radians(phi)) * np.cos(np.radians(theta))
    x = r * np.cos(np.radians(phi)) * np.sin(np.radians(theta))
    y = r * np.sin(np.radians(phi))
    return pose_matrix(x=x, y=y, z=z, theta=theta, phi=-phi)


def focal_length(w: int, camera_angle_x: float = np.pi / 3) -> float:
    return .5 * w / np.tan(.5 * camera_angle_x)


def camera_rays(h: int, w: int, pose: np.ndarray, camera_angle_x: float = np.pi / 3):
    """utils.py:50-54 -> (translation[h*w,3], direction[h*w,3]) float64, row-major pixels."""
    focal = focal_length(w, camera_angle_x)
    i, j = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="xy")
    dirs = np.stack([(i - w * .5) / focal, -(j - h * .5) / focal, -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * pose[:3, :3], -1)
    rays_o = np.broadcast_to(pose[:3, -1], rays_d.shape)
    return rays_o.reshape(-1, 3).astype(np.float64), rays_d.reshape(-1, 3).astype(np.float64)


def coarse_samples(rays_o, rays_d, near: float, far: float, n: int, jitter):
    """datasets/transforms.py:80-89 batched; `jitter` is the per-ray np.random.rand() scalar."""
    t = np.linspace(0., 1., n)
    z = 1. / (1. / near * (1. - t) + 1. / far * t)
    mids = .5 * (z[1:] + z[:-1])
    upper = np.concatenate([mids, z[-1:]], -1)
    lower = np.concatenate([z[:1], mids], -1)
    jitter = np.broadcast_to(np.asarray(jitter, np.float64).reshape(-1, 1), (rays_o.shape[0], 1))
    zs = lower[None, :] + (upper - lower)[None, :] * jitter
    pts = rays_o[:, None, :] + rays_d[:, None, :] * zs[:, :, None]
    return pts.astype(F32), rays_o.astype(F32), rays_d.astype(F32), zs.astype(F32)


def frame_batch(h=128, w=128, phi=0.0, theta=0.0, radius=2.4, near=1.0, far=4.0, n_coarse=64,
                jitter=0.5, seed=None):
    """One frame worth of pipeline inputs [ray_samples, ray_translation, ray_direction, z_vals,
    rgb_truth] as fp32 numpy arrays (the list layout of solver/nerf_solver.py:77-81)."""
    o, d = camera_rays(h, w, sphere_pose(phi, theta, radius))
    if seed is not None:
        jitter = np.random.default_rng(seed).random(o.shape[0])
    pts, o32, d32, z = coarse_samples(o, d, near, far, n_coarse, jitter)
    rgb = procedural_image(h, w, phi, theta).reshape(-1, 3)
    return [pts, o32, d32, z, rgb]


def procedural_image(h: int, w: int, phi: float = 0.0, theta: float = 0.0) -> np.ndarray:
    """A fixed analytic BGR image in [0,1] (values only feed the loss)."""
    yy, xx = np.meshgrid(np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    r2 = (xx - 0.2 * np.sin(np.radians(theta))) ** 2 + (yy + 0.1 * np.sin(np.radians(phi))) ** 2
    img = np.stack([np.exp(-4 * r2), 0.5 + 0.5 * np.cos(6 * xx) * np.exp(-2 * r2), np.clip(1 - r2, 0, 1)], -1)
    return img.astype(F32)


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
def render_ray_net_shapes(n_layers=8, width=256, positions_dim=60, directions_dim=24,
                          additional_input_dim=0, skips=(4,), use_directional_input=1):
    """state_dict (name, shape) pairs in registration order (models/render_ray_net.py:19-40)."""
    pin = positions_dim + additional_input_dim
    layers = [("positions_pose_input", width, pin)]
    for i in range(n_layers - 1):
        layers.append((f"positional_net.{i}", width, width + pin if i in skips else width))
    layers.append(("additional_linear_layer", width, width))
    layers.append(("sigma_out_layer", 1, width))
    dw = width // 2
    layers.append(("directional_input", dw, width + directions_dim if use_directional_input else width))
    layers.append(("directional_net.0", dw, dw))
    layers.append(("rgb_out_layer", 3, dw))
    return layers


def init_linear_stack(layers, seed: int) -> dict:
    rng = np.random.default_rng(seed)
    params = {}
    for name, fan_out, fan_in in layers:
        bound = 1.0 / np.sqrt(fan_in)
        params[name + ".weight"] = rng.uniform(-bound, bound, (fan_out, fan_in)).astype(F32)
        params[name + ".bias"] = rng.uniform(-bound, bound, (fan_out,)).astype(F32)
    return params


def make_render_ray_net_params(seed: int, sigma_scale: float = 1.0, rgb_scale: float = 1.0, **net_kw) -> dict:
    """Seeded RenderRayNet parameters.  `sigma_scale`/`rgb_scale` multiply the two head weight
    matrices so that a random-init net produces non-trivial densities and colours."""
    params = init_linear_stack(render_ray_net_shapes(**net_kw), seed)
    params["sigma_out_layer.weight"] = (params["sigma_out_layer.weight"] * F32(sigma_scale)).astype(F32)
    params["rgb_out_layer.weight"] = (params["rgb_out_layer.weight"] * F32(rgb_scale)).astype(F32)
    return params


def _probe_trunk(params, pts, dirs, n_layers=8, skips=(4,), pos_L=10, dir_L=4, add=None, add_first=False):
    """Plain numpy evaluation of a RenderRayNet up to the two head inputs (used only to calibrate
    the synthetic scene below; fp64, not a reference for anything)."""
    def enc(x, L):
        out = []
        for k in range(L):
            out += [np.sin(x * 2.0 ** k), np.cos(x * 2.0 ** k)]
        return np.concatenate(out, -1)
    P = {k: v.astype(np.float64) for k, v in params.items()}
    pe, de = enc(pts.astype(np.float64), pos_L), enc(dirs.astype(np.float64), dir_L)
    if add is not None:
        pe = np.concatenate([add.astype(np.float64), pe] if add_first else [pe, add.astype(np.float64)], -1)
    lin = lambda x, n: x @ P[n + ".weight"].T + P[n + ".bias"]
    o = np.maximum(lin(pe, "positions_pose_input"), 0)
    for i in range(n_layers - 1):
        o = np.maximum(lin(np.concatenate([o, pe], -1) if i in skips else o, f"positional_net.{i}"), 0)
    o = lin(o, "additional_linear_layer")
    h = lin(np.concatenate([o, de], -1), "directional_input")
    h = np.maximum(lin(h, "directional_net.0"), 0)
    return o, h


def make_scene_net_params(seed: int, sigma_std: float = 10.0, rgb_std: float = 1.5, gamma: float = 0.7,
                          add_first: bool = False, **net_kw) -> dict:
    """A random-init RenderRayNet turned into a well-conditioned synthetic scene:

    * the weight columns that read positional-encoding band k are damped by 2^(-gamma*k), so the
      field is spatially smooth at the sample spacing (an undamped random init is white noise in
      space: there even the reference's own fp32 and fp64 renderings differ by 6e-2, while a trained
      NeRF - and this scene - sit at the 1e-6 fp32 round-off floor);
    * the two heads are rescaled and re-centred so that over the camera frustum sigma ~ (0, sigma_std)
      and the colour logits ~ (0, rgb_std): a semi-transparent volume with non-trivial compositing
      weights (an untouched random init predicts a constant, usually negative sigma: an empty scene).
    """
    layers = render_ray_net_shapes(**net_kw)
    params = init_linear_stack(layers, seed)
    pos_dim = net_kw.get("positions_dim", 60)
    width = net_kw.get("width", 256)
    L = pos_dim // 6
    scale = np.repeat(2.0 ** (-gamma * np.arange(L)), 6).astype(F32)
    add_dim = net_kw.get("additional_input_dim", 0)
    c0 = add_dim if add_first else 0            # [add | PE(x)] rows (append_* pipelines) vs [PE(x) | add]
    params["positions_pose_input.weight"][:, c0:c0 + pos_dim] *= scale
    for i in net_kw.get("skips", (4,)):
        params[f"positional_net.{i}.weight"][:, width + c0:width + c0 + pos_dim] *= scale
    rng = np.random.default_rng(seed + 7919)
    pts = rng.uniform(-2.5, 2.5, (2048, 3))
    dirs = rng.normal(size=(2048, 3))
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    add = rng.uniform(-1, 1, (2048, add_dim)) if add_dim else None
    o, h = _probe_trunk(params, pts, dirs, n_layers=net_kw.get("n_layers", 8), skips=net_kw.get("skips", (4,)), add=add,
                        add_first=add_first)
    w = params["sigma_out_layer.weight"].astype(np.float64)
    s = o @ w.T
    sc = sigma_std / s.std()
    params["sigma_out_layer.weight"] = (w * sc).astype(F32)
    params["sigma_out_layer.bias"] = np.asarray(-s.mean(0) * sc, F32)
    w = params["rgb_out_layer.weight"].astype(np.float64)
    c = h @ w.T
    sc = rgb_std / c.std(0)
    params["rgb_out_layer.weight"] = (w * sc[:, None]).astype(F32)
    params["rgb_out_layer.bias"] = np.asarray(-c.mean(0) * sc, F32)
    return params


def make_scene_nets(seed: int, eps: float = 0.01, **kw):
    """(coarse, fine) parameters of ONE scene: the fine net is the coarse net with a 1 % relative
    perturbation of every weight, the way a trained coarse/fine pair describes the same volume (with
    two unrelated random fields the hierarchical samples would land in arbitrary fine densities)."""
    coarse = make_scene_net_params(seed, **kw)
    rng = np.random.default_rng(seed + 104729)
    fine = {k: (v * (1.0 + eps * rng.standard_normal(v.shape))).astype(F32) for k, v in coarse.items()}
    return coarse, fine


def make_warp_field_params(seed: int, positions_dim=60, pose_dim=40, width=256, out_scale=1.0) -> dict:
    """WarpFieldNet: linear1 (width, positions_dim+pose_dim), linear2 (3, width) (models/warp_field_net.py:14-15)."""
    params = init_linear_stack([("linear1", width, positions_dim + pose_dim), ("linear2", 3, width)], seed)
    params["linear2.weight"] = (params["linear2.weight"] * F32(out_scale)).astype(F32)
    return params


def append_vertices_shapes(n_layers=8, width=256, positions_dim=60, directions_dim=24, additional_input_dim=6890,
                           additional_input_layers=1, skips=(4,)):
    """AppendVerticesNet state_dict (name, fan_out, fan_in) in registration order
    (models/append_vertices_net.py:19-41)."""
    layers = [("positions_pose_input", width, positions_dim)]
    for i in range(n_layers - 1):
        layers.append((f"positional_net.{i}", width, width + positions_dim if i in skips else width))
    layers.append(("additional_linear_layer", width, width))
    layers.append(("sigma_out_layer", 1, width))
    layers.append(("vertices_net.0", width, additional_input_dim))
    for i in range(additional_input_layers):
        layers.append((f"vertices_net.{i + 1}", width, width))
    dw = width // 2
    layers.append(("directional_input", dw, width + directions_dim))
    layers.append(("directional_net.0", dw, dw))
    layers.append(("rgb_out_layer", 3, dw))
    return layers


def make_append_vertices_params(seed: int, sigma_scale=30.0, rgb_scale=10.0, **kw) -> dict:
    params = init_linear_stack(append_vertices_shapes(**kw), seed)
    params["sigma_out_layer.weight"] = (params["sigma_out_layer.weight"] * F32(sigma_scale)).astype(F32)
    params["sigma_out_layer.bias"] = (params["sigma_out_layer.bias"] + F32(2.0)).astype(F32)
    params["rgb_out_layer.weight"] = (params["rgb_out_layer.weight"] * F32(rgb_scale)).astype(F32)
    return params


def human_poses(joints=(41, 38), start=0.0, end=60.0, steps=10) -> np.ndarray:
    """render.py:190-220 (get_human_poses) -> [steps, 69] fp32, radians at the listed joints."""
    angles = np.linspace(start, end, steps)
    poses = np.zeros((steps, 69), F32)
    for i, a in enumerate(angles):
        for j in joints:
            poses[i, j] = np.deg2rad(a)
    return poses
