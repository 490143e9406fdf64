#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box never sees it).  The
reference's Python is imported from where it lies with stand-ins for the third-party modules it
imports at module scope but that the hot path never calls (cv2, imageio, trimesh, tensorboard),
and `torchsearchsorted.searchsorted` bound to torch.searchsorted (bit-identical to the native
extension, which does not build against this torch; SURVEY.md section 4 probe).  Nothing of the
reference is copied: the fixtures are arrays (inputs + the reference's outputs) in .npz files.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from smpl_nerf_amd import synthetic as syn

REF = os.environ.get("SNERF_REFERENCE", "/root/reference")
F32 = np.float32


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


SS_LOG = []


def _import_reference():
    _stub("cv2")
    _stub("imageio")
    tm = _stub("trimesh")
    tm.base = _stub("trimesh.base", Trimesh=object)
    tm.ray = _stub("trimesh.ray")
    tm.ray.ray_triangle = _stub("trimesh.ray.ray_triangle", RayMeshIntersector=object)
    try:
        import importlib
        importlib.import_module("torch.utils.tensorboard")
    except Exception:
        _stub("torch.utils.tensorboard", SummaryWriter=object)

    def searchsorted(a, v, out=None, side="left"):
        r = torch.searchsorted(a, v, right=(side != "left"))
        SS_LOG.append((a.detach().clone(), v.detach().clone(), r.clone()))
        return r

    _stub("torchsearchsorted", searchsorted=searchsorted)
    sys.path.insert(0, REF)
    import utils as U
    from models.render_ray_net import RenderRayNet
    from models.nerf_pipeline import NerfPipeline
    from models.smpl_nerf_pipeline import SmplNerfPipeline
    from models.warp_field_net import WarpFieldNet
    return U, RenderRayNet, NerfPipeline, SmplNerfPipeline, WarpFieldNet


class Args:
    def __init__(self, **kw):
        self.default_device = torch.device("cpu")
        self.sigma_noise_std = 0.0
        self.white_background = 0
        self.run_fine = 1
        self.number_fine_samples = 128
        self.human_pose_encoding = 1
        self.__dict__.update(kw)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def load_params(module, params):
    module.load_state_dict({k: t(v) for k, v in params.items()})
    return module


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    U, RenderRayNet, NerfPipeline, SmplNerfPipeline, WarpFieldNet = _import_reference()
    torch.set_grad_enabled(False) if __name__ == "__main__" else None
    rng = np.random.default_rng(1234)

    # ---- G1 positional encoding (utils.py:114-131) ---------------------------------------------
    x = rng.uniform(-4, 4, (64, 3)).astype(F32)
    g1 = {"x": x}
    for L, ident in [(10, 0), (4, 0), (10, 1), (4, 1), (0, 1)]:
        g1[f"enc_L{L}_id{ident}"] = U.PositionalEncoder(L, ident).encode(t(x)).numpy()
    pose2 = rng.uniform(-1.5, 1.5, (16, 2)).astype(F32)
    g1["pose2"] = pose2
    g1["pose2_enc_L10_id0"] = U.PositionalEncoder(10, 0).encode(t(pose2)).numpy()
    save("g1_posenc.npz", **g1)

    # ---- G2 RenderRayNet forward (models/render_ray_net.py:42-61) ------------------------------
    pe, de = U.PositionalEncoder(10, 0), U.PositionalEncoder(4, 0)
    pts = rng.uniform(-2, 2, (160, 3)).astype(F32)
    dirs = rng.normal(size=(160, 3)).astype(F32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    g2 = {"pts": pts, "dirs": dirs}
    inp = torch.cat([pe.encode(t(pts)), de.encode(t(dirs))], -1)
    g2["inputs"] = inp.numpy()
    for tag, kw, seed in [("skip4", dict(skips=(4,)), 11), ("noskip", dict(skips=()), 12),
                          ("d4w128", dict(n_layers=4, width=128, skips=(1,)), 13)]:
        params = syn.make_render_ray_net_params(seed, sigma_scale=30.0, rgb_scale=10.0, **kw)
        net = load_params(RenderRayNet(n_layers=kw.get("n_layers", 8), width=kw.get("width", 256),
                                       positions_dim=60, directions_dim=24, skips=list(kw["skips"])), params)
        g2[f"raw_{tag}"] = net(inp).numpy()
    params = syn.make_scene_nets(101)[1]
    net = load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), params)
    g2["raw_scene101"] = net(inp).numpy()
    # additional per-ray input (append_smpl_params style, train.py:154-159) and no-direction ablation
    add = rng.uniform(-1, 1, (160, 6)).astype(F32)
    g2["add6"] = add
    params = syn.make_render_ray_net_params(14, 30.0, 10.0, additional_input_dim=6, skips=(4,))
    net = load_params(RenderRayNet(8, 256, 60, 24, additional_input_dim=6, skips=[4]), params)
    g2["raw_add6"] = net(torch.cat([pe.encode(t(pts)), t(add), de.encode(t(dirs))], -1)).numpy()
    params = syn.make_render_ray_net_params(15, 30.0, 10.0, skips=(4,), use_directional_input=0)
    net = load_params(RenderRayNet(8, 256, 60, 24, skips=[4], use_directional_input=0), params)
    g2["raw_nodir"] = net(inp).numpy()
    # WarpFieldNet (models/warp_field_net.py:17-21)
    wparams = syn.make_warp_field_params(21)
    wnet = load_params(WarpFieldNet(8, 256, 60, 40), wparams)
    winp = rng.uniform(-1, 1, (96, 100)).astype(F32)
    g2["warp_inputs"] = winp
    g2["warp_out"] = wnet(t(winp)).numpy()
    save("g2_mlp.npz", **g2)

    # ---- G3 raw2outputs (utils.py:134-191) -----------------------------------------------------
    g3 = {}
    B = 24
    for N in (1, 2, 64, 192, 100):
        raw = rng.normal(0, 2.0, (B, N, 4)).astype(F32)
        raw[..., 3] *= 8.0
        z = np.sort(rng.uniform(1, 4, (B, N)).astype(F32), -1)
        d_ray = rng.normal(size=(B, 3)).astype(F32)
        d_smp = rng.normal(size=(B, N, 3)).astype(F32)
        g3[f"raw_N{N}"], g3[f"z_N{N}"], g3[f"dray_N{N}"], g3[f"dsmp_N{N}"] = raw, z, d_ray, d_smp
        for wb in (0, 1):
            args = Args(white_background=wb)
            for tag, d in (("ray", t(d_ray)[:, None, :].expand(B, N, 3)), ("smp", t(d_smp))):
                if N == 1 and tag == "smp":
                    continue
                out = U.raw2outputs(t(raw), t(z), d, args)
                for nm, o in zip(("rgb", "weights", "alpha"), out):
                    g3[f"{nm}_N{N}_wb{wb}_{tag}"] = o.numpy()
    # injected sigma noise (utils.py:171-173): same tensor added on both sides
    noise = rng.normal(0, 1.0, (B, 64)).astype(F32)
    g3["noise_N64"] = noise
    rawn = g3["raw_N64"].copy()
    rawn[..., 3] += noise
    out = U.raw2outputs(t(rawn), t(g3["z_N64"]), t(g3["dray_N64"])[:, None, :].expand(B, 64, 3), Args())
    g3["rgb_N64_noise"], g3["weights_N64_noise"], g3["alpha_N64_noise"] = [o.numpy() for o in out]
    save("g3_raw2outputs.npz", **g3)

    # ---- G4 sample_pdf / fine_sampling (utils.py:194-264) --------------------------------------
    B, Nc, Nf = 48, 64, 128
    o, d = syn.camera_rays(16, 16, syn.sphere_pose(10, 20, 2.4))
    sel = rng.choice(o.shape[0], B, replace=False)
    _, o32, d32, z = syn.coarse_samples(o[sel], d[sel], 1.0, 4.0, Nc, rng.random(B))
    w = rng.random((B, Nc)).astype(F32) ** 4
    w[0] = 0.0                                    # all zeros -> 1e-5 floor, uniform pdf
    w[1] = 0.37                                   # all equal
    w[2] = 0.0; w[2, 30] = 1.0                    # single spike (denominator branch, cdf ties)
    w[3] = 0.0; w[3, 1] = 0.5; w[3, 62] = 0.5     # mass at both ends
    w[4] = 0.0; w[4, 0] = 1.0; w[4, 63] = 1.0     # only the excluded end weights are non-zero
    w[5] = 1e-9                                   # tiny
    w[6, ::2] = 0.0                               # alternating zeros
    args = Args(number_fine_samples=Nf)
    SS_LOG.clear()
    z_f, pts_f = U.fine_sampling(t(o32), t(d32), t(z), t(w), args)
    cdf, u, inds = SS_LOG[-1]
    z_mid = .5 * (t(z)[..., 1:] + t(z)[..., :-1])
    z_samples = U.sample_pdf(z_mid, t(w)[..., 1:-1], args)
    # the one host-dependent intermediate of the path: the normalising sum of utils.py:200-201 as THIS torch build
    # evaluates it on THIS host (same call, same operand as inside sample_pdf => same bits)
    tot = torch.sum(t(w)[..., 1:-1] + 1e-5, -1, keepdim=True)
    g4 = dict(o=o32, d=d32, z=z, w=w, cdf=cdf.numpy(), u=u.numpy(), inds=inds.numpy(), tot=tot.numpy(),
              z_samples=z_samples.numpy(), z_fine=z_f.numpy(), pts_fine=pts_f.numpy())
    # other (Nc, Nf) shapes
    for nc, nf in ((16, 8), (32, 64), (64, 64), (48, 200)):
        _, oo, dd, zz = syn.coarse_samples(o[sel], d[sel], 1.6, 3.1, nc, rng.random(B))
        ww = rng.random((B, nc)).astype(F32) ** 3
        zf, pf = U.fine_sampling(t(oo), t(dd), t(zz), t(ww), Args(number_fine_samples=nf))
        g4[f"z_{nc}_{nf}"], g4[f"w_{nc}_{nf}"] = zz, ww
        g4[f"zf_{nc}_{nf}"], g4[f"pf_{nc}_{nf}"] = zf.numpy(), pf.numpy()
        g4[f"o_{nc}_{nf}"], g4[f"d_{nc}_{nf}"] = oo, dd
        g4[f"tot_{nc}_{nf}"] = torch.sum(t(ww)[..., 1:-1] + 1e-5, -1, keepdim=True).numpy()
        g4[f"u_{nc}_{nf}"] = torch.linspace(0., 1., steps=nf).numpy()     # utils.py:206 on this host
    save("g4_sampler.npz", **g4)

    # ---- searchsorted known answers (torchsearchsorted semantics incl. ties / out of range) -----
    a = np.array([[0.0, 0.0, 0.25, 0.25, 0.5, 1.0, 1.0]], F32)
    v = np.array([[-1.0, 0.0, 0.1, 0.25, 0.3, 0.5, 0.75, 1.0, 2.0]], F32)
    gs = {"a": a, "v": v}
    for side in ("left", "right"):
        gs[f"out_{side}"] = torch.searchsorted(t(a), t(v), right=(side == "right")).numpy()
    a2 = np.sort(rng.random((7, 50)).astype(F32), -1)
    v2 = rng.random((7, 33)).astype(F32)
    v2[:, :5] = a2[:, 10:15]                      # exact hits
    gs["a2"], gs["v2"] = a2, v2
    for side in ("left", "right"):
        gs[f"out2_{side}"] = torch.searchsorted(t(a2), t(v2), right=(side == "right")).numpy()
    save("g_searchsorted.npz", **gs)

    # ---- G5 NerfPipeline.forward (models/nerf_pipeline.py:14-67), 128x128 frame ----------------
    pc, pf = syn.make_scene_nets(101)
    mc = load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pc)
    mf = load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pf)
    g5 = {}
    for tag, kw in (("nf14", dict(near=1.0, far=4.0)), ("nf1631wb", dict(near=1.6, far=3.1))):
        args = Args(white_background=1 if tag.endswith("wb") else 0)
        pipe = NerfPipeline(mc, mf, args, pe, de)
        data = syn.frame_batch(128, 128, phi=0.0, theta=0.0, seed=7, **kw)
        outs = [[], [], [], []]
        for s in range(0, 16384, 2048):
            r = pipe([t(a[s:s + 2048]) for a in data])
            for lst, o_ in zip(outs, r):
                lst.append(o_.numpy())
        rgb, rgb_fine, pts_fine, alpha_fine = [np.concatenate(l) for l in outs]
        sub = np.arange(0, 16384, 64) + (np.arange(256) % 64)       # 256 strided rays, all columns hit
        g5[f"sub_{tag}"] = sub
        g5[f"rgb_{tag}"], g5[f"rgb_fine_{tag}"] = rgb, rgb_fine
        g5[f"pts_fine_sub_{tag}"], g5[f"alpha_fine_sub_{tag}"] = pts_fine[sub], alpha_fine[sub]
        if tag == "nf14":
            pipe_c = NerfPipeline(mc, mf, Args(run_fine=0), pe, de)
            r = pipe_c([t(a[sub]) for a in data])
            g5["coarse_only_rgb"], g5["coarse_only_alpha"] = r[0].numpy(), r[3].numpy()
    save("g5_nerf_pipeline.npz", **g5)

    # ---- G6 SmplNerfPipeline.forward (models/smpl_nerf_pipeline.py:16-100) ---------------------
    pw = syn.make_warp_field_params(103, out_scale=0.3)
    mw = load_params(WarpFieldNet(8, 256, 60, 40), pw)
    pose_enc = U.PositionalEncoder(10, 0)
    poses = syn.human_poses((41, 38), 0, 60, 10)
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    sub = np.arange(0, 16384, 256) + (np.arange(64) * 5 % 128)
    gp = poses[np.arange(64) % 10]
    g6 = {"sub": sub, "goal_pose": gp}
    for wb in (0, 1):
        pipe = SmplNerfPipeline(mc, mf, mw, Args(white_background=wb), pe, de, pose_enc)
        r = pipe([t(a[sub]) for a in data[:4]] + [t(gp), t(data[4][sub])])
        for nm, o_ in zip(("rgb", "rgb_fine", "warp_fine", "pts_fine", "warped_fine", "alpha_fine"), r):
            g6[f"{nm}_wb{wb}"] = o_.numpy()
    pipe = SmplNerfPipeline(mc, mf, mw, Args(run_fine=0), pe, de, pose_enc)
    r = pipe([t(a[sub]) for a in data[:4]] + [t(gp), t(data[4][sub])])
    g6["coarse_rgb"], g6["coarse_warp"], g6["coarse_warped"], g6["coarse_alpha"] = (
        r[0].numpy(), r[2].numpy(), r[4].numpy(), r[5].numpy())
    save("g6_smpl_nerf_pipeline.npz", **g6)

    # ---- G8 get_rays + CoarseSampling (utils.py:26-54, datasets/transforms.py:58-90) -----------
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_transforms", os.path.join(REF, "datasets", "transforms.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    from camera import get_sphere_pose
    pose = get_sphere_pose(12.0, -30.0, 2.4)
    ro, rd = U.get_rays(16, 24, syn.focal_length(24), pose)
    cs, tt = tr.CoarseSampling(1.0, 4.0, 64), tr.ToTensor()
    np.random.seed(5)
    jit = []
    rows = []
    flat_o, flat_d = ro.reshape(-1, 3), rd.reshape(-1, 3)
    st = np.random.get_state()
    for k in range(0, flat_o.shape[0], 37):
        rows.append(k)
        out = tt(cs((flat_o[k], flat_d[k], np.zeros(3, F32))))
        jit.append([o_.numpy() for o_ in out[:4]])
    np.random.set_state(st)
    jitter = np.array([np.random.rand() for _ in rows])
    save("g8_rays.npz", pose=pose, rays_o=ro, rays_d=rd, rows=np.array(rows), jitter=jitter,
         samples=np.stack([j[0] for j in jit]), o=np.stack([j[1] for j in jit]),
         d=np.stack([j[2] for j in jit]), z=np.stack([j[3] for j in jit]),
         pose_ref_0_0=get_sphere_pose(0, 0, 2.4), pose_ref_b=get_sphere_pose(-20.0, 75.0, 2.4))


if __name__ == "__main__":
    main()
