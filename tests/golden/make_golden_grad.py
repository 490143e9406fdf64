#!/usr/bin/env python3
"""Gradient / training golden vectors, produced by running the REFERENCE under torch autograd
(build container only; see make_golden.py for how the reference is imported).

    python tests/golden/make_golden_grad.py        # writes g7_grads.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch

import make_golden as MG
from smpl_nerf_amd import synthetic as syn

F32 = np.float32
t = MG.t


def param_digest(named):
    """Per-parameter (sum, l2, first 16 values) - small but sensitive to any element."""
    out = {}
    for k, v in named:
        a = v.detach().numpy().astype(np.float64).reshape(-1)
        out[k] = np.concatenate([[a.sum(), np.sqrt((a * a).sum())], a[:16]])
    return out


def main():
    U, RenderRayNet, NerfPipeline, _, _ = MG._import_reference()
    rng = np.random.default_rng(4321)
    g = {}

    # ---- raw2outputs backward (autograd through utils.py:161-191) ---------------------------------
    g3 = MG.np.load(os.path.join(HERE, "g3_raw2outputs.npz"))
    for N in (1, 2, 64, 192, 100):
        B = g3[f"raw_N{N}"].shape[0]
        gout = rng.normal(size=(B, 3)).astype(F32)
        g[f"c_gout_N{N}"] = gout
        for wb in (0, 1):
            for mode in ("ray", "smp"):
                if N == 1 and mode == "smp":
                    continue
                raw = t(g3[f"raw_N{N}"]).requires_grad_(True)
                d = t(g3[f"dray_N{N}"])[:, None, :].expand(B, N, 3) if mode == "ray" else t(g3[f"dsmp_N{N}"])
                rgb, _, _ = U.raw2outputs(raw, t(g3[f"z_N{N}"]), d, MG.Args(white_background=wb))
                (rgb * t(gout)).sum().backward()
                g[f"c_draw_N{N}_wb{wb}_{mode}"] = raw.grad.numpy()

    # ---- RenderRayNet backward: every parameter gradient of a small net, digests of the 8x256 net ------
    g2 = MG.np.load(os.path.join(HERE, "g2_mlp.npz"))
    inp = t(g2["inputs"])
    gout = rng.normal(size=(inp.shape[0], 4)).astype(F32)
    g["m_gout"] = gout
    kw = dict(n_layers=4, width=128, skips=(1,))
    net = MG.load_params(RenderRayNet(4, 128, 60, 24, skips=[1]), syn.make_render_ray_net_params(13, 30.0, 10.0, **kw))
    (net(inp) * t(gout)).sum().backward()
    for k, p in net.named_parameters():
        g[f"m_d4w128/{k}"] = p.grad.numpy()
    for tag, params, skips in (("skip4", syn.make_render_ray_net_params(11, 30.0, 10.0, skips=(4,)), [4]),
                               ("noskip", syn.make_render_ray_net_params(12, 30.0, 10.0, skips=()), []),
                               ("scene", syn.make_scene_nets(101)[1], [4])):
        net = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=skips), params)
        (net(inp) * t(gout)).sum().backward()
        for k, v in param_digest((k, p.grad) for k, p in net.named_parameters()).items():
            g[f"m_{tag}/{k}"] = v

    # ---- NerfPipeline training: loss + grads of one step, three Adam steps (solver/nerf_solver.py:31-33,48-52,83-87)
    pc, pf = syn.make_scene_nets(101)
    mc = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pc)
    mf = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pf)
    pe, de = U.PositionalEncoder(10, 0), U.PositionalEncoder(4, 0)
    pipe = NerfPipeline(mc, mf, MG.Args(), pe, de)
    data = syn.frame_batch(128, 128, seed=7)
    sub = np.arange(0, 16384, 128) + (np.arange(128) * 3 % 128)
    g["t_sub"] = sub
    batch = [t(a[sub]) for a in data]
    optim = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4, betas=(0.9, 0.999), eps=1e-8,
                             weight_decay=0)
    loss_fn = torch.nn.MSELoss()
    losses = []
    for step in range(3):
        rgb, rgb_fine, _, _ = pipe(batch)
        optim.zero_grad()
        loss = loss_fn(rgb, batch[-1]) + loss_fn(rgb_fine, batch[-1])
        loss.backward()
        if step == 0:
            g["t_rgb0"], g["t_rgb_fine0"] = rgb.detach().numpy(), rgb_fine.detach().numpy()
            for k, v in param_digest((f"coarse.{k}", p.grad) for k, p in mc.named_parameters()).items():
                g[f"t_grad0/{k}"] = v
            for k, v in param_digest((f"fine.{k}", p.grad) for k, p in mf.named_parameters()).items():
                g[f"t_grad0/{k}"] = v
        optim.step()
        losses.append(loss.item())
    g["t_losses"] = np.array(losses)
    for k, v in param_digest((f"coarse.{k}", p) for k, p in mc.named_parameters()).items():
        g[f"t_param3/{k}"] = v
    for k, v in param_digest((f"fine.{k}", p) for k, p in mf.named_parameters()).items():
        g[f"t_param3/{k}"] = v
    # coarse-only training (run_fine=0): loss = 2*MSE_coarse, fine net gets no gradient (quirk Q10)
    mc = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pc)
    pipe = NerfPipeline(mc, mf, MG.Args(run_fine=0), pe, de)
    rgb, rgb_fine, _, _ = pipe(batch)
    loss = loss_fn(rgb, batch[-1]) + loss_fn(rgb_fine, batch[-1])
    mc.zero_grad()
    loss.backward()
    g["t_coarse_only_loss"] = np.array([loss.item()])
    for k, v in param_digest((f"coarse.{k}", p.grad) for k, p in mc.named_parameters()).items():
        g[f"t_coarse_only_grad/{k}"] = v
    MG.save("g7_grads.npz", **g)


if __name__ == "__main__":
    torch.set_grad_enabled(True)
    main()
