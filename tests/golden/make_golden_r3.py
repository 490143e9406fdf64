#!/usr/bin/env python3
"""Round-3 golden vectors, produced by running the REFERENCE under torch autograd (build container only; see
make_golden.py for how the reference is imported).  Nothing of the reference is copied: arrays in .npz files.

    python tests/golden/make_golden_r3.py

    g12_smpl_raw_pose.npz   SmplNerfPipeline with human_pose_encoding = 0 (the parser default, config_parser.py:72;
                            models/smpl_nerf_pipeline.py:40-45; WarpFieldNet(positions_dim=3, pose_dim=2), train.py:111-114),
                            run_fine = 0 - the only mode in which the reference runs it: outputs, loss, gradients.
    g13_warp_net_grad.npz   WarpFieldNet.forward(x) (models/warp_field_net.py:17-21) under autograd: output, d x, parameter
                            gradients, for the encoded (100 columns) and the raw (5 columns) input widths.
    g14_ops_grads.npz       PositionalEncoder.encode (utils.py:114-131) and raw2outputs (utils.py:134-191) as differentiable
                            stand-alone ops: gradients through all three outputs to raw, z_vals and the directions;
                            SmplNerfSolver's density loss on the returned alpha (solver/smpl_nerf_solver.py:35-43).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch

import make_golden as MG
import make_golden_grad as GG
from smpl_nerf_amd import synthetic as syn

F32 = np.float32
t = MG.t


def main():
    U, RenderRayNet, NerfPipeline, SmplNerfPipeline, WarpFieldNet = MG._import_reference()
    torch.set_grad_enabled(True)
    rng = np.random.default_rng(20250929)
    mse = torch.nn.functional.mse_loss

    # ---- g12: human_pose_encoding = 0, run_fine = 0 ---------------------------------------------------------------
    g6 = np.load(os.path.join(HERE, "g6_smpl_nerf_pipeline.npz"))
    pc, pf = syn.make_scene_nets(101)
    pw = syn.make_warp_field_params(105, positions_dim=3, pose_dim=2, out_scale=0.3)
    pe, de, he = U.PositionalEncoder(10, 0), U.PositionalEncoder(4, 0), U.PositionalEncoder(10, 0)
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    sub = g6["sub"]
    g = {"sub": sub, "goal_pose": g6["goal_pose"]}
    for k, v in pw.items():
        g[f"warp_param/{k}"] = v
    for wb in (0, 1):
        batch = [t(a[sub]) for a in data[:4]] + [t(g6["goal_pose"]), t(data[4][sub])]
        mc = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pc)
        mf = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pf)
        mw = MG.load_params(WarpFieldNet(8, 256, 3, 2), pw)
        args = MG.Args(white_background=wb, run_fine=0, human_pose_encoding=0)
        pipe = SmplNerfPipeline(mc, mf, mw, args, pe, de, he)
        out = pipe(batch)
        loss = mse(out[0], batch[-1]) + mse(out[1], batch[-1])
        loss.backward()
        for nm, o_ in zip(("rgb", "rgb_fine", "warp", "samples", "warped", "alpha"), out):
            g[f"{nm}_wb{wb}"] = o_.detach().numpy()
        g[f"loss_wb{wb}"] = np.array([loss.item()])
        for k, v in GG.param_digest((f"coarse.{k}", p.grad) for k, p in mc.named_parameters()).items():
            g[f"grad_wb{wb}/{k}"] = v
        for k, p in mw.named_parameters():
            g[f"warpgrad_wb{wb}/{k}"] = p.grad.numpy()
        assert all(p.grad is None for p in mf.parameters())
    # the reference fails in the fine branch in this mode (quirk Q5): record that it does
    try:
        SmplNerfPipeline(mc, mf, mw, MG.Args(run_fine=1, human_pose_encoding=0), pe, de, he)(batch)
        g["fine_branch_raises"] = np.array([0])
    except RuntimeError:
        g["fine_branch_raises"] = np.array([1])
    MG.save("g12_smpl_raw_pose.npz", **g)

    # ---- g13: WarpFieldNet.forward(x) under autograd --------------------------------------------------------------
    g = {}
    for tag, pdim, qdim, seed in (("enc", 60, 40, 21), ("raw", 3, 2, 105)):
        params = syn.make_warp_field_params(seed, positions_dim=pdim, pose_dim=qdim)
        net = MG.load_params(WarpFieldNet(8, 256, pdim, qdim), params)
        x = t(rng.uniform(-1, 1, (200, pdim + qdim)).astype(F32)).requires_grad_(True)
        gout = t(rng.normal(size=(200, 3)).astype(F32))
        out = net(x)
        (out * gout).sum().backward()
        g[f"x_{tag}"], g[f"gout_{tag}"], g[f"out_{tag}"], g[f"dx_{tag}"] = (x.detach().numpy(), gout.numpy(),
                                                                            out.detach().numpy(), x.grad.numpy())
        for k, p in net.named_parameters():
            g[f"grad_{tag}/{k}"] = p.grad.numpy()
            g[f"param_{tag}/{k}"] = params[k]
    MG.save("g13_warp_net_grad.npz", **g)

    # ---- g14: differentiable stand-alone ops ----------------------------------------------------------------------
    g = {}
    x = rng.uniform(-2, 2, (96, 3)).astype(F32)
    g["pe_x"] = x
    for L, ident in ((10, 0), (4, 1), (0, 1), (6, 0)):
        enc = U.PositionalEncoder(L, ident)
        xt = t(x).requires_grad_(True)
        out = enc.encode(xt)
        gout = rng.normal(size=tuple(out.shape)).astype(F32)
        (out * t(gout)).sum().backward()
        g[f"pe_gout_L{L}_id{ident}"], g[f"pe_dx_L{L}_id{ident}"] = gout, xt.grad.numpy()
    pose = rng.uniform(-1.5, 1.5, (32, 2)).astype(F32)          # a 2-channel input (the two joint angles)
    pt = t(pose).requires_grad_(True)
    out = U.PositionalEncoder(10, 0).encode(pt)
    gout = rng.normal(size=tuple(out.shape)).astype(F32)
    (out * t(gout)).sum().backward()
    g["pe_pose"], g["pe_pose_gout"], g["pe_pose_dx"] = pose, gout, pt.grad.numpy()

    g3 = np.load(os.path.join(HERE, "g3_raw2outputs.npz"))
    for N in (1, 2, 64, 192, 100):
        B = g3[f"raw_N{N}"].shape[0]
        g_rgb = rng.normal(size=(B, 3)).astype(F32)
        g_w = rng.normal(size=(B, N)).astype(F32)
        g_a = rng.normal(size=(B, N)).astype(F32)
        g[f"c_grgb_N{N}"], g[f"c_gw_N{N}"], g[f"c_ga_N{N}"] = g_rgb, g_w, g_a
        for wb in (0, 1):
            for mode in ("ray", "smp"):
                if N == 1 and mode == "smp":
                    continue
                raw = t(g3[f"raw_N{N}"]).requires_grad_(True)
                z = t(g3[f"z_N{N}"]).requires_grad_(True)
                d0 = t(g3[f"dray_N{N}"] if mode == "ray" else g3[f"dsmp_N{N}"]).requires_grad_(True)
                d = d0[:, None, :].expand(B, N, 3) if mode == "ray" else d0
                rgb, w, a = U.raw2outputs(raw, z, d, MG.Args(white_background=wb))
                ((rgb * t(g_rgb)).sum() + (w * t(g_w)).sum() + (a * t(g_a)).sum()).backward()
                key = f"N{N}_wb{wb}_{mode}"
                g[f"c_draw_{key}"] = raw.grad.numpy()
                g[f"c_dz_{key}"] = z.grad.numpy() if z.grad is not None else np.zeros((B, N), F32)
                g[f"c_ddir_{key}"] = d0.grad.numpy() if d0.grad is not None else np.zeros(tuple(d0.shape), F32)
    MG.save("g14_ops_grads.npz", **g)


def sample_pdf_grads():
    """g16_sample_pdf_grad.npz: sample_pdf (utils.py:194-228) under autograd - the gradient w.r.t. bins and weights that the
    reference's own autograd gives (fine_sampling detaches the result, sample_pdf itself does not)."""
    U = MG._import_reference()[0]
    torch.set_grad_enabled(True)
    g4 = np.load(os.path.join(HERE, "g4_sampler.npz"))
    rng = np.random.default_rng(424242)
    z, w = g4["z"], g4["w"]
    bins = t(.5 * (z[..., 1:] + z[..., :-1])).requires_grad_(True)
    wi = t(np.ascontiguousarray(w[..., 1:-1])).requires_grad_(True)
    out = U.sample_pdf(bins, wi, MG.Args(number_fine_samples=128))
    gout = rng.normal(size=tuple(out.shape)).astype(F32)
    (out * t(gout)).sum().backward()
    MG.save("g16_sample_pdf_grad.npz", bins=bins.detach().numpy(), weights=wi.detach().numpy(), gout=gout,
            samples=out.detach().numpy(), d_bins=bins.grad.numpy(), d_weights=wi.grad.numpy())


if __name__ == "__main__":
    if "--only-g16" in sys.argv:
        sample_pdf_grads()
    else:
        main()
        sample_pdf_grads()
