#!/usr/bin/env python3
"""Training golden for SmplNerfPipeline (models/smpl_nerf_pipeline.py + solver/smpl_nerf_solver.py:35-43):
one step under autograd in the reference -> loss and gradient digests of the coarse, fine and warp nets.
    python tests/golden/make_golden_smpl_grad.py     # writes g11_smpl_grads.npz"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch

import make_golden as MG
import make_golden_grad as GG
from smpl_nerf_amd import synthetic as syn

t = MG.t


def main():
    U, RenderRayNet, NerfPipeline, SmplNerfPipeline, WarpFieldNet = MG._import_reference()
    torch.set_grad_enabled(True)
    g6 = np.load(os.path.join(HERE, "g6_smpl_nerf_pipeline.npz"))
    pc, pf = syn.make_scene_nets(101)
    pw = syn.make_warp_field_params(103, out_scale=0.3)
    pe, de, he = U.PositionalEncoder(10, 0), U.PositionalEncoder(4, 0), U.PositionalEncoder(10, 0)
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    sub = g6["sub"]
    batch = [t(a[sub]) for a in data[:4]] + [t(g6["goal_pose"]), t(data[4][sub])]
    g = {}
    for wb in (0, 1):
        mc = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pc)
        mf = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pf)
        mw = MG.load_params(WarpFieldNet(8, 256, 60, 40), pw)
        pipe = SmplNerfPipeline(mc, mf, mw, MG.Args(white_background=wb), pe, de, he)
        out = pipe(batch)
        loss = torch.nn.functional.mse_loss(out[0], batch[-1]) + torch.nn.functional.mse_loss(out[1], batch[-1])
        loss.backward()
        g[f"loss_wb{wb}"] = np.array([loss.item()])
        for name, m in (("coarse", mc), ("fine", mf), ("warp", mw)):
            for k, v in GG.param_digest((f"{name}.{k}", p.grad) for k, p in m.named_parameters()).items():
                g[f"grad_wb{wb}/{k}"] = v
            if name == "warp":
                for k, p in m.named_parameters():
                    g[f"warpfull_wb{wb}/{k}"] = p.grad.numpy()
    MG.save("g11_smpl_grads.npz", **g)


if __name__ == "__main__":
    main()
