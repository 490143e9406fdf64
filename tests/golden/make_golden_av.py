#!/usr/bin/env python3
"""Golden vectors for AppendVerticesPipeline (models/append_vertices_pipeline.py) from the reference
itself, with the synthetic body model / index estimator standing in for smplx + the SMPL .pkl.
    python tests/golden/make_golden_av.py    # writes g9_append_vertices.npz"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch

import make_golden as MG
from smpl_nerf_amd import synthetic as syn
from smpl_nerf_amd.synthetic_smpl import IndexPoseEstimator, LinearBodyModel

t = MG.t


def main():
    U, RenderRayNet, NerfPipeline, _, _ = MG._import_reference()
    from models.append_vertices_net import AppendVerticesNet
    from models.append_vertices_pipeline import AppendVerticesPipeline
    torch.set_grad_enabled(False)
    pe, de = U.PositionalEncoder(10, 0), U.PositionalEncoder(4, 0)
    pc, pf = syn.make_append_vertices_params(201), syn.make_append_vertices_params(202)
    mc = MG.load_params(AppendVerticesNet(8, 256, 60, 24, 6890, additional_input_layers=1, skips=[4]), pc)
    mf = MG.load_params(AppendVerticesNet(8, 256, 60, 24, 6890, additional_input_layers=1, skips=[4]), pf)
    poses = torch.from_numpy(syn.human_poses((41, 38), 0, 60, 10))
    est = IndexPoseEstimator(poses, torch.zeros(1, 10))
    body = LinearBodyModel(seed=3)
    data = syn.frame_batch(128, 128, phi=3.0, theta=-10.0, seed=11)
    B = 24
    sub = np.arange(B) * 683
    idx = np.arange(B) % 10
    g = {"sub": sub, "images": idx}
    # NOTE: with run_fine=1 the reference raises (models/append_vertices_pipeline.py:71 expands an already
    # 3-D `vertices_flat`), so only the coarse-only path has reference results to pin.
    for wb in (0, 1):
        pipe = AppendVerticesPipeline(mc, mf, est, body, MG.Args(white_background=wb, run_fine=0), pe, de)
        pipe.global_orient = torch.zeros([1, 3])
        out = pipe([t(a[sub]) for a in data[:4]] + [torch.from_numpy(idx), t(data[4][sub])])
        g[f"coarse_rgb_wb{wb}"], g[f"coarse_alpha_wb{wb}"] = out[0].numpy(), out[3].numpy()
        assert out[0] is out[1] and tuple(out[2].shape) == (B, 64, 3)
    try:
        pipe = AppendVerticesPipeline(mc, mf, est, body, MG.Args(run_fine=1), pe, de)
        pipe.global_orient = torch.zeros([1, 3])
        pipe([t(a[sub]) for a in data[:4]] + [torch.from_numpy(idx), t(data[4][sub])])
        g["reference_fine_branch_runs"] = np.array([1])
    except RuntimeError:
        g["reference_fine_branch_runs"] = np.array([0])
    # the literal AppendVerticesNet.forward(x) on full-width rows (20670 + 60 + 24 columns)
    rng = np.random.default_rng(5)
    x = rng.normal(0, 0.5, (40, 20670 + 60 + 24)).astype(np.float32)
    g["net_rows"] = x[:, np.r_[0:60, 20754 - 24:20754]]          # the only columns the net reads
    g["net_out"] = mc(t(x)).numpy()
    MG.save("g9_append_vertices.npz", **g)

    # ---- AppendSmplParamsPipeline / AppendToNerfPipeline (models/append_smpl_params_pipeline.py, append_to_nerf_pipeline.py)
    from models.append_smpl_params_pipeline import AppendSmplParamsPipeline
    from models.append_to_nerf_pipeline import AppendToNerfPipeline
    h = {"sub": sub, "images": idx}
    hpe = U.PositionalEncoder(10, 0)
    gp = poses[idx] + 0.05 * torch.from_numpy(np.random.default_rng(9).normal(size=(B, 69)).astype(np.float32))
    h["goal_pose"] = gp.numpy()
    batch = [t(a[sub]) for a in data[:4]] + [gp, t(data[4][sub])]
    for name, cls, npose in (("smpl", AppendSmplParamsPipeline, 69), ("two", AppendToNerfPipeline, 2)):
        for enc in (0, 1):
            add = npose * (20 if enc else 1)
            pcs = syn.make_scene_net_params(301 + enc, add_first=True, additional_input_dim=add)
            pfs = syn.make_scene_net_params(303 + enc, add_first=True, additional_input_dim=add)
            mcs = MG.load_params(RenderRayNet(8, 256, 60, 24, add, skips=[4]), pcs)
            mfs = MG.load_params(RenderRayNet(8, 256, 60, 24, add, skips=[4]), pfs)
            pipe = cls(mcs, mfs, MG.Args(human_pose_encoding=enc), pe, de, hpe)
            out = pipe(batch)
            for nm, o in zip(("rgb", "rgb_fine", "pts_fine", "alpha_fine"), out):
                h[f"{name}{enc}_{nm}"] = o.numpy()
    # one training step of append_smpl_params (raw pose, 69 extra columns): loss + gradient digests
    torch.set_grad_enabled(True)
    import make_golden_grad as GG
    pcs, pfs = (syn.make_scene_net_params(301, add_first=True, additional_input_dim=69),
                syn.make_scene_net_params(303, add_first=True, additional_input_dim=69))
    mcs = MG.load_params(RenderRayNet(8, 256, 60, 24, 69, skips=[4]), pcs)
    mfs = MG.load_params(RenderRayNet(8, 256, 60, 24, 69, skips=[4]), pfs)
    pipe = AppendSmplParamsPipeline(mcs, mfs, MG.Args(human_pose_encoding=0), pe, de, hpe)
    rgb, rgb_fine, _, _ = pipe(batch)
    loss = torch.nn.functional.mse_loss(rgb, batch[-1]) + torch.nn.functional.mse_loss(rgb_fine, batch[-1])
    loss.backward()
    h["train_loss"] = np.array([loss.item()])
    for k, v in GG.param_digest((f"coarse.{k}", p.grad) for k, p in mcs.named_parameters()).items():
        h[f"train_grad/{k}"] = v
    for k, v in GG.param_digest((f"fine.{k}", p.grad) for k, p in mfs.named_parameters()).items():
        h[f"train_grad/{k}"] = v
    MG.save("g10_append_pose.npz", **h)


if __name__ == "__main__":
    main()
