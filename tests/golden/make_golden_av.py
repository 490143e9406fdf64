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


if __name__ == "__main__":
    main()
