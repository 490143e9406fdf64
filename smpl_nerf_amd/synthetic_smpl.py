"""Stand-ins for the SMPL assets the reference needs but does not ship (SMPL .pkl, smplx): a linear
"body model" with the call contract AppendVerticesPipeline uses (models/append_vertices_pipeline.py:38-40)
and the index-based pose estimator (models/dummy_smpl_estimator_model.py:6-27).  Torch modules, device
agnostic; used by the golden generator, the tests and the synthetic benchmarks."""
from __future__ import annotations

import types

import numpy as np
import torch


class LinearBodyModel(torch.nn.Module):
    """vertices[b] = V0 + sum_j body_pose[b, j] * Vj for the two animated joints 38 and 41
    (6890 x 3 like SMPL).  `betas`/`global_orient` are accepted and ignored."""

    def __init__(self, seed: int = 0, n_vertices: int = 6890, scale: float = 0.3):
        super().__init__()
        rng = np.random.default_rng(seed)
        self.register_buffer("v0", torch.from_numpy(rng.normal(0, scale, (n_vertices, 3)).astype(np.float32)))
        self.register_buffer("v38", torch.from_numpy(rng.normal(0, scale, (n_vertices, 3)).astype(np.float32)))
        self.register_buffer("v41", torch.from_numpy(rng.normal(0, scale, (n_vertices, 3)).astype(np.float32)))

    def forward(self, betas=None, return_verts=True, body_pose=None, global_orient=None):
        v = (self.v0[None] + body_pose[:, 38, None, None] * self.v38[None]
             + body_pose[:, 41, None, None] * self.v41[None])
        return types.SimpleNamespace(vertices=v)


class IndexPoseEstimator(torch.nn.Module):
    """models/dummy_smpl_estimator_model.py:21-27: x are indices into preset goal poses; betas are shared."""

    def __init__(self, goal_poses, betas):
        super().__init__()
        self.betas = torch.nn.Parameter(betas.data, requires_grad=False)
        self.goal_poses = torch.nn.Parameter(goal_poses.data, requires_grad=False)

    def forward(self, x):
        return self.goal_poses[x], self.betas.expand(len(x), -1)
