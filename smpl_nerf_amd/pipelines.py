"""NerfPipeline drop-in (models/nerf_pipeline.py:8-67).

Same constructor, same `forward(data)` contract - `data` is the list
[ray_samples[B,Nc,3], ray_translation[B,3], ray_direction[B,3], z_vals[B,Nc], rgb_truth[B,3]] a Solver
hands over (solver/nerf_solver.py:77-81) and the result is the 4-tuple
(rgb, rgb_fine, ray_samples_fine, densities) - but the whole march is five HIP launches:

    fused encode+MLP (coarse)  ->  composite  ->  inverse-CDF sampler+merge+points
                               ->  fused encode+MLP (fine)  ->  composite

No [B*N, 84] encoded input, no [B*N, 256] activation and no [B, 128, 63] gather operand is ever
materialised in HBM.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class NerfPipeline(nn.Module):

    def __init__(self, model_coarse, model_fine, args, position_encoder, direction_encoder):
        super().__init__()
        self.device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")  # singe_sample_pipeline.py:11
        self.args = args
        self.model_coarse = model_coarse
        self.model_fine = model_fine
        self.position_encoder = position_encoder
        self.direction_encoder = direction_encoder

    def _noise(self, shape, device):
        # utils.py:171-173: drawn whenever sigma_noise_std > 0, also in eval mode (quirk Q3)
        std = getattr(self.args, "sigma_noise_std", 0.)
        return torch.normal(0, std, shape, device=device) if std > 0. else None

    def forward(self, data):
        ray_samples, ray_translation, ray_direction, z_vals, _ = data
        args = self.args
        B, Nc = z_vals.shape
        wb = bool(args.white_background)
        dev = ray_samples.device
        # coarse net on the given samples (:29-41)
        raw = self.model_coarse.forward_fused(ray_samples, ray_direction, Nc, self.position_encoder,
                                              self.direction_encoder)
        rgb, weights, densities = ops.composite(raw.view(B, Nc, 4), z_vals, ray_direction, wb,
                                                self._noise((B, Nc), dev))                        # :42
        if not args.run_fine:
            return rgb, rgb, ray_samples, densities                                               # :43-44
        # hierarchical samples (:47) and the fine net on them (:49-60)
        hs = ops.hierarchical_samples(ray_translation, ray_direction, z_vals, weights, args.number_fine_samples)
        z_fine, ray_samples_fine = hs["z_fine"], hs["pts"]
        N = z_fine.shape[1]
        raw_fine = self.model_fine.forward_fused(ray_samples_fine, ray_direction, N, self.position_encoder,
                                                 self.direction_encoder)
        rgb_fine, _, densities_fine = ops.composite(raw_fine.view(B, N, 4), z_fine, ray_direction, wb,
                                                    self._noise((B, N), dev), want_weights=False)  # :65
        return rgb, rgb_fine, ray_samples_fine, densities_fine                                    # :67
