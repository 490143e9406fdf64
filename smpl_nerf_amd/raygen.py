"""On-device replacement of the reference's per-ray input pipeline:
RaysFromImagesDataset.__init__/__getitem__ (datasets/rays_from_images_dataset.py:41-79) + get_rays
(utils.py:50-54) + CoarseSampling / ToTensor (datasets/transforms.py:13-21, 58-90) for whole batches, with
the images and poses resident on the GPU.  Bit-identical to the numpy path (fp64 in the reference's operation
order, rounded to fp32 once)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, ptr


def coarse_bin_tables(near: float, far: float, number_samples: int):
    """lower[k], upper[k]-lower[k] of CoarseSampling.__call__ (datasets/transforms.py:82-86), evaluated with
    the reference's own numpy expression."""
    t_vals = np.linspace(0., 1., number_samples)
    z_vals = 1. / (1. / near * (1. - t_vals) + 1. / far * (t_vals))
    mids = .5 * (z_vals[1:] + z_vals[:-1])
    upper = np.concatenate([mids, z_vals[-1:]], -1)
    lower = np.concatenate([z_vals[:1], mids], -1)
    return lower, upper - lower


class RayGenerator:
    """Holds poses [F,4,4] (fp64) and optionally images [F,H,W,3] (fp32 in [0,1], BGR like cv2.imread / 255,
    datasets/transforms.py:33) on the device and produces pipeline input lists for arbitrary ray batches."""

    def __init__(self, poses, h: int, w: int, camera_angle_x: float, near: float, far: float, number_samples: int,
                 device, images=None):
        self.h, self.w = int(h), int(w)
        self.focal = float(.5 * w / np.tan(.5 * camera_angle_x))       # rays_from_images_dataset.py:44
        self.n = int(number_samples)
        self.device = torch.device(device)
        poses = np.ascontiguousarray(np.asarray(poses, np.float64).reshape(-1, 4, 4))
        self.n_frames = poses.shape[0]
        self.frame_ids = list(range(self.n_frames))     # global frame numbers (for_rank: this rank's shard)
        self.poses = torch.from_numpy(poses).to(self.device)
        lower, span = coarse_bin_tables(near, far, number_samples)
        self.lower = torch.from_numpy(np.ascontiguousarray(lower)).to(self.device)
        self.span = torch.from_numpy(np.ascontiguousarray(span)).to(self.device)
        self.images = None
        if images is not None:
            self.images = torch.as_tensor(images, dtype=torch.float32, device=self.device).reshape(-1, 3).contiguous()

    @classmethod
    def for_rank(cls, poses, h, w, camera_angle_x, near, far, number_samples, device, images=None, world=None, rank=None):
        """This rank's shard of a data set, by image: frames rank, rank + world, ... of `poses` / `images`
        (host arrays) - only those are moved to the device.  `frame_ids` records which global frames they are."""
        from . import dist as sdist
        if world is None:
            world, rank = sdist.world_rank()
        if len(poses) < world:
            # checked on EVERY rank: a rank with an empty shard would draw nothing while the others enter the step's
            # all-reduce, and the job would hang
            raise ValueError(f"RayGenerator.for_rank: {len(poses)} frames cannot be sharded by image over {world} ranks "
                             "(every rank needs at least one frame)")
        ids = sdist.shard_frame_indices(len(poses), world, rank)
        poses = np.asarray(poses, np.float64).reshape(-1, 4, 4)[ids]
        if images is not None:
            images = np.asarray(images)[ids]
        gen = cls(poses, h, w, camera_angle_x, near, far, number_samples, device, images)
        gen.frame_ids = ids
        return gen

    @property
    def n_rays(self):
        return self.n_frames * self.h * self.w

    def batch(self, ray_index: torch.Tensor, jitter: torch.Tensor):
        """ray_index int64 [B] (frame*H*W + row*W + col), jitter fp64 [B] in [0,1) ->
        [ray_samples [B,Nc,3], ray_translation [B,3], ray_direction [B,3], z_vals [B,Nc], rgb_truth [B,3]].
        Without images the rgb_truth slot holds zeros, so the list always has the five entries the pipelines unpack.
        Indices outside [0, n_rays) are not checked on the host (that would synchronise); the kernel never reads out
        of bounds for them and returns NaN rays (and indexing the image table raises a device-side assert)."""
        if not ray_index.is_cuda or not jitter.is_cuda:
            raise RuntimeError("RayGenerator.batch: ray_index and jitter must be on the GPU (no CPU path)")
        idx = ray_index.to(torch.int64).contiguous()
        jit = jitter.to(torch.float64).contiguous()
        B = idx.shape[0]
        dev = self.device
        samples = torch.empty((B, self.n, 3), device=dev, dtype=torch.float32)
        o = torch.empty((B, 3), device=dev, dtype=torch.float32)
        d = torch.empty((B, 3), device=dev, dtype=torch.float32)
        z = torch.empty((B, self.n), device=dev, dtype=torch.float32)
        lib = _lib.load()
        with torch.cuda.device(dev), _lib.timed("raygen"):
            check(lib.snerf_raygen_f64(ptr(self.poses), self.n_frames, self.h, self.w, self.focal, ptr(self.lower),
                                       ptr(self.span), self.n, ptr(idx), ptr(jit), B, ptr(samples), ptr(o), ptr(d),
                                       ptr(z), current_stream()), "snerf_raygen_f64")
        truth = self.images[idx] if self.images is not None else torch.zeros((B, 3), device=dev, dtype=torch.float32)
        return [samples, o, d, z, truth]

    def random_batch(self, batch_size: int, generator=None):
        """A shuffled-DataLoader-like draw (train.py:100): uniform ray indices, one uniform jitter per ray."""
        idx = torch.randint(0, self.n_rays, (batch_size,), device=self.device, generator=generator)
        jit = torch.rand((batch_size,), device=self.device, dtype=torch.float64, generator=generator)
        return self.batch(idx, jit)
