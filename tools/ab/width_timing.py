"""--netwidth above 256: per-launch time and roofline fraction of the fused forward (inference), and of the one-call training step,
for RenderRayNets of a given width against the fp32 MFMA peak (HIP events around the launches; algorithmic FLOPs of the UNPADDED
net: 2 x parameters of its weight matrices per sample).   python tools/ab/width_timing.py [widths] [samples]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smpl_nerf_amd.nets import RenderRayNet  # noqa: E402
from smpl_nerf_amd.ops import PositionalEncoder  # noqa: E402
from smpl_nerf_amd.pipelines import NerfPipeline, PipelineArgs  # noqa: E402
from smpl_nerf_amd.trainer import DataParallelTrainer  # noqa: E402

PEAK = 157.3
dev = torch.device("cuda:0")
widths = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256, 320, 384, 512]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128 * 128 * 64
pe, de = PositionalEncoder(10, 0), PositionalEncoder(4, 0)
rng = np.random.default_rng(0)
for width in widths:
    torch.manual_seed(width)
    nets = [RenderRayNet(8, width, 60, 24, skips=[4]).to(dev) for _ in range(2)]
    flop = 2 * sum(p.numel() for k, p in nets[0].named_parameters() if k.endswith("weight"))      # per sample, forward
    rays = n // 64
    pts = torch.from_numpy(rng.uniform(-2, 2, (rays, 64, 3)).astype(np.float32)).to(dev)
    d = torch.from_numpy(rng.normal(size=(rays, 3)).astype(np.float32)).to(dev)
    with torch.no_grad():
        for _ in range(3):
            nets[0].forward_fused(pts, d, 64, pe, de)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            nets[0].forward_fused(pts, d, 64, pe, de)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = n * flop / ms / 1e9
    padded = (width + 63) // 64 * 64 if width > 256 else 256 if width > 128 else 128
    print(f"width {width:3d} (kernel width {padded}) inference {n} samples: {ms:8.3f} ms/launch  {tf:6.1f} TFLOP/s algorithmic = "
          f"{tf / PEAK:.3f} of the fp32 MFMA peak", flush=True)
    if len(sys.argv) > 3 and sys.argv[3] == "infer":
        continue
    # the one-call training step: 4096 rays x (64 + 192) samples, fwd + dgrad + wgrad of both nets
    B = 4096
    for m in nets:
        m.train()
    pipe = NerfPipeline(nets[0], nets[1], PipelineArgs(), pe, de)
    tr = DataParallelTrainer(pipe, nets, lr=5e-4)
    o = rng.normal(0, 0.2, (B, 3)).astype(np.float32) + np.array([0, 0, 2.4], np.float32)
    dd = rng.normal(0, 0.3, (B, 3)).astype(np.float32) + np.array([0, 0, -1], np.float32)
    z = np.sort(rng.uniform(1.0, 4.0, (B, 64)).astype(np.float32), -1)
    smp = (o[:, None, :] + dd[:, None, :] * z[:, :, None]).astype(np.float32)
    gt = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    batch = [torch.from_numpy(a).to(dev) for a in (smp, o, dd, z, gt)]
    for _ in range(2):
        tr.step(batch)
    torch.cuda.synchronize()
    reps = 5
    e0.record()
    for _ in range(reps):
        tr.step(batch)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 3 * flop * B * 256 / ms / 1e9
    print(f"width {width:3d} training step {B} rays (one call: {tr._one_call_state() is not None}): {ms:8.3f} ms/step  {tf:6.1f} TFLOP/s "
          f"algorithmic = {tf / PEAK:.3f} of the fp32 MFMA peak (whole step incl. compositing, sampler, loss, Adam)", flush=True)
    del tr, pipe, nets
    torch.cuda.empty_cache()
