"""Training-step operating points: ms per step, host enqueue time, peak memory for the one-call step (and the autograd form).

    python tools/ab/train_points.py [--rays 64,800,2048,4096] [--chunks 0,1024,2048] [--autograd] [--precision fp32]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from smpl_nerf_amd.trainer import DataParallelTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--rays", default="64,800,2048,4096")
ap.add_argument("--chunks", default="2048")
ap.add_argument("--autograd", action="store_true")
ap.add_argument("--precision", default="fp32")
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
data = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in bench.frame_inputs("nerf", 128, 0)]
FLOP = 3 * 2 * 607872 * 256
for rays in [min(int(v), data[0].shape[0]) for v in a.rays.split(",")]:      # (at most the frame's 16 384 rays)
    for mode in (["one_call"] + (["autograd"] if a.autograd else [])):
        for chunk in ([int(v) for v in a.chunks.split(",")] if mode == "one_call" else [0]):
            pipe, _, models = bench.build_pipeline(dev, a.precision, "nerf")
            for m in models:
                m.train()
                for p in m.parameters():
                    p.requires_grad_(True)
            tr = DataParallelTrainer(pipe, models, lr=bench.TRAIN_LR, one_call=None if mode == "one_call" else False)
            tr.rays_per_chunk = chunk
            g = torch.Generator(device="cpu").manual_seed(1234)
            batches = [[t[torch.randperm(data[0].shape[0], generator=g)[:rays].to(dev)].contiguous() for t in data] for _ in range(4)]
            for i in range(3):
                tr.step(batches[i % 4])
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(dev)
            t0 = time.perf_counter()
            for i in range(a.steps):
                loss = tr.step(batches[i % 4])
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ms = dt / a.steps * 1e3
            print(f"rays {rays:6d} {mode:9s} chunk {chunk:5d}: {ms:8.3f} ms/step  host {th / a.steps * 1e3:6.3f} ms  "
                  f"{rays * 256 / ms * 1e3:.3e} ray-samples/s  frac {FLOP * rays / (ms * 1e-3) / 157.3e12:.3f}  "
                  f"peak {torch.cuda.max_memory_allocated(dev) / 2**30:6.2f} GiB ({torch.cuda.max_memory_allocated(dev) / rays / 2**20:.2f} MiB/ray)  loss {float(loss):.5f}",
                  flush=True)
            del tr, pipe, models
            torch.cuda.empty_cache()
