"""Where the host time of a one-call training step goes: cProfile over a few steps (the GPU is far behind; only enqueue cost counts).

    python tools/ab/host_profile_train.py [rays] [steps]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from smpl_nerf_amd.trainer import DataParallelTrainer

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
data = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in bench.frame_inputs("nerf", 128, 0)]
pipe, _, models = bench.build_pipeline(dev, "fp32", "nerf")
for m in models:
    m.train()
tr = DataParallelTrainer(pipe, models, lr=bench.TRAIN_LR)
batch = [t[:rays].contiguous() for t in data]
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(batch)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"rays {rays}: host {th / steps * 1e3:.3f} ms per step (plain)")
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(steps):
    tr.step(batch)
pr.disable()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"rays {rays}: host {th / steps * 1e3:.3f} ms per step (under cProfile)")
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
