"""A loop of one-call training steps for traces: python tools/ab/train_loop.py <rays> [workload] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from smpl_nerf_amd.trainer import DataParallelTrainer

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
workload = sys.argv[2] if len(sys.argv) > 2 else "nerf"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
data = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in bench.frame_inputs(workload, 128, 0)]
pipe, _, models = bench.build_pipeline(dev, "fp32", workload)
for m in models:
    m.train()
tr = DataParallelTrainer(pipe, models, lr=bench.TRAIN_LR)
batch = [t[:rays].contiguous() for t in data]
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(batch)
th = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{workload} rays {rays}: {dt / steps * 1e3:.4f} ms per step (host {th / steps * 1e3:.4f})")
