"""bench.train_section's own loop at 64 rays (smpl_nerf): per-step GPU times, to find where an occasional 8 ms/step average comes from."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "smpl_nerf"
data = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in bench.frame_inputs(wl, 128, 0)]
pipe, _, _ = bench.build_pipeline(dev, "fp32", wl)
with torch.no_grad():
    for _ in range(3):
        pipe(data)          # the render section of the bench comes first
torch.cuda.synchronize()
for rep in range(4):
    r = bench.train_section("fp32", wl, data, 64, 50, 1, 0, dev)
    print(rep, "ms/step", round(r["ms_per_step"], 3), "kernels", round(r["mlp_kernels_ms_per_step"], 3), "host", round(r["host_enqueue_ms_per_step"], 3), flush=True)
# per-step events with the same construction
from smpl_nerf_amd.trainer import DataParallelTrainer
pipe, _, models = bench.build_pipeline(dev, "fp32", wl)
for m in models:
    m.train()
tr = DataParallelTrainer(pipe, models, lr=bench.TRAIN_LR)
g = torch.Generator(device="cpu").manual_seed(1234)
batches = []
for _ in range(4):
    idx = torch.randperm(data[0].shape[0], generator=g)[:64].to(dev)
    batches.append([t[idx].contiguous() for t in data])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
for i in range(2):
    tr.step(batches[i % 4])
torch.cuda.synchronize()
ev[0].record()
for i in range(60):
    tr.step(batches[i % 4])
    ev[i + 1].record()
torch.cuda.synchronize()
print("per step:", " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.2f}" for i in range(60)))
