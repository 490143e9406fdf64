"""ms per 64-ray training step in blocks of 50 steps (smpl_nerf and nerf one-call trainers): is the step time stable?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from smpl_nerf_amd.trainer import DataParallelTrainer
dev = torch.device("cuda:0")
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for wl in ("smpl_nerf", "nerf"):
    data = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in bench.frame_inputs(wl, 128, 0)]
    pipe, _, models = bench.build_pipeline(dev, "fp32", wl)
    for m in models:
        m.train()
    tr = DataParallelTrainer(pipe, models, lr=bench.TRAIN_LR)
    batch = [t[:rays].contiguous() for t in data]
    for _ in range(5):
        tr.step(batch)
    torch.cuda.synchronize()
    out = []
    for blk in range(16):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(50):
            tr.step(batch)
        e1.record()
        torch.cuda.synchronize()
        out.append((e0.elapsed_time(e1) / 50, (time.perf_counter() - t0) / 50 * 1e3))
        if blk == 7:
            time.sleep(1.0)          # an idle second: does the next block run at idle clocks?
    print(wl, rays, "rays: GPU ms/step per block", " ".join(f"{a:.2f}" for a, _ in out), flush=True)
    print(wl, rays, "rays: wall ms/step per block", " ".join(f"{b:.2f}" for _, b in out), flush=True)
