"""Where the host time of a small inference call goes (inference.py:251-252: `pipeline(data)` under no_grad): plain timing of the
dispatching forward() against the five-launch form, then cProfile over the single-call form.

    python tools/ab/host_profile_render.py [rays] [steps] [workload]
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

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
workload = sys.argv[3] if len(sys.argv) > 3 else "nerf"
dev = torch.device("cuda:0")
data = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in bench.frame_inputs(workload, 128, 0)]
pipe, _, models = bench.build_pipeline(dev, "fp32", workload)
sub = [t[:rays].contiguous() for t in data]


def loop(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn(sub)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th / n * 1e3, (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for fn in (pipe, pipe._forward_calls):
        loop(fn, 5)
    for name, fn in (("forward (dispatches to the single call)", pipe), ("five-launch form", pipe._forward_calls), ("forward again", pipe)):
        h, t = loop(fn, steps)
        print(f"{workload} rays {rays}: {name}: host {h:.4f} ms, per step {t:.4f} ms")
    a, b = pipe(sub), pipe._forward_calls(sub)
    print("single call == five launches bit for bit:", all(torch.equal(x, y) for x, y in zip(a, b)))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        pipe(sub)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
