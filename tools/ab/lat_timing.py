"""Per-launch time of the fused forward at small sample counts: SNERF_LAT=0/1 python tools/ab/lat_timing.py (HIP events)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smpl_nerf_amd import synthetic as syn  # noqa: E402
from smpl_nerf_amd.nets import RenderRayNet  # noqa: E402
from smpl_nerf_amd.ops import PositionalEncoder  # noqa: E402

dev = torch.device("cuda:0")
pc, _ = syn.make_scene_nets(101)
net = RenderRayNet(8, 256, 60, 24, skips=[4])
net.load_state_dict({k: torch.from_numpy(v) for k, v in pc.items()})
net = net.to(dev)
pe, de = PositionalEncoder(10, 0), PositionalEncoder(4, 0)
rng = np.random.default_rng(0)
train = len(sys.argv) > 1 and sys.argv[1] in ("train", "bwd")
bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"      # forward + backward through autograd (run under rocprofv3 for the kernels' own times)
sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (4096, 12288, 16384, 32768, 51200, 65536, 153600, 262144, 524288)
for n in sizes:
    rays = n // 64
    pts = torch.from_numpy(rng.uniform(-2, 2, (rays, 64, 3)).astype(np.float32)).to(dev)
    d = torch.from_numpy(rng.normal(size=(rays, 3)).astype(np.float32)).to(dev)
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        for _ in range(3):
            net.forward_fused(pts, d, 64, pe, de)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            raw = net.forward_fused(pts, d, 64, pe, de)
            if bwd:
                raw.backward(torch.ones_like(raw))
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"n={n:7d} {'train' if train else 'infer'} LAT={os.environ.get('SNERF_LAT', '1')} "
          f"MAXT={os.environ.get('SNERF_LAT_MAX_TILES_PER_CU', '8')}: {ms * 1e3:8.1f} us  {n * 1215744 / ms / 1e9 / 157.3:.3f} of peak")
