"""Dev: per-wave error pattern of the split-bf16 kernel vs the fp32 kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smpl_nerf_amd import synthetic as syn
from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder
dev = torch.device("cuda:0")
p = syn.make_scene_nets(101)[1]
m = RenderRayNet(8, 256, 60, 24, skips=[4]); m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}); m = m.to(dev)
rng = np.random.default_rng(0)
B, Ns = 64, 8   # 512 samples = 4 workgroups x 8 waves x 16
pts = torch.from_numpy(rng.uniform(-2, 2, (B, Ns, 3)).astype(np.float32)).to(dev)
d = torch.from_numpy(rng.normal(size=(B, 3)).astype(np.float32)).to(dev)
enc = (PositionalEncoder(10, 0), PositionalEncoder(4, 0))
with torch.no_grad():
    a = m.forward_fused(pts, d, Ns, *enc).cpu().numpy().reshape(-1, 4)
    for prec in ("bf16x3", "bf16x6", "bf16x3"):
        m.precision = prec
        b = m.forward_fused(pts, d, Ns, *enc).cpu().numpy().reshape(-1, 4)
        print(prec, "identical to fp32 output:", bool((a == b).all()), "max|b|", np.abs(b).max())
        err = np.abs(a - b).max(1).reshape(-1, 16).max(1)   # per wave
        print(prec, "scale", np.abs(a).max(), "per-wave max err:")
        print(np.array2string(err.reshape(-1, 8), precision=2, max_line_width=200))
