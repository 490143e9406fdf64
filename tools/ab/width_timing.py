import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from smpl_nerf_amd import synthetic as syn
from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B, Ns = 16384, 128
pts = torch.from_numpy(rng.uniform(-2, 2, (B, Ns, 3)).astype(np.float32)).to(dev)
dirs = torch.from_numpy(rng.normal(size=(B, 3)).astype(np.float32)).to(dev)
pe, de = PositionalEncoder(10, 0), PositionalEncoder(4, 0)
for width in (256, 200, 128, 100, 64, 30):
    params = syn.make_render_ray_net_params(1, 30.0, 10.0, n_layers=8, width=width, skips=(4,))
    net = RenderRayNet(8, width, 60, 24, skips=[4]); net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}); net = net.to(dev)
    with torch.no_grad():
        for _ in range(2): net.forward_fused(pts, dirs, Ns, pe, de)
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(5): net.forward_fused(pts, dirs, Ns, pe, de)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    macs = sum(v.size for k, v in params.items() if k.endswith("weight"))
    print(f"width {width:4d}: {ms:7.3f} ms per {B*Ns} samples = {B*Ns/ms*1e3:.3e} samples/s; algorithmic {2*macs*B*Ns/ms*1e-9:.1f} TFLOP/s")
