"""Layer-by-layer path (widths above the fused kernels): inference and forward + backward of one RenderRayNet, with the fraction of
the fp32 MFMA peak on the algorithmic FLOPs.   python tools/ab/layered_timing.py [width] [n_infer] [n_train]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder

width = int(sys.argv[1]) if len(sys.argv) > 1 else 768
n_inf = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
n_tr = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = RenderRayNet(8, width, 60, 24, skips=[4]).to(dev)
flop = 2 * sum(p.numel() for k, p in net.named_parameters() if k.endswith("weight"))
pe, de = PositionalEncoder(10, 0), PositionalEncoder(4, 0)


def run(n, train, steps=3):
    x = torch.rand(n, 3, device=dev) * 2 - 1
    d = torch.randn(n // 64, 3, device=dev)
    g = torch.randn(n, 4, device=dev)

    def once():
        if train:
            for p in net.parameters():
                p.grad = None
            out = net.forward_fused(x, d, 64, pe, de)
            (out * g).sum().backward()
        else:
            with torch.no_grad():
                net.forward_fused(x, d, 64, pe, de)
    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    f = flop * n * (3 if train else 1)
    print(f"width {width} n {n} {'fwd+bwd' if train else 'inference'}: {dt * 1e3:.2f} ms, {f / dt / 1e12:.1f} TFLOP/s = {f / dt / 1e12 / 157.3:.3f} of the fp32 MFMA peak")


run(n_inf, False)
run(n_tr, True)
