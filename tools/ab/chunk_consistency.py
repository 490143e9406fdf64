"""Parameter gradients of one backward over n samples against the sum of two backwards over an uneven split of the same samples:
different n take different chunk counts / kernel variants (small-call forms, one-round chunks, whole-rounds rule), so any
chunking error shows up as a mismatch.  fp32, bf16x6, f16x3; n from tens to a million.

    python tools/ab/chunk_consistency.py
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from smpl_nerf_amd.nets import RenderRayNet

dev = torch.device("cuda:0")
torch.manual_seed(3)
bad = 0
for prec in ("fp32", "bf16x6", "f16x3"):
    net = RenderRayNet(8, 256, 60, 24, skips=[4]).to(dev).train()
    net.precision = prec
    for n in (37, 1000, 4097, 16383, 16385, 20000, 65537, 131072, 150001, 262144, 300007, 524288, 1000003):
        x = torch.randn(n, 84, device=dev)
        w = torch.randn(n, 4, device=dev)

        def grads(lo, hi):
            for p in net.parameters():
                p.grad = None
            (net(x[lo:hi]) * w[lo:hi]).sum().backward()
            return [p.grad.detach().clone() for p in net.parameters()]

        full = grads(0, n)
        h = max(1, (n * 3) // 7)
        a, b = grads(0, h), grads(h, n)
        worst = 0.0
        for f, ga, gb in zip(full, a, b):
            worst = max(worst, float((f - (ga + gb)).norm()) / max(float(f.norm()), 1e-20))
        tol = 2e-5 if prec == "fp32" else 2e-3
        ok = worst <= tol and all(bool(torch.isfinite(f).all()) for f in full)
        bad += not ok
        print(("ok  " if ok else "BAD ") + f"{prec} n {n}: worst relative difference {worst:.2e}", flush=True)
# inference: a sample's output does not depend on how many samples the call has or where in the call it sits - bit for bit
# (persistent workgroups, wrap-around of the weight stream, ragged last tiles, the small-call tile size)
for prec in ("fp32", "bf16x6", "f16x3", "bf16x3"):
    net = RenderRayNet(8, 256, 60, 24, skips=[4]).to(dev).eval()
    net.precision = prec
    x = torch.randn(300007, 84, device=dev)
    with torch.no_grad():
        full = net(x)
        for n, h in ((37, 5), (4097, 129), (16385, 16000), (65537, 1), (300007, 131072)):
            parts = torch.cat([net(x[:h].contiguous()), net(x[h:n].contiguous())])
            ok = torch.equal(parts, full[:n])
            bad += not ok
            print(("ok  " if ok else "BAD ") + f"inference {prec} n {n} split at {h}: {'bit-identical' if ok else 'DIFFERS'}", flush=True)
# the warp net (linear1 -> relu -> linear2 on [encoded position | encoded pose] rows): its own dgrad / wgrad chunking
from smpl_nerf_amd.nets import WarpFieldNet
for width in (256, 128, 100):
    mw = WarpFieldNet(8, width, 60, 40).to(dev).train()
    for n in (5, 1000, 16385, 65537, 200003, 786432):
        x = torch.randn(n, 100, device=dev)
        w = torch.randn(n, 3, device=dev)

        def wgrads(lo, hi):
            for p in mw.parameters():
                p.grad = None
            xi = x[lo:hi].clone().requires_grad_(True)
            (mw(xi) * w[lo:hi]).sum().backward()
            return [p.grad.detach().clone() for p in mw.parameters()], xi.grad

        (full, gx), h = wgrads(0, n), max(1, (n * 3) // 7)
        (a, gxa), (b, gxb) = wgrads(0, h), wgrads(h, n)
        worst = max(float((f - (ga + gb)).norm()) / max(float(f.norm()), 1e-20) for f, ga, gb in zip(full, a, b))
        worst = max(worst, float((gx - torch.cat([gxa, gxb])).norm()) / max(float(gx.norm()), 1e-20))
        ok = worst <= 2e-5
        bad += not ok
        print(("ok  " if ok else "BAD ") + f"warp width {width} n {n}: worst relative difference {worst:.2e}", flush=True)
print("bad:", bad)
