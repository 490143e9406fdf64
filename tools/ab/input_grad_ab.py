"""Input gradients of the per-ray additional inputs: the r03 form (gather the [n, 256] d Y blocks out of the tile-row-major
buffer with permute().reshape(), then a library GEMM per reading layer, then a per-ray sum - nets.py:58-93 at 699df94)
against snerf_dy_contract_f32 on the stored tile-rows (csrc/contract.hip), same buffers, same result.

    python tools/ab/input_grad_ab.py [--rays 4096] [--spr 192] [--cols 69]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from smpl_nerf_amd import nets
from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--spr", type=int, default=192)
ap.add_argument("--cols", type=int, default=69)
a = ap.parse_args()
dev = torch.device("cuda:0")
net = RenderRayNet(8, 256, 60, 24, additional_input_dim=a.cols, skips=[4]).to(dev)
desc = net.desc_for_encoders(PositionalEncoder(10, 0), PositionalEncoder(4, 0), add_first=True)
n = a.rays * a.spr
layout = nets._dy_rows(desc)
rows_total = max(r + (f + 15) // 16 for r, f in layout)
dy = torch.randn(rows_total * n * 16, device=dev)
readers = [(0, net.positions_pose_input.weight, 0), (5, net.positional_net[4].weight, net.width)]


def old():
    d_pa = None
    for l, w, c0 in readers:
        r0, f = layout[l]
        t = (f + 15) // 16
        g = dy[r0 * n * 16:(r0 + t) * n * 16].view(t, n, 16).permute(1, 0, 2).reshape(n, t * 16)[:, :f]
        term = g @ w[:, c0:c0 + 60 + a.cols]
        d_pa = term if d_pa is None else d_pa + term
    return d_pa[:, :a.cols].reshape(-1, a.spr, a.cols).sum(1)


def new():
    out = torch.empty((a.rays, a.cols), device=dev)
    for k, (l, w, c0) in enumerate(readers):
        nets._contract(dy, n, layout[l][0], layout[l][1], w, c0, a.cols, a.spr, out, 0, k > 0)
    return out


with torch.no_grad():
    ra, rb = old(), new()
    err = float((ra - rb).abs().max() / ra.abs().max())
    for name, fn in (("r03: gather + rocBLAS GEMM + per-ray sum", old), ("snerf_dy_contract_f32", new)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"{name:45s} {(time.perf_counter() - t0) * 100:8.3f} ms per backward   peak extra memory "
              f"{(torch.cuda.max_memory_allocated() - dy.numel() * 4) / 2**20:8.1f} MiB")
    print(f"n = {n} samples ({a.rays} rays x {a.spr}), {a.cols} additional columns, two reading layers; max rel diff {err:.2e}")
