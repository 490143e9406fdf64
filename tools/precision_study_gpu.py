"""GPU measurement: error of every precision mode of the fused encode+MLP kernel against a float64 evaluation, over the
net shapes / batch shapes of tests/test_gpu_parity.py::test_split_bf16_stress_against_fp32_kernel plus a full 128x128
frame of the bench scene.  Prints max and RMS errors and their ratios to the exact-fp32 kernel's own error (the figures
DESIGN.md section 3.1b quotes and the bound the tests hold bf16x6 to)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from smpl_nerf_amd import synthetic as syn
from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder
from test_gpu_parity import _mlp_ref64

F32 = np.float32
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
MODES = ("fp32", "bf16x6", "f16x3", "bf16x3")


def net_of(params, **kw):
    m = RenderRayNet(kw.get("n_layers", 8), 256, 60, 24, kw.get("additional_input_dim", 0), skips=list(kw.get("skips", (4,))),
                     use_directional_input=kw.get("use_directional_input", 1))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return m.to(dev)


def main():
    rng = np.random.default_rng(2024)
    enc = (PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    cases = [dict(n_layers=8, skips=(4,)), dict(n_layers=8, skips=()), dict(n_layers=5, skips=(2,)), dict(n_layers=3, skips=(1,)),
             dict(n_layers=8, skips=(4,), additional_input_dim=69), dict(n_layers=8, skips=(3, 6), additional_input_dim=5)]
    sq = {m: 0.0 for m in MODES}
    cnt = 0
    rows = []
    for ci, kw in enumerate(cases):
        add_dim = kw.get("additional_input_dim", 0)
        add_first = bool(add_dim and ci % 2)
        params = syn.make_scene_net_params(500 + ci, add_first=add_first, **kw)
        net = net_of(params, **kw)
        for B, Ns, per_sample in ((37, 5, 0), (129, 64, 1), (700, 192, 0), (4096, 64, 0)):
            pts = rng.uniform(-2.5, 2.5, (B, Ns, 3)).astype(F32)
            dirs = rng.normal(size=(B, Ns, 3) if per_sample else (B, 1, 3)).astype(F32)
            add = rng.uniform(-1, 1, (B, 1, add_dim)).astype(F32) if add_dim else None
            kwf = dict(additional=T(add[:, 0]), add_first=add_first) if add_dim else {}
            ref = _mlp_ref64(params, pts, np.broadcast_to(dirs, (B, Ns, 3)),
                             None if add is None else np.broadcast_to(add, (B, Ns, add_dim)), add_first, **kw)
            scale = float(np.abs(ref).max())
            row = {"case": ci, "B": B, "Ns": Ns}
            with torch.no_grad():
                for m in MODES:
                    net.precision = m
                    out = net.forward_fused(T(pts), T(dirs.reshape(-1, 3)), Ns, *enc, **kwf).cpu().numpy().reshape(B, Ns, 4)
                    e = (out.astype(np.float64) - ref) / scale
                    row[m + "_max"] = float(np.abs(e).max())
                    row[m + "_rms"] = float(np.sqrt((e * e).mean()))
                    sq[m] += float((e * e).sum())
            cnt += ref.size
            rows.append(row)
            print(json.dumps(row))
    agg = {m: float(np.sqrt(sq[m] / cnt)) for m in MODES}
    print("aggregate rms (relative to each case's largest |raw|):", json.dumps(agg))
    print("ratios to fp32:", json.dumps({m: agg[m] / agg["fp32"] for m in MODES}))
    print("per-case max ratio bf16x6/fp32:", max(r["bf16x6_max"] / r["fp32_max"] for r in rows),
          " rms ratio:", max(r["bf16x6_rms"] / r["fp32_rms"] for r in rows))
    print("per-case max ratio f16x3/fp32:", max(r["f16x3_max"] / r["fp32_max"] for r in rows),
          " rms ratio:", max(r["f16x3_rms"] / r["fp32_rms"] for r in rows))


if __name__ == "__main__":
    main()
