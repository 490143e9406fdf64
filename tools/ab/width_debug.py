"""Per-parameter gradient error of RenderRayNets of several widths against torch (debug aid for the widths above 256)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch_ref as R  # noqa: E402
from smpl_nerf_amd import synthetic as syn  # noqa: E402
from smpl_nerf_amd.nets import RenderRayNet  # noqa: E402

F32 = np.float32
dev = torch.device("cuda:0")
cases = [(3, 384, ()), (3, 320, ()), (3, 352, ()), (3, 400, ()), (3, 512, ()), (1, 384, ()), (2, 384, ())]
if len(sys.argv) > 2:      # "depth:width:skip,skip;..."
    cases = [(int(a), int(b), tuple(int(v) for v in c.split(",") if v)) for a, b, c in (t.split(":") for t in sys.argv[2].split(";"))]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 777
for n_layers, width, skips in cases:
    rng = np.random.default_rng(width)
    kw = dict(n_layers=n_layers, width=width, skips=skips)
    params = syn.make_render_ray_net_params(7 + width, 30.0, 10.0, **kw)
    if n_layers > 8:      # (as tests/test_gpu_round3.py: keep the variance through 16 ReLU layers)
        for i in range(n_layers - 1):
            params[f"positional_net.{i}.weight"] = (params[f"positional_net.{i}.weight"] * F32(np.sqrt(6.0))).astype(F32)
    net = RenderRayNet(n_layers, width, 60, 24, skips=list(skips))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net = net.to(dev)
    pts, dirs = rng.uniform(-2, 2, (n, 1, 3)).astype(F32), rng.normal(size=(n, 3)).astype(F32)
    gout = rng.normal(size=(n, 4)).astype(F32)
    dn = dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)
    x_enc = torch.cat([R.posenc(torch.from_numpy(pts[:, 0]), 10, 0), R.posenc(torch.from_numpy(dn.astype(F32)), 4, 0)], -1)
    P = R.tparams(params)
    ref = R.render_ray_net(P, x_enc, n_layers=n_layers, skips=skips)
    (ref * torch.from_numpy(gout)).sum().backward()
    raw = net(x_enc.to(dev))
    (raw * torch.from_numpy(gout).to(dev)).sum().backward()
    # fp64 pre-activations of every ReLU layer (adjudication of single wrong rows: a flipped ReLU of a borderline value)
    P64 = {k: torch.from_numpy(v).double() for k, v in params.items()}
    xe = x_enc.double()
    xp, xd = xe[:, :60], xe[:, 60:]
    pre = {}
    h = xp @ P64["positions_pose_input.weight"].T + P64["positions_pose_input.bias"]
    pre["positions_pose_input"] = h
    h = torch.relu(h)
    for i in range(n_layers - 1):
        inp = torch.cat([h, xp], -1) if i in skips else h
        h = inp @ P64[f"positional_net.{i}.weight"].T + P64[f"positional_net.{i}.bias"]
        pre[f"positional_net.{i}"] = h
        h = torch.relu(h)
    o = h @ P64["additional_linear_layer.weight"].T + P64["additional_linear_layer.bias"]
    h1 = torch.cat([o, xd], -1) @ P64["directional_input.weight"].T + P64["directional_input.bias"]
    pre["directional_net.0"] = h1 @ P64["directional_net.0.weight"].T + P64["directional_net.0.bias"]
    print(f"depth {n_layers} width {width}: fwd err {float((raw.detach().cpu() - ref.detach()).abs().max()):.2e}")
    for k, p in net.named_parameters():
        g = P[k].grad.numpy()
        e = np.abs(p.grad.cpu().numpy() - g)
        tol = 5e-4 * np.abs(g) + 5e-5 * np.abs(g).max()          # the rule of tests/test_gpu_round3.py
        bad_rows = np.nonzero((e > tol).reshape(e.shape[0], -1).any(-1))[0]
        note = ""
        layer = k.rsplit(".", 1)[0]
        if len(bad_rows) == 1 and layer in pre:
            col = pre[layer][:, int(bad_rows[0])]
            i = int(col.abs().argmin())
            note = f"   <- fp64: the smallest |pre-activation| of this feature is {float(col[i]):.2e} (sample {i})"
        print(f"   {k:36s} max err {e.max():.2e} (|g| {np.abs(g).max():.2e})  bad rows {len(bad_rows)}: {bad_rows[:12].tolist()}{note}")
