"""Differentiable fp32 torch (CPU) restatement of the floating-point part of the path, used ONLY by
tests as the gradient reference for the HIP backward kernels (the numpy oracle has no autograd).
It mirrors oracle/nerf_oracle.py op for op; tests/test_grad_golden.py pins it against gradients
captured from the reference itself (tests/golden/make_golden_grad.py)."""
import numpy as np
import torch

from oracle import nerf_oracle as O


def posenc(x, L, ident):
    outs = [x] if ident else []
    for k in range(L):
        f = float(2.0 ** k)
        outs += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(outs, -1)


def render_ray_net(P, x, n_layers=8, positions_dim=60, directions_dim=24, additional_input_dim=0, skips=(4,),
                   use_directional_input=1):
    lin = lambda v, n: torch.nn.functional.linear(v, P[n + ".weight"], P[n + ".bias"])
    pin = positions_dim + additional_input_dim
    pp, dd = x[..., :pin], x[..., x.shape[-1] - directions_dim:]
    o = torch.relu(lin(pp, "positions_pose_input"))
    for i in range(n_layers - 1):
        o = torch.relu(lin(torch.cat([o, pp], -1) if i in skips else o, f"positional_net.{i}"))
    o = lin(o, "additional_linear_layer")
    sigma = lin(o, "sigma_out_layer")
    o = lin(torch.cat([o, dd], -1) if use_directional_input else o, "directional_input")
    o = torch.relu(lin(o, "directional_net.0"))
    return torch.cat([lin(o, "rgb_out_layer"), sigma], -1)


def raw2outputs(raw, z, dirs, wb, noise=None):
    dists = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1) * torch.norm(dirs, dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    if z.shape[-1] == 1:
        return rgb.view(raw.shape[0], 3), torch.ones(raw.shape[0], 1), torch.ones(raw.shape[0], 1)
    sig = raw[..., 3] if noise is None else raw[..., 3] + noise
    a = 1. - torch.exp(-torch.relu(sig) * dists)
    om = 1. - a + 1e-10
    T = torch.cumprod(torch.cat([torch.ones_like(om[..., :1]), om[..., :-1]], -1), -1)
    w = a * T
    out = torch.sum(w[..., None] * rgb, -2)
    if wb:
        out = out + (1. - torch.sum(w, -1)[..., None])
    return out, w, a


def tparams(params, requires_grad=True):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(requires_grad) for k, v in params.items()}


def nerf_pipeline(Pc, Pf, data, wb=0, run_fine=1, nf=128, u=None, net_kw=None):
    """models/nerf_pipeline.py:14-67 with the hierarchical samples taken from the numpy oracle (they
    are detached in the reference, utils.py:260)."""
    net_kw = net_kw or {}
    x, o, d, z = [torch.from_numpy(np.ascontiguousarray(a)) for a in data[:4]]
    B, Nc = z.shape
    dirs = d[:, None, :].expand(B, Nc, 3)
    dn = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    denc = posenc(dn, 4, 0)
    inp = torch.cat([posenc(x, 10, 0).view(B * Nc, -1), denc.reshape(B * Nc, -1)], -1)
    raw = render_ray_net(Pc, inp, **net_kw).view(B, Nc, 4)
    rgb, w, a = raw2outputs(raw, z, dirs, wb)
    if not run_fine:
        return rgb, rgb, x, a
    zf, pts = O.fine_sampling(data[1], data[2], data[3], w.detach().numpy(), nf, u=u)
    zf, pts = torch.from_numpy(zf), torch.from_numpy(pts)
    N = zf.shape[1]
    inp_f = torch.cat([posenc(pts, 10, 0).view(B * N, -1), denc[:, :1, :].expand(B, N, 24).reshape(B * N, -1)], -1)
    raw_f = render_ray_net(Pf, inp_f, **net_kw).view(B, N, 4)
    rgb_f, _, a_f = raw2outputs(raw_f, zf, d[:, None, :].expand(B, N, 3), wb)
    return rgb, rgb_f, pts, a_f


def digest(t):
    a = t.detach().cpu().numpy().astype(np.float64).reshape(-1)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum())], a[:16]])
