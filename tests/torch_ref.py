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


RELU_LAYERS = lambda n_layers: ["positions_pose_input"] + [f"positional_net.{i}" for i in range(n_layers - 1)] + ["directional_net.0"]


def layer_order(n_layers):
    """nn.Linear names of models/render_ray_net.py in forward order."""
    return (["positions_pose_input"] + [f"positional_net.{i}" for i in range(n_layers - 1)] +
            ["additional_linear_layer", "sigma_out_layer", "directional_input", "directional_net.0", "rgb_out_layer"])


def render_ray_net_preacts(P, x, n_layers=8, positions_dim=60, directions_dim=24, additional_input_dim=0, skips=(4,),
                           use_directional_input=1):
    """The pre-activations of every ReLU layer of render_ray_net() in the dtype of P / x (float64 for adjudication)."""
    lin = lambda v, n: torch.nn.functional.linear(v, P[n + ".weight"], P[n + ".bias"])
    pin = positions_dim + additional_input_dim
    pp, dd = x[..., :pin], x[..., x.shape[-1] - directions_dim:]
    pre = {"positions_pose_input": lin(pp, "positions_pose_input")}
    o = torch.relu(pre["positions_pose_input"])
    for i in range(n_layers - 1):
        pre[f"positional_net.{i}"] = lin(torch.cat([o, pp], -1) if i in skips else o, f"positional_net.{i}")
        o = torch.relu(pre[f"positional_net.{i}"])
    o = lin(o, "additional_linear_layer")
    o = lin(torch.cat([o, dd], -1) if use_directional_input else o, "directional_input")
    pre["directional_net.0"] = lin(o, "directional_net.0")
    return pre


def check_grads_or_one_relu_kink(got, want, params64, x64, n_layers, skips, rtol=5e-4, atol_rel=5e-5, kink=1e-6, below=5e-2):
    """Every parameter gradient `got[name]` against `want[name]` (numpy) at |err| <= rtol |g| + atol_rel max|g| - or ONE ReLU kink.

    Two fp32 summation orders (MFMA k-blocks against the CPU's dot products) need not agree on the sign of a pre-activation that is
    zero to round-off; the ReLU mask bit of that one (sample, feature) then differs and the gradients differ by that sample's
    contribution: in ONE row of that layer's weight gradient (one element of its bias gradient), nowhere in the layers towards the
    output, and by a bounded amount in every layer towards the input.  A miss passes only with exactly that
    signature, the feature's smallest |pre-activation| recomputed in float64 below `kink`; anything else fails.  Returns None or a
    description of the adjudicated kink."""
    viol = {}
    for k, g in want.items():
        a = got[k]
        bad = np.abs(a - g) > rtol * np.abs(g) + atol_rel * np.abs(g).max()
        if bad.any():
            viol[k] = bad
    if not viol:
        return None
    order = layer_order(n_layers)
    layers_bad = [l for l in order if l + ".weight" in viol or l + ".bias" in viol]
    top = layers_bad[-1]
    if "sigma_out_layer" in layers_bad:
        # a sibling of the colour branch: it reads the trunk's output, which a kink leaves unchanged - unless the kink sits in the
        # trunk, where the head's own gradient (d sigma x o) still does not change
        raise AssertionError(f"gradient mismatch in sigma_out_layer (no ReLU kink explains it); layers off: {layers_bad}")
    assert top in RELU_LAYERS(n_layers), f"gradient mismatch with its topmost layer {top} not a ReLU layer; layers off: {layers_bad}"
    wbad = viol.get(top + ".weight")
    rows = sorted(set(np.nonzero(wbad)[0].tolist())) if wbad is not None else []
    brows = sorted(np.nonzero(viol[top + ".bias"])[0].tolist()) if top + ".bias" in viol else []
    rows_all = sorted(set(rows) | set(brows))
    assert len(rows_all) == 1, f"gradient mismatch in {len(rows_all)} rows of {top} (a single kink touches one): {rows_all[:8]}"
    f = rows_all[0]
    pre = render_ray_net_preacts(params64, x64, n_layers=n_layers, skips=skips)[top][:, f]
    smallest = float(pre.abs().min())
    assert smallest < kink, f"row {f} of {top} is off but its smallest |pre-activation| in float64 is {smallest:.3e} (not a kink)"
    # towards the input: the one sample's contribution, bounded
    for l in layers_bad[:-1]:
        for suffix in (".weight", ".bias"):
            k = l + suffix
            err = np.linalg.norm(got[k] - want[k]) / max(np.linalg.norm(want[k]), 1e-30)
            assert err <= below, f"{k}: relative error {err:.2e} below the kink of {top} row {f} exceeds {below}"
    return f"ReLU kink: {top} row {f}, float64 pre-activation {float(pre[pre.abs().argmin()]):.3e} at sample {int(pre.abs().argmin())}"


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
