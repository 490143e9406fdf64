"""Pin tests/torch_ref.py (the gradient reference of the GPU tests) against gradients captured from the
reference under autograd (tests/golden/g7_grads.npz).  CPU only."""
import numpy as np
import pytest
import torch

import torch_ref as R
from smpl_nerf_amd import synthetic as syn
from conftest import load_golden


def close(a, b, rtol, atol):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("N", [1, 2, 64, 192, 100])
@pytest.mark.parametrize("wb", [0, 1])
@pytest.mark.parametrize("mode", ["ray", "smp"])
def test_raw2outputs_backward(N, wb, mode):
    if N == 1 and mode == "smp":
        pytest.skip("N==1 ignores directions")
    g3, g7 = load_golden("g3_raw2outputs.npz"), load_golden("g7_grads.npz")
    B = g3[f"raw_N{N}"].shape[0]
    raw = torch.from_numpy(g3[f"raw_N{N}"]).requires_grad_(True)
    d = torch.from_numpy(g3[f"dray_N{N}"])[:, None, :].expand(B, N, 3) if mode == "ray" else torch.from_numpy(g3[f"dsmp_N{N}"])
    rgb, _, _ = R.raw2outputs(raw, torch.from_numpy(g3[f"z_N{N}"]), d, wb)
    (rgb * torch.from_numpy(g7[f"c_gout_N{N}"])).sum().backward()
    close(raw.grad.numpy(), g7[f"c_draw_N{N}_wb{wb}_{mode}"], 1e-4, 1e-6)


def test_mlp_backward_small_net_all_params():
    g2, g7 = load_golden("g2_mlp.npz"), load_golden("g7_grads.npz")
    kw = dict(n_layers=4, width=128, skips=(1,))
    P = R.tparams(syn.make_render_ray_net_params(13, 30.0, 10.0, **kw))
    out = R.render_ray_net(P, torch.from_numpy(g2["inputs"]), n_layers=4, skips=(1,))
    (out * torch.from_numpy(g7["m_gout"])).sum().backward()
    for k, p in P.items():
        ref = g7[f"m_d4w128/{k}"]
        close(p.grad.numpy(), ref, 1e-4, 1e-5 * np.abs(ref).max())


@pytest.mark.parametrize("tag", ["skip4", "noskip", "scene"])
def test_mlp_backward_digests(tag):
    g2, g7 = load_golden("g2_mlp.npz"), load_golden("g7_grads.npz")
    params = {"skip4": lambda: syn.make_render_ray_net_params(11, 30.0, 10.0, skips=(4,)),
              "noskip": lambda: syn.make_render_ray_net_params(12, 30.0, 10.0, skips=()),
              "scene": lambda: syn.make_scene_nets(101)[1]}[tag]()
    P = R.tparams(params)
    out = R.render_ray_net(P, torch.from_numpy(g2["inputs"]), skips=() if tag == "noskip" else (4,))
    (out * torch.from_numpy(g7["m_gout"])).sum().backward()
    for k, p in P.items():
        ref = g7[f"m_{tag}/{k}"]
        close(R.digest(p.grad), ref, 2e-4, 2e-5 * max(np.abs(ref[2:]).max(), ref[1] / np.sqrt(p.numel())))


def test_pipeline_first_step_gradients():
    g7 = load_golden("g7_grads.npz")
    pc, pf = syn.make_scene_nets(101)
    Pc, Pf = R.tparams(pc), R.tparams(pf)
    data = syn.frame_batch(128, 128, seed=7)
    batch = [a[g7["t_sub"]] for a in data]
    u = torch.linspace(0., 1., 128).numpy()
    rgb, rgb_f, _, _ = R.nerf_pipeline(Pc, Pf, batch, u=u)
    gt = torch.from_numpy(batch[4])
    loss = torch.nn.functional.mse_loss(rgb, gt) + torch.nn.functional.mse_loss(rgb_f, gt)
    loss.backward()
    assert abs(loss.item() - g7["t_losses"][0]) <= 1e-6
    close(rgb.detach().numpy(), g7["t_rgb0"], 0, 1e-5)
    close(rgb_f.detach().numpy(), g7["t_rgb_fine0"], 0, 1e-4)
    for name, P in (("coarse", Pc), ("fine", Pf)):
        for k, p in P.items():
            ref = g7[f"t_grad0/{name}.{k}"]
            scale = max(np.abs(ref[2:]).max(), ref[1] / np.sqrt(p.numel()), 1e-12)
            close(R.digest(p.grad), ref, 5e-3, 2e-3 * scale)
