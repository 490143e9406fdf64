"""Round-3 GPU parity: the API surface VERDICT r02 listed as missing, against vectors the reference itself produced
(tests/golden/make_golden_r3.py): SmplNerfPipeline with human_pose_encoding = 0, WarpFieldNet.forward(x) under autograd,
differentiable PositionalEncoder.encode / raw2outputs, searchsorted for every scalar type the reference dispatches."""
import numpy as np
import pytest
import torch

import torch_ref as R
from oracle import nerf_oracle as O
from smpl_nerf_amd import synthetic as syn
from conftest import load_golden

pytestmark = pytest.mark.gpu
F32 = np.float32
PRECISIONS = ["fp32", "bf16x6", "f16x3"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def close(a, b, rtol, atol):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def _net(dev, params, precision="fp32"):
    from smpl_nerf_amd.nets import RenderRayNet
    net = RenderRayNet(8, 256, 60, 24, skips=[4])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net.precision = precision
    return net.to(dev)


def _warp(dev, params, pdim, qdim, precision="fp32"):
    from smpl_nerf_amd.nets import WarpFieldNet
    mw = WarpFieldNet(8, 256, pdim, qdim)
    mw.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    mw.precision = precision
    return mw.to(dev)


# ------------------------------------------------------------------------------------------ a7: human_pose_encoding = 0
@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("wb", [0, 1])
def test_smpl_nerf_raw_pose_inputs_forward_and_gradients(dev, wb, prec):
    """human_pose_encoding = 0 (config_parser.py:72, the parser default): the warp net reads [x | two joint angles]
    (models/smpl_nerf_pipeline.py:40-45).  run_fine = 0 - the reference's only working mode - forward tuple, loss and the
    gradients of the coarse and warp nets against the reference's."""
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import SmplNerfPipeline
    g = load_golden("g12_smpl_raw_pose.npz")
    pc, pf = syn.make_scene_nets(101)
    pw = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith("warp_param/")}
    mc, mf, mw = _net(dev, pc, prec), _net(dev, pf, prec), _warp(dev, pw, 3, 2, prec)
    args = O.Args(white_background=wb, run_fine=0, human_pose_encoding=0)
    pipe = SmplNerfPipeline(mc, mf, mw, args, PositionalEncoder(10, 0), PositionalEncoder(4, 0), PositionalEncoder(10, 0))
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    d = [T(a[g["sub"]], dev) for a in data[:4]] + [T(g["goal_pose"], dev), T(data[4][g["sub"]], dev)]
    with torch.no_grad():
        out = pipe(d)
    tol = 1e-4 if prec == "fp32" else 2e-4
    for nm, o_, t_ in zip(("rgb", "rgb_fine", "warp", "samples", "warped", "alpha"), out, (tol, tol, 1e-5, 0, 1e-5, 2e-4)):
        assert o_.shape == g[f"{nm}_wb{wb}"].shape, nm
        close(o_.cpu().numpy(), g[f"{nm}_wb{wb}"], 0, max(t_, 1e-12))
    out = pipe(d)
    loss = torch.nn.functional.mse_loss(out[0], d[-1]) + torch.nn.functional.mse_loss(out[1], d[-1])
    loss.backward()
    close([loss.item()], g[f"loss_wb{wb}"], 2e-5, 1e-7)
    for k, p in mc.named_parameters():
        ref = g[f"grad_wb{wb}/coarse.{k}"]
        scale = max(np.abs(ref[2:]).max(), ref[1] / np.sqrt(p.numel()), 1e-12)
        close(R.digest(p.grad), ref, 2e-2, 1e-2 * scale)
    assert all(p.grad is None for p in mf.parameters())      # run_fine = 0: the fine net is not evaluated (quirk Q10)
    for k, p in mw.named_parameters():
        ref = g[f"warpgrad_wb{wb}/{k}"].astype(np.float64)
        got = p.grad.cpu().numpy().astype(np.float64)
        assert np.linalg.norm(got - ref) <= 1e-2 * np.linalg.norm(ref), (k, np.linalg.norm(got - ref) / np.linalg.norm(ref))


def test_smpl_nerf_raw_pose_inputs_fine_branch_fails_like_the_reference(dev):
    """Quirk Q5: with human_pose_encoding = 0 the fine branch feeds 100 encoded columns into the 5-column linear1
    (models/smpl_nerf_pipeline.py:71-77) - a RuntimeError in the reference (recorded in the fixture) and here."""
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import SmplNerfPipeline
    g = load_golden("g12_smpl_raw_pose.npz")
    assert int(g["fine_branch_raises"][0]) == 1
    pc, pf = syn.make_scene_nets(101)
    pw = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith("warp_param/")}
    pipe = SmplNerfPipeline(_net(dev, pc), _net(dev, pf), _warp(dev, pw, 3, 2), O.Args(run_fine=1, human_pose_encoding=0),
                            PositionalEncoder(10, 0), PositionalEncoder(4, 0), PositionalEncoder(10, 0))
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    d = [T(a[g["sub"]], dev) for a in data[:4]] + [T(g["goal_pose"], dev), T(data[4][g["sub"]], dev)]
    with pytest.raises(RuntimeError), torch.no_grad():
        pipe(d)


@pytest.mark.parametrize("tag,pdim,qdim", [("enc", 60, 40), ("raw", 3, 2)])
def test_warp_field_net_forward_under_autograd(dev, tag, pdim, qdim):
    """WarpFieldNet.forward(x) in training mode (models/warp_field_net.py:17-22): output, the gradient of the rows and
    every parameter gradient against the reference under autograd."""
    g = load_golden("g13_warp_net_grad.npz")
    params = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith(f"param_{tag}/")}
    net = _warp(dev, params, pdim, qdim).train()
    x = T(g[f"x_{tag}"], dev).requires_grad_(True)
    out = net(x)
    assert out.requires_grad
    close(out.detach().cpu().numpy(), g[f"out_{tag}"], 1e-5, 1e-6)
    (out * T(g[f"gout_{tag}"], dev)).sum().backward()
    close(x.grad.cpu().numpy(), g[f"dx_{tag}"], 1e-4, 1e-5)
    for k, p in net.named_parameters():
        ref = g[f"grad_{tag}/{k}"]
        close(p.grad.cpu().numpy(), ref, 2e-4, 2e-5 * np.abs(ref).max())
    # 3-D input (batch, samples, columns) keeps its leading shape; parameters frozen -> rows still get their gradient
    for p in net.parameters():
        p.requires_grad_(False)
    x3 = T(g[f"x_{tag}"].reshape(8, 25, -1), dev).requires_grad_(True)
    out3 = net(x3)
    assert out3.shape == (8, 25, 3)
    (out3 * T(g[f"gout_{tag}"].reshape(8, 25, 3), dev)).sum().backward()
    close(x3.grad.cpu().numpy().reshape(200, -1), g[f"dx_{tag}"], 1e-4, 1e-5)
    with torch.no_grad():
        close(net(x3).cpu().numpy().reshape(200, 3), g[f"out_{tag}"], 1e-5, 1e-6)


def test_smpl_nerf_density_loss_reaches_the_nets(dev):
    """SmplNerfSolver's optional density loss (solver/smpl_nerf_solver.py:35-43) differentiates the returned `densities`
    and `warped_samples`: both carry gradient here.  The fine net's gradient of such a loss is checked by central
    differences of the loss along the gradient direction (the fine net does not move the hierarchical samples, so the loss
    is a smooth function of it; the warp net does, which autograd ignores like the reference - utils.py:260 - so for it
    only "finite and non-zero" is asserted)."""
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import SmplNerfPipeline
    g6 = load_golden("g6_smpl_nerf_pipeline.npz")
    pc, pf = syn.make_scene_nets(101)
    mc, mf = _net(dev, pc), _net(dev, pf)
    mw = _warp(dev, syn.make_warp_field_params(103, out_scale=0.3), 60, 40)
    pipe = SmplNerfPipeline(mc, mf, mw, O.Args(), PositionalEncoder(10, 0), PositionalEncoder(4, 0), PositionalEncoder(10, 0))
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    d = [T(a[g6["sub"]], dev) for a in data[:4]] + [T(g6["goal_pose"], dev), T(data[4][g6["sub"]], dev)]
    target = torch.rand((d[0].shape[0], 192), device=dev, generator=torch.Generator(device=dev).manual_seed(3))

    def loss_of():
        out = pipe(d)
        return torch.nn.functional.mse_loss(out[5], target) + 0.1 * out[4].square().mean()

    loss = loss_of()
    loss.backward()
    grads = {id(p): p.grad.clone() for m in (mf, mw) for p in m.parameters()}
    assert all(bool(torch.isfinite(v).all()) for v in grads.values())
    assert sum(float(v.abs().sum()) for v in grads.values()) > 0
    assert sum(float(grads[id(p)].abs().sum()) for p in mw.parameters()) > 0
    # directional central difference along the gradient of two fine-net tensors
    for p in (mf.sigma_out_layer.weight, mf.additional_linear_layer.bias):
        gdir = grads[id(p)]
        gn = float(gdir.norm())
        if gn == 0:
            continue
        eps = 1e-2 / gn * float(p.detach().abs().max() + 1e-3)
        with torch.no_grad():
            p.add_(eps * gdir)
            for m in (mf, mw):
                m.mark_weights_changed()
            lp = float(loss_of())
            p.sub_(2 * eps * gdir)
            for m in (mf, mw):
                m.mark_weights_changed()
            lm = float(loss_of())
            p.add_(eps * gdir)
            for m in (mf, mw):
                m.mark_weights_changed()
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - gn * gn) <= 0.1 * gn * gn + 1e-7, (fd, gn * gn)


# ------------------------------------------------------------------------------------------ differentiable stand-alone ops
@pytest.mark.parametrize("L,ident", [(10, 0), (4, 1), (0, 1), (6, 0)])
def test_positional_encoder_is_differentiable(dev, L, ident):
    """PositionalEncoder.encode under autograd (utils.py:123-131) vs the reference's gradient."""
    from smpl_nerf_amd.ops import PositionalEncoder
    g = load_golden("g14_ops_grads.npz")
    x = T(g["pe_x"], dev).requires_grad_(True)
    out = PositionalEncoder(L, ident).encode(x)
    (out * T(g[f"pe_gout_L{L}_id{ident}"], dev)).sum().backward()
    ref = g[f"pe_dx_L{L}_id{ident}"]
    close(x.grad.cpu().numpy(), ref, 1e-5, 1e-5 * np.abs(ref).max())
    if (L, ident) == (10, 0):
        p = T(g["pe_pose"], dev).requires_grad_(True)
        (PositionalEncoder(10, 0).encode(p) * T(g["pe_pose_gout"], dev)).sum().backward()
        close(p.grad.cpu().numpy(), g["pe_pose_dx"], 1e-5, 1e-5 * np.abs(g["pe_pose_dx"]).max())


@pytest.mark.parametrize("N", [1, 2, 64, 192, 100])
@pytest.mark.parametrize("wb", [0, 1])
@pytest.mark.parametrize("mode", ["ray", "smp"])
def test_raw2outputs_all_outputs_and_inputs_differentiable(dev, N, wb, mode):
    """raw2outputs(raw, z_vals, dirs, args) under autograd with gradients arriving at rgb, weights AND alpha, flowing to
    raw, z_vals and the directions (utils.py:134-191) - against the reference's autograd."""
    from smpl_nerf_amd import ops
    if N == 1 and mode == "smp":
        pytest.skip("N==1 ignores directions")
    g3, g = load_golden("g3_raw2outputs.npz"), load_golden("g14_ops_grads.npz")
    B = g3[f"raw_N{N}"].shape[0]
    raw = T(g3[f"raw_N{N}"], dev).requires_grad_(True)
    z = T(g3[f"z_N{N}"], dev).requires_grad_(True)
    d0 = T(g3[f"dray_N{N}"] if mode == "ray" else g3[f"dsmp_N{N}"], dev).requires_grad_(True)
    d = d0[:, None, :].expand(B, N, 3) if mode == "ray" else d0
    rgb, w, a = ops.raw2outputs(raw, z, d, O.Args(white_background=wb))
    ((rgb * T(g[f"c_grgb_N{N}"], dev)).sum() + (w * T(g[f"c_gw_N{N}"], dev)).sum() + (a * T(g[f"c_ga_N{N}"], dev)).sum()).backward()
    key = f"N{N}_wb{wb}_{mode}"
    for got, ref in ((raw.grad, g[f"c_draw_{key}"]), (z.grad, g[f"c_dz_{key}"]), (d0.grad, g[f"c_ddir_{key}"])):
        got = np.zeros_like(ref) if got is None else got.cpu().numpy()
        close(got, ref, 2e-4, 2e-5 * max(np.abs(ref).max(), 1e-12))


# ------------------------------------------------------------------------------------------ a6: every scalar type
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.int32, torch.int64, torch.int16, torch.int8, torch.uint8])
@pytest.mark.parametrize("side", ["left", "right"])
def test_searchsorted_every_scalar_type(dev, dtype, side):
    """AT_DISPATCH_ALL_TYPES (searchsorted_cpu_wrapper.cpp:100): bit-exact against numpy.searchsorted, the reference's
    own test oracle (test/test_searchsorted.py:41-44), incl. ties, broadcast rows and rows longer than the LDS stage."""
    from smpl_nerf_amd.ops import searchsorted
    rng = np.random.default_rng(11)
    npdt = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64,
            torch.int16: np.int16, torch.int8: np.int8, torch.uint8: np.uint8}[dtype]

    def draw(shape):
        if np.issubdtype(npdt, np.floating):
            return rng.normal(size=shape).astype(npdt)
        info = np.iinfo(npdt)
        lo, hi = max(info.min, -1000), min(info.max, 1000)
        return rng.integers(lo, hi + 1, size=shape).astype(npdt)      # narrow range -> many ties

    for (ra, ca, rv, cv) in [(7, 63, 7, 128), (1, 200, 9, 33), (5, 50, 1, 17), (3, 40000, 3, 70), (4, 0, 4, 5), (2, 1, 2, 300)]:
        a = np.sort(draw((ra, ca)), -1)
        v = draw((rv, cv))
        if ca >= 8:
            v[:, :4] = a[0, 2:6] if ra == 1 else (a[:rv, 2:6] if rv <= ra else a[:1, 2:6])     # exact hits
        out = searchsorted(T(a, dev), T(v, dev), side=side).cpu().numpy()
        rows = max(ra, rv)
        ref = np.stack([np.searchsorted(a[0 if ra == 1 else r], v[0 if rv == 1 else r], side=side) for r in range(rows)])
        assert out.dtype == np.int64 and out.shape == ref.shape
        np.testing.assert_array_equal(out, ref)


def test_searchsorted_rejects_mixed_and_unsupported_types(dev):
    from smpl_nerf_amd.ops import searchsorted
    a = torch.zeros((2, 4), device=dev)
    with pytest.raises(RuntimeError):
        searchsorted(a, a.double())
    with pytest.raises(RuntimeError):
        searchsorted(a.half(), a.half())


def test_composite_forward_accepts_long_rays(dev):
    """The forward has no upper bound on N (the backward takes N <= 4096): 5000 samples per ray against the oracle."""
    from smpl_nerf_amd import ops
    rng = np.random.default_rng(5)
    B, N = 3, 5000
    raw = rng.normal(0, 1.0, (B, N, 4)).astype(F32)
    raw[..., 3] *= 3.0
    z = np.sort(rng.uniform(1, 4, (B, N)).astype(F32), -1)
    d = rng.normal(size=(B, 3)).astype(F32)
    rgb, w, a = ops.composite(T(raw, dev), T(z, dev), T(d, dev), False)
    ref = O.raw2outputs(raw, z, np.broadcast_to(d[:, None, :], (B, N, 3)), 0)
    close(rgb.cpu().numpy(), ref[0], 0, 2e-6)
    close(w.cpu().numpy(), ref[1], 0, 2e-6)
    close(a.cpu().numpy(), ref[2], 0, 2e-6)


# ------------------------------------------------------------------------------------------ memory-bounded backward
def _grads(models):
    return [None if p.grad is None else p.grad.detach().clone() for m in models for p in m.parameters()]


def _assert_grads_close(got, ref, rel):
    for a, b in zip(got, ref):
        assert (a is None) == (b is None)
        if a is not None:
            a, b = a.double(), b.double()
            assert float((a - b).norm()) <= rel * float(b.norm()) + 1e-12, float((a - b).norm() / b.norm())


@pytest.mark.parametrize("prec", PRECISIONS)
def test_block_wise_backward_equals_the_stored_backward(dev, prec):
    """_FusedMlpFn with a small activation budget (forward keeps only its inputs; backward recomputes the layer inputs
    block by block and sums the parameter gradients) against the stored form: same loss bits, gradients equal to the
    additivity tolerance of the split-K sums.  NerfPipeline, 1024 rays, blocks of ~100 rays."""
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import NerfPipeline
    pc, pf = syn.make_scene_nets(101)
    data = syn.frame_batch(128, 128, seed=7)
    sub = np.arange(0, 16384, 16)
    d = [T(a[sub], dev) for a in data]
    out = {}
    for budget in (0, 40 << 20):        # 40 MB: 21 KB per sample -> ~1900 samples = 29 coarse rays / 9 fine rays per block
        mc, mf = _net(dev, pc, prec), _net(dev, pf, prec)
        mc.activation_budget_bytes = mf.activation_budget_bytes = budget
        pipe = NerfPipeline(mc, mf, O.Args(), PositionalEncoder(10, 0), PositionalEncoder(4, 0))
        r = pipe(d)
        loss = torch.nn.functional.mse_loss(r[0], d[-1]) + torch.nn.functional.mse_loss(r[1], d[-1])
        loss.backward()
        out[budget] = (float(loss.detach()), _grads((mc, mf)))
    assert out[0][0] == out[40 << 20][0]                     # the inference kernel and the training forward agree bit for bit
    _assert_grads_close(out[40 << 20][1], out[0][1], 2e-5 if prec == "fp32" else 2e-4)


def test_block_wise_backward_with_input_gradients_and_per_ray_inputs(dev):
    """The same through SmplNerfPipeline (positions / per-sample directions receive gradients, the warp net trains) and
    through a pose-conditioned net (per-ray additional inputs receive theirs)."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import SmplNerfPipeline
    g6 = load_golden("g6_smpl_nerf_pipeline.npz")
    pc, pf = syn.make_scene_nets(101)
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    d = [T(a[g6["sub"]], dev) for a in data[:4]] + [T(g6["goal_pose"], dev), T(data[4][g6["sub"]], dev)]
    out = {}
    for budget in (0, 24 << 20):
        mc, mf = _net(dev, pc), _net(dev, pf)
        mw = _warp(dev, syn.make_warp_field_params(103, out_scale=0.3), 60, 40)
        mc.activation_budget_bytes = mf.activation_budget_bytes = budget
        pipe = SmplNerfPipeline(mc, mf, mw, O.Args(), PositionalEncoder(10, 0), PositionalEncoder(4, 0), PositionalEncoder(10, 0))
        r = pipe(d)
        loss = torch.nn.functional.mse_loss(r[0], d[-1]) + torch.nn.functional.mse_loss(r[1], d[-1])
        loss.backward()
        out[budget] = (float(loss.detach()), _grads((mc, mf, mw)))
    assert out[0][0] == out[24 << 20][0]
    _assert_grads_close(out[24 << 20][1], out[0][1], 1e-4)
    # per-ray additional inputs (69 pose columns in front of the encoding) with their own gradient
    params = syn.make_scene_net_params(301, add_first=True, additional_input_dim=69)
    rng = np.random.default_rng(2)
    x = T(rng.uniform(-1.5, 1.5, (96 * 64, 3)).astype(F32), dev)
    dirs = T(rng.normal(size=(96, 3)).astype(F32), dev)
    gout = T(rng.normal(size=(96 * 64, 4)).astype(F32), dev)
    res = {}
    for budget in (0, 16 << 20):
        net = RenderRayNet(8, 256, 60, 24, 69, skips=[4])
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        net = net.to(dev)
        net.activation_budget_bytes = budget
        add = T(rng.uniform(-1, 1, (96, 69)).astype(F32), dev) if budget == 0 else res["add"].detach().clone()
        add.requires_grad_(True)
        res["add"] = add
        raw = net.forward_fused(x, dirs, 64, PositionalEncoder(10, 0), PositionalEncoder(4, 0), additional=add, add_first=True)
        (raw * gout).sum().backward()
        res[budget] = (raw.detach().clone(), add.grad.clone(), _grads((net,)))
    assert torch.equal(res[0][0], res[16 << 20][0])
    _assert_grads_close([res[16 << 20][1]] + res[16 << 20][2], [res[0][1]] + res[0][2], 1e-4)


def test_one_256x256_frame_trains_in_one_step(dev):
    """a frame-sized batch of BASELINE configs[3]/[4] (65 536 rays = 16.8 M ray-samples, 357 GB of layer
    inputs + d Y if stored) in ONE step under the default 64 GB activation budget: the gradient equals the mean of the
    gradients of its four 16 384-ray quarters (each computed with everything stored), and the peak allocation stays
    below 100 GB."""
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import NerfPipeline
    pc, pf = syn.make_scene_nets(101)
    data = syn.frame_batch(256, 256, seed=7)
    full = [T(a, dev) for a in data]
    assert full[0].shape[0] == 65536

    def run(batch, budget):
        mc, mf = _net(dev, pc), _net(dev, pf)
        if budget is not None:
            mc.activation_budget_bytes = mf.activation_budget_bytes = budget
        pipe = NerfPipeline(mc, mf, O.Args(), PositionalEncoder(10, 0), PositionalEncoder(4, 0))
        r = pipe(batch)
        loss = torch.nn.functional.mse_loss(r[0], batch[-1]) + torch.nn.functional.mse_loss(r[1], batch[-1])
        loss.backward()
        return float(loss.detach()), _grads((mc, mf))

    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    loss_full, g_full = run(full, None)                    # default budget: 64 GB per call -> block-wise
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated(dev)
    assert peak < 100 * (1 << 30), peak
    acc, losses = None, []
    for q in range(4):
        part = [t[q * 16384:(q + 1) * 16384].contiguous() for t in full]
        l, g = run(part, 0)                                # everything stored (89 GB per quarter)
        losses.append(l)
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        del g
        torch.cuda.empty_cache()
    mean = [a / 4 for a in acc]
    assert abs(loss_full - sum(losses) / 4) <= 1e-6 * abs(loss_full) + 1e-8
    _assert_grads_close(g_full, mean, 5e-5)


# ------------------------------------------------------------------------------------------ 200 Adam steps vs the reference
@pytest.mark.parametrize("prec", ["fp32", "bf16x6"])
def test_200_adam_steps_follow_the_reference_loss_curve(dev, prec):
    """the reference's own NerfSolver objects trained 200 steps (B = 256, Adam lr 3e-5) on the synthetic
    scene (tests/golden/make_golden_train.py -> g15_train200.npz: loss curve 0.46 -> 0.0085, validation PSNR 24.69 dB).
    The HIP path on the same batches, through DataParallelTrainer.step (solver/nerf_solver.py:76-89):

      * the yardstick is the reference against ITSELF: the fixture also holds its curve with 3 instead of 8 CPU threads
        (only MKL's summation order changes): max |relative deviation| 4.0 %, mean 0.38 % - 200 optimiser steps amplify
        fp32 round-off.  Bounds here: every step within 2 x that maximum (8 %), the mean within 1.5 % (the HIP kernels'
        summation structure differs more from MKL's than MKL's from itself: 0.7 % measured), the first step to 1e-5;
      * the trained model: PSNR of the fine rendering on the validation rays within 0.1 dB of the reference's 24.69 dB
        (measured +0.045 / -0.018 dB), validation loss within 10 % - the coarse + fine validation loss of a 200-step
        trajectory is itself chaotic at that level: across arithmetically equivalent variants of this path (fused vs
        unfused Adam, folded vs unfolded weight-gradient jobs, fp32 / bf16x6 / f16x3) it lands between -0.9 % and +7.9 % of
        the reference's (tools/ab/train200.py), while the PSNR moves by < 0.05 dB;
      * the validation rendering within 3e-2 of the reference's pixel values."""
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import NerfPipeline
    from smpl_nerf_amd.trainer import DataParallelTrainer
    g = load_golden("g15_train200.npz")
    pc, pf = syn.make_scene_nets(101)
    mc, mf = _net(dev, pc, prec).train(), _net(dev, pf, prec).train()
    pipe = NerfPipeline(mc, mf, O.Args(), PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    tr = DataParallelTrainer(pipe, [mc, mf], lr=float(g["lr"][0]))
    data = [T(a, dev) for a in syn.frame_batch(128, 128, seed=7)]
    idx = torch.from_numpy(g["idx"]).to(dev)
    losses = [tr.step([t[idx[i]] for t in data]) for i in range(int(g["steps"][0]))]
    losses = torch.stack(losses).double().cpu().numpy()
    ref = g["losses"]
    rel = np.abs(losses - ref) / ref
    print(f"[{prec}] rel dev: first {rel[0]:.2e} max {rel.max():.3e} mean {rel.mean():.3e}; by 50: "
          f"{[float('%.3g' % rel[i:i + 50].max()) for i in range(0, 200, 50)]}")
    own = np.abs(g["losses_3_threads"] - ref) / ref          # the reference against itself (other summation order)
    assert rel[0] <= 1e-5, rel[0]
    assert rel.max() <= 2 * own.max() and rel.mean() <= 1.5e-2, (rel.max(), rel.mean(), own.max(), own.mean())
    mc.eval(), mf.eval()
    vi = torch.from_numpy(g["val_idx"]).to(dev)
    with torch.no_grad():
        vb = [t[vi] for t in data]
        out = pipe(vb)
        val_loss = float(tr.loss(out[0], out[1], vb[-1]))
        psnr = -10.0 * np.log10(float(torch.mean((out[1] - vb[-1]) ** 2)))
    assert abs(val_loss - g["val_loss"][0]) <= 0.10 * g["val_loss"][0], (val_loss, g["val_loss"][0])
    assert abs(psnr - g["val_psnr_fine"][0]) <= 0.1, (psnr, g["val_psnr_fine"][0])
    assert float(np.abs(out[1].cpu().numpy() - g["val_rgb_fine"]).max()) <= 3e-2
    print(f"[{prec}] loss curve: max rel dev {rel.max():.2e}, mean {rel.mean():.2e}; val PSNR {psnr:.3f} dB "
          f"(reference {g['val_psnr_fine'][0]:.3f})")


def test_sample_pdf_is_differentiable(dev):
    """sample_pdf(bins, weights, args) under autograd (utils.py:194-228): the kernel's samples, the reference's gradient
    w.r.t. bins and weights (g16_sample_pdf_grad.npz incl. the adversarial rows of g4).  Strict mode (normalising sums as
    the reference's host evaluates them), so the indices - the pieces of the piecewise-linear map - are the reference's."""
    from smpl_nerf_amd import ops
    g = load_golden("g16_sample_pdf_grad.npz")
    bins = T(g["bins"], dev).requires_grad_(True)
    w = T(g["weights"], dev).requires_grad_(True)
    out = ops.sample_pdf(bins, w, O.Args(number_fine_samples=128, strict_cumsum=1))
    assert out.requires_grad
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g["samples"])
    (out * T(g["gout"], dev)).sum().backward()
    for got, ref in ((bins.grad.cpu().numpy(), g["d_bins"]), (w.grad.cpu().numpy(), g["d_weights"])):
        rows = np.abs(got - ref).max(-1) <= 1e-3 * np.maximum(np.abs(ref).max(-1), 1e-6)      # per row, relative to the row
        # the adversarial rows (single spike, mass at the ends) sit on the `denom < 1e-5` kink of utils.py:224: the backward's
        # cdf (device cumsum) and the reference's (host cumsum) fall on different sides of it for a few entries
        assert rows.mean() >= 0.9, (rows.mean(), np.abs(got - ref).max())
    with torch.no_grad():
        np.testing.assert_array_equal(ops.sample_pdf(bins, w, O.Args(number_fine_samples=128, strict_cumsum=1)).cpu().numpy(), g["samples"])


# ------------------------------------------------------------------------------------------ RCCL on the GPU box
def _rccl_worker(port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dv = torch.device("cuda", 0)
    torch.cuda.set_device(dv)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dv)      # what bench.py / the trainer do on an 8-GPU node
    try:
        from smpl_nerf_amd import dist as sd
        flat = torch.arange(1220872, device=dv, dtype=torch.float32)          # the nerf gradient buffer's size
        ref = flat.clone()
        sd.allreduce_mean_(flat)                                              # RCCL all-reduce on the compute stream
        sd.broadcast_(flat, 0)
        sd.barrier(dv)                                                        # barrier(device_ids=[...]) as in bench.py
        t = sd.max_over_ranks(1.25, dv)
        rows = sd.gather_rows(torch.ones((5, 3), device=dv), 5)
        torch.cuda.synchronize()
        q.put((bool(torch.equal(flat, ref)), t, tuple(rows.shape), dist.get_backend()))
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_of_the_path_run_on_this_gpu():
    """N > 1 cannot run on a 1-GPU box, but RCCL can: a world-size-1 "nccl" group through every collective the path uses
    (flat-buffer all-reduce, replica broadcast, device barrier, max over ranks, row gather) - initialisation, stream
    semantics and the dmabuf IPC setting of this image are exercised for real, only the xGMI transport is not."""
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    same, t, shape, backend = q.get()
    assert same and t == 1.25 and shape == (5, 3) and backend == "nccl"


def test_bench_takes_its_rccl_branch_at_world_size_one():
    """bench.py with SNERF_BENCH_FORCE_GROUP=1: process group on "nccl" (= RCCL) with device_id, device barrier and
    max-over-ranks on the device - the branch the driver's 8-GPU run takes, here with one rank."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SNERF_BENCH_FORCE_GROUP="1", SNERF_DIST_BACKEND="nccl")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-rays", "0",
                        "--train-rays", "512", "--train-steps", "2", "--no-alt", "--no-pmc", "--points="],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and "nccl" in line["config"]["parallelism"] and line["value"] > 5e7
    assert line["train"]["value"] > 1e7
    # the record of a grouped run: the backend's own view, the all-reduce timed on RCCL, per-rank entries
    assert line["collective"]["backend"] == "nccl (RCCL)" and line["collective"]["world_size_seen_by_backend"] == 1
    assert len(line["per_rank"]) == 1 and line["per_rank"][0]["rank"] == 0 and "gfx950" in line["per_rank"][0]["arch"]
    c = line["train"]["collective"]
    assert c["backend"] == "nccl (RCCL)" and c["bytes"] == 4 * 1220872 and c["allreduce_calls"] == 2
    assert 0 < c["allreduce_ms_per_step"] < 50 and len(line["train"]["per_rank"]) == 1
    # r05: the grouped step is ONE C-ABI call with the all-reduce inside (snerf_nerf_train_step_dp_f32)
    assert c["allreduce_inside_the_step_call"] and c["rccl_comm_world_rank"] == [1, 0]
    assert line["train"]["c_abi_calls_per_step"] == 1.0


def test_smpl_render_rays_covers_the_modes_without_a_one_call_entry(dev):
    """SmplNerfPipeline.render_rays with run_fine = 0 (and with human_pose_encoding = 0) - the modes the one-call C entry
    does not cover - returns what forward() returns, under no_grad, instead of raising."""
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import SmplNerfPipeline
    g = load_golden("g12_smpl_raw_pose.npz")
    pc, pf = syn.make_scene_nets(101)
    pw = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith("warp_param/")}
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    d = [T(a[g["sub"]], dev) for a in data[:4]] + [T(g["goal_pose"], dev), T(data[4][g["sub"]], dev)]
    pipe = SmplNerfPipeline(_net(dev, pc), _net(dev, pf), _warp(dev, pw, 3, 2), O.Args(run_fine=0, human_pose_encoding=0),
                            PositionalEncoder(10, 0), PositionalEncoder(4, 0), PositionalEncoder(10, 0))
    out = pipe.render_rays(d)
    with torch.no_grad():
        ref = pipe(d)
    assert len(out) == 6 and all(not o.requires_grad for o in out)
    for a, b in zip(ref, out):
        assert torch.equal(a, b)
    close(out[0].cpu().numpy(), g["rgb_wb0"], 0, 1e-4)


def test_block_wise_backward_of_encoded_rows(dev):
    """RenderRayNet.forward(x_enc) under a small activation budget: the encoded rows' gradient and the parameter gradients of
    the block-wise backward equal the stored form's."""
    from smpl_nerf_amd.nets import RenderRayNet
    g2, g7 = load_golden("g2_mlp.npz"), load_golden("g7_grads.npz")
    params = syn.make_render_ray_net_params(11, 30.0, 10.0, skips=(4,))
    res = {}
    for budget in (0, 1 << 20):        # 1 MB: ~49 rows per block of the 160
        net = RenderRayNet(8, 256, 60, 24, skips=[4])
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        net = net.to(dev)
        net.activation_budget_bytes = budget
        x = T(g2["inputs"], dev).requires_grad_(True)
        raw = net(x)
        (raw * T(g7["m_gout"], dev)).sum().backward()
        res[budget] = (raw.detach().clone(), x.grad.clone(), _grads((net,)))
    assert torch.equal(res[0][0], res[1 << 20][0])
    _assert_grads_close([res[1 << 20][1]] + res[1 << 20][2], [res[0][1]] + res[0][2], 2e-5)
    for k, p in zip([k for k, _ in net.named_parameters()], res[1 << 20][2]):
        ref = g7[f"m_skip4/{k}"]
        close(R.digest(p), ref, 5e-4, 5e-5 * max(np.abs(ref[2:]).max(), ref[1] / np.sqrt(p.numel())))


# ------------------------------------------------------------------------------------------ sampler store paths (r03)
@pytest.mark.parametrize("Nc,Nf", [(64, 128), (64, 64), (33, 31), (16, 16), (8, 4), (100, 92)])
def test_sample_pdf_store_paths_agree(dev, Nc, Nf):
    """sample_pdf_kernel writes rows of a multiple of four samples as 16-byte vectors (the points through an LDS staging
    area) and everything else dword by dword; the 64 + 128 shape runs a compile-time-sized instance.  Every path must give
    the same bits: outputs at 16-byte-aligned addresses against outputs one float off alignment (which forces the dword
    path), for shapes on both sides of each condition, and both against the numpy oracle's z_fine / points."""
    from smpl_nerf_amd import _lib
    from smpl_nerf_amd.ops import ptr, current_stream, check, uniform_u
    rng = np.random.default_rng(100 * Nc + Nf)
    B = 257
    z = np.sort(rng.uniform(1, 4, (B, Nc)).astype(F32), axis=-1)
    w = (rng.uniform(0, 1, (B, Nc)) ** 4).astype(F32)
    o, d = rng.normal(size=(B, 3)).astype(F32), rng.normal(size=(B, 3)).astype(F32)
    tz, tw, to, td = T(z, dev), T(w, dev), T(o, dev), T(d, dev)
    u = uniform_u(Nf, dev)
    Nt = Nc + Nf
    lib = _lib.load()
    outs = []
    for off in (0, 1):
        zf_buf = torch.full((B * Nt + 4,), float("nan"), device=dev)
        pts_buf = torch.full((B * Nt * 3 + 4,), float("nan"), device=dev)
        zf, pts = zf_buf[off:off + B * Nt], pts_buf[off:off + B * Nt * 3]
        assert zf.data_ptr() % 16 == 4 * off and pts.data_ptr() % 16 == 4 * off
        with torch.cuda.device(dev):
            check(lib.snerf_sample_pdf_f32(ptr(tz), ptr(tw), ptr(u), ptr(to), ptr(td), B, Nc, Nf, None, None, ptr(zf), ptr(pts),
                                           current_stream()), "snerf_sample_pdf_f32")
        torch.cuda.synchronize()
        assert torch.isnan(zf_buf[:off]).all() and torch.isnan(zf_buf[off + B * Nt:]).all()        # nothing outside the rows
        assert torch.isnan(pts_buf[:off]).all() and torch.isnan(pts_buf[off + B * Nt * 3:]).all()
        outs.append((zf.cpu().numpy().reshape(B, Nt), pts.cpu().numpy().reshape(B, Nt, 3)))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    z_ref, pts_ref = O.fine_sampling(o, d, z, w, Nf, u=u.cpu().numpy())
    np.testing.assert_array_equal(outs[0][0], z_ref)
    np.testing.assert_array_equal(outs[0][1], pts_ref)


# ------------------------------------------------------------------------------------------ small calls (r03)
@pytest.mark.parametrize("rays", [1, 64, 200])
def test_small_calls_use_64_sample_tiles_and_equal_the_large_call(dev, rays):
    """Calls of <= 64 x CUs samples run the 4-wave (64-sample-tile) form of the fp32 forward (csrc/mlp.hip: launch_fwd) -
    half the latency at the README's 64-ray batches.  Same per-sample arithmetic: the rows of a small call equal the same
    rows evaluated inside a frame-sized call bit for bit, in inference and in the training forward (whose gradients then
    match to summation order)."""
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(rays)
    pc, _ = syn.make_scene_nets(101)
    Ns, big = 64, 1024
    pts = rng.uniform(-2, 2, (big, Ns, 3)).astype(F32)
    dirs = rng.normal(size=(big, 3)).astype(F32)
    pe, de = PositionalEncoder(10, 0), PositionalEncoder(4, 0)
    net = _net(dev, pc)
    with torch.no_grad():
        full = net.forward_fused(T(pts, dev), T(dirs, dev), Ns, pe, de)                    # 65 536 samples: 128-sample tiles
        small = net.forward_fused(T(pts[:rays], dev), T(dirs[:rays], dev), Ns, pe, de)     # <= 12 800 samples: 64-sample tiles
    assert torch.equal(small.reshape(-1, 4), full.reshape(-1, 4)[:rays * Ns])
    gout = T(rng.normal(size=(rays * Ns, 4)).astype(F32), dev)
    grads = []
    for n_rays in (rays, big):
        net.zero_grad(set_to_none=True)
        raw = net.forward_fused(T(pts[:n_rays], dev), T(dirs[:n_rays], dev), Ns, pe, de).reshape(-1, 4)
        (raw[:rays * Ns] * gout).sum().backward()
        grads.append([p.grad.clone() for p in net.parameters()])
        if n_rays == rays:
            assert torch.equal(raw.detach(), full.reshape(-1, 4)[:rays * Ns])
    for a, b in zip(*grads):
        scale = max(b.abs().max().item(), 1e-20)
        assert (a - b).abs().max().item() <= 2e-5 * scale


# ------------------------------------------------------------------------------------------ any --netwidth (r03)
@pytest.mark.parametrize("n_layers,width,skips", [(8, 64, (4,)), (4, 100, (1,)), (8, 200, (4,)), (3, 250, ()), (5, 30, (2,)),
                                                  (2, 7, ()), (1, 256, ()), (1, 40, ()), (16, 128, (0, 7, 14))])
def test_render_ray_net_of_any_width_up_to_256(dev, n_layers, width, skips):
    _render_ray_net_width_case(dev, n_layers, width, skips)


def _render_ray_net_width_case(dev, n_layers, width, skips):
    """config_parser.py:20 `--netwidth` is free; the kernels exist for trunks of 128 and 256 features, other widths run
    zero-padded inside the next larger one (csrc/mlp_plan.h: make_plan).  Output of the fused forward (positions +
    directions), of forward(encoded rows), and every parameter gradient against the torch fp32 restatement of
    models/render_ray_net.py:42-61 (pinned against the reference in test_grad_golden.py)."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(width)
    kw = dict(n_layers=n_layers, width=width, skips=skips)
    params = syn.make_render_ray_net_params(7 + width, 30.0, 10.0, **kw)
    if n_layers > 8:      # torch's default init shrinks the signal by ~2.4x per ReLU layer: at depth 16 every pre-activation
        for i in range(n_layers - 1):           # would sit within round-off of zero - keep the variance instead (He scaling)
            params[f"positional_net.{i}.weight"] = (params[f"positional_net.{i}.weight"] * F32(np.sqrt(6.0))).astype(F32)
    net = RenderRayNet(n_layers, width, 60, 24, skips=list(skips))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net = net.to(dev)
    n = 777
    pts, dirs = rng.uniform(-2, 2, (n, 1, 3)).astype(F32), rng.normal(size=(n, 3)).astype(F32)
    gout = rng.normal(size=(n, 4)).astype(F32)
    dn = dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)
    x_enc = torch.cat([R.posenc(torch.from_numpy(pts[:, 0]), 10, 0), R.posenc(torch.from_numpy(dn.astype(F32)), 4, 0)], -1)
    P = R.tparams(params)
    ref = R.render_ray_net(P, x_enc, n_layers=n_layers, skips=skips)
    (ref * torch.from_numpy(gout)).sum().backward()
    tol = 2e-5 * max(1.0, ref.detach().abs().max().item())
    with torch.no_grad():
        raw = net.forward_fused(T(pts, dev), T(dirs, dev), 1, PositionalEncoder(10, 0), PositionalEncoder(4, 0))
        close(raw.cpu().numpy(), ref.detach().numpy(), 0, 5 * tol)       # (the kernel encodes and normalises itself)
        close(net(x_enc.to(dev)).cpu().numpy(), ref.detach().numpy(), 0, tol)
    raw = net(x_enc.to(dev))
    (raw * T(gout, dev)).sum().backward()
    got, want = {}, {}
    for k, p in net.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape
        got[k], want[k] = p.grad.cpu().numpy(), P[k].grad.numpy()
    # |err| <= 5e-4 |g| + 5e-5 max|g| for every parameter - or exactly ONE ReLU mask bit of difference, adjudicated in float64
    # (torch_ref.check_grads_or_one_relu_kink: one row of one ReLU layer, nothing towards the output, |pre-activation| < 1e-6)
    note = R.check_grads_or_one_relu_kink(got, want, {k: torch.from_numpy(v).double() for k, v in params.items()}, x_enc.double(),
                                         n_layers, skips)
    if note:
        print(f"({n_layers}, {width}, {skips}): {note}")
    return note


@pytest.mark.parametrize("width", [64, 100, 200])
def test_warp_field_net_of_any_width_up_to_256(dev, width):
    """`--netwidth_warp` likewise (config_parser.py:30): WarpFieldNet.forward(rows) and its gradients against torch."""
    rng = np.random.default_rng(width)
    params = syn.make_warp_field_params(5 + width, width=width)
    from smpl_nerf_amd.nets import WarpFieldNet
    net = WarpFieldNet(8, width, 60, 40)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net = net.to(dev).train()
    x = rng.uniform(-1, 1, (300, 100)).astype(F32)
    gout = rng.normal(size=(300, 3)).astype(F32)
    P = R.tparams(params)
    xt = torch.from_numpy(x).requires_grad_(True)
    ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(xt, P["linear1.weight"], P["linear1.bias"])),
                                     P["linear2.weight"], P["linear2.bias"])
    (ref * torch.from_numpy(gout)).sum().backward()
    xg = T(x, dev).requires_grad_(True)
    out = net(xg)
    close(out.detach().cpu().numpy(), ref.detach().numpy(), 1e-5, 1e-6)
    (out * T(gout, dev)).sum().backward()
    close(xg.grad.cpu().numpy(), xt.grad.numpy(), 1e-4, 1e-5)
    for k, p in net.named_parameters():
        g = P[k].grad.numpy()
        close(p.grad.cpu().numpy(), g, 2e-4, 2e-5 * np.abs(g).max())


# ------------------------------------------------------------------------------------------ warp net: per-ray pose fold (r03)
@pytest.mark.parametrize("tag,pdim,qdim,pos_enc", [("enc", 60, 40, (10, 0)), ("raw", 3, 2, (0, 1))])
def test_warp_inference_folds_the_pose_columns_per_ray(dev, tag, pdim, qdim, pos_enc):
    """The pose encoding is a per-ray constant (models/smpl_nerf_pipeline.py:40-45), so the inference kernel adds its
    columns of linear1 once per ray (csrc/warp.hip: warp_ray_bias_kernel) instead of per sample; the training forward does
    not fold.  Both must give the reference's warp: compare inference (folded), training forward (unfolded) and the torch
    evaluation of models/warp_field_net.py:17-21 on rays of 64 and of 7 samples (a wave's 16 samples then span 3 rays)."""
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(pdim)
    params = syn.make_warp_field_params(31 + pdim, positions_dim=pdim, pose_dim=qdim, out_scale=0.3)
    net = _warp(dev, params, pdim, qdim)
    P = R.tparams(params, requires_grad=False)
    pe = PositionalEncoder(*pos_enc)
    for B, Ns in ((37, 64), (301, 9), (50, 7)):      # 9: a wave's 16 samples span 3 rays; 7: below the fold's threshold of 8
        x = rng.uniform(-1.5, 1.5, (B, Ns, 3)).astype(F32)
        pose = rng.uniform(-1, 1, (B, qdim)).astype(F32)          # what the pipeline passes: the encoded (or raw) pose per ray
        o = rng.normal(size=(B, 3)).astype(F32)
        rows = torch.cat([R.posenc(torch.from_numpy(x), pos_enc[0], pos_enc[1]),
                          torch.from_numpy(pose)[:, None, :].expand(B, Ns, qdim)], -1)
        ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(rows, P["linear1.weight"], P["linear1.bias"])),
                                         P["linear2.weight"], P["linear2.bias"]).numpy()
        with torch.no_grad():
            inf = net.forward_fused(T(x, dev), T(pose, dev), T(o, dev), Ns, pe)
        trn = net.forward_fused(T(x, dev), T(pose, dev).requires_grad_(True), T(o, dev), Ns, pe)
        for out in (inf, trn):
            close(out[0].detach().cpu().numpy().reshape(B, Ns, 3), ref, 1e-5, 2e-6)
        close(inf[0].cpu().numpy(), trn[0].detach().cpu().numpy(), 1e-5, 1e-6)
        close(inf[1].cpu().numpy(), trn[1].detach().cpu().numpy(), 1e-6, 1e-6)       # warped points


# ------------------------------------------------------------------------------------------ per-ray additional inputs folded (r03)
@pytest.mark.parametrize("add_dim,add_first,skips,width", [(69, True, (4,), 256), (2, False, (4,), 256), (69, True, (), 256),
                                                           (20, True, (1, 5), 128), (69, False, (3,), 200)])
def test_inference_folds_per_ray_additional_inputs(dev, add_dim, add_first, skips, width):
    """The additional inputs of append_smpl_params / append_to_nerf (and the vertex floats of AppendVerticesNet) are
    per-ray constants: the fp32 inference kernel takes W_add . add once per ray and layer (csrc/mlp.hip:
    mlp_add_fold_kernel) and skips those k-blocks; the training forward multiplies them per sample.  Folded inference,
    unfolded training forward and the torch evaluation of models/render_ray_net.py:42-61 must agree - on rays of 64 samples
    and of 7 (a wave's 16 samples then span 3 rays), ragged against the 128-sample tile."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(add_dim + width)
    kw = dict(n_layers=8, width=width, skips=skips, additional_input_dim=add_dim)
    params = syn.make_render_ray_net_params(41 + add_dim, 30.0, 10.0, **kw)
    net = RenderRayNet(8, width, 60, 24, add_dim, skips=list(skips))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net = net.to(dev)
    P = R.tparams(params, requires_grad=False)
    pe, de = PositionalEncoder(10, 0), PositionalEncoder(4, 0)
    for B, Ns in ((41, 64), (333, 9), (50, 7)):      # 9: a wave's 16 samples span 3 rays; 7: below the fold's threshold of 8
        x = rng.uniform(-2, 2, (B, Ns, 3)).astype(F32)
        d = rng.normal(size=(B, 3)).astype(F32)
        add = rng.uniform(-1, 1, (B, add_dim)).astype(F32)
        dn = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(F32)
        pex = R.posenc(torch.from_numpy(x), 10, 0)
        addx = torch.from_numpy(add)[:, None, :].expand(B, Ns, add_dim)
        ded = R.posenc(torch.from_numpy(dn), 4, 0)[:, None, :].expand(B, Ns, 24)
        rows = torch.cat([addx, pex, ded] if add_first else [pex, addx, ded], -1).reshape(B * Ns, -1)
        ref = R.render_ray_net(P, rows, n_layers=8, additional_input_dim=add_dim, skips=skips).numpy()
        with torch.no_grad():
            inf = net.forward_fused(T(x, dev), T(d, dev), Ns, pe, de, additional=T(add, dev), add_first=add_first)
        trn = net.forward_fused(T(x, dev), T(d, dev), Ns, pe, de, additional=T(add, dev), add_first=add_first)
        assert trn.requires_grad and not inf.requires_grad
        tol = 5e-5 * max(1.0, np.abs(ref).max())
        close(inf.cpu().numpy().reshape(-1, 4), ref, 0, tol)
        close(trn.detach().cpu().numpy().reshape(-1, 4), ref, 0, tol)
        close(inf.cpu().numpy(), trn.detach().cpu().numpy(), 0, 0.2 * tol)


def test_random_network_shapes_against_torch(dev):
    """A seeded sweep over what the reference's parser leaves free - depth, width, skip layers, encoder sizes, additional
    (per-ray) inputs in both column orders, directional input on / off, samples per ray - inference (with the per-ray
    folds and the 64-sample tiles where they apply) and the training forward against the torch restatement of
    models/render_ray_net.py:42-61."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(20260929)
    for case in range(24):
        n_layers = int(rng.integers(1, 11))
        width = int(rng.choice([8, 33, 64, 90, 128, 177, 256]))
        skips = tuple(sorted(set(int(v) for v in rng.integers(0, max(1, n_layers - 1), size=int(rng.integers(0, 3))))))
        pL, pid = int(rng.integers(1, 11)), int(rng.integers(0, 2))
        dL, did = int(rng.integers(1, 5)), int(rng.integers(0, 2))
        add_dim = int(rng.choice([0, 0, 2, 20, 69]))
        add_first = bool(rng.integers(0, 2))
        use_dir = int(rng.integers(0, 4) > 0)
        B, Ns = int(rng.integers(3, 40)), int(rng.choice([5, 16, 64]))
        pdim, ddim = 3 * (pid + 2 * pL), 3 * (did + 2 * dL)
        kw = dict(n_layers=n_layers, width=width, positions_dim=pdim, directions_dim=ddim, additional_input_dim=add_dim,
                  skips=skips, use_directional_input=use_dir)
        params = syn.make_render_ray_net_params(1000 + case, 10.0, 5.0, **kw)
        net = RenderRayNet(n_layers, width, pdim, ddim, add_dim, skips=list(skips), use_directional_input=use_dir)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        net = net.to(dev)
        x = rng.uniform(-1.5, 1.5, (B, Ns, 3)).astype(F32)
        d = rng.normal(size=(B, 3)).astype(F32)
        add = rng.uniform(-1, 1, (B, add_dim)).astype(F32) if add_dim else None
        dn = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(F32)
        cols = [R.posenc(torch.from_numpy(x), pL, pid)]
        if add_dim:
            a = torch.from_numpy(add)[:, None, :].expand(B, Ns, add_dim)
            cols = [a] + cols if add_first else cols + [a]
        cols.append(R.posenc(torch.from_numpy(dn), dL, did)[:, None, :].expand(B, Ns, ddim))
        rows = torch.cat(cols, -1).reshape(B * Ns, -1)
        ref = R.render_ray_net(R.tparams(params, requires_grad=False), rows, n_layers=n_layers, positions_dim=pdim,
                               directions_dim=ddim, additional_input_dim=add_dim, skips=skips,
                               use_directional_input=use_dir).numpy()
        pe, de = PositionalEncoder(pL, pid), PositionalEncoder(dL, did)
        extra = dict(additional=T(add, dev), add_first=add_first) if add_dim else {}
        with torch.no_grad():
            inf = net.forward_fused(T(x, dev), T(d, dev), Ns, pe, de, **extra)
        trn = net.forward_fused(T(x, dev), T(d, dev), Ns, pe, de, **extra)
        tol = 1e-4 * max(1.0, np.abs(ref).max())
        msg = f"case {case}: {kw}, add_first {add_first}, B {B}, Ns {Ns}"
        np.testing.assert_allclose(inf.cpu().numpy().reshape(-1, 4), ref, rtol=0, atol=tol, err_msg=msg)
        np.testing.assert_allclose(trn.detach().cpu().numpy().reshape(-1, 4), ref, rtol=0, atol=tol, err_msg=msg)


def test_random_network_shapes_gradients_against_torch(dev):
    """The same sweep for the backward: every parameter gradient and the gradient of the per-ray additional inputs (where
    the case has them) against torch autograd over the restated net."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(929)
    for case in range(14):
        n_layers = int(rng.integers(1, 9))
        width = int(rng.choice([16, 50, 64, 100, 128, 200, 256]))
        skips = tuple(sorted(set(int(v) for v in rng.integers(0, max(1, n_layers - 1), size=int(rng.integers(0, 3))))))
        pL, dL = int(rng.integers(1, 11)), int(rng.integers(1, 5))
        add_dim = int(rng.choice([0, 2, 20, 69]))
        add_first = bool(rng.integers(0, 2))
        B, Ns = int(rng.integers(3, 30)), int(rng.choice([5, 16, 64]))
        pdim, ddim = 6 * pL, 6 * dL
        kw = dict(n_layers=n_layers, width=width, positions_dim=pdim, directions_dim=ddim, additional_input_dim=add_dim, skips=skips)
        params = syn.make_render_ray_net_params(2000 + case, 10.0, 5.0, **kw)
        net = RenderRayNet(n_layers, width, pdim, ddim, add_dim, skips=list(skips))
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        net = net.to(dev)
        x = rng.uniform(-1.5, 1.5, (B, Ns, 3)).astype(F32)
        d = rng.normal(size=(B, 3)).astype(F32)
        gout = rng.normal(size=(B * Ns, 4)).astype(F32)
        dn = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(F32)
        P = R.tparams(params)
        cols = [R.posenc(torch.from_numpy(x), pL, 0)]
        add_t = None
        if add_dim:
            add_t = torch.from_numpy(rng.uniform(-1, 1, (B, add_dim)).astype(F32)).requires_grad_(True)
            a = add_t[:, None, :].expand(B, Ns, add_dim)
            cols = [a] + cols if add_first else cols + [a]
        cols.append(R.posenc(torch.from_numpy(dn), dL, 0)[:, None, :].expand(B, Ns, ddim))
        ref = R.render_ray_net(P, torch.cat(cols, -1).reshape(B * Ns, -1), n_layers=n_layers, positions_dim=pdim,
                               directions_dim=ddim, additional_input_dim=add_dim, skips=skips)
        (ref * torch.from_numpy(gout)).sum().backward()
        extra = {}
        if add_dim:
            add_g = T(add_t.detach().numpy(), dev).requires_grad_(True)
            extra = dict(additional=add_g, add_first=add_first)
        raw = net.forward_fused(T(x, dev), T(d, dev), Ns, PositionalEncoder(pL, 0), PositionalEncoder(dL, 0), **extra)
        (raw.reshape(-1, 4) * T(gout, dev)).sum().backward()
        msg = f"case {case}: {kw}, add_first {add_first}, B {B}, Ns {Ns}"
        for k, p in net.named_parameters():
            g = P[k].grad.numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), g, rtol=1e-3, atol=1e-4 * max(np.abs(g).max(), 1e-12), err_msg=msg + " " + k)
        if add_dim:
            g = add_t.grad.numpy()
            np.testing.assert_allclose(add_g.grad.cpu().numpy(), g, rtol=1e-3, atol=1e-4 * max(np.abs(g).max(), 1e-12), err_msg=msg + " d_add")


def test_random_warp_net_shapes_against_torch(dev):
    """Seeded sweep for WarpFieldNet: width, position encoder, pose columns, samples per ray - fused inference (per-ray pose
    fold from 8 samples per ray), training forward and the gradients of linear1 / linear2 / the pose rows against torch."""
    from smpl_nerf_amd.nets import WarpFieldNet
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(4242)
    for case in range(12):
        width = int(rng.choice([16, 64, 100, 128, 200, 256]))
        pL, pid = int(rng.integers(0, 11)), int(rng.integers(0, 2))
        if pL == 0:
            pid = 1
        qdim = int(rng.choice([2, 40, 69]))
        B, Ns = int(rng.integers(3, 40)), int(rng.choice([5, 9, 64]))
        pdim = 3 * (pid + 2 * pL)
        params = syn.make_warp_field_params(3000 + case, positions_dim=pdim, pose_dim=qdim, width=width, out_scale=0.5)
        net = WarpFieldNet(8, width, pdim, qdim)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        net = net.to(dev)
        x = rng.uniform(-1.5, 1.5, (B, Ns, 3)).astype(F32)
        o = rng.normal(size=(B, 3)).astype(F32)
        gout = rng.normal(size=(B * Ns, 3)).astype(F32)
        P = R.tparams(params)
        pose_t = torch.from_numpy(rng.uniform(-1, 1, (B, qdim)).astype(F32)).requires_grad_(True)
        rows = torch.cat([R.posenc(torch.from_numpy(x), pL, pid), pose_t[:, None, :].expand(B, Ns, qdim)], -1).reshape(B * Ns, -1)
        ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(rows, P["linear1.weight"], P["linear1.bias"])),
                                         P["linear2.weight"], P["linear2.bias"])
        (ref * torch.from_numpy(gout)).sum().backward()
        pe = PositionalEncoder(pL, pid)
        msg = f"case {case}: width {width}, pos ({pL}, {pid}), pose {qdim}, B {B}, Ns {Ns}"
        with torch.no_grad():
            inf = net.forward_fused(T(x, dev), T(pose_t.detach().numpy(), dev), T(o, dev), Ns, pe)
        pose_g = T(pose_t.detach().numpy(), dev).requires_grad_(True)
        trn = net.forward_fused(T(x, dev), pose_g, T(o, dev), Ns, pe)
        tol = 2e-5 * max(1.0, ref.detach().abs().max().item())
        np.testing.assert_allclose(inf[0].cpu().numpy(), ref.detach().numpy(), rtol=0, atol=tol, err_msg=msg)
        np.testing.assert_allclose(trn[0].detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=tol, err_msg=msg)
        np.testing.assert_allclose(inf[1].cpu().numpy(), x.reshape(-1, 3) + ref.detach().numpy(), rtol=0, atol=2 * tol, err_msg=msg)
        (trn[0] * T(gout, dev)).sum().backward()
        for k, p in net.named_parameters():
            g = P[k].grad.numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), g, rtol=1e-3, atol=1e-4 * max(np.abs(g).max(), 1e-12), err_msg=msg + " " + k)
        g = pose_t.grad.numpy()
        np.testing.assert_allclose(pose_g.grad.cpu().numpy(), g, rtol=1e-3, atol=1e-4 * max(np.abs(g).max(), 1e-12), err_msg=msg + " d_pose")


def test_split_precision_on_other_widths_runs_exact_fp32(dev):
    """The split-precision kernels exist for width 256; a pipeline of narrower nets with set_precision("bf16x6") runs the
    exact-fp32 kernels in forward() and in the one-call render_rays - same results as precision "fp32", no error."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import NerfPipeline
    nets = []
    for seed in (5, 6):
        params = syn.make_render_ray_net_params(seed, 30.0, 10.0, n_layers=4, width=96, skips=(1,))
        m = RenderRayNet(4, 96, 60, 24, skips=[1])
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        nets.append(m.to(dev).eval())
    data = [T(a[:300], dev) for a in syn.frame_batch(32, 32, seed=2)]
    outs = {}
    for prec in ("fp32", "bf16x6", "f16x3"):
        pipe = NerfPipeline(nets[0], nets[1], O.Args(), PositionalEncoder(10, 0), PositionalEncoder(4, 0)).set_precision(prec)
        with torch.no_grad():
            outs[prec] = (pipe(data), pipe.render_rays(data))
    for prec in ("bf16x6", "f16x3"):
        for a, b in zip(outs["fp32"][0] + outs["fp32"][1], outs[prec][0] + outs[prec][1]):
            assert torch.equal(a, b)
