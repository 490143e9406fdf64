"""GPU tests added in round 2: full-size runs of BASELINE configs[3]/[4], guard bands around every caller-allocated
output and scratch buffer of the C-ABI, input gradients (encoded rows, per-ray additional inputs, additional inputs
together with position/direction gradients), the multi-rank data-parallel path through the real pipeline, bench.py's
self-launch, and a measured ReLU-mask-flip count behind the gradient tolerances."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import torch_ref as R
from oracle import nerf_oracle as O
from smpl_nerf_amd import synthetic as syn
from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


def _net(dev, params, **kw):
    from smpl_nerf_amd.nets import RenderRayNet
    m = RenderRayNet(kw.get("n_layers", 8), kw.get("width", 256), 60, 24, kw.get("additional_input_dim", 0),
                     skips=list(kw.get("skips", (4,))))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    if kw.get("precision"):
        m.precision = kw["precision"]
    return m.to(dev)


def _encoders():
    from smpl_nerf_amd.ops import PositionalEncoder
    return PositionalEncoder(10, 0), PositionalEncoder(4, 0)


def _nerf_pipeline(dev, precision="fp32", **argkw):
    from smpl_nerf_amd.pipelines import NerfPipeline
    pc, pf = syn.make_scene_nets(101)
    pipe = NerfPipeline(_net(dev, pc), _net(dev, pf), O.Args(**argkw), *_encoders())
    return pipe.set_precision(precision), (pc, pf)


def _smpl_pipeline(dev, precision="fp32", **argkw):
    from smpl_nerf_amd.nets import WarpFieldNet
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import SmplNerfPipeline
    pc, pf = syn.make_scene_nets(101)
    pw = syn.make_warp_field_params(103, out_scale=0.3)
    mw = WarpFieldNet(8, 256, 60, 40)
    mw.load_state_dict({k: torch.from_numpy(v) for k, v in pw.items()})
    pipe = SmplNerfPipeline(_net(dev, pc), _net(dev, pf), mw.to(dev), O.Args(**argkw), *_encoders(), PositionalEncoder(10, 0))
    return pipe.set_precision(precision), (pc, pf, pw)


def _av_pipeline(dev, precision="fp32", n_poses=60, **argkw):
    from smpl_nerf_amd.nets import AppendVerticesNet
    from smpl_nerf_amd.pipelines import AppendVerticesPipeline
    from smpl_nerf_amd.synthetic_smpl import IndexPoseEstimator, LinearBodyModel
    params = [syn.make_append_vertices_params(s) for s in (201, 202)]
    nets = []
    for p in params:
        m = AppendVerticesNet(8, 256, 60, 24, 6890, additional_input_layers=1, skips=[4])
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        nets.append(m.to(dev))
    est = IndexPoseEstimator(torch.from_numpy(syn.human_poses((41, 38), 0, 60, n_poses)), torch.zeros(1, 10)).to(dev)
    pipe = AppendVerticesPipeline(nets[0], nets[1], est, LinearBodyModel(seed=3).to(dev), O.Args(**argkw), *_encoders())
    return pipe.set_precision(precision), params


def _invariance(pipe, full, R_, dev, nout):
    """One call == 7 ragged chunks == a permuted ray order, bit for bit (rays are independent)."""
    with torch.no_grad():
        ref = [N(t) for t in pipe(full)]
        cuts = [0, 1, 130, 9000, 9001, 30000, 50011, R_]
        parts = [pipe([t[a:b] for t in full]) for a, b in zip(cuts[:-1], cuts[1:])]
        for k in range(nout):
            assert np.array_equal(np.concatenate([N(p[k]) for p in parts]), ref[k]), k
        del parts
        perm = torch.from_numpy(np.random.default_rng(5).permutation(R_)).to(dev)
        shuf = pipe([t[perm] for t in full])
        for k in range(nout):
            assert np.array_equal(N(shuf[k]), ref[k][N(perm)]), k
    return ref


# ------------------------------------------------------------------------------------------ BASELINE configs[3]
@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
def test_smpl_nerf_full_256_frame_60_poses(dev, prec):
    """configs[3]: model_type=smpl_nerf at 256 x 256 (65 536 rays, 16.8 M MLP evaluations + the warp field) with the 60
    arm poses of the data set mixed ray by ray in one batch: chunk / permutation invariance bit for bit, sortedness and
    range properties, and a 1/16 subset against the CPU oracle to the north-star tolerance."""
    from smpl_nerf_amd.ops import uniform_u
    pipe, (pc, pf, pw) = _smpl_pipeline(dev, prec)
    data = syn.frame_batch(256, 256, phi=12.0, theta=-35.0, seed=23)
    R_ = data[0].shape[0]
    assert R_ == 65536
    poses = syn.human_poses((41, 38), 0, 60, 60)
    goal = poses[(np.arange(R_) * 7919) % 60].astype(F32)
    dn = list(data[:4]) + [goal, data[4]]
    full = [T(a, dev) for a in dn]
    rgb, rgb_fine, warp_f, pts_f, warped_f, dens = _invariance(pipe, full, R_, dev, 6)
    assert np.allclose(warped_f, pts_f + warp_f, atol=1e-6)
    o, d = data[1], data[2]
    zf = ((pts_f - o[:, None, :]) * d[:, None, :]).sum(-1) / (d * d).sum(-1)[:, None]
    assert np.all(np.diff(zf, axis=1) >= -1e-4)
    assert rgb.min() >= 0.0 and rgb.max() <= 1.0 + 1e-6 and rgb_fine.min() >= 0.0 and rgb_fine.max() <= 1.0 + 1e-6
    assert dens.min() >= 0.0 and dens.max() <= 1.0
    sub = np.arange(0, R_, 16)
    enc = O.PositionalEncoder
    want = O.smpl_nerf_pipeline_forward(pc, pf, pw, O.Args(u=N(uniform_u(128, dev))), enc(10, 0), enc(4, 0), enc(10, 0),
                                        [a[sub] for a in dn])
    assert maxabs(rgb[sub], want[0]) <= 1e-4 and maxabs(rgb_fine[sub], want[1]) <= 1e-4
    assert np.mean(np.abs(warp_f[sub] - want[2]) > 1e-4) <= 0.02


# ------------------------------------------------------------------------------------------ BASELINE configs[4]
@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
def test_append_vertices_full_256_frame_60_poses(dev, prec):
    """configs[4]: model_type=append_vertices at 256 x 256 with per-ray image indices into 60 poses (estimator -> body
    model -> 6890 vertices per ray, of which the net reads the first 60 floats - quirk Q7), coarse + fine."""
    from smpl_nerf_amd.ops import uniform_u
    from smpl_nerf_amd.synthetic_smpl import LinearBodyModel
    pipe, params = _av_pipeline(dev, prec, run_fine=1)
    data = syn.frame_batch(256, 256, phi=-8.0, theta=50.0, seed=29)
    R_ = data[0].shape[0]
    images = ((np.arange(R_) * 104729) % 60).astype(np.int64)
    full = [T(a, dev) for a in data[:4]] + [torch.from_numpy(images).to(dev), T(data[4], dev)]
    rgb, rgb_fine, pts_f, dens = _invariance(pipe, full, R_, dev, 4)
    assert rgb.min() >= 0.0 and rgb.max() <= 1.0 + 1e-6 and rgb_fine.min() >= 0.0 and rgb_fine.max() <= 1.0 + 1e-6
    assert dens.min() >= 0.0 and dens.max() <= 1.0 and pts_f.shape == (R_, 192, 3)
    sub = np.arange(0, R_, 16)
    verts = LinearBodyModel(seed=3)(body_pose=torch.from_numpy(syn.human_poses((41, 38), 0, 60, 60)[images[sub]])).vertices.numpy()
    want = O.append_vertices_pipeline_forward(params[0], params[1], verts, O.Args(u=N(uniform_u(128, dev))),
                                              O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0), [a[sub] for a in data])
    assert maxabs(rgb[sub], want[0]) <= 1e-4 and maxabs(rgb_fine[sub], want[1]) <= 1e-4


# ------------------------------------------------------------------------------------------ guard bands
class _GuardBands:
    """While active, every CUDA tensor the host side allocates through torch.empty / zeros / empty_like - i.e. every
    caller-allocated output and scratch buffer handed to the C-ABI - sits inside a larger allocation whose 1 KiB bands on
    both sides carry a byte pattern; check() asserts that no kernel touched a band."""
    PAD, FILL = 1024, 0xA5

    def __init__(self):
        self.records = []
        self._real = {}

    def _guarded(self, shape, dtype, device):
        dtype = dtype or torch.float32
        esize = torch.empty((), dtype=dtype).element_size()
        nbytes = int(np.prod(shape, dtype=np.int64)) * esize if len(shape) else esize
        buf = self._real["empty"](nbytes + 2 * self.PAD, dtype=torch.uint8, device=device)
        buf.fill_(self.FILL)
        self.records.append((buf, nbytes))
        return buf[self.PAD:self.PAD + nbytes].view(dtype).view(tuple(shape))

    def __enter__(self):
        g = self

        def shape_of(size):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                return tuple(size[0])
            return tuple(size)

        def empty(*size, **kw):
            dv = kw.get("device")
            if dv is None or torch.device(dv).type != "cuda" or kw.get("out") is not None:
                return g._real["empty"](*size, **kw)
            return g._guarded(shape_of(size), kw.get("dtype"), dv)

        def zeros(*size, **kw):
            dv = kw.get("device")
            if dv is None or torch.device(dv).type != "cuda" or kw.get("out") is not None:
                return g._real["zeros"](*size, **kw)
            return g._guarded(shape_of(size), kw.get("dtype"), dv).zero_()

        def empty_like(t, **kw):
            if not t.is_cuda or kw:
                return g._real["empty_like"](t, **kw)
            return g._guarded(tuple(t.shape), t.dtype, t.device)

        for name, fn in (("empty", empty), ("zeros", zeros), ("empty_like", empty_like)):
            self._real[name] = getattr(torch, name)
            setattr(torch, name, fn)
        return self

    def __exit__(self, *exc):
        for name, fn in self._real.items():
            setattr(torch, name, fn)
        return False

    def check(self):
        torch.cuda.synchronize()
        assert self.records
        for buf, nbytes in self.records:
            lo, hi = buf[:self.PAD], buf[self.PAD + nbytes:]
            assert bool((lo == self.FILL).all()) and bool((hi == self.FILL).all()), (buf.numel(), nbytes)
        n = len(self.records)
        self.records = []
        return n


@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
def test_guard_bands_around_every_output_and_scratch_buffer(dev, prec):
    """Ragged sizes (n = 1, 127, 129, 5003 samples; B = 1, 37 rays) through every entry point: stand-alone ops, the
    pipelines (five launches and the single-call render), and full training steps (act / dy / gpart scratch sized by
    snerf_mlp_train_sizes / snerf_warp_train_sizes, flat gradients, input gradients)."""
    from smpl_nerf_amd import ops
    from smpl_nerf_amd.raygen import RayGenerator
    rng = np.random.default_rng(8)
    pe, de = _encoders()
    total = 0
    with _GuardBands() as g:
        # ---- stand-alone ops
        for n in (1, 127, 129, 5003):
            x = T(rng.uniform(-3, 3, (n, 3)).astype(F32), dev)
            pe.encode(x)
            a = T(np.sort(rng.normal(size=(n, 63)).astype(F32), -1), dev)
            v = T(rng.normal(size=(n, 17)).astype(F32), dev)
            ops.searchsorted(a, v, side="right")
            ops.searchsorted(a[:1].contiguous(), v, side="left")
        for B, Ns in ((1, 1), (1, 64), (37, 192), (37, 100), (5, 1025)):
            raw = T(rng.normal(0, 2, (B, Ns, 4)).astype(F32), dev)
            z = T(np.sort(rng.uniform(1, 4, (B, Ns)).astype(F32), -1), dev)
            d = T(rng.normal(size=(B, 3)).astype(F32), dev)
            ops.composite(raw, z, d, True)
            ds = T(rng.normal(size=(B, Ns, 3)).astype(F32), dev).requires_grad_(True)
            rr = raw.clone().requires_grad_(True)
            rgb, _, _ = ops.composite(rr, z, ds, False)
            rgb.sum().backward()
        for B, nc, nf in ((1, 64, 128), (37, 64, 128), (37, 3, 1), (5, 100, 77)):
            z = T(np.sort(rng.uniform(1, 4, (B, nc)).astype(F32), -1), dev)
            w = T(rng.uniform(0, 1, (B, nc)).astype(F32), dev)
            o, d = T(rng.normal(size=(B, 3)).astype(F32), dev), T(rng.normal(size=(B, 3)).astype(F32), dev)
            ops.hierarchical_samples(o, d, z, w, nf, want_inds=True, want_samples=True)
            ops.hierarchical_samples(o, d, z, w, nf, want_inds=True, strict=True)      # snerf_sample_pdf_strict_f32
            ops.sample_pdf(T(np.sort(rng.uniform(1, 4, (B, nc)).astype(F32), -1), dev), w[:, :nc - 1].contiguous(),
                           O.Args(number_fine_samples=nf))
            ops.sample_pdf(T(np.sort(rng.uniform(1, 4, (B, nc)).astype(F32), -1), dev), w[:, :nc - 1].contiguous(),
                           O.Args(number_fine_samples=nf, strict_cumsum=1))
        gen = RayGenerator(np.stack([syn.sphere_pose(3.0, 4.0, 2.4)] * 2), 16, 24, np.pi / 3, 1.0, 4.0, 64, dev)
        gen.random_batch(37)
        gen.batch(torch.tensor([0, 16 * 24 * 2 - 1, 16 * 24 * 2, -1], device=dev), torch.zeros(4, dtype=torch.float64, device=dev))
        total += g.check()
        # ---- pipelines, inference
        data = syn.frame_batch(128, 128, phi=2.0, theta=9.0, seed=31)
        for B in (1, 37):
            sub = np.arange(B) * 401
            batch = [T(a[sub], dev) for a in data]
            pose = T(syn.human_poses()[np.arange(B) % 10], dev)
            nerf, _ = _nerf_pipeline(dev, prec)
            smpl, _ = _smpl_pipeline(dev, prec)
            av, _ = _av_pipeline(dev, prec, n_poses=10, run_fine=1)
            with torch.no_grad():
                nerf(batch)
                nerf.render_rays(batch)
                smpl(batch[:4] + [pose, batch[4]])
                smpl.render_rays(batch[:4] + [pose, batch[4]])
                av(batch[:4] + [torch.arange(B, device=dev) % 10, batch[4]])
            total += g.check()
            # ---- training steps (forward with saved activations, backward, flat gradients)
            for pipe, b in ((nerf, batch), (smpl, batch[:4] + [pose, batch[4]])):
                for p in pipe.parameters():
                    p.requires_grad_(True)
                out = pipe(b)
                (torch.nn.functional.mse_loss(out[0], b[-1]) + torch.nn.functional.mse_loss(out[1], b[-1])).backward()
                total += g.check()
        # ---- encoded rows with input gradient, additional inputs with input gradient
        for n in (1, 127, 129, 5003):
            net = _net(dev, syn.make_scene_nets(101)[0], precision=prec)
            rows = T(rng.normal(size=(n, 84)).astype(F32), dev).requires_grad_(True)
            net(rows).sum().backward()
            total += g.check()
    assert total > 200


# ------------------------------------------------------------------------------------------ input gradients (ADVICE)
def test_encoded_rows_receive_their_gradient(dev):
    """RenderRayNet.forward(x) is the reference's module call; like nn.Module it must propagate into x (the reference's
    smpl_nerf / dynamic pipelines train upstream modules through model(inputs), models/render_ray_net.py:42-61)."""
    g2, g7 = load_golden("g2_mlp.npz"), load_golden("g7_grads.npz")
    params = syn.make_render_ray_net_params(11, 30.0, 10.0, skips=(4,))
    net = _net(dev, params)
    x = T(g2["inputs"], dev).requires_grad_(True)
    (net(x) * T(g7["m_gout"], dev)).sum().backward()
    P = R.tparams(params)
    xc = torch.from_numpy(g2["inputs"]).clone().requires_grad_(True)
    (R.render_ray_net(P, xc) * torch.from_numpy(g7["m_gout"])).sum().backward()
    ref = xc.grad.numpy()
    assert x.grad is not None and x.grad.shape == x.shape
    np.testing.assert_allclose(N(x.grad), ref, rtol=5e-4, atol=5e-5 * np.abs(ref).max())
    for k, p in net.named_parameters():                     # parameter gradients unchanged by the extra output
        r = P[k].grad.numpy()
        np.testing.assert_allclose(N(p.grad), r, rtol=5e-4, atol=5e-5 * np.abs(r).max())
    # no parameter gradients wanted, input gradient only
    net2 = _net(dev, params)
    for p in net2.parameters():
        p.requires_grad_(False)
    x2 = T(g2["inputs"], dev).requires_grad_(True)
    (net2(x2) * T(g7["m_gout"], dev)).sum().backward()
    np.testing.assert_allclose(N(x2.grad), ref, rtol=5e-4, atol=5e-5 * np.abs(ref).max())


def test_append_vertices_estimator_receives_gradient(dev):
    """AppendVerticesSolver optimises the pose estimator in its own parameter group: the gradient must reach it through
    smpl_model and the first positions_dim vertex floats (models/append_vertices_pipeline.py:30-58)."""
    from smpl_nerf_amd.synthetic_smpl import LinearBodyModel
    pipe, params = _av_pipeline(dev, n_poses=10, run_fine=1)
    pipe.smpl_estimator.goal_poses.requires_grad_(True)
    g = load_golden("g9_append_vertices.npz")
    data = syn.frame_batch(128, 128, phi=3.0, theta=-10.0, seed=11)
    sub, images = g["sub"], g["images"]
    d = [T(a[sub], dev) for a in data[:4]] + [torch.from_numpy(images).to(dev), T(data[4][sub], dev)]
    out = pipe(d)
    loss = torch.nn.functional.mse_loss(out[0], d[-1]) + torch.nn.functional.mse_loss(out[1], d[-1])
    loss.backward()
    got = N(pipe.smpl_estimator.goal_poses.grad)
    # torch reference of the same graph: poses -> linear body model -> first 60 vertex floats -> nets -> compositing
    poses = torch.from_numpy(syn.human_poses((41, 38), 0, 60, 10)).clone().requires_grad_(True)
    verts = LinearBodyModel(seed=3)(body_pose=poses[torch.from_numpy(images)]).vertices
    B = len(sub)
    ray_in = verts.reshape(B, -1)[:, :60]
    dr = torch.from_numpy(data[2][sub])
    denc = R.posenc(dr / torch.norm(dr, dim=-1, keepdim=True), 4, 0)
    total = 0.0
    z, w_c = torch.from_numpy(data[3][sub]), None
    for P_np, run in ((params[0], "coarse"), (params[1], "fine")):
        P = R.tparams({k: v for k, v in P_np.items() if not k.startswith("vertices_net")}, requires_grad=False)
        if run == "fine":
            from smpl_nerf_amd.ops import uniform_u
            zf, _ = O.fine_sampling(data[1][sub], data[2][sub], data[3][sub], w_c, 128, u=N(uniform_u(128, dev)))
            z = torch.from_numpy(zf)
        Ns = z.shape[1]
        rows = torch.cat([ray_in[:, None, :].expand(B, Ns, 60), denc[:, None, :].expand(B, Ns, 24)], -1)
        raw = R.render_ray_net(P, rows.reshape(B * Ns, -1)).view(B, Ns, 4)
        rgb, w, _ = R.raw2outputs(raw, z, dr[:, None, :].expand(B, Ns, 3), 0)
        w_c = w.detach().numpy()
        total = total + torch.nn.functional.mse_loss(rgb, torch.from_numpy(data[4][sub]))
    total.backward()
    ref = poses.grad.numpy()
    assert abs(loss.item() - total.item()) <= 2e-6
    assert np.abs(ref).max() > 0
    assert np.linalg.norm(got - ref) <= 2e-3 * np.linalg.norm(ref)


@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("add_first", [False, True])
def test_additional_inputs_together_with_position_and_direction_gradients(dev, prec, add_first):
    """A pose-conditioned net (additional_input_dim > 0) fed by positions / per-sample directions that carry gradient:
    d x and d dir from the dgrad kernel, d additional from the stored d Y, parameter gradients - all at once (DESIGN
    section 9 listed this combination as not built), against the pinned torch reference."""
    rng = np.random.default_rng(17)
    add_dim, B, Ns = 5, 29, 7
    kw = dict(n_layers=8, skips=(3, 6), additional_input_dim=add_dim)
    params = syn.make_scene_net_params(611, add_first=add_first, **kw)
    net = _net(dev, params, precision=prec, **kw)
    pts = rng.uniform(-2, 2, (B, Ns, 3)).astype(F32)
    sd = rng.normal(size=(B, Ns, 3)).astype(F32)
    add = rng.uniform(-1, 1, (B, add_dim)).astype(F32)
    gout = rng.normal(size=(B * Ns, 4)).astype(F32)
    x, d, a = (T(v, dev).requires_grad_(True) for v in (pts, sd, add))
    raw = net.forward_fused(x, d, Ns, *_encoders(), additional=a, add_first=add_first)
    (raw * T(gout, dev)).sum().backward()
    P = R.tparams(params)
    xc, dc, ac = (torch.from_numpy(v).clone().requires_grad_(True) for v in (pts, sd, add))
    dn = dc / torch.norm(dc, dim=-1, keepdim=True)
    ae = ac[:, None, :].expand(B, Ns, add_dim)
    cols = [ae, R.posenc(xc, 10, 0)] if add_first else [R.posenc(xc, 10, 0), ae]
    rows = torch.cat(cols + [R.posenc(dn, 4, 0)], -1).view(B * Ns, -1)
    ref_raw = R.render_ray_net(P, rows, **kw)
    (ref_raw * torch.from_numpy(gout)).sum().backward()
    np.testing.assert_allclose(N(raw), ref_raw.detach().numpy(), rtol=0, atol=2e-4 * float(ref_raw.detach().abs().max()))
    for got, ref, tag in ((x.grad, xc.grad, "x"), (d.grad, dc.grad, "d"), (a.grad, ac.grad, "add")):
        ref = ref.numpy().astype(np.float64)
        assert got is not None, tag
        assert np.linalg.norm(N(got) - ref) <= 2e-3 * np.linalg.norm(ref), (tag, np.linalg.norm(N(got) - ref) / np.linalg.norm(ref))
    for k, p in net.named_parameters():
        ref = P[k].grad.numpy().astype(np.float64)
        assert np.linalg.norm(N(p.grad) - ref) <= 2e-3 * np.linalg.norm(ref), k


# ------------------------------------------------------------------------------------------ ReLU-mask flips, measured
def _relu_masks_from_act(act, n, nh=7, width=256):
    """[(nh + 2), n, width] booleans (output > 0) of layer 0 .. nh and directional_net[0], from the saved activation
    tile-rows of a training forward (mlp_plan.h TrainLayout; default encoders: 4 + 2 encoder tile-rows)."""
    rows = act.size // (n * 16)
    a = act[:rows * n * 16].reshape(rows, n, 16)
    t_w, t_d = width // 16, width // 32
    x1 = 4 + 2
    h2 = x1 + (nh + 2) * t_w + t_d
    out = []
    for idx in range(nh + 2):
        r0, nt = (x1 + idx * t_w, t_w) if idx <= nh else (h2, t_d)
        # tile-row t holds features 16 t + 4 g + r at position 4 g + r
        out.append(np.transpose(a[r0:r0 + nt], (1, 0, 2)).reshape(n, nt * 16) > 0)
    return out


@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
def test_relu_mask_flips_are_counted_and_everything_else_is_tight(dev, prec):
    """The ragged 5003-sample gradient test of test_gpu_grad.py allows outliers "because a handful of pre-activations sit
    within round-off of zero".  Here that is measured: the ReLU masks of the HIP training forward are compared with the
    torch fp32 reference's, sample by sample; the samples with a flipped mask are counted (a handful in 10 M
    activations) and taken out of the loss on both sides - the remaining gradient must then agree tightly, element by
    element, with no outlier allowance."""
    rng = np.random.default_rng(77)
    params = syn.make_scene_nets(101)[0]
    B, Ns = 5003 // 7 + 1, 7
    n = B * Ns
    pts = rng.uniform(-2, 2, (B, Ns, 3)).astype(F32)
    dray = rng.normal(size=(B, 3)).astype(F32)
    gout = rng.normal(size=(n, 4)).astype(F32)
    net = _net(dev, params, precision=prec)
    net.activation_budget_bytes = 1 << 40          # the stored form: the activation buffer is read below
    raw = net.forward_fused(T(pts, dev), T(dray, dev), Ns, *_encoders())
    masks = _relu_masks_from_act(N(raw.grad_fn.act), n)
    # torch reference activations
    P = R.tparams(params)
    dn = torch.from_numpy(dray)[:, None, :].expand(B, Ns, 3)
    dn = dn / torch.norm(dn, dim=-1, keepdim=True)
    xr = torch.cat([R.posenc(torch.from_numpy(pts), 10, 0), R.posenc(dn, 4, 0)], -1).view(n, -1)
    with torch.no_grad():
        lin = lambda v, k: torch.nn.functional.linear(v, P[k + ".weight"], P[k + ".bias"])
        pp, dd = xr[:, :60], xr[:, 60:]
        o = torch.relu(lin(pp, "positions_pose_input"))
        ref_masks = [o.numpy() > 0]
        for i in range(7):
            o = torch.relu(lin(torch.cat([o, pp], -1) if i == 4 else o, f"positional_net.{i}"))
            ref_masks.append(o.numpy() > 0)
        o = lin(o, "additional_linear_layer")
        h = torch.relu(lin(lin(torch.cat([o, dd], -1), "directional_input"), "directional_net.0"))
        ref_masks.append(h.numpy() > 0)
    flipped = np.zeros(n, bool)
    n_flips = 0
    for m, r in zip(masks, ref_masks):
        diff = m != r
        n_flips += int(diff.sum())
        flipped |= diff.any(1)
    n_act = sum(m.size for m in masks)
    print(f"[{prec}] ReLU mask flips: {n_flips} of {n_act} activations, {int(flipped.sum())} of {n} samples affected")
    assert n_flips <= 2e-5 * n_act and flipped.sum() <= 0.01 * n
    gout[flipped] = 0.0
    (raw * T(gout, dev)).sum().backward()
    (R.render_ray_net(P, xr) * torch.from_numpy(gout)).sum().backward()
    for k, p in net.named_parameters():
        ref = P[k].grad.numpy().astype(np.float64)
        got = N(p.grad).astype(np.float64)
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max(), err_msg=k)
        assert np.linalg.norm(got - ref) <= 2e-4 * np.linalg.norm(ref), (k, np.linalg.norm(got - ref) / np.linalg.norm(ref))


# ------------------------------------------------------------------------------------------ multi-rank path on the GPU
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, backend, prec, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    ndev = torch.cuda.device_count()
    dv = torch.device("cuda", rank % ndev)
    torch.cuda.set_device(dv)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dv)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smpl_nerf_amd import dist as sd
        from smpl_nerf_amd.trainer import DataParallelTrainer
        pipe, _ = _nerf_pipeline(dv, prec)
        if rank:                                            # a deliberately different replica: the trainer must fix it
            with torch.no_grad():
                for p in pipe.parameters():
                    p.mul_(1.5)
        models = [pipe.model_coarse, pipe.model_fine]
        for p in pipe.parameters():
            p.requires_grad_(True)
        tr = DataParallelTrainer(pipe, models, lr=1e-4)
        data = syn.frame_batch(128, 128, phi=4.0, theta=-6.0, seed=41)
        sub = np.arange(0, 16384, 16)                       # 1024 rays, 512 per rank
        b, e = sd.shard_range(len(sub), world, rank)
        batch = [T(a[sub][b:e], dv) for a in data]
        tr._arm_grad_sinks()
        out = pipe(batch)
        loss = tr.loss(out[0], out[1], batch[-1])
        loss.backward()
        sinks = sum(int(p.grad.data_ptr() == v.data_ptr()) for p, v in zip(tr.params, tr._views))
        tr.sync_gradients()
        if rank == 0:
            q.put(([N(p.grad) for p in tr.params], loss.item(), sinks, len(tr.params)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_two_rank_data_parallel_gradients_equal_the_single_rank_gradient(dev, prec):
    """N-rank gradients (mean over ranks of the flat buffer the HIP backward wrote, one all-reduce) equal the
    1-rank gradient of the concatenated batch, through the real NerfPipeline + HIP backward.  RCCL ("nccl") when two
    GPUs are visible; on a 1-GPU box both ranks share the device and the group runs on gloo."""
    from smpl_nerf_amd.trainer import DataParallelTrainer
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, backend, prec, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, loss0, sinks, nparams = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sinks == nparams                      # the backward kernels wrote straight into the flat all-reduce buffer
    pipe, _ = _nerf_pipeline(dev, prec)
    for p in pipe.parameters():
        p.requires_grad_(True)
    tr = DataParallelTrainer(pipe, [pipe.model_coarse, pipe.model_fine], lr=1e-4)
    data = syn.frame_batch(128, 128, phi=4.0, theta=-6.0, seed=41)
    sub = np.arange(0, 16384, 16)
    batch = [T(a[sub], dev) for a in data]
    out = pipe(batch)
    tr.loss(out[0], out[1], batch[-1]).backward()
    for g, p in zip(got, tr.params):
        ref = N(p.grad).astype(np.float64)
        np.testing.assert_allclose(g, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())


def test_bench_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE) starts two ranks itself and prints one line with
    n_gpus = 2 (gloo when only one GPU is visible)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--cpu-rays", "0", "--train-rays", "512", "--train-steps", "2", "--no-alt", "--no-pmc"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["dtype"] == "f32"
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-12
    # the N > 1 record proves itself: which ranks on which devices, what the backend saw, what the one
    # collective of the training step moved and how long it took
    assert [p["rank"] for p in line["per_rank"]] == [0, 1] and all(p["value"] > 0 and p["device"] for p in line["per_rank"])
    assert abs(line["value"] - 2 * 2 * 16384 * 256 / (line["ms_per_step"] * 2e-3)) <= 1e-6 * line["value"]
    assert line["collective"]["world_size_seen_by_backend"] == 2 and line["collective"]["bytes"] == 0
    c = line["train"]["collective"]
    assert c["world_size_seen_by_backend"] == 2 and c["bytes"] == 4 * 1220872 and c["allreduce_calls"] == 2
    assert c["allreduce_ms_per_step"] > 0 and c["broadcast_ms"] > 0 and c["backend"] in ("gloo", "nccl (RCCL)")
    assert [p["rank"] for p in line["train"]["per_rank"]] == [0, 1]


def test_bench_strong_scaling_mode_assembles_one_frame(dev):
    """`bench.py --gpus 2 --scaling strong`: ONE 128 x 128 frame split by rows over the ranks (dist.shard_rays), the rendered
    rows all-gathered into the frame on every rank inside the timed region (dist.gather_rows): the assembled frame
    equals the single-rank render bit for bit (rays are independent), value counts the frame once."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--steps", "2",
                        "--warmup", "1", "--cpu-rays", "0", "--train-rays", "0", "--no-alt", "--no-pmc"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert sum(p["rays_per_step"] for p in line["per_rank"]) == 16384
    assert abs(line["value"] - 16384 * 256 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    c = line["collective"]
    assert c["bytes"] == 16384 * 3 * 4 and c["strong_frame_max_abs_diff_vs_single_rank_render"] == 0.0


# ------------------------------------------------------------------------------------------ f-3: the reference's files
def test_reference_checkpoints_render_on_the_hip_path(dev):
    """Weights the reference saved (utils.save_run -> tests/golden/io_fixture/model_*.pt) loaded through io.load_run give
    the outputs the reference's own modules computed from them (expect.npz), on encoded rows."""
    from smpl_nerf_amd import io as sio
    from smpl_nerf_amd.nets import RenderRayNet, WarpFieldNet
    fix = os.path.join(ROOT, "tests", "golden", "io_fixture")
    e = dict(np.load(os.path.join(fix, "expect.npz")))
    mc, mf = (RenderRayNet(n_layers=2, width=128, positions_dim=60, directions_dim=24, skips=[0]) for _ in range(2))
    mw = WarpFieldNet(8, 128, 60, 40)
    sio.load_run(fix, [mc, mf, mw], ["model_coarse.pt", "model_fine.pt", "model_warp_field.pt"])
    with torch.no_grad():
        oc = mc.to(dev)(T(e["rows"], dev))
        of = mf.to(dev)(T(e["rows"], dev))
        ow = mw.to(dev)(T(e["warp_rows"], dev))
    assert maxabs(N(oc), e["out_coarse"]) <= 2e-6 and maxabs(N(of), e["out_fine"]) <= 2e-6
    assert maxabs(N(ow), e["out_warp"]) <= 2e-6


# ------------------------------------------------------------------------------------------ input gradients, other encoders
@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("pos,dirs", [((10, 1), (4, 1)), ((16, 0), (8, 1)), ((4, 0), (2, 0)), ((10, 0), (4, 0))])
def test_position_and_direction_gradients_for_other_encoders(dev, prec, pos, dirs):
    """d x / d dir through encoders other than the default L = 10 / 4: identity columns (63 / 27 inputs), more
    frequencies (the wide dgrad variant, 8 encoder k-blocks; split precisions route their backward to it) and fewer
    (padded encoder tiles).  Against the pinned torch reference: inputs and every parameter."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(23)
    (pL, pid), (dL, did) = pos, dirs
    pdim, ddim = 3 * (pid + 2 * pL), 3 * (did + 2 * dL)
    kw = dict(n_layers=6, skips=(2,), positions_dim=pdim, directions_dim=ddim)
    params = syn.make_render_ray_net_params(901, 5.0, 3.0, **kw)
    # damp the high bands so that fp32 round-off in 2^15 x does not dominate the comparison
    net = RenderRayNet(6, 256, pdim, ddim, skips=[2])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net = net.to(dev)
    net.precision = prec
    B, Ns = 19, 5
    pts = rng.uniform(-1, 1, (B, Ns, 3)).astype(F32) * (2.0 ** -6 if pL > 10 else 1.0)
    sd = rng.normal(size=(B, Ns, 3)).astype(F32)
    gout = rng.normal(size=(B * Ns, 4)).astype(F32)
    x, d = T(pts, dev).requires_grad_(True), T(sd, dev).requires_grad_(True)
    raw = net.forward_fused(x, d, Ns, PositionalEncoder(pL, pid), PositionalEncoder(dL, did))
    (raw * T(gout, dev)).sum().backward()
    P = R.tparams(params)
    xc, dc = torch.from_numpy(pts).clone().requires_grad_(True), torch.from_numpy(sd).clone().requires_grad_(True)
    dn = dc / torch.norm(dc, dim=-1, keepdim=True)
    rows = torch.cat([R.posenc(xc, pL, pid), R.posenc(dn, dL, did)], -1).view(B * Ns, -1)
    ref_raw = R.render_ray_net(P, rows, **kw)
    (ref_raw * torch.from_numpy(gout)).sum().backward()
    np.testing.assert_allclose(N(raw), ref_raw.detach().numpy(), rtol=0, atol=3e-4 * float(ref_raw.detach().abs().max()))
    for got, ref, tag in ((x.grad, xc.grad, "x"), (d.grad, dc.grad, "d")):
        ref = ref.numpy().astype(np.float64)
        assert got is not None and np.linalg.norm(N(got) - ref) <= 3e-3 * np.linalg.norm(ref), (tag, np.linalg.norm(N(got) - ref) / np.linalg.norm(ref))
    for k, p in net.named_parameters():
        ref = P[k].grad.numpy().astype(np.float64)
        assert np.linalg.norm(N(p.grad) - ref) <= 3e-3 * np.linalg.norm(ref), k
