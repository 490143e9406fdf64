"""GPU parity tests: the HIP path (through the C-ABI, via the drop-in Python mirror) against the CPU
oracle on the same seeded inputs and against the golden vectors captured from the reference.

Tolerances (written next to each assert):
  * integer / index outputs: bit-exact;
  * sampler floats: bit-exact against the oracle (same fp64-sum / fp32-op contract);
  * fp32 elementwise + reductions: a few ulp;
  * rendered RGB: 1e-4 absolute, PSNR 0.01 dB (BASELINE.json north_star).
"""
import os
from itertools import product

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from smpl_nerf_amd import synthetic as syn
from conftest import load_golden

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from smpl_nerf_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


# ------------------------------------------------------------------------------------------ a6
@pytest.mark.parametrize("Ba,Bv,A,V,side", list(product([1, 100, 200], [1, 100, 200], [1, 50, 500], [1, 12, 120],
                                                        ["left", "right"])))
def test_searchsorted_reference_grid(dev, Ba, Bv, A, V, side):
    """The reference's own test grid (torchsearchsorted/test/test_searchsorted.py:27-44)."""
    from smpl_nerf_amd.ops import searchsorted
    if Ba > 1 and Bv > 1 and Ba != Bv:
        pytest.skip("skipped by the reference too")
    rng = np.random.default_rng(Ba + 3 * Bv + 5 * A + 7 * V)
    for _ in range(4):
        a = np.sort(rng.random((Ba, A), dtype=F32), axis=1)
        v = rng.random((Bv, V), dtype=F32)
        out = searchsorted(T(a, dev), T(v, dev), side=side)
        assert out.dtype == torch.long
        np.testing.assert_array_equal(N(out), O.searchsorted(a, v, side))


def test_searchsorted_out_argument_and_goldens(dev):
    from smpl_nerf_amd.ops import searchsorted
    g = load_golden("g_searchsorted.npz")
    for side in ("left", "right"):
        np.testing.assert_array_equal(N(searchsorted(T(g["a"], dev), T(g["v"], dev), side=side)), g[f"out_{side}"])
        out = torch.empty((7, 33), dtype=torch.long, device=dev)
        r = searchsorted(T(g["a2"], dev), T(g["v2"], dev), out, side=side)
        assert r is out
        np.testing.assert_array_equal(N(out), g[f"out2_{side}"])
    g4 = load_golden("g4_sampler.npz")
    np.testing.assert_array_equal(N(searchsorted(T(g4["cdf"], dev), T(g4["u"], dev), side="right")), g4["inds"])
    np.testing.assert_array_equal(N(searchsorted(T(g4["cdf"], dev), T(g4["u"][:1], dev), side="right")), g4["inds"])


def test_searchsorted_large_shapes(dev):
    """README benchmark shape class (50000x300 / 50000x1000) scaled to seconds, plus the long-row path."""
    from smpl_nerf_amd.ops import searchsorted
    rng = np.random.default_rng(0)
    a = np.sort(rng.random((5000, 300), dtype=F32), axis=1)
    v = rng.random((5000, 100), dtype=F32)
    np.testing.assert_array_equal(N(searchsorted(T(a, dev), T(v, dev), side="left")), O._searchsorted_rows(a, v, "left"))
    a = np.sort(rng.random((3, 20000), dtype=F32), axis=1)           # > LDS budget -> global bisection
    v = rng.random((3, 1000), dtype=F32)
    v[:, :10] = a[:, 5:15]
    for side in ("left", "right"):
        np.testing.assert_array_equal(N(searchsorted(T(a, dev), T(v, dev), side=side)), O.searchsorted(a, v, side))
    # size-independent property at NeRF frame size: result is the insertion point
    B = 16384
    cdf = np.sort(rng.random((B, 63), dtype=F32), axis=1)
    u = np.broadcast_to(O.linspace01(128), (B, 128)).copy()
    inds = N(searchsorted(T(cdf, dev), T(u, dev), side="right"))
    lo = np.take_along_axis(np.concatenate([np.full((B, 1), -np.inf, F32), cdf], 1), inds, 1)
    hi = np.take_along_axis(np.concatenate([cdf, np.full((B, 1), np.inf, F32)], 1), inds, 1)
    assert np.all(lo <= u) and np.all(u < hi)


# ------------------------------------------------------------------------------------------ a1
@pytest.mark.parametrize("L,ident", [(10, 0), (4, 0), (10, 1), (4, 1), (0, 1)])
def test_posenc(dev, L, ident):
    from smpl_nerf_amd.ops import PositionalEncoder
    g = load_golden("g1_posenc.npz")
    enc = PositionalEncoder(L, ident)
    assert enc.output_dim == O.PositionalEncoder(L, ident).output_dim
    out = N(enc.encode(T(g["x"], dev)))
    assert out.shape == g[f"enc_L{L}_id{ident}"].shape
    assert maxabs(out, g[f"enc_L{L}_id{ident}"]) <= 4e-7        # sin/cos at |arg| <= 2e3: <= 2 ulp of 1 each side
    x = np.random.default_rng(3).uniform(-4, 4, (1000, 77, 3)).astype(F32)   # ragged vs the 64-point tile
    out = N(enc.encode(T(x, dev)))
    assert maxabs(out, O.PositionalEncoder(L, ident).encode(x)) <= 4e-7


def test_posenc_pose_channels(dev):
    from smpl_nerf_amd.ops import PositionalEncoder
    g = load_golden("g1_posenc.npz")
    assert maxabs(N(PositionalEncoder(10, 0).encode(T(g["pose2"], dev))), g["pose2_enc_L10_id0"]) <= 4e-7


# ------------------------------------------------------------------------------------------ a4
@pytest.mark.parametrize("Ns", [1, 2, 64, 192, 100])
@pytest.mark.parametrize("wb", [0, 1])
@pytest.mark.parametrize("mode", ["ray", "smp"])
def test_raw2outputs(dev, Ns, wb, mode):
    from smpl_nerf_amd.ops import raw2outputs
    if Ns == 1 and mode == "smp":
        pytest.skip("N==1 ignores directions")
    g = load_golden("g3_raw2outputs.npz")
    raw, z = g[f"raw_N{Ns}"], g[f"z_N{Ns}"]
    B = raw.shape[0]
    if mode == "ray":
        d = T(g[f"dray_N{Ns}"], dev)[:, None, :].expand(B, Ns, 3)        # the pipeline's expanded view
    else:
        d = T(g[f"dsmp_N{Ns}"], dev)
    args = O.Args(white_background=wb)
    rgb, w, a = raw2outputs(T(raw, dev), T(z, dev), d, args)
    dn = np.broadcast_to(g[f"dray_N{Ns}"][:, None, :], (B, Ns, 3)) if mode == "ray" else g[f"dsmp_N{Ns}"]
    orgb, ow, oa = O.raw2outputs(raw, z, dn, wb)
    for got, orc, gold, tol in ((rgb, orgb, g[f"rgb_N{Ns}_wb{wb}_{mode}"], 1e-6),
                                (w, ow, g[f"weights_N{Ns}_wb{wb}_{mode}"], 5e-7),
                                (a, oa, g[f"alpha_N{Ns}_wb{wb}_{mode}"], 5e-7)):
        assert tuple(got.shape) == gold.shape
        assert maxabs(N(got), orc) <= tol        # vs oracle
        assert maxabs(N(got), gold) <= tol       # vs reference


def test_raw2outputs_noise(dev):
    from smpl_nerf_amd import ops
    g = load_golden("g3_raw2outputs.npz")
    rgb, w, a = ops.composite(T(g["raw_N64"], dev), T(g["z_N64"], dev), T(g["dray_N64"], dev), False,
                              T(g["noise_N64"], dev))
    assert maxabs(N(rgb), g["rgb_N64_noise"]) <= 1e-6
    assert maxabs(N(w), g["weights_N64_noise"]) <= 5e-7
    # sigma_noise_std > 0 draws noise (utils.py:171-173): result must differ from the noiseless one
    args = O.Args(sigma_noise_std=1.0)
    torch.manual_seed(0)
    rgb_n, _, _ = ops.raw2outputs(T(g["raw_N64"], dev), T(g["z_N64"], dev), T(g["dray_N64"], dev), args)
    assert maxabs(N(rgb_n), g["rgb_N64_wb0_ray"]) > 1e-4


def test_composite_frame_size_properties(dev):
    """Full 128x128 frame (16384 rays x 192): weights are a sub-probability vector, alpha in [0,1],
    white background completes rgb to 1 - linearity in the colours."""
    from smpl_nerf_amd import ops
    rng = np.random.default_rng(5)
    B, Ns = 16384, 192
    raw = rng.normal(0, 2, (B, Ns, 4)).astype(F32)
    z = np.sort(rng.uniform(1, 4, (B, Ns)).astype(F32), -1)
    d = rng.normal(size=(B, 3)).astype(F32)
    rgb, w, a = ops.composite(T(raw, dev), T(z, dev), T(d, dev), False)
    rgb_w, _, _ = ops.composite(T(raw, dev), T(z, dev), T(d, dev), True)
    w, a, rgb, rgb_w = N(w), N(a), N(rgb), N(rgb_w)
    assert np.all(a >= 0) and np.all(a <= 1) and np.all(w >= 0)
    acc = w.sum(-1, dtype=np.float64)
    assert np.all(acc <= 1 + 1e-5)
    np.testing.assert_allclose(rgb_w - rgb, np.broadcast_to((1 - acc)[:, None], (B, 3)), atol=2e-6)
    sub = slice(0, 512)
    orgb, ow, _ = O.raw2outputs(raw[sub], z[sub], np.broadcast_to(d[sub, None, :], (512, Ns, 3)), 0)
    assert maxabs(rgb[sub], orgb) <= 1e-6 and maxabs(w[sub], ow) <= 5e-7


# ------------------------------------------------------------------------------------------ a5
def _check_sampler(dev, o, d, z, w, nf, u=None, tot=None):
    from smpl_nerf_amd import ops
    if u is not None:
        ops._U_CACHE[(nf, str(dev))] = T(u, dev)
    try:
        r = ops.hierarchical_samples(T(o, dev), T(d, dev), T(z, dev), T(w, dev), nf, want_inds=True, want_samples=True,
                                     tot=None if tot is None else T(tot, dev))
    finally:
        ops._U_CACHE.pop((nf, str(dev)), None)
    uu = N(ops.uniform_u(nf, dev)) if u is None else u
    z_mid = (F32(0.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
    det = O.sample_pdf_detail(z_mid, w[:, 1:-1], nf, u=uu, tot=tot)
    zf, pts = O.fine_sampling(o, d, z, w, nf, u=uu, tot=tot)
    np.testing.assert_array_equal(N(r["inds"]), det["inds"])             # bit-exact indices
    np.testing.assert_array_equal(N(r["z_samples"]), det["samples"])     # bit-exact fp32
    np.testing.assert_array_equal(N(r["z_fine"]), zf)
    np.testing.assert_array_equal(N(r["pts"]), pts)
    return r


def test_sampler_golden_inputs(dev):
    g = load_golden("g4_sampler.npz")
    r = _check_sampler(dev, g["o"], g["d"], g["z"], g["w"], 128, u=g["u"][0])
    # and against the reference's own outputs, up to its sensitivity to the 1-ulp normalising sum
    assert np.mean(N(r["inds"]) != g["inds"]) <= 5e-3
    assert np.mean(np.abs(N(r["z_fine"]) - g["z_fine"]) > 5e-6) <= 5e-3
    assert np.mean(np.abs(N(r["pts"]) - g["pts_fine"]) > 2e-5) <= 5e-3


def test_sampler_strict_mode_is_the_reference_bit_for_bit_from_the_weights(dev):
    """SURVEY 8b `strict_cumsum` / north_star "sample indices bit-exact": given the same weights and the normalising sums
    as the reference's host evaluated them (the one host-dependent step, recorded in g4), the kernel's indices, samples,
    merged depths and points ARE the reference's - including the adversarial rows (all-zero, all-equal, single spike,
    mass at the ends, tiny, alternating zeros)."""
    from smpl_nerf_amd import ops
    g = load_golden("g4_sampler.npz")
    r = _check_sampler(dev, g["o"], g["d"], g["z"], g["w"], 128, u=g["u"][0], tot=g["tot"])
    np.testing.assert_array_equal(N(r["inds"]), g["inds"])
    np.testing.assert_array_equal(N(r["z_samples"]), g["z_samples"])
    np.testing.assert_array_equal(N(r["z_fine"]), g["z_fine"])
    np.testing.assert_array_equal(N(r["pts"]), g["pts_fine"])
    for nc, nf in ((16, 8), (32, 64), (64, 64), (48, 200)):
        key = f"{nc}_{nf}"
        ops._U_CACHE[(nf, str(dev))] = T(g["u_" + key], dev)          # the reference host's linspace bits
        try:
            r = ops.hierarchical_samples(T(g["o_" + key], dev), T(g["d_" + key], dev), T(g["z_" + key], dev),
                                         T(g["w_" + key], dev), nf, tot=T(g["tot_" + key], dev))
        finally:
            ops._U_CACHE.pop((nf, str(dev)), None)
        np.testing.assert_array_equal(N(r["z_fine"]), g["zf_" + key])
        np.testing.assert_array_equal(N(r["pts"]), g["pf_" + key])
    # the literal sample_pdf(bins, weights, args) convention with args.strict_cumsum: the host's own torch.sum supplies the
    # sums, i.e. on the host that wrote the fixtures this is the reference bit for bit; elsewhere within the 1-ulp class
    z_mid = (F32(0.5) * (g["z"][:, 1:] + g["z"][:, :-1])).astype(F32)
    zs = ops.sample_pdf(T(z_mid, dev), T(np.ascontiguousarray(g["w"][:, 1:-1]), dev), O.Args(number_fine_samples=128, strict_cumsum=1))
    assert np.mean(np.abs(N(zs) - g["z_samples"]) > 5e-6) <= 5e-3


@pytest.mark.parametrize("nc,nf", [(16, 8), (32, 64), (64, 64), (48, 200), (3, 1), (200, 700)])
def test_sampler_shapes(dev, nc, nf):
    rng = np.random.default_rng(nc * 1000 + nf)
    B = 301                                                              # ragged vs 4 rays per workgroup
    o, d = syn.camera_rays(20, 20, syn.sphere_pose(10, 20, 2.4))
    sel = rng.choice(o.shape[0], B, replace=False)
    _, oo, dd, zz = syn.coarse_samples(o[sel], d[sel], 1.6, 3.1, nc, rng.random(B))
    ww = (rng.random((B, nc)).astype(F32) ** 3)
    ww[0] = 0
    ww[1] = 1
    _check_sampler(dev, oo, dd, zz, ww, nf)


def test_sampler_unsorted_input_falls_back_to_exact_sort(dev):
    rng = np.random.default_rng(11)
    B, nc, nf = 64, 64, 128
    z = rng.uniform(1, 4, (B, nc)).astype(F32)                           # NOT sorted
    z[::2] = np.sort(z[::2], -1)
    w = rng.random((B, nc)).astype(F32)
    o = rng.normal(size=(B, 3)).astype(F32)
    d = rng.normal(size=(B, 3)).astype(F32)
    _check_sampler(dev, o, d, z, w, nf)


def test_sample_pdf_and_fine_sampling_signatures(dev):
    from smpl_nerf_amd import ops
    g = load_golden("g4_sampler.npz")
    args = O.Args(number_fine_samples=128)
    ops._U_CACHE[(128, str(dev))] = T(g["u"][0], dev)
    try:
        z = g["z"]
        z_mid = (F32(0.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
        zs = ops.sample_pdf(T(z_mid, dev), T(g["w"][:, 1:-1].copy(), dev), args)
        np.testing.assert_array_equal(N(zs), O.sample_pdf(z_mid, g["w"][:, 1:-1], 128, u=g["u"][0]))
        zf, pts = ops.fine_sampling(T(g["o"], dev), T(g["d"], dev), T(z, dev), T(g["w"], dev), args)
        ozf, opts = O.fine_sampling(g["o"], g["d"], z, g["w"], 128, u=g["u"][0])
        np.testing.assert_array_equal(N(zf), ozf)
        np.testing.assert_array_equal(N(pts), opts)
    finally:
        ops._U_CACHE.pop((128, str(dev)), None)


def test_sampler_frame_size_properties(dev):
    from smpl_nerf_amd import ops
    rng = np.random.default_rng(2)
    data = syn.frame_batch(128, 128, seed=3)
    B = 16384
    w = (rng.random((B, 64)).astype(F32) ** 6)
    r = ops.hierarchical_samples(T(data[1], dev), T(data[2], dev), T(data[3], dev), T(w, dev), 128, want_samples=True)
    zf, pts, zs = N(r["z_fine"]), N(r["pts"]), N(r["z_samples"])
    assert np.all(np.diff(zf, axis=-1) >= 0)                             # sortedness
    both = np.sort(np.concatenate([data[3], zs], -1), -1)
    np.testing.assert_array_equal(zf, both)                              # a permutation of cat(z, samples)
    np.testing.assert_array_equal(pts, data[1][:, None, :] + data[2][:, None, :] * zf[..., None])
    assert np.all(zs >= data[3][:, :1]) and np.all(zs <= data[3][:, -1:])


# ------------------------------------------------------------------------------------------ a2
def _net(dev, params, **kw):
    from smpl_nerf_amd.nets import RenderRayNet
    net = RenderRayNet(n_layers=kw.get("n_layers", 8), width=kw.get("width", 256), positions_dim=60,
                       directions_dim=24, additional_input_dim=kw.get("additional_input_dim", 0),
                       skips=list(kw.get("skips", (4,))), use_directional_input=kw.get("use_directional_input", 1))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return net.to(dev)


@pytest.mark.parametrize("tag,kw,seed", [("skip4", dict(skips=(4,)), 11), ("noskip", dict(skips=()), 12),
                                         ("d4w128", dict(n_layers=4, width=128, skips=(1,)), 13)])
def test_render_ray_net_encoded_and_fused(dev, tag, kw, seed):
    from smpl_nerf_amd.ops import PositionalEncoder
    g = load_golden("g2_mlp.npz")
    params = syn.make_render_ray_net_params(seed, sigma_scale=30.0, rgb_scale=10.0, **kw)
    net = _net(dev, params, **kw)
    ref = g[f"raw_{tag}"]
    tol = 4e-6 * max(1.0, float(np.max(np.abs(ref))))                   # fp32 round-off of a 10-layer chain
    with torch.no_grad():
        out = N(net(T(g["inputs"], dev)))                               # RenderRayNet.forward(x_enc)
        assert out.shape == ref.shape
        assert maxabs(out, ref) <= tol
        fused = N(net.forward_fused(T(g["pts"], dev), T(g["dirs"], dev), 1, PositionalEncoder(10, 0),
                                    PositionalEncoder(4, 0)))
        assert maxabs(fused, ref) <= 4 * tol                            # + sin/cos ulps through 2^9 frequencies
    fkw = {k: v for k, v in kw.items() if k != "width"}
    assert maxabs(out, O.render_ray_net_forward(params, g["inputs"], **fkw)) <= tol


def test_render_ray_net_scene_weights(dev):
    from smpl_nerf_amd.ops import PositionalEncoder
    g = load_golden("g2_mlp.npz")
    net = _net(dev, syn.make_scene_nets(101)[1])
    with torch.no_grad():
        ref = g["raw_scene101"]
        # fp32 round-off times the head scale; the MFMA accumulates each K=256 dot product as one
        # sequential fmaf chain (MKL's blocked sgemm sums in shorter chains), hence 1e-5 and not 4e-6
        tol = 1e-5 * float(np.max(np.abs(ref)))
        assert maxabs(N(net(T(g["inputs"], dev))), ref) <= tol
        fused = net.forward_fused(T(g["pts"], dev), T(g["dirs"], dev), 1, PositionalEncoder(10, 0), PositionalEncoder(4, 0))
        assert maxabs(N(fused), ref) <= 2 * tol


def test_render_ray_net_additional_input_and_no_direction(dev):
    from smpl_nerf_amd.ops import PositionalEncoder
    g = load_golden("g2_mlp.npz")
    pe, de = O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0)
    params = syn.make_render_ray_net_params(14, 30.0, 10.0, additional_input_dim=6, skips=(4,))
    net = _net(dev, params, additional_input_dim=6, skips=(4,))
    x = np.concatenate([pe.encode(g["pts"]), g["add6"], de.encode(g["dirs"])], -1)
    with torch.no_grad():
        assert maxabs(N(net(T(x, dev))), g["raw_add6"]) <= 1e-4
        fused = net.forward_fused(T(g["pts"], dev), T(g["dirs"], dev), 1, PositionalEncoder(10, 0),
                                  PositionalEncoder(4, 0), additional=T(g["add6"], dev))
        assert maxabs(N(fused), g["raw_add6"]) <= 4e-4
    params = syn.make_render_ray_net_params(15, 30.0, 10.0, skips=(4,), use_directional_input=0)
    net = _net(dev, params, skips=(4,), use_directional_input=0)
    with torch.no_grad():
        assert maxabs(N(net(T(g["inputs"], dev))), g["raw_nodir"]) <= 1e-4


def test_render_ray_net_ragged_and_ray_broadcast(dev):
    """n not a multiple of the 64-sample workgroup tile; per-ray directions broadcast over samples."""
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(8)
    params = syn.make_render_ray_net_params(31, 30.0, 10.0, skips=(4,))
    net = _net(dev, params)
    B, Ns = 37, 5
    pts = rng.uniform(-2, 2, (B, Ns, 3)).astype(F32)
    dray = rng.normal(size=(B, 3)).astype(F32)
    pe, de = O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0)
    dn = O._normalize(np.broadcast_to(dray[:, None, :], (B, Ns, 3)))
    x = np.concatenate([pe.encode(pts), de.encode(dn)], -1).reshape(B * Ns, -1)
    ref = O.render_ray_net_forward(params, x)
    with torch.no_grad():
        out = net.forward_fused(T(pts, dev), T(dray, dev), Ns, PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    assert maxabs(N(out), ref) <= 2e-5 * max(1.0, float(np.max(np.abs(ref))))
    # weights updated in place -> the packed stream must be rebuilt
    with torch.no_grad():
        net.rgb_out_layer.bias.add_(1.0)
        out2 = net.forward_fused(T(pts, dev), T(dray, dev), Ns, PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    np.testing.assert_allclose(N(out2)[:, :3], N(out)[:, :3] + 1.0, atol=1e-5)


# ------------------------------------------------------------------------------------------ a3
def _pipeline(dev, wb=0, run_fine=1):
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import NerfPipeline
    pc, pf = syn.make_scene_nets(101)
    args = O.Args(white_background=wb, run_fine=run_fine)
    return NerfPipeline(_net(dev, pc), _net(dev, pf), args, PositionalEncoder(10, 0), PositionalEncoder(4, 0))


def psnr(img, gt):
    return O.mse2psnr_img(np.mean((np.asarray(img, np.float64) - gt) ** 2))


@pytest.mark.parametrize("tag,near,far,wb", [("nf14", 1.0, 4.0, 0), ("nf1631wb", 1.6, 3.1, 1)])
def test_nerf_pipeline_full_frame_vs_reference(dev, tag, near, far, wb):
    """BASELINE config 2: 128x128 frame, 64 coarse + 128 fine samples, netdepth 8 - against the frame
    the reference itself rendered on CPU (tests/golden/g5)."""
    g = load_golden("g5_nerf_pipeline.npz")
    pipe = _pipeline(dev, wb=wb)
    data = syn.frame_batch(128, 128, phi=0.0, theta=0.0, seed=7, near=near, far=far)
    with torch.no_grad():
        rgb, rgb_fine, pts_fine, alpha_fine = pipe([T(a, dev) for a in data])
    rgb, rgb_fine, pts_fine, alpha_fine = N(rgb), N(rgb_fine), N(pts_fine), N(alpha_fine)
    assert rgb.shape == (16384, 3) and pts_fine.shape == (16384, 192, 3) and alpha_fine.shape == (16384, 192)
    assert maxabs(rgb, g[f"rgb_{tag}"]) <= 1e-4                          # coarse frame
    assert maxabs(rgb_fine, g[f"rgb_fine_{tag}"]) <= 1e-4                # fine frame: north_star tolerance
    gt = data[4]
    assert abs(psnr(rgb_fine, gt) - psnr(g[f"rgb_fine_{tag}"], gt)) <= 0.01      # dB
    assert psnr(rgb_fine, g[f"rgb_fine_{tag}"].astype(np.float64)) > 90.0
    sub = g[f"sub_{tag}"]
    assert np.mean(np.abs(pts_fine[sub] - g[f"pts_fine_sub_{tag}"]) > 1e-4) <= 0.02   # fp32 noise floor (H2)
    assert np.mean(np.abs(alpha_fine[sub] - g[f"alpha_fine_sub_{tag}"]) > 1e-3) <= 0.02


def test_nerf_pipeline_coarse_only_and_oracle(dev):
    g = load_golden("g5_nerf_pipeline.npz")
    data = syn.frame_batch(128, 128, phi=0.0, theta=0.0, seed=7, near=1.0, far=4.0)
    sub = g["sub_nf14"]
    pipe = _pipeline(dev, run_fine=0)
    with torch.no_grad():
        out = pipe([T(a[sub], dev) for a in data])
    assert out[0] is out[1]                                             # quirk Q10
    assert maxabs(N(out[0]), g["coarse_only_rgb"]) <= 1e-5
    assert maxabs(N(out[3]), g["coarse_only_alpha"]) <= 5e-5            # sigma head scale (~20) x fp32 round-off x dist
    assert tuple(out[2].shape) == (256, 64, 3)
    # full pipeline against the oracle on the same subset
    pipe = _pipeline(dev)
    pc, pf = syn.make_scene_nets(101)
    with torch.no_grad():
        got = pipe([T(a[sub], dev) for a in data])
    from smpl_nerf_amd.ops import uniform_u
    args = O.Args(u=N(uniform_u(128, dev)))
    ref = O.nerf_pipeline_forward(pc, pf, args, O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0),
                                  [a[sub] for a in data])
    assert maxabs(N(got[0]), ref[0]) <= 1e-5
    assert maxabs(N(got[1]), ref[1]) <= 1e-4


# ------------------------------------------------------------------------------------------ a7
def _smpl_pipeline(dev, wb=0, run_fine=1):
    from smpl_nerf_amd.nets import WarpFieldNet
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import SmplNerfPipeline
    pc, pf = syn.make_scene_nets(101)
    pw = syn.make_warp_field_params(103, out_scale=0.3)
    mw = WarpFieldNet(8, 256, 60, 40)
    mw.load_state_dict({k: torch.from_numpy(v) for k, v in pw.items()})
    args = O.Args(white_background=wb, run_fine=run_fine)
    pipe = SmplNerfPipeline(_net(dev, pc), _net(dev, pf), mw.to(dev), args, PositionalEncoder(10, 0),
                            PositionalEncoder(4, 0), PositionalEncoder(10, 0))
    return pipe, (pc, pf, pw)


def test_warp_field_net(dev):
    from smpl_nerf_amd.nets import WarpFieldNet
    g = load_golden("g2_mlp.npz")
    mw = WarpFieldNet(8, 256, 60, 40)
    mw.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_warp_field_params(21).items()})
    mw = mw.to(dev)
    with torch.no_grad():
        out = mw(T(g["warp_inputs"], dev))                               # WarpFieldNet.forward(x_rows)
    assert maxabs(N(out), g["warp_out"]) <= 2e-6


@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("wb", [0, 1])
def test_smpl_nerf_pipeline_vs_reference(dev, wb, prec):
    """a7 against the frames the reference rendered; bf16x6 = the warp net and both RenderRayNets on the bf16 matrix
    cores (split-bf16, fp32-class), held to the same tolerances."""
    g = load_golden("g6_smpl_nerf_pipeline.npz")
    pipe, _ = _smpl_pipeline(dev, wb=wb)
    pipe.model_coarse.precision = pipe.model_fine.precision = pipe.model_warp_field.precision = prec
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    sub = g["sub"]
    d = [T(a[sub], dev) for a in data[:4]] + [T(g["goal_pose"], dev), T(data[4][sub], dev)]
    with torch.no_grad():
        out = pipe(d)
    names = ("rgb", "rgb_fine", "warp_fine", "pts_fine", "warped_fine", "alpha_fine")
    assert [tuple(o.shape) for o in out] == [g[f"{n}_wb{wb}"].shape for n in names]
    assert maxabs(N(out[0]), g[f"rgb_wb{wb}"]) <= 1e-4
    assert maxabs(N(out[1]), g[f"rgb_fine_wb{wb}"]) <= 1e-4               # north_star tolerance
    for i in (2, 3, 4):
        assert np.mean(np.abs(N(out[i]) - g[f"{names[i]}_wb{wb}"]) > 1e-4) <= 0.02


def test_smpl_nerf_pipeline_coarse_only_and_oracle(dev):
    g = load_golden("g6_smpl_nerf_pipeline.npz")
    pipe, (pc, pf, pw) = _smpl_pipeline(dev, run_fine=0)
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    sub = g["sub"]
    d = [T(a[sub], dev) for a in data[:4]] + [T(g["goal_pose"], dev), T(data[4][sub], dev)]
    with torch.no_grad():
        out = pipe(d)
    assert out[0] is out[1]
    assert maxabs(N(out[0]), g["coarse_rgb"]) <= 1e-5
    assert maxabs(N(out[2]), g["coarse_warp"]) <= 2e-6
    assert maxabs(N(out[4]), g["coarse_warped"]) <= 2e-6
    assert maxabs(N(out[5]), g["coarse_alpha"]) <= 5e-5
    from smpl_nerf_amd.ops import uniform_u
    pipe, _ = _smpl_pipeline(dev)
    with torch.no_grad():
        got = pipe(d)
    enc = O.PositionalEncoder
    dn = [a[sub] for a in data[:4]] + [g["goal_pose"], data[4][sub]]
    ref = O.smpl_nerf_pipeline_forward(pc, pf, pw, O.Args(u=N(uniform_u(128, dev))), enc(10, 0), enc(4, 0), enc(10, 0), dn)
    assert maxabs(N(got[0]), ref[0]) <= 1e-5 and maxabs(N(got[1]), ref[1]) <= 1e-4


# ------------------------------------------------------------------------------------------ a8
def _av_pipeline(dev, wb=0, run_fine=0):
    from smpl_nerf_amd.nets import AppendVerticesNet
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import AppendVerticesPipeline
    from smpl_nerf_amd.synthetic_smpl import IndexPoseEstimator, LinearBodyModel
    nets = []
    for seed in (201, 202):
        m = AppendVerticesNet(8, 256, 60, 24, 6890, additional_input_layers=1, skips=[4])
        m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_append_vertices_params(seed).items()})
        nets.append(m.to(dev))
    est = IndexPoseEstimator(torch.from_numpy(syn.human_poses((41, 38), 0, 60, 10)), torch.zeros(1, 10)).to(dev)
    body = LinearBodyModel(seed=3).to(dev)
    return AppendVerticesPipeline(nets[0], nets[1], est, body, O.Args(white_background=wb, run_fine=run_fine),
                                  PositionalEncoder(10, 0), PositionalEncoder(4, 0)), nets


@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("wb", [0, 1])
def test_append_vertices_pipeline(dev, wb, prec):
    g = load_golden("g9_append_vertices.npz")
    data = syn.frame_batch(128, 128, phi=3.0, theta=-10.0, seed=11)
    d = [T(a[g["sub"]], dev) for a in data[:4]] + [torch.from_numpy(g["images"]).to(dev), T(data[4][g["sub"]], dev)]
    pipe, nets = _av_pipeline(dev, wb=wb)
    pipe.set_precision(prec)
    with torch.no_grad():
        out = pipe(d)
        assert out[0] is out[1] and tuple(out[2].shape) == (24, 64, 3)
        assert maxabs(N(out[0]), g[f"coarse_rgb_wb{wb}"]) <= 1e-5
        assert maxabs(N(out[3]), g[f"coarse_alpha_wb{wb}"]) <= 5e-5
        # the fine branch (which the reference cannot run, pipeline.py:71) against the oracle's restatement
        pipe, _ = _av_pipeline(dev, wb=wb, run_fine=1)
        got = pipe.set_precision(prec)(d)
    from smpl_nerf_amd.ops import uniform_u
    from smpl_nerf_amd.synthetic_smpl import LinearBodyModel
    verts = LinearBodyModel(seed=3)(body_pose=torch.from_numpy(syn.human_poses((41, 38), 0, 60, 10)[g["images"]])).vertices.numpy()
    ref = O.append_vertices_pipeline_forward(syn.make_append_vertices_params(201), syn.make_append_vertices_params(202), verts,
                                             O.Args(white_background=wb, u=N(uniform_u(128, dev))), O.PositionalEncoder(10, 0),
                                             O.PositionalEncoder(4, 0), [a[g["sub"]] for a in data])
    assert maxabs(N(got[0]), ref[0]) <= 1e-5 and maxabs(N(got[1]), ref[1]) <= 1e-4
    assert tuple(got[2].shape) == (24, 192, 3)


def test_append_vertices_net_forward_rows(dev):
    """AppendVerticesNet.forward(x) on literal 20754-column rows: only columns [:60] and [-24:] matter."""
    g = load_golden("g9_append_vertices.npz")
    _, nets = _av_pipeline(dev)
    rows = np.zeros((40, 20754), F32)
    rows[:, :60] = g["net_rows"][:, :60]
    rows[:, -24:] = g["net_rows"][:, 60:]
    rows[:, 60:-24] = 123.0                                              # must be ignored
    with torch.no_grad():
        out = nets[0](T(rows, dev))
    assert maxabs(N(out), g["net_out"]) <= 2e-5 * max(1.0, float(np.abs(g["net_out"]).max()))
    sd = nets[0].state_dict()
    assert "vertices_net.0.weight" in sd and tuple(sd["vertices_net.0.weight"].shape) == (256, 6890)


# ------------------------------------------------------------------------------------------ f-4
@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("name,npose", [("smpl", 69), ("two", 2)])
@pytest.mark.parametrize("enc", [0, 1])
def test_append_pose_pipelines(dev, name, npose, enc, prec):
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import AppendSmplParamsPipeline, AppendToNerfPipeline
    g = load_golden("g10_append_pose.npz")
    add = npose * (20 if enc else 1)
    nets = []
    for seed in (301 + enc, 303 + enc):
        m = RenderRayNet(8, 256, 60, 24, add, skips=[4])
        m.load_state_dict({k: torch.from_numpy(v) for k, v in
                           syn.make_scene_net_params(seed, add_first=True, additional_input_dim=add).items()})
        nets.append(m.to(dev))
    cls = AppendSmplParamsPipeline if name == "smpl" else AppendToNerfPipeline
    pipe = cls(nets[0], nets[1], O.Args(human_pose_encoding=enc), PositionalEncoder(10, 0), PositionalEncoder(4, 0),
               PositionalEncoder(10, 0)).set_precision(prec)
    data = syn.frame_batch(128, 128, phi=3.0, theta=-10.0, seed=11)
    d = [T(a[g["sub"]], dev) for a in data[:4]] + [T(g["goal_pose"], dev), T(data[4][g["sub"]], dev)]
    with torch.no_grad():
        out = pipe(d)
    assert maxabs(N(out[0]), g[f"{name}{enc}_rgb"]) <= 1e-5
    assert maxabs(N(out[1]), g[f"{name}{enc}_rgb_fine"]) <= 1e-4
    assert np.mean(np.abs(N(out[2]) - g[f"{name}{enc}_pts_fine"]) > 1e-4) <= 0.02
    assert tuple(out[3].shape) == g[f"{name}{enc}_alpha_fine"].shape


# ------------------------------------------------------------------------------------------ edge cases
def test_empty_and_maximum_sizes(dev):
    """Empty batches are no-ops; the sampler/compositor limits (Nc, Nf, N <= 1024) work at the limit."""
    from smpl_nerf_amd import ops
    z = torch.empty((0, 64), device=dev)
    r = ops.hierarchical_samples(torch.empty((0, 3), device=dev), torch.empty((0, 3), device=dev), z, z.clone(), 128)
    assert tuple(r["z_fine"].shape) == (0, 192) and tuple(r["pts"].shape) == (0, 192, 3)
    rgb, w, a = ops.composite(torch.empty((0, 64, 4), device=dev), z, torch.empty((0, 3), device=dev), False)
    assert tuple(rgb.shape) == (0, 3) and tuple(w.shape) == (0, 64)
    out = ops.searchsorted(torch.empty((0, 5), device=dev), torch.empty((0, 7), device=dev))
    assert tuple(out.shape) == (0, 7)
    assert tuple(ops.PositionalEncoder(10, 0).encode(torch.empty((0, 3), device=dev)).shape) == (0, 60)
    rng = np.random.default_rng(21)
    B, nc, nf = 9, 1024, 1024
    zz = np.sort(rng.uniform(1, 4, (B, nc)).astype(F32), -1)
    ww = rng.random((B, nc)).astype(F32) ** 2
    o = rng.normal(size=(B, 3)).astype(F32)
    d = rng.normal(size=(B, 3)).astype(F32)
    _check_sampler(dev, o, d, zz, ww, nf)
    raw = rng.normal(0, 2, (B, 1024, 4)).astype(F32)
    rgb, w, a = ops.composite(T(raw, dev), T(zz, dev), T(d, dev), True)
    orgb, ow, oa = O.raw2outputs(raw, zz, np.broadcast_to(d[:, None, :], (B, 1024, 3)), 1)
    assert maxabs(N(rgb), orgb) <= 2e-6 and maxabs(N(w), ow) <= 5e-7 and maxabs(N(a), oa) <= 5e-7
    with pytest.raises(RuntimeError, match="1024"):
        ops.hierarchical_samples(T(o, dev), T(d, dev), T(zz, dev), T(ww, dev), 1025)


def test_searchsorted_ties_nan_free_duplicates(dev):
    """Collisions: long runs of equal keys and queries equal to keys, both sides."""
    from smpl_nerf_amd.ops import searchsorted
    rng = np.random.default_rng(4)
    a = np.sort(rng.integers(0, 8, (50, 63)).astype(F32), -1)            # many duplicates
    v = rng.integers(-1, 9, (50, 128)).astype(F32)
    for side in ("left", "right"):
        np.testing.assert_array_equal(N(searchsorted(T(a, dev), T(v, dev), side=side)), O.searchsorted(a, v, side))


# ------------------------------------------------------------------------------------------ 8(f)-1 ray generation
def test_raygen_bit_exact_vs_reference_path(dev):
    """get_rays + CoarseSampling + ToTensor of the reference (golden g8) reproduced on the device."""
    from smpl_nerf_amd.raygen import RayGenerator
    g = load_golden("g8_rays.npz")
    gen = RayGenerator(g["pose"][None], 16, 24, np.pi / 3, 1.0, 4.0, 64, dev)
    out = gen.batch(T(g["rows"].astype(np.int64), dev), T(g["jitter"], dev))
    np.testing.assert_array_equal(N(out[0]), g["samples"])
    np.testing.assert_array_equal(N(out[1]), g["o"])
    np.testing.assert_array_equal(N(out[2]), g["d"])
    np.testing.assert_array_equal(N(out[3]), g["z"])


def test_raygen_frame_and_random_batches(dev):
    """A whole 128x128 frame (several frames resident, ragged batch) equals the host path used by every other
    test; random batches gather the matching ground-truth pixels."""
    from smpl_nerf_amd.raygen import RayGenerator
    poses = np.stack([syn.sphere_pose(7.0 * f, 25.0 * f, 2.4) for f in range(3)])
    imgs = np.stack([syn.procedural_image(128, 128, 7.0 * f, 25.0 * f) for f in range(3)])
    gen = RayGenerator(poses, 128, 128, np.pi / 3, 1.6, 3.1, 64, dev, images=imgs)
    rng = np.random.default_rng(0)
    jit = rng.random(16384)
    out = gen.batch(torch.arange(16384, 2 * 16384, device=dev), T(jit, dev))      # frame 1
    o, d = syn.camera_rays(128, 128, poses[1])
    pts, o32, d32, z = syn.coarse_samples(o, d, 1.6, 3.1, 64, jit)
    np.testing.assert_array_equal(N(out[0]), pts)
    np.testing.assert_array_equal(N(out[1]), o32)
    np.testing.assert_array_equal(N(out[2]), d32)
    np.testing.assert_array_equal(N(out[3]), z)
    np.testing.assert_array_equal(N(out[4]), imgs[1].reshape(-1, 3))
    g = torch.Generator(device=dev).manual_seed(1)
    b = gen.random_batch(1001, generator=g)
    assert [tuple(t.shape) for t in b] == [(1001, 64, 3), (1001, 3), (1001, 3), (1001, 64), (1001, 3)]
    assert bool(torch.all(b[3][:, 1:] > b[3][:, :-1]))


# ------------------------------------------------------------------------------------------ split-bf16 MFMA modes
@pytest.mark.parametrize("prec,tol_raw,tol_rgb", [("bf16x6", 1e-5, 1e-4), ("f16x3", 1e-5, 1e-4), ("bf16x3", 2e-3, 2e-4)])
def test_split_bf16_modes(dev, prec, tol_raw, tol_rgb):
    """The 16-bit-matrix-core modes of the inference kernel: bf16x6 (three bf16 parts, six products) and f16x3 (two fp16
    parts of power-of-two-scaled operands, three products) are in the fp32 kernel's parity class (RGB <= 1e-4 vs the
    reference's frame, raw <= 1e-5 * head scale), bf16x3 trades ~2^-16 relative error for speed."""
    from smpl_nerf_amd.ops import PositionalEncoder
    g = load_golden("g2_mlp.npz")
    net = _net(dev, syn.make_scene_nets(101)[1])
    net.precision = prec
    ref = g["raw_scene101"]
    with torch.no_grad():
        fused = net.forward_fused(T(g["pts"], dev), T(g["dirs"], dev), 1, PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    assert maxabs(N(fused), ref) <= 2 * tol_raw * float(np.max(np.abs(ref)))
    # ragged n, per-ray directions broadcast over samples
    rng = np.random.default_rng(8)
    B, Ns = 37, 5
    pts = rng.uniform(-2, 2, (B, Ns, 3)).astype(F32)
    dray = rng.normal(size=(B, 3)).astype(F32)
    with torch.no_grad():
        out = net.forward_fused(T(pts, dev), T(dray, dev), Ns, PositionalEncoder(10, 0), PositionalEncoder(4, 0))
        net.precision = "fp32"
        out32 = net.forward_fused(T(pts, dev), T(dray, dev), Ns, PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    assert maxabs(N(out), N(out32)) <= 2 * tol_raw * float(np.abs(N(out32)).max())
    # whole frame against the reference's rendering
    g5 = load_golden("g5_nerf_pipeline.npz")
    pipe = _pipeline(dev)
    pipe.model_coarse.precision = pipe.model_fine.precision = prec
    data = syn.frame_batch(128, 128, phi=0.0, theta=0.0, seed=7, near=1.0, far=4.0)
    with torch.no_grad():
        rgb, rgb_fine, _, _ = pipe([T(a, dev) for a in data])
    assert maxabs(N(rgb), g5["rgb_nf14"]) <= tol_rgb
    assert maxabs(N(rgb_fine), g5["rgb_fine_nf14"]) <= tol_rgb
    assert abs(psnr(N(rgb_fine), data[4]) - psnr(g5["rgb_fine_nf14"], data[4])) <= 0.01


def test_split_bf16_additional_inputs_and_per_sample_dirs(dev):
    """bf16x6 with pose columns in front of the encoding (append_smpl_params) and with per-sample directions."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(3)
    m = RenderRayNet(8, 256, 60, 24, 69, skips=[4])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in
                       syn.make_scene_net_params(301, add_first=True, additional_input_dim=69).items()})
    m = m.to(dev)
    B, Ns = 33, 7
    pts = rng.uniform(-2, 2, (B, Ns, 3)).astype(F32)
    dsm = rng.normal(size=(B * Ns, 3)).astype(F32)
    pose = rng.uniform(-1, 1, (B, 69)).astype(F32)
    enc = (PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    with torch.no_grad():
        a = m.forward_fused(T(pts, dev), T(dsm, dev), Ns, *enc, additional=T(pose, dev), add_first=True)
        m.precision = "bf16x6"
        b = m.forward_fused(T(pts, dev), T(dsm, dev), Ns, *enc, additional=T(pose, dev), add_first=True)
    assert maxabs(N(a), N(b)) <= 2e-5 * float(np.abs(N(a)).max())


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "bf16x3", "f16x3"])
@pytest.mark.parametrize("run_fine,white", [(1, 0), (1, 1), (0, 0)])
def test_render_rays_single_call(dev, prec, run_fine, white):
    """snerf_render_rays_f32 (the whole NerfPipeline.forward behind one C-ABI call) returns exactly what the five
    separate entry points return through NerfPipeline.forward - same kernels, same order, same stream."""
    pipe = _pipeline(dev)
    pipe.args.run_fine, pipe.args.white_background = run_fine, white
    pipe.model_coarse.precision = pipe.model_fine.precision = prec
    data = syn.frame_batch(128, 128, phi=20.0, theta=10.0, seed=3, near=1.0, far=4.0)
    sub = np.arange(0, 16384, 37)  # ragged: 443 rays
    batch = [T(a[sub], dev) for a in data]
    with torch.no_grad():
        ref = pipe(batch)
        out = pipe.render_rays(batch)
    for a, b in zip(ref, out):
        assert a.shape == b.shape
        assert torch.equal(a, b)


def _mlp_ref64(params, pts, dirs, add, add_first, n_layers=8, skips=(4,), use_directional_input=1, **_):
    """RenderRayNet on [PE(x) | add | PE(normalised d)] rows in float64 (the layer order of models/render_ray_net.py:42-61)."""
    P = {k: np.asarray(v, np.float64) for k, v in params.items()}

    def pe(x, L):
        x = np.asarray(x, np.float64)
        return np.concatenate([f(x * 2.0 ** k) for k in range(L) for f in (np.sin, np.cos)], -1)

    d = np.asarray(dirs, np.float64)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    cols = [pe(pts, 10)]
    if add is not None:
        cols = [add.astype(np.float64)] + cols if add_first else cols + [add.astype(np.float64)]
    xin = np.concatenate(cols, -1)
    lin = lambda x, n: x @ P[n + ".weight"].T + P[n + ".bias"]
    o = np.maximum(lin(xin, "positions_pose_input"), 0)
    for i in range(n_layers - 1):
        o = np.maximum(lin(np.concatenate([o, xin], -1) if i in skips else o, f"positional_net.{i}"), 0)
    o = lin(o, "additional_linear_layer")
    sigma = lin(o, "sigma_out_layer")
    o = lin(np.concatenate([o, pe(d, 4)], -1) if use_directional_input else o, "directional_input")
    o = np.maximum(lin(o, "directional_net.0"), 0)
    return np.concatenate([lin(o, "rgb_out_layer"), sigma], -1)


@pytest.mark.gpu
def test_split_bf16_stress_against_fp32_kernel(dev):
    """The split kernels lay their instruction stream out by hand (asm loads with counted waits, DMA ring with
    alternating issuers): sweep depths, skip positions, additional inputs, direction modes and ragged sizes; repeated
    launches must be bit-identical (a timing-dependent hazard would show up as run-to-run noise).

    Accuracy is held against a float64 evaluation, RELATIVE TO THE EXACT-fp32 KERNEL'S OWN ERROR (the nets here are not
    all well conditioned).  bf16x6 (three bf16 parts = 24 significand bits, the three dropped cross terms are below
    2^-24) must be in the fp32 kernel's class in the strict sense: its RMS error over the whole sweep <= 1.0 x the fp32
    kernel's (measured 0.89: the MFMA sums a 32-long block before rounding, the fp32 MFMA a 4-long one), and no case with
    >= 1000 outputs worse than 1.25 x in RMS / 1.6 x in its maximum (the maximum of ~1e5 errors fluctuates by that much
    between two equally accurate evaluations).  f16x3 (two fp16 parts of scaled operands, 22 bits) is held to the same
    figures - measured 0.81 / 1.11 / 1.28 - and bf16x3 (16 bits) to 2^-16-class error."""
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(2024)
    enc = (PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    cases = [dict(n_layers=8, skips=(4,)), dict(n_layers=8, skips=()), dict(n_layers=5, skips=(2,)), dict(n_layers=3, skips=(1,)),
             dict(n_layers=8, skips=(4,), additional_input_dim=69), dict(n_layers=8, skips=(3, 6), additional_input_dim=5),
             dict(n_layers=6, skips=(4,), use_directional_input=0)]
    modes = ("fp32", "bf16x6", "bf16x3", "f16x3")
    sq, cnt = {m: 0.0 for m in modes}, 0
    for ci, kw in enumerate(cases):
        add_dim = kw.get("additional_input_dim", 0)
        add_first = bool(add_dim and ci % 2)
        if kw.get("use_directional_input", 1):
            params = syn.make_scene_net_params(500 + ci, add_first=add_first, **kw)
        else:   # the scene calibration probes the directional branch: plain random init for the ablation switch
            params = syn.make_render_ray_net_params(500 + ci, 30.0, 10.0, **kw)
        net = _net(dev, params, **kw)
        for B, Ns, per_sample in ((1, 1, 0), (37, 5, 0), (129, 64, 1), (700, 192, 0)):
            pts = rng.uniform(-2.5, 2.5, (B, Ns, 3)).astype(F32)
            dirs = rng.normal(size=(B, Ns, 3) if per_sample else (B, 1, 3)).astype(F32)
            add = rng.uniform(-1, 1, (B, 1, add_dim)).astype(F32) if add_dim else None
            kwf = dict(additional=T(add[:, 0], dev), add_first=add_first) if add_dim else {}
            tdirs = T(dirs.reshape(-1, 3), dev)
            outs = {}
            with torch.no_grad():
                for prec in modes:
                    net.precision = prec
                    a = net.forward_fused(T(pts, dev), tdirs, Ns, *enc, **kwf)
                    b = net.forward_fused(T(pts, dev), tdirs, Ns, *enc, **kwf)
                    assert torch.equal(a, b), (kw, B, Ns, prec)
                    outs[prec] = N(a).reshape(B, Ns, 4).astype(np.float64)
                    assert np.isfinite(outs[prec]).all()
            ref = _mlp_ref64(params, pts, np.broadcast_to(dirs, (B, Ns, 3)),
                             None if add is None else np.broadcast_to(add, (B, Ns, add_dim)), add_first, **kw)
            scale = float(np.abs(ref).max()) + 1e-6
            err = {m: (outs[m] - ref) / scale for m in modes}
            emax = {m: float(np.abs(err[m]).max()) for m in modes}
            erms = {m: float(np.sqrt((err[m] ** 2).mean())) for m in modes}
            for m in modes:
                sq[m] += float((err[m] ** 2).sum())
            cnt += ref.size
            tag = (kw, B, Ns, emax, erms)
            if ref.size >= 1000:
                for m in ("bf16x6", "f16x3"):
                    assert erms[m] <= 1.25 * erms["fp32"], (m,) + tag
                    assert emax[m] <= 1.6 * emax["fp32"], (m,) + tag
            else:       # a handful of outputs: no statistics (the fp32 kernel can be right to the last bit by luck), only
                for m in ("bf16x6", "f16x3"):       # "same class": a few units of the 1e-6 x scale round-off level
                    assert emax[m] <= 4.0 * emax["fp32"] + 3e-6, (m,) + tag
            assert emax["bf16x3"] <= 300.0 * emax["fp32"] + 2e-3, tag
    agg = {m: float(np.sqrt(sq[m] / cnt)) for m in modes}
    print("aggregate RMS error vs float64, relative to the fp32 kernel:", {m: round(agg[m] / agg["fp32"], 4) for m in modes})
    assert agg["bf16x6"] <= 1.0 * agg["fp32"], agg
    assert agg["f16x3"] <= 1.0 * agg["fp32"], agg


@pytest.mark.gpu
@pytest.mark.parametrize("wscale,xscale", [(1.0, 1.0), (1e-3, 1.0), (300.0, 1.0), (1.0, 1e-4), (3e-3, 40.0), (1e3, 1e3)])
def test_f16x3_operand_ranges(dev, wscale, xscale):
    """f16x3 scales every operand into fp16's range by exact powers of two (weights per layer, activations per sample):
    hidden weights scaled by `wscale` (activations then grow or shrink by wscale^depth: 1e-21 .. 1e21 relative to the
    inputs) and sample positions by `xscale` must neither overflow nor lose the result - error against float64 bounded
    like the fp32 kernel's own."""
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(int(wscale * 1000) % 97 + 3)
    params = syn.make_render_ray_net_params(77, 30.0, 10.0)
    for k in list(params):
        if k.startswith("positional_net.") and k.endswith(".weight"):
            params[k] = (params[k] * np.float32(wscale)).astype(F32)
    net = _net(dev, params)
    B, Ns = 64, 16
    pts = (rng.uniform(-2.5, 2.5, (B, Ns, 3)) * xscale).astype(F32)
    dirs = rng.normal(size=(B, 1, 3)).astype(F32)
    enc = (PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    outs = {}
    with torch.no_grad():
        for prec in ("fp32", "f16x3"):
            net.precision = prec
            outs[prec] = N(net.forward_fused(T(pts, dev), T(dirs.reshape(-1, 3), dev), Ns, *enc)).reshape(B, Ns, 4).astype(np.float64)
    ref = _mlp_ref64(params, pts, np.broadcast_to(dirs, (B, Ns, 3)), None, False)
    scale = float(np.abs(ref).max()) + 1e-30
    assert np.isfinite(ref).all() and np.isfinite(outs["f16x3"]).all()
    e32, ef = (float(np.abs(outs[k] - ref).max()) for k in ("fp32", "f16x3"))
    assert ef <= 8.0 * e32 + 1e-5 * scale, (wscale, xscale, e32, ef, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("xscale", [1.0, 300.0])
def test_f16x3_identity_encoder_columns(dev, xscale):
    """Encoders with include_identity=1 put the raw coordinates beside the sin / cos columns (utils.py:114-131): f16x3
    must take the operand scale of those layers from the sample's largest coordinate (|x| up to 750 here; 2^14 |x| would
    overflow fp16 from |x| = 4 on)."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(17)
    params = syn.make_render_ray_net_params(79, 30.0, 10.0, positions_dim=63, directions_dim=27)
    net = RenderRayNet(8, 256, 63, 27, skips=[4])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net = net.to(dev)
    B, Ns = 48, 9
    pts = (rng.uniform(-2.5, 2.5, (B, Ns, 3)) * xscale).astype(F32)
    dirs = rng.normal(size=(B, 1, 3)).astype(F32)
    enc = (PositionalEncoder(10, 1), PositionalEncoder(4, 1))
    outs = {}
    with torch.no_grad():
        for prec in ("fp32", "f16x3"):
            net.precision = prec
            outs[prec] = N(net.forward_fused(T(pts, dev), T(dirs.reshape(-1, 3), dev), Ns, *enc)).astype(np.float64)
    P = {k: np.asarray(v, np.float64) for k, v in params.items()}
    d = np.broadcast_to(dirs, (B, Ns, 3)).reshape(-1, 3).astype(np.float64)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    pe = lambda x, L: np.concatenate([x] + [f(x * 2.0 ** k) for k in range(L) for f in (np.sin, np.cos)], -1)
    xin, dpe = pe(pts.reshape(-1, 3).astype(np.float64), 10), pe(d, 4)
    lin = lambda x, n: x @ P[n + ".weight"].T + P[n + ".bias"]
    o = np.maximum(lin(xin, "positions_pose_input"), 0)
    for i in range(7):
        o = np.maximum(lin(np.concatenate([o, xin], -1) if i == 4 else o, f"positional_net.{i}"), 0)
    o = lin(o, "additional_linear_layer")
    sigma = lin(o, "sigma_out_layer")
    o = np.maximum(lin(lin(np.concatenate([o, dpe], -1), "directional_input"), "directional_net.0"), 0)
    ref = np.concatenate([lin(o, "rgb_out_layer"), sigma], -1)
    scale = float(np.abs(ref).max())
    e32, ef = (float(np.abs(outs[k] - ref).max()) for k in ("fp32", "f16x3"))
    assert np.isfinite(outs["f16x3"]).all()
    assert e32 <= 1e-4 * scale                      # (the layout of the identity columns is what the fp32 kernel uses)
    assert ef <= 8.0 * e32 + 1e-5 * scale, (xscale, e32, ef, scale)


@pytest.mark.gpu
def test_torchsearchsorted_shim_on_gpu(dev):
    """`from torchsearchsorted import searchsorted` as written in the reference (utils.py:14, :212) resolves to the HIP
    kernel through shims/ and reproduces numpy.searchsorted (the reference's test oracle)."""
    import importlib
    import sys
    shims = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shims")
    sys.path.insert(0, shims)
    try:
        sys.modules.pop("torchsearchsorted", None)
        searchsorted = importlib.import_module("torchsearchsorted").searchsorted
        rng = np.random.default_rng(5)
        a = np.sort(rng.random((300, 63)).astype(F32), 1)
        v = rng.random((300, 128)).astype(F32)
        for side in ("left", "right"):
            got = N(searchsorted(T(a, dev), T(v, dev), side=side))
            want = np.stack([np.searchsorted(a[i], v[i], side=side) for i in range(a.shape[0])])
            assert got.dtype == np.int64 and np.array_equal(got, want)
    finally:
        sys.path.remove(shims)
        sys.modules.pop("torchsearchsorted", None)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [0, 3, 16])
def test_c_host_example(dev, tmp_path, precision):
    """examples/c_host/render_rays.c - plain C99, the C-ABI of include/smplnerf.h plus the HIP runtime, no Python in
    the process - renders the same rays as NerfPipeline.forward, bit for bit."""
    import shutil
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("gcc") or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc / ROCm headers not available")
    exe = str(tmp_path / "render_rays")
    lib = os.path.join(root, "smpl_nerf_amd", "csrc", "libsmplnerf_hip.so")
    subprocess.run(["gcc", "-std=c99", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_host", "render_rays.c"), lib, "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + os.path.dirname(lib), "-o", exe], check=True)
    pipe = _pipeline(dev)
    prec = {0: "fp32", 3: "bf16x6", 16: "f16x3"}[precision]
    pipe.model_coarse.precision = pipe.model_fine.precision = prec
    data = syn.frame_batch(128, 128, phi=10.0, theta=5.0, seed=11, near=1.0, far=4.0)
    sub = np.arange(0, 16384, 29)
    batch = [np.ascontiguousarray(a[sub]) for a in data]
    B, Nc, Nf = batch[3].shape[0], batch[3].shape[1], 128
    from smpl_nerf_amd.ops import uniform_u
    blob = struct.pack("<5i", 0x534e5246, B, Nc, Nf, 0)
    for m in (pipe.model_coarse, pipe.model_fine):
        blob += bytes(m.desc_for_encoders(pipe.position_encoder, pipe.direction_encoder, False))
    for m in (pipe.model_coarse, pipe.model_fine):
        blob += torch.cat([p.detach().reshape(-1).float() for p in m._ordered_params()]).cpu().numpy().tobytes()
    for a in batch[:4]:
        blob += a.astype(F32).tobytes()
    blob += uniform_u(Nf, dev).cpu().numpy().astype(F32).tobytes()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(blob)
    r = subprocess.run([exe, str(fin), str(fout), str(precision)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.frombuffer(fout.read_bytes(), dtype=F32)
    with torch.no_grad():
        ref = pipe([T(a, dev) for a in batch])
    want = np.concatenate([N(t).reshape(-1) for t in ref])
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16x6", "f16x3"])
def test_full_256_frame_properties(dev, prec):
    """BASELINE configs[3]/[4] frame size (256 x 256 = 65 536 rays, 64 + 128 samples = 16.8 M MLP evaluations) through
    size-independent properties: rays are independent, so (i) rendering the frame in one call, in 7 ragged chunks and in a
    permuted ray order gives bit-identical per-ray results, (ii) merged samples are sorted and contain the coarse ones,
    (iii) compositing weights are a sub-probability (alpha in [0,1], rgb in [0,1] without white background), and
    (iv) a 1/16 subset matches the CPU oracle to the north-star tolerance."""
    from smpl_nerf_amd.ops import uniform_u
    pipe = _pipeline(dev)
    pipe.model_coarse.precision = pipe.model_fine.precision = prec
    data = syn.frame_batch(256, 256, phi=30.0, theta=20.0, seed=21, near=1.0, far=4.0)
    R = data[0].shape[0]
    assert R == 65536
    full = [T(a, dev) for a in data]
    with torch.no_grad():
        ref = [N(t) for t in pipe(full)]
        cuts = [0, 1, 130, 9000, 9001, 30000, 50011, R]
        parts = [pipe([t[a:b] for t in full]) for a, b in zip(cuts[:-1], cuts[1:])]
        perm = torch.from_numpy(np.random.default_rng(5).permutation(R)).to(dev)
        shuf = pipe([t[perm] for t in full])
    for k in range(4):
        assert np.array_equal(np.concatenate([N(p[k]) for p in parts]), ref[k])
        assert np.array_equal(N(shuf[k]), ref[k][N(perm)])
    rgb, rgb_fine, pts_fine, dens = ref
    o, d = data[1], data[2]
    zf = ((pts_fine - o[:, None, :]) * d[:, None, :]).sum(-1) / (d * d).sum(-1)[:, None]      # depth along the ray
    assert np.all(np.diff(zf, axis=1) >= -1e-4)
    assert rgb.min() >= 0.0 and rgb.max() <= 1.0 + 1e-6 and rgb_fine.min() >= 0.0 and rgb_fine.max() <= 1.0 + 1e-6
    assert dens.min() >= 0.0 and dens.max() <= 1.0
    sub = np.arange(0, R, 16)
    pc, pf = syn.make_scene_nets(101)
    want = O.nerf_pipeline_forward(pc, pf, O.Args(u=N(uniform_u(128, dev))), O.PositionalEncoder(10, 0),
                                   O.PositionalEncoder(4, 0), [a[sub] for a in data])
    assert maxabs(rgb[sub], want[0]) <= 1e-4 and maxabs(rgb_fine[sub], want[1]) <= 1e-4


@pytest.mark.gpu
def test_warp_field_net_split_bf16(dev):
    """The fused warp stage (warp, x' = x + warp, sdir = x' - o) on the bf16 matrix cores against the fp32 kernel: six
    products per MAC, so the warp agrees to fp32 round-off (it feeds the 2^9 band of the position encoding), for ragged
    sizes and both precision spellings."""
    from smpl_nerf_amd.nets import WarpFieldNet
    from smpl_nerf_amd.ops import PositionalEncoder
    rng = np.random.default_rng(17)
    mw = WarpFieldNet(8, 256, 60, 40)
    mw.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_warp_field_params(103, out_scale=0.3).items()})
    mw = mw.to(dev)
    pos_enc, pose_enc = PositionalEncoder(10, 0), PositionalEncoder(10, 0)
    for B, Ns in ((1, 1), (37, 5), (300, 64), (513, 192)):
        pts = T(rng.uniform(-2.5, 2.5, (B, Ns, 3)).astype(F32), dev)
        o = T(rng.uniform(-3, 3, (B, 3)).astype(F32), dev)
        pose = pose_enc.encode(T(rng.uniform(0, 1, (B, 2)).astype(F32), dev))
        outs = {}
        with torch.no_grad():
            for prec in ("fp32", "bf16x6", "bf16x3"):
                mw.precision = prec
                a = mw.forward_fused(pts, pose, o, Ns, pos_enc)
                b = mw.forward_fused(pts, pose, o, Ns, pos_enc)
                assert all(torch.equal(x, y) for x, y in zip(a, b))
                outs[prec] = [N(t) for t in a]
        assert all(np.array_equal(x, y) for x, y in zip(outs["bf16x6"], outs["bf16x3"]))      # always three parts
        scale = float(np.abs(outs["fp32"][0]).max()) + 1e-6
        assert maxabs(outs["bf16x6"][0], outs["fp32"][0]) <= 4e-6 * scale + 1e-7
        assert maxabs(outs["bf16x6"][1], outs["fp32"][1]) <= 4e-6 * scale + 5e-7
        assert maxabs(outs["bf16x6"][2], outs["fp32"][2]) <= 4e-6 * scale + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16x6"])
@pytest.mark.parametrize("wb", [0, 1])
def test_render_rays_smpl_single_call(dev, prec, wb):
    """snerf_render_rays_smpl_f32 (SmplNerfPipeline.forward behind one C-ABI call) returns exactly what the separate entry
    points return through SmplNerfPipeline.forward."""
    g = load_golden("g6_smpl_nerf_pipeline.npz")
    pipe, _ = _smpl_pipeline(dev, wb=wb)
    pipe.set_precision(prec)
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    sub = g["sub"]
    d = [T(a[sub], dev) for a in data[:4]] + [T(g["goal_pose"], dev), T(data[4][sub], dev)]
    with torch.no_grad():
        ref = pipe(d)
        out = pipe.render_rays(d)
    assert len(ref) == len(out) == 6
    for a, b in zip(ref, out):
        assert a.shape == b.shape and torch.equal(a.reshape(b.shape), b)
