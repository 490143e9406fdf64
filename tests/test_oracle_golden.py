"""Pin the numpy oracle (oracle/nerf_oracle.py) against vectors captured from the reference
itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import nerf_oracle as O
from smpl_nerf_amd import synthetic as syn
from conftest import load_golden

F32 = np.float32


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


# ---------------------------------------------------------------- a1
@pytest.mark.parametrize("L,ident", [(10, 0), (4, 0), (10, 1), (4, 1), (0, 1)])
def test_posenc(L, ident):
    g = load_golden("g1_posenc.npz")
    enc = O.PositionalEncoder(L, ident)
    out = enc.encode(g["x"])
    ref = g[f"enc_L{L}_id{ident}"]
    assert out.shape == ref.shape and out.shape[-1] == 3 * enc.output_dim
    # sin/cos of fp32 arguments up to ~2000 rad: libm vs torch's vectorised kernels, <= 2 ulp of 1
    assert maxabs(out, ref) <= 3e-7


def test_posenc_pose():
    g = load_golden("g1_posenc.npz")
    assert maxabs(O.PositionalEncoder(10, 0).encode(g["pose2"]), g["pose2_enc_L10_id0"]) <= 3e-7


# ---------------------------------------------------------------- a2
@pytest.mark.parametrize("tag,kw,seed", [("skip4", dict(skips=(4,)), 11), ("noskip", dict(skips=()), 12),
                                         ("d4w128", dict(n_layers=4, width=128, skips=(1,)), 13)])
def test_render_ray_net(tag, kw, seed):
    g = load_golden("g2_mlp.npz")
    params = syn.make_render_ray_net_params(seed, sigma_scale=30.0, rgb_scale=10.0, **kw)
    fkw = {k: v for k, v in kw.items() if k != "width"}
    out = O.render_ray_net_forward(params, g["inputs"], **fkw)
    ref = g[f"raw_{tag}"]
    scale = float(np.max(np.abs(ref)))
    assert maxabs(out, ref) <= 2e-6 * max(scale, 1.0)


def test_render_ray_net_scene_weights():
    """The calibrated scene net used by the pipeline goldens / bench (large head scales)."""
    g = load_golden("g2_mlp.npz")
    out = O.render_ray_net_forward(syn.make_scene_nets(101)[1], g["inputs"], skips=(4,))
    ref = g["raw_scene101"]
    assert maxabs(out, ref) <= 4e-6 * float(np.max(np.abs(ref)))      # fp32 round-off times the head scale


def test_render_ray_net_additional_input_and_nodir():
    g = load_golden("g2_mlp.npz")
    pe, de = O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0)
    params = syn.make_render_ray_net_params(14, 30.0, 10.0, additional_input_dim=6, skips=(4,))
    x = np.concatenate([pe.encode(g["pts"]), g["add6"], de.encode(g["dirs"])], -1)
    out = O.render_ray_net_forward(params, x, additional_input_dim=6, skips=(4,))
    assert maxabs(out, g["raw_add6"]) <= 2e-5
    params = syn.make_render_ray_net_params(15, 30.0, 10.0, skips=(4,), use_directional_input=0)
    out = O.render_ray_net_forward(params, g["inputs"], skips=(4,), use_directional_input=0)
    assert maxabs(out, g["raw_nodir"]) <= 2e-5


def test_param_count_matches_survey():
    shapes = O.render_ray_net_param_shapes(skips=(4,))
    assert sum(int(np.prod(s)) for _, s in shapes) == 610436
    shapes = O.render_ray_net_param_shapes(skips=())
    assert sum(int(np.prod(s)) for _, s in shapes) == 595076


def test_warp_field_net():
    g = load_golden("g2_mlp.npz")
    out = O.warp_field_net_forward(syn.make_warp_field_params(21), g["warp_inputs"])
    assert maxabs(out, g["warp_out"]) <= 2e-6


# ---------------------------------------------------------------- a4
@pytest.mark.parametrize("N", [1, 2, 64, 192, 100])
@pytest.mark.parametrize("wb", [0, 1])
@pytest.mark.parametrize("mode", ["ray", "smp"])
def test_raw2outputs(N, wb, mode):
    if N == 1 and mode == "smp":
        pytest.skip("reference special case N==1 ignores directions")
    g = load_golden("g3_raw2outputs.npz")
    B = g[f"raw_N{N}"].shape[0]
    d = np.broadcast_to(g[f"dray_N{N}"][:, None, :], (B, N, 3)) if mode == "ray" else g[f"dsmp_N{N}"]
    rgb, w, a = O.raw2outputs(g[f"raw_N{N}"], g[f"z_N{N}"], d, wb)
    assert maxabs(rgb, g[f"rgb_N{N}_wb{wb}_{mode}"]) <= 1e-6
    assert maxabs(w, g[f"weights_N{N}_wb{wb}_{mode}"]) <= 5e-7
    assert maxabs(a, g[f"alpha_N{N}_wb{wb}_{mode}"]) <= 5e-7


def test_raw2outputs_noise_injection():
    g = load_golden("g3_raw2outputs.npz")
    d = np.broadcast_to(g["dray_N64"][:, None, :], (24, 64, 3))
    rgb, w, a = O.raw2outputs(g["raw_N64"], g["z_N64"], d, 0, noise=g["noise_N64"])
    assert maxabs(rgb, g["rgb_N64_noise"]) <= 1e-6
    assert maxabs(w, g["weights_N64_noise"]) <= 5e-7


# ---------------------------------------------------------------- a6
@pytest.mark.parametrize("side", ["left", "right"])
def test_searchsorted_known_answers(side):
    g = load_golden("g_searchsorted.npz")
    np.testing.assert_array_equal(O.searchsorted(g["a"], g["v"], side), g[f"out_{side}"])
    np.testing.assert_array_equal(O.searchsorted(g["a2"], g["v2"], side), g[f"out2_{side}"])
    np.testing.assert_array_equal(O._searchsorted_rows(g["a2"], g["v2"], side), g[f"out2_{side}"])


# ---------------------------------------------------------------- a5
def test_linspace_within_one_ulp_of_torch():
    g = load_golden("g4_sampler.npz")
    u = O.linspace01(128)
    assert u[0] == 0.0 and u[-1] == 1.0 and np.all(np.diff(u) > 0)
    assert maxabs(u, g["u"][0]) <= 6e-8          # torch's SIMD/FMA evaluation: <= 1 ulp apart


def _bin_width_at(z, zs):
    """Width of the coarse-midpoint bin each sample falls into (upper bound on a bin flip)."""
    z_mid = 0.5 * (z[:, 1:] + z[:, :-1])
    idx = np.clip(np.sum(z_mid[:, None, :] <= zs[:, :, None], -1), 1, z_mid.shape[1] - 1)
    return np.take_along_axis(z_mid, idx, -1) - np.take_along_axis(z_mid, idx - 1, -1)


def test_invert_cdf_exact_given_reference_cdf():
    """Kernel-boundary contract: same (cdf, u) in => bit-identical indices out, samples <= 1 ulp."""
    g = load_golden("g4_sampler.npz")
    z = g["z"]
    z_mid = (F32(0.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
    inds, samples = O.invert_cdf(z_mid, g["cdf"], g["u"])
    np.testing.assert_array_equal(inds, g["inds"])
    assert maxabs(samples, g["z_samples"]) <= 5e-7


def test_sample_pdf_main_case():
    g = load_golden("g4_sampler.npz")
    z, w = g["z"], g["w"]
    z_mid = (F32(0.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
    det = O.sample_pdf_detail(z_mid, w[:, 1:-1], 128, u=g["u"][0])
    # cdf: only the normalising sum differs from torch (<= 1 ulp of the sum => <= 2 ulp of 1 here)
    assert maxabs(det["cdf"], g["cdf"]) <= 2.4e-7
    # With its own cdf an index may flip only where u sits within that distance of a cdf knot, and
    # the inverse cdf is ill-conditioned in near-empty bins (1e-5 floor, utils.py:200/:224): the
    # reference's own result moves by a fraction of a bin under a 1-ulp change of the sum.  So:
    # nearly all samples agree to fp32 round-off, every sample agrees to within its bin.
    err = np.abs(det["samples"].astype(np.float64) - g["z_samples"])
    assert np.mean(det["inds"] != g["inds"]) <= 5e-3
    assert np.mean(err > 5e-6) <= 5e-3
    assert np.all(err <= _bin_width_at(z, g["z_samples"]) + 1e-6)
    zf, pts = O.fine_sampling(g["o"], g["d"], z, w, 128, u=g["u"][0])
    errz = np.abs(zf.astype(np.float64) - g["z_fine"])
    assert np.mean(errz > 5e-6) <= 5e-3
    assert np.mean(np.abs(pts - g["pts_fine"]) > 2e-5) <= 5e-3
    assert np.all(np.diff(zf, axis=-1) >= 0)


def test_strict_sampler_is_the_reference_bit_for_bit_from_the_weights():
    """With the recorded normalising sums (the one host-dependent intermediate, utils.py:201) the oracle's cdf, indices,
    samples, merged depths and points equal the reference's exactly - all 48 rays x 128 samples of g4, the adversarial
    rows included, and the four other (Nc, Nf) shapes."""
    g = load_golden("g4_sampler.npz")
    z_mid = (F32(0.5) * (g["z"][:, 1:] + g["z"][:, :-1])).astype(F32)
    det = O.sample_pdf_detail(z_mid, g["w"][:, 1:-1], 128, u=g["u"][0], tot=g["tot"])
    np.testing.assert_array_equal(det["cdf"], g["cdf"])
    np.testing.assert_array_equal(det["inds"], g["inds"])
    np.testing.assert_array_equal(det["samples"], g["z_samples"])
    zf, pts = O.fine_sampling(g["o"], g["d"], g["z"], g["w"], 128, u=g["u"][0], tot=g["tot"])
    np.testing.assert_array_equal(zf, g["z_fine"])
    np.testing.assert_array_equal(pts, g["pts_fine"])
    for nc, nf in ((16, 8), (32, 64), (64, 64), (48, 200)):
        k = f"{nc}_{nf}"
        zf, pts = O.fine_sampling(g["o_" + k], g["d_" + k], g["z_" + k], g["w_" + k], nf, u=g["u_" + k], tot=g["tot_" + k])
        np.testing.assert_array_equal(zf, g["zf_" + k])
        np.testing.assert_array_equal(pts, g["pf_" + k])


@pytest.mark.parametrize("nc,nf", [(16, 8), (32, 64), (64, 64), (48, 200)])
def test_fine_sampling_shapes(nc, nf):
    g = load_golden("g4_sampler.npz")
    zf, pts = O.fine_sampling(g[f"o_{nc}_{nf}"], g[f"d_{nc}_{nf}"], g[f"z_{nc}_{nf}"], g[f"w_{nc}_{nf}"], nf)
    assert zf.shape == (48, nc + nf)
    assert np.mean(np.abs(zf - g[f"zf_{nc}_{nf}"]) > 5e-6) <= 1e-2
    assert np.mean(np.abs(pts - g[f"pf_{nc}_{nf}"]) > 2e-5) <= 1e-2


# ---------------------------------------------------------------- a3
def _nerf_nets():
    return syn.make_scene_nets(101)


@pytest.mark.parametrize("tag,near,far,wb", [("nf14", 1.0, 4.0, 0), ("nf1631wb", 1.6, 3.1, 1)])
def test_nerf_pipeline_subset(tag, near, far, wb):
    g = load_golden("g5_nerf_pipeline.npz")
    pc, pf = _nerf_nets()
    data = syn.frame_batch(128, 128, phi=0.0, theta=0.0, seed=7, near=near, far=far)
    sub = g[f"sub_{tag}"]
    args = O.Args(white_background=wb)
    pe, de = O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0)
    rgb, rgb_fine, pts_fine, alpha_fine = O.nerf_pipeline_forward(pc, pf, args, pe, de, [a[sub] for a in data])
    assert maxabs(rgb, g[f"rgb_{tag}"][sub]) <= 1e-5
    assert maxabs(rgb_fine, g[f"rgb_fine_{tag}"][sub]) <= 1e-4       # north_star tolerance
    ref_pts = g[f"pts_fine_sub_{tag}"]
    # sample positions are continuous in the coarse weights; allow the fp32 noise floor
    assert np.mean(np.abs(pts_fine - ref_pts) > 1e-4) <= 0.02
    mse = float(np.mean((rgb_fine.astype(np.float64) - g[f"rgb_fine_{tag}"][sub]) ** 2))
    assert mse < 1e-10


def test_nerf_pipeline_coarse_only():
    g = load_golden("g5_nerf_pipeline.npz")
    pc, pf = _nerf_nets()
    data = syn.frame_batch(128, 128, phi=0.0, theta=0.0, seed=7, near=1.0, far=4.0)
    sub = g["sub_nf14"]
    pe, de = O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0)
    out = O.nerf_pipeline_forward(pc, pf, O.Args(run_fine=0), pe, de, [a[sub] for a in data])
    assert out[0] is out[1]
    assert maxabs(out[0], g["coarse_only_rgb"]) <= 1e-5
    assert maxabs(out[3], g["coarse_only_alpha"]) <= 1e-5
    assert out[2].shape == (256, 64, 3)


# ---------------------------------------------------------------- a7
@pytest.mark.parametrize("wb", [0, 1])
def test_smpl_nerf_pipeline(wb):
    g = load_golden("g6_smpl_nerf_pipeline.npz")
    pc, pf = _nerf_nets()
    pw = syn.make_warp_field_params(103, out_scale=0.3)
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    sub = g["sub"]
    pe, de, he = O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0), O.PositionalEncoder(10, 0)
    d = [a[sub] for a in data[:4]] + [g["goal_pose"], data[4][sub]]
    out = O.smpl_nerf_pipeline_forward(pc, pf, pw, O.Args(white_background=wb), pe, de, he, d)
    names = ("rgb", "rgb_fine", "warp_fine", "pts_fine", "warped_fine", "alpha_fine")
    assert maxabs(out[0], g[f"rgb_wb{wb}"]) <= 1e-5
    assert maxabs(out[1], g[f"rgb_fine_wb{wb}"]) <= 1e-4
    for i in (2, 3, 4):
        assert np.mean(np.abs(out[i] - g[f"{names[i]}_wb{wb}"]) > 1e-4) <= 0.02


def test_smpl_nerf_pipeline_coarse_only():
    g = load_golden("g6_smpl_nerf_pipeline.npz")
    pc, pf = _nerf_nets()
    pw = syn.make_warp_field_params(103, out_scale=0.3)
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    sub = g["sub"]
    pe, de, he = O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0), O.PositionalEncoder(10, 0)
    d = [a[sub] for a in data[:4]] + [g["goal_pose"], data[4][sub]]
    out = O.smpl_nerf_pipeline_forward(pc, pf, pw, O.Args(run_fine=0), pe, de, he, d)
    assert maxabs(out[0], g["coarse_rgb"]) <= 1e-5
    assert maxabs(out[2], g["coarse_warp"]) <= 2e-6
    assert maxabs(out[4], g["coarse_warped"]) <= 2e-6
    assert maxabs(out[5], g["coarse_alpha"]) <= 5e-5   # warp round-off (2e-6) re-enters the 2^9 band of the encoder


# ---------------------------------------------------------------- a8
def _av_setup():
    import torch
    from smpl_nerf_amd.synthetic_smpl import LinearBodyModel
    g = load_golden("g9_append_vertices.npz")
    poses = syn.human_poses((41, 38), 0, 60, 10)
    body = LinearBodyModel(seed=3)
    verts = body(body_pose=torch.from_numpy(poses[g["images"]])).vertices.numpy()
    data = syn.frame_batch(128, 128, phi=3.0, theta=-10.0, seed=11)
    return g, verts, [a[g["sub"]] for a in data]


@pytest.mark.parametrize("wb", [0, 1])
def test_append_vertices_pipeline_coarse(wb):
    g, verts, d = _av_setup()
    assert g["reference_fine_branch_runs"][0] == 0      # the reference raises with run_fine=1 (pipeline.py:71)
    pc, pf = syn.make_append_vertices_params(201), syn.make_append_vertices_params(202)
    out = O.append_vertices_pipeline_forward(pc, pf, verts, O.Args(white_background=wb, run_fine=0),
                                             O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0), d)
    assert out[0] is out[1]
    assert maxabs(out[0], g[f"coarse_rgb_wb{wb}"]) <= 1e-5
    assert maxabs(out[3], g[f"coarse_alpha_wb{wb}"]) <= 1e-5


def test_append_vertices_net_rows():
    g = load_golden("g9_append_vertices.npz")
    out = O.append_vertices_net_forward(syn.make_append_vertices_params(201), g["net_rows"])
    assert maxabs(out, g["net_out"]) <= 2e-5 * max(1.0, float(np.abs(g["net_out"]).max()))


# ---------------------------------------------------------------- f-4 pose-append pipelines
@pytest.mark.parametrize("name,npose", [("smpl", 69), ("two", 2)])
@pytest.mark.parametrize("enc", [0, 1])
def test_append_pose_pipelines(name, npose, enc):
    g = load_golden("g10_append_pose.npz")
    add = npose * (20 if enc else 1)
    pc = syn.make_scene_net_params(301 + enc, add_first=True, additional_input_dim=add)
    pf = syn.make_scene_net_params(303 + enc, add_first=True, additional_input_dim=add)
    data = syn.frame_batch(128, 128, phi=3.0, theta=-10.0, seed=11)
    d = [a[g["sub"]] for a in data[:4]] + [g["goal_pose"], data[4][g["sub"]]]
    E = O.PositionalEncoder
    out = O.append_pose_pipeline_forward(pc, pf, O.Args(human_pose_encoding=enc), E(10, 0), E(4, 0), E(10, 0), d,
                                         two_joints=(name == "two"))
    assert maxabs(out[0], g[f"{name}{enc}_rgb"]) <= 1e-5
    assert maxabs(out[1], g[f"{name}{enc}_rgb_fine"]) <= 1e-4
    assert np.mean(np.abs(out[2] - g[f"{name}{enc}_pts_fine"]) > 1e-4) <= 0.02


# ---------------------------------------------------------------- adjacent: rays / coarse samples
def test_rays_and_coarse_sampling():
    g = load_golden("g8_rays.npz")
    np.testing.assert_allclose(syn.sphere_pose(0, 0, 2.4), g["pose_ref_0_0"], atol=1e-12)
    np.testing.assert_allclose(syn.sphere_pose(-20.0, 75.0, 2.4), g["pose_ref_b"], atol=1e-12)
    o, d = syn.camera_rays(16, 24, g["pose"])
    np.testing.assert_allclose(o.reshape(16, 24, 3), g["rays_o"], atol=0)
    np.testing.assert_allclose(d.reshape(16, 24, 3), g["rays_d"], atol=1e-15)
    rows = g["rows"]
    pts, o32, d32, z = syn.coarse_samples(o[rows], d[rows], 1.0, 4.0, 64, g["jitter"])
    np.testing.assert_array_equal(z, g["z"])
    np.testing.assert_array_equal(pts, g["samples"])
    np.testing.assert_array_equal(o32, g["o"])
    np.testing.assert_array_equal(d32, g["d"])
    ro, rd = O.get_rays(16, 24, syn.focal_length(24), g["pose"])
    np.testing.assert_array_equal(rd, g["rays_d"])
    p2, _, _, z2 = O.coarse_sampling(o[rows], d[rows], 1.0, 4.0, 64, g["jitter"])
    np.testing.assert_array_equal(z2, g["z"])
    np.testing.assert_array_equal(p2, g["samples"])


# ---------------------------------------------------------------- the torch-CPU restatement bench.py times as cpu_baseline
@pytest.mark.parametrize("tag,near,far,wb", [("nf14", 1.0, 4.0, 0), ("nf1631wb", 1.6, 3.1, 1)])
def test_torch_cpu_path_nerf_pipeline_is_the_reference_bit_for_bit(tag, near, far, wb):
    """oracle/torch_cpu_path.py issues the reference's own ATen calls in the reference's order: on the host that produced
    the fixtures it reproduces the reference's frames exactly (calibrate_cpu_baseline.py re-checks that against the live
    reference); elsewhere (another torch build / SIMD width) within the fixtures' tolerance."""
    import torch
    from oracle import torch_cpu_path as TP
    g = load_golden("g5_nerf_pipeline.npz")
    pc, pf = _nerf_nets()
    data = syn.frame_batch(128, 128, phi=0.0, theta=0.0, seed=7, near=near, far=far)
    sub = g[f"sub_{tag}"]
    with torch.no_grad():
        out = TP.nerf_pipeline_forward(TP.tparams(pc), TP.tparams(pf), TP.Args(white_background=wb),
                                       TP.PositionalEncoder(10, False), TP.PositionalEncoder(4, False),
                                       [torch.from_numpy(np.ascontiguousarray(a[sub])) for a in data])
    assert maxabs(out[0].numpy(), g[f"rgb_{tag}"][sub]) <= 1e-6
    assert maxabs(out[1].numpy(), g[f"rgb_fine_{tag}"][sub]) <= 1e-5
    assert np.mean(np.abs(out[2].numpy() - g[f"pts_fine_sub_{tag}"]) > 1e-4) <= 0.002


@pytest.mark.parametrize("wb", [0, 1])
def test_torch_cpu_path_smpl_nerf_pipeline(wb):
    import torch
    from oracle import torch_cpu_path as TP
    g = load_golden("g6_smpl_nerf_pipeline.npz")
    pc, pf = _nerf_nets()
    pw = syn.make_warp_field_params(103, out_scale=0.3)
    data = syn.frame_batch(128, 128, phi=5.0, theta=15.0, seed=9)
    sub = g["sub"]
    d = [torch.from_numpy(np.ascontiguousarray(a[sub])) for a in data[:4]] + [torch.from_numpy(g["goal_pose"]),
                                                                              torch.from_numpy(data[4][sub])]
    enc = TP.PositionalEncoder
    with torch.no_grad():
        out = TP.smpl_nerf_pipeline_forward(TP.tparams(pc), TP.tparams(pf), TP.tparams(pw), TP.Args(white_background=wb),
                                            enc(10, False), enc(4, False), enc(10, False), d)
    names = ("rgb", "rgb_fine", "warp_fine", "pts_fine", "warped_fine", "alpha_fine")
    assert maxabs(out[0].numpy(), g[f"rgb_wb{wb}"]) <= 1e-6
    assert maxabs(out[1].numpy(), g[f"rgb_fine_wb{wb}"]) <= 1e-5
    for i in (2, 3, 4):
        assert np.mean(np.abs(out[i].numpy() - g[f"{names[i]}_wb{wb}"]) > 1e-4) <= 0.002


def test_cpu_baseline_calibration_record():
    """The committed calibration of the CPU baseline against the live reference (oracle/calibrate_cpu_baseline.py)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "cpu_baseline_calibration.json")) as f:
        c = json.load(f)
    assert c["outputs_bit_identical"] is True
    assert 0.9 <= c["port_over_reference_speed"] <= 1.1
