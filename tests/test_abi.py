"""CPU-side checks of the C-ABI library: it builds, loads, exports every symbol the public header
declares, and validates arguments before touching a device.  No GPU needed, no compute launched."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from smpl_nerf_amd import _lib, build


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "smplnerf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(snerf_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    syms = header_symbols()
    assert len(syms) >= 12
    assert sorted(_lib.SIGNATURES) == syms          # the ctypes table lists exactly the header's functions
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/smplnerf.h but not exported"


def test_version_and_error_string(lib):
    assert lib.snerf_version() == 109
    assert isinstance(lib.snerf_last_error_string(), bytes)
    assert lib.snerf_device_count() >= 0


def test_only_the_header_symbols_are_exported(lib):
    """-fvisibility=hidden + SNERF_API: `nm -D --defined-only` of the library lists the header's entry points and nothing else
    (no C++ symbols of namespace snerf, no template instantiations a host's own code could collide with)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert names == header_symbols(), sorted(set(names) ^ set(header_symbols()))
    assert lib.snerf_shutdown() == 0 and lib.snerf_shutdown() == 0          # idempotent, nothing to destroy on a CPU box


def test_argument_validation_happens_on_the_host(lib):
    # mismatched row counts (searchsorted.py:23-28) -> BADARG, nothing launched
    rc = lib.snerf_searchsorted_f32(None, 3, 4, None, 2, 4, None, 0, None)
    assert rc == -1 and b"rows" in lib.snerf_last_error_string()
    assert lib.snerf_searchsorted_f32(None, 0, 4, None, 0, 4, None, 0, None) == 0      # empty is a no-op
    assert lib.snerf_posenc_f32(None, 5, 0, 10, 0, None, None) == -1
    assert lib.snerf_composite_fwd_f32(None, None, None, 0, None, 4, 0, 0, None, None, None, None) == -1
    assert lib.snerf_sample_pdf_f32(None, None, None, None, None, 4, 2, 128, None, None, None, None, None) == -1
    assert lib.snerf_sample_pdf_f32(None, None, None, None, None, 0, 64, 128, None, None, None, None, None) == 0


def test_mlp_descriptor_arithmetic(lib):
    d = _lib.MlpDesc(8, 256, 10, 0, 4, 0, 0, 1 << 4, 1)
    assert lib.snerf_mlp_param_floats(d) == 610436        # skips=[4]
    d0 = _lib.MlpDesc(8, 256, 10, 0, 4, 0, 0, 0, 1)
    assert lib.snerf_mlp_param_floats(d0) == 595076       # skips=[]
    # 77 slabs of 33 KiB (32 A tiles + bias) + 3 pad slabs (mlp_plan.h)
    assert lib.snerf_mlp_packed_floats(d) == (77 + 3) * 8448
    # any width up to 256 (other widths than 128 / 256 run zero-padded): the parameter count is RenderRayNet's own
    w200 = _lib.MlpDesc(8, 200, 10, 0, 4, 0, 0, 0, 1)
    assert lib.snerf_mlp_param_floats(w200) == (200 * 61 + 7 * 200 * 201 + 200 * 201 + 201 + 100 * 225 + 100 * 101 + 3 * 101)
    assert lib.snerf_mlp_packed_floats(w200) == lib.snerf_mlp_packed_floats(d0)       # the stream of the 256-wide kernel
    # above 256 (r05): kernels of 320 / 384 / 448 / 512 features (other widths zero-padded); a slab is one k-block of all output tiles
    w320 = _lib.MlpDesc(8, 320, 10, 0, 4, 0, 0, 0, 1)
    assert lib.snerf_mlp_param_floats(w320) == (320 * 61 + 7 * 320 * 321 + 320 * 321 + 321 + 160 * 345 + 160 * 161 + 3 * 161)
    w300 = _lib.MlpDesc(8, 300, 10, 0, 4, 0, 0, 0, 1)
    slabs320 = 4 + 7 * 20 + 20 + 1 + (20 + 2 + 2) // 3 + (10 + 2) // 3 + 1    # k-blocks per layer / (32 tiles per slab // output tiles)
    assert lib.snerf_mlp_packed_floats(w320) == lib.snerf_mlp_packed_floats(w300) == (slabs320 + 3) * 8448
    w512 = _lib.MlpDesc(8, 512, 10, 0, 4, 0, 0, 0, 1)
    slabs512 = 4 + 7 * 32 + 32 + 1 + (32 + 2) // 2 + 16 // 2 + 1
    assert lib.snerf_mlp_packed_floats(w512) == (slabs512 + 3) * 8448
    bad = _lib.MlpDesc(8, 640, 10, 0, 4, 0, 0, 0, 1)
    assert lib.snerf_mlp_param_floats(bad) < 0
    assert lib.snerf_mlp_pack_f32(bad, None, None, None) == -1 and b"width must be in [2, 512]" in lib.snerf_last_error_string()
    from smpl_nerf_amd.nets import RenderRayNet, WarpFieldNet
    RenderRayNet(8, 320, 60, 24)                                 # config_parser.py:20 accepts it, and so do the kernels since r05
    # r06: any width constructs - above the fused kernels' limits (the snerf_mlp_* entries above stay at 512: SNERF_E_BADARG) the net
    # runs one nn.Linear at a time through snerf_linear_* (smpl_nerf_amd/layered.py)
    assert RenderRayNet(8, 640, 60, 24)._layered and not RenderRayNet(8, 512, 60, 24)._layered
    assert WarpFieldNet(8, 512, 60, 40)._layered and not WarpFieldNet(8, 256, 60, 40)._layered
    assert RenderRayNet(20, 64, 60, 24)._layered                 # ... and any --netdepth (config_parser.py:19)
    with pytest.raises(ValueError):
        RenderRayNet(8, 1, 60, 24)
    assert lib.snerf_linear_bwd_weight_scratch_floats(1000, 768, 828) == 4 * (768 * 828 + 768)
    assert lib.snerf_mlp_packed_bf16_bytes(w200, 3) < 0     # the split-precision entry points: width 256 only
    assert lib.snerf_mlp_pack_f32(d, None, None, None) == -1


def test_missing_library_is_a_hard_error(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsmplnerf_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected():
    import torch
    from smpl_nerf_amd import ops
    a = torch.sort(torch.rand(4, 8), dim=1)[0]
    v = torch.rand(4, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.searchsorted(a, v)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.PositionalEncoder(4, 0).encode(torch.rand(5, 3))
    with pytest.raises(AssertionError):
        ops.searchsorted(torch.rand(3, 4), torch.rand(2, 4))       # reference's shape assert


def test_directions_argument_views():
    import torch
    from smpl_nerf_amd import ops
    d = torch.rand(5, 3)
    t, per = ops._directions_arg(d[:, None, :].expand(5, 7, 3), 5, 7)
    assert per == 0 and t.shape == (5, 3) and torch.equal(t, d)
    t, per = ops._directions_arg(torch.rand(5, 7, 3), 5, 7)
    assert per == 1 and t.shape == (5, 7, 3)
    t, per = ops._directions_arg(d, 5, 7)
    assert per == 0


def test_render_ray_net_state_dict_contract():
    """Checkpoints written by the reference (utils.py:282-283) must load: same keys, same shapes."""
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd import synthetic as syn
    net = RenderRayNet(8, 256, 60, 24, skips=[4])
    sd = net.state_dict()
    expect = []
    for name, fo, fi in syn.render_ray_net_shapes(skips=(4,)):
        expect += [(name + ".weight", (fo, fi)), (name + ".bias", (fo,))]
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == expect
    assert sum(v.numel() for v in sd.values()) == 610436
    d = net.desc_for_encoded()
    assert (d.pos_freqs, d.pos_identity, d.dir_freqs, d.dir_identity, d.add_dim, d.skip_mask) == (10, 0, 4, 0, 0, 16)
    odd = RenderRayNet(4, 128, 5, 24, additional_input_dim=2, skips=[1])
    d = odd.desc_for_encoded()
    assert (d.pos_freqs, d.add_dim) == (0, 7)


def test_torchsearchsorted_shim_imports_and_has_no_cpu_path():
    """shims/torchsearchsorted is the package reference code imports (utils.py:14); on CPU tensors it must raise -
    the product has no CPU fallback."""
    import importlib
    import sys
    import numpy as np
    import pytest
    import torch
    shims = os.path.join(ROOT, "shims")
    sys.path.insert(0, shims)
    try:
        sys.modules.pop("torchsearchsorted", None)
        mod = importlib.import_module("torchsearchsorted")
        assert mod.__file__.startswith(shims)
        a = torch.from_numpy(np.sort(np.random.default_rng(0).random((3, 9)).astype(np.float32), 1))
        v = torch.rand(3, 5)
        with pytest.raises(RuntimeError):
            mod.searchsorted(a, v, side="right")
        with pytest.raises(AssertionError):          # the reference's own shape asserts come first
            mod.searchsorted(a[0], v)
    finally:
        sys.path.remove(shims)
        sys.modules.pop("torchsearchsorted", None)


def test_split_bf16_and_render_entry_points_validate_on_the_host(lib):
    """Host-side argument checks of the entry points added for the bf16 matrix-core path, the training variants and the
    one-call renderer: descriptor arithmetic, nsplit / precision ranges, null pointers - nothing is launched."""
    d = _lib.MlpDesc(8, 256, 10, 0, 4, 0, 0, 1 << 4, 1)
    # slab = one 32-wide k-block x 16 tiles x nsplit parts (+1 KiB bias): 79 slabs + 3 pad slabs for this net
    b3, b2 = lib.snerf_mlp_packed_bf16_bytes(d, 3), lib.snerf_mlp_packed_bf16_bytes(d, 2)
    assert b3 % (3 * 16384 + 1024) == 0 and b2 % (2 * 16384 + 1024) == 0
    assert b3 // (3 * 16384 + 1024) == b2 // (2 * 16384 + 1024) > 3
    assert lib.snerf_mlp_packed_bf16_bytes(d, 4) == -1 and b"nsplit" in lib.snerf_last_error_string()
    # nsplit = SNERF_SPLIT_F16X3 (16): two fp16 parts - the layout of nsplit = 2, accepted by every split entry point
    assert _lib.SPLIT_F16X3 == 16
    assert lib.snerf_mlp_packed_bf16_bytes(d, _lib.SPLIT_F16X3) == b2
    assert lib.snerf_mlp_packed_t_bf16_bytes(d, _lib.SPLIT_F16X3, 1) == lib.snerf_mlp_packed_t_bf16_bytes(d, 2, 1)
    assert lib.snerf_mlp_fwd_bf16_f32(d, None, _lib.SPLIT_F16X3, None, None, 0, None, 0, 64, None, None) == 0
    assert lib.snerf_mlp_bwd_bf16_f32(d, None, _lib.SPLIT_F16X3, None, None, 0, None, None, None, None) == 0
    narrow = _lib.MlpDesc(4, 128, 10, 0, 4, 0, 0, 2, 1)
    assert lib.snerf_mlp_packed_bf16_bytes(narrow, 3) == -1 and b"256" in lib.snerf_last_error_string()
    assert lib.snerf_mlp_packed_t_bf16_bytes(d, 3, 0) > 0
    assert lib.snerf_mlp_packed_t_bf16_bytes(d, 3, 1) > lib.snerf_mlp_packed_t_bf16_bytes(d, 3, 0)   # + encoder columns
    assert lib.snerf_mlp_packed_t_bf16_bytes(d, 1, 0) == -1
    assert lib.snerf_mlp_pack_bf16(d, None, None, 3, None) == -1
    assert lib.snerf_mlp_pack_t_bf16(d, None, None, 3, 0, None) == -1
    assert lib.snerf_mlp_fwd_bf16_f32(d, None, 3, None, None, 0, None, 0, 64, None, None) == 0        # n = 0: no-op
    assert lib.snerf_mlp_fwd_bf16_f32(d, None, 3, None, None, 0, None, 128, 64, None, None) == -1     # null pointers
    assert lib.snerf_mlp_fwd_train_bf16_f32(d, None, 5, None, None, 0, None, 128, 64, None, None, None) == -1
    assert lib.snerf_mlp_bwd_bf16_f32(d, None, 3, None, None, 0, None, None, None, None) == 0
    assert lib.snerf_mlp_bwd_bf16_f32(d, None, 3, None, None, 128, None, None, None, None) == -1
    assert lib.snerf_mlp_bwd_inputs_bf16_f32(d, None, 3, None, None, None, None, 0, 1, 128, None, None, None, None, None,
                                             None) == -1
    # one-call renderer
    ws = lib.snerf_render_rays_workspace_bytes(16384, 64, 128)
    assert ws >= 16384 * (64 * 16 + 64 * 8 + 128 * 4 + 192 * 4 + 192 * 16) and ws % 16 == 0
    assert lib.snerf_render_rays_workspace_bytes(16384, 0, 128) == -1
    args = [d, None, d, None, 3] + [None] * 7 + [0, 64, 128, 0, None, None, None, None, None, None]
    assert lib.snerf_render_rays_f32(*args) == 0                                                         # B = 0: no-op
    args[12] = 4
    assert lib.snerf_render_rays_f32(*args) == -1 and b"null" in lib.snerf_last_error_string()
    args[4] = 1
    assert lib.snerf_render_rays_f32(*args) == -1 and b"precision" in lib.snerf_last_error_string()


def test_pipeline_set_precision_reaches_every_net():
    import torch
    from smpl_nerf_amd.nets import RenderRayNet, WarpFieldNet
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import PipelineArgs, SmplNerfPipeline
    enc = PositionalEncoder(10, 0)
    pipe = SmplNerfPipeline(RenderRayNet(), RenderRayNet(), WarpFieldNet(8, 256, 60, 40), PipelineArgs(), enc,
                            PositionalEncoder(4, 0), enc)
    assert pipe.set_precision("bf16x6") is pipe
    assert {pipe.model_coarse.precision, pipe.model_fine.precision, pipe.model_warp_field.precision} == {"bf16x6"}
    with pytest.raises(ValueError):
        pipe.set_precision("fp16")


def test_training_buffer_sizes_of_widths_above_256(lib):
    """Host-side layout of the 320 .. 512 kernels (csrc/mlp_plan.h: make_train_layout): tile-rows of 16 features per layer input,
    one tile-row per ReLU sign mask (four words per lane) instead of half a row, d Y rows per layer output; the transposed stream
    of the dgrad; a width is laid out as the next kernel width above it (config_parser.py:20 --netwidth)."""
    import ctypes

    def sizes(width, n, depth=8, skip_mask=1 << 4):
        d = _lib.MlpDesc(depth, width, 10, 0, 4, 0, 0, skip_mask, 1)
        v = [ctypes.c_int64() for _ in range(4)]
        cnt = ctypes.c_int32()
        assert lib.snerf_mlp_train_sizes(d, n, *[ctypes.byref(x) for x in v], ctypes.byref(cnt)) == 0
        return [x.value for x in v] + [cnt.value]

    n = 1000
    for width, T in ((512, 32), (500, 32), (448, 28), (384, 24), (330, 24), (320, 20), (257, 20)):
        act, dy, pt, gp, cnt = sizes(width, n)
        nh = 7
        rows = 4 + 2 + (nh + 1) * T + T + T // 2 + T // 2 + (nh + 2)      # encoders, x[1..nh+1], o, h1, h2, one row per mask
        assert act >= rows * n * 16 and act - rows * n * 16 <= 64, (width, act, rows)
        dy_rows = (nh + 2) * T + 1 + T // 2 + T // 2 + 1                  # trunk layers + additional, sigma, directional x 2, rgb
        assert dy >= dy_rows * n * 16 and dy - dy_rows * n * 16 <= 64, (width, dy, dy_rows)
        assert cnt == 8 and gp % cnt == 0                                  # wgrad_chunks(1000) partials of one slot-ordered gradient
        assert pt % 8448 == 0                                              # whole slabs
    # the same description below 256 keeps two masks per tile-row
    act256 = sizes(256, n)[0]
    assert act256 >= (4 + 2 + 8 * 16 + 16 + 8 + 8 + 5) * n * 16 and act256 - (4 + 2 + 8 * 16 + 16 + 8 + 8 + 5) * n * 16 <= 64
    # a 512-wide first layer needs two k-blocks of input: 3 identity columns alone are refused by name
    tiny = _lib.MlpDesc(2, 512, 0, 1, 0, 1, 0, 0, 1)
    assert lib.snerf_mlp_param_floats(tiny) < 0
    assert lib.snerf_mlp_pack_f32(tiny, None, None, None) == -1 and b"two k-blocks" in lib.snerf_last_error_string()
