"""smpl_nerf_amd.dropin against the real reference checkout (build container only: skipped where /root/reference is
absent, i.e. on the GPU box).  Wiring only - no compute: after install() the reference's own Solver / inference modules
hold the HIP-backed classes and operators."""
import importlib
import importlib.util
import os
import sys

import pytest

REF = os.environ.get("SNERF_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference checkout not present")


def test_install_rebinds_the_reference_modules():
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    before = set(sys.modules)
    for name in ("cv2", "imageio"):                      # third-party modules the image lacks; never called here
        mg._stub(name)
    tm = mg._stub("trimesh")
    tm.base = mg._stub("trimesh.base", Trimesh=object)
    tm.ray = mg._stub("trimesh.ray")
    tm.ray.ray_triangle = mg._stub("trimesh.ray.ray_triangle", RayMeshIntersector=object)
    try:
        importlib.import_module("torch.utils.tensorboard")
    except Exception:
        mg._stub("torch.utils.tensorboard", SummaryWriter=object)
    saved = {k: sys.modules.get(k) for k in list(sys.modules)
             if k.split(".")[0] in ("utils", "models", "solver", "torchsearchsorted")}
    saved_path, saved_meta = list(sys.path), list(sys.meta_path)
    from smpl_nerf_amd import dropin, nets, ops, pipelines
    try:
        for k in list(saved):
            sys.modules.pop(k, None)
        dropin._installed = False
        dropin._originals.clear()
        dropin.install(REF)
        import torchsearchsorted
        assert torchsearchsorted.searchsorted is ops.searchsorted
        U = importlib.import_module("utils")                       # the reference's utils.py (imports torchsearchsorted)
        assert U.searchsorted is ops.searchsorted and U.raw2outputs is ops.raw2outputs
        assert U.fine_sampling is ops.fine_sampling and U.sample_pdf is ops.sample_pdf
        NS = importlib.import_module("solver.nerf_solver")         # solver/nerf_solver.py:5 `from models.nerf_pipeline import ...`
        assert NS.NerfPipeline is pipelines.NerfPipeline
        SS = importlib.import_module("solver.smpl_nerf_solver")
        assert SS.SmplNerfPipeline is pipelines.SmplNerfPipeline
        M = importlib.import_module("models.render_ray_net")
        assert M.RenderRayNet is nets.RenderRayNet
        NP = importlib.import_module("models.nerf_pipeline")
        assert NP.NerfPipeline is pipelines.NerfPipeline and NP.raw2outputs is ops.raw2outputs
        assert importlib.import_module("models.warp_field_net").WarpFieldNet is nets.WarpFieldNet
        AP = importlib.import_module("models.append_smpl_params_pipeline")
        assert AP.AppendSmplParamsPipeline is pipelines.AppendSmplParamsPipeline
        # the reference's own encoder class is kept and is what our pipelines accept
        enc = U.PositionalEncoder(10, False)
        pipe = NS.NerfPipeline(nets.RenderRayNet(), nets.RenderRayNet(), object(), enc, U.PositionalEncoder(4, False))
        d = pipe.model_coarse.desc_for_encoders(pipe.position_encoder, pipe.direction_encoder)
        assert (d.pos_freqs, d.dir_freqs, d.pos_identity) == (10, 4, 0)
        assert dropin.install(REF) == 0                            # idempotent
    finally:
        sys.meta_path[:] = saved_meta
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "models", "solver", "torchsearchsorted")]:
            sys.modules.pop(k, None)
        sys.modules.update({k: v for k, v in saved.items() if v is not None})
        dropin._installed = False
        for k in set(sys.modules) - before:               # the stand-ins and whatever the reference pulled in
            if k.split(".")[0] in ("cv2", "imageio", "trimesh", "camera", "render", "make_golden") or k == "torch.utils.tensorboard":
                sys.modules.pop(k, None)


def test_install_normalises_the_root(tmp_path):
    """a symlinked root matches the (absolute, resolved) module paths; a sibling directory whose name merely
    starts with the root's does not; a missing directory is an error."""
    from smpl_nerf_amd import dropin
    link = tmp_path / "ref_link"
    os.symlink(REF, link)
    root = dropin._norm_root(str(link))
    assert root == os.path.join(os.path.realpath(REF), "")
    assert dropin._under(os.path.join(REF, "utils.py"), root)
    assert not dropin._under(os.path.realpath(REF) + "2/utils.py", root)
    with pytest.raises(FileNotFoundError):
        dropin.install(str(tmp_path / "nope"))


def test_install_wraps_the_render_entry_point_in_no_grad(tmp_path):
    """inference.inference (inference.py:222-265) calls pipeline(data) with autograd recording on (:247-253); the drop-in
    runs it under torch.no_grad() so that the pipelines take their inference kernels.  A stand-in module of that name under a
    scratch root (the reference's own inference.py needs imageio / torchvision at import)."""
    import torch
    from smpl_nerf_amd import dropin
    root = tmp_path / "checkout"
    root.mkdir()
    (root / "inference.py").write_text("import torch\n\ndef inference(x=1):\n    'render'\n    return torch.is_grad_enabled(), x\n")
    saved_path, saved_meta, saved_mod = list(sys.path), list(sys.meta_path), sys.modules.pop("inference", None)
    try:
        dropin._installed = False
        sys.path.insert(0, str(root))
        import inference as inf                                    # imported BEFORE install: swept
        assert inf.inference() == (True, 1)
        assert dropin.install(str(root)) >= 1
        assert inf.inference(5) == (False, 5) and inf.inference.__doc__ == "render"
        assert torch.is_grad_enabled()
        assert dropin.install(str(root)) == 0                      # idempotent: not wrapped twice
        sys.modules.pop("inference")
        inf2 = importlib.import_module("inference")                # imported AFTER install: patched on import
        assert inf2.inference() == (False, 1)
    finally:
        sys.meta_path[:] = saved_meta
        sys.path[:] = saved_path
        sys.modules.pop("inference", None)
        if saved_mod is not None:
            sys.modules["inference"] = saved_mod
        dropin._installed = False
