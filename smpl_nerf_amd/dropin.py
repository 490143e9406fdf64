"""Turn-key drop-in: make an UNMODIFIED checkout of HannesStark/SMPL-NeRF run its ray-march path on
libsmplnerf_hip.so.

    import smpl_nerf_amd.dropin as dropin
    dropin.install("/path/to/SMPL-NeRF")      # before (or after) importing the reference's train / inference modules

What it does (INTEGRATION.md sections 1-2 as code):
  * registers a `torchsearchsorted` package whose `searchsorted` is ops.searchsorted (the reference imports it at
    utils.py:14 and calls it at utils.py:212);
  * rebinds, in every module of the reference that is (or later gets) imported, the names of the path's operators and
    plugin classes to their HIP-backed mirrors: utils.{raw2outputs, sample_pdf, fine_sampling, searchsorted},
    models.render_ray_net.RenderRayNet, models.warp_field_net.WarpFieldNet, models.append_vertices_net.AppendVerticesNet,
    models.{nerf,smpl_nerf,append_vertices,append_smpl_params,append_to_nerf}_pipeline.* - including the copies that
    `from x import y` left in solver/*.py, train.py, inference.py.
  * wraps the reference's render entry point, inference.inference (inference.py:222-265), in torch.no_grad(): the reference
    calls `pipeline(data)` there in eval mode but WITH autograd recording (inference.py:247-253) - the graph is built and
    thrown away per batch - and with recording on, these pipelines would run their training forward (every layer input
    stored, 10.7 KB per ray-sample) instead of the inference kernels.  No result changes: nothing in that function
    differentiates.
The reference's PositionalEncoder is left alone: the pipelines only read its number_frequencies / include_identity (the
encoding itself is fused into the MLP kernel), and its encode() works on GPU tensors where a pipeline calls it.

Nothing of the reference is imported unless it already is or `reference_root` is given; nothing is copied.
"""
from __future__ import annotations

import importlib
import importlib.abc
import os
import sys
import types

from . import nets, ops, pipelines

# reference module -> {attribute: replacement}
REPLACEMENTS = {
    "utils": {"raw2outputs": ops.raw2outputs, "sample_pdf": ops.sample_pdf, "fine_sampling": ops.fine_sampling,
              "searchsorted": ops.searchsorted},
    "models.render_ray_net": {"RenderRayNet": nets.RenderRayNet},
    "models.warp_field_net": {"WarpFieldNet": nets.WarpFieldNet},
    "models.append_vertices_net": {"AppendVerticesNet": nets.AppendVerticesNet},
    "models.nerf_pipeline": {"NerfPipeline": pipelines.NerfPipeline},
    "models.smpl_nerf_pipeline": {"SmplNerfPipeline": pipelines.SmplNerfPipeline},
    "models.append_vertices_pipeline": {"AppendVerticesPipeline": pipelines.AppendVerticesPipeline},
    "models.append_smpl_params_pipeline": {"AppendSmplParamsPipeline": pipelines.AppendSmplParamsPipeline},
    "models.append_to_nerf_pipeline": {"AppendToNerfPipeline": pipelines.AppendToNerfPipeline},
}
_NAMES = {name: obj for table in REPLACEMENTS.values() for name, obj in table.items()}
_originals = {}      # name -> the reference's own object (once seen), to recognise `from x import y` copies
_installed = False


def _register_torchsearchsorted():
    mod = types.ModuleType("torchsearchsorted")
    mod.__doc__ = "smpl_nerf_amd drop-in for the reference's native extension package"
    mod.searchsorted = ops.searchsorted
    mod.__all__ = ["searchsorted"]
    sys.modules["torchsearchsorted"] = mod


# reference module -> functions that only render (no backward inside): run under torch.no_grad()
NO_GRAD_FUNCTIONS = {"inference": ("inference",)}


def _no_grad(fn):
    import functools

    import torch

    @functools.wraps(fn)
    def rendered_without_autograd(*args, **kw):
        with torch.no_grad():
            return fn(*args, **kw)

    rendered_without_autograd._snerf_no_grad = True
    return rendered_without_autograd


def _patch_module(mod):
    """Rebind the path's names in one (reference) module; returns how many bindings changed."""
    table = REPLACEMENTS.get(getattr(mod, "__name__", ""), {})
    changed = 0
    for name in NO_GRAD_FUNCTIONS.get(getattr(mod, "__name__", ""), ()):
        cur = mod.__dict__.get(name)
        if callable(cur) and not getattr(cur, "_snerf_no_grad", False):
            setattr(mod, name, _no_grad(cur))
            changed += 1
    for name, new in table.items():           # the defining module: remember the original, then replace it
        cur = mod.__dict__.get(name)
        if cur is not None and cur is not new:
            _originals.setdefault(name, cur)
            setattr(mod, name, new)
            changed += 1
    for name, new in _NAMES.items():          # `from models.x import Y` / `from utils import y` copies elsewhere
        cur = mod.__dict__.get(name)
        if cur is not None and cur is not new and cur is _originals.get(name):
            setattr(mod, name, new)
            changed += 1
    return changed


def _norm_root(root):
    """Absolute, symlink-free root with a trailing separator (module __file__ / spec.origin are absolute paths, so a
    relative or symlinked root would match nothing; the separator keeps '/x/SMPL-NeRF' from matching '/x/SMPL-NeRF2')."""
    return None if not root else os.path.join(os.path.realpath(root), "")


def _under(path, root):
    return bool(path) and (root is None or os.path.realpath(path).startswith(root))


def _is_reference_module(mod, root):
    return _under(getattr(mod, "__file__", None), root)


class _PatchOnImport(importlib.abc.MetaPathFinder):
    """After any later import of a reference module, patch it (and re-sweep: it may have pulled in others)."""

    def __init__(self, root):
        self.root = root
        self._busy = False

    def find_spec(self, fullname, path, target=None):
        if self._busy or not (fullname in REPLACEMENTS or fullname.split(".")[0] in ("solver", "models", "train", "inference",
                                                                                     "utils", "render")):
            return None
        self._busy = True
        try:
            spec = importlib.util.find_spec(fullname)
        finally:
            self._busy = False
        if spec is None or spec.loader is None or not _under(spec.origin, self.root):
            return None
        loader, root = spec.loader, self.root

        class _Loader(importlib.abc.Loader):
            def create_module(self, s):
                return loader.create_module(s)

            def exec_module(self, module):
                loader.exec_module(module)
                sweep(root)

        spec.loader = _Loader()
        return spec


def sweep(reference_root=None) -> int:
    """Patch every already-imported module of the reference (defining modules first); returns the number of rebindings."""
    mods = [m for m in list(sys.modules.values()) if isinstance(m, types.ModuleType) and _is_reference_module(m, reference_root)]
    mods.sort(key=lambda m: 0 if m.__name__ in REPLACEMENTS else 1)
    return sum(_patch_module(m) for m in mods)


def install(reference_root: str = None) -> int:
    """See the module docstring.  Idempotent; returns the number of names rebound by this call."""
    global _installed
    import importlib.util  # noqa: F401  (used by the finder)
    _register_torchsearchsorted()
    if reference_root:
        if not os.path.isdir(reference_root):
            raise FileNotFoundError(f"dropin.install: {reference_root!r} is not a directory")
        reference_root = _norm_root(reference_root)
        if reference_root.rstrip(os.sep) not in [os.path.realpath(p) for p in sys.path if p]:
            sys.path.insert(0, reference_root.rstrip(os.sep))
    n = sweep(reference_root)
    if not _installed:
        sys.meta_path.insert(0, _PatchOnImport(reference_root))
        _installed = True
    if n == 0 and reference_root and importlib.util.find_spec("utils") is None:
        import warnings
        warnings.warn(f"dropin.install({reference_root!r}): nothing rebound and the reference's modules are not importable "
                      "from there - the reference would keep running its own torch path")
    return n
