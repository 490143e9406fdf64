"""Drop-in replacement of the reference's native extension package `torchsearchsorted`
(torchsearchsorted/src/torchsearchsorted/__init__.py, searchsorted.py:20-53): put this directory's parent
(`shims/`) on sys.path and unmodified reference code - `from torchsearchsorted import searchsorted`
(utils.py:14, called at utils.py:212) - runs on libsmplnerf_hip.so's snerf_searchsorted_f32.

Same signature, asserts and return value as the reference function.  CUDA (ROCm) tensors only: there is no CPU
implementation behind this package (the reference's CPU build of the extension is what the oracle restates).
"""
from smpl_nerf_amd.ops import searchsorted  # noqa: F401

__all__ = ["searchsorted"]
