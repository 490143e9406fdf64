"""smpl_nerf_amd - MI355X-native NeRF ray-march path (HIP kernels behind a C-ABI).

Importing the package is cheap and does not touch the GPU; the HIP library is loaded on first
use of an op (see smpl_nerf_amd._lib) and its absence is a hard error, never a CPU fallback.
"""
__version__ = "0.1.0"
