"""Builds smpl_nerf_amd/csrc/libsmplnerf_hip.so for gfx950 with hipcc (in-tree, no torch involved).

    python -m smpl_nerf_amd.build [--force]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libsmplnerf_hip.so")
SOURCES = ["api.hip", "searchsorted.hip", "posenc.hip", "composite.hip", "sampler.hip", "mlp.hip", "mlp_train.hip", "warp.hip", "raygen.hip", "mlp_bf16.hip", "render.hip", "mlp_train_bf16.hip", "warp_bf16.hip", "train_step.hip", "contract.hip", "mlp_lat.hip", "dp_comm.hip", "linear.hip"]
HEADERS = ["exports.map", "snerf_common.h", "mlp_plan.h", "mlp_device.h", "mlp_bf16_device.h", "mlp_train_device.h", "mlp_lat_device.h", "warp_plan.h", os.path.join("..", "..", "include", "smplnerf.h")]
# -fvisibility=hidden: the SNERF_API entry points of include/smplnerf.h are the library's only dynamic symbols
# -ffp-contract=off: the HBM-bound ops reproduce the reference's eager (unfused) fp32 op order.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# mlp_bf16.hip: keep the operand-split subtractions scalar - SLP-packed v_pk_add_f32 beside MFMAs costs matrix-pipe
# issue slots (MI355X_MICROARCH.md, per-instruction constants)
EXTRA_FLAGS = {"mlp_bf16.hip": ["-fno-slp-vectorize"], "mlp_train_bf16.hip": ["-fno-slp-vectorize"],
               "warp_bf16.hip": ["-fno-slp-vectorize"]}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libsmplnerf_hip.so)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    cc = hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [cc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", LIB] + objs + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
