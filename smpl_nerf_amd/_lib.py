"""ctypes binding of libsmplnerf_hip.so (include/smplnerf.h).

There is exactly one implementation of every op in this package: the HIP library.  If it is
missing, or a call fails, a RuntimeError is raised - there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libsmplnerf_hip.so")
SPLIT_F16X3 = 16   # include/smplnerf.h SNERF_SPLIT_F16X3: nsplit / precision code of the two-part fp16 kernel

SNERF_OK = 0


class MlpDesc(ctypes.Structure):
    """struct snerf_mlp_desc (include/smplnerf.h) - mirrors RenderRayNet.__init__ (models/render_ray_net.py:8)."""
    _fields_ = [("n_layers", c_int32), ("width", c_int32), ("pos_freqs", c_int32), ("pos_identity", c_int32),
                ("dir_freqs", c_int32), ("dir_identity", c_int32), ("add_dim", c_int32), ("skip_mask", c_uint32),
                ("use_dir", c_int32), ("add_first", c_int32)]


class WarpDesc(ctypes.Structure):
    """struct snerf_warp_desc - mirrors WarpFieldNet.__init__ (models/warp_field_net.py:8)."""
    _fields_ = [("width", c_int32), ("pos_freqs", c_int32), ("pos_identity", c_int32), ("pose_dim", c_int32)]


_P = c_void_p  # device pointers travel as integers (tensor.data_ptr())


class AdamState(ctypes.Structure):
    """struct snerf_adam_state - torch.optim.Adam over one flat fp32 buffer (solver/nerf_solver.py:11-14, 31-33)."""
    _fields_ = [("params", _P), ("grads", _P), ("exp_avg", _P), ("exp_avg_sq", _P), ("n_params", c_int64),
                ("scratch", _P), ("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("eps", c_double),
                ("weight_decay", c_double)]


class AdamRange(ctypes.Structure):
    """struct snerf_adam_range - adjacent parameter tensors taking part in a step, with their device step counters."""
    _fields_ = [("begin", c_int64), ("end", c_int64), ("step", _P), ("n_steps", c_int32)]


class AdamNet(ctypes.Structure):
    """struct snerf_adam_net - a RenderRayNet inside the flat buffer whose weight streams the optimiser step keeps current."""
    _fields_ = [("desc", POINTER(MlpDesc)), ("param_offset", c_int64), ("precision", c_int32), ("packed", _P),
                ("packed_t", _P), ("slot_fwd", _P), ("slot_t", _P)]


class NerfBatch(ctypes.Structure):
    """struct snerf_nerf_batch - the Solver's batch (solver/nerf_solver.py:77-81) plus u and the sigma noise."""
    _fields_ = [("ray_samples", _P), ("rays_o", _P), ("rays_d", _P), ("z_vals", _P), ("rgb_truth", _P), ("u", _P),
                ("noise_coarse", _P), ("noise_fine", _P), ("additional", _P), ("B", c_int64), ("Nc", c_int32), ("Nf", c_int32),
                ("white_background", c_int32)]


class InputGrads(ctypes.Structure):      # snerf_input_grads
    _fields_ = [("d_additional", _P), ("params_coarse", _P), ("params_fine", _P)]


# name -> (restype, argtypes); must list every symbol include/smplnerf.h declares
SIGNATURES = {
    "snerf_version": (c_int, []),
    "snerf_last_error_string": (c_char_p, []),
    "snerf_device_count": (c_int, []),
    "snerf_shutdown": (c_int, []),
    "snerf_searchsorted_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int, _P]),
    "snerf_searchsorted": (c_int, [c_int, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int, _P]),
    "snerf_posenc_bwd_f32": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P, _P]),
    "snerf_composite_bwd_all_f32": (c_int, [_P, _P, _P, c_int, _P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "snerf_posenc_f32": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, _P]),
    "snerf_composite_fwd_f32": (c_int, [_P, _P, _P, c_int, _P, c_int64, c_int, c_int, _P, _P, _P, _P]),
    "snerf_composite_bwd_f32": (c_int, [_P, _P, _P, c_int, _P, c_int64, c_int, c_int, _P, _P, _P, _P]),
    "snerf_sample_pdf_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, c_int, _P, _P, _P, _P, _P]),
    "snerf_sample_pdf_strict_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, _P, _P, _P, _P, _P]),
    "snerf_sample_pdf_bins_strict_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "snerf_sample_pdf_bins_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "snerf_sample_pdf_bins_f32": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "snerf_mlp_param_floats": (c_int64, [POINTER(MlpDesc)]),
    "snerf_mlp_packed_floats": (c_int64, [POINTER(MlpDesc)]),
    "snerf_mlp_pack_f32": (c_int, [POINTER(MlpDesc), _P, _P, _P]),
    "snerf_mlp_fwd_f32": (c_int, [POINTER(MlpDesc), _P, _P, _P, c_int, _P, c_int64, c_int, _P, _P]),
    "snerf_mlp_packed_bf16_bytes": (c_int64, [POINTER(MlpDesc), c_int]),
    "snerf_mlp_pack_bf16": (c_int, [POINTER(MlpDesc), _P, _P, c_int, _P]),
    "snerf_mlp_fwd_bf16_f32": (c_int, [POINTER(MlpDesc), _P, c_int, _P, _P, c_int, _P, c_int64, c_int, _P, _P]),
    "snerf_mlp_fwd_train_bf16_f32": (c_int, [POINTER(MlpDesc), _P, c_int, _P, _P, c_int, _P, c_int64, c_int, _P, _P, _P]),
    "snerf_mlp_packed_t_bf16_bytes": (c_int64, [POINTER(MlpDesc), c_int, c_int]),
    "snerf_mlp_pack_t_bf16": (c_int, [POINTER(MlpDesc), _P, _P, c_int, c_int, _P]),
    "snerf_mlp_bwd_bf16_f32": (c_int, [POINTER(MlpDesc), _P, c_int, _P, _P, c_int64, _P, _P, _P, _P]),
    "snerf_mlp_bwd_inputs_bf16_f32": (c_int, [POINTER(MlpDesc), _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int64, _P, _P, _P,
                                              _P, _P, _P]),
    "snerf_mlp_train_sizes": (c_int, [POINTER(MlpDesc), c_int64, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
                                      POINTER(c_int64), POINTER(c_int32)]),
    "snerf_mlp_dy_layout": (c_int, [POINTER(MlpDesc), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "snerf_mlp_fwd_train_f32": (c_int, [POINTER(MlpDesc), _P, _P, _P, c_int, _P, c_int64, c_int, _P, _P, _P]),
    "snerf_mlp_fwd_encoded_train_f32": (c_int, [POINTER(MlpDesc), _P, _P, c_int64, c_int64, _P, _P, _P]),
    "snerf_mlp_pack_t_f32": (c_int, [POINTER(MlpDesc), _P, _P, c_int, _P]),
    "snerf_mlp_bwd_inputs_f32": (c_int, [POINTER(MlpDesc), _P, _P, _P, _P, _P, c_int, c_int, c_int64, _P, _P, _P, _P, _P,
                                         _P]),
    "snerf_mlp_bwd_f32": (c_int, [POINTER(MlpDesc), _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "snerf_warp_param_floats": (c_int64, [POINTER(WarpDesc)]),
    "snerf_warp_packed_floats": (c_int64, [POINTER(WarpDesc)]),
    "snerf_warp_pack_f32": (c_int, [POINTER(WarpDesc), _P, _P, _P]),
    "snerf_warp_fwd_f32": (c_int, [POINTER(WarpDesc), _P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P]),
    "snerf_warp_packed_bf16_bytes": (c_int64, [POINTER(WarpDesc)]),
    "snerf_warp_pack_bf16": (c_int, [POINTER(WarpDesc), _P, _P, _P]),
    "snerf_warp_fwd_bf16_f32": (c_int, [POINTER(WarpDesc), _P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P]),
    "snerf_warp_train_sizes": (c_int, [POINTER(WarpDesc), c_int64, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
                                       POINTER(c_int64)]),
    "snerf_warp_fwd_train_f32": (c_int, [POINTER(WarpDesc), _P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P]),
    "snerf_warp_pack_t_f32": (c_int, [POINTER(WarpDesc), _P, _P, _P]),
    "snerf_warp_bwd_f32": (c_int, [POINTER(WarpDesc), _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "snerf_raygen_f64": (c_int, [_P, c_int64, c_int, c_int, c_double, _P, _P, c_int, _P, _P, c_int64, _P, _P, _P, _P, _P]),
    "snerf_mlp_fwd_encoded_f32": (c_int, [POINTER(MlpDesc), _P, _P, c_int64, c_int64, _P, _P]),
    "snerf_render_rays_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "snerf_render_rays_smpl_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "snerf_render_rays_smpl_f32": (c_int, [POINTER(MlpDesc), _P, POINTER(MlpDesc), _P, POINTER(WarpDesc), _P, c_int, _P, _P, _P, _P,
                                           _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "snerf_render_rays_add_workspace_bytes": (c_int64, [POINTER(MlpDesc), POINTER(MlpDesc), c_int64, c_int, c_int]),
    "snerf_render_rays_add_f32": (c_int, [POINTER(MlpDesc), _P, POINTER(MlpDesc), _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int64,
                                          c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "snerf_render_rays_f32": (c_int, [POINTER(MlpDesc), _P, POINTER(MlpDesc), _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int64,
                                      c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "snerf_mlp_fold_workspace_bytes": (c_int64, [POINTER(MlpDesc), c_int64, c_int]),
    "snerf_mlp_fwd_ws_f32": (c_int, [POINTER(MlpDesc), _P, _P, _P, c_int, _P, c_int64, c_int, _P, _P, c_int64, _P]),
    "snerf_warp_fold_workspace_bytes": (c_int64, [POINTER(WarpDesc), c_int64, c_int]),
    "snerf_warp_fwd_ws_f32": (c_int, [POINTER(WarpDesc), _P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P, c_int64, _P]),
    "snerf_dy_contract_scratch_floats": (c_int64, [c_int64, c_int, c_int]),
    "snerf_dy_contract_f32": (c_int, [_P, c_int64, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int64, c_int, c_int, _P, _P]),
    "snerf_mlp_stream_slots": (c_int, [POINTER(MlpDesc), _P, _P, c_int, _P]),
    "snerf_smpl_nerf_train_workspace_bytes": (c_int64, [POINTER(MlpDesc), POINTER(MlpDesc), POINTER(WarpDesc), c_int64, c_int, c_int,
                                                        c_int64]),
    "snerf_smpl_nerf_train_grads_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, POINTER(WarpDesc), _P, _P, c_int,
                                                POINTER(NerfBatch), _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "snerf_smpl_nerf_train_step_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, POINTER(WarpDesc), _P, _P, c_int,
                                               POINTER(NerfBatch), _P, c_int64, _P, _P, _P, _P, _P, _P, _P, POINTER(AdamState),
                                               POINTER(AdamRange), c_int, POINTER(AdamNet), c_int, c_int64, _P]),
    "snerf_warp_repack_f32": (c_int, [POINTER(WarpDesc), _P, c_int64, c_int64, _P, _P, _P]),
    "snerf_adam_step_f32": (c_int, [POINTER(AdamState), POINTER(AdamRange), c_int, POINTER(AdamNet), c_int, _P]),
    "snerf_nerf_train_workspace_bytes": (c_int64, [POINTER(MlpDesc), POINTER(MlpDesc), c_int64, c_int, c_int, c_int64]),
    "snerf_nerf_train_grads_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, c_int, POINTER(NerfBatch), c_int64,
                                           _P, _P, _P, _P, _P, _P, _P, _P]),
    "snerf_nerf_train_step_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, c_int, POINTER(NerfBatch), c_int64,
                                          _P, _P, _P, _P, _P, _P, POINTER(AdamState), POINTER(AdamRange), c_int, POINTER(AdamNet), c_int, _P, _P]),
    "snerf_nerf_train_grads_ig_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, c_int, POINTER(NerfBatch), c_int64,
                                              _P, _P, _P, _P, _P, _P, POINTER(InputGrads), _P, _P]),
    "snerf_nerf_train_step_ig_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, c_int, POINTER(NerfBatch), c_int64,
                                             _P, _P, _P, _P, _P, _P, POINTER(AdamState), POINTER(AdamRange), c_int, POINTER(AdamNet), c_int,
                                             POINTER(InputGrads), _P, _P]),
    # any --netwidth: nn.Linear as stand-alone GEMMs (csrc/linear.hip)
    "snerf_linear_fwd_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int64, c_int, _P, c_int, c_int, _P, c_int64, _P]),
    "snerf_linear_bwd_input_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int64, c_int, c_int, _P, c_int64, _P]),
    "snerf_linear_bwd_weight_scratch_floats": (c_int64, [c_int64, c_int, c_int]),
    "snerf_linear_bwd_weight_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int64, c_int, c_int, _P, c_int64, _P, _P, _P]),
    "snerf_relu_bwd_f32": (c_int, [_P, _P, c_int64, c_int, c_int64, c_int64, _P]),
    # 8(e): RCCL inside the boundary
    "snerf_comm_unique_id": (c_int, [_P]),
    "snerf_comm_init_rank": (c_int, [_P, c_int, c_int, POINTER(_P)]),
    "snerf_comm_destroy": (c_int, [_P]),
    "snerf_comm_info": (c_int, [_P, POINTER(ctypes.c_int32), POINTER(ctypes.c_int32)]),
    "snerf_comm_allreduce_avg_f32": (c_int, [_P, _P, c_int64, _P]),
    "snerf_nerf_train_step_dp_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, c_int, POINTER(NerfBatch), c_int64,
                                             _P, _P, _P, _P, _P, _P, POINTER(AdamState), POINTER(AdamRange), c_int, POINTER(AdamNet), c_int,
                                             _P, _P, _P]),
    "snerf_nerf_train_step_dp_ig_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, c_int, POINTER(NerfBatch), c_int64,
                                                _P, _P, _P, _P, _P, _P, POINTER(AdamState), POINTER(AdamRange), c_int, POINTER(AdamNet), c_int,
                                                POINTER(InputGrads), _P, _P, _P]),
    "snerf_smpl_nerf_train_step_dp_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, POINTER(WarpDesc), _P, _P, c_int,
                                                  POINTER(NerfBatch), _P, c_int64, _P, _P, _P, _P, _P, _P, _P, POINTER(AdamState),
                                                  POINTER(AdamRange), c_int, POINTER(AdamNet), c_int, c_int64, _P, _P]),
    "snerf_smpl_nerf_train_grads_aux_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, POINTER(WarpDesc), _P, _P, c_int,
                                                    POINTER(NerfBatch), _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "snerf_smpl_nerf_train_step_aux_f32": (c_int, [POINTER(MlpDesc), _P, _P, POINTER(MlpDesc), _P, _P, POINTER(WarpDesc), _P, _P, c_int,
                                                   POINTER(NerfBatch), _P, c_int64, _P, _P, _P, _P, _P, _P, _P, POINTER(AdamState),
                                                   POINTER(AdamRange), c_int, POINTER(AdamNet), c_int, c_int64, _P, _P, _P]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m smpl_nerf_amd.build` (hipcc, gfx950). "
            "smpl_nerf_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first so both share one HIP runtime)
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != SNERF_OK:
        msg = load().snerf_last_error_string()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


# ---- optional per-launch timing (HIP events on the launch stream) ---------------------------------
_PROFILE = None


class profile:
    """Context manager: records a pair of events (on the stream the kernels are launched on, i.e.
    PyTorch's current stream) around every C-ABI launch made through `timed()`.

        with _lib.profile() as prof: pipeline(data)
        prof.summary() -> {name: (calls, total_ms)}
    """

    def __enter__(self):
        global _PROFILE
        self.records = []
        self._prev = _PROFILE
        _PROFILE = self
        return self

    def __exit__(self, *exc):
        global _PROFILE
        _PROFILE = self._prev
        return False

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.records:
            calls, ms = out.get(name, (0, 0.0))
            out[name] = (calls + 1, ms + e0.elapsed_time(e1))
        return out


CALLS = 0      # C-ABI launches made through timed() since import (bench.py: calls per step of an unprofiled loop)


class timed:
    """with timed("mlp_fwd"): lib.snerf_...(...)  - counts the call; records an event pair when a profile() is active."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global CALLS
        CALLS += 1
        if _PROFILE is not None:
            import torch
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _PROFILE is not None:
            self.e1.record()
            _PROFILE.records.append((self.name, self.e0, self.e1))
        return False
