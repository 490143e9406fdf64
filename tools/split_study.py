"""CPU study (numpy, no GPU): accuracy of operand-split schemes for the RenderRayNet layers against float64.

    python tools/split_study.py

Emulates what the matrix cores would compute - operands split into low-precision parts, the kept cross products exact,
summed in float64 (the accumulation error of the fp32 accumulator is common to all schemes and to the exact-fp32
kernel, and left out) - for
  bf16x6 : 3 bf16 parts (RNE), 6 products          (csrc/mlp_bf16.hip, the bench default)
  bf16x3 : 2 bf16 parts (RNE), 3 products
  f16x3  : 2 fp16 parts, 3 products, activations scaled per sample by a power of two so that the largest input of the
           layer sits at 2^14 (fp16 overflows at 65504), weights scaled per layer the same way; activation parts
           rounded toward zero (v_cvt_pkrtz_f16_f32), weight parts RNE (packed off-line)
on the scene nets of the bench and on a 128x128 frame's coarse samples; prints max |raw - raw64| and the RGB error
after sigmoid.  DESIGN.md section 9 quotes the numbers.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smpl_nerf_amd import synthetic as syn  # noqa: E402


def bf16_rne(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def f16_rne(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def f16_rtz(x):
    x = np.asarray(x, np.float32)
    h = x.astype(np.float16)
    over = np.abs(h.astype(np.float32)) > np.abs(x)
    h = np.where(over, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float32)


def split(x, n, rnd):
    parts, r = [], np.asarray(x, np.float32)
    for _ in range(n):
        p = rnd(r)
        parts.append(p)
        r = (r - p).astype(np.float32)
    return parts


def pow2_scale(maxabs, target_exp=14):
    """2^k with maxabs * 2^k in [2^target_exp, 2^(target_exp+1))."""
    m = np.maximum(np.asarray(maxabs, np.float64), 1e-300)
    return 2.0 ** (target_exp - np.floor(np.log2(m)))


def linear(x, w, b, scheme):
    x32, w32 = np.asarray(x, np.float32), np.asarray(w, np.float32)
    if scheme == "f64":
        return x @ w.T + b
    if scheme == "fp32":
        return x32.astype(np.float64) @ w32.astype(np.float64).T + b
    if scheme in ("bf16x6", "bf16x3"):
        n = 3 if scheme == "bf16x6" else 2
        xp, wp = split(x32, n, bf16_rne), split(w32, n, bf16_rne)
        keep = [(i, j) for i in range(n) for j in range(n) if i + j < n]
        out = sum(xp[i].astype(np.float64) @ wp[j].astype(np.float64).T for i, j in keep)
        return out + b
    if scheme == "f16x3":
        sx = pow2_scale(np.abs(x32).max(axis=1, keepdims=True))          # per sample
        sw = pow2_scale(np.abs(w32).max())                               # per layer
        xp = split((x32.astype(np.float64) * sx).astype(np.float32), 2, f16_rtz)
        wp = split((w32.astype(np.float64) * sw).astype(np.float32), 2, f16_rne)
        assert np.isfinite(xp[0]).all() and np.isfinite(wp[0]).all()
        keep = [(0, 0), (0, 1), (1, 0)]
        out = sum(xp[i].astype(np.float64) @ wp[j].astype(np.float64).T for i, j in keep)
        return out / (sx * sw) + b
    raise ValueError(scheme)


def net(P, xin, dpe, scheme, n_layers=8, skips=(4,)):
    f32 = (lambda a: a) if scheme == "f64" else (lambda a: a.astype(np.float32).astype(np.float64))   # fp32 activations
    lin = lambda x, n: f32(linear(x, P[n + ".weight"], P[n + ".bias"], scheme))
    o = np.maximum(lin(xin, "positions_pose_input"), 0)
    for i in range(n_layers - 1):
        o = np.maximum(lin(np.concatenate([o, xin], -1) if i in skips else o, f"positional_net.{i}"), 0)
    o = lin(o, "additional_linear_layer")
    sigma = lin(o, "sigma_out_layer")
    o = lin(np.concatenate([o, dpe], -1), "directional_input")
    o = np.maximum(lin(o, "directional_net.0"), 0)
    return np.concatenate([lin(o, "rgb_out_layer"), sigma], -1)


def pe(x, L):
    x = np.asarray(x, np.float64)
    return np.concatenate([f(x * 2.0 ** k) for k in range(L) for f in (np.sin, np.cos)], -1)


def main():
    data = syn.frame_batch(128, 128, seed=7)
    pts = data[0][::16].reshape(-1, 3)                      # 1024 rays x 64 samples
    d = np.repeat(data[2][::16], 64, axis=0).astype(np.float64)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    xin, dpe = pe(pts, 10), pe(d, 4)
    sig = lambda r: 1 / (1 + np.exp(-r))
    for name, params in zip(("coarse", "fine"), syn.make_scene_nets(101)):
        P = {k: np.asarray(v, np.float64) for k, v in params.items()}
        ref = net(P, xin, dpe, "f64")
        print(f"{name} net, {xin.shape[0]} samples; |raw| max {np.abs(ref).max():.1f}")
        for scheme in ("fp32", "bf16x6", "f16x3", "bf16x3"):
            out = net(P, xin, dpe, scheme)
            print(f"  {scheme:7s} max|raw - raw64| {np.abs(out - ref).max():.2e}   max|sigmoid(rgb) - ref| "
                  f"{np.abs(sig(out[:, :3]) - sig(ref[:, :3])).max():.2e}   sigma rel {np.abs(out[:, 3] - ref[:, 3]).max() / np.abs(ref[:, 3]).max():.2e}")


if __name__ == "__main__":
    main()
