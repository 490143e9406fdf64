"""CPU check of the arithmetic behind the split kernels (tools/split_study.py emulates them in numpy with exact
float64 products): bf16x6 and f16x3 stay in the fp32 class, bf16x3 is the 2^-16 mode."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import split_study as S  # noqa: E402
from smpl_nerf_amd import synthetic as syn  # noqa: E402


def test_split_schemes_against_float64():
    data = syn.frame_batch(128, 128, seed=7)
    pts = data[0][::256, ::4].reshape(-1, 3)                       # 64 rays x 16 samples
    d = np.repeat(data[2][::256], 16, axis=0).astype(np.float64)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    xin, dpe = S.pe(pts, 10), S.pe(d, 4)
    P = {k: np.asarray(v, np.float64) for k, v in syn.make_scene_nets(101)[1].items()}
    ref = S.net(P, xin, dpe, "f64")
    err = {k: float(np.abs(S.net(P, xin, dpe, k) - ref).max()) for k in ("fp32", "bf16x6", "f16x3", "bf16x3")}
    scale = float(np.abs(ref).max())
    assert err["fp32"] <= 2e-6 * scale
    assert err["bf16x6"] <= 2.0 * err["fp32"] + 1e-7 * scale
    assert err["f16x3"] <= 10.0 * err["fp32"] + 1e-7 * scale
    assert err["bf16x3"] >= 10.0 * err["f16x3"]                   # the two-bf16-part mode is a different class


def test_fp16_rtz_emulation():
    x = np.array([1.0, 1.0009765625, 1.0004, -1.0004, 65503.9, 6.1e-5, 3.1e-8, 0.0], np.float32)
    h = S.f16_rtz(x)
    assert np.all(np.abs(h) <= np.abs(x)) and np.all(np.sign(h) * np.sign(x) >= 0)
    assert h[2] == 1.0 and h[3] == -1.0 and h[4] == 65472.0
    assert np.all(np.abs(x - h) <= np.maximum(np.abs(x) * 2.0 ** -10, 6e-8))
