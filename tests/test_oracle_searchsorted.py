"""Pin the C restatement of the reference's native binary search (oracle/searchsorted_ref.c)
against numpy.searchsorted with the reference's own test grid
(torchsearchsorted/test/test_searchsorted.py:27-44).  CPU only."""
import ctypes
import os
import subprocess
from itertools import product

import numpy as np
import pytest

from oracle import nerf_oracle as O
from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def cref():
    so = os.path.join(ROOT, "oracle", "_build", "libsearchsorted_ref.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
    lib.searchsorted_ref_f32.argtypes = [fp, ctypes.c_int64, ctypes.c_int64, fp, ctypes.c_int64, ctypes.c_int64,
                                         ip, ctypes.c_int]
    lib.searchsorted_ref_f32.restype = ctypes.c_int

    def run(a, v, side):
        a = np.ascontiguousarray(a, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty((max(a.shape[0], v.shape[0]), v.shape[1]), np.int64)
        rc = lib.searchsorted_ref_f32(a.ctypes.data_as(fp), a.shape[0], a.shape[1], v.ctypes.data_as(fp),
                                      v.shape[0], v.shape[1], out.ctypes.data_as(ip), 1 if side == "left" else 0)
        assert rc == 0
        return out
    return run


@pytest.mark.parametrize("Ba,Bv,A,V,side", list(product([1, 100, 200], [1, 100, 200], [1, 50, 500], [1, 12, 120],
                                                        ["left", "right"])))
def test_reference_grid(cref, Ba, Bv, A, V, side):
    if Ba > 1 and Bv > 1 and Ba != Bv:
        pytest.skip("mismatched batch sizes are skipped by the reference too (test_searchsorted.py:36-37)")
    rng = np.random.default_rng(Ba * 7 + Bv * 3 + A + V)
    for _ in range(3):
        a = np.sort(rng.random((Ba, A), dtype=np.float32), axis=1)
        v = rng.random((Bv, V), dtype=np.float32)
        np.testing.assert_array_equal(cref(a, v, side), O.searchsorted(a, v, side))


@pytest.mark.parametrize("side", ["left", "right"])
def test_ties_and_out_of_range(cref, side):
    g = load_golden("g_searchsorted.npz")
    np.testing.assert_array_equal(cref(g["a"], g["v"], side), g[f"out_{side}"])
    np.testing.assert_array_equal(cref(g["a2"], g["v2"], side), g[f"out2_{side}"])


def test_nerf_shape_cdf(cref):
    g = load_golden("g4_sampler.npz")
    np.testing.assert_array_equal(cref(g["cdf"], g["u"], "right"), g["inds"])
    np.testing.assert_array_equal(cref(g["cdf"], g["u"][:1], "right"), g["inds"])      # row-broadcast v
