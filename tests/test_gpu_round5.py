"""Round-5 GPU parity: what VERDICT r04 / ADVICE r04 asked for - strided callers of sample_pdf under autograd, the library's
shutdown entry, the latency-class kernels of small calls (README.md:23 `--batchsize=64`, inference.py:231 800 rays) against
the throughput kernels and the reference's fixtures."""
import ctypes

import numpy as np
import pytest
import torch

import torch_ref as R
from oracle import nerf_oracle as O
from smpl_nerf_amd import _lib
from smpl_nerf_amd import synthetic as syn
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def test_sample_pdf_gradient_through_the_reference_callers_strided_view(dev):
    """utils.py:259 calls sample_pdf(z_vals_mid, weights[..., 1:-1], args): a view with storage offset 1 and row stride Nc.
    Under autograd the gradient must be that of the dense copy (the backward took the view's data pointer)."""
    from smpl_nerf_amd import ops
    g = load_golden("g16_sample_pdf_grad.npz")
    bins_np, w_np = g["bins"], g["weights"]                       # [B, Nb], [B, Nb - 1]
    B, Nb = bins_np.shape
    gout = T(g["gout"], dev)
    args = O.Args(number_fine_samples=128)

    def run(bins, w):
        out = ops.sample_pdf(bins, w, args)
        (out * gout).sum().backward()
        return out.detach()

    bd, wd = T(bins_np, dev).requires_grad_(True), T(w_np, dev).requires_grad_(True)
    out_dense = run(bd, wd)
    # the same numbers as interior columns of a wider tensor (weights[..., 1:-1]) and bins as every second column
    wide = torch.full((B, Nb + 1), 7.0, device=dev)
    wide[:, 1:-1] = T(w_np, dev)
    wide.requires_grad_(True)
    bins2 = torch.zeros((B, 2 * Nb), device=dev)
    bins2[:, ::2] = T(bins_np, dev)
    bins2.requires_grad_(True)
    wv, bv = wide[..., 1:-1], bins2[:, ::2]
    assert not wv.is_contiguous() and not bv.is_contiguous()
    out_view = run(bv, wv)
    torch.testing.assert_close(out_view, out_dense, rtol=0, atol=0)
    torch.testing.assert_close(wide.grad[:, 1:-1], wd.grad, rtol=0, atol=0)
    assert float(wide.grad[:, 0].abs().max()) == 0.0 and float(wide.grad[:, -1].abs().max()) == 0.0
    torch.testing.assert_close(bins2.grad[:, ::2], bd.grad, rtol=0, atol=0)
    assert float(bins2.grad[:, 1::2].abs().max()) == 0.0


def test_shutdown_destroys_the_events_and_the_next_step_recreates_them(dev):
    """snerf_shutdown (include/smplnerf.h "State"): after a small training step (which forks the coarse net's backward onto
    the auxiliary stream: two events per thread and device) the events are destroyed; the next step creates them again
    and gives the same loss trajectory as a run without the shutdown in between."""
    from test_gpu_round4 import _trainer, _batch
    lib = _lib.load()
    losses = []
    for with_shutdown in (False, True):
        tr, pipe, mc, mf = _trainer(dev, "fp32", one_call=True, lr=1e-4)
        batch = _batch(dev, 64)
        ls = []
        for i in range(3):
            ls.append(float(tr.step(batch)))
            if with_shutdown:
                torch.cuda.synchronize()
                assert lib.snerf_shutdown() == 0
        losses.append(ls)
    assert losses[0] == losses[1]
    assert lib.snerf_shutdown() == 0 and lib.snerf_shutdown() == 0
