#!/usr/bin/env python3
"""Longer-horizon training golden: the REFERENCE's own NerfSolver objects (its NerfPipeline, its Adam, its
nerf_loss; solver/nerf_solver.py:31-33, 48-52, 76-89) run for 200 steps at B = 256 on the synthetic scene - the loss curve,
the validation loss / PSNR of the trained nets on held-out rays (util/scores.py:47-48: PSNR = -10 log10(mse)) and digests
of the trained parameters.  Build container only (see make_golden.py); about four minutes of CPU.

    python tests/golden/make_golden_train.py        # writes g15_train200.npz

Batches: step i uses rays `perm_i[:256]` of the 128x128 frame syn.frame_batch(seed=7), perm_i = permutation drawn from
numpy's default_rng(777) (recorded in the fixture as the index table, so the test does not depend on numpy's generator).
Learning rate 3e-5: on this synthetic scene Adam at the parser's 5e-4 collapses the fine net's densities to zero within a
few steps (every gradient becomes exactly 0 - nothing left to compare); at 3e-5 both nets train."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch

import make_golden as MG
import make_golden_grad as GG
from smpl_nerf_amd import synthetic as syn

STEPS, B, LR = 200, 256, 3e-5
t = MG.t


def run_reference(U, RenderRayNet, NerfSolver, threads, idx, data, log):
    """The 200 steps on the reference's NerfSolver objects with `threads` intra-op threads -> (losses, mc, mf, solver)."""
    torch.set_num_threads(threads)
    pc, pf = syn.make_scene_nets(101)
    mc = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pc).train()
    mf = MG.load_params(RenderRayNet(8, 256, 60, 24, skips=[4]), pf).train()
    solver = NerfSolver(mc, mf, U.PositionalEncoder(10, False), U.PositionalEncoder(4, False),
                        MG.Args(lrate=LR, weight_decay=0.0), torch.optim.Adam, torch.nn.MSELoss())
    losses = []
    for i in range(STEPS):
        batch = [t(a[idx[i]]) for a in data]
        rgb, rgb_fine, _, _ = solver.pipeline(batch)                      # nerf_solver.py:81
        solver.optim.zero_grad()                                          # :83
        loss = solver.nerf_loss(rgb, rgb_fine, batch[-1])                 # :85
        loss.backward()                                                   # :86
        solver.optim.step()                                               # :87
        losses.append(loss.item())
        if log and i % 20 == 19:
            print(f"step {i + 1}: loss {losses[-1]:.6f}", flush=True)
    return losses, mc, mf, solver


def main():
    U, RenderRayNet, NerfPipeline, _, _ = MG._import_reference()
    from solver.nerf_solver import NerfSolver
    torch.set_grad_enabled(True)
    data = syn.frame_batch(128, 128, seed=7)
    rng = np.random.default_rng(777)
    idx = np.stack([rng.permutation(16384)[:B] for _ in range(STEPS)]).astype(np.int64)
    val_idx = np.arange(37, 16384, 16)[:1024].astype(np.int64)          # strided validation rays of the same frame
    losses, mc, mf, solver = run_reference(U, RenderRayNet, NerfSolver, os.cpu_count(), idx, data, True)
    # The reference against ITSELF: the same 200 steps with 3 intra-op threads instead of 8 - only MKL's GEMM blocking, i.e.
    # the fp32 summation order, changes.  200 optimiser steps amplify that round-off; how far the reference's own curve moves
    # is the yardstick for how far another fp32 implementation's may.
    losses3, _, _, _ = run_reference(U, RenderRayNet, NerfSolver, 3, idx, data, False)
    rel = np.abs(np.array(losses3) - np.array(losses)) / np.array(losses)
    print(f"reference, 3 threads vs {os.cpu_count()}: max rel dev {rel.max():.3e}, mean {rel.mean():.3e}")
    torch.set_num_threads(os.cpu_count())
    mc.eval(), mf.eval()
    with torch.no_grad():                                                 # the validation pass of :107-150 on the held-out rays
        vb = [t(a[val_idx]) for a in data]
        rgb, rgb_fine, _, _ = solver.pipeline(vb)
        val_loss = solver.nerf_loss(rgb, rgb_fine, vb[-1]).item()
        mse_fine = torch.mean((rgb_fine - vb[-1]) ** 2).item()
    g = {"idx": idx, "val_idx": val_idx, "losses": np.array(losses, np.float64), "val_loss": np.array([val_loss]),
         "val_psnr_fine": np.array([-10.0 * np.log10(mse_fine)]), "val_rgb_fine": rgb_fine.numpy(),
         "lr": np.array([LR]), "steps": np.array([STEPS]),
         "losses_3_threads": np.array(losses3, np.float64)}
    for k, v in GG.param_digest((f"coarse.{k}", p) for k, p in mc.named_parameters()).items():
        g[f"param/{k}"] = v
    for k, v in GG.param_digest((f"fine.{k}", p) for k, p in mf.named_parameters()).items():
        g[f"param/{k}"] = v
    print("val loss", val_loss, "val PSNR (fine)", g["val_psnr_fine"][0])
    MG.save("g15_train200.npz", **g)


if __name__ == "__main__":
    main()
