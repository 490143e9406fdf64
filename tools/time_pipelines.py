"""Dev/measurement: render throughput (ray-samples/s, one 128x128 frame, 64+128 samples) of the pose-conditioned
pipelines of SURVEY section 8 next to NerfPipeline, per precision mode.  Numbers quoted in DESIGN.md section 8."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from smpl_nerf_amd import synthetic as syn
from smpl_nerf_amd.nets import RenderRayNet, WarpFieldNet
from smpl_nerf_amd.ops import PositionalEncoder
from smpl_nerf_amd.pipelines import AppendSmplParamsPipeline, NerfPipeline, PipelineArgs, SmplNerfPipeline

dev = torch.device("cuda:0")


def net(p, **kw):
    m = RenderRayNet(8, 256, 60, 24, kw.get("add", 0), skips=[4])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return m.to(dev)


def timed(pipe, data, nets, label):
    for prec in ("fp32", "bf16x6", "f16x3", "bf16x3"):
        for m in nets:
            m.precision = prec
        with torch.no_grad():
            pipe(data)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                pipe(data)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{label:22s} {prec:7s} {dt * 1e3:7.2f} ms/frame  {16384 * 256 / dt:.3e} ray-samples/s")


enc = (PositionalEncoder(10, 0), PositionalEncoder(4, 0))
frame = [torch.from_numpy(a).to(dev) for a in syn.frame_batch(128, 128, seed=7)]
pc, pf = syn.make_scene_nets(101)
mc, mf = net(pc), net(pf)
timed(NerfPipeline(mc, mf, PipelineArgs(), *enc), frame, (mc, mf), "nerf")

mw = WarpFieldNet(8, 256, 60, 40)
mw.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_warp_field_params(103, out_scale=0.3).items()})
mw = mw.to(dev)
pose = torch.from_numpy(np.tile(syn.human_poses()[3][None], (16384, 1)).astype(np.float32)).to(dev)
smpl = SmplNerfPipeline(mc, mf, mw, PipelineArgs(), *enc, PositionalEncoder(10, 0))
timed(smpl, frame[:4] + [pose, frame[4]], (mc, mf, mw), "smpl_nerf (warp)")

pa = syn.make_scene_net_params(301, add_first=True, additional_input_dim=69)
ma, mb = net(pa, add=69), net(pa, add=69)
asp = AppendSmplParamsPipeline(ma, mb, PipelineArgs(human_pose_encoding=0), *enc, PositionalEncoder(10, 0))
timed(asp, frame[:4] + [pose, frame[4]], (ma, mb), "append_smpl_params")

# ---- training steps (4096 rays, fwd + bwd + Adam), bf16x6 forward/dgrad and all-fp32 ---------------------------
from smpl_nerf_amd.trainer import DataParallelTrainer


def timed_train(make, label):
    for prec in ("fp32", "bf16x6", "f16x3"):
        pipe, models, batch = make(prec)
        tr = DataParallelTrainer(pipe, models, lr=3e-5)  # cf. bench.py: larger steps kill the fine net (all gradients 0)
        for _ in range(2):
            tr.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tr.step(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"train {label:16s} {prec:7s} {dt * 1e3:7.2f} ms/step  {4096 * 256 / dt:.3e} ray-samples/s")


sub = torch.arange(0, 16384, 4, device=dev)


def make_nerf(prec):
    a, b = net(pc), net(pf)
    a.precision = b.precision = prec
    return NerfPipeline(a, b, PipelineArgs(), *enc), [a, b], [t[sub] for t in frame]


def make_smpl(prec):
    a, b = net(pc), net(pf)
    a.precision = b.precision = prec
    w = WarpFieldNet(8, 256, 60, 40)
    w.precision = prec
    w.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_warp_field_params(103, out_scale=0.3).items()})
    w = w.to(dev)
    p = SmplNerfPipeline(a, b, w, PipelineArgs(), *enc, PositionalEncoder(10, 0))
    return p, [a, b, w], [t[sub] for t in frame[:4]] + [pose[sub], frame[4][sub]]


timed_train(make_nerf, "nerf")
timed_train(make_smpl, "smpl_nerf (warp)")
