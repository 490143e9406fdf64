"""Randomised sweep of the inference path against the CPU torch restatement of the reference (oracle/torch_cpu_path.py): random
depth / width / skips / encoders / use_directional_input / ray and sample counts / white background / precision, NerfPipeline
under no_grad (the pipeline's kernels) and the single-call render.  rgb of the coarse pass must agree closely on every ray; the
fine pass on all but a handful (a sample index may flip where a cdf value sits on a u).  Not part of the suite.

    python tools/ab/fuzz_render.py [cases] [seed]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import functools

import numpy as np
import torch

from oracle import torch_cpu_path as TP
from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder
from smpl_nerf_amd import synthetic as syn
from smpl_nerf_amd.nets import WarpFieldNet
from smpl_nerf_amd.pipelines import (AppendSmplParamsPipeline, AppendToNerfPipeline, NerfPipeline, PipelineArgs, SmplNerfPipeline)

_tp_net = TP.render_ray_net

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    kind = str(rng.choice(["nerf", "nerf", "smpl_nerf", "append_smpl_params", "append_smpl_params_encoded", "append_to_nerf"]))
    if os.environ.get("FUZZ_KIND"):
        kind = os.environ["FUZZ_KIND"]
    depth = int(rng.integers(1, 13)) if not os.environ.get("FUZZ_EDGES") else int(rng.choice([1, 2, 15, 16]))     # FUZZ_EDGES=1: the limits
    width = int(rng.choice([16, 33, 64, 100, 128, 200, 256, 256, 256]))
    if os.environ.get("FUZZ_EDGES"):
        width = int(rng.choice([2, 3, 15, 17, 63, 65, 127, 129, 255, 256]))
    prec = str(rng.choice(["fp32", "fp32", "bf16x6", "f16x3"])) if width == 256 else "fp32"
    if os.environ.get("FUZZ_PREC") and width == 256:      # e.g. FUZZ_PREC=bf16x3
        prec = os.environ["FUZZ_PREC"]
    if os.environ.get("FUZZ_WIDE"):      # FUZZ_WIDE=1: --netwidth above 256 (the kernels of 320 / 384 / 448 / 512 features, fp32)
        width, prec = [257, 300, 320, 352, 384, 400, 448, 500, 512][case % 9], "fp32"
    skips = sorted(set(int(v) for v in rng.integers(0, depth, rng.integers(0, 3)))) if depth > 1 else []
    B = int(rng.choice([1, 2, 5, 31, 64, 100, 129, 257, 600, 2049]))
    Nc = int(rng.choice([3, 4, 7, 16, 33, 64, 100]))
    Nf = int(rng.choice([0, 1, 5, 64, 128, 150]))
    wb = int(rng.integers(0, 2))
    use_dir = int(rng.integers(0, 4) != 0)
    Lp, Ld = int(rng.choice([10, 10, 6, 3, 12])), int(rng.choice([4, 4, 2, 6]))
    idp, idd = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    run_fine = 1 if Nf > 0 else 0
    o = rng.normal(0, 0.2, (B, 3)).astype(np.float32) + np.array([0, 0, 2.4], np.float32)
    d = rng.normal(0, 0.3, (B, 3)).astype(np.float32) + np.array([0, 0, -1], np.float32)
    z = np.sort(rng.uniform(1.0, 4.0, (B, Nc)).astype(np.float32), -1)
    samples = (o[:, None, :] + d[:, None, :] * z[:, :, None]).astype(np.float32)
    gt = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    cpu = [torch.from_numpy(np.ascontiguousarray(a)) for a in (samples, o, d, z, gt)]
    add_dim = {"append_smpl_params": 69, "append_smpl_params_encoded": 69 * 20, "append_to_nerf": 2}.get(kind, 0)
    if kind != "nerf":
        use_dir = 1 if kind == "smpl_nerf" else use_dir
        cpu = cpu[:4] + [torch.from_numpy(syn.human_poses()[np.arange(B) % 10].astype(np.float32)), cpu[4]]
    batch = [t.to(dev) for t in cpu]
    if os.environ.get("FUZZ_MISALIGN"):      # every input 4 bytes off a 16-byte boundary (a view one float into its storage)
        batch = [torch.cat([t.new_zeros(1), t.reshape(-1)])[1:].view(t.shape) for t in batch]
    desc = f"{kind} {prec} depth {depth} width {width} skips {skips} B {B} Nc {Nc} Nf {Nf} wb {wb} dir {use_dir} L {Lp}/{Ld} id {idp}/{idd}"
    try:
        torch.manual_seed(2000 + case)
        pe, de = PositionalEncoder(Lp, idp), PositionalEncoder(Ld, idd)
        nets = []
        for _ in range(2):
            m = RenderRayNet(depth, width, 3 * pe.output_dim, 3 * de.output_dim, add_dim, skips=list(skips), use_directional_input=use_dir)
            with torch.no_grad():
                m.sigma_out_layer.weight.mul_(20.0)
            m.precision = prec
            nets.append(m)
        P = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in nets]
        nets = [m.to(dev).eval() for m in nets]
        args = PipelineArgs(white_background=wb, number_fine_samples=max(Nf, 1), run_fine=run_fine, strict_cumsum=1)
        net_kw = dict(n_layers=depth, positions_dim=3 * pe.output_dim, directions_dim=3 * de.output_dim, skips=tuple(skips),
                      use_directional_input=use_dir)
        targs = TP.Args(white_background=wb, number_fine_samples=max(Nf, 1), run_fine=run_fine)
        tpe, tde, the = TP.PositionalEncoder(Lp, idp), TP.PositionalEncoder(Ld, idd), TP.PositionalEncoder(10, 0)
        TP.render_ray_net = functools.partial(_tp_net, **net_kw)          # the pipelines below call it with their own defaults
        if kind == "nerf":
            pipe = NerfPipeline(nets[0], nets[1], args, pe, de)
            ref_fn = lambda: TP.nerf_pipeline_forward(P[0], P[1], targs, tpe, tde, cpu)
        elif kind == "smpl_nerf":
            mw = WarpFieldNet(int(2 + case % 7), int([256, 128, 100][case % 3]), 3 * pe.output_dim, 40)
            with torch.no_grad():
                for q in mw.parameters():
                    q.mul_(0.3)
            Pw = {k: v.detach().clone() for k, v in mw.state_dict().items()}
            mw.precision = prec if mw.width == 256 else "fp32"
            pipe = SmplNerfPipeline(nets[0], nets[1], mw.to(dev).eval(), args, pe, de, PositionalEncoder(10, 0))
            ref_fn = lambda: TP.smpl_nerf_pipeline_forward(P[0], P[1], Pw, targs, tpe, tde, the, cpu)
        else:
            args.human_pose_encoding = targs.human_pose_encoding = 1 if kind.endswith("encoded") else 0
            cls = AppendToNerfPipeline if kind == "append_to_nerf" else AppendSmplParamsPipeline
            pipe = cls(nets[0], nets[1], args, pe, de, PositionalEncoder(10, 0))
            ref_fn = lambda: TP.append_pose_pipeline_forward(P[0], P[1], targs, tpe, tde, the, cpu, two_joints=kind == "append_to_nerf")
        pipe.keep_fine = True           # (opt-in: forward() leaves its hierarchical samples in pipe.last_fine)
        with torch.no_grad():
            out = pipe(batch)
            ref_own = ref_fn()          # the CPU path on its own hierarchical samples
            # ... and on the HIP path's samples: the reference's sampler is discontinuous where a bin's mass sits at
            # its 1e-5 threshold (utils.py:224), so a last-bit difference of the coarse weights moves a ray's samples - with equal
            # samples the fine pass is held to the tolerance of the coarse one, no "rays off" allowance
            with TP.fine_override(tuple(t.detach().cpu() for t in pipe.last_fine) if run_fine else None):
                ref = ref_fn()
        # the single-call entries (snerf_render_rays_f32 / _smpl_f32 / _add_f32) against the five-launch forward: bit for bit
        one_call_ok = True
        mixed = kind == "smpl_nerf" and pipe.model_warp_field.precision != prec      # (the one-call entry wants one precision for all nets)
        if not (kind == "smpl_nerf" and Nf == 0) and not mixed:
            args.strict_cumsum = 0
            with torch.no_grad():      # (keep_fine is on: pipe(batch) is the five-launch forward)
                fwd, one = pipe(batch), pipe.render_rays(batch)
            one_call_ok = all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(fwd, one))
            args.strict_cumsum = 1
        tol = 2e-5 if prec == "fp32" else (3e-3 if prec == "bf16x3" else 3e-4)
        if kind == "smpl_nerf":
            tol *= 100          # the warp net's round-off passes through two 2^9 encoders before it reaches a colour
        ec = (out[0].cpu() - ref[0]).abs().max(-1).values
        ef = (out[1].cpu() - ref[1]).abs().max(-1).values
        # information only: rays whose fine colour differs visibly when the CPU path draws its OWN samples (the sampler itself is
        # held to the oracle bit for bit from equal weights by fuzz_ops.py; the bench line reports sampler_index_equal_frac)
        flips = int(((out[1].cpu() - ref_own[1]).abs().max(-1).values > 10 * tol).sum())
        ok = float(ec.max()) <= tol and float(ef.max()) <= 3 * tol and bool(torch.isfinite(out[1]).all()) and one_call_ok
        bad += not ok
        print(("ok  " if ok else "BAD ") + desc + f": coarse max {float(ec.max()):.2e}, fine on equal samples median {float(ef.median()):.2e} max {float(ef.max()):.2e} (own samples: {flips} rays off)" + ("" if one_call_ok else "  ONE-CALL RENDER DIFFERS"),
              flush=True)
        if not ok and os.environ.get("FUZZ_ADJUDICATE", "1") != "0":
            # the flagged case in float64 on the CPU, same samples: who is further from the exact value?
            try:
                with TP.fine_override(tuple(t.detach().cpu().double() for t in pipe.last_fine) if run_fine else None):
                    P64 = [{k: v.double() for k, v in p_.items()} for p_ in P]
                    cpu64 = [t.double() if t.is_floating_point() else t for t in cpu]
                    if kind == "nerf":
                        r64 = TP.nerf_pipeline_forward(P64[0], P64[1], targs, tpe, tde, cpu64)
                    elif kind == "smpl_nerf":
                        r64 = TP.smpl_nerf_pipeline_forward(P64[0], P64[1], {k: v.double() for k, v in Pw.items()}, targs, tpe, tde, the, cpu64)
                    else:
                        r64 = TP.append_pose_pipeline_forward(P64[0], P64[1], targs, tpe, tde, the, cpu64, two_joints=kind == "append_to_nerf")
                    print(f"     fp64 adjudication (max abs error of the fine colours): HIP {float((out[1].cpu().double() - r64[1]).abs().max()):.2e}, "
                          f"CPU fp32 restatement {float((ref[1].double() - r64[1]).abs().max()):.2e}; coarse: HIP {float((out[0].cpu().double() - r64[0]).abs().max()):.2e}, "
                          f"CPU fp32 {float((ref[0].double() - r64[0]).abs().max()):.2e}", flush=True)
            except Exception as e:      # noqa: BLE001
                print(f"     fp64 adjudication failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("EXC " + desc + f": {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{cases - bad} of {cases} cases agree")
