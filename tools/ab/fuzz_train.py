"""Randomised sweep of the one-call training step against the autograd form: random depth / width / skips / B / Nc / Nf / chunk /
white background / use_directional_input / additional inputs, first-step loss and gradients compared.  Not part of the suite
(minutes on the GPU); prints one line per case and a summary.

    python tools/ab/fuzz_train.py [cases] [seed]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder
from smpl_nerf_amd.pipelines import NerfPipeline, PipelineArgs
from smpl_nerf_amd.trainer import DataParallelTrainer

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    depth = int(rng.integers(1, 13)) if not os.environ.get("FUZZ_EDGES") else int(rng.choice([1, 2, 15, 16]))     # FUZZ_EDGES=1: the limits
    width = int(rng.choice([16, 33, 64, 100, 128, 200, 256, 256, 256]))
    if os.environ.get("FUZZ_EDGES"):
        width = int(rng.choice([2, 3, 15, 17, 63, 65, 127, 129, 255, 256]))
    prec = str(rng.choice(["fp32", "bf16x6", "f16x3"])) if width == 256 else "fp32"
    if os.environ.get("FUZZ_PREC") and width == 256:      # e.g. FUZZ_PREC=bf16x3
        prec = os.environ["FUZZ_PREC"]
    if os.environ.get("FUZZ_WIDE"):      # FUZZ_WIDE=1: --netwidth above 256 (the kernels of 320 / 384 / 448 / 512 features, fp32)
        width, prec = [257, 300, 320, 352, 384, 400, 448, 500, 512][case % 9], "fp32"
    kind = str(rng.choice(["nerf", "nerf", "smpl_nerf", "append_smpl_params", "append_smpl_params_encoded", "append_to_nerf"]))
    if os.environ.get("FUZZ_KIND"):
        kind = os.environ["FUZZ_KIND"]
    skips = sorted(set(int(v) for v in rng.integers(0, depth, rng.integers(0, 3)))) if depth > 1 else []
    B = int(rng.choice([1, 2, 5, 31, 64, 100, 129, 257, 600]))
    Nc = int(rng.choice([3, 4, 7, 16, 33, 64, 100]))   # the sampler needs 3 coarse samples (one interior weight)
    Nf = int(rng.choice([0, 1, 5, 64, 128, 150]))
    if os.environ.get("FUZZ_LARGE"):      # FUZZ_LARGE=1: the sizes of real steps (use with FUZZ_VS_TORCH=0)
        B, Nc, Nf = int(rng.choice([2048, 3001, 4096])), 64, 128
    chunk = int(rng.choice([0, 1, 17, 64, 200]))
    if os.environ.get("FUZZ_LARGE"):
        chunk = int(rng.choice([0, 1000, 2048]))
    wb = int(rng.integers(0, 2))
    use_dir = int(rng.integers(0, 4) != 0)
    Lp, Ld = int(rng.choice([10, 10, 6, 3])), int(rng.choice([4, 4, 2]))
    idp, idd = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    run_fine = 1 if Nf > 0 else 0
    o = rng.normal(0, 0.2, (B, 3)).astype(np.float32) + np.array([0, 0, 2.4], np.float32)
    d = rng.normal(0, 0.3, (B, 3)).astype(np.float32) + np.array([0, 0, -1], np.float32)
    z = np.sort(rng.uniform(1.0, 4.0, (B, Nc)).astype(np.float32), -1)
    samples = (o[:, None, :] + d[:, None, :] * z[:, :, None]).astype(np.float32)
    gt = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    batch = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (samples, o, d, z, gt)]
    add_dim = {"append_smpl_params": 69, "append_smpl_params_encoded": 69 * 20, "append_to_nerf": 2}.get(kind, 0)
    if kind != "nerf":
        use_dir = 1 if kind == "smpl_nerf" else use_dir
        from smpl_nerf_amd import synthetic as syn
        pose = torch.from_numpy(syn.human_poses()[np.arange(B) % 10].astype(np.float32)).to(dev)
        batch = batch[:4] + [pose, batch[4]]
    if os.environ.get("FUZZ_MISALIGN"):      # every input 4 bytes off a 16-byte boundary (a view one float into its storage)
        batch = [torch.cat([t.new_zeros(1), t.reshape(-1)])[1:].view(t.shape) for t in batch]
    desc = f"{kind} {prec} depth {depth} width {width} skips {skips} B {B} Nc {Nc} Nf {Nf} chunk {chunk} wb {wb} dir {use_dir} L {Lp}/{Ld} id {idp}/{idd}"
    try:
        runs = []
        # third run (split precisions): the same step in exact fp32 - the split modes against the arithmetic they stand in for
        for one_call, run_prec in ((None, prec), (False, prec)) + (((None, "fp32"),) if prec != "fp32" else ()):
            torch.manual_seed(1000 + case)
            pe, de = PositionalEncoder(Lp, idp), PositionalEncoder(Ld, idd)
            nets = []
            for _ in range(2):
                m = RenderRayNet(depth, width, 3 * pe.output_dim, 3 * de.output_dim, add_dim, skips=list(skips),
                                 use_directional_input=use_dir).to(dev).train()
                with torch.no_grad():
                    m.sigma_out_layer.weight.mul_(20.0)
                m.precision = run_prec
                nets.append(m)
            args = PipelineArgs(white_background=wb, number_fine_samples=max(Nf, 1), run_fine=run_fine)
            if kind == "smpl_nerf":
                from smpl_nerf_amd.nets import WarpFieldNet
                from smpl_nerf_amd.pipelines import SmplNerfPipeline
                wdepth, wwidth = int(2 + case % 7), int([256, 128, 100][case % 3])
                mw = WarpFieldNet(wdepth, wwidth, 3 * pe.output_dim, 2 * 20).to(dev).train()
                with torch.no_grad():
                    for q in mw.parameters():
                        q.mul_(0.3)
                mw.precision = run_prec if wwidth == 256 else "fp32"
                pipe = SmplNerfPipeline(nets[0], nets[1], mw, args, pe, de, PositionalEncoder(10, 0))
                nets = nets + [mw]
            elif add_dim:
                from smpl_nerf_amd.pipelines import AppendSmplParamsPipeline, AppendToNerfPipeline
                args.human_pose_encoding = 1 if kind.endswith("encoded") else 0
                cls = AppendToNerfPipeline if kind == "append_to_nerf" else AppendSmplParamsPipeline
                pipe = cls(nets[0], nets[1], args, pe, de, PositionalEncoder(10, 0))
            else:
                pipe = NerfPipeline(nets[0], nets[1], args, pe, de)
            tr = DataParallelTrainer(pipe, nets, lr=1e-3, one_call=one_call)
            tr.rays_per_chunk = chunk
            if len(runs) == 0 and run_fine:      # the HIP path's hierarchical samples at the initial weights (for the CPU comparison below)
                pipe.keep_fine = True      # (opt-in: forward() leaves its hierarchical samples in pipe.last_fine)
                with torch.no_grad():
                    pipe(batch)
                gpu_fine = tuple(t.detach().cpu() for t in pipe.last_fine)
                pipe.keep_fine, pipe.last_fine = False, None
            loss = float(tr.step(batch))
            # (split-precision smpl_nerf with identity columns / more frequencies: the trainer keeps the autograd path by design)
            by_design = kind == "smpl_nerf" and prec != "fp32" and (idp or Lp > 10)
            assert (tr._one_call_state() is not None) == (one_call is None) or by_design, "path"
            runs.append((loss, [None if p.grad is None else p.grad.clone() for p in tr.params]))
        (la, ga), (lb, gb) = runs[:2]
        e32 = -1.0
        if len(runs) == 3:       # |g - g_fp32| against the largest gradient tensor of the net
            top = max((float(r.norm()) for r in runs[2][1] if r is not None), default=0.0)
            # (the coarse net only: the fine net's gradient moves with the fine samples, which a last-bit difference of the coarse
            # weights can move - the sampler's threshold, see fuzz_render.py)
            n0 = len(list(nets[0].parameters()))
            e32 = max((float((a - r).norm()) / top for a, r in list(zip(ga, runs[2][1]))[:n0] if a is not None and r is not None),
                      default=0.0) if top > 0 else 0.0
            e32_fine = max((float((a - r).norm()) / top for a, r in list(zip(ga, runs[2][1]))[n0:] if a is not None and r is not None),
                           default=0.0) if top > 0 else 0.0
            if e32_fine > 5e-2:
                e32 = max(e32, e32_fine)
        err = 0.0
        for a, b in zip(ga, gb):
            if b is None or a is None:
                continue
            err = max(err, float((a - b).norm()) / (float(b.norm()) + 1e-12)) if float(b.norm()) > 1e-9 else err
        et, et_fine, lt = -1.0, 0.0, 0.0
        if kind == "nerf" and prec == "fp32" and os.environ.get("FUZZ_VS_TORCH", "1") != "0":
            # the same step on the CPU torch restatement of the reference (oracle/torch_cpu_path.py), from the same initial weights
            from oracle import torch_cpu_path as TP
            torch.manual_seed(1000 + case)
            P = []
            for _ in range(2):
                m = RenderRayNet(depth, width, 3 * pe.output_dim, 3 * de.output_dim, add_dim, skips=list(skips), use_directional_input=use_dir)
                with torch.no_grad():
                    m.sigma_out_layer.weight.mul_(20.0)
                P.append({k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()})
            cb = [t.cpu() for t in batch]
            # on the HIP path's fine samples: the reference's sampler is discontinuous where a bin's mass sits at its 1e-5
            # threshold (utils.py:224: nearly opaque rays), so a last-bit difference of the coarse weights would move samples - with
            # equal samples the fine net's gradient is held as tightly as the coarse net's
            with TP.fine_override(gpu_fine if run_fine else None):
                out = TP.nerf_pipeline_forward(P[0], P[1], TP.Args(white_background=wb, number_fine_samples=max(Nf, 1), run_fine=run_fine),
                                               TP.PositionalEncoder(Lp, idp), TP.PositionalEncoder(Ld, idd), cb,
                                               net_kw=dict(n_layers=depth, positions_dim=3 * pe.output_dim, directions_dim=3 * de.output_dim,
                                                           skips=tuple(skips), use_directional_input=use_dir))
            lt = torch.nn.functional.mse_loss(out[0], cb[-1]) + torch.nn.functional.mse_loss(out[1], cb[-1])
            lt.backward()
            ref = [v.grad for p_ in P for v in p_.values()]
            top = max(float(r.norm()) for r in ref if r is not None) if any(r is not None for r in ref) else 0.0
            # the coarse net's gradient does not depend on the fine samples (they are detached): tight.  The fine net's does, and
            # the reference's sampler is discontinuous where a bin's mass sits at its 1e-5 threshold (utils.py:224: nearly opaque
            # rays) - there a last-bit difference of the coarse weights moves samples: loose.
            et = et_fine = 0.0
            n_coarse = len(P[0])
            for i, (a, r) in enumerate(zip(ga, ref)):
                if r is None or a is None:
                    continue
                e = float((a.cpu() - r).norm()) / max(float(r.norm()), 1e-3 * top, 1e-12)
                if i < n_coarse:
                    et = max(et, e)
                else:
                    et_fine = max(et_fine, e)
            lt = float(lt.detach())
            if not (abs(la - lt) <= 2e-3 * abs(lt) + 1e-7 and et <= 1e-2 and et_fine <= 1e-2):
                err = max(err, 1.0)      # flag the case
        # (a sample within rounding of a ReLU kink takes the other side in the other arithmetic: ~1 / sqrt(samples) of a gradient)
        if e32 > max(2e-4 if prec != "bf16x3" else 2e-2, 0.05 / np.sqrt(B * Nc)):
            err = max(err, 1.0)      # flag the case
        ok = abs(la - lb) <= 2e-6 * abs(lb) + 1e-8 and err <= (2e-4 if prec == "fp32" else 2e-3) and np.isfinite(la)
        bad += not ok
        print(("ok  " if ok else "BAD ") + desc + f": loss {la:.6f} / {lb:.6f}, max rel grad err {err:.2e}" + (f", vs fp32 step {e32:.1e}" if e32 >= 0 else "") + (f", vs CPU torch: loss {lt:.6f}, grads coarse {et:.2e} fine {et_fine:.2e}" if et >= 0 else ""), flush=True)
    except Exception as e:   # noqa: BLE001 - the sweep reports and goes on
        bad += 1
        print("EXC " + desc + f": {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{cases - bad} of {cases} cases agree")
