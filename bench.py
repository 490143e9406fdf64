#!/usr/bin/env python3
"""bench.py - ray-samples/s of the NeRF ray-march path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path - NerfPipeline.forward (models/nerf_pipeline.py:14-67) - over one
batch of synthetic input: a whole 128x128 frame (16 384 rays) with 64 coarse + 128 fine samples per ray
through two 8-layer / 256-wide RenderRayNets (BASELINE configs[1]).  One ray-sample = one MLP
evaluation of one sample point: 64 (coarse) + 192 (fine) = 256 per ray, 4 194 304 per step.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Inputs are resident in HBM before the timed region.  Rays of independent frames shard across ranks
(each rank renders its own camera pose); rendering has no exchange step, so there is no data-path
collective - RCCL is used only for the barrier and the max-over-ranks of the elapsed time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FLOP_PER_EVAL = 2 * 607872           # RenderRayNet 8x256, pos 60, dir 24, skips=[4]
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0       # MI355X dense bf16 matrix peak (MI355X_MICROARCH.md)
# precision modes of the inference kernel: (kernel name for rocprof, products per fp32-accurate MAC)
MODES = {"fp32": ("snerf::mlp_fwd_kernel<256, 8, false, false>", 1),
         "f16x3": ("snerf::mlp_fwd_bf16_kernel<256, 8, 2, false, 1>", 3),
         "bf16x6": ("snerf::mlp_fwd_bf16_kernel<256, 8, 3, false, 0>", 6),
         "bf16x3": ("snerf::mlp_fwd_bf16_kernel<256, 8, 2, false, 0>", 3)}
# HBM bytes per average launch of the MLP kernel, per precision mode, from the rocprofv3 PMC passes committed under
# profiles/ (FETCH_SIZE as reported plus WRITE_SIZE); the kernels are MFMA-bound, this is informational.
TRAFFIC_PER_LAUNCH = {}
try:
    with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as _f:
        TRAFFIC_PER_LAUNCH = {k: v["avg_launch"]["hbm_bytes"] for k, v in json.load(_f)["kernels"].items()}
except Exception:
    pass


def build_pipeline(dev, precision="fp32", workload="nerf"):
    from smpl_nerf_amd import synthetic as syn
    from smpl_nerf_amd.nets import RenderRayNet
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import NerfPipeline, PipelineArgs, SmplNerfPipeline

    params = list(syn.make_scene_nets(101))
    nets = []
    for p in params:
        m = RenderRayNet(8, 256, 60, 24, skips=[4])
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        m.precision = precision
        nets.append(m.to(dev).eval())
    args = PipelineArgs(white_background=0, run_fine=1, number_fine_samples=128, sigma_noise_std=0.0)
    if workload == "smpl_nerf":   # BASELINE configs[2]: pose-conditioned warp field in front of both nets
        from smpl_nerf_amd.nets import WarpFieldNet
        pw = syn.make_warp_field_params(103, out_scale=0.3)
        mw = WarpFieldNet(8, 256, 60, 40)
        mw.load_state_dict({k: torch.from_numpy(v) for k, v in pw.items()})
        params.append(pw)
        pipe = SmplNerfPipeline(nets[0], nets[1], mw.to(dev).eval(), args, PositionalEncoder(10, 0), PositionalEncoder(4, 0),
                                PositionalEncoder(10, 0))
        return pipe.set_precision(precision), params
    pipe = NerfPipeline(nets[0], nets[1], args, PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    return pipe, params


def cpu_baseline(params, data_np, n_rays, u):
    """The numpy oracle (a port of the reference's CPU path) timed on the host cores on a bounded
    sample of the same workload: the first n_rays rays of the same frame, same weights."""
    from oracle import nerf_oracle as O
    args = O.Args(u=u)
    pe, de = O.PositionalEncoder(10, 0), O.PositionalEncoder(4, 0)
    sub = [a[:n_rays] for a in data_np]
    if len(params) == 3:   # smpl_nerf: data = [samples, o, d, z, goal_pose, rgb]
        fwd = lambda d: O.smpl_nerf_pipeline_forward(params[0], params[1], params[2], args, pe, de, O.PositionalEncoder(10, 0), d)
    else:
        fwd = lambda d: O.nerf_pipeline_forward(params[0], params[1], args, pe, de, d)
    fwd([a[:64] for a in data_np])   # warm-up
    t0 = time.perf_counter()
    out = fwd(sub)
    dt = time.perf_counter() - t0
    return n_rays * 256 / dt, dt, out


def train_section(pipe, data, rays, steps, world, rank, dev):
    """Secondary measurement (not `value`): data-parallel training steps - forward with saved
    activations, MSE coarse+fine, HIP backward, one flat RCCL all-reduce of the gradients, Adam - on
    `rays` rays per GPU drawn from this rank's frame (solver/nerf_solver.py:76-87)."""
    from smpl_nerf_amd import _lib
    from smpl_nerf_amd.dist import barrier, max_over_ranks
    from smpl_nerf_amd.trainer import DataParallelTrainer
    models = [pipe.model_coarse, pipe.model_fine] + ([pipe.model_warp_field] if hasattr(pipe, "model_warp_field") else [])
    for m in models:
        m.train()
        for p in m.parameters():
            p.requires_grad_(True)
    # lr: small enough that both nets stay alive on this synthetic scene (at 1e-4 and above Adam's first steps push the
    # fine net's densities below zero everywhere: the rendered colour and every gradient become exactly 0, and the
    # backward kernels would be timed on all-zero operands)
    tr = DataParallelTrainer(pipe, models, lr=3e-5)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    n_total = data[0].shape[0]
    batches = []
    for _ in range(4):
        idx = torch.randperm(n_total, generator=g)[:rays].to(dev)
        batches.append([t[idx].contiguous() for t in data])
    losses = []
    for i in range(2):
        losses.append(tr.step(batches[i % 4]))
    barrier(dev)
    torch.cuda.synchronize()
    with _lib.profile() as prof:
        t0 = time.perf_counter()
        for i in range(steps):
            losses.append(tr.step(batches[i % 4]))
        torch.cuda.synchronize()
        barrier(dev)
        dt = time.perf_counter() - t0
    kern = prof.summary()
    dt = max_over_ranks(dt, dev)
    with torch.no_grad():   # the trained nets still render something (not collapsed to zero density)
        fine_std = float(pipe(batches[0])[1].std())
    losses = [float(l) for l in losses]
    evals = world * steps * rays * 256
    bwd = {k: v for k, v in kern.items() if k.startswith("mlp_bwd")}
    fwd = {k: v for k, v in kern.items() if k.startswith("mlp_fwd_train")}
    flop_step = 3 * FLOP_PER_EVAL * rays * 256          # fwd + dgrad + wgrad
    mlp_ms = (sum(v[1] for v in bwd.values()) + sum(v[1] for v in fwd.values())) / steps
    return {"metric": "ray-samples/s, training step (fwd+bwd+all-reduce+Adam)", "value": evals / dt,
            "rays_per_step_per_gpu": rays, "ms_per_step": dt / steps * 1e3, "steps": steps,
            "loss_first": losses[0], "loss_last": losses[-1], "rgb_fine_std_last_step": fine_std,
            "kernels": {"f16x3": "f16x3 forward (activations saved in fp32), dgrad and wgrad (fp32 reduce)",
                        "bf16x6": "bf16x6 forward, dgrad and wide wgrad, fp32 narrow wgrad", "bf16x3": "bf16x3 forward, dgrad "
                        "and wide wgrad, fp32 narrow wgrad", "fp32": "fp32"}.get(getattr(pipe.model_coarse, "precision", ""), ""),
            "mlp_kernels_ms_per_step": mlp_ms, "mlp_tflops": flop_step / (mlp_ms * 1e-3) / 1e12,
            "kernels_ms_per_step": {k: v[1] / steps for k, v in sorted(kern.items())},
            "collective": "one all-reduce of 1 220 872 fp32 gradients per step" if world > 1 else "none (1 GPU)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays of the frame the CPU baseline renders (0 = skip)")
    ap.add_argument("--precision", choices=sorted(MODES), default="f16x3",
                    help="matrix-core arithmetic of the render kernel: f16x3 (two fp16 parts of power-of-two-scaled "
                         "operands, 3 products per MAC, fp32-class accuracy, default), bf16x6 (three bf16 parts, 6 products, "
                         "fp32-class accuracy), fp32 (v_mfma_f32_16x16x4_f32), bf16x3 (two bf16 parts, ~1e-5 relative)")
    ap.add_argument("--workload", choices=["nerf", "smpl_nerf"], default="nerf",
                    help="nerf = BASELINE configs[1] (the metric's configuration); smpl_nerf = configs[2] (warp field + per-sample "
                         "directions in front of the same nets), same frame size and sample counts")
    ap.add_argument("--train-rays", type=int, default=4096, help="rays per GPU per training step (0 = skip the train section)")
    ap.add_argument("--train-steps", type=int, default=10)
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(ndev, 1))   # one rank per GPU; the modulo only matters for
    torch.cuda.set_device(dev)                               # single-GPU dry runs of the multi-rank path
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SNERF_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from smpl_nerf_amd import _lib, synthetic as syn
    from smpl_nerf_amd.dist import barrier, max_over_ranks, shard_frames

    pipe, params = build_pipeline(dev, a.precision, a.workload)
    # each rank renders its own frame: rays of independent images shard across GPUs (weak scaling)
    frame_id = shard_frames(world, rank)
    data_np = syn.frame_batch(128, 128, phi=7.0 * frame_id, theta=25.0 * frame_id, seed=7 + frame_id)
    if a.workload == "smpl_nerf":   # one of the 10 arm poses of configs[2] for the whole frame (render.py:190-220)
        pose = np.tile(syn.human_poses()[3 + frame_id % 7][None], (data_np[0].shape[0], 1)).astype(np.float32)
        data_np = list(data_np[:4]) + [pose, data_np[4]]
    data = [torch.from_numpy(x).to(dev) for x in data_np]
    rays = data[0].shape[0]
    evals_per_step = rays * 256

    with torch.no_grad():
        for _ in range(a.warmup):
            out = pipe(data)
        barrier(dev)
        torch.cuda.synchronize()
        with _lib.profile() as prof:
            t0 = time.perf_counter()
            for _ in range(a.steps):
                out = pipe(data)
            torch.cuda.synchronize()
            barrier(dev)
            elapsed = time.perf_counter() - t0
        kern = prof.summary()
    elapsed = max_over_ranks(elapsed, dev)

    alt = {}
    if rank == 0 or world > 1:
        with torch.no_grad():
            for prec in sorted(MODES):
                if prec == a.precision:
                    continue
                pipe.set_precision(prec)
                pipe(data)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    o2 = pipe(data)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
                with _lib.profile() as p2:
                    pipe(data)
                k2 = {k: v for k, v in p2.summary().items() if k.startswith("mlp_fwd")}
                mlp_s = sum(v[1] for v in k2.values()) * 1e-3
                prod = MODES[prec][1]
                pk = PEAK_F32_MFMA_TFLOPS if prec == "fp32" else PEAK_BF16_MFMA_TFLOPS
                tf = FLOP_PER_EVAL * prod * evals_per_step / mlp_s / 1e12
                alt[prec] = {"ray_samples_per_s_per_gpu": evals_per_step / dt, "ms_per_step": dt * 1e3,
                             "rgb_fine_max_abs_diff_vs_" + a.precision: float((o2[1] - out[1]).abs().max()),
                             "mlp_kernel": MODES[prec][0], "mlp_kernel_ms_per_step": mlp_s * 1e3,
                             "mfma_tflops": tf, "mfma_peak": pk, "mfma_frac": tf / pk}
            pipe.set_precision(a.precision)

    train = None
    if a.train_rays > 0:
        try:
            train = train_section(pipe, data, a.train_rays, a.train_steps, world, rank, dev)
        except Exception as e:  # the render metric above stays valid
            train = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        value = world * a.steps * evals_per_step / elapsed
        # dominant kernel = the fused encode+MLP kernel; it is launched twice per step (coarse: 16384*64
        # samples, fine: 16384*192).  Roofline over ALL its launches in the timed region, so that the
        # average launch duration is the number rocprofv3 --stats reports for the kernel.
        mlp = {k: v for k, v in kern.items() if k.startswith("mlp_fwd")}
        calls = sum(v[0] for v in mlp.values())
        ms = sum(v[1] for v in mlp.values())
        avg_ms = ms / calls
        units_per_launch = a.steps * evals_per_step / calls
        kname, products = MODES[a.precision]
        alg_tflops = FLOP_PER_EVAL * units_per_launch / (avg_ms * 1e-3) / 1e12    # fp32 MACs of the network
        if a.precision == "fp32":
            peak = PEAK_F32_MFMA_TFLOPS
            achieved = alg_tflops
            flop_per_unit = FLOP_PER_EVAL
            roof_extra = {"peak_note": "fp32-input MFMA (v_mfma_f32_16x16x4_f32), exact fp32, 157.3 TFLOP/s"}
            dtype = "f32"
        else:
            # split operands: every fp32 MAC of the network is `products` exact 16-bit products on the bf16 / fp16 matrix
            # cores (same dense peak), accumulated in fp32 - that is the algorithm of this kernel, so its flop count per
            # ray-sample is products x 1 215 744 and its roofline is the dense 16-bit MFMA peak.  The fp32-equivalent rate
            # (`fp32_equivalent_tflops`, what the network needs) is reported beside it.
            peak = PEAK_BF16_MFMA_TFLOPS
            achieved = alg_tflops * products
            flop_per_unit = FLOP_PER_EVAL * products
            roof_extra = {"fp32_equivalent_tflops": alg_tflops,
                          "fp32_equivalent_vs_fp32_mfma_peak": alg_tflops / PEAK_F32_MFMA_TFLOPS,
                          "fp32_equivalent_vs_bf16_mfma_peak": alg_tflops / PEAK_BF16_MFMA_TFLOPS,
                          "algorithmic_flop_per_unit": FLOP_PER_EVAL,
                          "products_per_fp32_mac": products,
                          "peak_note": f"dense {'fp16' if a.precision == 'f16x3' else 'bf16'} MFMA peak 2500 TFLOP/s "
                                       f"(v_mfma_f32_16x16x32_{'f16' if a.precision == 'f16x3' else 'bf16'}); operands split into "
                                       f"{'two fp16' if a.precision == 'f16x3' else 'bf16'} parts, {products} products per fp32 MAC, fp32 accumulate; padded tiles "
                                       f"(84->96, 280->288 inputs, 4-wide heads) are not counted"}
            dtype = {"f16x3": "f32 via f16x3 (operands scaled by exact powers of two and split into two fp16 parts, three "
                              "products per MAC, fp32 accumulate; RGB parity class of the fp32 kernel)",
                     "bf16x6": "f32 via bf16x6 (split-bf16 operands, fp32 accumulate; RGB parity class of the fp32 kernel)",
                     "bf16x3": "f32 via bf16x3 (split-bf16, ~2^-16 relative)"}[a.precision]
        line = {
            "metric": "ray-samples/sec (coarse+fine) at 128^2 / 64+128 samples",
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": ("nerf 128x128 frame per GPU, coarse+fine 64+128 samples/ray, run_fine=1, netdepth 8, "
                                    "width 256, skips [4], forward render (BASELINE configs[1])") if a.workload == "nerf" else
                                   ("smpl_nerf 128x128 frame per GPU (one arm pose), coarse+fine 64+128 samples/ray, warp field + "
                                    "netdepth 8 / width 256 nets, forward render (BASELINE configs[2]); roofline counts the "
                                    "RenderRayNet kernel only"),
                       "rays_per_step_per_gpu": rays, "ray_samples_per_ray": 256, "parallelism": f"dp{world} (rays of "
                       "independent frames per rank, no data-path collective)"},
            "rays_per_s": world * a.steps * rays / elapsed,
            "roofline": dict({"bound": "mfma", "kernel": kname + " (coarse + fine launches)",
                              "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                              "frac": achieved / peak, "traffic": TRAFFIC_PER_LAUNCH.get(a.precision),
                              "avg_launch_ms": avg_ms, "launches": calls, "flop_per_unit": flop_per_unit,
                              "units_per_launch": units_per_launch,
                              "traffic_note": "(FETCH_SIZE + WRITE_SIZE) per average launch from the PMC passes under "
                                              "profiles/, not measured live"}, **roof_extra),
            "precision": a.precision,
            "kernels_ms_per_step": {k: v[1] / a.steps for k, v in sorted(kern.items())},
        }
        line["other_precisions_1gpu"] = alt
        if train is not None:
            line["train"] = train
        if world == 1 and a.cpu_rays > 0:
            from smpl_nerf_amd.ops import uniform_u
            u = uniform_u(128, dev).cpu().numpy()
            cpu_val, cpu_dt, ref = cpu_baseline(params, data_np, min(a.cpu_rays, rays), u)
            n = min(a.cpu_rays, rays)
            err = float(np.max(np.abs(out[1][:n].cpu().numpy() - ref[1])))
            line["cpu_baseline"] = {"value": cpu_val, "unit": "ray-samples/s", "cores": os.cpu_count(),
                                    "kind": "port", "sample": f"first {n} rays of the same frame, same weights "
                                    f"({n * 256} ray-samples, {cpu_dt:.1f} s): numpy oracle, BLAS threads = host cores"}
            line["rgb_fine_max_abs_diff_vs_oracle"] = err
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
