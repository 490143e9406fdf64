#!/usr/bin/env python3
"""bench.py - ray-samples/s of the NeRF ray-march path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path - NerfPipeline.forward (models/nerf_pipeline.py:14-67) - over one
batch of synthetic input: a whole 128x128 frame (16 384 rays) with 64 coarse + 128 fine samples per ray
through two 8-layer / 256-wide RenderRayNets (BASELINE configs[1]).  One ray-sample = one MLP
evaluation of one sample point: 64 (coarse) + 192 (fine) = 256 per ray, 4 194 304 per step.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`); when a
launcher already set RANK / WORLD_SIZE it joins that group.  One rank per GPU over RCCL ("nccl"); when fewer
GPUs than ranks are visible (dry runs of the multi-rank path on a 1-GPU box) the ranks share devices and the
group runs on gloo.

The headline `value` is measured in the reference's own arithmetic: exact fp32 (`--precision fp32`,
v_mfma_f32_16x16x4_f32).  The operand-split modes on the 16-bit matrix cores are timed in the same run and
reported under `other_precisions_1gpu`, each with its algorithmic roofline fraction.

Inputs are resident in HBM before the timed region.  Rays of independent frames shard across ranks
(each rank renders its own camera pose); rendering has no exchange step, so there is no data-path
collective - the process group is used only for the barrier and the max-over-ranks of the elapsed time
(and, in the `train` section, for the one gradient all-reduce per step).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_EVAL = 2 * 607872           # RenderRayNet 8x256, pos 60, dir 24, skips=[4] (SURVEY.md 8d / BASELINE.md 3)
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_16BIT_MFMA_TFLOPS = 2500.0      # MI355X dense bf16 / fp16 matrix peak (MI355X_MICROARCH.md)
# precision modes of the render kernel: (kernel name as rocprofv3 prints it, 16-bit products per fp32 MAC, dtype text)
MODES = {
    "fp32": ("snerf::mlp_fwd_kernel<256, 8, false, false>", 1, "f32"),
    "bf16x6": ("snerf::mlp_fwd_bf16_kernel<256, 8, 3, false, 0>", 6,
               "f32 via bf16x6 (operands split into three bf16 parts, six products per MAC, fp32 accumulate)"),
    "f16x3": ("snerf::mlp_fwd_bf16_kernel<256, 8, 2, false, 1>", 3,
              "f32 via f16x3 (power-of-two-scaled operands split into two fp16 parts: 22 significand bits, three products "
              "per MAC, fp32 accumulate) - narrower than fp32"),
    "bf16x3": ("snerf::mlp_fwd_bf16_kernel<256, 8, 2, false, 0>", 3,
               "f32 via bf16x3 (two bf16 parts: 16 significand bits) - narrower than fp32"),
}
COARSE_ONLY = ("nerf {r}x{r} frame per GPU, coarse-only 64 samples/ray, run_fine=0, netdepth 8, width 256, skips [4], forward "
               "render (BASELINE configs[0], the reference's CPU-runnable case)")
INPUT_DGRAD_FLOP_PER_EVAL = 2 * (60 * 256 * 2 + 24 * 128)   # encoder columns of layer 0 + the skip layer, and of directional_input
WARP_FLOP_PER_EVAL = 2 * (256 * 100 + 3 * 256)      # WarpFieldNet 100 -> 256 -> 3: 52 736 FLOP (SURVEY.md 8d / BASELINE.md 3)
WARP_FLOP_PADDED = 2 * (256 * 112 + 16 * 256)       # as executed: 7 k-blocks of 16 input slots, the 3-wide head in a 16-wide tile
WARP_FLOP_FOLDED = 2 * (256 * 64 + 16 * 256)        # fp32 inference: the 3 pose k-blocks are one 256-vector per ray (csrc/warp.hip)
WORKLOADS = {
    "nerf": "nerf {r}x{r} frame per GPU, coarse+fine 64+128 samples/ray, run_fine=1, netdepth 8, width 256, skips [4], "
            "forward render (BASELINE configs[1])",
    "smpl_nerf": "smpl_nerf {r}x{r} frame per GPU (one arm pose), coarse+fine 64+128 samples/ray, warp field + netdepth 8 / "
                 "width 256 nets, forward render (BASELINE configs[2] at 128, configs[3] at 256); roofline counts the "
                 "RenderRayNet kernel only",
    "append_vertices": "append_vertices {r}x{r} frame per GPU, SMPL-vertex-conditioned nets (per-ray constant inputs, quirk "
                       "Q7), coarse+fine 64+128 samples/ray (BASELINE configs[4]); vertices from synthetic_smpl.LinearBodyModel "
                       "(no smplx / SMPL model file in the image: parity unpinned for real SMPL vertices and for the fine "
                       "branch, which the reference itself cannot run)",
    "append_smpl_params": "append_smpl_params {r}x{r} frame per GPU, 69 pose columns in front of the encoding, coarse+fine "
                          "64+128 samples/ray (SURVEY 8f-4, the paper's headline model)",
}


# ---------------------------------------------------------------------------------------------------- launch
def _flush_c_stdio():
    """RCCL prints its version banner to the C library's stdout at the first collective; redirected to a pipe that stream is
    block-buffered and would surface at process exit - AFTER the JSON line.  Flush it where it was written."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL between processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------- workload
def build_pipeline(dev, precision="fp32", workload="nerf", run_fine=1):
    """(pipeline, parameter dicts, trainable models) of `workload` with seeded synthetic-scene weights."""
    import torch
    from smpl_nerf_amd import synthetic as syn
    from smpl_nerf_amd.nets import AppendVerticesNet, RenderRayNet, WarpFieldNet
    from smpl_nerf_amd.ops import PositionalEncoder
    from smpl_nerf_amd.pipelines import (AppendSmplParamsPipeline, AppendVerticesPipeline, NerfPipeline, PipelineArgs,
                                         SmplNerfPipeline)

    def load(m, p):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        return m.to(dev).eval()

    args = PipelineArgs(white_background=0, run_fine=run_fine, number_fine_samples=128, sigma_noise_std=0.0,
                        human_pose_encoding=1)
    enc = (PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    if workload == "append_vertices":
        from smpl_nerf_amd.synthetic_smpl import IndexPoseEstimator, LinearBodyModel
        params = [syn.make_append_vertices_params(s) for s in (201, 202)]
        nets = [load(AppendVerticesNet(8, 256, 60, 24, 6890, additional_input_layers=1, skips=[4]), p) for p in params]
        est = IndexPoseEstimator(torch.from_numpy(syn.human_poses((41, 38), 0, 60, 60)), torch.zeros(1, 10)).to(dev)
        pipe = AppendVerticesPipeline(nets[0], nets[1], est, LinearBodyModel(seed=3).to(dev), args, *enc)
        return pipe.set_precision(precision), params, nets
    if workload == "append_smpl_params":
        args.human_pose_encoding = 0
        params = [syn.make_scene_net_params(s, add_first=True, additional_input_dim=69) for s in (301, 303)]
        nets = [load(RenderRayNet(8, 256, 60, 24, 69, skips=[4]), p) for p in params]
        pipe = AppendSmplParamsPipeline(nets[0], nets[1], args, *enc, PositionalEncoder(10, 0))
        return pipe.set_precision(precision), params, nets
    params = list(syn.make_scene_nets(101))
    nets = [load(RenderRayNet(8, 256, 60, 24, skips=[4]), p) for p in params]
    if workload == "smpl_nerf":   # pose-conditioned warp field in front of both nets
        pw = syn.make_warp_field_params(103, out_scale=0.3)
        mw = load(WarpFieldNet(8, 256, 60, 40), pw)
        params.append(pw)
        pipe = SmplNerfPipeline(nets[0], nets[1], mw, args, *enc, PositionalEncoder(10, 0))
        return pipe.set_precision(precision), params, nets + [mw]
    pipe = NerfPipeline(nets[0], nets[1], args, *enc)
    return pipe.set_precision(precision), params, nets


def frame_inputs(workload, res, frame_id):
    """Numpy pipeline input list of the frame rank `frame_id` renders (camera on the sphere, per-ray jitter)."""
    import numpy as np
    from smpl_nerf_amd import synthetic as syn
    data = syn.frame_batch(res, res, phi=7.0 * frame_id, theta=25.0 * frame_id, seed=7 + frame_id)
    n = data[0].shape[0]
    if workload in ("smpl_nerf", "append_smpl_params"):   # one of the arm poses (render.py:190-220) for the whole frame
        pose = np.tile(syn.human_poses()[(3 + frame_id) % 10][None], (n, 1)).astype(np.float32)
        data = list(data[:4]) + [pose, data[4]]
    elif workload == "append_vertices":                    # image index -> estimator -> body model (dummy_dynamic_dataset.py:93)
        data = list(data[:4]) + [np.full((n,), (7 + frame_id) % 60, np.int64), data[4]]
    return data


# ---------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _calibration(section=None):
    try:
        with open(os.path.join(ROOT, "oracle", "cpu_baseline_calibration.json")) as f:
            c = json.load(f)
        if section:
            c = dict(c[section], host_cpu=c["host_cpu"], threads=c["threads"])
            keys = ("host_cpu", "threads", "rays", "reference_ray_samples_per_s", "port_ray_samples_per_s",
                    "port_over_reference_speed", "losses_bit_identical", "steps_compared")
        else:
            keys = ("host_cpu", "threads", "reference_ray_samples_per_s", "port_ray_samples_per_s",
                    "port_over_reference_speed", "outputs_bit_identical")
        return {k: c[k] for k in keys if k in c}
    except Exception:
        return None


def cpu_baseline(workload, params, data_np, n_rays, run_fine=1):
    """The reference's CPU path, as restated op for op in PyTorch-CPU fp32 by oracle/torch_cpu_path.py (calibrated against
    the reference itself in the build container: oracle/cpu_baseline_calibration.json), timed on this host on a bounded
    sample of the same workload: the first n_rays rays of the same frame, same weights; warm-up 1, median of 5.
    torch's default intra-op thread count (what the reference would get) oversubscribes a 128-core host on these small
    GEMMs, so a short probe (256 rays per thread count) picks the fastest setting first and `cores` reports it."""
    import numpy as np
    import torch
    from oracle import torch_cpu_path as T
    default_threads = torch.get_num_threads()
    P = [T.tparams(p) for p in params]
    pe, de = T.PositionalEncoder(10, False), T.PositionalEncoder(4, False)
    if workload == "append_vertices":
        run_fine = 0          # the reference's AppendVerticesPipeline raises in its fine branch (:71): its coarse pass is the CPU figure
    targs = T.Args(run_fine=run_fine)
    per_ray = 256 if run_fine else 64

    def make_fwd(n):
        data = [torch.from_numpy(np.ascontiguousarray(a[:n])) for a in data_np]
        if workload == "smpl_nerf":
            return lambda: T.smpl_nerf_pipeline_forward(P[0], P[1], P[2], targs, pe, de, T.PositionalEncoder(10, False), data)
        if workload == "append_smpl_params":
            targs.human_pose_encoding = 0
            return lambda: T.append_pose_pipeline_forward(P[0], P[1], targs, pe, de, T.PositionalEncoder(10, False), data)
        if workload == "append_vertices":   # estimator -> body model -> [vertices | PE(x) | PE(d)] rows, like the reference's forward
            from smpl_nerf_amd import synthetic as syn
            from smpl_nerf_amd.synthetic_smpl import IndexPoseEstimator, LinearBodyModel
            est = IndexPoseEstimator(torch.from_numpy(syn.human_poses((41, 38), 0, 60, 60)), torch.zeros(1, 10))
            body = LinearBodyModel(seed=3)

            def fwd():
                goal_poses, betas = est(data[4])
                verts = body(betas=betas, return_verts=True, body_pose=goal_poses,
                             global_orient=torch.zeros(1, 3).expand(len(data[0]), -1)).vertices
                return T.append_vertices_pipeline_forward_coarse(P[0], verts, targs, pe, de, data)
            return fwd
        return lambda: T.nerf_pipeline_forward(P[0], P[1], targs, pe, de, data)

    probe, small = {}, make_fwd(min(256, n_rays))
    with torch.no_grad():
        for t in sorted({default_threads, 64, 32, 16, 8} & set(range(1, (os.cpu_count() or 1) + 1)) | {default_threads}):
            torch.set_num_threads(t)
            small()
            t0 = time.perf_counter()
            small()
            probe[t] = min(256, n_rays) * per_ray / (time.perf_counter() - t0)
        threads = max(probe, key=probe.get)
        torch.set_num_threads(threads)
        fwd = make_fwd(n_rays)
        fwd()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = fwd()
            ts.append(time.perf_counter() - t0)
        torch.set_num_threads(default_threads)
    dt = statistics.median(ts)
    what = {"nerf": "NerfPipeline.forward", "smpl_nerf": "SmplNerfPipeline.forward",
            "append_smpl_params": "AppendSmplParamsPipeline.forward (the calibration record is the nerf pipeline's: this "
                                  "pipeline adds 69 input columns to the same ops)",
            "append_vertices": "AppendVerticesPipeline.forward up to its coarse result - the fine branch raises in the reference "
                               "(append_vertices_pipeline.py:71) - with the [samples, 20754] rows materialised and the dead "
                               "vertices_net evaluated like the reference does; estimator and body model are the synthetic "
                               "stand-ins; speed not calibrated against the reference (it cannot run here: smplx / SMPL file absent)"
            }.get(workload, workload)
    info = {"value": n_rays * per_ray / dt, "unit": "ray-samples/s", "cores": threads, "kind": "port",
            "host_cpu": _cpu_model(), "host_logical_cpus": os.cpu_count(), "torch_default_threads": default_threads,
            "thread_probe_ray_samples_per_s": {str(k): v for k, v in sorted(probe.items())},
            "sample": f"first {n_rays} rays of the same frame, same weights ({n_rays * per_ray} ray-samples per pass; warm-up 1, "
                      f"median of 5 passes, {dt:.2f} s each): oracle/torch_cpu_path.py = the reference's {what} "
                      f"restated op for op on PyTorch-CPU fp32, torch.set_num_threads({threads}) = the fastest of the probed "
                      f"counts",
            "calibration_vs_reference_in_build_container": _calibration()}
    inds = T.LAST.get("inds") if (run_fine and workload == "nerf") else None
    info["_sampler_inds"] = None if inds is None else inds.numpy()
    return info, [o.numpy() for o in out]


def quality_keys(out, ref, gt, n, cpu_inds, pipe, data):
    """The quality half of BASELINE's metric ("...; PSNR vs ref"), on the CPU-baseline subset and outside the timed region:
    PSNR (util/scores.py:47-48: -10 ln(mse) / ln 10 over all pixels and channels) of the HIP render and of the CPU
    reference path's render against the same ground truth, their difference (north_star: within 0.01 dB), the PSNR between
    the two renders, and the fraction of (ray, u) pairs for which the HIP sampler picked the same searchsorted index as the
    CPU path (end to end from each side's own coarse weights, default - non-strict - normalising sum)."""
    import numpy as np
    import torch
    from smpl_nerf_amd import ops
    from smpl_nerf_amd.io import img2psnr
    q = {}
    k = 1 if len(ref) > 1 else 0
    hip, cpu = out[k][:n].float().cpu().numpy(), ref[k]
    q["psnr_db"] = img2psnr(hip, gt[:n])
    q["psnr_db_cpu_reference_path"] = img2psnr(cpu, gt[:n])
    q["psnr_delta_db_vs_oracle"] = q["psnr_db"] - q["psnr_db_cpu_reference_path"]
    q["psnr_db_hip_vs_cpu_reference_render"] = img2psnr(hip, cpu) if np.any(hip != cpu) else None   # (None: bit-identical)
    q["psnr_note"] = ("util/scores.py:47-48 on the fine colours of the CPU-baseline subset (%d rays) against the synthetic scene's "
                      "rgb_truth; tolerance of north_star: |delta| <= 0.01 dB" % n)
    if cpu_inds is not None:
        with torch.no_grad():
            sub = [t[:n] for t in data]
            B, Nc = sub[3].shape
            raw = pipe.model_coarse.forward_fused(sub[0], sub[2], Nc, pipe.position_encoder, pipe.direction_encoder)
            _, weights, _ = ops.composite(raw.view(B, Nc, 4), sub[3], sub[2], bool(pipe.args.white_background), None)
            hs = ops.hierarchical_samples(sub[1], sub[2], sub[3], weights, pipe.args.number_fine_samples, want_inds=True)
        q["sampler_index_equal_frac"] = float((hs["inds"].cpu().numpy() == cpu_inds).mean())
    return q


def cpu_train_baseline(workload, params, data_np, n_rays, threads, lr):
    """The reference's training step on this host's CPU: oracle/torch_cpu_path.train_step = the per-batch body of
    NerfSolver.train (solver/nerf_solver.py:76-89: forward under autograd, zero_grad, MSE coarse + fine, backward, Adam.step,
    loss.item()), calibrated against the reference's own NerfSolver object in the build container (losses bit-identical).
    Bounded sample: n_rays rays of the same frame, same initial weights; warm-up 1, median of 3 steps."""
    import numpy as np
    import torch
    from oracle import torch_cpu_path as T
    default_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        state = T.TrainState(params, lr=lr, workload=workload)
        data = [torch.from_numpy(np.ascontiguousarray(a[:n_rays])) for a in data_np]
        losses, ts = [T.train_step(state, data)], []
        for _ in range(3):
            t0 = time.perf_counter()
            losses.append(T.train_step(state, data))
            ts.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(default_threads)
    dt = statistics.median(ts)
    return {"value": n_rays * 256 / dt, "unit": "ray-samples/s", "cores": threads, "kind": "port",
            "seconds_per_step": dt, "loss_first": losses[0], "loss_last": losses[-1],
            "sample": f"{n_rays} rays of the same frame per step ({n_rays * 256} ray-samples), same initial weights, Adam lr {lr}; "
                      f"warm-up 1, median of 3 steps: oracle/torch_cpu_path.train_step = NerfSolver.train's per-batch body "
                      f"(solver/nerf_solver.py:76-89) on PyTorch-CPU fp32 autograd, torch.set_num_threads({threads})",
            "calibration_vs_reference_in_build_container": _calibration("train")}


# ---------------------------------------------------------------------------------------------------- HBM traffic
def _profiler_env_key(k):
    return k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX", "HSA_TOOLS", "RPD_", "OMNITRACE", "ROCPROFSYS"))


def under_profiler():
    if any(_profiler_env_key(k) for k in os.environ):
        return True
    return any(t in os.environ.get("LD_PRELOAD", "") for t in ("rocprof", "roctracer", "rocprofiler", "omnitrace"))


def pmc_traffic(argv_child, kernel_substr, timeout_s=240):
    """HBM bytes per average launch of the dominant kernel from two separate rocprofv3 --pmc passes over a short run of
    this same script (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots').  Units:
    KB as rocprofv3 reports them.  gfx950 correction: the guide's 2x under-count applies to wide (16 B/lane) streaming
    reads; this kernel's HBM reads are 4 B/lane position loads, and its FETCH_SIZE + WRITE_SIZE equals the known byte
    count of the exact-fp32 kernel, so the figure is used as reported."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    # never nest: when this process already runs under a profiler (rocprofv3 --kernel-trace ... -- python bench.py), a
    # --pmc child would combine counter collection with the parent's tracing in one process tree
    if under_profiler():
        return None, "skipped: bench.py itself runs under a profiler (counter passes are not nested inside a trace)"
    out = {}
    env = {k: v for k, v in os.environ.items() if not _profiler_env_key(k)}
    env["TMPDIR"] = "/tmp"
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="snerf_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__)] + argv_child
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"no {ctr} rows for the kernel (rc {r.returncode}): {r.stderr[-200:]}"
            out[ctr] = (sum(vals) / len(vals) * 1024.0, len(vals))
        except Exception as e:
            return None, f"{type(e).__name__}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out, None


# ---------------------------------------------------------------------------------------------------- training
TRAIN_LR = 3e-5


def gather_per_rank(entry, grouped):
    """[entry of rank 0, entry of rank 1, ...] on every rank (the record proves which ranks and devices took part)."""
    if not grouped:
        return [entry]
    import torch.distributed as dist
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, entry)
    return out


def collective_info(backend, n_floats, what):
    import torch.distributed as dist
    return {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world_size_seen_by_backend": dist.get_world_size(),
            "rank_seen_by_backend": dist.get_rank(), "op": what, "bytes": 4 * n_floats}


def raygen_loader(dev, res, n_frames, batch, world, rank):
    """SURVEY 8(f)-1 in the timed region: a multi-frame data set resident on the GPU, sharded BY IMAGE over the ranks
    (raygen.RayGenerator.for_rank), one shuffled epoch of `batch`-ray batches per pass (trainer.RayBatchLoader(shuffle=True) =
    DataLoader(shuffle=True), train.py:100).  Procedural images (no data set in the image)."""
    import numpy as np
    from smpl_nerf_amd import synthetic as syn
    from smpl_nerf_amd.raygen import RayGenerator
    from smpl_nerf_amd.trainer import RayBatchLoader
    poses = np.stack([syn.sphere_pose(360.0 * i / n_frames, 20.0 * np.sin(i), 2.4) for i in range(n_frames)])
    base = syn.procedural_image(res, res)          # one procedural frame, rolled per pose: distinct pixels per frame, cheap to build
    images = np.stack([np.roll(base, (3 * i) % res, axis=1) for i in range(n_frames)]).astype(np.float32)
    gen = RayGenerator.for_rank(poses, res, res, np.pi / 3, 1.0, 4.0, 64, dev, images=images, world=world, rank=rank)
    return gen, RayBatchLoader(gen, batch, seed=4321, shuffle=True)


def train_section(precision, workload, data, rays, steps, world, rank, dev, backend=None, loader=None, input_grads=False):
    """Secondary measurement (not `value`): data-parallel training steps - forward with saved activations, MSE
    coarse+fine, HIP backward, one flat all-reduce of the gradients, Adam - on `rays` rays per GPU drawn from this rank's
    frame (solver/nerf_solver.py:76-87), or, with `loader`, generated on the device per step from this rank's shard of a
    multi-frame data set (inside the timed region).  Fresh nets per call (the step updates them)."""
    import torch
    from smpl_nerf_amd import _lib
    from smpl_nerf_amd.dist import barrier, max_over_ranks
    from smpl_nerf_amd.trainer import DataParallelTrainer
    pipe, _, models = build_pipeline(dev, precision, workload)
    for m in models:
        m.train()
        for p in m.parameters():
            p.requires_grad_(True)
    if input_grads and workload == "append_vertices":      # the estimator is trained too: gradient through smpl_model into its poses
        pipe.smpl_estimator.goal_poses.requires_grad_(True)
        models = models + [pipe.smpl_estimator]
    # lr: small enough that both nets stay alive on this synthetic scene (at 1e-4 and above Adam's first steps push the
    # fine net's densities below zero everywhere: the rendered colour and every gradient become exactly 0, and the
    # backward kernels would be timed on all-zero operands)
    tr = DataParallelTrainer(pipe, models, lr=TRAIN_LR, sync_at_world_one=backend is not None)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    n_total = data[0].shape[0]
    if loader is not None:
        def endless():
            while True:
                for b in loader:
                    if b[0].shape[0] == rays:      # (the short last batch of an epoch is skipped: steps of equal size)
                        yield b
        feed = endless()
        next_batch = lambda i: next(feed)
    else:
        batches = []
        for _ in range(4):
            idx = torch.randperm(n_total, generator=g)[:rays].to(dev)
            batches.append([t[idx].contiguous() for t in data])
            if input_grads and workload == "append_smpl_params":
                batches[-1][4].requires_grad_(True)           # d loss / d goal_pose (69 per-ray columns)
        next_batch = lambda i: batches[i % 4]
    losses = []
    # warm-up: a quarter of a second of render passes, then two steps - after the CPU legs of this script the GPU has
    # been idle for tens of seconds and comes back at idle clocks; a 64-ray step (0.9 ms) measured right then ran at 8 ms
    # (profiles/r04: seen in one of two runs of the smpl_nerf line before this)
    # (inference passes, not steps: their number may differ between ranks, and they contain no collective)
    t_warm = time.perf_counter()
    with torch.no_grad():
        while time.perf_counter() - t_warm < 0.25:
            pipe([t[:4096].contiguous() for t in data] if data[0].shape[0] > 4096 else data)
            torch.cuda.synchronize()
    for i in range(2):
        losses.append(tr.step(next_batch(i)))
    barrier(dev)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(dev)
    tr.timing = {}
    with _lib.profile() as prof:
        t0 = time.perf_counter()
        for i in range(steps):
            losses.append(tr.step(next_batch(i)))
        t_host = time.perf_counter() - t0          # the Python loop returned: everything is enqueued
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0
        barrier(dev)
        dt = time.perf_counter() - t0
    kern = prof.summary()
    peak_mem = torch.cuda.max_memory_allocated(dev)
    dt = max_over_ranks(dt, dev)
    ar = [e0.elapsed_time(e1) for e0, e1 in tr.timing.get("allreduce_events", [])]
    tr.timing = None
    # The one-call data-parallel step (snerf_nerf_train_step_dp_f32) averages the gradients INSIDE the call: no event pair can
    # bracket the collective there, so it is timed on its own - the same ncclAllReduce(ncclAvg) of the same flat buffer on the
    # same communicator and stream, `steps` times back to back (outside the timed region above)
    in_call = getattr(tr, "_comm", None) not in (None, False)
    ar_calls_in_steps = getattr(tr, "collective_calls", 0)
    if in_call:
        scratch = torch.zeros_like(tr._flat_g)
        tr._comm.allreduce_avg_(scratch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            tr._comm.allreduce_avg_(scratch)
        e1.record()
        torch.cuda.synchronize()
        ar = [e0.elapsed_time(e1) / steps] * steps
        del scratch
    # what a user pays: the same steps without the event pairs of the profile above (r06: THIS loop is `ms_per_step` / `value` - two
    # hipEventRecord per step sit on the compute stream of the profiled loop, 3 % of a 64-ray step; the profiled loop's own figure is
    # kept as `ms_per_step_with_event_pairs`, its event pairs give `mlp_kernels_ms_per_step`)
    barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        losses.append(tr.step(next_batch(i)))
    t_host_plain = time.perf_counter() - t0
    torch.cuda.synchronize()
    barrier(dev)
    dt_events, dt = dt, max_over_ranks(time.perf_counter() - t0, dev)
    with torch.no_grad():   # the trained nets still render something (not collapsed to zero density)
        fine_std = float(pipe(next_batch(0))[1].std())
    losses = [float(l) for l in losses]
    evals = world * steps * rays * 256
    mlp_ms = sum(v[1] for k, v in kern.items() if k.startswith(("mlp_bwd", "mlp_fwd_train"))) / steps
    one_call = any(k.startswith("train_step") for k in kern)
    if one_call:
        # the step is one C-ABI call (snerf_nerf_train_step_f32): the event pair brackets ALL its kernels (compositing,
        # sampler, loss, Adam included), so the fraction below is a lower bound on the MLP kernels' own
        mlp_ms = sum(v[1] for k, v in kern.items() if k.startswith("train_step")) / steps
    flop_step = 3 * FLOP_PER_EVAL * rays * 256          # fwd + dgrad + wgrad of the two RenderRayNets
    peak = PEAK_F32_MFMA_TFLOPS if precision == "fp32" else PEAK_16BIT_MFMA_TFLOPS
    tf = flop_step / (mlp_ms * 1e-3) / 1e12 if mlp_ms else None
    return {"metric": "ray-samples/s, training step (fwd+bwd+all-reduce+Adam)", "value": evals / dt, "precision": precision,
            "rays_per_step_per_gpu": rays, "ms_per_step": dt / steps * 1e3, "ms_per_step_with_event_pairs": dt_events / steps * 1e3, "steps": steps,
            "host_enqueue_ms_per_step": t_host_plain / steps * 1e3, "host_ms_with_event_pairs": t_host / steps * 1e3,
            "c_abi_calls_per_step": sum(v[0] for v in kern.values()) / steps,
            "peak_allocated_bytes": int(peak_mem),
            "loss_first": losses[0], "loss_last": losses[-1], "rgb_fine_std_last_step": fine_std,
            "mlp_kernels_ms_per_step": mlp_ms, "mlp_algorithmic_tflops": tf, "mlp_peak_tflops": peak,
            "mlp_roofline_frac": tf / peak if tf else None,
            # smpl_nerf: the warp net's own FLOPs (forward + dgrad + wgrad of 100 -> 256 -> 3 on every ray-sample) are executed inside the
            # same call; with them in the numerator (VERDICT r04 #3 asks for this as a second key)
            "mlp_plus_warp_roofline_frac": ((flop_step + 3 * WARP_FLOP_PER_EVAL * rays * 256) / (mlp_ms * 1e-3) / 1e12 / peak
                                            if (tf and workload == "smpl_nerf" and one_call) else None),
            # ... and with the dgrad into the nets' inputs (d loss / d warped sample and d view direction through the encoder
            # columns of layer 0, the skip layer and directional_input: models/smpl_nerf_pipeline.py:49-56 under autograd),
            # which the nerf step does not compute: 2 x (60 x 256 x 2 + 24 x 128) FLOP per ray-sample
            "mlp_plus_warp_plus_input_dgrad_roofline_frac": (
                (flop_step + (3 * WARP_FLOP_PER_EVAL + INPUT_DGRAD_FLOP_PER_EVAL) * rays * 256) / (mlp_ms * 1e-3) / 1e12 / peak
                if (tf and workload == "smpl_nerf" and one_call) else None),
            "step_entry": ((("snerf_smpl_nerf_train_step_aux_f32" if any(k.startswith("train_step_smpl") for k in kern) else
                             "snerf_nerf_train_step_f32") + " (one C-ABI call per step; mlp_kernels_ms_per_step brackets the whole "
                            "call - for smpl_nerf that includes the warp net's kernels, which the FLOP count of the fraction leaves out)")
                           if one_call else "autograd (torch.autograd.Function per kernel group) + HipAdam"),
            "rays_per_chunk": tr.rays_per_chunk if one_call else None,
            "input_gradients": bool(input_grads),
            "kernels_ms_per_step": {k: v[1] / steps for k, v in sorted(kern.items())},
            "batches": ("generated on the device per step: RayBatchLoader(shuffle=True) over this rank's frames "
                        f"({loader.gen.n_frames} of the data set's frames, {loader.gen.n_rays} rays; raygen + pixel gather inside the "
                        "timed region)" if loader is not None else "4 resident batches drawn from this rank's frame"),
            "raygen_ms_per_step": (kern["raygen"][1] / steps if "raygen" in kern else None),
            "collective": (dict(collective_info(backend, sum(p.numel() for p in tr.params),
                                                ("ncclAllReduce(ncclAvg, fp32) of the flat gradient buffer INSIDE the one C-ABI call of a "
                                                 "step, on the compute stream between the backward and Adam (two launches: the coarse "
                                                 "net's segment, then the rest; the same on every rank whatever its batch size) - "
                                                 "snerf_nerf_train_step_dp_f32, RCCL bound by the library") if in_call else
                                                "one all-reduce (sum, then / world) of the flat fp32 gradient buffer per step"),
                                allreduce_ms_per_step=(sum(ar) / len(ar) if ar else None), allreduce_calls=len(ar),
                                allreduce_inside_the_step_call=bool(in_call),
                                rccl_comm_world_rank=([tr._comm.world, tr._comm.rank] if in_call else None),
                                broadcast_ms=tr.broadcast_ms,
                                timing=("HIP events around the same collective issued stand-alone on the same communicator, buffer and "
                                        "stream (inside the step call it cannot be bracketed)" if in_call else
                                        "HIP events on the compute stream around dist.sync (includes the / world)"))
                           if backend else "none (1 GPU, no process group)"),
            "per_rank": gather_per_rank({"rank": rank, "device": torch.cuda.get_device_name(dev),
                                         "arch": torch.cuda.get_device_properties(dev).gcnArchName, "device_index": dev.index,
                                         "ms_per_step": dt_own / steps * 1e3, "value": steps * rays * 256 / dt_own,
                                         "loss_last": losses[-1]}, backend is not None)}


# ---------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-rays", type=int, default=2048,
                    help="rays of the frame the CPU baseline renders per pass (2048 = the reference's default batch; 0 = skip)")
    ap.add_argument("--precision", choices=sorted(MODES), default="fp32",
                    help="matrix-core arithmetic of the headline measurement: fp32 (v_mfma_f32_16x16x4_f32, the reference's "
                         "arithmetic, default), bf16x6 (three bf16 parts, 6 products per MAC), f16x3 (two fp16 parts of "
                         "power-of-two-scaled operands, 3 products), bf16x3 (two bf16 parts)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="nerf")
    ap.add_argument("--res", type=int, default=128, help="frame edge in pixels (BASELINE configs[3]/[4] use 256)")
    ap.add_argument("--train-rays", type=int, default=4096, help="rays per GPU per training step (0 = skip the train section)")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--coarse-only", action="store_true",
                    help="BASELINE configs[0]: run_fine=0, 64 samples per ray (nerf workload)")
    ap.add_argument("--rays", type=int, default=0,
                    help="rays per step per GPU (default: the whole frame, res*res); the reference's own operating points are "
                         "2048 (train batch, config_parser.py:53), 800 (inference.py:231) and 64 (README quickstart)")
    ap.add_argument("--cpu-train-rays", type=int, default=512, help="rays per step of the CPU training baseline (0 = skip)")
    ap.add_argument("--points", default="64,800,2048",
                    help="comma-separated ray counts of the extra operating points (render; training at 2048 when among them); "
                         "empty = skip")
    ap.add_argument("--netwidth-points", default="512,768",
                    help="comma-separated --netwidth values above 256 (config_parser.py:20) measured as extra points of the nerf "
                         "workload: one frame rendered, one training step, each with its fraction of the fp32 MFMA peak; empty = skip")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): one frame per rank; strong: ONE frame split by rows over the ranks (dist.shard_rays), the "
                         "rendered rows all-gathered into the frame on every rank (dist.gather_rows) inside the timed region")
    ap.add_argument("--train-from-raygen", action="store_true",
                    help="training steps fed by RayBatchLoader(shuffle=True) over a multi-frame RayGenerator sharded by image "
                         "(rays generated on the device inside the timed region) instead of resident batches")
    ap.add_argument("--raygen-frames", type=int, default=64, help="frames of the synthetic data set of --train-from-raygen")
    ap.add_argument("--train-input-grads", action="store_true",
                    help="train section of the pose- / vertex-conditioned workloads with the gradient flowing into the per-ray "
                         "inputs: append_vertices trains its pose estimator (AppendVerticesSolver's second parameter group, "
                         "models/append_vertices_pipeline.py:30-58), append_smpl_params differentiates the goal pose")
    ap.add_argument("--no-alt", action="store_true", help="skip the other precision modes")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes behind roofline.traffic")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus))

    # stdout carries the ONE JSON line and nothing else: everything any library writes to file descriptor 1 from here on
    # (RCCL prints a version banner through the C library's stdout at its first collective) goes to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X (no CUDA/HIP device visible; there is no CPU path)")
    dev = torch.device("cuda", local_rank % ndev)   # one rank per GPU; the modulo only matters for dry runs of the
    torch.cuda.set_device(dev)                       # multi-rank path on fewer GPUs than ranks
    backend = None
    # SNERF_BENCH_FORCE_GROUP=1: form the process group at world size 1 too - the only way to take the RCCL branch of this
    # script (init with device_id, device barrier, max over ranks on the device) on a 1-GPU box
    grouped = world > 1 or bool(os.environ.get("SNERF_BENCH_FORCE_GROUP"))
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        backend = os.environ.get("SNERF_DIST_BACKEND", "nccl" if ndev >= world else "gloo")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from smpl_nerf_amd import _lib
    from smpl_nerf_amd.dist import barrier, max_over_ranks, shard_frames

    global FLOP_PER_EVAL
    if a.coarse_only and a.workload != "nerf":
        raise SystemExit("--coarse-only is BASELINE configs[0]: the nerf workload")
    run_fine = 0 if a.coarse_only else 1
    per_ray = 256 if run_fine else 64
    pipe, params, nets = build_pipeline(dev, a.precision, a.workload, run_fine)
    # algorithmic FLOPs of one RenderRayNet evaluation of THIS workload (2 x weight elements the output depends on: the
    # additional-input columns of the pose-conditioned nets count, AppendVerticesNet's dead vertices_net branch does not)
    FLOP_PER_EVAL = 2 * sum(p.numel() for k, p in nets[0].named_parameters() if k.endswith("weight") and not k.startswith("vertices_net"))
    # ... of which the fp32 inference kernel does not execute the additional-input columns per sample: they are per-ray
    # constants, folded into one vector per ray and layer (csrc/mlp.hip: mlp_add_fold_kernel; SNERF_MLP_FOLD=0 turns it off)
    d0 = nets[0].desc_for_rows() if hasattr(nets[0], "desc_for_rows") else None
    add_cols = int(getattr(nets[0], "additional_input_dim", 0)) if d0 is None else int(d0.add_dim)
    n_add_layers = 1 + sum(1 for i in range(nets[0].n_layers - 1) if i in nets[0].skips)
    FOLDED_FLOP_PER_EVAL = 2 * add_cols * nets[0].width * n_add_layers if os.environ.get("SNERF_MLP_FOLD", "1") != "0" else 0
    # weak scaling (default): each rank renders its own frame - rays of independent images shard across GPUs.
    # strong scaling: ONE frame, its rays split by rows over the ranks (dist.shard_rays), the rendered rows all-gathered into
    # the whole frame on every rank (dist.gather_rows) - the only collective a cooperative render needs.
    strong = a.scaling == "strong"
    from smpl_nerf_amd.dist import gather_rows, shard_rays
    frame_id = 0 if strong else shard_frames(world, rank)
    data_np = frame_inputs(a.workload, a.res, frame_id)
    if a.rays:
        if a.rays > data_np[0].shape[0]:
            raise SystemExit(f"--rays {a.rays} exceeds the {a.res}x{a.res} frame")
        data_np = [x[:a.rays] for x in data_np]
    frame_rays = data_np[0].shape[0]
    full_np = data_np
    if strong:
        data_np = shard_rays(data_np, world, rank)
    data = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in data_np]
    rays = data[0].shape[0]
    evals_per_step = rays * per_ray
    # what the whole job processes per step: `world` frames (weak) or the one frame (strong)
    job_evals_per_step = frame_rays * per_ray if strong else world * evals_per_step

    def render_step():
        out = pipe(data)
        if strong:
            return out, gather_rows(out[1], frame_rays)
        return out, None

    with torch.no_grad():
        for _ in range(a.warmup):
            out, frame = render_step()
        barrier(dev)
        torch.cuda.synchronize()
        _flush_c_stdio()       # (the barrier above was the group's first collective)
        with _lib.profile() as prof:
            t0 = time.perf_counter()
            for _ in range(a.steps):
                out, frame = render_step()
            t_host = time.perf_counter() - t0      # the Python loop returned: everything is enqueued
            torch.cuda.synchronize()
            elapsed_own = time.perf_counter() - t0
            barrier(dev)
            elapsed = time.perf_counter() - t0
        kern = prof.summary()
    elapsed = max_over_ranks(elapsed, dev)
    per_rank = gather_per_rank({"rank": rank, "device": torch.cuda.get_device_name(dev),
                                "arch": torch.cuda.get_device_properties(dev).gcnArchName, "device_index": dev.index,
                                "rays_per_step": rays, "ms_per_step": elapsed_own / a.steps * 1e3,
                                "value": a.steps * evals_per_step / elapsed_own}, grouped)
    strong_check = None
    if strong and rank == 0:      # the assembled frame against this rank rendering the whole frame alone (outside the timed region)
        with torch.no_grad():
            whole = pipe([torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in full_np])[1]
        strong_check = float((whole - frame).abs().max())

    def operating_point(n, steps=20):
        """Render `n` rays per step (the first n of the frame): ms per step, host enqueue time, C-ABI calls."""
        sub = [t[:n].contiguous() for t in data]
        with torch.no_grad():
            for _ in range(3):
                pipe(sub)
            torch.cuda.synchronize()
            with _lib.profile() as pp:
                t0 = time.perf_counter()
                for _ in range(steps):
                    pipe(sub)
                th = time.perf_counter() - t0
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            k = pp.summary()
            # the same steps without the per-call event pairs of the profile: the host cost a user pays
            torch.cuda.synchronize()
            calls0 = _lib.CALLS
            t0 = time.perf_counter()
            for _ in range(steps):
                pipe(sub)
            th_plain = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt_plain = time.perf_counter() - t0
            calls_plain = (_lib.CALLS - calls0) / steps
        # c_abi_calls_per_step: of the plain loop - what inference.py:251-252's `pipeline(data)` under no_grad costs (r06: the single-call
        # render entry); under the profile above the pipeline keeps one call per kernel so that every launch has its event pair
        return {"rays_per_step": n, "ms_per_step": dt_plain / steps * 1e3, "host_enqueue_ms_per_step": th_plain / steps * 1e3,
                "ray_samples_per_s": n * per_ray * steps / dt_plain, "c_abi_calls_per_step": calls_plain,
                "c_abi_calls_per_step_under_the_launch_profile": sum(v[0] for v in k.values()) / steps,
                "gpu_kernels_ms_per_step": sum(v[1] for v in k.values()) / steps,
                "ms_per_step_with_event_pairs": dt / steps * 1e3, "host_ms_with_event_pairs": th / steps * 1e3}

    def eval_without_no_grad(n, steps=20):
        """inference.py:247-253 as the reference ships it: eval mode, autograd recording ON - the pipelines then run their
        training forward (every layer input stored) and drop the graph.  dropin.install() wraps that loop in no_grad; this row
        is what an unwrapped caller pays."""
        sub = [t[:n].contiguous() for t in data]
        for _ in range(3):
            pipe(sub)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe(sub)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"rays_per_step": n, "mode": "eval_without_no_grad: inference.py:247-253 unwrapped (autograd recording on -> training "
                                           "forward, layer inputs stored and dropped)",
                "ms_per_step": dt / steps * 1e3, "ray_samples_per_s": n * per_ray * steps / dt,
                "peak_allocated_bytes": int(torch.cuda.max_memory_allocated(dev))}

    points = None
    if a.points and world == 1 and not grouped:
        points = [operating_point(n) for n in sorted({int(v) for v in a.points.split(",") if v.strip()}) if n <= rays]
        if 800 <= rays and run_fine:
            points.append(eval_without_no_grad(800))

    def netwidth_point(width, steps=3):
        """The nerf workload with RenderRayNets of `width` features (config_parser.py:20 --netwidth; the kernels of 320 / 384 / 448 /
        512 features, csrc/mlp_plan.h): this rank's rays rendered per step, and one-call training steps of min(4096, rays) rays -
        fractions of the fp32 MFMA peak on the ALGORITHMIC FLOPs of the unpadded nets, kernel times from HIP event pairs."""
        from smpl_nerf_amd import synthetic as syn
        from smpl_nerf_amd.nets import RenderRayNet
        from smpl_nerf_amd.ops import PositionalEncoder
        from smpl_nerf_amd.pipelines import NerfPipeline, PipelineArgs
        from smpl_nerf_amd.trainer import DataParallelTrainer
        wnets = []
        for seed in (401, 403):
            m = RenderRayNet(8, width, 60, 24, skips=[4])
            m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_render_ray_net_params(seed, 30.0, 10.0, width=width).items()})
            wnets.append(m.to(dev).eval())
        flop = 2 * sum(p.numel() for k, p in wnets[0].named_parameters() if k.endswith("weight"))
        wargs = PipelineArgs(white_background=0, run_fine=1, number_fine_samples=128, sigma_noise_std=0.0)
        wpipe = NerfPipeline(wnets[0], wnets[1], wargs, PositionalEncoder(10, 0), PositionalEncoder(4, 0))
        sub = [t.contiguous() for t in data[:5]]
        with torch.no_grad():
            wpipe(sub)
            torch.cuda.synchronize()
            with _lib.profile() as pp:
                t0 = time.perf_counter()
                for _ in range(steps):
                    wpipe(sub)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
            k = pp.summary()
        # the RenderRayNet launches' own time where the pipeline ran them as separate C-ABI calls; else the whole step
        # (above 512 features: the layer-by-layer path - its GEMM launches are the "linear_fwd" records, smpl_nerf_amd/layered.py)
        mlp_ms = sum(v[1] for n, v in k.items() if n.startswith(("mlp_fwd", "linear_fwd"))) / steps
        tf = flop * rays * 256 / (mlp_ms * 1e-3) / 1e12 if mlp_ms else None
        tf_step = flop * rays * 256 / dt / 1e12
        n_train = min(4096, rays)
        for m in wnets:
            m.train()
        tr = DataParallelTrainer(wpipe, wnets, lr=5e-4)
        tb = [t[:n_train].contiguous() for t in data[:5]]
        tr.step(tb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(tb)
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t0) / steps
        ttf = 3 * flop * n_train * 256 / dts / 1e12
        return {"netwidth": width, "kernel_width": (width + 63) // 64 * 64 if width <= 512 else None,
                "path": "fused register-resident chain" if width <= 512 else "layer by layer (snerf_linear_*: one fp32 MFMA GEMM per nn.Linear)",
                "flop_per_eval": flop, "rays_per_step": rays,
                "render_ms_per_step": dt * 1e3, "render_ray_samples_per_s": rays * 256 / dt,
                "render_mlp_kernels_ms_per_step": mlp_ms or None, "render_mlp_roofline_frac": tf / PEAK_F32_MFMA_TFLOPS if tf else None,
                "render_roofline_frac_whole_step": tf_step / PEAK_F32_MFMA_TFLOPS,
                "train_rays_per_step": n_train, "train_ms_per_step": dts * 1e3, "train_ray_samples_per_s": n_train * 256 / dts,
                "train_step_roofline_frac_whole_step": ttf / PEAK_F32_MFMA_TFLOPS, "train_one_call": tr._one_call_state() is not None,
                "peak_tflops": PEAK_F32_MFMA_TFLOPS, "dtype": "f32"}

    width_points = None
    if a.netwidth_points and world == 1 and not grouped and a.workload == "nerf" and run_fine and a.precision == "fp32":
        try:
            width_points = [netwidth_point(int(v)) for v in a.netwidth_points.split(",") if v.strip()]
        except Exception as e:
            width_points = [{"error": f"{type(e).__name__}: {e}"}]

    def mlp_launch_stats(k, steps):
        mlp = {n: v for n, v in k.items() if n.startswith("mlp_fwd")}
        calls = sum(v[0] for v in mlp.values())
        ms = sum(v[1] for v in mlp.values())
        return calls, ms / calls, steps * evals_per_step / calls

    def roofline_of(prec, calls, avg_ms, units_per_launch):
        kname, products, _ = MODES[prec]
        alg = FLOP_PER_EVAL * units_per_launch / (avg_ms * 1e-3) / 1e12     # 1 215 744 FLOP per ray-sample
        peak = PEAK_F32_MFMA_TFLOPS if prec == "fp32" else PEAK_16BIT_MFMA_TFLOPS
        r = {"bound": "mfma", "kernel": kname + " (coarse + fine launches)", "achieved": alg, "peak": peak,
             "unit": "TFLOP/s", "frac": alg / peak, "traffic": None, "avg_launch_ms": avg_ms, "launches": calls,
             "flop_per_unit": FLOP_PER_EVAL, "units_per_launch": units_per_launch}
        if prec == "fp32":
            if FOLDED_FLOP_PER_EVAL:    # algorithmic FLOPs the kernel does not pay per sample (per-ray inputs folded)
                r["flop_per_unit_executed_per_sample"] = FLOP_PER_EVAL - FOLDED_FLOP_PER_EVAL
                r["frac_on_executed_flops"] = alg / peak * (FLOP_PER_EVAL - FOLDED_FLOP_PER_EVAL) / FLOP_PER_EVAL
            r["peak_note"] = "fp32-input MFMA (v_mfma_f32_16x16x4_f32), exact fp32: 157.3 TFLOP/s"
        else:
            r["peak_note"] = ("dense 16-bit MFMA peak 2500 TFLOP/s; `achieved` counts the network's own fp32 MACs "
                              "(algorithmic), not the split products")
            r["products_per_fp32_mac"] = products
            r["mfma_issue_frac"] = alg * products / peak        # executed 16-bit products against the same peak
            r["achieved_vs_fp32_mfma_peak"] = alg / PEAK_F32_MFMA_TFLOPS
        return r

    alt = {}
    if not a.no_alt:
        with torch.no_grad():
            for prec in sorted(MODES):
                if prec == a.precision:
                    continue
                pipe.set_precision(prec)
                for _ in range(2):
                    o2 = pipe(data)
                torch.cuda.synchronize()
                with _lib.profile() as p2:
                    t0 = time.perf_counter()
                    for _ in range(a.steps):
                        o2 = pipe(data)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / a.steps
                calls, avg_ms, upl = mlp_launch_stats(p2.summary(), a.steps)
                r = roofline_of(prec, calls, avg_ms, upl)
                alt[prec] = {"ray_samples_per_s_per_gpu": evals_per_step / dt, "ms_per_step": dt * 1e3, "steps": a.steps,
                             "dtype": MODES[prec][2],
                             "rgb_fine_max_abs_diff_vs_" + a.precision: float((o2[1] - out[1]).abs().max()),
                             "roofline": r}
            pipe.set_precision(a.precision)

    train = train_alt = None
    if a.train_rays > 0 and run_fine:
        try:
            loader = None
            if a.train_from_raygen:
                if a.workload != "nerf":
                    raise SystemExit("--train-from-raygen feeds the nerf workload (five-tensor batches)")
                _, loader = raygen_loader(dev, a.res, a.raygen_frames, min(a.train_rays, frame_rays), world, rank)
            tb = backend if grouped else None
            train = train_section(a.precision, a.workload, data, min(a.train_rays, frame_rays if loader else rays), a.train_steps,
                                  world, rank, dev, tb, loader, a.train_input_grads)
            if a.points and world == 1:   # the reference's own batch sizes (config_parser.py:53, README quickstart)
                pts = []
                for n in sorted({int(v) for v in a.points.split(",") if v.strip()}):
                    if n == min(a.train_rays, rays) or n > rays:
                        continue
                    t = train_section(a.precision, a.workload, data, n, a.train_steps, world, rank, dev, tb)
                    pts.append({k: t[k] for k in ("value", "rays_per_step_per_gpu", "ms_per_step", "host_enqueue_ms_per_step",
                                                   "c_abi_calls_per_step", "ms_per_step_with_event_pairs", "mlp_kernels_ms_per_step", "mlp_roofline_frac",
                                                   "peak_allocated_bytes")})
                train["operating_points"] = pts
            if not a.no_alt:
                train_alt = {}
                for prec in ("bf16x6", "f16x3", "fp32"):
                    if prec != a.precision:
                        t = train_section(prec, a.workload, data, min(a.train_rays, rays), a.train_steps, world, rank, dev, tb)
                        train_alt[prec] = {k: t[k] for k in ("value", "ms_per_step", "loss_first", "loss_last",
                                                             "mlp_kernels_ms_per_step", "mlp_algorithmic_tflops",
                                                             "mlp_roofline_frac")}
        except Exception as e:  # the render metric above stays valid
            train = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        value = a.steps * job_evals_per_step / elapsed
        # dominant kernel = the fused encode+MLP kernel; it is launched twice per step (coarse: rays*64 samples, fine:
        # rays*192).  Roofline over ALL its launches in the timed region, so that the average launch duration is the
        # number rocprofv3 --stats reports for the kernel.
        calls, avg_ms, upl = mlp_launch_stats(kern, a.steps)
        roof = roofline_of(a.precision, calls, avg_ms, upl)
        alg_bytes = upl * (12 + 16) + rays * 12       # positions in, raw out, one direction per ray
        roof["algorithmic_hbm_bytes_per_launch"] = alg_bytes
        if world == 1 and not a.no_pmc:
            child = ["--steps", "3", "--warmup", "1", "--cpu-rays", "0", "--train-rays", "0", "--no-alt", "--no-pmc",
                     "--precision", a.precision, "--workload", a.workload, "--res", str(a.res), "--points", ""]
            if a.coarse_only:
                child.append("--coarse-only")
            if a.rays:
                child += ["--rays", str(a.rays)]
            t, err = pmc_traffic(child, MODES[a.precision][0].replace("snerf::", ""))
            if t:
                roof["traffic"] = t["FETCH_SIZE"][0] + t["WRITE_SIZE"][0]
                roof["traffic_detail"] = {"fetch_bytes": t["FETCH_SIZE"][0], "write_bytes": t["WRITE_SIZE"][0],
                                          "launches_sampled": t["FETCH_SIZE"][1],
                                          "source": "live: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over "
                                                    "`bench.py " + " ".join(child) + "`, per average launch"}
            else:
                roof["traffic_detail"] = {"error": err}
        if a.coarse_only:
            metric = f"ray-samples/sec (coarse-only) at {a.res}^2 / 64 samples"
        else:
            metric = f"ray-samples/sec (coarse+fine) at {a.res}^2 / 64+128 samples"
        line = {
            "metric": metric,
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": MODES[a.precision][2], "data": "synthetic",
            "config": {"workload": (COARSE_ONLY if a.coarse_only else WORKLOADS[a.workload]).format(r=a.res)
                                   + (f"; first {rays} rays of the frame per step" if a.rays else ""),
                       "rays_per_step_per_gpu": rays, "ray_samples_per_ray": per_ray,
                       "parallelism": (f"dp{world} (one frame split by rows over the ranks; the rendered rows are all-gathered into "
                                       f"the frame on every rank each step" if strong else
                                       f"dp{world} (rays of independent frames per rank, no data-path collective")
                                      + (f"; process group on {backend}" if backend else "") + ")"},
            "rays_per_s": a.steps * (frame_rays if strong else world * rays) / elapsed,
            "host_enqueue_ms_per_step": t_host / a.steps * 1e3,
            "c_abi_calls_per_step": sum(v[0] for v in kern.values()) / a.steps,
            "roofline": roof,
            "precision": a.precision,
            "kernels_ms_per_step": {k: v[1] / a.steps for k, v in sorted(kern.items())},
        }
        if grouped:     # an N > 1 record proves itself: who took part, on which devices, how long each rank took
            line["per_rank"] = per_rank
            line["collective"] = (dict(collective_info(backend, frame_rays * 3, "all-gather of the rendered rows (rgb_fine) into the "
                                                       "frame on every rank, once per step (dist.gather_rows)"),
                                       strong_frame_max_abs_diff_vs_single_rank_render=strong_check) if strong else
                                  dict(collective_info(backend, 0, "none in the render path (barrier + max-over-ranks of the elapsed "
                                                                   "time only); the train section has the gradient all-reduce")))
        if alt:
            line["other_precisions_1gpu"] = alt
            if "bf16x6" in alt and a.precision == "fp32":
                # the fastest mode whose error against float64 the tests hold to <= the exact-fp32 kernel's own
                # (test_split_bf16_stress_against_fp32_kernel: aggregate RMS <= 1.0 x; tools/precision_study_gpu.py measures
                # it) - reported, not the headline: `value` stays on the reference's own arithmetic
                b = alt["bf16x6"]
                line["fastest_mode_with_fp32_kernel_accuracy"] = {
                    "precision": "bf16x6", "ray_samples_per_s_per_gpu": b["ray_samples_per_s_per_gpu"],
                    "speedup_vs_value": b["ray_samples_per_s_per_gpu"] * world / value,
                    "rgb_fine_max_abs_diff_vs_fp32": b["rgb_fine_max_abs_diff_vs_fp32"],
                    "roofline_frac_vs_16bit_peak": b["roofline"]["frac"], "mfma_issue_frac": b["roofline"]["mfma_issue_frac"]}
        warp = {n: v for n, v in kern.items() if n.startswith("warp_fwd")}
        if warp:    # smpl_nerf: the warp-field kernel beside the RenderRayNet kernel, on ALGORITHMIC FLOPs
            wcalls, wms = sum(v[0] for v in warp.values()), sum(v[1] for v in warp.values())
            wupl = a.steps * evals_per_step / wcalls
            wtf = WARP_FLOP_PER_EVAL * wupl / (wms / wcalls * 1e-3) / 1e12
            wpeak = PEAK_F32_MFMA_TFLOPS if a.precision == "fp32" else PEAK_16BIT_MFMA_TFLOPS
            # what the matrix pipe executes per sample: the fp32 inference kernel folds the per-ray pose columns (unless
            # SNERF_WARP_FOLD=0), the split-precision kernel runs all 7 k-blocks
            folded = a.precision == "fp32" and os.environ.get("SNERF_WARP_FOLD", "1") != "0"
            wexec = WARP_FLOP_FOLDED if folded else WARP_FLOP_PADDED
            line["warp_roofline"] = {
                "bound": "mfma", "kernel": "snerf::warp_fwd_resident_kernel<256, 16, false>" if a.precision == "fp32"
                else "snerf::warp_fwd_bf16_kernel<256, 8>", "achieved": wtf, "peak": wpeak, "unit": "TFLOP/s",
                "frac": wtf / wpeak, "avg_launch_ms": wms / wcalls, "launches": wcalls, "flop_per_unit": WARP_FLOP_PER_EVAL,
                "units_per_launch": wupl, "frac_on_padded_flops_as_executed": wtf / wpeak * wexec / WARP_FLOP_PER_EVAL,
                "padded_flop_per_unit": wexec, "pose_columns_folded_per_ray": bool(folded)}
        if points:
            line["operating_points_render"] = points
        if width_points:
            line["netwidth_points"] = width_points
        if train is not None:
            line["train"] = train
        if train_alt:
            line["train_other_precisions"] = train_alt
        if world == 1 and a.cpu_rays > 0:
            # (append_vertices: the reference materialises 83 KB per sample - a far smaller sample fills the time budget)
            n = min(a.cpu_rays if a.workload != "append_vertices" else min(a.cpu_rays, 64), rays)
            info, ref = cpu_baseline(a.workload, params, data_np, n, run_fine)
            cpu_inds = info.pop("_sampler_inds", None)
            line["cpu_baseline"] = info
            if a.workload != "append_vertices":     # (its CPU figure is the coarse pass only)
                line.update(quality_keys(out, ref, data_np[-1], n, cpu_inds, pipe, data))
            if (isinstance(train, dict) and "error" not in train and a.cpu_train_rays > 0 and run_fine
                    and a.workload in ("nerf", "smpl_nerf")):
                train["cpu_baseline"] = cpu_train_baseline(a.workload, params, data_np, min(a.cpu_train_rays, rays),
                                                           info["cores"], TRAIN_LR)
            if a.workload != "append_vertices":     # (its CPU figure is the coarse pass only)
                line["rgb_fine_max_abs_diff_vs_oracle"] = float(np.max(np.abs(out[1][:n].cpu().numpy() - ref[1])))
            line["rgb_coarse_max_abs_diff_vs_oracle"] = float(np.max(np.abs(out[0][:n].cpu().numpy() - ref[0])))
        _flush_c_stdio()
        print(json.dumps(line), file=json_out, flush=True)
    if grouped:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
