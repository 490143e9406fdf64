#!/usr/bin/env python3
"""BUILD-CONTAINER ONLY (needs /root/reference; the GPU box never sees it).

Times the reference's own CPU forward path - NerfPipeline.forward (models/nerf_pipeline.py:14-67) with the
reference's RenderRayNet modules - beside oracle/torch_cpu_path.nerf_pipeline_forward on the same rays, weights and
thread count, checks that the outputs are identical and that the two speeds agree within +-10 %, and
writes oracle/cpu_baseline_calibration.json.  bench.py's cpu_baseline then times torch_cpu_path on the GPU host as
"the reference's CPU path" (kind "port", calibrated).

    python oracle/calibrate_cpu_baseline.py [--rays 2048] [--threads 8]
"""
import argparse
import importlib.util
import json
import os
import platform
import statistics
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import numpy as np
import torch


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def median_time(fn, repeats=5, warmup=1):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    a = ap.parse_args()
    torch.set_num_threads(a.threads)

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    U, RenderRayNet, NerfPipeline, _, _ = mg._import_reference()
    # the golden generator's searchsorted stand-in also records its operands; time the plain binding
    U.searchsorted = lambda a, v, out=None, side="left": torch.searchsorted(a, v, right=(side != "left"))

    from oracle import torch_cpu_path as T
    from smpl_nerf_amd import synthetic as syn

    pc, pf = syn.make_scene_nets(101)
    data_np = syn.frame_batch(128, 128, seed=7)
    data = [torch.from_numpy(np.ascontiguousarray(x[:a.rays])) for x in data_np]

    def ref_net(p):
        m = RenderRayNet(8, 256, 60, 24, skips=[4])
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        return m.eval()

    args = mg.Args()
    ref_pipe = NerfPipeline(ref_net(pc), ref_net(pf), args, U.PositionalEncoder(10, False), U.PositionalEncoder(4, False))
    Pc, Pf = T.tparams(pc), T.tparams(pf)
    targs = T.Args()
    pe, de = T.PositionalEncoder(10, False), T.PositionalEncoder(4, False)

    with torch.no_grad():
        t_ref, ts_ref, out_ref = median_time(lambda: ref_pipe(data))
        t_port, ts_port, out_port = median_time(lambda: T.nerf_pipeline_forward(Pc, Pf, targs, pe, de, data))
        # interleave once more to catch drift of the container's clocks
        t_ref2, _, _ = median_time(lambda: ref_pipe(data), repeats=3, warmup=0)
        t_port2, _, _ = median_time(lambda: T.nerf_pipeline_forward(Pc, Pf, targs, pe, de, data), repeats=3, warmup=0)
    t_ref, t_port = min(t_ref, t_ref2), min(t_port, t_port2)
    same = all(torch.equal(x, y) for x, y in zip(out_ref, out_port))
    evals = a.rays * 256
    res = {"host_cpu": cpu_model(), "threads": a.threads, "rays": a.rays, "ray_samples": evals,
           "reference_seconds": t_ref, "port_seconds": t_port,
           "reference_ray_samples_per_s": evals / t_ref, "port_ray_samples_per_s": evals / t_port,
           "port_over_reference_speed": t_ref / t_port, "outputs_bit_identical": bool(same),
           "reference": "NerfPipeline.forward (models/nerf_pipeline.py:14-67), torch %s CPU, no_grad" % torch.__version__,
           "port": "oracle/torch_cpu_path.nerf_pipeline_forward", "method": "warm-up 1, median of 5, best of two rounds"}
    print(json.dumps(res, indent=1))
    assert same, "the restatement's outputs differ from the reference's"
    assert 0.9 <= res["port_over_reference_speed"] <= 1.1, "restatement is not within +-10 % of the reference's speed"
    with open(os.path.join(HERE, "cpu_baseline_calibration.json"), "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
