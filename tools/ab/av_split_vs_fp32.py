"""AppendVerticesPipeline (configs[4], frozen estimator: the vertex floats are per-ray additional inputs, 20 670 columns = wide
wgrad jobs of their own) - one training step in bf16x6 / f16x3 against the same step in fp32, one-call and chunked."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch
import test_gpu_round2 as t2
import test_gpu_round4 as t4
from smpl_nerf_amd.trainer import DataParallelTrainer
dev = torch.device("cuda:0")
b = t4._batch(dev, 100, stride=53)
images = (torch.arange(100, device=dev) % 10)
batch = b[:4] + [images, b[4]]
ref = None
bad = 0
for prec, chunk in (("fp32", 0), ("bf16x6", 0), ("bf16x6", 37), ("f16x3", 0), ("f16x3", 37), ("f16x3", 7)):
    pipe, _ = t2._av_pipeline(dev, prec)
    nets = [pipe.model_coarse, pipe.model_fine]
    for m in nets:
        m.train()
    tr = DataParallelTrainer(pipe, nets, lr=1e-6)
    tr.rays_per_chunk = chunk
    loss = float(tr.step(batch))
    g = {f"{i}.{k}": p.grad.clone() for i, m in enumerate(nets) for k, p in m.named_parameters() if p.grad is not None}
    if ref is None:
        ref, top = g, max(float(v.norm()) for v in g.values())
        print(f"fp32: loss {loss:.6f}, one-call {tr._one_call_state() is not None}, largest |g| {top:.3e}")
        continue
    worst = sorted(((float((g[k] - ref[k]).norm()) / top, k) for k in ref if not k.startswith("1.")), reverse=True)[:2]
    worst_f = sorted(((float((g[k] - ref[k]).norm()) / top, k) for k in ref if k.startswith("1.")), reverse=True)[:1]
    ok = worst[0][0] <= 2e-4 and all(bool(torch.isfinite(v).all()) for v in g.values())
    bad += not ok
    print(("ok  " if ok else "BAD ") + f"{prec} chunk {chunk}: loss {loss:.6f}; coarse net worst |g - g32| / largest: {[(f'{e:.1e}', k) for e, k in worst]}; fine net {[(f'{e:.1e}', k) for e, k in worst_f]}", flush=True)
print("bad:", bad)
