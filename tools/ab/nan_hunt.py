"""Hunt for the intermittent non-finite gradients: device properties, then fresh trainers (nerf / smpl_nerf, one-call / autograd,
three precisions), every step checked, the location of every non-finite gradient element printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch

from smpl_nerf_amd.nets import RenderRayNet, WarpFieldNet
from smpl_nerf_amd.ops import PositionalEncoder
from smpl_nerf_amd.pipelines import NerfPipeline, PipelineArgs, SmplNerfPipeline
from smpl_nerf_amd.trainer import DataParallelTrainer
import test_gpu_round4 as t4

dev = torch.device("cuda:0")
pr = torch.cuda.get_device_properties(0)
print("device:", pr.name, getattr(pr, "gcnArchName", "?"), "CUs", pr.multi_processor_count, "mem GiB", pr.total_memory / 2**30,
      "free GiB", torch.cuda.mem_get_info()[0] / 2**30, flush=True)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
total_bad = 0
for rnd in range(rounds):
    for kind in ("nerf", "smpl_nerf"):
        for prec in ("fp32", "f16x3"):
            for one_call in (None, False):
                for ident in (0, 1):
                    torch.manual_seed(5)
                    pe, de = PositionalEncoder(10, ident), PositionalEncoder(4, 0)
                    nets = [RenderRayNet(4, 256, 3 * pe.output_dim, 3 * de.output_dim, skips=[2]).to(dev).train() for _ in range(2)]
                    for m in nets:
                        m.precision = prec
                    if kind == "smpl_nerf":
                        mw = WarpFieldNet(3, 128, 3 * pe.output_dim, 40).to(dev).train()
                        pipe = SmplNerfPipeline(nets[0], nets[1], mw, PipelineArgs(), pe, de, PositionalEncoder(10, 0))
                        models, batch = nets + [mw], t4._smpl_batch(dev, 40)
                    else:
                        pipe = NerfPipeline(nets[0], nets[1], PipelineArgs(), pe, de)
                        models, batch = nets, t4._batch(dev, 40, stride=61)
                    tr = DataParallelTrainer(pipe, models, lr=1e-4, one_call=one_call)
                    tag = f"round {rnd} {kind} {prec} {'one_call' if tr._one_call_state() is not None else 'autograd'} ident {ident}"
                    for step in range(3):
                        loss = float(tr.step(batch))
                        torch.cuda.synchronize()
                        for mi, m in enumerate(models):
                            for k, p in m.named_parameters():
                                g = p.grad
                                if g is None:
                                    print(tag, "step", step, "net", mi, k, "grad is None", flush=True)
                                    continue
                                nf = ~torch.isfinite(g)
                                if bool(nf.any()):
                                    total_bad += 1
                                    idx = nf.reshape(-1).nonzero().reshape(-1)
                                    print(tag, "step", step, f"loss {loss:.6f}", "net", mi, k, tuple(g.shape), "non-finite", int(idx.numel()), "of", g.numel(),
                                          "first", int(idx[0]), "last", int(idx[-1]), flush=True)
                    del tr, pipe, models, nets
print("non-finite gradient tensors seen:", total_bad)
