"""RenderRayNet forward + backward over sample counts that give the wgrad ragged / empty trailing chunks; with
SNERF_DEBUG_POISON_LDS=1 every kernel starts on LDS full of NaNs (csrc/snerf_common.h), so anything that reads LDS it has not
written shows up as a non-finite gradient.  Prints one line per count and `bad: K`.

    SNERF_DEBUG_POISON_LDS=1 python tools/ab/lds_poison_sweep.py
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from smpl_nerf_amd.nets import RenderRayNet
dev = torch.device('cuda:0')
torch.manual_seed(1)
total = 0
for prec in ("fp32", "bf16x6", "f16x3"):
    net = RenderRayNet(4, 256, 60, 24, skips=[2]).to(dev).train()
    net.precision = prec
    for n in (1, 100, 2560, 7680, 8160, 8000, 4096, 12288, 16384, 16400, 40000, 262144):
        x = torch.randn(n, 84, device=dev)
        for p in net.parameters():
            p.grad = None
        out = net(x)
        out.square().mean().backward()
        torch.cuda.synchronize()
        bad = [k for k, p in net.named_parameters() if not bool(torch.isfinite(p.grad).all())]
        total += len(bad) + (0 if bool(torch.isfinite(out).all()) else 1)
        print(prec, n, 'out finite', bool(torch.isfinite(out).all()), 'bad grads:', bad, flush=True)
print("bad:", total)
