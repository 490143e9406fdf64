"""HipAdam against torch.optim.Adam (CPU) element by element: prints the worst elements with their histories."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from smpl_nerf_amd.trainer import HipAdam
dev = torch.device("cuda:0")
for wd in (0.0, 0.01):
    n = 200000
    gen = torch.Generator().manual_seed(11)
    p0 = (torch.rand(n, generator=gen) - 0.5) * 0.25
    q = torch.nn.Parameter(p0.clone())
    ref = torch.optim.Adam([q], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    flat_p = p0.clone().to(dev); flat_g = torch.zeros_like(flat_p)
    P = torch.nn.Parameter(flat_p)   # shares storage
    opt = HipAdam([P], flat_p, flat_g, [flat_g.view(n)], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    hist = []
    for t in range(6):
        g = torch.randn(n, generator=gen) * (10.0 ** float(torch.randint(-4, 1, (1,), generator=gen)))
        q.grad = g.clone(); ref.step()
        P.grad = g.to(dev); opt.step()
        torch.cuda.synchronize()
        d = (flat_p.cpu() - q.detach()).abs()
        hist.append((g, flat_p.cpu().clone(), q.detach().clone(), opt.exp_avg.cpu().clone(), ref.state[q]["exp_avg"].clone(),
                     opt.exp_avg_sq.cpu().clone(), ref.state[q]["exp_avg_sq"].clone()))
        if t == 0 and wd == 0.0:
            bad = torch.nonzero(d > 0).flatten()[:6]
            for j in bad.tolist():
                print("MISMATCH", float(p0[j]).hex(), float(g[j]).hex(), float(hist[-1][3][j]).hex(), float(hist[-1][5][j]).hex(),
                      float(hist[-1][1][j]).hex(), float(hist[-1][2][j]).hex())
        print(f"wd {wd} step {t+1}: max abs {float(d.max()):.3e}  p!=ref {int((d > 0).sum())}  m!=ref {int((hist[-1][3] != hist[-1][4]).sum())} "
              f"v!=ref {int((hist[-1][5] != hist[-1][6]).sum())} of {n}  gscale {float(g.abs().mean()):.2e}")
    i = int(d.argmax())
    for t, (g, a, b, m1, m2, v1, v2) in enumerate(hist):
        print(f"  t={t+1} g={float(g[i]):.9e} p_hip={float(a[i]):.9e} p_ref={float(b[i]):.9e} m={float(m1[i]):.9e}/{float(m2[i]):.9e} v={float(v1[i]):.9e}/{float(v2[i]):.9e}")
