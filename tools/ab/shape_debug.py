"""One-call step vs autograd step for one network / batch shape: per-tensor gradient and parameter differences (debug aid)."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder
from smpl_nerf_amd.pipelines import NerfPipeline, PipelineArgs
from smpl_nerf_amd.trainer import DataParallelTrainer
dev = torch.device("cuda:0")
shape = dict(n_layers=3, width=100, skips=(0, 1), B=129, Nc=64, Nf=128, chunk=50, wb=0)
if len(sys.argv) > 1 and sys.argv[1] == "4":
    shape = dict(n_layers=8, width=256, skips=(), B=300, Nc=8, Nf=200, chunk=128, wb=1)
B, Nc, Nf = shape["B"], shape["Nc"], shape["Nf"]
rng = np.random.default_rng(B + Nc)
o = rng.normal(0, 0.2, (B, 3)).astype(np.float32) + np.array([0, 0, 2.4], np.float32)
d = rng.normal(0, 0.3, (B, 3)).astype(np.float32) + np.array([0, 0, -1], np.float32)
z = np.sort(rng.uniform(1.0, 4.0, (B, Nc)).astype(np.float32), -1)
samples = o[:, None, :] + d[:, None, :] * z[:, :, None]
gt = rng.uniform(0, 1, (B, 3)).astype(np.float32)
batch = [torch.from_numpy(a).to(dev) for a in (samples.astype(np.float32), o, d, z, gt)]
runs = []
for one_call in (None, False):
    torch.manual_seed(21)
    nets = []
    for _ in range(2):
        m = RenderRayNet(shape["n_layers"], shape["width"], 60, 24, skips=list(shape["skips"])).to(dev).train()
        with torch.no_grad():
            m.sigma_out_layer.weight.mul_(20.0)
        nets.append(m)
    pipe = NerfPipeline(nets[0], nets[1], PipelineArgs(white_background=shape["wb"], number_fine_samples=Nf), PositionalEncoder(10, 0), PositionalEncoder(4, 0))
    tr = DataParallelTrainer(pipe, nets, lr=1e-3, one_call=one_call)
    tr.rays_per_chunk = shape["chunk"]
    l1 = float(tr.step(batch))
    g1 = [p.grad.clone() for p in tr.params]
    p1 = [p.detach().clone() for p in tr.params]
    l2 = float(tr.step(batch))
    g2 = [p.grad.clone() for p in tr.params]
    runs.append((l1, l2, g1, p1, g2, [p.detach().clone() for p in tr.params], [n for m in nets for n, _ in m.named_parameters()]))
a, b = runs
print("losses", a[0], b[0], a[1], b[1])
for i, name in enumerate(a[6]):
    g1a, g1b, p1a, p1b, g2a, g2b, p2a, p2b = a[2][i], b[2][i], a[3][i], b[3][i], a[4][i], b[4][i], a[5][i], b[5][i]
    gm = float(g1b.abs().max())
    print(f"{i:2d} {name:32s} |g1| {gm:.2e} d g1 {float((g1a-g1b).abs().max()):.2e}  d p1 {float((p1a-p1b).abs().max()):.2e}  |g2| {float(g2b.abs().max()):.2e} d g2 {float((g2a-g2b).abs().max()):.2e}  d p2 {float((p2a-p2b).abs().max()):.2e}")
