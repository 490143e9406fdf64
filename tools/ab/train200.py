"""Diagnostic: the 200-step reference training run (tests/golden/g15_train200.npz) on the HIP path under variants."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smpl_nerf_amd import synthetic as syn
from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder
from smpl_nerf_amd.pipelines import NerfPipeline, PipelineArgs
from smpl_nerf_amd.trainer import DataParallelTrainer

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
strict = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fused = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "g15_train200.npz")))
pc, pf = syn.make_scene_nets(101)
def net(p):
    m = RenderRayNet(8, 256, 60, 24, skips=[4]); m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}); m.precision = prec
    return m.to(dev).train()
mc, mf = net(pc), net(pf)
pipe = NerfPipeline(mc, mf, PipelineArgs(strict_cumsum=strict), PositionalEncoder(10, 0), PositionalEncoder(4, 0))
tr = DataParallelTrainer(pipe, [mc, mf], lr=float(g["lr"][0]), fused=bool(fused))
data = [torch.from_numpy(a).to(dev) for a in syn.frame_batch(128, 128, seed=7)]
idx = torch.from_numpy(g["idx"]).to(dev)
L = torch.stack([tr.step([t[idx[i]] for t in data]) for i in range(200)]).double().cpu().numpy()
rel = np.abs(L - g["losses"]) / g["losses"]
vi = torch.from_numpy(g["val_idx"]).to(dev)
mc.eval(); mf.eval()
with torch.no_grad():
    vb = [t[vi] for t in data]; out = pipe(vb)
    vl = float(tr.loss(out[0], out[1], vb[-1])); psnr = -10 * np.log10(float(torch.mean((out[1] - vb[-1]) ** 2)))
print(f"prec={prec} strict={strict} fused_adam={fused} fold={os.environ.get('SNERF_WGRAD_FOLD','1')}: rel dev max {rel.max():.3e} mean {rel.mean():.3e} windows {[float('%.3g'%rel[i:i+50].max()) for i in range(0,200,50)]} "
      f"val_loss {vl:.6f} (ref {g['val_loss'][0]:.6f}, {100*(vl/g['val_loss'][0]-1):+.2f} %) psnr {psnr:.3f} (ref {g['val_psnr_fine'][0]:.3f})")
