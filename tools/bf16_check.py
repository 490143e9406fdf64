"""Dev check: split-bf16 MLP modes vs the fp32 kernel and the reference frame; timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smpl_nerf_amd import synthetic as syn, _lib
from smpl_nerf_amd.nets import RenderRayNet
from smpl_nerf_amd.ops import PositionalEncoder
from smpl_nerf_amd.pipelines import NerfPipeline, PipelineArgs as Args
dev = torch.device("cuda:0")
pc, pf = syn.make_scene_nets(101)
def net(p):
    m = RenderRayNet(8, 256, 60, 24, skips=[4]); m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}); return m.to(dev)
mc, mf = net(pc), net(pf)
pipe = NerfPipeline(mc, mf, Args(), PositionalEncoder(10, 0), PositionalEncoder(4, 0))
data = [torch.from_numpy(a).to(dev) for a in syn.frame_batch(128, 128, seed=7)]
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g5_nerf_pipeline.npz")))
res = {}
with torch.no_grad():
    for prec in ("fp32", "bf16x6", "bf16x3", "f16x3"):
        mc.precision = mf.precision = prec
        out = pipe(data); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): out = pipe(data)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        res[prec] = [o.cpu().numpy() for o in out]
        e_ref_c = np.abs(res[prec][0] - g["rgb_nf14"]).max(); e_ref_f = np.abs(res[prec][1] - g["rgb_fine_nf14"]).max()
        print(f"{prec}: {dt*1e3:.2f} ms/frame  {16384*256/dt:.3e} samples/s | vs reference frame: coarse {e_ref_c:.2e} fine {e_ref_f:.2e}", end="")
        if prec != "fp32":
            print(f" | vs fp32 kernel: coarse {np.abs(res[prec][0]-res['fp32'][0]).max():.2e} fine {np.abs(res[prec][1]-res['fp32'][1]).max():.2e}")
        else: print()
