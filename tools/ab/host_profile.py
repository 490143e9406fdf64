"""Where the host time of a training step goes (cProfile over 100 steps at 2048 rays)."""
import cProfile, pstats, os, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from smpl_nerf_amd.trainer import DataParallelTrainer
dev = torch.device("cuda:0")
pipe, _, models = bench.build_pipeline(dev, "fp32", "nerf")
for m in models:
    m.train()
tr = DataParallelTrainer(pipe, models, lr=3e-5)
data = [torch.from_numpy(x).to(dev) for x in bench.frame_inputs("nerf", 128, 0)]
batch = [t[:int(sys.argv[1]) if len(sys.argv) > 1 else 2048].contiguous() for t in data]
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    tr.step(batch)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38)
print(s.getvalue()[:7000])
