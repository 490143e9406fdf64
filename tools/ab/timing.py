import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0")
pipe, _, _ = bench.build_pipeline(dev, "fp32", "nerf")
data = [torch.from_numpy(x).to(dev) for x in bench.frame_inputs("nerf", 128, 0)]
with torch.no_grad():
    for _ in range(2): pipe(data)
    torch.cuda.synchronize()
    os.environ["SNERF_TIMING"] = "/tmp/timing.txt"
    net = pipe.model_coarse
    raw = net.forward_fused(data[0], data[2], 64, pipe.position_encoder, pipe.direction_encoder)
    torch.cuda.synchronize()
d = np.loadtxt("/tmp/timing.txt", dtype=np.int64)
d = d[d[:, 2] > 0]
for w in (0, 4, 1):
    x = d[d[:, 0] == w]
    t0, t1, t2 = x[:, 2], x[:, 3], x[:, 4]
    period = np.diff(t2)
    print(f"wave {w}: {len(x)} slabs recorded; slab period (barrier exit to barrier exit): median {np.median(period):.0f} mean {period.mean():.0f} min {period.min()} max {period.max()} cycles")
    print("   store+load issue (t1-t0): median %.0f; barrier wait (t2-t1): median %.0f mean %.0f max %d" % (np.median(t1 - t0), np.median(t2 - t1), (t2 - t1).mean(), (t2 - t1).max()))
    print("   compute between barrier exit and next release entry (t0[i+1]-t2[i]): median %.0f mean %.0f" % (np.median(t0[1:] - t2[:-1]), (t0[1:] - t2[:-1]).mean()))
    print("   first 80 periods:", period[:80].tolist())
