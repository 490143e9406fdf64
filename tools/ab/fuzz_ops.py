"""Randomised sweep of the stand-alone operators against the numpy oracle: the hierarchical sampler (bit-exact: indices, samples,
merged depths, points), compositing (1e-6), searchsorted (bit-exact, every dtype of the reference's dispatch, both sides, row
broadcast), positional encoding.  Random row counts / lengths / weight patterns (zeros, spikes, ties).  Not part of the suite.

    python tools/ab/fuzz_ops.py [cases] [seed]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from oracle import nerf_oracle as O
from smpl_nerf_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
N = lambda t: t.detach().cpu().numpy()
bad = 0


def weights_pattern(B, n):
    kind = rng.integers(0, 5)
    w = rng.uniform(0, 1, (B, n)).astype(np.float32)
    if kind == 1:
        w[:] = 0                                         # nothing hit: uniform pdf from the 1e-5 floor
    elif kind == 2:
        w[:] = 0
        w[np.arange(B), rng.integers(0, n, B)] = 1      # one spike
    elif kind == 3:
        w = (w > 0.7).astype(np.float32)                # plateaus: ties in the cdf
    elif kind == 4:
        w *= np.float32(1e-6)                           # below the floor
    return w


for case in range(cases):
    what = ["sampler", "composite", "searchsorted", "posenc", "raygen", "encoded_forward", "warp_rows"][case % 7]
    try:
        if what == "sampler":
            B, Nc, Nf = int(rng.choice([1, 3, 64, 257, 1000])), int(rng.choice([3, 4, 5, 17, 64, 100, 255, 1024])), int(rng.choice([1, 2, 7, 64, 128, 333, 1024]))
            z = np.sort(rng.uniform(1, 4, (B, Nc)).astype(np.float32), -1)
            if rng.integers(0, 3) == 0:
                z[:, Nc // 2] = z[:, Nc // 2 - 1]                  # a repeated depth
            w = weights_pattern(B, Nc)
            o = rng.normal(0, 1, (B, 3)).astype(np.float32)
            d = rng.normal(0, 1, (B, 3)).astype(np.float32)
            u = N(ops.uniform_u(Nf, dev))
            r = ops.hierarchical_samples(T(o), T(d), T(z), T(w), Nf, want_inds=True, want_samples=True)
            zmid = (np.float32(0.5) * (z[:, 1:] + z[:, :-1])).astype(np.float32)
            det = O.sample_pdf_detail(zmid, w[:, 1:-1], Nf, u=u)
            zf, pts = O.fine_sampling(o, d, z, w, Nf, u=u)
            ok = np.array_equal(N(r["inds"]), det["inds"]) and np.array_equal(N(r["z_samples"]), det["samples"]) and \
                np.array_equal(N(r["z_fine"]), zf) and np.array_equal(N(r["pts"]), pts)
            desc = f"sampler B {B} Nc {Nc} Nf {Nf}"
        elif what == "composite":
            B, n = int(rng.choice([1, 5, 64, 300, 1025])), int(rng.choice([1, 2, 3, 64, 100, 192, 256, 500]))
            raw = rng.normal(0, 2, (B, n, 4)).astype(np.float32)
            raw[..., 3] *= rng.choice([1, 20, 200])
            z = np.sort(rng.uniform(1, 4, (B, n)).astype(np.float32), -1)
            d = rng.normal(0, 1, (B, 3)).astype(np.float32)
            wb = int(rng.integers(0, 2))
            noise = rng.normal(0, 1, (B, n)).astype(np.float32) if rng.integers(0, 2) else None
            rgb, wt, al = ops.composite(T(raw), T(z), T(d), bool(wb), noise=None if noise is None else T(noise), want_weights=True, want_alpha=True)
            er, ew, ea = O.raw2outputs(raw, z, np.broadcast_to(d[:, None, :], (B, n, 3)), wb, noise)
            err = max(np.abs(N(rgb) - er).max(), np.abs(N(wt) - ew).max(), np.abs(N(al) - ea).max())
            ok = err <= (2e-6 if n <= 256 else 5e-6)
            desc = f"composite B {B} N {n} wb {wb} noise {noise is not None}: max err {err:.2e}"
        elif what == "searchsorted":
            dt = rng.choice([np.float32, np.float64, np.int32, np.int64, np.int16, np.int8, np.uint8])
            ra, rv = int(rng.choice([1, 1, 7, 100])), int(rng.choice([1, 7, 100]))
            if ra != rv and ra != 1 and rv != 1:
                rv = ra
            na, nv = int(rng.choice([1, 2, 33, 128, 1000, 5000])), int(rng.choice([1, 5, 64, 129, 2000]))
            if np.issubdtype(dt, np.floating):
                a = np.sort(rng.normal(0, 1, (ra, na)).astype(dt), -1)
                v = rng.normal(0, 1.2, (rv, nv)).astype(dt)
                v[:, ::3] = rng.choice(a[0], size=v[:, ::3].shape)         # exact hits: the two sides differ there
            else:
                hi = min(50, np.iinfo(dt).max)
                a = np.sort(rng.integers(0, hi, (ra, na)).astype(dt), -1)      # many ties
                v = rng.integers(0, hi, (rv, nv)).astype(dt)
            side = str(rng.choice(["left", "right"]))
            out = ops.searchsorted(T(a), T(v), side=side)
            ok = np.array_equal(N(out), O.searchsorted(a, v, side))
            desc = f"searchsorted {np.dtype(dt).name} a {a.shape} v {v.shape} {side}"
        elif what == "raygen":
            from smpl_nerf_amd.raygen import RayGenerator
            H, W = int(rng.choice([1, 7, 64, 128, 200])), int(rng.choice([1, 9, 64, 128, 333]))
            F, Nc = int(rng.choice([1, 3, 10])), int(rng.choice([1, 2, 3, 64, 100]))
            near, far = float(rng.uniform(0.1, 2.0)), float(rng.uniform(2.5, 8.0))
            ang = float(rng.uniform(0.3, 2.0))
            poses = np.tile(np.eye(4), (F, 1, 1))
            for f in range(F):
                q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
                poses[f, :3, :3] = q
                poses[f, :3, 3] = rng.normal(0, 2, 3)
            gen = RayGenerator(poses, H, W, ang, near, far, Nc, dev)
            Bq = int(rng.choice([1, 5, 257, 4096]))
            idx = rng.integers(0, F * H * W, Bq)
            jit = rng.uniform(0, 1, Bq)
            got = [N(t) for t in gen.batch(T(idx.astype(np.int64)), T(jit))[:4]]
            focal = .5 * W / np.tan(.5 * ang)
            exp = [[], [], [], []]
            fr, pix = idx // (H * W), idx % (H * W)
            for f in np.unique(fr):
                o_all, d_all = O.get_rays(H, W, focal, poses[f])
                sel = fr == f
                r = O.coarse_sampling(o_all.reshape(-1, 3)[pix[sel]], d_all.reshape(-1, 3)[pix[sel]], near, far, Nc, jit[sel])
                for k in range(4):
                    exp[k].append((np.nonzero(sel)[0], r[k]))
            ok = True
            for k in range(4):
                full = np.empty_like(got[k])
                for where, val in exp[k]:
                    full[where] = val
                ok = ok and np.array_equal(full, got[k])
            desc = f"raygen {F} frames {H}x{W} Nc {Nc} B {Bq}"
        elif what == "encoded_forward":
            # RenderRayNet.forward(x) on already-encoded rows (models/render_ray_net.py:42-61) - any positions_dim / directions_dim /
            # additional_input_dim, the three precisions - against the CPU torch restatement
            from oracle import torch_cpu_path as TP
            from smpl_nerf_amd.nets import RenderRayNet
            depth, width = int(rng.integers(1, 11)), int(rng.choice([16, 64, 100, 128, 256, 256]))
            pd, dd, ad = int(rng.choice([60, 63, 30, 7, 100])), int(rng.choice([24, 27, 12, 3])), int(rng.choice([0, 0, 5, 69, 300]))
            skips = sorted(set(int(v) for v in rng.integers(0, depth, rng.integers(0, 3)))) if depth > 1 else []
            use_dir = int(rng.integers(0, 4) != 0)
            prec = str(rng.choice(["fp32", "bf16x6", "f16x3"])) if width == 256 else "fp32"
            if os.environ.get("FUZZ_WIDE"):      # FUZZ_WIDE=1: --netwidth above 256 (the kernels of 320 / 384 / 448 / 512 features, fp32)
                width, prec = [257, 300, 320, 352, 384, 400, 448, 500, 512][case % 9], "fp32"
            n = int(rng.choice([1, 17, 128, 1000, 20001]))
            torch.manual_seed(4000 + case)
            net = RenderRayNet(depth, width, pd, dd, ad, skips=list(skips), use_directional_input=use_dir)
            P = {k: v.detach().clone() for k, v in net.state_dict().items()}
            net.precision = prec
            net = net.to(dev).eval()
            x = torch.randn(n, pd + ad + dd)
            with torch.no_grad():
                got = net(x.to(dev)).cpu()
                ref = TP.render_ray_net(P, x, n_layers=depth, positions_dim=pd, directions_dim=dd, additional_input_dim=ad, skips=tuple(skips),
                                        use_directional_input=use_dir)
            err = float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))
            ok = err <= (2e-5 if prec == "fp32" else 2e-4) and bool(torch.isfinite(got).all())
            desc = f"encoded_forward {prec} depth {depth} width {width} skips {skips} dims {pd}/{ad}/{dd} dir {use_dir} n {n}: max err {err:.2e}"
        elif what == "warp_rows":
            # WarpFieldNet.forward(rows) (models/warp_field_net.py:17-21) with its backward (rows and parameters) against torch
            from oracle import torch_cpu_path as TP
            from smpl_nerf_amd.nets import WarpFieldNet
            width = int(rng.choice([7, 64, 100, 128, 200, 256]))
            pd, qd = int(rng.choice([60, 63, 3, 33])), int(rng.choice([40, 2, 24]))
            n = int(rng.choice([1, 19, 128, 5000, 70001]))
            torch.manual_seed(5000 + case)
            mw = WarpFieldNet(8, width, pd, qd)
            P = {k: v.detach().clone().requires_grad_(True) for k, v in mw.state_dict().items()}
            mw = mw.to(dev).train()
            x, w = torch.randn(n, pd + qd), torch.randn(n, 3)
            xg, xc = x.clone().to(dev).requires_grad_(True), x.clone().requires_grad_(True)
            out = mw(xg)
            (out * w.to(dev)).sum().backward()
            ref = TP.warp_field_net(P, xc)
            (ref * w).sum().backward()
            errs = [float((out.detach().cpu() - ref.detach()).abs().max()) / max(1.0, float(ref.abs().max())),
                    float((xg.grad.cpu() - xc.grad).norm()) / max(float(xc.grad.norm()), 1e-12)]
            errs += [float((p.grad.cpu() - P[k].grad).norm()) / max(float(P[k].grad.norm()), 1e-12) for k, p in mw.named_parameters()]
            err = max(errs)
            # the output is continuous in the pre-activations; the gradients are not: a sample within rounding of a ReLU kink
            # takes the other side in one of the two fp32 evaluations (checked against fp64: sometimes the GPU's, sometimes
            # torch's) and moves a sum over n samples by ~1/sqrt(n) of itself
            ok = errs[0] <= 2e-5 and err <= (2e-5 if n < 1000 else 1e-2)
            names = ["out", "d rows"] + [k for k, _ in mw.named_parameters()]
            desc = f"warp_rows width {width} dims {pd}+{qd} n {n}: max err {err:.2e} ({names[errs.index(err)]})"
        else:
            L, ident = int(rng.choice([0, 1, 4, 10, 16])), int(rng.integers(0, 2))
            if L == 0 and not ident:
                ident = 1
            shape = tuple(int(v) for v in rng.choice([1, 3, 17, 200], int(rng.integers(1, 3)))) + (int(rng.choice([2, 3])),)
            x = rng.normal(0, 1.5, shape).astype(np.float32)
            got = N(ops.PositionalEncoder(L, ident).encode(T(x)))
            ref = O.PositionalEncoder(L, ident).encode(x)
            err = float(np.abs(got - ref).max()) if got.shape == ref.shape else float("inf")
            ok = err <= 1e-6
            desc = f"posenc L {L} ident {ident} x {shape}: max err {err:.2e}"
        bad += not ok
        print(("ok  " if ok else "BAD ") + desc, flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print(f"EXC {what}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{cases - bad} of {cases} cases agree")
