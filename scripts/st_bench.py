#!/usr/bin/env python3
"""Spatial-transformer / warp.py samplers in isolation at BASELINE shapes (configs[2]: batch 32 x 720 x 1280 x 3): kernel time
per launch from the library's dispatch-timestamp events, achieved ALGORITHMIC GB/s and fraction of 8 TB/s.
Algorithmic bytes per output pixel: 12 read (every source pixel is needed about once at unit scale) + 12 written = 24;
bilinear_interp with explicit coordinates reads x and y too: 32.  theta is 24-36 bytes per SAMPLE."""
import argparse
import json
import math
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coupe.optical_flow_based_deep_video_stabilization_amd import runtime, spatial_transformer as st, warp as vwarp   # noqa: E402

PEAK = 8000.0


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    runtime.hbm_profile(1)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    runtime.hbm_profile(0)
    ms = sum(v[0] for v in runtime.hbm_profile_read().values())
    return ms / iters * 1e3


def thetas(B, kind, g):
    """identity | stab (what a stabiliser applies: +-2 degrees, +-3 % scale, +-3 % shift, per sample) | rot30"""
    rows = []
    for _ in range(B):
        if kind == "identity":
            a, s, tx, ty = 0.0, 1.0, 0.0, 0.0
        elif kind == "rot30":
            a, s, tx, ty = math.radians(30), 1.0, 0.0, 0.0
        else:
            r = torch.rand(4, generator=g)
            a, s = math.radians(float(r[0]) * 4 - 2), 1 + (float(r[1]) - 0.5) * 0.06
            tx, ty = (float(r[2]) - 0.5) * 0.06, (float(r[3]) - 0.5) * 0.06
        rows.append([s * math.cos(a), -s * math.sin(a), tx, s * math.sin(a), s * math.cos(a), ty])
    return torch.tensor(rows, dtype=torch.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--shapes", default="32x720x1280,8x512x512,16x1080x1920")
    ap.add_argument("--kinds", default="identity,stab,rot30")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    rows = []
    for sh in args.shapes.split(","):
        B, H, W = (int(v) for v in sh.split("x"))
        g = torch.Generator().manual_seed(1)
        img = torch.rand(B, H, W, 3, generator=g).cuda()
        px = B * H * W
        ref_m = torch.tensor([[(W - 1) / 2, 0, (W - 1) / 2], [0, (H - 1) / 2, (H - 1) / 2], [0, 0, 1]])
        cfg = types.SimpleNamespace(warpType="homography", warpApprox=20, batch_size=B, height=H, width=W, refMtrx=ref_m.cuda())
        for kind in args.kinds.split(","):
            th6 = thetas(B, kind, g)
            th8 = torch.cat([th6, (torch.rand(B, 2, generator=g) - 0.5) * (0.0 if kind == "identity" else 0.02)], 1).cuda()
            th6 = th6.cuda()
            M3 = torch.cat([th8, torch.ones(B, 1, device="cuda")], 1).reshape(B, 3, 3)
            aff, proj = st.AffineTransformer((H, W)), st.ProjectiveTransformer((H, W))
            grid = aff.pixel_grid.reshape(3, -1)
            t23 = th6.reshape(B, 2, 3)          # explicit coordinates for bilinear_interp, elementwise (no library GEMM in this script's trace)
            T = (t23[:, :, 0:1] * grid[0] + t23[:, :, 1:2] * grid[1]) + t23[:, :, 2:3] * grid[2]
            xs, ys = T[:, 0].reshape(-1).contiguous(), T[:, 1].reshape(-1).contiguous()
            r = {"shape": sh, "pixels": px, "theta": kind}
            for name, fn, bpp in (
                    ("AffineTransformer.transform", lambda: aff.transform(img, th6), 24),
                    ("ProjectiveTransformer.transform", lambda: proj.transform(img, th8), 24),
                    ("bilinear_interp (explicit x, y)", lambda: st.bilinear_interp(img, xs, ys, (H, W)), 32),
                    ("warp.transformImage", lambda: vwarp.transformImage(cfg, img, M3), 24)):
                us = timeit(fn, args.iters)
                gbs = px * bpp / us * 1e-3
                r[name] = {"us": round(us, 2), "alg_bytes_per_px": bpp, "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK, 4)}
                print(f"{sh:>14} {kind:<9} {name:<34} {us:9.2f} us  {gbs:8.1f} GB/s  {gbs / PEAK:.3f}", file=sys.stderr, flush=True)
            rows.append(r)
            del xs, ys, T
    s = json.dumps(rows, indent=1)
    if args.out:
        open(args.out, "w").write(s)
    print(s)


if __name__ == "__main__":
    main()
