#!/usr/bin/env python3
"""What the benchmark's flows look like (random-weight network on uniform-noise frames): magnitude and pixel-to-pixel variation
of the output-resolution flow that drives the warp's gathers.  usage: flow_stats.py B H W"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coupe.optical_flow_based_deep_video_stabilization_amd as vs   # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4])
vs.initialize_global_variables(seed=1, cin=27)
g = torch.Generator().manual_seed(1000)
feats = torch.rand(B, H, W, 27, generator=g).cuda()
frame = torch.rand(B, H, W, 3, generator=g).cuda()
flows, outflow, warped = vs.stabilise_originalsize(feats, frame)
f = outflow
if len(sys.argv) > 4:
    f.cpu().numpy().tofile(sys.argv[4])       # raw fp32 [B,H,W,2] for tools/warp_bench
print(f"outflow {tuple(f.shape)}: mean |f| {f.abs().mean():.3f}  max |f| {f.abs().max():.3f}  "
      f"mean |df/dx| {(f[:, :, 1:] - f[:, :, :-1]).abs().mean():.4f}  mean |df/dy| {(f[:, 1:] - f[:, :-1]).abs().mean():.4f}  "
      f"p99 |df/dx| {(f[:, :, 1:] - f[:, :, :-1]).abs().flatten()[::97].quantile(0.99):.4f}")

# source bounding box of output tiles (what an LDS-staged source window would have to hold)
ys = torch.arange(H, device="cuda").view(1, H, 1).float()
xs = torch.arange(W, device="cuda").view(1, 1, W).float()
sx = (xs + f[..., 0]).clamp(0, W - 1)
sy = (ys + f[..., 1]).clamp(0, H - 1)
for th, tw in ((8, 32), (16, 16), (16, 64), (32, 32), (8, 128), (4, 256), (16, 32)):
    hh, ww = H // th * th, W // tw * tw
    def tiles(t):
        return t[:, :hh, :ww].reshape(B, hh // th, th, ww // tw, tw).permute(0, 1, 3, 2, 4).reshape(B, hh // th, ww // tw, th * tw)
    bw = tiles(sx).amax(-1).floor() - tiles(sx).amin(-1).floor() + 2
    bh = tiles(sy).amax(-1).floor() - tiles(sy).amin(-1).floor() + 2
    area = (bw * bh).flatten()
    q = torch.quantile(area[::7].float(), torch.tensor([0.5, 0.9, 0.99], device="cuda"))
    print(f"tile {th:3d}x{tw:<3d} ({th * tw:5d} px): bbox w mean {bw.mean():6.1f} h mean {bh.mean():6.1f}  area/px mean {area.mean() / (th * tw):5.2f} "
          f"p50 {q[0] / (th * tw):5.2f} p90 {q[1] / (th * tw):5.2f} p99 {q[2] / (th * tw):5.2f}  max area {area.max():.0f} px")
