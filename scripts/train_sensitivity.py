"""How far two arithmetically equivalent builds of the training step drift apart: the same 13 steps with the 3x3 stages in
Winograd form and in direct form, per-step loss_main side by side (rounding differences only; see profiles/README.md)."""
import sys
import torch
sys.path.insert(0, ".")
from coupe.optical_flow_based_deep_video_stabilization_amd import train_step, weights as wts

B, H, W = 8, 512, 512
g = torch.Generator().manual_seed(0)
feats = torch.rand(B, H, W, 27, generator=g).cuda()
gt, un = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
runs = []
for wino in (3.0e9, 1e30, 3.0e9):
    tr = train_step.Trainer(wts.synthetic_weights(seed=1, cin=27, random_bn=False, flow_gain=0.2), B, H, W)
    tr.wino_min_flops = wino
    runs.append([float(tr.step(feats, gt, un, lr=1e-4)) for _ in range(13)])
    del tr
for i in range(13):
    print(i, " ".join(f"{r[i]:.9f}" for r in runs), f"wino-direct {runs[0][i] - runs[1][i]:+.2e}  repeat {runs[0][i] - runs[2][i]:+.2e}")
