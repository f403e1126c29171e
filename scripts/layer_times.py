#!/usr/bin/env python3
"""Per-launch-slot time of the forward's 15 conv-like launches (dispatch timestamps taken by the library).
usage: layer_times.py [B H W]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coupe.optical_flow_based_deep_video_stabilization_amd as vs
from coupe.optical_flow_based_deep_video_stabilization_amd import runtime, weights as wts

B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 512, 512)
vs.assign_weights(wts.synthetic_weights(seed=1, cin=27))
ctx = runtime.get_context()
feats = torch.rand(B, H, W, 27).cuda()
for _ in range(5):
    vs.flownetS_pyramid(feats, B)
torch.cuda.synchronize()
ctx.profile(True)
for _ in range(10):
    vs.flownetS_pyramid(feats, B)
torch.cuda.synchronize()
ms, fl, n = ctx.profile_read()
dfl = ctx.profile_read_direct()
names = ctx.profile_kernel_names()
slots = ["conv1", "conv2", "conv3", "conv3_1", "conv4", "conv4_1", "conv5", "conv5_1", "conv6", "conv6_1", "deconv5", "deconv4", "deconv3", "deconv2", "pf2_taps"]
for s, m, f, d, k in zip(slots, ms, fl, dfl, names):
    print(f"{s:9s} {m / n * 1e3:8.1f} us  issued {f / m / 1e9 if m else 0:6.1f} TF/s  direct {d / m / 1e9 if m else 0:6.1f} TF/s  {k}")
print("sum", sum(ms) / n * 1e3, "us")
