#!/usr/bin/env python3
"""scratch: time the input-gradient launches of the stride-2 encoder layers (B=8, 512x512 input)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coupe.optical_flow_based_deep_video_stabilization_amd import training
shapes = [("conv2", 8, 256, 256, 64, 128, 5, 2, 2), ("conv3", 8, 128, 128, 128, 256, 5, 2, 2), ("conv4", 8, 64, 64, 256, 512, 3, 2, 1),
          ("conv5", 8, 32, 32, 512, 512, 3, 2, 1), ("conv6", 8, 16, 16, 512, 1024, 3, 2, 1)]
for name, B, Hi, Wi, cin, cout, k, s, p in shapes:
    Ho, Wo = (Hi + 2 * p - k) // s + 1, (Wi + 2 * p - k) // s + 1
    g = torch.randn(B, Ho, Wo, cout, device="cuda") * 0.1
    Wt = torch.randn(k, k, cin, cout, device="cuda") * 0.05
    for _ in range(3): training.conv_dgrad(g, Wt, s, p, (Hi, Wi))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): training.conv_dgrad(g, Wt, s, p, (Hi, Wi))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * k * k * cin * cout * B * Ho * Wo
    print(f"{name:8s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  frac {fl/ms/1e9/157.3:.3f}", flush=True)
