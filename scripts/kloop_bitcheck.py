#!/usr/bin/env python3
"""sha256 of every output of one forward (B=8 512x512x27, seeded) -- run once per library build (VSTAB_LIB) and compare the lines:
the assembly K loop keeps the C++ loop's MFMA order per accumulator, so the two builds must print the same digests."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coupe.optical_flow_based_deep_video_stabilization_amd as vs   # noqa: E402

B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 512, 512)))
vs.initialize_global_variables(seed=1, cin=27)
g = torch.Generator().manual_seed(1000)
feats = torch.rand(B, H, W, 27, generator=g).cuda()
outs = vs.flownetS_pyramid(feats, B)
torch.cuda.synchronize()
def walk(name, o):
    if isinstance(o, dict):
        for k in sorted(o):
            walk(f"{name}.{k}", o[k])
    elif isinstance(o, (tuple, list)):
        for i, v in enumerate(o):
            walk(f"{name}[{i}]", v)
    elif torch.is_tensor(o):
        print(name, tuple(o.shape), hashlib.sha256(o.detach().cpu().numpy().tobytes()).hexdigest()[:24])


walk("out", outs)

# the filter-gradient kernel's loop (both tile widths, split-K and not)
from coupe.optical_flow_based_deep_video_stabilization_amd import training   # noqa: E402

for name, Bn, Hi, Wi, cin, cout, k, s_, p_ in (("wgrad conv3", 8, 128, 128, 128, 256, 5, 2, 2), ("wgrad conv1-like", 2, 64, 64, 28, 64, 7, 2, 3),
                                              ("wgrad conv6_1", 8, 8, 8, 1024, 1024, 3, 1, 1)):
    gg = torch.Generator().manual_seed(7)
    Ho, Wo = (Hi + 2 * p_ - k) // s_ + 1, (Wi + 2 * p_ - k) // s_ + 1
    x = torch.randn(Bn, Hi, Wi, cin, generator=gg).cuda()
    go = torch.randn(Bn, Ho, Wo, cout, generator=gg).cuda()
    dW, db = training.conv_wgrad(x, go, k, s_, p_)
    torch.cuda.synchronize()
    walk("out." + name.replace(" ", "_"), [dW, db])
