import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from coupe.optical_flow_based_deep_video_stabilization_amd import train_step, weights as wts
B, H, W = 8, 512, 512
w = wts.synthetic_weights(seed=1, cin=27, random_bn=False, flow_gain=0.2)
for prepack in (False, True):
    tr = train_step.Trainer(w, B, H, W)
    tr.prepack = prepack
    g = torch.Generator().manual_seed(0)
    feats = torch.rand(B, H, W, 27, generator=g).cuda()
    gt, un = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
    for _ in range(3):
        tr.step(feats, gt, un, lr=1e-4)
    torch.cuda.synchronize()
    # host issue time of ONE step into an empty stream
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); tr.step(feats, gt, un, lr=1e-4); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    print("prepack", prepack, "host issue ms", [round(a * 1e3, 2) for a, b in ts], "issue+drain ms", [round(b * 1e3, 2) for a, b in ts])
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(feats, gt, un, lr=1e-4)
    torch.cuda.synchronize()
    print("  10 steps back to back:", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms/step")
    del tr
