"""Gradient agreement of two arithmetically equivalent builds of the training step (3x3 stages in Winograd / direct form)
at the benchmark size: through the loss (tf_warp's cell boundaries amplify 1e-6 flow differences) or, with `fixed`, from
fixed upstream flow gradients (rounding only)."""
import sys, torch
sys.path.insert(0, ".")
from coupe.optical_flow_based_deep_video_stabilization_amd import train_step, weights as wts
B, H, W = 8, 512, 512
g = torch.Generator().manual_seed(0)
feats = torch.rand(B, H, W, 27, generator=g).cuda()
gt, un = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
G = []; P = []
for wino in (3.0e9, 1e30):
    tr = train_step.Trainer(wts.synthetic_weights(seed=1, cin=27, random_bn=False, flow_gain=0.2), B, H, W)
    tr.wino_min_flops = wino
    flows = tr.forward(feats)
    if len(sys.argv) > 1 and sys.argv[1] == "fixed":        # fixed upstream flow gradients: no tf_warp cell boundaries in the way
        gr = torch.Generator().manual_seed(5)
        tr.backward_from_flow_grads({k: torch.randn(v.shape, generator=gr).cuda() for k, v in flows.items() if k != "flow"})
        l = 0.0
    else:
        l = tr.loss_and_backward(gt, un)
    G.append({k: v.clone() for k, v in tr.g.items()})
    tr.adam(1e-4, 0.9)
    P.append({k: v.clone() for k, v in tr.p.items()})
    print("loss", float(l))
    del tr
tot_flip = 0
for k in G[0]:
    a, b = G[0][k].double(), G[1][k].double()
    rel = float((a - b).norm() / (b.norm() + 1e-300))
    flips = int(((a * b) < 0).sum())
    tot_flip += flips
    dp = float((P[0][k].double() - P[1][k].double()).abs().max())
    print(f"{k:28s} |g| {float(b.norm()):.3e} absmax {float(b.abs().max()):.2e} median {float(b.abs().median()):.2e} rel diff {rel:.2e} sign flips {flips}/{a.numel()} max dparam {dp:.2e}")
print("total flips", tot_flip)
