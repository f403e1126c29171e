"""Generates tests/golden/*.npz with the CPU oracle (oracle/vstab_oracle.py, fp64 arbiter).
The reference itself cannot run here (TensorFlow 1.10 / tensorlayer are not installable,
SURVEY.md 8c), so these are vectors of the restatement, not of TensorFlow: parity is
"unpinned" at that boundary.  Weights are NOT stored (155 MB): they are regenerated from
the seed by coupe...weights.synthetic_weights.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coupe.optical_flow_based_deep_video_stabilization_amd import weights as wts  # noqa: E402
from oracle import vstab_oracle as vo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (B, H, W, Cin, out_h, out_w, weight seed, random_bn, flow_gain)
    "net_48x64_c27": (1, 48, 64, 27, 48, 64, 11, True, 2.0),
    "net_70x90_c6": (1, 70, 90, 6, 96, 120, 12, True, 2.0),
}


def make(name):
    B, H, W, Cin, oh, ow, seed, rbn, gain = CASES[name]
    rng = np.random.default_rng(seed + 100)
    feats = rng.random((B, H, W, Cin), dtype=np.float32)
    frame = rng.random((B, oh, ow, 3), dtype=np.float32)
    w = wts.synthetic_weights(seed=seed, cin=Cin, random_bn=rbn, flow_gain=gain)
    flows, outflow, warped = vo.stabilise_originalsize(feats, frame, w, torch.float64)
    out = {"feats": feats, "frame": frame, "meta": np.array([B, H, W, Cin, oh, ow, seed, int(rbn)], np.int64),
           "flow_gain": np.float64(gain), "outflow": outflow.numpy().astype(np.float32),
           "warped": warped.numpy().astype(np.float32)}
    for k in vo.FLOW_KEYS:
        out[k] = flows[k].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def make_warp():
    rng = np.random.default_rng(7)
    img = rng.random((2, 37, 53, 3), dtype=np.float32)
    flow = ((rng.random((2, 37, 53, 2), dtype=np.float32) - 0.5) * 40).astype(np.float32)
    flow[0, 0, :8, 0] = np.array([-1.0, -0.5, 0.0, 0.5, 51.0, 52.0, 1e6, -1e6], np.float32)   # x edge cases at xx=0..7
    flow[0, 0, :8, 1] = 0.0
    flow[1, :6, 0, 1] = np.array([-1.0, -1.5, 35.0, 36.0, 30.5, -0.25], np.float32)
    flow[1, :6, 0, 0] = 0.25
    out = vo.tf_warp(torch.from_numpy(img), torch.from_numpy(flow), 37, 53, torch.float32).numpy()
    np.savez_compressed(os.path.join(HERE, "warp_37x53.npz"), img=img, flow=flow, warped=out)
    print("warp_37x53", out.shape)


def make_widening():
    """Small vectors for the rows built around the path: cv2-style resizes, homography evaluator, training loss."""
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    out = {"img_u8": img, "resize_u8_48x64": vo.cv_resize_u8(img, 48, 64)}
    imgf = rng.random((46, 62, 3), dtype=np.float32)
    out["img_f32"] = imgf
    out["resize_f32_48x64"] = vo.cv_resize_f32(imgf, 48, 64)
    # a homography field with 30 % gross outliers
    H, W = 48, 64
    Ht = np.array([[1.02, 0.01, 1.5], [-0.015, 0.99, -1.0], [1e-5, -2e-5, 1.0]])
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    d = Ht[2, 0] * xs + Ht[2, 1] * ys + Ht[2, 2]
    flow = np.stack([xs - (Ht[0, 0] * xs + Ht[0, 1] * ys + Ht[0, 2]) / d, ys - (Ht[1, 0] * xs + Ht[1, 1] * ys + Ht[1, 2]) / d], -1).astype(np.float32)
    bad = rng.random((H, W)) < 0.3
    flow[bad] += rng.uniform(-25, 25, (int(bad.sum()), 2)).astype(np.float32)
    M, n_in = vo.homography_fit(flow, K=64, seed=9, thresh=3.0, refine=2)
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    out.update(homo_flow=flow, homo_H=M, homo_inliers=np.int64(n_in), homo_frame=frame,
               homo_warped=vo.cv_warp_perspective_u8(frame, M, H, W))
    # loss_main on seeded flows
    B, Hh, Ww = 2, 64, 96
    sizes = [(1, 2), (2, 3), (4, 6), (8, 12), (62, 94)]
    g = torch.Generator().manual_seed(5)
    gt, un = torch.rand(B, Hh, Ww, 3, generator=g), torch.rand(B, Hh, Ww, 3, generator=g)
    flows = {k: (torch.randn(B, h, w, 2, generator=g) * 1.5).requires_grad_(True) for k, (h, w) in zip(vo.LOSS_LEVELS, sizes)}
    loss = vo.loss_main(flows, gt, un)
    loss.backward()
    out.update(loss_gt=gt.numpy(), loss_un=un.numpy(), loss_value=np.float64(loss.detach()))
    for k in vo.LOSS_LEVELS:
        out["loss_flow_" + k] = flows[k].detach().numpy()
        out["loss_grad_" + k] = flows[k].grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "widening_small.npz"), **out)
    print("widening_small", {k: getattr(v, "shape", ()) for k, v in out.items()})


if __name__ == "__main__":
    for n in CASES:
        make(n)
    make_warp()
    make_widening()
