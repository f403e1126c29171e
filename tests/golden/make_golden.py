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


if __name__ == "__main__":
    for n in CASES:
        make(n)
    make_warp()
