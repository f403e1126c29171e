#!/usr/bin/env python3
"""HBM-side kernels of the path in isolation (W1 tf_warp, G1 flow glue, the fused launch) at BASELINE shapes:
time per launch from torch events over back-to-back launches, achieved algorithmic GB/s and fraction of 8 TB/s.
Algorithmic bytes: warp 32 B/px (12 img + 8 flow + 12 out); glue 16 B/px (8 src flow + 8 out);
fused glue+warp with the output flow written 40 B/px (8 src + 8 outflow + 12 + 12), without 32 B/px."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coupe.optical_flow_based_deep_video_stabilization_amd as vs   # noqa: E402
from coupe.optical_flow_based_deep_video_stabilization_amd import runtime   # noqa: E402

PEAK = 8000.0


def timeit(fn, iters):
    """Average KERNEL time (us) from the library's dispatch-timestamp events (not host time: at 512x512 a Python call is
    slower than the kernel)."""
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


def insitu_study(B, H, W, iters):
    """Why the fused glue + warp launch takes 22.7 us inside a step and 16.6 us here (B=8 512x512): the same launch timed (library
    dispatch timestamps) with its operands in the cache states a step leaves them in.  Between timed launches:
      warm        nothing (this script's default: frame and predict_flow2 were read by the previous iteration)
      cold        1 GiB written to another buffer (L2 and the 256 MB Infinity Cache hold neither operand)
      frame cold  the same flush, then predict_flow2 re-written by a copy kernel (as pf2_tile_kernel leaves it: fresh, dirty in L2) --
                  what a step looks like: the frame was last touched a whole step (~700 MB of workspace traffic) ago
      pf2 cold    the flush, then the FRAME re-read by a copy (frame resident, predict_flow2 from HBM)"""
    g = torch.Generator().manual_seed(1)
    img = torch.rand(B, H, W, 3, generator=g).cuda()
    lo = torch.randn(B, 2, max(H // 16, 2), max(W // 16, 2), generator=g) * 6
    pf2 = torch.nn.functional.interpolate(lo, size=(H - 2, W - 2), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous().cuda()
    pf2_src, sink = pf2.clone(), torch.empty_like(img)
    junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    modes = {"warm": lambda: None,
             "cold": lambda: junk.fill_(1.0),
             "frame cold, predict_flow2 fresh": lambda: (junk.fill_(1.0), pf2.copy_(pf2_src)),
             "predict_flow2 cold, frame resident": lambda: (junk.fill_(1.0), sink.copy_(img))}
    out = {}
    for name, prep in modes.items():
        for _ in range(3):
            prep(); vs.flow_glue_warp(pf2, img, H, W)
        torch.cuda.synchronize()
        runtime.hbm_profile(1)
        for _ in range(iters):
            prep()
            vs.flow_glue_warp(pf2, img, H, W)
        torch.cuda.synchronize()
        runtime.hbm_profile(0)
        hp = runtime.hbm_profile_read()["flow_glue_warp"]
        us = hp[0] / hp[1] * 1e3
        gbs = B * H * W * 40 / us * 1e-3
        out[name] = {"us": round(us, 2), "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK, 4)}
        print(f"{B}x{H}x{W} glue+warp fused, {name:<36} {us:8.2f} us  {gbs:8.1f} GB/s  {gbs / PEAK:.3f}", file=sys.stderr, flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--flows", default="smooth,random")
    ap.add_argument("--shapes", default="8x512x512,16x1080x1920,32x720x1280,1x384x512")
    ap.add_argument("--insitu-study", action="store_true", help="only the cache-state study of the fused launch at the first shape")
    args = ap.parse_args()
    if args.insitu_study:
        B, H, W = (int(v) for v in args.shapes.split(",")[0].split("x"))
        print(json.dumps({"shape": args.shapes.split(",")[0], "insitu_study": insitu_study(B, H, W, args.iters)}))
        return
    rows = []
    for sh in args.shapes.split(","):
        B, H, W = (int(v) for v in sh.split("x"))
        g = torch.Generator().manual_seed(1)
        img = torch.rand(B, H, W, 3, generator=g).cuda()
        px = B * H * W
        for kind in args.flows.split(","):
            if kind == "random":        # every pixel's flow independent: neighbouring pixels gather from unrelated cache lines
                pf2 = (torch.randn(B, H - 2, W - 2, 2, generator=g) * 3).cuda()
            else:                       # "smooth": a coarse random field upsampled 16x, like a real (or a network's) flow
                lo = torch.randn(B, 2, max(H // 16, 2), max(W // 16, 2), generator=g) * 6
                pf2 = torch.nn.functional.interpolate(lo, size=(H - 2, W - 2), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous().cuda()
            flow = vs.flow_to_output_res(pf2, H, W, H, W)
            r = {"shape": sh, "pixels": px, "flow": kind}
            for name, fn, bpp in (
                    ("warp", lambda: vs.tf_warp(img, flow, H, W), 32),
                    ("glue", lambda: vs.flow_to_output_res(pf2, H, W, H, W), 16),
                    ("glue+warp fused (flow written)", lambda: vs.flow_glue_warp(pf2, img, H, W), 40),
                    ("glue+warp fused (no flow)", lambda: vs.flow_glue_warp(pf2, img, H, W, want_outflow=False), 32)):
                us = timeit(fn, args.iters)
                gbs = px * bpp / us * 1e-3
                r[name] = {"us": round(us, 2), "alg_bytes_per_px": bpp, "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK, 4)}
                print(f"{sh:>14} {kind:<7} {name:<32} {us:9.2f} us  {gbs:8.1f} GB/s  {gbs / PEAK:.3f}", file=sys.stderr, flush=True)
            rows.append(r)
        # G2 (main:806): the unstable frame resized to the flow grid.  (a) from a contiguous 3-channel tensor: 12 B read + 12 B written
        # per pixel; (b) in place from channels 24:27 of the 27-channel stack, as stabilise_native does: every 128-byte line of the
        # stack is touched, so the bytes that move are 108 + 12 per pixel.  Stream events over back-to-back launches.
        def ev_time(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3
        us = ev_time(lambda: vs.resize_images(img, (H - 2, W - 2)))
        gbs = px * 24 / us * 1e-3
        rows.append({"shape": sh, "resize3 contiguous (G2)": {"us": round(us, 2), "bytes_per_px": 24, "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK, 4)}})
        print(f"{sh:>14} {'-':<7} {'resize of a 3-channel frame (G2)':<32} {us:9.2f} us  {gbs:8.1f} GB/s  {gbs / PEAK:.3f}", file=sys.stderr, flush=True)
        if B * H * W * 27 * 4 < 8e9:
            stack = torch.rand(B, H, W, 27, generator=g).cuda()
            us = ev_time(lambda: vs.resize_images_slice3(stack, 24, (H - 2, W - 2)))
            gbs = px * 120 / us * 1e-3
            rows.append({"shape": sh, "resize 24:27 of the stack in place (G2)": {"us": round(us, 2), "bytes_per_px": 120, "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK, 4)}})
            print(f"{sh:>14} {'-':<7} {'resize 24:27 of the stack':<32} {us:9.2f} us  {gbs:8.1f} GB/s  {gbs / PEAK:.3f} (120 B/px moved)", file=sys.stderr, flush=True)
            del stack
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
