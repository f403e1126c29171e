#!/usr/bin/env python3
"""Real-video mode: the reference's per-frame loop (main:535-630) through clip_driver.ClipStabiliser -- frame i
depends on the stabilised frames before it, so one clip is sequential and the figure of merit is ms per frame
(the reference prints exactly that, main:627); several clips run in lockstep as the batch dimension.

    python bench_stream.py [--clips 1] [--frames 200] [--out-height 720 --out-width 1280]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=1)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--out-height", type=int, default=720)
    ap.add_argument("--out-width", type=int, default=1280)
    ap.add_argument("--net-height", type=int, default=384)
    ap.add_argument("--net-width", type=int, default=512)
    ap.add_argument("--issue-burst", type=int, default=16, help="frames issued back to back in the host-issue-cost leg before the GPU drains")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_stream.py needs a GPU")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    import coupe.optical_flow_based_deep_video_stabilization_amd as vs
    from coupe.optical_flow_based_deep_video_stabilization_amd.clip_driver import ClipStabiliser

    vs.initialize_global_variables(seed=1, cin=27)
    n, oh, ow = args.clips, args.out_height, args.out_width
    drv = ClipStabiliser(oh, ow, n_clips=n, net_hw=(args.net_height, args.net_width))
    g = torch.Generator().manual_seed(7)
    frames = [torch.randint(0, 256, (n, oh, ow, 3), dtype=torch.uint8, generator=g).cuda() for _ in range(8)]
    outs = [torch.empty_like(frames[0]) for _ in range(4)]          # a writer would consume frame i before frame i + 4 is produced
    for i in range(args.warmup):
        drv.step(frames[i % 8], out=outs[i % 4])
    torch.cuda.synchronize()
    # (a) throughput with the host running ahead (frames already decoded): launch all, sync once
    t0 = time.perf_counter()
    for i in range(args.frames):
        drv.step(frames[i % 8], out=outs[i % 4])
    torch.cuda.synchronize()
    t_async = (time.perf_counter() - t0) / args.frames
    # (b) latency of one frame with a sync after each (frame handed to a writer before the next is read)
    t0 = time.perf_counter()
    for i in range(args.frames):
        drv.step(frames[i % 8], out=outs[i % 4])
        torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / args.frames
    # (c) host-side cost of issuing one step, WITHOUT back-pressure: bursts of at most `--issue-burst` frames into an empty stream
    # (a frame is ~30 kernel launches: a burst stays far below the depth of the stream's queue), the clock stopped while the GPU drains.
    # Rounds 1-4 issued all frames into a queue that fills, so their figure mixed issue cost with waiting for the GPU.
    t_issue_sum, issued = 0.0, 0
    while issued < args.frames:
        nb = min(args.issue_burst, args.frames - issued)
        t0 = time.perf_counter()
        for i in range(nb):
            drv.step(frames[(issued + i) % 8], out=outs[(issued + i) % 4])
        t_issue_sum += time.perf_counter() - t0
        torch.cuda.synchronize()
        issued += nb
    t_issue = t_issue_sum / args.frames
    print(json.dumps({
        "metric": f"stabilised frames/sec, autoregressive clip driver, {n} clip(s) in lockstep, {oh}x{ow} output",
        "value": round(n / t_async, 2), "unit": "frames/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f32",
        "ms_per_step_async": round(t_async * 1e3, 4), "ms_per_step_synced": round(t_sync * 1e3, 4),
        "ms_host_issue_per_step": round(t_issue * 1e3, 4), "host_issue_burst": args.issue_burst,
        "host_calls_per_step": "1 (vstab_clip_step, every buffer pre-allocated)",
        "data": "synthetic uint8 frames, seeded He-normal weights",
        "config": {"workload": f"{n} clip(s) x {args.frames} frames, net {args.net_height}x{args.net_width}x27, "
                               f"uint8 resize + history ring + forward + flow glue + tf_warp + quantise per frame"}}), flush=True)


if __name__ == "__main__":
    main()
