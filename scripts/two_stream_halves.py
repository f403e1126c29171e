#!/usr/bin/env python3
"""Experiment (round 5): the headline batch of 8 as two half-batches of 4 on two HIP streams -- does one half's memory-bound launches
(Winograd transforms, combines, the tail: 7 % of a step) hide under the other half's MFMA launches?  Two contexts (own weights, own
workspace), plan pinned to 8 so the halves compute the bits of the whole batch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coupe.optical_flow_based_deep_video_stabilization_amd as vs
from coupe.optical_flow_based_deep_video_stabilization_amd import runtime, weights as wts

B, H, W = 8, 512, 512
w = wts.synthetic_weights(seed=1, cin=27)
for sc in ("flownetS", "half_a", "half_b"):
    runtime.assign_weights(w, sc)
g = torch.Generator().manual_seed(1000)
feats = torch.rand(B, H, W, 27, generator=g).cuda()
frame = torch.rand(B, H, W, 3, generator=g).cuda()
whole = vs.OriginalSizeStabiliser(B, H, W, 27, H, W)
halves = []
for sc in ("half_a", "half_b"):
    runtime.get_context(sc).set_plan_batch(8)
    halves.append(vs.OriginalSizeStabiliser(B // 2, H, W, 27, H, W, scope=sc))
runtime.get_context("flownetS").set_plan_batch(8)
fa, fb = feats[:4].contiguous(), feats[4:].contiguous()
ra, rb = frame[:4].contiguous(), frame[4:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def step_whole():
    whole(feats, frame)


def step_halves():
    with torch.cuda.stream(s1):
        halves[0](fa, ra)
    with torch.cuda.stream(s2):
        halves[1](fb, rb)


def step_halves_serial():
    halves[0](fa, ra)
    halves[1](fb, rb)


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for r in range(3):
    print(f"round {r}: whole batch {timeit(step_whole):.4f} ms | two halves, two streams {timeit(step_halves):.4f} ms | two halves, one stream {timeit(step_halves_serial):.4f} ms", flush=True)
o = whole(feats, frame)
ow = o[2].clone()
step_halves()
torch.cuda.synchronize()
print("halves == whole (warped):", bool(torch.equal(torch.cat([halves[0].warped, halves[1].warped]), ow)))
