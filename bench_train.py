#!/usr/bin/env python3
"""Training-step throughput: one `sess.run(optim_main)` of the reference = train-mode forward + loss_main
+ backward through the network + Adam, on synthetic data.  One JSON line.

    python bench_train.py [--batch 8 --height 512 --width 512 --steps 10 --warmup 3]

FLOP accounting: forward = netspec's 2*MACs; backward = input gradient + weight gradient of every conv-like layer ~ 2x forward
(the first layer has no input gradient; the full-resolution head runs literally here, on the materialised upsampled tensor)."""
import argparse
import json
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (what RCCL needs on this driver); before any HIP init
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA


def cpu_baseline(weights, H, W, seconds_budget=25.0):
    """One optimiser step of the reference's training graph (main:176-335) as the oracle restates it, timed on the host cores: train-mode
    forward (BatchNorm on batch statistics), loss_main, torch autograd backward through the restated graph, Adam -- on a bounded sample
    (batch 1 of the same frame size; the GPU number is per sample too).  kind = "port": TensorFlow 1.10 cannot exist on this box."""
    import numpy as np
    from oracle import vstab_oracle as vo
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    threads = max(1, min(n, 16))
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    feats, gt, un = torch.rand(1, H, W, 27, generator=g), torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
    Wt = {k: torch.tensor(v, dtype=torch.float32, requires_grad=("moving_" not in k)) for k, v in weights.items()}
    m = {k: torch.zeros_like(v) for k, v in Wt.items() if v.requires_grad}
    v2 = {k: torch.zeros_like(v) for k, v in Wt.items() if v.requires_grad}

    def step(t):
        flows = vo.flownetS_pyramid(feats, Wt, torch.float32, is_train=True)
        loss = vo.loss_main(flows, gt, un, torch.float32)
        loss.backward()
        with torch.no_grad():
            for k, p in Wt.items():
                if p.grad is None:
                    continue
                m[k].mul_(0.9).add_(p.grad, alpha=0.1)
                v2[k].mul_(0.999).addcmul_(p.grad, p.grad, value=0.001)
                p.sub_(1e-4 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t) * m[k] / (v2[k].sqrt() + 1e-8))
                p.grad = None
        return float(loss.detach())

    t0 = time.perf_counter()
    step(1)
    warm = time.perf_counter() - t0
    times = []
    while len(times) < 3 and sum(times) + warm < seconds_budget:
        t0 = time.perf_counter()
        step(len(times) + 2)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times or [warm]))
    return {"value": round(1.0 / med, 3), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{len(times) or 1} timed optimiser steps on ONE {H}x{W}x27 sample (train-mode forward, loss_main, autograd backward, Adam), "
                      "torch-CPU fp32 restatement of the TF training graph"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the three extra steps with an event pair around every conv-family call")
    ap.add_argument("--phases", action="store_true", help="also time forward / loss+backward / Adam separately (adds syncs)")
    ap.add_argument("--gpus", type=int, default=1, help="data-parallel replicas to start when not already under a launcher")
    args = ap.parse_args()
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "vstab_launch", os.path.join(ROOT, "coupe", "optical_flow_based_deep_video_stabilization_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    rc = launch.maybe_self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus,
                                  force=os.environ.get("VSTAB_FORCE_DIST") == "1")     # child job; nothing here touches the GPU
    if rc is not None:
        raise SystemExit(rc)
    real_stdout = launch.claim_stdout()       # fd 1 -> stderr from here on (RCCL prints its banner to stdout); the JSON line goes to the real one
    if not torch.cuda.is_available():
        raise SystemExit("bench_train.py needs a GPU")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("VSTAB_FORCE_DIST") == "1"      # data-parallel replicas, gradients averaged over RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from coupe.optical_flow_based_deep_video_stabilization_amd import netspec, train_step, weights as wts

    B, H, W = args.batch, args.height, args.width
    w = wts.synthetic_weights(seed=1, cin=27, random_bn=False, flow_gain=0.2)
    tr = train_step.Trainer(w, B, H, W)
    g = torch.Generator().manual_seed(rank)                # every replica trains on its own shard
    feats = torch.rand(B, H, W, 27, generator=g).cuda()
    gt, un = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
    for _ in range(args.warmup):
        loss = tr.step(feats, gt, un, lr=1e-4)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step(feats, gt, un, lr=1e-4)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    phases = None
    if args.phases:
        tf = tb = ta = 0.0
        for _ in range(args.steps):
            torch.cuda.synchronize(); a = time.perf_counter()
            tr.forward(feats); torch.cuda.synchronize(); b = time.perf_counter()
            tr.loss_and_backward(gt, un); torch.cuda.synchronize(); c = time.perf_counter()
            tr.adam(1e-4); torch.cuda.synchronize(); d = time.perf_counter()
            tf += b - a; tb += c - b; ta += d - c
        phases = {"forward_ms": round(tf / args.steps * 1e3, 3), "loss_backward_ms": round(tb / args.steps * 1e3, 3),
                  "adam_ms": round(ta / args.steps * 1e3, 3)}
    # ---- roofline of the step's MFMA-bound calls: three more steps with an event pair (on the launch stream) around every conv-family
    # library call; a call's time includes its helper launches (operand packing, split-K combine, pixel tables), so the fractions are
    # lower bounds on the MFMA kernels' own.  flops = 2*MAC of the layer each call computes (forward, input gradient and filter gradient
    # of a layer cost the same; Winograd-form calls are priced as the direct 3x3 convolution)
    roofline = None
    if not args.no_roofline:
        tr.profile_calls(True)
        for _ in range(3):
            tr.step(feats, gt, un, lr=1e-4)
        fam = tr.profile_read()
        if rank == 0 and tr.calls:          # per call, in issue order (a step's calls repeat three times): the layer-level view
            n1 = len(tr.calls) // 3
            print(f"{'#':>3} {'family':<24}{'GFLOP':>9}{'ms':>9}{'TFLOP/s':>9}{'frac':>7}", file=sys.stderr)
            for i in range(n1):
                kind, fl = tr.calls[i][0], tr.calls[i][1]
                ms = sum(tr.calls[i + r * n1][2].elapsed_time(tr.calls[i + r * n1][3]) for r in range(3)) / 3
                tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                print(f"{i:>3} {kind:<24}{fl / 1e9:>9.2f}{ms:>9.4f}{tf:>9.1f}{tf / MFMA_F32_PEAK_TFLOPS:>7.3f}", file=sys.stderr)
        tr.profile_calls(False)
        if fam:
            rows = {k: {"ms_per_step": round(ms / 3, 4), "calls_per_step": n // 3, "gflop_per_step": round(fl / 3 / 1e9, 2),
                        "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
                    for k, (ms, fl, n) in fam.items() if ms > 0}
            dom = max(rows, key=lambda k: rows[k]["ms_per_step"])
            tot_ms = sum(ms for ms, _, _ in fam.values()) / 3
            tot_fl = sum(fl for _, fl, _ in fam.values()) / 3
            kern = {"conv_wgrad": "wgrad_mfma_kernel<128> / <64> (+ pixel table, split-K combine)",
                    "conv_dgrad": "conv_mfma_kernel (input gradients and DeConv2dLayer forwards; + operand packing, split-K combine)",
                    "conv_forward": "conv_mfma_kernel / conv_rowwin_kernel (+ operand packing, split-K combine)",
                    "conv3x3_winograd": "winograd transforms + 16-position conv_mfma_kernel launch", "conv3x3_winograd_wgrad": "winograd transforms + batched wgrad_mfma_kernel"}
            roofline = {"bound": "mfma", "achieved": rows[dom]["achieved"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rows[dom]["frac"],
                        "traffic": None, "kernel": kern.get(dom, dom), "family": dom, "families": rows,
                        "all_mfma_calls": {"ms_per_step": round(tot_ms, 3), "gflop_per_step": round(tot_fl / 1e9, 1),
                                           "achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                           "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)},
                        "note": "event pairs around whole library calls (helper launches included): lower bounds on the MFMA kernels' own fractions; "
                                "Winograd-form calls priced as direct 3x3 convolutions"}
    gf = netspec.gflop_per_sample(H, W, 27)
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    print(json.dumps({
        "metric": f"training samples/sec @{H}x{W}", "value": round(B * world / dt, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "scaling": "weak", "vs_baseline": None,
        "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "dtype": "f32",
        "data": "synthetic (uniform [0,1) frames, seeded He-normal weights)", "final_loss": float(loss),
        "approx_tflops": round(3.0 * gf * B * world / dt / 1e3, 2), "phases": phases, "roofline": roofline,
        "cpu_baseline": (cpu_baseline(w, H, W) if (world == 1 and not args.no_cpu_baseline) else None),
        "config": {"workload": f"batch={B} per GPU {H}x{W}x27: train-mode forward + loss_main + backward + Adam (38.7 M parameters)"
                               + ("; gradients averaged with one RCCL all-reduce of a 155 MB bucket" if use_dist else "")}}), file=real_stdout, flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
