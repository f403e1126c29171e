#!/usr/bin/env python3
"""BASELINE.json configs[3]: an F-frame 1080p synthetic clip sharded across the GPUs of one node,
reassembled over RCCL: every finished micro-batch is all-gathered into its place of the full clip while the next one is computed
(`distributed.SequenceGatherer`).

Teacher-forced: every frame's 27-channel input stack is given (synthetic), so frames are independent
samples and the clip shards by contiguous blocks (the autoregressive real-video mode does
not shard by frame -- that is clip_driver.ClipStabiliser, batched over clips instead).  Strong scaling:
the clip is fixed, ranks split it.

    python bench_clip.py --frames 1000 [--gpus N --height 1080 --width 1920 --micro-batch 8]     (starts its own N ranks)
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 bench_clip.py --gpus N --frames 1000
"""
import argparse
import json
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (what RCCL needs on this driver); before any HIP init
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cin", type=int, default=27)
    ap.add_argument("--micro-batch", type=int, default=8)
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start when not already under a launcher")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="gloo: ranks may share one GPU and frames are reassembled through host memory -- a rehearsal of the N>1 control "
                         "flow (ragged shards, overlapped gather) on a one-GPU box; its numbers mean nothing")
    ap.add_argument("--check", action="store_true",
                    help="every frame gets its own seeded input (all of them generated on every rank: use small frames), and rank 0 also "
                         "computes the UNSHARDED sequence with the same micro-batch and compares the reassembled clip with it bit for bit "
                         "(`sharded_equals_unsharded` in the JSON line; a mismatch is exit code 3)")
    ap.add_argument("--no-plan-pin", action="store_true", help="do not pin the launch plan to the micro-batch (ragged tails then differ in the last bits)")
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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench_clip.py needs a GPU")
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("VSTAB_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import coupe.optical_flow_based_deep_video_stabilization_amd as vs
    from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, runtime, distributed as vdist

    F_, H, W, Cin, MB = args.frames, args.height, args.width, args.cin, args.micro_batch
    vs.initialize_global_variables(seed=1, cin=Cin)
    # a shard's last micro-batch is ragged (125 frames per rank in micro-batches of 8 end with one of 5): the launch plan is pinned to
    # the full micro-batch, so every frame comes out bit-identical to the unsharded run whatever it is batched with (vstab_set_plan_batch)
    if not args.no_plan_pin:
        runtime.get_context().set_plan_batch(MB)
    lo, hi = vdist.shard_range(F_, rank, world)
    n_local = hi - lo
    if args.check:
        g = torch.Generator().manual_seed(4242)                      # the SAME clip on every rank; a rank works on frames [lo, hi)
        all_feats = torch.rand(F_, H, W, Cin, generator=g)
        all_frame = torch.rand(F_, H, W, 3, generator=g)
        clip_feats, clip_frame = all_feats[lo:hi].cuda(), all_frame[lo:hi].cuda()
        feats, frame = clip_feats[:MB].contiguous(), clip_frame[:MB].contiguous()
    else:
        g = torch.Generator().manual_seed(2000 + rank)
        feats = torch.rand(MB, H, W, Cin, generator=g).cuda()            # one synthetic micro-batch, reused
        frame = torch.rand(MB, H, W, 3, generator=g).cuda()
    L = _lib.lib()
    host = args.backend == "gloo"
    seq = vdist.SequenceGatherer(F_, (H, W, 3), torch.uint8, torch.device("cpu") if host else torch.device("cuda", local_rank)) if use_dist else None
    shard = torch.empty((n_local, H, W, 3), dtype=torch.uint8, device="cuda")
    common = seq.common if seq is not None else n_local

    def run_shard():
        """Micro-batches of this rank's shard through the HIP path; every finished micro-batch of the part all shards have in
        common is all-gathered into its place of the full clip at once, beside the next micro-batch's kernels."""
        def compute(b0, bc):
            if args.check:
                _, _, warped = vs.stabilise_originalsize(clip_feats[b0:b0 + bc].contiguous(), clip_frame[b0:b0 + bc].contiguous())
            else:
                _, _, warped = vs.stabilise_originalsize(feats[:bc], frame[:bc])
            _lib.check(L.vstab_quantise_output(warped.data_ptr(), bc * H * W, shard[b0:b0 + bc].data_ptr(), runtime.stream_ptr()))

        for b0 in range(0, common, MB):
            bc = min(MB, common - b0)
            compute(b0, bc)
            if seq is not None:
                seq.submit(shard[b0:b0 + bc].cpu() if host else shard[b0:b0 + bc], b0)
        if n_local > common:                    # a block partition gives some ranks one item more: it goes through finish()
            compute(common, n_local - common)

    # warm-up: one micro-batch (kernels loaded, workspaces allocated)
    vs.stabilise_originalsize(feats[:min(MB, max(n_local, 1))], frame[:min(MB, max(n_local, 1))])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_shard()
    t_issue = time.perf_counter() - t0
    full = seq.finish(shard[common:].cpu() if host else shard[common:]) if seq is not None else shard
    torch.cuda.synchronize()
    t_compute = time.perf_counter() - t0          # compute with the reassembly overlapped; what is left of it shows in `elapsed`
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed, t_compute], dtype=torch.float64, device="cpu" if host else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_compute = float(t[0]), float(t[1])
    assert full.shape[0] == F_
    same = None
    if args.check and rank == 0:          # the unsharded sequence: one rank, the same micro-batch size, the clip from frame 0
        ref = torch.empty((F_, H, W, 3), dtype=torch.uint8, device="cuda")
        for b0 in range(0, F_, MB):
            bc = min(MB, F_ - b0)
            _, _, warped = vs.stabilise_originalsize(all_feats[b0:b0 + bc].cuda(), all_frame[b0:b0 + bc].cuda())
            _lib.check(L.vstab_quantise_output(warped.data_ptr(), bc * H * W, ref[b0:b0 + bc].data_ptr(), runtime.stream_ptr()))
        torch.cuda.synchronize()
        same = bool(torch.equal(ref.cpu(), full.cpu()))
    if rank == 0:
        print(json.dumps({
            "metric": f"stabilised frames/sec, {F_}-frame {H}x{W} clip sharded over {world} GPU(s), all-gather reassembly",
            "value": round(F_ / elapsed, 2), "unit": "frames/s", "n_gpus": world, "higher_is_better": True,
            "scaling": "strong", "dtype": "f32", "data": "synthetic, teacher-forced history",
            "seconds_total": round(elapsed, 4), "seconds_compute_and_overlapped_gather": round(t_compute, 4),
            "seconds_host_issue": round(t_issue, 4), "sharded_equals_unsharded": same, "plan_batch": 0 if args.no_plan_pin else MB,
            "config": {"workload": f"{F_} frames {H}x{W}x{Cin}, micro-batch {MB}, uint8 frames all-gathered per micro-batch, overlapped with compute",
                       "gathered_bytes": int(F_) * H * W * 3}}), file=real_stdout, flush=True)
    if use_dist:
        dist.destroy_process_group()
    if same is False:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
