#!/usr/bin/env python3
"""Headline benchmark: stabilised frame-pairs/sec @512x512 (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch of synthetic input, per rank:
  feats [8,512,512,27] -> flownetS_pyramid (5 flows) -> flow to output resolution
  (main:497-498) -> tf_warp of the [8,512,512,3] frame (main:514).
Inputs and weights are resident in HBM before the timed region.  With N > 1 ranks every
rank processes its own batch (weak scaling, samples are independent) and the warped frames
of each step are all-gathered over RCCL on a side stream, overlapped with the next step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the keys below); a per-layer table goes to stderr.
"""
import argparse
import json
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (what RCCL needs on this driver); before any HIP init
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA = vector fp32 peak
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def load_launch_module():
    """coupe/.../launch.py loaded by path: the parent of a self-launched job must not import the package
    (which loads the HIP library) or touch the GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "vstab_launch", os.path.join(ROOT, "coupe", "optical_flow_based_deep_video_stabilization_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def host_cpu_share():
    """Cores this process may really use: cgroup quota if one is set, else the affinity mask,
    capped at 16 (a 1-GPU box's CPU share; more threads than that only oversubscribe)."""
    if os.environ.get("VSTAB_CPU_THREADS"):
        return max(1, int(os.environ["VSTAB_CPU_THREADS"]))
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def newest_steady_trace():
    """The newest committed steady-state rocprofv3 summary of the headline command (profiles/rocprof_rNN*_cfg1_steady.md, by name)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "rocprof_r*_cfg1_steady.md")))
    return "profiles/" + os.path.basename(c[-1]) if c else "profiles/ (no steady-state trace committed)"


def cpu_baseline(feats, frame, weights, seconds_budget=20.0):
    """The CPU restatement (oracle, torch fp32 on all host cores) timed on a bounded sample
    of the same workload: `feats` / `frame` are the FIRST samples of the batch the GPU was timed on (host copies).
    kind = "port": TensorFlow 1.10 cannot exist on this box.  Returns (record, outputs of one pass)."""
    from oracle import vstab_oracle as vo
    threads = host_cpu_share()
    torch.set_num_threads(threads)
    nb, H, W, Cin = feats.shape
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = vo.stabilise_originalsize(feats, frame, weights, torch.float32)       # warm-up; its outputs feed flow_err
        warm = time.perf_counter() - t0
        times = []
        while len(times) < 5 and (sum(times) + warm) < seconds_budget:
            t0 = time.perf_counter()
            vo.stabilise_originalsize(feats, frame, weights, torch.float32)
            times.append(time.perf_counter() - t0)
        if not times:
            times = [warm]
    med = float(np.median(times))
    return {"value": nb / med, "unit": "frame-pairs/s", "cores": threads, "cores_box": os.cpu_count(), "cpu_model": cpu_model(), "kind": "port",
            "cores_note": "cores = threads the timed pass used (this process's CPU share: cgroup quota / affinity, capped at 16); "
                          "cores_box = logical CPUs the host reports",
            "sample": f"{len(times)} timed passes of the first {nb} samples of the timed GPU batch ({H}x{W}x{Cin}: "
                      f"network + flow glue + warp), torch-CPU fp32 restatement of the TF graph"}, ref


EPS32 = 1.1920929e-07
FLOW_ERR_TOL = 1e-3          # BASELINE.json north_star: flows and warped frames within 1e-3 max-abs on fp32


def flow_error(gpu_out, feats, frame, weights, ref32=None, want_fp64=True):
    """The metric's second half ("+ max-abs flow err vs TF CPU", BASELINE.json): the outputs of the LAST timed step on the GPU
    against the CPU restatement run on the same first `nb` samples of the timed batch.  `vs_fp32` = the torch-CPU fp32
    restatement (the pass cpu_baseline times; its own rounding is in the figure), `vs_fp64` = the same graph in fp64 (the
    arbiter the parity tests use).  Warped frames are compared away from tf_warp's discontinuity lines (x = -1, W-1; y = -1, H-1),
    where a flow that differs by rounding legitimately picks another branch."""
    from oracle import vstab_oracle as vo
    torch.set_num_threads(host_cpu_share())
    flows_g, outflow_g, warped_g = gpu_out
    nb = feats.shape[0]
    oh, ow = frame.shape[1], frame.shape[2]
    g_flows = {k: flows_g[k][:nb].double().cpu() for k in vo.FLOW_KEYS}
    g_out = outflow_g[:nb].double().cpu() if outflow_g is not None else None
    g_warp = warped_g[:nb].double().cpu()

    def against(ref):
        rf, ro, rw = ref
        lv = {k: float((g_flows[k] - rf[k].double()).abs().max()) for k in vo.FLOW_KEYS}
        if g_out is not None:
            lv["outflow"] = float((g_out - ro.double()).abs().max())
        mask = vo.warp_discontinuity_mask(ro, oh, ow)
        if g_out is not None:
            mask &= vo.warp_discontinuity_mask(g_out, oh, ow)
        dw = (g_warp - rw.double()).abs().amax(dim=3)
        return {"max_abs": {k: float(f"{v:.3e}") for k, v in lv.items()},
                "warped_max_abs_masked": float(f"{float(dw[mask].max()):.3e}"),
                "warped_pixels_masked_out": int((~mask).sum()),
                "max_abs_flow": {k: round(float(rf[k].abs().max()), 2) for k in vo.FLOW_KEYS},
                # the error is a RELATIVE one: in fp32 epsilons of the level's largest flow
                "eps_of_max_flow": {k: round(lv[k] / (EPS32 * max(float(rf[k].abs().max()), 1e-30)), 2) for k in vo.FLOW_KEYS}}

    with torch.no_grad():
        if ref32 is None:
            ref32 = vo.stabilise_originalsize(feats, frame, weights, torch.float32)
        r32 = against(ref32)
        r64 = against(vo.stabilise_originalsize(feats, frame, weights, torch.float64)) if want_fp64 else None
    worst = max(list(r32["max_abs"].values()) + [r32["warped_max_abs_masked"]])
    res = {"max_abs": r32["max_abs"], "warped_max_abs_masked": r32["warped_max_abs_masked"],
           "warped_pixels_masked_out": r32["warped_pixels_masked_out"], "max_abs_flow": r32["max_abs_flow"],
           "eps_of_max_flow": r32["eps_of_max_flow"],
           "vs": "torch-CPU fp32 restatement of the TF graph on the same inputs (parity unpinned: TensorFlow 1.10 cannot run here)",
           "samples": nb, "of_step": "last timed step", "tol": FLOW_ERR_TOL, "worst": worst, "within_tol": bool(worst <= FLOW_ERR_TOL)}
    if r64 is not None:
        w64 = max(list(r64["max_abs"].values()) + [r64["warped_max_abs_masked"]])
        res["vs_fp64"] = {"max_abs": r64["max_abs"], "warped_max_abs_masked": r64["warped_max_abs_masked"], "worst": w64,
                          "eps_of_max_flow": r64["eps_of_max_flow"],
                          "within_tol": bool(w64 <= FLOW_ERR_TOL), "vs": "the same restatement in fp64 (the parity tests' arbiter)"}
    return res


def secondary_rows(vs, runtime, log):
    """Short forms of bench_clip.py and bench_train.py on this GPU; every row is best-effort (a failure is reported, not raised)."""
    rows = {}
    try:        # configs[3]'s per-GPU work: a 1080p clip shard in micro-batches of 8, frames quantised to uint8 as the writer does
        from coupe.optical_flow_based_deep_video_stabilization_amd import _lib
        F_, MB, H, W = 24, 8, 1080, 1920
        g = torch.Generator().manual_seed(5)
        feats = torch.rand(MB, H, W, 27, generator=g).cuda()
        frame = torch.rand(MB, H, W, 3, generator=g).cuda()
        shard = torch.empty((F_, H, W, 3), dtype=torch.uint8, device="cuda")
        stab = vs.OriginalSizeStabiliser(MB, H, W, 27, H, W, want_outflow=False)

        def run():
            for b0 in range(0, F_, MB):
                _, _, warped = stab(feats, frame)
                _lib.check(_lib.lib().vstab_quantise_output(warped.data_ptr(), MB * H * W, shard[b0:b0 + MB].data_ptr(), runtime.stream_ptr()))
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows["clip_shard_1080p"] = {"frames_per_s": round(F_ / dt, 1), "frames": F_, "micro_batch": MB, "ms_per_micro_batch": round(dt / (F_ // MB) * 1e3, 2),
                                    "what": "BASELINE configs[3] per-GPU work (bench_clip.py without the all-gather): network + glue + warp + uint8 quantise at 1080x1920"}
        del stab, feats, frame, shard
        log(f"secondary: 1080p clip shard {F_ / dt:.1f} frames/s")
    except Exception as e:      # noqa: BLE001
        rows["clip_shard_1080p"] = {"error": repr(e)[:200]}
    try:        # one optimiser step of the reference's training graph (main:176-335), what bench_train.py times
        from coupe.optical_flow_based_deep_video_stabilization_amd import train_step, weights as wts
        torch.cuda.empty_cache()
        B, H, W = 8, 512, 512
        tr = train_step.Trainer(wts.synthetic_weights(seed=1, cin=27, random_bn=False, flow_gain=0.2), B, H, W)
        g = torch.Generator().manual_seed(0)
        feats = torch.rand(B, H, W, 27, generator=g).cuda()
        gt, un = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
        for _ in range(2):
            tr.step(feats, gt, un, lr=1e-4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            loss = tr.step(feats, gt, un, lr=1e-4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        rows["train_step_512"] = {"ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(B / dt, 1), "batch": B, "final_loss": float(loss),
                                  "what": "train-mode forward + loss_main + backward + Adam at 8 x 512x512x27 (bench_train.py)"}
        log(f"secondary: training step {dt * 1e3:.2f} ms")
        del tr
    except Exception as e:      # noqa: BLE001
        rows["train_step_512"] = {"error": repr(e)[:200]}
    torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)        # 1.2 s of timed region at the headline shape: a sustained figure, not a burst
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--cin", type=int, default=27)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-flow-err", action="store_true",
                    help="skip the flow_err block (the last timed step's flows and warped frame against the CPU restatement on the same inputs)")
    ap.add_argument("--err-samples", type=int, default=2, help="samples of the timed batch the CPU restatement is run on")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of warped frames (N>1)")
    ap.add_argument("--gather-schedule", choices=("allgather", "direct"), default="allgather",
                    help="reassembly over RCCL: one all-gather per step, or world-1 point-to-point pushes per rank (all xGMI links at once)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend; gloo (frames gathered through host memory, ranks may share one GPU) only rehearses the "
                         "N>1 control flow on a one-GPU box -- its numbers mean nothing")
    ap.add_argument("--gather-every", type=int, default=4,
                    help="uint8 reassembly: stage this many steps' frames and gather them with ONE collective (fewer, larger collectives; "
                         "the same bytes, all of them inside the timed region)")
    ap.add_argument("--gather-fp32", action="store_true",
                    help="all-gather the fp32 warped frames instead of the uint8 video frames the reference writes (main:630)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket conv launches with HIP events")
    ap.add_argument("--event-every", type=int, default=8,
                    help="record the per-launch HIP events on every n-th step of the timed region (an event-carrying step costs ~2 %% more at the "
                         "headline shape, ~15 %% more for one sample; at least two steps carry them)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph (torch.cuda.CUDAGraph) and replay it; single GPU, no kernel events")
    ap.add_argument("--roctx", action="store_true", help="run every layer inside a named roctx range (rocprofv3 --marker-trace)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short secondary measurements after the timed region (BASELINE configs[3]-shaped sharded clip, one "
                         "training step; single GPU only)")
    ap.add_argument("--alloc-per-step", action="store_true",
                    help="call stabilise_originalsize (two library calls, seven output allocations per step) instead of the pre-allocated "
                         "one-call OriginalSizeStabiliser")
    ap.add_argument("--st-warp", choices=("none", "affine", "projective", "homography"), default="none",
                    help="BASELINE configs[2]'s spatial_transformer leg: also warp the stabilised frame with AffineTransformer / "
                         "ProjectiveTransformer (spatial_transformer.py:400-452, 539-608) or warp.transformImage (warp.py:46-86) inside the step")
    ap.add_argument("--plan-flags", type=int, default=0,
                    help="vstab_set_plan_flags bits for A/B runs: 1 = few-row layers on the tiled kernel + combine launch (no weight-stream "
                         "kernel: the round-3 schedule), 2 = refinement levels as four launches, 4 = predict_flow2's gather and the glue + warp "
                         "as two launches (the round-4 tail)")
    ap.add_argument("--plan-batch", type=int, default=0,
                    help="vstab_set_plan_batch: pin the arithmetic-changing plan decisions to this batch (0 = plan for --batch itself)")
    ap.add_argument("--vgg16", action="store_true",
                    help="BASELINE config 5: also run the VGG16 trunk (preprocess + 13 conv + 5 pool) on the warped frames")
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks ourselves as a CHILD job, before anything touches the GPU
    # (VSTAB_FORCE_DIST=1 rehearses the same launcher + RCCL path with however many ranks --gpus names, also 1)
    rc = load_launch_module().maybe_self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus,
                                                force=os.environ.get("VSTAB_FORCE_DIST") == "1")
    if rc is not None:
        raise SystemExit(rc)
    real_stdout = load_launch_module().claim_stdout()      # fd 1 -> stderr from here on: RCCL's banner must not sit beside the JSON line
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())       # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("VSTAB_FORCE_DIST") == "1"      # rehearse the RCCL path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import coupe.optical_flow_based_deep_video_stabilization_amd as vs
    from coupe.optical_flow_based_deep_video_stabilization_amd import runtime, netspec
    from coupe.optical_flow_based_deep_video_stabilization_amd import distributed as vdist

    B, H, W, Cin = args.batch, args.height, args.width, args.cin
    weights = vs.initialize_global_variables(seed=1, cin=Cin)
    g = torch.Generator().manual_seed(1000 + rank)
    feats = torch.rand(B, H, W, Cin, generator=g).cuda()
    frame = torch.rand(B, H, W, 3, generator=g).cuda()
    ctx = runtime.get_context()
    if args.plan_flags:
        ctx.set_plan_flags(args.plan_flags)
    if args.plan_batch:
        ctx.set_plan_batch(args.plan_batch)
    if args.roctx:
        runtime.trace_ranges(True)

    gather = None            # fp32 frames: one collective per step, straight from the step's output tensor
    grouped = None           # uint8 video frames (what the reference writes, main:625,630): `gather_every` steps per collective
    G = max(1, args.gather_every) if not args.gather_fp32 else 1       # steps per collective
    gdev = torch.device("cpu") if args.backend == "gloo" else torch.device("cuda", local_rank)
    host = args.backend == "gloo"
    if (world > 1 or force_dist) and not args.no_gather:
        if args.gather_fp32:
            gather = vdist.FrameGatherer((B, H, W, 3), world, gdev, dtype=torch.float32, schedule=args.gather_schedule)
        else:
            # staging buffers and the gatherers of the short groups the warm-up and the timed region end with are allocated HERE:
            # nothing allocates between the opening barrier and the closing synchronise
            grouped = vdist.StepGroupGatherer(G, (B, H, W, 3), world, torch.device("cuda", local_rank), gdev, dtype=torch.uint8,
                                              schedule=args.gather_schedule, tail_steps=(args.warmup, args.steps),
                                              to_comm=(lambda t: t.cpu()) if host else None)
            from coupe.optical_flow_based_deep_video_stabilization_amd import _lib

    def submit_frames(warped):
        """Hand one step's frames to the reassembly (overlapped with the next steps)."""
        if grouped is not None:          # np.uint8(cvtColor(warped*255)) as the reference's writer does, into the group's staging buffer
            dst = grouped.stage()
            _lib.check(_lib.lib().vstab_quantise_output(warped.data_ptr(), B * H * W, dst.data_ptr(), runtime.stream_ptr()))
            grouped.commit()
        else:
            gather.submit(warped.cpu() if host else warped)

    def flush_and_drain():
        if grouped is not None:
            grouped.flush()
        if gather is not None:
            gather.drain()

    vgg = None
    if args.vgg16:
        from coupe.optical_flow_based_deep_video_stabilization_amd import vgg16 as vvgg
        vgg = vvgg.Vgg16(seed=7, reuse_outputs=True)

    # the spatial-transformer leg (--st-warp: inside the step; otherwise a few launches after the timed region for roofline_hbm.other_rows)
    from coupe.optical_flow_based_deep_video_stabilization_amd import spatial_transformer as vst, warp as vwarp
    import math
    import types
    st_rows = []
    for _ in range(B):            # what a stabiliser applies: +-2 degrees, +-3 % scale, +-3 % shift, per sample
        r = torch.rand(4, generator=g)
        a, sc = math.radians(float(r[0]) * 4 - 2), 1 + (float(r[1]) - 0.5) * 0.06
        st_rows.append([sc * math.cos(a), -sc * math.sin(a), (float(r[2]) - 0.5) * 0.06, sc * math.sin(a), sc * math.cos(a), (float(r[3]) - 0.5) * 0.06])
    th6 = torch.tensor(st_rows, dtype=torch.float32).cuda()
    th8 = torch.cat([th6, (torch.rand(B, 2, generator=g).cuda() - 0.5) * 0.02], 1)
    st_aff, st_proj = vst.AffineTransformer((H, W)), vst.ProjectiveTransformer((H, W))
    st_cfg = types.SimpleNamespace(warpType="homography", warpApprox=20, batch_size=B, height=H, width=W,
                                   refMtrx=torch.tensor([[(W - 1) / 2, 0, (W - 1) / 2], [0, (H - 1) / 2, (H - 1) / 2], [0, 0, 1.0]]).cuda())
    st_M = torch.cat([th8, torch.ones(B, 1, device="cuda")], 1).reshape(B, 3, 3)
    st_fns = {"affine": lambda im: st_aff.transform(im, th6), "projective": lambda im: st_proj.transform(im, th8),
              "homography": lambda im: vwarp.transformImage(st_cfg, im, st_M)}

    nstep = [0]
    gather_on = [True]          # the second timed region of an N > 1 run (the same steps without the reassembly) clears it

    dbg = [] if os.environ.get("VSTAB_BENCH_DEBUG") else None

    # one library call per step into buffers allocated once (the fp32-gather path hands `warped` to an asynchronous collective and
    # therefore needs a fresh tensor per step)
    stab = None if (args.alloc_per_step or args.gather_fp32) else vs.OriginalSizeStabiliser(B, H, W, Cin, H, W)

    def step():
        t0 = time.perf_counter()
        flows, outflow, warped = stab(feats, frame) if stab is not None else vs.stabilise_originalsize(feats, frame)
        t1 = time.perf_counter()
        if args.st_warp != "none":
            warped = st_fns[args.st_warp](warped)
        if vgg is not None:
            vgg.build(vvgg.preprocess(warped))
        if gather_on[0] and (gather is not None or grouped is not None):
            t2 = time.perf_counter()
            submit_frames(warped)
            if dbg is not None:
                dbg.append((t1 - t0, 0.0, time.perf_counter() - t2))
        nstep[0] += 1
        return flows, outflow, warped

    graph = None
    if args.graph:
        if gather is not None or grouped is not None:
            raise SystemExit("--graph is for the single-GPU path (no collective inside the capture)")
        args.no_kernel_events = True
        step()                                   # allocate workspaces / load kernels outside the capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_out = step()
        eager_step = step
        step = lambda: (graph.replay(), graph_out)[1]

    from coupe.optical_flow_based_deep_video_stabilization_amd import benchloop
    use_events = not args.no_kernel_events
    ev_every = max(1, min(args.event_every, max(1, args.steps // 2)))
    n_event_steps = 0

    def profilers_on():
        ctx.profile(use_events)
        runtime.hbm_profile(1 if use_events else 0)     # dispatch-timestamp events around the glue+warp launch too

    def timed_step(k):
        nonlocal n_event_steps
        if use_events and k >= 0:
            on = (k % ev_every == 0)
            ctx.profile_set(on)
            runtime.hbm_profile(2 if on else 0)
            n_event_steps += int(on)
        return step()

    # W warm-up steps, then exactly K steps between barrier + synchronize pairs; elapsed = max over ranks (benchloop.py)
    elapsed, out = benchloop.timed_region(timed_step, args.steps, args.warmup, torch.cuda.synchronize, dist=dist,
                                          drain=flush_and_drain if (gather is not None or grouped is not None) else None, before_timed=profilers_on,
                                          device="cpu" if args.backend == "gloo" else "cuda", prime=profilers_on if use_events else None)

    # ---- N > 1: what the reassembly costs.  The SAME ranks run the same K steps again with the collective (and the uint8 staging
    # launch that feeds it) switched off; exposed = with - without.  After the timed region, never part of `value`.
    gather_cost = None
    if (gather is not None or grouped is not None):
        gather_on[0] = False
        if use_events:
            ctx.profile_set(False)
            runtime.hbm_profile(0)
        e2, _ = benchloop.timed_region(lambda k: step(), args.steps, min(args.warmup, 3), torch.cuda.synchronize, dist=dist,
                                       device="cpu" if args.backend == "gloo" else "cuda")
        gather_on[0] = True
        per_rank = B * H * W * 3 * (4 if args.gather_fp32 else 1)
        gather_cost = {"schedule": args.gather_schedule, "transport": "gloo through host memory (rehearsal: the figures mean nothing)" if host else "RCCL",
                       "dtype": "fp32" if args.gather_fp32 else "uint8", "steps_per_collective": G,
                       "bytes_per_rank_per_step": per_rank, "bytes_gathered_per_step": per_rank * world,
                       "ms_per_step_with": round(elapsed / args.steps * 1e3, 4), "ms_per_step_without": round(e2 / args.steps * 1e3, 4),
                       "exposed_ms": round((elapsed - e2) / args.steps * 1e3, 4),
                       "how": f"{args.steps} more steps on the same ranks and buffers after the timed region, reassembly (staging launch + collective) off"}

    # the last timed step's outputs (first --err-samples samples), copied now: later legs reuse the workspace and the output buffers
    err_nb = max(1, min(B, args.err_samples))
    err_out = None
    if rank == 0 and not args.no_flow_err and args.st_warp == "none":
        fl_, of_, wf_ = out
        err_out = ({k: v[:err_nb].clone() for k, v in fl_.items()}, of_[:err_nb].clone() if of_ is not None else None, wf_[:err_nb].clone())
        torch.cuda.synchronize()

    if dbg:
        for i, r in enumerate(dbg[-args.steps:]):
            log(f"step {i:3d} host ms: path {r[0] * 1e3:7.3f}  quantise {r[1] * 1e3:7.3f}  submit {r[2] * 1e3:7.3f}")

    # ---- roofline of the dominant kernel (implicit-GEMM conv on the fp32 MFMA).  Every conv-like
    # launch is bracketed by HIP events on its own stream inside the C library; launches are grouped
    # by kernel instantiation (as rocprofv3 names them) and the one with the most time is reported.
    roofline = None
    tap_launch_ms = None
    if use_events:
        ms, flops, _nrec = ctx.profile_read()      # sums over every recorded pass of the timed region
        dflops = ctx.profile_read_direct()         # the same layers counted as direct convolutions
        ctx.profile(False)
        nf = max(n_event_steps, 1)                 # steps of the timed region that carried events
        flops = [f / nf for f in flops]
        dflops = [f / nf for f in dflops]
        inst = ctx.profile_kernel_names()
        groups = {}
        for name, m, f, df in zip(inst, ms, flops, dflops):
            g_ = groups.setdefault(name, [0.0, 0.0, 0, 0.0])
            g_[0] += m / nf; g_[1] += f; g_[2] += 1; g_[3] += df
        dom = max(groups, key=lambda k: groups[k][0])
        tot_ms, tot_fl = sum(ms) / nf, sum(flops)
        if "predict2_taps" in ctx.LAUNCH_SLOTS:        # the one HBM-bound launch among them: priced by bytes in roofline_hbm.other_rows
            i14 = list(ctx.LAUNCH_SLOTS).index("predict2_taps")
            tap_launch_ms = (ms[i14] / nf, inst[i14])
        if rank == 0:
            log(f"{'launch':<14}{'ms':>9}{'GFLOP':>10}{'TFLOP/s':>10}{'frac':>8}  kernel")
            for name, m, f, k in zip(ctx.LAUNCH_SLOTS, ms, flops, inst):
                m /= nf
                tf = f / (m * 1e-3) / 1e12 if m > 0 else 0.0
                log(f"{name:<14}{m:>9.4f}{f / 1e9:>10.2f}{tf:>10.1f}{tf / MFMA_F32_PEAK_TFLOPS:>8.3f}  {k}")
            log(f"{'all conv':<14}{tot_ms:>9.4f}{tot_fl / 1e9:>10.2f}{tot_fl / (tot_ms * 1e-3) / 1e12:>10.1f}"
                f"{tot_fl / (tot_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS:>8.3f}   step {elapsed / args.steps * 1e3:.3f} ms")
        traffic, traffic_source = None, "none"
        if (B, H, W, Cin) == (8, 512, 512, 27):
            try:     # NOT measured by this run: HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC pass of
                     # this workload (counters need their own profiler runs; scripts/gpu_profile.sh + scripts/pmc_summary.py)
                import glob
                pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_*.json")))
                if pj:
                    traffic = json.load(open(pj[-1])).get(dom, {}).get("hbm_bytes_per_launch_corrected")
                    if traffic is not None:
                        traffic_source = "static: profiles/" + os.path.basename(pj[-1]) + " (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE pass of the same command, corrected per MI355X_MICROARCH.md)"
            except Exception:
                traffic, traffic_source = None, "none"
        d_ms, d_fl, d_n, d_dfl = groups[dom]
        tot_dfl = sum(dflops)
        achieved = d_fl / (d_ms * 1e-3) / 1e12
        all_tf = tot_fl / (tot_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": dom, "launches_per_step": d_n, "event_steps": n_event_steps,
                    "avg_launch_us": round(d_ms / d_n * 1e3, 2),
                    "alg_flops_per_launch_avg": d_fl / d_n,
                    # achieved / frac count the flops the MFMA kernel ISSUES (hardware utilisation).  The 3x3 stride-1 stages run in
                    # Winograd F(2x2,3x3) form (4/9 of the direct convolution's multiply-adds); *_direct count those layers as the
                    # direct convolutions SURVEY.md 8d prices -- a work rate, which may exceed what the matrix pipe itself does
                    "flops_counted": "issued by the MFMA kernel; *_direct = same layers as direct convolutions (Winograd stages x9/4)",
                    "note": "peak = 157.3 TFLOP/s at the nominal 2.4 GHz; in situ the chip holds 2.34-2.39 GHz in these launches (in-kernel stamps, "
                            "profiles/insitu_stamps_r03k_asm.txt); the K loops are assembly blocks (0.98 of the pipe with two workgroups per CU, 0.94 with one), "
                            "what is left is prologue + epilogue that the workgroups of a launch run in lockstep.  Launches are grouped by kernel instantiation as "
                            "rocprofv3 names them: since round 4 a refinement level's transposed convolution shares its launch with the level's tap-table GEMM "
                            "(conv_dual_kernel: its own group; the events bracket both, the flops counted are the transposed convolution's), so this group is the "
                            "plain conv_mfma_kernel launches (conv2 ... conv6 at B=8 512x512); " + newest_steady_trace() + " is the steady-state rocprofv3 "
                            "trace of the same command, whose average for this kernel agrees with avg_launch_us",
                    "achieved_direct": round(d_dfl / (d_ms * 1e-3) / 1e12, 2),
                    "frac_direct": round(d_dfl / (d_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                    "all_mfma_launches": {"launches_per_step": 15, "ms_per_step": round(tot_ms, 4),
                                          "achieved": round(all_tf, 2), "frac": round(all_tf / MFMA_F32_PEAK_TFLOPS, 4),
                                          "alg_flops_per_step": tot_fl, "direct_flops_per_step": tot_dfl,
                                          "frac_direct": round(tot_dfl / (tot_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}}

    # ---- HBM-side roofline: the glue + warp launch of the same timed steps.  achieved = ALGORITHMIC bytes of the
    # launch (8 B per source-flow pixel + 8 B output flow + 12 B frame read + 12 B written per output pixel) / its kernel time
    # from dispatch-timestamp events on the launch stream; peak = 8 TB/s (HBM3E spec; 6.3 TB/s is what a float4 copy reaches).
    roofline_hbm = None
    if use_events:
        runtime.hbm_profile(0)
        hp = runtime.hbm_profile_read()
        pf2_row = hp.pop("pf2_gather", None)
        st_in_step = {k: hp.pop(k, None) for k in ("st_sampler", "homography_warp")}
        name = max(hp, key=lambda k: hp[k][0])
        ms_sum, nl, by = hp[name]
        if nl > 0 and ms_sum > 0:
            kernels = {"pf2_glue_warp": "pf2_glue_warp_kernel<true, " + ("true" if W % 4 == 0 else "false") + ", false>",      # <WRITE_FLOW, STAGE, U8>
                       "flow_glue_warp": "warp3_tile_kernel<true, true, 4, 16, 32, 2, true, false, " + ("true" if W % 4 == 0 else "false") + ">",
                       "warp_flow": "warp3_tile_kernel<false, false, ...>", "flow_resize_scale": "flow_resize_scale_kernel"}
            gbs = by / (ms_sum * 1e-3) / 1e9
            h_traffic, h_src = None, "not measured by this run"
            try:     # static, like roofline.traffic: the newest committed PMC pass of this workload (cfg1) or of B=16 1080p
                import glob
                tag = {(8, 512, 512, 27): "pmc_r*.json", (16, 1080, 1920, 27): "pmc1080_r*.json"}.get((B, H, W, Cin))
                pj = sorted(glob.glob(os.path.join(ROOT, "profiles", tag))) if tag else []
                if pj:
                    h_traffic = json.load(open(pj[-1])).get(kernels[name], {}).get("hbm_bytes_per_launch_corrected")
                    if h_traffic is not None:
                        h_src = "static: profiles/" + os.path.basename(pj[-1]) + " (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of the same command)"
            except Exception:
                h_traffic, h_src = None, "not measured by this run"
            roofline_hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": h_traffic, "traffic_source": h_src,
                            "kernel": kernels[name],
                            "entry_point": "vstab_stabilise_originalsize (its last launch: predict_flow2 gather + flow glue + tf_warp)" if name == "pf2_glue_warp" else "vstab_" + name,
                            "launches": nl,
                            "avg_launch_us": round(ms_sum / nl * 1e3, 2), "alg_bytes_per_launch": by / nl,
                            "alg_bytes_per_output_pixel": round(by / nl / (B * H * W), 2),
                            "note": "limit is the L1 tag pipe (12-byte gathers), not HBM: profiles/README.md, r02 warp study; "
                                    "flows of a random-weight network on noise frames move neighbouring sample points ~0.6 px apart per pixel"}
            if pf2_row and pf2_row[1] > 0 and pf2_row[0] > 0:        # K9: LDS / L2-bound gather; its compulsory bytes against the same peak
                pms, pn, pby = pf2_row
                pg = pby / (pms * 1e-3) / 1e9
                roofline_hbm["other_rows"] = [{"row": "K9 predict_flow2 gather (pf2_tile_kernel; not HBM-bound: nine LDS taps per pixel)",
                                               "launches": pn, "avg_launch_us": round(pms / pn * 1e3, 2), "alg_bytes_per_launch": pby / pn,
                                               "achieved": round(pg, 1), "unit": "GB/s", "frac": round(pg / HBM_PEAK_GBS, 4)}]
            if tap_launch_ms and tap_launch_ms[0] > 0:               # K9's table: 784 B read + 128 B written per quarter-resolution pixel
                h1, w1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
                h2, w2 = (h1 + 4 - 5) // 2 + 1, (w1 + 4 - 5) // 2 + 1
                tby = float(B * h2 * w2 * (784 + 128))
                tg = tby / (tap_launch_ms[0] * 1e-3) / 1e9
                roofline_hbm.setdefault("other_rows", []).append(
                    {"row": "K9 predict_flow2 tap table (" + tap_launch_ms[1] + "; HBM-bound, 7.8 FLOP/B)", "launches": 1,
                     "avg_launch_us": round(tap_launch_ms[0] * 1e3, 2), "alg_bytes_per_launch": tby, "achieved": round(tg, 1), "unit": "GB/s",
                     "frac": round(tg / HBM_PEAK_GBS, 4)})
            # S1-S3: the spatial-transformer / warp.py samplers on this run's frames (24 B per output pixel: 12 gathered + 12 written;
            # theta is 24-36 B per SAMPLE).  Inside the step with --st-warp; otherwise five launches each AFTER the timed region.
            st_kernel = {"affine": "st3_tile_kernel<0, %s>", "projective": "st3_tile_kernel<0, %s>", "homography": "st3_tile_kernel<2, %s>"}
            slot_of = {"affine": "st_sampler", "projective": "st_sampler", "homography": "homography_warp"}
            for mode in ("affine", "projective", "homography"):
                if args.st_warp == mode:
                    sms, sn, sby = st_in_step[slot_of[mode]]
                    where = "inside the timed steps (--st-warp)"
                elif args.st_warp == "none":
                    _, _, wf = out
                    st_fns[mode](wf)
                    torch.cuda.synchronize()
                    runtime.hbm_profile(1)
                    for _ in range(5):
                        st_fns[mode](wf)
                    torch.cuda.synchronize()
                    runtime.hbm_profile(0)
                    sms, sn, sby = runtime.hbm_profile_read()[slot_of[mode]]
                    where = "5 launches on the last step's stabilised frames, after the timed region"
                else:
                    continue
                if sn > 0 and sms > 0:
                    sg = sby / (sms * 1e-3) / 1e9
                    roofline_hbm.setdefault("other_rows", []).append(
                        {"row": {"affine": "S2 AffineTransformer.transform", "projective": "S2 ProjectiveTransformer.transform",
                                 "homography": "S3 warp.transformImage"}[mode] + f" at {B}x{H}x{W}x3 (" + where + ")",
                         "kernel": st_kernel[mode] % ("true" if W % 4 == 0 else "false"), "launches": sn,
                         "avg_launch_us": round(sms / sn * 1e3, 2), "alg_bytes_per_launch": sby / sn,
                         "alg_bytes_per_output_pixel": round(sby / sn / (B * H * W), 2),
                         "achieved": round(sg, 1), "unit": "GB/s", "frac": round(sg / HBM_PEAK_GBS, 4)})
                    if rank == 0:
                        log(f"{'st ' + mode:<18}{sms / sn:>9.4f} ms  {sby / sn / 1e6:>9.1f} MB  {sg:>8.1f} GB/s  frac {sg / HBM_PEAK_GBS:.3f} of 8 TB/s")
            if rank == 0:
                log(f"{name:<18}{ms_sum / nl:>9.4f} ms  {by / nl / 1e6:>9.1f} MB  {gbs:>8.1f} GB/s  frac {gbs / HBM_PEAK_GBS:.3f} of 8 TB/s")

    value = benchloop.aggregate_value(B, world, args.steps, elapsed)
    res = {
        "metric": f"stabilised frame-pairs/sec @{H}x{W}",
        "value": round(value, 2),
        "unit": "frame-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (uniform [0,1) frames, seeded He-normal weights; no checkpoint offline)",
        "config": {"workload": f"batch={B} {H}x{W}x{Cin} frame stacks per GPU: FlowNetS-pyramid forward "
                               f"(5 flows) + flow resize/scale + tf_warp at {H}x{W}",
                   "batch_per_gpu": B, "height": H, "width": W, "cin": Cin,
                   "gflop_per_sample": round(netspec.gflop_per_sample(H, W, Cin), 2),
                   "all_gather": (("fp32" if args.gather_fp32 else "uint8") + " warped frames, async over " + ("gloo (host memory; rehearsal)" if host else "RCCL") + ", schedule " + args.gather_schedule + f", one collective per {G} step(s)") if (gather is not None or grouped is not None) else False,
                   "vgg16_trunk": bool(args.vgg16), "st_warp": args.st_warp, "plan_flags": args.plan_flags, "plan_batch": args.plan_batch,
                   "host_calls_per_step": "1 (vstab_stabilise_originalsize, outputs pre-allocated)" if stab is not None else "2 + 7 allocations"},
        "roofline": roofline,
        "roofline_hbm": roofline_hbm,
        "all_gather": gather_cost,
    }
    # ---- secondary rows, AFTER the timed region and never part of `value`: what bench_clip.py (BASELINE configs[3]) and
    # bench_train.py measure at length, in short form, so that a driver that only runs bench.py sees them
    if rank == 0 and world == 1 and dist is None and not args.no_secondary and graph is None:
        res["secondary"] = secondary_rows(vs, runtime, log)
    if rank == 0:
        nb = err_nb
        feats_h, frame_h = feats[:nb].cpu().numpy(), frame[:nb].cpu().numpy()
        ref32 = None
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"], ref32 = cpu_baseline(feats_h, frame_h, weights, args.cpu_seconds)
            except Exception as e:   # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "frame-pairs/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        else:
            res["cpu_baseline"] = None
        # ---- the metric's second half: max-abs flow error of the timed workload (last timed step) vs the CPU restatement, same inputs
        if err_out is not None:
            try:
                res["flow_err"] = flow_error(err_out, feats_h, frame_h, weights, ref32)
                fe = res["flow_err"]
                log(f"flow_err vs fp32 restatement: worst {fe['worst']:.3e} (tol {FLOW_ERR_TOL:g}), max |flow| {fe['max_abs_flow']}")
            except Exception as e:   # noqa: BLE001
                res["flow_err"] = {"error": repr(e)[:300]}
        else:
            res["flow_err"] = None
        print(json.dumps(res), file=real_stdout, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
