"""GPU tests of the round-4 one-sample path: few-row layers as weight streams with the split-K reduction finished inside the launch
(csrc/conv_skinny.hip; model.py:838-852 at main:568-569's one sess.run per frame), and pinned plan batches (vstab_set_plan_batch:
a sample's bits do not depend on what it is batched with; SURVEY.md 8e, main:553-558)."""
import numpy as np
import pytest
import torch

import coupe.optical_flow_based_deep_video_stabilization_amd as vs
from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, runtime, weights as wts
from oracle import vstab_oracle as vo

pytestmark = pytest.mark.gpu
KEYS = ("predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3", "predict_flow2")
NO_SKINNY = 1
EPS32 = 1.1920929e-07
# Two kernel families (or launch schedules) for the same layers associate the same sums differently.  Their flows differ by a few fp32
# epsilons of the flow itself: profiles/flow_err_margin_r05.md measures at most 11.9 eps of a level's largest flow over every shape,
# weight set and plan there (the pyramid doubles a level's rounding into the next one and predict_flow2 takes 8 x up(predict_flow3)).
PLAN_TO_PLAN_EPS = 16


@pytest.fixture
def ctx():
    runtime.reset()
    vs.assign_weights(wts.synthetic_weights(seed=1, cin=27, random_bn=True, flow_gain=2.0))
    c = runtime.get_context()
    yield c
    runtime.reset()


def layer_tiles(B, H, W, flags=0, plan_batch=0):
    import ctypes
    out = (ctypes.c_int32 * 200)()
    tiles = []
    for layer in range(19):
        n = _lib.lib().vstab_host_layer_plan_pinned(plan_batch, flags, B, H, W, 27, layer, out, 200)
        assert n > 0
        tiles.append((out[21], out[19], out[25]))          # tile id, split-K factor, Winograd form
    return tiles


@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (1, 384, 512), (2, 96, 128), (1, 200, 264), (3, 64, 64), (4, 128, 128)])
def test_weight_stream_layers_match_the_tiled_form_and_the_oracle(ctx, B, H, W):
    # same layers through the two kernels: different association of the same sums -> equal up to fp32 rounding, far inside the
    # flow budget; and the default (weight-stream) path against the fp64 restatement, every level
    assert any(t[0] == 6 for t in layer_tiles(B, H, W)), "this shape should have weight-stream layers"
    assert not any(t[0] == 6 for t in layer_tiles(B, H, W, NO_SKINNY))
    g = torch.Generator().manual_seed(H * 7 + W)
    feats = torch.rand(B, H, W, 27, generator=g)
    a = vs.flownetS_pyramid(feats.cuda(), B)
    ia = {k: v.clone() for k, v in ctx.internals(B, H, W, 27).items() if k in ("conv5", "concat5", "conv6", "conv6_1", "concat4")}
    a = {k: a[k].clone() for k in KEYS}
    ctx.set_plan_flags(NO_SKINNY)
    b = vs.flownetS_pyramid(feats.cuda(), B)
    ib = {k: v.clone() for k, v in ctx.internals(B, H, W, 27).items() if k in ia}
    ctx.set_plan_flags(0)
    for k in ia:
        scale = max(1.0, float(ib[k].abs().max()))
        assert float((ia[k] - ib[k]).abs().max()) <= 2e-5 * scale, k
    ref = vo.flownetS_pyramid(feats.numpy(), wts.synthetic_weights(seed=1, cin=27, random_bn=True, flow_gain=2.0), torch.float64)
    for k in KEYS:
        mag = max(1.0, float(ref[k].abs().max()))
        assert float((a[k] - b[k]).abs().max()) <= PLAN_TO_PLAN_EPS * EPS32 * mag, (k, mag)
        assert float((a[k].double().cpu() - ref[k]).abs().max()) <= 1e-3, k


@pytest.mark.parametrize("H,W", [(256, 256), (384, 512)])
def test_in_launch_split_k_reduction_is_reproducible_under_load(ctx, H, W):
    # the last arriver of a tile must read every slab as its writers left it (agent-scope hand-off: cdna_hip_programming.md G16).
    # Two different inputs alternate through the SAME slab and ticket memory, forty forwards back to back with no host waits in
    # between: a stale slab line (L1 of the reducer's CU, a not-yet-written-through store) shows as a result that is not
    # bit-identical to the first one of its input.
    g = torch.Generator().manual_seed(3)
    xs = [torch.rand(1, H, W, 27, generator=g).cuda() for _ in range(2)]
    first = [None, None]
    outs = []
    for it in range(300 if H == 256 else 60):
        r = vs.flownetS_pyramid(xs[it & 1], 1)
        outs.append((it & 1, {k: r[k].clone() for k in KEYS}))
    torch.cuda.synchronize()
    for which, r in outs:
        if first[which] is None:
            first[which] = r
            continue
        for k in KEYS:
            assert torch.equal(r[k], first[which][k]), k
    assert not torch.equal(first[0]["predict_flow2"], first[1]["predict_flow2"])
    # and the ticket words are zero again (the library's invariant between launches): one more forward still agrees
    r = vs.flownetS_pyramid(xs[0], 1)
    assert torch.equal(r["predict_flow2"], first[0]["predict_flow2"])


@pytest.mark.parametrize("P,H,W", [(8, 256, 256), (8, 512, 512), (5, 136, 200)])
def test_pinned_plan_batch_makes_a_sample_independent_of_its_batch(ctx, P, H, W):
    # split-K factors, Winograd or direct form and the kernel family are functions of the batch; pinned to the plan of P samples
    # every smaller batch reproduces its samples bit for bit (what a ragged last micro-batch of a sharded clip needs)
    g = torch.Generator().manual_seed(P * 31 + H)
    feats = torch.rand(P, H, W, 27, generator=g).cuda()
    unpinned = {b: layer_tiles(b, H, W) for b in (1, P)}
    if H >= 256:
        assert unpinned[1] != unpinned[P], "the shapes of this test should plan a lone sample differently"
    ctx.set_plan_batch(P)
    whole = vs.flownetS_pyramid(feats, P)
    whole = {k: whole[k].clone() for k in KEYS}
    for lo, hi in ((0, 1), (P - 1, P), (1, 4), (2, P), (0, P - 1)):
        part = vs.flownetS_pyramid(feats[lo:hi].contiguous(), hi - lo)
        for k in KEYS:
            assert torch.equal(part[k], whole[k][lo:hi]), (k, lo, hi)
    with pytest.raises(Exception):
        vs.flownetS_pyramid(torch.cat([feats, feats[:1]]), P + 1)        # beyond the pinned batch
    ctx.set_plan_batch(0)
    again = vs.flownetS_pyramid(feats, P)                                   # unpinned at the batch itself = the pinned plan's own batch
    for k in KEYS:
        assert torch.equal(again[k], whole[k]), k


@pytest.mark.parametrize("B,H,W", [(1, 384, 512), (1, 256, 256), (8, 512, 512), (3, 88, 104), (2, 720, 1280)])
def test_two_launch_refinement_levels_equal_the_four_launch_sequence_bit_for_bit(ctx, B, H, W):
    # a level's transposed convolution + tap-table GEMM in one launch (conv_dual_kernel) and its split-K combine + predict_up in one
    # launch (combine_predict_up_kernel) are the same workgroup programs as the four separate launches: identical bits, every level,
    # every internal tensor of the decoder.  (Flag 1 keeps the few-row layers on the tiled kernel in both runs: the four-launch
    # sequence may otherwise put a transposed convolution on the weight-stream kernel, whose sums associate differently; flag 8 keeps the
    # transposed convolutions in their direct form in both runs: the Winograd form of round 6 exists in the two-problem launch only.)
    g = torch.Generator().manual_seed(B * 1000 + H)
    feats = torch.rand(B, H, W, 27, generator=g).cuda()
    names = ("concat5", "concat4", "concat3", "concat2")
    ctx.set_plan_flags(1 | 8)
    a = vs.flownetS_pyramid(feats, B)
    a = {k: a[k].clone() for k in KEYS}
    ia = {k: v.clone() for k, v in ctx.internals(B, H, W, 27).items() if k in names}
    ctx.set_plan_flags(1 | 2 | 8)
    b = vs.flownetS_pyramid(feats, B)
    ib = {k: v for k, v in ctx.internals(B, H, W, 27).items() if k in names}
    for k in names:
        assert torch.equal(ia[k], ib[k]), k
    for k in KEYS:
        assert torch.equal(a[k], b[k]), k
    ctx.set_plan_flags(0)


def _raw_forward(ctx, feats, ws, stream):
    """vstab_flownets_forward through ctypes with a caller-chosen workspace and stream."""
    B, H, W, Cin = feats.shape
    lv = vs.netspec.sizes_for(H, W).level
    flows = [torch.empty((B, lv[k][0], lv[k][1], 2), dtype=torch.float32, device="cuda") for k in (6, 5, 4, 3)]
    flows.append(torch.empty((B, H - 2, W - 2, 2), dtype=torch.float32, device="cuda"))
    _lib.check(_lib.lib().vstab_flownets_forward(ctx._h, feats.data_ptr(), B, H, W, Cin, *[f.data_ptr() for f in flows], ws.data_ptr(), ws.numel(),
                                                 stream.cuda_stream), ctx._h)
    return flows


@pytest.mark.parametrize("H,W", [(256, 256), (384, 512)])
def test_forwards_on_one_context_with_distinct_workspaces_may_overlap(ctx, H, W):
    # vstab.h: the context owns only the packed weights; everything a forward writes -- the ticket words of the in-launch split-K
    # reductions included -- lies in the caller's workspace, so two streams may run forwards on ONE context side by side when each
    # has its own workspace.  The workspaces start as garbage (0xFF bytes: non-zero ticket words, NaN slabs): the forward zeroes
    # what it needs.  Every overlapped result is bit-identical to the serial one.
    n = _lib.lib().vstab_workspace_bytes_ctx(ctx._h, 1, H, W, 27)
    g = torch.Generator().manual_seed(11)
    xs = [torch.rand(1, H, W, 27, generator=g).cuda() for _ in range(2)]
    serial = []
    for x in xs:
        r = vs.flownetS_pyramid(x, 1)
        serial.append([r[k].clone() for k in KEYS])
    wss = [torch.full((n,), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    outs = [[], []]
    for it in range(40):
        for s in range(2):
            outs[s].append(_raw_forward(ctx, xs[s], wss[s], streams[s]))
    torch.cuda.synchronize()
    for s in range(2):
        for fl in outs[s]:
            for a, b, k in zip(fl, serial[s], KEYS):
                assert torch.equal(a, b), (s, k)
    # a workspace whose ticket words were left dirty (what a launch that failed mid-flight leaves) is healed by the next forward
    ent = (_lib.VstabWsEntry * 24)()
    m = _lib.lib().vstab_workspace_layout_ctx(ctx._h, 1, H, W, 27, ent, 24)
    tk = [e for e in ent[:m] if e.name == b"tickets"][0]
    wss[0][tk.offset_bytes:tk.offset_bytes + 4 * tk.w] = 7
    fl = _raw_forward(ctx, xs[0], wss[0], streams[0])
    torch.cuda.synchronize()
    for a, b, k in zip(fl, serial[0], KEYS):
        assert torch.equal(a, b), k
    assert int(wss[0][tk.offset_bytes:tk.offset_bytes + 4 * tk.w].view(torch.int32).abs().max()) == 0       # and every launch left them zero


def test_forwards_on_one_context_from_two_host_threads(ctx):
    # vstab.h's threading contract as a caller reads it: one context, two HOST threads, each with its own stream and workspace, issuing
    # forwards concurrently (ctypes releases the GIL around the call).  A successful forward only READS the context (round 5 assigned
    # std::string name slots on every call: a data race between two threads) -- results are bit-identical to the serial ones.
    import threading
    H, W = 256, 256
    n = _lib.lib().vstab_workspace_bytes_ctx(ctx._h, 1, H, W, 27)
    g = torch.Generator().manual_seed(12)
    xs = [torch.rand(1, H, W, 27, generator=g).cuda() for _ in range(2)]
    serial = []
    for x in xs:
        r = vs.flownetS_pyramid(x, 1)
        serial.append([r[k].clone() for k in KEYS])
    wss = [torch.empty((n,), dtype=torch.uint8, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    outs, errs = [[], []], []

    def worker(s):
        try:
            with torch.cuda.device(0):
                for _ in range(200):
                    outs[s].append(_raw_forward(ctx, xs[s], wss[s], streams[s]))
        except Exception as e:          # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=worker, args=(s,)) for s in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for s in range(2):
        assert len(outs[s]) == 200
        for fl in outs[s][::20] + [outs[s][-1]]:
            for a, b, k in zip(fl, serial[s], KEYS):
                assert torch.equal(a, b), (s, k)


@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (1, 384, 512)])
def test_four_phase_transposed_convolutions_on_the_weight_stream_kernel(ctx, B, H, W):
    # plan flag 2 alone (no dual launch) sends few-row transposed convolutions through conv_skinny_kernel: per-phase weight offsets,
    # slab base phase*ks*slab, ticket index (phase*gridDim.x+bx)*gridDim.y+by.  Against the tiled form (flags 1|2) to fp32 rounding
    # and against the fp64 restatement, every decoder tensor and every flow.
    tiles = layer_tiles(B, H, W, 2)
    assert any(t[0] == 6 for t in tiles[10:14]), "flag 2 should put a transposed convolution on the weight-stream kernel at this shape"
    g = torch.Generator().manual_seed(H + 13 * W)
    feats = torch.rand(B, H, W, 27, generator=g)
    names = ("concat5", "concat4", "concat3", "concat2")
    ctx.set_plan_flags(2)
    a = vs.flownetS_pyramid(feats.cuda(), B)
    a = {k: a[k].clone() for k in KEYS}
    ia = {k: v.clone() for k, v in ctx.internals(B, H, W, 27).items() if k in names}
    ctx.set_plan_flags(1 | 2)
    b = vs.flownetS_pyramid(feats.cuda(), B)
    ib = {k: v.clone() for k, v in ctx.internals(B, H, W, 27).items() if k in names}
    ctx.set_plan_flags(0)
    w = wts.synthetic_weights(seed=1, cin=27, random_bn=True, flow_gain=2.0)
    ref, internals = vo.flownetS_pyramid(feats.numpy(), w, torch.float64, return_internals=True)
    for k in names:
        scale = max(1.0, float(ib[k].abs().max()))
        assert float((ia[k] - ib[k]).abs().max()) <= 2e-5 * scale, k
        assert float((ia[k].double().cpu() - internals[k]).abs().max()) <= 1e-3 * scale, k
    for k in KEYS:
        mag = max(1.0, float(ref[k].abs().max()))
        assert float((a[k] - b[k]).abs().max()) <= PLAN_TO_PLAN_EPS * EPS32 * mag, (k, mag)
        assert float((a[k].double().cpu() - ref[k]).abs().max()) <= 1e-3, k


def test_workspace_layout_of_a_pinned_context_matches_what_the_forward_uses(ctx):
    # vstab_workspace_layout_ctx / vstab_workspace_bytes_ctx describe the plan THIS context runs: under a pinned batch the plan-dependent
    # buffers (split-K slabs, Winograd V / M: always the last three entries) may differ from the unpinned plan's, the activation offsets
    # never do; a workspace of exactly vstab_workspace_bytes_ctx bytes is enough for the forward, one byte less is refused.
    import ctypes
    L = _lib.lib()
    B, H, W = 3, 256, 256
    ent_u, ent_p = (_lib.VstabWsEntry * 24)(), (_lib.VstabWsEntry * 24)()
    nu = L.vstab_workspace_layout(B, H, W, 27, ent_u, 24)
    ctx.set_plan_batch(8)
    npn = L.vstab_workspace_layout_ctx(ctx._h, B, H, W, 27, ent_p, 24)
    assert nu == npn == 19
    for a, b in zip(ent_u[:nu - 3], ent_p[:npn - 3]):
        assert (a.name, a.offset_bytes, a.h, a.w, a.c, a.c_stride) == (b.name, b.offset_bytes, b.h, b.w, b.c, b.c_stride)
    assert [e.name for e in ent_p[npn - 3:npn]] == [b"splitk", b"winograd_in", b"winograd_out"]
    need = L.vstab_workspace_bytes_ctx(ctx._h, B, H, W, 27)
    last = ent_p[npn - 1]
    assert last.offset_bytes + 4 * last.w <= need
    feats = torch.rand(B, H, W, 27).cuda()
    lv = vs.netspec.sizes_for(H, W).level
    flows = [torch.empty((B, lv[k][0], lv[k][1], 2), device="cuda") for k in (6, 5, 4, 3)] + [torch.empty((B, H - 2, W - 2, 2), device="cuda")]
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    args = [ctx._h, feats.data_ptr(), B, H, W, 27] + [f.data_ptr() for f in flows]
    assert L.vstab_flownets_forward(*args, ws.data_ptr(), need, runtime.stream_ptr()) == 0
    assert L.vstab_flownets_forward(*args, ws.data_ptr(), need - 1, runtime.stream_ptr()) == -4        # VSTAB_E_NOMEM
    torch.cuda.synchronize()
    ref = vs.flownetS_pyramid(feats, B)
    assert torch.equal(flows[4], ref["predict_flow2"])
    ctx.set_plan_batch(0)
