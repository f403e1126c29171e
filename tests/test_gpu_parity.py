"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors.  Tolerance for flows: 1e-3 max-abs (BASELINE.json north_star), fp32."""
import os

import numpy as np
import pytest
import torch

import coupe.optical_flow_based_deep_video_stabilization_amd as vs
from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, netspec, runtime, weights as wts
from oracle import vstab_oracle as vo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FLOW_TOL = 1e-3


def dev(a):
    return torch.as_tensor(np.asarray(a)).to("cuda")


def maxabs(a, b):
    return float((a.double().cpu() - torch.as_tensor(np.asarray(b)).double()).abs().max())


# ------------------------------------------------------------------------- warp + glue
def test_warp_golden_bit_exact():
    z = np.load(os.path.join(GOLD, "warp_37x53.npz"))
    out = vs.tf_warp(dev(z["img"]), dev(z["flow"]), 37, 53)
    torch.cuda.synchronize()
    # same fp32 op sequence as the oracle, contraction disabled -> identical bits
    assert np.array_equal(out.cpu().numpy(), z["warped"])


@pytest.mark.parametrize("B,H,W,C", [(1, 5, 7, 3), (3, 64, 96, 3), (2, 33, 41, 1), (1, 30, 30, 4), (1, 1, 1, 3)])
def test_warp_random_vs_oracle(B, H, W, C):
    g = torch.Generator().manual_seed(H * 1000 + W)
    img = torch.rand(B, H, W, C, generator=g)
    flow = (torch.rand(B, H, W, 2, generator=g) - 0.5) * 2 * max(H, W)
    flow[0, 0, 0] = torch.tensor([0.0, 0.0])
    out = vs.tf_warp(img.cuda(), flow.cuda(), H, W)
    ref = vo.tf_warp(img, flow, H, W, torch.float32)
    assert maxabs(out, ref) <= 1e-6
    # identity flow zeroes the last row and column
    ident = vs.tf_warp(img.cuda(), torch.zeros(B, H, W, 2, device="cuda"), H, W).cpu()
    assert torch.equal(ident[:, :H - 1, :W - 1], img[:, :H - 1, :W - 1])
    assert ident[:, H - 1].abs().sum() == 0 and ident[:, :, W - 1].abs().sum() == 0


def test_warp_nonfinite_flow_does_not_fault():
    img = torch.rand(1, 8, 8, 3).cuda()
    flow = torch.zeros(1, 8, 8, 2)
    flow[0, 0, 0] = torch.tensor([float("nan"), float("inf")])
    flow[0, 1, 1] = torch.tensor([-float("inf"), 3e38])
    out = vs.tf_warp(img, flow.cuda(), 8, 8)
    torch.cuda.synchronize()
    assert out.shape == (1, 8, 8, 3)


@pytest.mark.parametrize("B,H,W", [(1, 37, 53), (2, 64, 96), (3, 31, 1031), (1, 270, 480), (5, 8, 9), (1, 1, 1)])
def test_warp_tiled_kernel_bit_exact_vs_fp32_oracle(B, H, W):
    # the 3-channel warp works on 1024-pixel tiles of the flat pixel index (ragged last tile, rows shorter and longer than a
    # wave pass, tiles that span samples): same fp32 statement sequence as the oracle -> identical bits
    g = torch.Generator().manual_seed(B * 7919 + H * 31 + W)
    img = torch.rand(B, H, W, 3, generator=g)
    flow = torch.randn(B, H, W, 2, generator=g) * 6
    flow[0, 0, 0] = torch.tensor([-0.5, -0.25])          # extrapolating corner (A.6)
    flow[-1, -1, -1] = torch.tensor([3.0, 3.0])          # beyond the far edge -> 0
    out = vs.tf_warp(img.cuda(), flow.cuda(), H, W)
    ref = vo.tf_warp(img, flow, H, W, torch.float32)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("B,hn,wn,oh,ow", [(2, 64, 64, 64, 64), (1, 70, 90, 96, 120), (3, 64, 96, 37, 53), (1, 48, 64, 135, 240),
                                           (2, 384, 512, 384, 512)])
def test_fused_glue_warp_bit_identical(B, hn, wn, oh, ow):
    # main:497-514 in ONE launch must give the bits of vstab_flow_resize_scale followed by vstab_warp_flow (and the oracle's
    # fp32 restatement of the two steps), with and without the output-resolution flow being written
    g = torch.Generator().manual_seed(hn * 131 + ow)
    pf2 = (torch.randn(B, hn - 2, wn - 2, 2, generator=g) * 4).cuda()
    frame = torch.rand(B, oh, ow, 3, generator=g).cuda()
    of_ref = vs.flow_to_output_res(pf2, hn, wn, oh, ow)
    wp_ref = vs.tf_warp(frame, of_ref, oh, ow)
    of, wp = vs.flow_glue_warp(pf2, frame, hn, wn)
    assert torch.equal(of, of_ref) and torch.equal(wp, wp_ref)
    none, wp2 = vs.flow_glue_warp(pf2, frame, hn, wn, want_outflow=False)
    assert none is None and torch.equal(wp2, wp_ref)
    # the glue is the graph's op sequence -- (pf2*net_h)/h, legacy bilinear, (x*ow)/net_w, (y*oh)/net_h, one fp32 rounding per
    # TF op (main:497-498) -- in the kernels as in the fp32 oracle: identical bits
    cpu_of = vo.flow_to_output_res(pf2.cpu(), hn, wn, oh, ow)
    assert cpu_of.dtype == torch.float32 and torch.equal(of.cpu(), cpu_of)
    assert torch.equal(wp.cpu(), vo.tf_warp(frame.cpu(), of.cpu(), oh, ow, torch.float32))


@pytest.mark.parametrize("B,hn,wn,oh,ow,fused", [
    (8, 512, 512, 512, 512, True),       # BASELINE configs[1]: the headline step
    (1, 256, 256, 256, 256, True),       # configs[0]
    (1, 384, 512, 720, 1280, True),      # the clip driver's shape: the output 1.9x the flow grid
    (2, 200, 264, 200, 264, True),       # ragged tiles on both axes
    (3, 136, 200, 250, 333, True),       # output width not a multiple of 4: unstaged stores
    (2, 128, 160, 120, 150, True),       # output slightly SMALLER than the flow grid (126 x 158): source steps of more than one pixel
    (1, 384, 512, 96, 128, False),       # output a quarter of the flow grid: the rectangles do not fit LDS -> two launches, same bits
])
def test_fused_tail_bit_identical(B, hn, wn, oh, ow, fused):
    # vstab_stabilise_originalsize ends in ONE launch for predict_flow2's gather (model.py:882-887), the glue (main:497-498) and tf_warp
    # (main:514) when the geometry allows: every output -- the five flows, the output-resolution flow, the warped frame -- must carry
    # the bits of the separate launches (plan flag 4), with and without the output-resolution flow; predict_flow2 is written once by
    # the tiles that own its pixels (sentinel check: no pixel left unwritten).
    import ctypes as C
    runtime.reset()
    vs.assign_weights(wts.synthetic_weights(seed=6, cin=27, random_bn=True, flow_gain=1.0))
    ctx = runtime.get_context()
    g = torch.Generator().manual_seed(hn * 7 + ow)
    feats = torch.rand(B, hn, wn, 27, generator=g).cuda()
    frame = torch.rand(B, oh, ow, 3, generator=g).cuda()
    for want in (True, False):
        ctx.set_plan_flags(4)
        ref = vs.OriginalSizeStabiliser(B, hn, wn, 27, oh, ow, want_outflow=want)
        rf, ro, rw = ref(feats, frame)
        rf = {k: v.clone() for k, v in rf.items()}
        ro, rw = (ro.clone() if want else None), rw.clone()
        ctx.set_plan_flags(0)
        st = vs.OriginalSizeStabiliser(B, hn, wn, 27, oh, ow, want_outflow=want)
        st.flows[4].fill_(float("nan"))                 # predict_flow2: every pixel must be written by its owner tile
        st.warped.fill_(float("nan"))
        runtime.hbm_profile(1)
        f, o, w_ = st(feats, frame)
        torch.cuda.synchronize()
        runtime.hbm_profile(0)
        prof = runtime.hbm_profile_read()
        assert (prof["pf2_glue_warp"][1] == 1) == fused and (prof["flow_glue_warp"][1] == 1) == (not fused)
        for k in rf:
            assert torch.equal(f[k], rf[k]), (k, want)
        assert torch.equal(w_, rw), want
        if want:
            assert torch.equal(o, ro)
        else:
            assert o is None
    runtime.reset()


def test_glue_division_by_launch_constants_is_the_ieee_quotient():
    # The glue divides by three constants of the launch (main:497-498: /382, /512, /384).  The kernels do it with five fused
    # operations on a host-side reciprocal instead of the ~11-instruction run-time division; this compares the two ON THE DEVICE:
    # for the divisors of the reference's and BASELINE's sizes over EVERY fp32 bit pattern (2^32 numerators each: -0.0, denormal
    # quotients and infinities are compared bit for bit too, only NaN numerators are skipped), for every integer divisor up to 4096
    # over a stride through the patterns, and checks which divisors take the plain division.
    import ctypes as C
    L = _lib.lib()
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = runtime.stream_ptr()
    for d in (382.0, 510.0, 384.0, 512.0, 718.0, 1278.0, 720.0, 1280.0, 1078.0, 1918.0, 1080.0, 1920.0, 254.0, 3.0, 7.0, 1023.0, 16777213.0):
        _lib.check(L.vstab_selftest_div_const(C.c_float(d), 0, 1 << 32, bad.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    for d in range(1, 4097):
        _lib.check(L.vstab_selftest_div_const(C.c_float(float(d)), (d * 2654435761) & 0xFFFFFFFF, 1 << 21, bad.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    for d in (0.5, 16777215.0, float(1 << 25), float("nan")):          # < 1, all-ones significand, > 2^24: plain division there
        assert L.vstab_selftest_div_const(C.c_float(d), 0, 16, bad.data_ptr(), st) == -1
    # the zero / denormal corner explicitly: the 2^24 patterns around +0 and around -0 (every denormal and the smallest normals)
    for first in (0, 0x80000000):
        _lib.check(L.vstab_selftest_div_const(C.c_float(382.0), first, 1 << 24, bad.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    # and through the glue itself on values that once told the single-multiply form apart (tests/test_oracle_kat.py)
    pf2 = torch.linspace(-40.0, 40.0, 382 * 4).view(1, 382, 4, 1).repeat(1, 1, 1, 2).contiguous()
    assert torch.equal(vs.flow_to_output_res(pf2.cuda(), 384, 512, 382, 4).cpu(), vo.flow_to_output_res(pf2, 384, 512, 382, 4))


def test_one_call_stabiliser_bit_identical_and_reuses_its_buffers():
    # vstab_stabilise_originalsize (one library call, outputs allocated once) against the two-call path
    w = wts.synthetic_weights(seed=3, cin=27, random_bn=True, flow_gain=2.0)
    runtime.reset()
    vs.assign_weights(w)
    g = torch.Generator().manual_seed(5)
    for (B, H, W, oh, ow) in ((2, 64, 96, 64, 96), (1, 70, 90, 96, 120)):
        feats = torch.rand(B, H, W, 27, generator=g).cuda()
        frame = torch.rand(B, oh, ow, 3, generator=g).cuda()
        flows, outflow, warped = vs.stabilise_originalsize(feats, frame)
        stab = vs.OriginalSizeStabiliser(B, H, W, 27, oh, ow)
        f2, o2, w2 = stab(feats, frame)
        for k in vo.FLOW_KEYS:
            assert torch.equal(f2[k], flows[k]), k
        assert torch.equal(o2, outflow) and torch.equal(w2, warped)
        ptrs = (w2.data_ptr(), o2.data_ptr(), f2["predict_flow2"].data_ptr())
        feats2 = torch.rand(B, H, W, 27, generator=g).cuda()
        f3, o3, w3 = stab(feats2, frame)
        assert (w3.data_ptr(), o3.data_ptr(), f3["predict_flow2"].data_ptr()) == ptrs          # same buffers, new contents
        assert torch.equal(w3, vs.stabilise_originalsize(feats2, frame)[2])
        nof = vs.OriginalSizeStabiliser(B, H, W, 27, oh, ow, want_outflow=False)
        assert nof(feats2, frame)[1] is None and torch.equal(nof(feats2, frame)[2], w3)
        with pytest.raises(ValueError):
            stab(feats[:, :-2], frame)
        with pytest.raises(ValueError):
            stab(feats, frame.double())
    with pytest.raises(ValueError):
        vs.OriginalSizeStabiliser(1, 64, 64, 6, 64, 64)          # weights are for 27 channels
    # a batch-sliced view of odd-sized frames is contiguous but not 16-byte aligned: refused BEFORE the network runs (the one-call path
    # ends in the fused glue + warp launch; stabilise_originalsize takes the two-launch path for such a view)
    big = torch.rand(3, 37, 53, 3, generator=g).cuda()
    fe = torch.rand(2, 64, 96, 27, generator=g).cuda()
    odd = vs.OriginalSizeStabiliser(2, 64, 96, 27, 37, 53)
    assert big[1:].data_ptr() % 16 != 0 and big[1:].is_contiguous()
    with pytest.raises(ValueError, match="16-byte"):
        odd(fe, big[1:])
    assert torch.equal(odd(fe, big[1:].clone())[2], vs.stabilise_originalsize(fe, big[1:])[2])


def test_batch_sliced_odd_frame_takes_the_two_launch_path():
    # a view like frame[1:] of an odd-sized frame starts at an address that is not 16-byte aligned: the fused launch cannot
    # take it (its contract), the evaluator must still work -- through vstab_flow_resize_scale + vstab_warp_flow
    w = wts.synthetic_weights(seed=3, cin=27, random_bn=True, flow_gain=2.0)
    runtime.reset()
    vs.assign_weights(w)
    g = torch.Generator().manual_seed(9)
    feats = torch.rand(2, 64, 64, 27, generator=g).cuda()
    big = torch.rand(3, 37, 53, 3, generator=g).cuda()
    frame = big[1:]
    assert frame.data_ptr() % 16 != 0 and frame.is_contiguous()
    flows, outflow, warped = vs.stabilise_originalsize(feats, frame)
    _, outflow_c, warped_c = vs.stabilise_originalsize(feats, frame.clone())
    assert torch.equal(outflow, outflow_c) and torch.equal(warped, warped_c)


def test_get_pixel_value():
    img = torch.rand(2, 9, 11, 3)
    x = torch.randint(0, 11, (2, 4, 5), dtype=torch.int32)
    y = torch.randint(0, 9, (2, 4, 5), dtype=torch.int32)
    out = vs.get_pixel_value(img.cuda(), x.cuda(), y.cuda())
    assert torch.equal(out.cpu(), vo.get_pixel_value(img, x, y))


@pytest.mark.parametrize("h,w,oh,ow", [(6, 8, 12, 16), (24, 32, 48, 64), (48, 64, 382, 510), (17, 30, 34, 60), (12, 12, 12, 12), (40, 40, 20, 13)])
def test_resize_images_vs_oracle(h, w, oh, ow):
    for C in (3, 2, 5):                      # 3 channels: the tiled kernel; other counts: the one-element-per-thread kernel
        x = torch.rand(2, h, w, C)
        out = vs.resize_images(x.cuda(), (oh, ow))
        ref = vo.resize_bilinear_legacy(x, oh, ow)
        assert maxabs(out, ref) <= 1e-6, C


@pytest.mark.parametrize("B,h,w,cs,c_off,oh,ow", [(2, 64, 96, 27, 24, 62, 94), (1, 48, 64, 27, 24, 46, 62), (3, 33, 41, 6, 3, 70, 52),
                                                  (1, 20, 24, 3, 0, 37, 53), (2, 40, 40, 5, 1, 20, 13)])
def test_resize_slice3_in_place_bit_identical(B, h, w, cs, c_off, oh, ow):
    # main:806 reads the unstable frame (channels 24:27) straight out of the 27-channel stack: same bits as resizing a copy
    x = torch.rand(B, h, w, cs, generator=torch.Generator().manual_seed(h * w + cs)).cuda()
    got = vs.resize_images_slice3(x, c_off, (oh, ow))
    ref = vs.resize_images(x[..., c_off:c_off + 3].contiguous(), (oh, ow))
    assert torch.equal(got, ref)
    assert maxabs(got, vo.resize_bilinear_legacy(x[..., c_off:c_off + 3].cpu(), oh, ow)) <= 1e-6


@pytest.mark.parametrize("hn,wn,oh,ow", [(384, 512, 384, 512), (64, 64, 64, 64), (70, 90, 96, 120), (64, 96, 32, 48), (50, 70, 37, 53),
                                         (12, 3, 9, 5)])      # a one-column flow: the tiled kernel's paired taps do not apply
def test_flow_glue_vs_oracle(hn, wn, oh, ow):
    pf2 = torch.randn(2, hn - 2, wn - 2, 2) * 5
    out = vs.flow_to_output_res(pf2.cuda(), hn, wn, oh, ow)
    ref = vo.flow_to_output_res(pf2, hn, wn, oh, ow)
    assert maxabs(out, ref) <= 2e-5


# ------------------------------------------------------------------------- network
def run_net(feats, w):
    runtime.reset()
    vs.assign_weights(w)
    out = vs.flownetS_pyramid(dev(feats), feats.shape[0], is_train=False)
    torch.cuda.synchronize()
    return out


def check_internals(feats, w, ref_int):
    B, H, W, Cin = feats.shape
    ints = runtime.get_context().internals(B, H, W, Cin)
    names = {"conv1": "conv1", "concat2": "concat2", "conv3": "conv3", "concat3": "concat3", "conv4": "conv4",
             "concat4": "concat4", "conv5": "conv5", "concat5": "concat5", "conv6": "conv6", "conv6_1": "conv6_1"}
    report = {}
    for mine, theirs in names.items():
        r = ref_int[theirs]
        err = maxabs(ints[mine], r)
        report[mine] = (err, float(r.abs().max()))
    return report


# sizes chosen to hit every first-layer path: row-window kernel (W*Cin % 4 == 0, Cin 27 and 6),
# generic dword-gather kernel (70x90x27: W*Cin % 4 == 2) and the generic 16-byte kernel (Cin 28)
@pytest.mark.parametrize("B,H,W,cin,seed", [(2, 64, 64, 27, 3), (1, 88, 104, 27, 4), (1, 70, 90, 6, 5), (3, 48, 64, 28, 6),
                                            (1, 70, 90, 27, 7), (2, 136, 264, 27, 8),
                                            # one sample at BASELINE configs[0]'s size and at the reference's own (main:540-630): the round-4
                                            # schedule -- weight-stream layers, two-problem launches, 4x4 predict_up tiles -- layer by layer
                                            (1, 256, 256, 27, 9), (1, 384, 512, 27, 10)])
def test_network_every_layer_vs_oracle(B, H, W, cin, seed):
    w = wts.synthetic_weights(seed=seed, cin=cin, random_bn=True, flow_gain=2.0)
    feats = np.random.default_rng(seed).random((B, H, W, cin), dtype=np.float32)
    ref, ref_int = vo.flownetS_pyramid(feats, w, torch.float64, return_internals=True)
    out = run_net(feats, w)
    report = check_internals(feats, w, ref_int)
    bad = {k: v for k, v in report.items() if v[0] > 2e-4 * max(1.0, v[1])}
    assert not bad, f"layer errors (max-abs err, max-abs ref): {report}"
    errs = {k: maxabs(out[k], ref[k]) for k in vo.FLOW_KEYS}
    assert all(e <= FLOW_TOL for e in errs.values()), errs
    assert out["flow"] is out["predict_flow2"]
    assert float(ref["predict_flow2"].abs().max()) > 0.5


@pytest.mark.parametrize("name", ["net_48x64_c27", "net_70x90_c6"])
def test_network_golden_end_to_end(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    B, H, W, Cin, oh, ow, seed, rbn = [int(v) for v in z["meta"]]
    w = wts.synthetic_weights(seed=seed, cin=Cin, random_bn=bool(rbn), flow_gain=float(z["flow_gain"]))
    runtime.reset()
    vs.assign_weights(w)
    flows, outflow, warped = vs.stabilise_originalsize(dev(z["feats"]), dev(z["frame"]))
    torch.cuda.synchronize()
    for k in vo.FLOW_KEYS:
        assert maxabs(flows[k], z[k]) <= FLOW_TOL, k
    assert maxabs(outflow, z["outflow"]) <= FLOW_TOL
    # warped frame: compare away from tf_warp's discontinuity lines
    ok = vo.warp_discontinuity_mask(torch.from_numpy(z["outflow"]), oh, ow, delta=5e-3)
    diff = (warped.cpu() - torch.from_numpy(z["warped"])).abs().amax(dim=3)
    assert float(diff[ok].max()) <= 2e-3
    assert ok.float().mean() > 0.9


def test_cfg0_256x256_single_pair_vs_fp64_oracle():
    # BASELINE.json configs[0]: one 256x256 synthetic sample through network + glue + warp
    w = wts.synthetic_weights(seed=1, cin=27, random_bn=False)
    g = np.random.default_rng(0)
    feats = g.random((1, 256, 256, 27), dtype=np.float32)
    ref = vo.flownetS_pyramid(feats, w, torch.float64)
    out = run_net(feats, w)
    errs = {k: maxabs(out[k], ref[k]) for k in vo.FLOW_KEYS}
    assert all(e <= FLOW_TOL for e in errs.values()), errs


def test_native_path_matches_oracle():
    w = wts.synthetic_weights(seed=9, cin=27, random_bn=True, flow_gain=2.0)
    feats = np.random.default_rng(9).random((1, 64, 80, 27), dtype=np.float32)
    runtime.reset()
    vs.assign_weights(w)
    flows, warped = vs.stabilise_native(dev(feats))
    rflows, rwarped = vo.stabilise_native(feats, w, torch.float64)
    assert maxabs(flows["predict_flow2"], rflows["predict_flow2"]) <= FLOW_TOL
    ok = vo.warp_discontinuity_mask(rflows["predict_flow2"].float(), 62, 78, delta=5e-3)
    diff = (warped.cpu().double() - rwarped).abs().amax(dim=3)
    assert float(diff[ok].max()) <= 2e-3


def test_batch_independence_and_determinism():
    # samples are independent (no cross-sample op at inference): sharding by sample is exact
    w = wts.synthetic_weights(seed=2, cin=27, random_bn=True)
    feats = np.random.default_rng(2).random((4, 64, 64, 27), dtype=np.float32)
    runtime.reset()
    vs.assign_weights(w)
    full = vs.flownetS_pyramid(dev(feats), 4)["predict_flow2"].clone()
    again = vs.flownetS_pyramid(dev(feats), 4)["predict_flow2"].clone()
    assert torch.equal(full, again)
    halves = torch.cat([vs.flownetS_pyramid(dev(feats[:2]), 2)["predict_flow2"].clone(),
                        vs.flownetS_pyramid(dev(feats[2:]), 2)["predict_flow2"].clone()])
    # split-K factors may differ with the batch -> not bitwise (vstab_set_plan_batch pins them: test_gpu_skinny.py); two plans for the same
    # layers differ by a few fp32 epsilons of the flow itself (profiles/flow_err_margin_r05.md: <= 11.9 measured, 16 allowed)
    assert maxabs(full, halves.cpu()) <= 16 * 1.1920929e-07 * max(1.0, float(full.abs().max()))


def test_errors_through_the_abi():
    runtime.reset()
    with pytest.raises(RuntimeError, match="weights"):
        vs.flownetS_pyramid(torch.zeros(1, 64, 64, 27, device="cuda"), 1)
    vs.initialize_global_variables(seed=1, cin=27)
    with pytest.raises(ValueError):
        vs.flownetS_pyramid(torch.zeros(1, 64, 64, 6, device="cuda"), 1)      # channel mismatch
    with pytest.raises(ValueError):
        vs.flownetS_pyramid(torch.zeros(1, 2, 2, 27, device="cuda"), 1)       # too small
    with pytest.raises(ValueError):
        vs.tf_warp(torch.zeros(1, 4, 4, 3, device="cuda"), torch.zeros(1, 4, 5, 2, device="cuda"), 4, 4)


def test_new_entry_points_reject_bad_arguments():
    """The round-2 entry points return VSTAB_E_* (never fault, never throw across the ABI) for arguments outside their contract."""
    import ctypes as C
    from coupe.optical_flow_based_deep_video_stabilization_amd import _lib
    L = _lib.lib()
    st = runtime.stream_ptr()
    flow = torch.zeros(1, 8, 8, 2, device="cuda")
    img3, img4 = torch.zeros(1, 8, 8, 3, device="cuda"), torch.zeros(1, 8, 8, 4, device="cuda")
    of, out = torch.zeros(1, 8, 8, 2, device="cuda"), torch.zeros(1, 8, 8, 3, device="cuda")
    ok = L.vstab_flow_glue_warp(flow.data_ptr(), 1, 8, 8, img3.data_ptr(), of.data_ptr(), out.data_ptr(), 8, 8, 3, 10, 10, st)
    assert ok == 0
    assert L.vstab_flow_glue_warp(None, 1, 8, 8, img3.data_ptr(), of.data_ptr(), out.data_ptr(), 8, 8, 3, 10, 10, st) == -6      # NULL
    assert L.vstab_flow_glue_warp(flow.data_ptr(), 1, 8, 8, img4.data_ptr(), of.data_ptr(), out.data_ptr(), 8, 8, 4, 10, 10, st) == -1   # C != 3
    assert L.vstab_flow_glue_warp(flow.data_ptr(), 1, 8, 1, img3.data_ptr(), of.data_ptr(), out.data_ptr(), 8, 8, 3, 10, 10, st) == -1   # one-column flow
    assert L.vstab_flow_glue_warp(flow.data_ptr(), 0, 8, 8, img3.data_ptr(), of.data_ptr(), out.data_ptr(), 8, 8, 3, 10, 10, st) == -1   # empty batch
    assert L.vstab_flow_glue_warp(flow.data_ptr(), 1, 8, 8, img3.data_ptr(), of.data_ptr(), out.data_ptr() + 4, 8, 8, 3, 10, 10, st) == -2   # alignment
    assert b"flow_glue_warp" in L.vstab_last_error(None)
    x = torch.zeros(1, 8, 8, 27, device="cuda")
    assert L.vstab_resize_bilinear_slice3(x.data_ptr(), 1, 8, 8, 27, 24, out.data_ptr(), 8, 8, st) == 0
    assert L.vstab_resize_bilinear_slice3(x.data_ptr(), 1, 8, 8, 27, 25, out.data_ptr(), 8, 8, st) == -1        # slice runs past the pixel
    assert L.vstab_resize_bilinear_slice3(x.data_ptr(), 1, 8, 8, 2, 0, out.data_ptr(), 8, 8, st) == -1
    assert L.vstab_resize_bilinear_slice3(None, 1, 8, 8, 27, 24, out.data_ptr(), 8, 8, st) == -6
    # the Python shims turn them into exceptions, and fall back where the reference's semantics allow it
    with pytest.raises(ValueError):
        vs.flow_glue_warp(flow, img4, 10, 10)
    f1 = torch.randn(1, 6, 1, 2, device="cuda")                      # a 1-pixel-wide flow: stabilise path uses the two-launch form
    fr = torch.rand(1, 5, 7, 3, device="cuda")
    o1 = vs.flow_to_output_res(f1, 8, 3, 5, 7)
    assert torch.equal(vs.tf_warp(fr, o1, 5, 7).cpu(), vo.tf_warp(fr.cpu(), o1.cpu(), 5, 7, torch.float32))
    torch.cuda.synchronize()


def test_round3_entry_points_reject_bad_arguments():
    """vstab_stabilise_originalsize, the glue's new (net_h, net_w) arguments, vstab_st_bilinear_interp(oh, ow) and the division
    self-test: VSTAB_E_* for arguments outside their contract, through the ABI."""
    import ctypes as C
    L = _lib.lib()
    st = runtime.stream_ptr()
    w = wts.synthetic_weights(seed=3, cin=27, random_bn=False)
    runtime.reset()
    vs.assign_weights(w)
    ctx = runtime.get_context()
    B, H, W = 1, 64, 64
    feats, frame = torch.rand(B, H, W, 27, device="cuda"), torch.rand(B, H, W, 3, device="cuda")
    stab = vs.OriginalSizeStabiliser(B, H, W, 27, H, W)
    pf = [f.data_ptr() for f in stab.flows]
    ws = stab.ws
    args = lambda fe=feats.data_ptr(), fr=frame.data_ptr(), wp=stab.warped.data_ptr(), oh=H, ow=W, wsb=ws.numel(): (
        ctx._h, fe, B, H, W, 27, fr, oh, ow, *pf, stab.outflow.data_ptr(), wp, ws.data_ptr(), wsb, st)
    assert L.vstab_stabilise_originalsize(*args()) == 0
    assert L.vstab_stabilise_originalsize(*args(fr=None)) == -6                     # NULL frame
    assert L.vstab_stabilise_originalsize(*args(wp=None)) == -6                     # NULL output
    assert L.vstab_stabilise_originalsize(*args(fe=None)) == -6                     # NULL feats (caught by the forward)
    assert L.vstab_stabilise_originalsize(*args(oh=0)) == -1                        # empty output
    assert L.vstab_stabilise_originalsize(*args(wsb=1024)) < 0                      # workspace too small
    assert L.vstab_stabilise_originalsize(None, *args()[1:]) == -6                  # no context
    flow = torch.zeros(1, 8, 8, 2, device="cuda")
    out = torch.zeros(1, 8, 8, 2, device="cuda")
    assert L.vstab_flow_resize_scale(flow.data_ptr(), 1, 8, 8, out.data_ptr(), 8, 8, 10, 10, st) == 0
    assert L.vstab_flow_resize_scale(flow.data_ptr(), 1, 8, 8, out.data_ptr(), 8, 8, 0, 10, st) == -1      # net_h < 1
    img = torch.zeros(1, 8, 8, 3, device="cuda")
    xy = torch.zeros(16, device="cuda")
    o = torch.zeros(16, 3, device="cuda")
    assert L.vstab_st_bilinear_interp(img.data_ptr(), 1, 8, 8, 3, xy.data_ptr(), xy.data_ptr(), 4, 4, o.data_ptr(), st) == 0
    assert L.vstab_st_bilinear_interp(img.data_ptr(), 1, 8, 8, 3, xy.data_ptr(), xy.data_ptr(), 0, 4, o.data_ptr(), st) == -1
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    assert L.vstab_selftest_div_const(C.c_float(382.0), 0, 0, bad.data_ptr(), st) == -1                  # empty range
    assert L.vstab_selftest_div_const(C.c_float(382.0), 0, 16, None, st) == -6
    torch.cuda.synchronize()


# ------------------------------------------------------------------------- predict_flow2 gather (K9 / F8) on its own
def _pf2_reference(T, bias2, pf3, H, W):
    """pf2[y,x,o] = b[o] + sum_{dy,dx} T[ny(y+dy)-1, nx(x+dx)-1][3dy+dx][o] over the in-image taps, then eight sequential adds of
    the legacy-bilinear upsampled pf3 (model.py:882-887; index map SURVEY.md A.4), in fp32 and in the kernel's order."""
    B, h2, w2, _ = T.shape
    iy = vo.nearest_align_corners_index(h2 + 2, H) - 1
    ix = vo.nearest_align_corners_index(w2 + 2, W) - 1
    oh, ow = H - 2, W - 2
    acc = torch.from_numpy(np.asarray(bias2, np.float32)).view(1, 1, 1, 2).expand(B, oh, ow, 2).clone()
    Tp = torch.zeros(B, h2 + 2, w2 + 2, 32)
    Tp[:, 1:-1, 1:-1] = T                                      # zero ring: a skipped tap adds +0.0
    for dy in range(3):
        for dx in range(3):
            ys = torch.from_numpy(iy[dy:dy + oh] + 1)
            xs = torch.from_numpy(ix[dx:dx + ow] + 1)
            t = Tp[:, ys][:, :, xs][..., (3 * dy + dx) * 2:(3 * dy + dx) * 2 + 2]
            acc = acc + t
    up = vo.resize_bilinear_legacy(pf3, oh, ow)
    for _ in range(8):
        acc = acc + up
    return acc


@pytest.mark.parametrize("B,h2,w2,H,W", [(2, 24, 32, 96, 128),        # the network's own geometry: LDS-staged window per 16x64 tile
                                         (1, 13, 17, 50, 66),         # odd sizes, ragged tiles
                                         (1, 64, 64, 34, 34),         # downsampling head: the tile's window exceeds the LDS budget ->
                                         (2, 40, 90, 20, 70)])        # ... the direct-gather kernel runs instead
def test_pf2_gather_tiled_and_direct_kernels(B, h2, w2, H, W):
    import ctypes as C
    from coupe.optical_flow_based_deep_video_stabilization_amd import _lib
    g = torch.Generator().manual_seed(h2 * 100 + W)
    T = torch.randn(B, h2, w2, 32, generator=g)
    h3, w3 = (h2 + 1) // 2, (w2 + 1) // 2
    pf3 = torch.randn(B, h3, w3, 2, generator=g) * 3
    bias2 = torch.tensor([0.25, -0.5])
    out = torch.empty(B, H - 2, W - 2, 2, device="cuda")
    Td, pd, bd = T.cuda(), pf3.cuda(), bias2.cuda()
    _lib.check(_lib.lib().vstab_pf2_from_taps(Td.data_ptr(), B, h2, w2, bd.data_ptr(), pd.data_ptr(), h3, w3, out.data_ptr(), H, W,
                                              runtime.stream_ptr()))
    ref = _pf2_reference(T, bias2.numpy(), pf3, H, W)
    assert maxabs(out, ref) <= 1e-5 * max(1.0, float(ref.abs().max()))
