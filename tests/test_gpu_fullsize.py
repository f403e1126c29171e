"""GPU parity at BASELINE.json's full sizes.  The CPU oracle needs seconds per sample there, so
each case checks ONE sample against the oracle and the rest of the batch through
size-independent properties: samples are independent (no cross-sample op at inference), so
copies of the same input must give bit-identical outputs wherever they sit in the batch --
including on both sides of the internal 2 GiB chunk boundary -- and runs are deterministic."""
import numpy as np
import pytest
import torch

import coupe.optical_flow_based_deep_video_stabilization_amd as vs
from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, runtime, weights as wts
from oracle import vstab_oracle as vo

pytestmark = pytest.mark.gpu
FLOW_TOL = 1e-3


EPS32 = 1.1920929e-07


def _run_case(B, H, W, cin, oracle_dtype, wkw=None, rel_eps=None):
    """wkw: keyword arguments of the weight set (default: He-normal filters, identity BatchNorm -- what bench.py times).
    rel_eps: None = the absolute 1e-3 gate; k = |err(pf_k)| <= k * eps32 * max|pf_k| per level instead (flows wider than the image:
    the error is a relative one)."""
    w = wts.synthetic_weights(cin=cin, **(wkw or dict(seed=1, random_bn=False)))
    runtime.reset()
    vs.assign_weights(w)
    rng = np.random.default_rng(H + W)
    two = rng.random((2, H, W, cin), dtype=np.float32)
    pattern = np.array([i % 2 for i in range(B)])
    pattern[-1] = 0                                    # first and last sample are the same input
    feats = torch.from_numpy(two).cuda()[torch.from_numpy(pattern).cuda()]
    frame = torch.rand(B, H, W, 3, device="cuda")
    flows, outflow, warped = vs.stabilise_originalsize(feats, frame)
    torch.cuda.synchronize()
    pf2 = flows["predict_flow2"]
    assert pf2.shape == (B, H - 2, W - 2, 2) and torch.isfinite(pf2).all()
    for k in vo.FLOW_KEYS:                             # copies agree bit for bit, across (equal) chunks too
        f = flows[k]
        for i in range(B):
            assert torch.equal(f[i], f[int(pattern[i])]), (k, i)
    again = vs.flownetS_pyramid(feats, B)["predict_flow2"]
    assert torch.equal(again, pf2)                     # deterministic
    ref = vo.flownetS_pyramid(two[:1], w, oracle_dtype)
    errs = {k: float((flows[k][0].double().cpu() - ref[k][0].double()).abs().max()) for k in vo.FLOW_KEYS}
    if rel_eps is None:
        assert all(e <= FLOW_TOL for e in errs.values()), errs
    else:
        mags = {k: float(ref[k].abs().max()) for k in vo.FLOW_KEYS}
        in_eps = {k: errs[k] / (EPS32 * max(1.0, mags[k])) for k in vo.FLOW_KEYS}
        assert all(v <= rel_eps for v in in_eps.values()), (in_eps, errs, mags)
    # glue + warp of sample 0 against the oracle run on the GPU's own flow (isolates W1/G1)
    of_ref = vo.flow_to_output_res(pf2[:1].cpu(), H, W, H, W)
    assert float((outflow[:1].cpu() - of_ref).abs().max()) <= 2e-5
    wr = vo.tf_warp(frame[:1].cpu(), outflow[:1].cpu(), H, W, torch.float32)
    assert float((warped[:1].cpu() - wr).abs().max()) <= 1e-6
    return errs


def test_cfg1_batch8_512x512():
    _run_case(8, 512, 512, 27, torch.float64)


def test_cfg2_batch32_720p_chunked():
    L = _lib.lib()
    # the batch does not fit under the 2 GiB tensor limit in one piece -> two chunks of 16
    assert L.vstab_workspace_bytes(32, 720, 1280, 27) == L.vstab_workspace_bytes(21, 720, 1280, 27)
    assert L.vstab_workspace_bytes(22, 720, 1280, 27) == L.vstab_workspace_bytes(21, 720, 1280, 27)
    _run_case(32, 720, 1280, 27, torch.float32)


def test_unequal_chunks_are_bit_equal_under_a_pinned_plan():
    # 23 samples at 720p -> chunks of 12 and 11.  Unpinned, their split-K plans may differ (same values up to fp32 summation order,
    # far inside the flow tolerance); with the plan pinned to the batch (vstab_set_plan_batch) both chunks take the decisions of a
    # chunk of 12 and every copy of the sample comes out with the same bits
    w = wts.synthetic_weights(seed=1, cin=27, random_bn=False)
    runtime.reset()
    vs.assign_weights(w)
    one = torch.rand(1, 720, 1280, 27, device="cuda")
    feats = one.expand(23, -1, -1, -1).contiguous()
    pf2 = vs.flownetS_pyramid(feats, 23)["predict_flow2"]
    assert float((pf2 - pf2[:1]).abs().max()) <= 5e-4          # unpinned: half the 1e-3 flow tolerance
    runtime.get_context().set_plan_batch(23)
    pf2 = vs.flownetS_pyramid(feats, 23)["predict_flow2"]
    assert torch.equal(pf2, pf2[:1].expand_as(pf2))
    lone = vs.flownetS_pyramid(one, 1)["predict_flow2"]         # and a lone sample under the same pin
    assert torch.equal(lone, pf2[:1])
    runtime.reset()


def test_cfg3_1080p_samples():
    # cfg3/cfg4 resolution (1080x1920); a short batch keeps the CPU side of the test bounded
    _run_case(3, 1080, 1920, 27, torch.float32)


def test_cfg2_full_pipeline_ends_in_the_spatial_transformer_warp():
    """BASELINE configs[2] as one pipeline at its stated size: batch 32 of 720 x 1280 x 27 -> FlowNetS pyramid (two internal chunks of 16)
    -> flow glue + tf_warp -> AffineTransformer on the stabilised frames (spatial_transformer.py:400-452).  Copies of an input agree
    bit for bit wherever they sit in the batch; sample 0's last stage is checked against the fp32 oracle on the GPU's own stabilised frame."""
    from coupe.optical_flow_based_deep_video_stabilization_amd import spatial_transformer as st
    B, H, W = 32, 720, 1280
    w = wts.synthetic_weights(seed=1, cin=27, random_bn=False)
    runtime.reset()
    vs.assign_weights(w)
    rng = np.random.default_rng(2)
    two = torch.from_numpy(rng.random((2, H, W, 27), dtype=np.float32)).cuda()
    fr2 = torch.from_numpy(rng.random((2, H, W, 3), dtype=np.float32)).cuda()
    pattern = torch.tensor([i % 2 for i in range(B)])
    pattern[-1] = 0
    idx = pattern.cuda()
    feats, frame = two[idx], fr2[idx]
    th2 = torch.tensor([[1.01, 0.02, 0.01, -0.02, 0.99, -0.015], [0.98, -0.03, -0.02, 0.03, 1.02, 0.01]])
    theta = th2[pattern].cuda()
    stab = vs.OriginalSizeStabiliser(B, H, W, 27, H, W)
    flows, outflow, warped = stab(feats, frame)
    out = st.AffineTransformer((H, W)).transform(warped, theta)
    torch.cuda.synchronize()
    assert out.shape == (B, H, W, 3) and torch.isfinite(out).all()
    for i in range(B):
        assert torch.equal(out[i], out[int(pattern[i])]) and torch.equal(warped[i], warped[int(pattern[i])]), i
    assert torch.equal(out[:1].cpu(), vo.st_transform(warped[:1].cpu(), th2[:1], (H, W), matmul="unfused"))


# ---- the parity envelope at full size ("what's weak" 2).  The cases above use the weight set bench.py times (0.22-0.27 of
# the 1e-3 budget).  These pin the harder sets of profiles/flow_err_margin_r05.md at BASELINE's own sizes, against the fp64 restatement:
# random BatchNorm statistics with flow_gain 1 (flows of ~200 px; 0.56-0.62 of the budget) under the ABSOLUTE gate, and flow_gain 2
# (flows of 420-446 px, wider than the image) under the RELATIVE bound the error really follows -- 32 fp32 epsilons of the level's
# largest flow (measured 14-21) -- because an absolute 1e-3 on a 420 px flow is 20 eps: below any fp32 evaluation's noise floor (the
# torch-CPU fp32 restatement is off by 1.84e-3 on that cell).  A regression in summation order shows up in either.
GAIN1 = dict(seed=2, random_bn=True, flow_gain=1.0)
GAIN2 = dict(seed=1, random_bn=True, flow_gain=2.0)


def test_cfg1_random_bn_gain1_absolute_gate():
    errs = _run_case(8, 512, 512, 27, torch.float64, GAIN1)
    assert max(errs.values()) <= 0.8e-3, errs              # measured 0.62e-3: the headroom itself is pinned


def test_cfg2_size_random_bn_gain1_absolute_gate():
    errs = _run_case(32, 720, 1280, 27, torch.float64, GAIN1)
    assert max(errs.values()) <= 0.8e-3, errs              # measured 0.56e-3


def test_cfg4_size_random_bn_gain1_absolute_gate():
    _run_case(16, 1080, 1920, 27, torch.float64, GAIN1)


def test_cfg1_random_bn_gain2_relative_bound():
    _run_case(8, 512, 512, 27, torch.float64, GAIN2, rel_eps=32)


def test_720p_random_bn_gain2_relative_bound():
    _run_case(4, 720, 1280, 27, torch.float64, GAIN2, rel_eps=32)
