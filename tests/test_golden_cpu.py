"""CPU: the oracle reproduces the committed golden vectors; the C-ABI library loads and
exports every symbol include/vstab.h declares; host-side error paths."""
import os
import re

import numpy as np
import pytest
import torch

from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, netspec, weights as wts
from oracle import vstab_oracle as vo

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    B, H, W, Cin, oh, ow, seed, rbn = [int(v) for v in z["meta"]]
    w = wts.synthetic_weights(seed=seed, cin=Cin, random_bn=bool(rbn), flow_gain=float(z["flow_gain"]))
    return z, w


@pytest.mark.parametrize("name", ["net_48x64_c27", "net_70x90_c6"])
def test_oracle_matches_golden(name):
    z, w = load_case(name)
    flows, outflow, warped = vo.stabilise_originalsize(z["feats"], z["frame"], w, torch.float64)
    for k in vo.FLOW_KEYS:
        assert np.abs(flows[k].numpy() - z[k]).max() < 1e-5
    assert np.abs(outflow.numpy() - z["outflow"]).max() < 1e-5
    assert np.abs(flows["predict_flow2"].numpy()).max() > 1.0       # flows of pixel magnitude


def test_oracle_fp32_within_tolerance_of_fp64():
    # the 1e-3 budget of BASELINE.json is comfortably above fp32 rounding noise of this net
    z, w = load_case("net_48x64_c27")
    f32 = vo.flownetS_pyramid(z["feats"], w, torch.float32)
    for k in vo.FLOW_KEYS:
        assert np.abs(f32[k].double().numpy() - z[k]).max() < 3e-4


def test_warp_golden():
    z = np.load(os.path.join(GOLD, "warp_37x53.npz"))
    out = vo.tf_warp(torch.from_numpy(z["img"]), torch.from_numpy(z["flow"]), 37, 53, torch.float32).numpy()
    assert np.array_equal(out, z["warped"])


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vstab.h")).read()
    declared = set(re.findall(r"VSTAB_API[^;(]*?\b(vstab_\w+)\s*\(", hdr))
    assert len(declared) >= 15
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert L.vstab_version().startswith(b"vstab-hip")


def test_no_cpu_fallback_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import coupe.optical_flow_based_deep_video_stabilization_amd as vs
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vs.flownetS_pyramid(torch.zeros(1, 64, 64, 27), 1)
    with pytest.raises(ValueError):
        vs.tf_warp(torch.zeros(1, 4, 4, 3), torch.zeros(1, 4, 4, 2), 4, 4)   # CPU tensors are rejected


def test_is_train_and_batch_size_errors():
    import coupe.optical_flow_based_deep_video_stabilization_amd as vs
    from coupe.optical_flow_based_deep_video_stabilization_amd import runtime
    runtime.reset()
    with pytest.raises(RuntimeError):                      # the training graph needs variables first (and a GPU: no CPU path)
        vs.flownetS_pyramid(torch.zeros(1, 64, 64, 27), 1, is_train=True)
    with pytest.raises(ValueError):
        vs.flownetS_pyramid(torch.zeros(2, 64, 64, 27), 1)


def test_checkpoint_key_roundtrip(tmp_path):
    w = wts.synthetic_weights(seed=2, cin=6)
    p = str(tmp_path / "ckpt.npz")
    wts.save_npz_dict(p, w)
    with np.load(p) as z:
        assert "main_net/flownetS/1/W_conv2d:0" in z.files and "main_net/flownetS/deconv5_bn/moving_variance:0" in z.files
    back = wts.load_npz_dict(p)
    assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w)
    bad = dict(w); bad.pop("predict4/b_conv2d")
    with pytest.raises(KeyError):
        wts.validate(bad)
    bad = dict(w); bad["3/W_conv2d"] = bad["3/W_conv2d"][..., :-1]
    with pytest.raises(ValueError):
        wts.validate(bad)


def test_sizes_rule():
    assert netspec.sizes_for(384, 512).level == {1: (192, 256), 2: (96, 128), 3: (48, 64), 4: (24, 32), 5: (12, 16), 6: (6, 8)}
    assert netspec.sizes_for(1080, 1920).level[4] == (68, 120)
    with pytest.raises(ValueError):
        netspec.sizes_for(2, 2)
    assert abs(netspec.gflop_per_sample(512, 512) - 52.43) < 0.01


def test_widening_golden_oracle():
    """The oracle reproduces the committed vectors of the rows built around the path (tests/golden/make_golden.py: make_widening)."""
    z = np.load(os.path.join(GOLD, "widening_small.npz"))
    assert np.array_equal(vo.cv_resize_u8(z["img_u8"], 48, 64), z["resize_u8_48x64"])
    assert np.array_equal(vo.cv_resize_f32(z["img_f32"], 48, 64), z["resize_f32_48x64"])
    M, n = vo.homography_fit(z["homo_flow"], K=64, seed=9, thresh=3.0, refine=2)
    assert n == int(z["homo_inliers"]) and np.abs(M - z["homo_H"]).max() < 1e-10
    assert np.array_equal(vo.cv_warp_perspective_u8(z["homo_frame"], z["homo_H"], 48, 64), z["homo_warped"])
    flows = {k: torch.from_numpy(z["loss_flow_" + k]) for k in vo.LOSS_LEVELS}
    loss = vo.loss_main(flows, torch.from_numpy(z["loss_gt"]), torch.from_numpy(z["loss_un"]))
    assert abs(float(loss) - float(z["loss_value"])) < 1e-10
