"""The checkpoint loader on the GPU against a file in the REFERENCE's layout (missing 4): the file is what the reference's
training script writes (main:329-330, 424-426 -- full `main_net/flownetS/<layer>/<leaf>:0` names, HWIO conv filters, [kh,kw,Cout,Cin]
transposed-conv filters; names and shapes transcribed from model.py:805-887 by tests/golden/make_tl_checkpoint.py, which does not use the
package's own tables).  It goes through `load_and_assign_npz_dict` (main:520) -> BatchNorm fold -> MFMA operand packing -> forward, and
all five flows are compared with the fp64 restatement fed the SAME arrays keyed by their reference names (prefix / ':0' stripped here,
not by the loader)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import coupe.optical_flow_based_deep_video_stabilization_amd as vs
from coupe.optical_flow_based_deep_video_stabilization_amd import model, runtime
from oracle import vstab_oracle as vo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_tl_checkpoint", os.path.join(HERE, "golden", "make_tl_checkpoint.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)


@pytest.mark.parametrize("shape", [(1, 96, 128), (2, 64, 96)])
def test_reference_layout_checkpoint_forward_matches_oracle(tmp_path, shape):
    B, H, W = shape
    path = str(tmp_path / "flownetS_pyramid.npz")
    written = mk.write(path, cin=27, seed=11)
    by_ref_name = {}
    for full, a in written:
        assert full.startswith("main_net/flownetS/") and full.endswith(":0"), full
        by_ref_name[full[len("main_net/flownetS/"):-2]] = a
    runtime.reset()
    model.load_and_assign_npz_dict(path)                                   # main:520
    rng = np.random.default_rng(5)
    feats = rng.random((B, H, W, 27), dtype=np.float32)
    out = model.flownetS_pyramid(torch.from_numpy(feats).cuda(), B, is_train=False)
    torch.cuda.synchronize()
    ref = vo.flownetS_pyramid(feats, by_ref_name, torch.float64)
    for k in vo.FLOW_KEYS:
        err = float((out[k].double().cpu() - ref[k]).abs().max())
        assert err <= 1e-3, (k, err)
        assert float(ref[k].abs().max()) > 1e-2, k                         # the comparison is not one of zeros
    assert torch.equal(out["flow"], out["predict_flow2"])                  # model.py:889
    # a transposed deconv filter (the conv layout) must be refused, not silently packed
    bad = dict(written)
    bad["main_net/flownetS/deconv3/W_deconv2d:0"] = np.ascontiguousarray(bad["main_net/flownetS/deconv3/W_deconv2d:0"].transpose(0, 1, 3, 2))
    p2 = str(tmp_path / "bad.npz")
    np.savez(p2, **bad)
    with pytest.raises(ValueError, match="deconv3/W_deconv2d"):
        model.load_and_assign_npz_dict(p2)
    runtime.reset()
