"""The checkpoint loader (main:520 `tl.files.load_and_assign_npz_dict`) against a file laid out the way the reference's training script
writes one (main:329-330, 424-426; names and shapes transcribed from model.py:805-887 by tests/golden/make_tl_checkpoint.py, which does not
use the package's own tables).  CPU only: loading registers the variables; packing happens when a context is created on a GPU."""
import importlib.util
import os

import numpy as np
import pytest

import coupe.optical_flow_based_deep_video_stabilization_amd as vs
from coupe.optical_flow_based_deep_video_stabilization_amd import netspec, runtime, weights as wts

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_tl_checkpoint", os.path.join(HERE, "golden", "make_tl_checkpoint.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("ckpt") / "flownetS_pyramid.npz")
    return p, mk.write(p, cin=27, seed=3)


def test_reference_style_checkpoint_loads_by_name(ckpt):
    path, written = ckpt
    names = [n for n, _ in written]
    # the file is what main:424-426 writes: 5 variables per BatchNorm-ed conv, 2 per head / flow deconv, none of the optimiser's
    assert len(names) == 10 * 5 + 5 * 2 + 4 * (2 + 3) + 4 * 2 == 88
    assert names[0] == "main_net/flownetS/1/W_conv2d:0" and names[2] == "main_net/flownetS/1/beta:0"
    assert "main_net/flownetS/deconv5_bn/moving_variance:0" in names and "main_net/flownetS/upsample3_2/W_deconv2d:0" in names
    assert not any("gamma" in n or "Adam" in n for n in names)
    assert sum(a.size for _, a in written) == sum(int(np.prod(s_)) for s_ in netspec.weight_shapes(27).values())       # 38.7 M parameters
    runtime._pending_weights.clear()
    w = vs.load_and_assign_npz_dict(path)
    assert set(w) == set(netspec.weight_shapes(27))                          # every variable of the graph, nothing else
    for full, a in written:
        short = full[len("main_net/flownetS/"):-2]
        assert w[short].dtype == np.float32 and np.array_equal(w[short], a), full
    assert w["deconv5/W_deconv2d"].shape == (4, 4, 512, 1024)                # [kh, kw, Cout, Cin]: model.py:850's shape= argument
    assert w["1/W_conv2d"].shape == (7, 7, 27, 64) and w["predict4/W_conv2d"].shape == (3, 3, 770, 2)
    assert runtime._pending_weights["flownetS"]["6_1/moving_mean"].shape == (1024,)
    runtime._pending_weights.clear()


def test_reference_style_checkpoint_rejects_wrong_layouts(ckpt, tmp_path):
    path, written = ckpt
    d = dict(written)
    # a transposed-convolution filter stored [kh,kw,Cin,Cout] (the conv layout) instead of [kh,kw,Cout,Cin]
    bad = dict(d)
    bad["main_net/flownetS/deconv4/W_deconv2d:0"] = np.ascontiguousarray(d["main_net/flownetS/deconv4/W_deconv2d:0"].transpose(0, 1, 3, 2))
    p = str(tmp_path / "transposed.npz")
    np.savez(p, **bad)
    with pytest.raises(ValueError, match="deconv4/W_deconv2d"):
        vs.load_and_assign_npz_dict(p)
    # a conv filter stored OIHW
    bad = dict(d)
    bad["main_net/flownetS/3/W_conv2d:0"] = np.ascontiguousarray(d["main_net/flownetS/3/W_conv2d:0"].transpose(3, 2, 0, 1))
    p = str(tmp_path / "oihw.npz")
    np.savez(p, **bad)
    with pytest.raises(ValueError, match="3/W_conv2d"):
        vs.load_and_assign_npz_dict(p)
    # a missing variable (a checkpoint of a network without the BatchNorm moving statistics)
    bad = {k: v for k, v in d.items() if not k.endswith("deconv3_bn/moving_mean:0")}
    p = str(tmp_path / "missing.npz")
    np.savez(p, **bad)
    with pytest.raises(KeyError, match="deconv3_bn/moving_mean"):
        vs.load_and_assign_npz_dict(p)
    # another scope (main:183 builds under 'main_net'; a file written under a different model scope must not load silently)
    bad = {k.replace("/flownetS/", "/other_net/"): v for k, v in d.items()}
    p = str(tmp_path / "scope.npz")
    np.savez(p, **bad)
    with pytest.raises(KeyError):
        vs.load_and_assign_npz_dict(p)
    # extra variables are ignored (e.g. a file saved after the optimiser was created would carry Adam slots)
    ok = dict(d)
    ok["Optimizer/main_net/flownetS/1/W_conv2d/Adam:0"] = np.zeros((7, 7, 27, 64), np.float32)
    p = str(tmp_path / "extra.npz")
    np.savez(p, **ok)
    w = wts.load_npz_dict(p)
    assert np.array_equal(w["1/W_conv2d"], d["main_net/flownetS/1/W_conv2d:0"])
    runtime._pending_weights.clear()
