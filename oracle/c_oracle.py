"""ctypes binding of oracle/libvstab_oracle.so (plain-C restatement, double precision).
TEST INFRASTRUCTURE ONLY -- see vstab_oracle.c."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvstab_oracle.so")

ENC_NAMES = ("1", "2", "3", "3_1", "4", "4_1", "5", "5_1", "6", "6_1")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vstab_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libvstab_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.vo_nearest_index.restype = C.c_int
    return _lib


def _d(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def weight_pointer_list(weights):
    """88 arrays in the order vo_flownetS_pyramid documents."""
    arrs = []
    for n in ENC_NAMES:
        arrs += [weights[f"{n}/W_conv2d"], weights[f"{n}/b_conv2d"], weights[f"{n}/beta"],
                 weights[f"{n}/moving_mean"], weights[f"{n}/moving_variance"]]
    for l in (6, 5, 4, 3, 2):
        arrs += [weights[f"predict{l}/W_conv2d"], weights[f"predict{l}/b_conv2d"]]
    for d in ("deconv5", "deconv4", "deconv3", "deconv2"):
        arrs += [weights[f"{d}/W_deconv2d"], weights[f"{d}/b_deconv2d"], weights[f"{d}_bn/beta"],
                 weights[f"{d}_bn/moving_mean"], weights[f"{d}_bn/moving_variance"]]
    for u in ("upsample6_5", "upsample5_4", "upsample4_3", "upsample3_2"):
        arrs += [weights[f"{u}/W_deconv2d"], weights[f"{u}/b_deconv2d"]]
    return [_d(a) for a in arrs]


def flownetS_pyramid(feats, weights, level_sizes):
    """level_sizes: [(h,w)] of pf6, pf5, pf4, pf3 (pf2 is (H-2, W-2))."""
    x = _d(feats)
    B, H, W, Cin = x.shape
    arrs = weight_pointer_list(weights)
    ptrs = (C.POINTER(C.c_double) * len(arrs))(*[_p(a) for a in arrs])
    outs = [np.zeros((B, h, w, 2)) for (h, w) in level_sizes] + [np.zeros((B, H - 2, W - 2, 2))]
    lib().vo_flownetS_pyramid(_p(x), B, H, W, Cin, ptrs, *[_p(o) for o in outs])
    return dict(zip(("predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3",
                     "predict_flow2"), outs))


def tf_warp(img, flow):
    im = _d(img)
    fl = np.ascontiguousarray(np.asarray(flow, dtype=np.float32))
    B, H, W, Cc = im.shape
    out = np.zeros_like(im)
    lib().vo_tf_warp(_p(im), fl.ctypes.data_as(C.POINTER(C.c_float)), B, H, W, Cc, _p(out))
    return out


def resize_bilinear(x, oh, ow):
    a = _d(x)
    B, h, w, Cc = a.shape
    out = np.zeros((B, oh, ow, Cc))
    lib().vo_resize_bilinear(_p(a), B, h, w, Cc, oh, ow, _p(out))
    return out


def flow_to_output_res(pf2, net_h, net_w, oh, ow):
    a = _d(pf2)
    B, h, w, _ = a.shape
    out = np.zeros((B, oh, ow, 2))
    lib().vo_flow_to_output_res(_p(a), B, h, w, net_h, net_w, oh, ow, _p(out))
    return out


def nearest_index(i, n_in, n_out):
    return int(lib().vo_nearest_index(int(i), int(n_in), int(n_out)))
