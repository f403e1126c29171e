"""Drop-ins for the warp ops the reference defines inside its main scripts:
`tf_warp` (main:70-130), `get_pixel_value` (main:44-68) and the flow glue of
`evaluate_originalSize` (main:497-498) and `evaluate` (main:806).
("main" = main_flownetS_pyramid_noprevloss_dataloader.py.)  HIP kernels only."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, runtime


def _f32_cuda(t, name):
    if not torch.is_tensor(t) or not t.is_cuda or t.dtype != torch.float32:
        raise ValueError(f"{name} must be a float32 CUDA tensor")
    return t.contiguous()


def tf_warp(img, flow, H, W):
    """img [B,H,W,C], flow [B,H,W,2] (channel 0 = x, 1 = y) -> warped [B,H,W,C]."""
    img = _f32_cuda(img, "img")
    flow = _f32_cuda(flow, "flow")
    B, h, w, Cc = img.shape
    if (h, w) != (H, W) or tuple(flow.shape) != (B, H, W, 2):
        raise ValueError(f"tf_warp: img {tuple(img.shape)} / flow {tuple(flow.shape)} do not match H={H}, W={W}")
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        _lib.check(_lib.lib().vstab_warp_flow(img.data_ptr(), flow.data_ptr(), out.data_ptr(), B, H, W, Cc,
                                              runtime.stream_ptr()))
    return out


def get_pixel_value(img, x, y):
    """img [B,Hi,Wi,C]; x, y int32 [B,H,W] -> img[b, y, x, :] as [B,H,W,C]."""
    img = _f32_cuda(img, "img")
    if x.shape != y.shape or x.dim() != 3:
        raise ValueError("x and y must both be [B,H,W]")
    x = x.to(device=img.device, dtype=torch.int32).contiguous()
    y = y.to(device=img.device, dtype=torch.int32).contiguous()
    B, Hi, Wi, Cc = img.shape
    _, H, W = x.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.lib().vstab_get_pixel_value(img.data_ptr(), x.data_ptr(), y.data_ptr(), out.data_ptr(),
                                                    B, H, W, Cc, Hi, Wi, runtime.stream_ptr()))
    return out


def resize_images(x, size):
    """tf.image.resize_images(x, size): legacy bilinear, align_corners=False (main:806)."""
    x = _f32_cuda(x, "x")
    B, h, w, Cc = x.shape
    oh, ow = int(size[0]), int(size[1])
    out = torch.empty((B, oh, ow, Cc), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().vstab_resize_bilinear(x.data_ptr(), B, h, w, Cc, out.data_ptr(), oh, ow,
                                                    runtime.stream_ptr()))
    return out


def resize_images_slice3(x, c_off, size):
    """resize_images(x[..., c_off:c_off+3], size) read in place from the Cs-channel tensor (main:806: the unstable frame is
    channels 24:27 of the input stack) -- no intermediate 3-channel copy; bit-identical to resize_images of that copy."""
    x = _f32_cuda(x, "x")
    B, h, w, Cs = x.shape
    oh, ow = int(size[0]), int(size[1])
    if (h, w) == (oh, ow):
        return x[..., c_off:c_off + 3].contiguous()
    out = torch.empty((B, oh, ow, 3), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().vstab_resize_bilinear_slice3(x.data_ptr(), B, h, w, Cs, int(c_off), out.data_ptr(), oh, ow,
                                                           runtime.stream_ptr()))
    return out


def flow_to_output_res(predict_flow2, net_h, net_w, out_h, out_w):
    """main:497-498 with 384 -> net_h, 512 -> net_w:
    resize_images(pf2 * net_h / pf2.shape[1], [out_h, out_w]); x = x * out_w / net_w; y = y * out_h / net_h
    (each `*` and `/` its own fp32 operation, in the reference's order)."""
    f = _f32_cuda(predict_flow2, "predict_flow2")
    B, h, w, two = f.shape
    if two != 2:
        raise ValueError("flow must have 2 channels")
    out = torch.empty((B, out_h, out_w, 2), dtype=torch.float32, device=f.device)
    with torch.cuda.device(f.device):
        _lib.check(_lib.lib().vstab_flow_resize_scale(f.data_ptr(), B, h, w, out.data_ptr(), out_h, out_w,
                                                      int(net_h), int(net_w), runtime.stream_ptr()))
    return out


def flow_glue_warp(predict_flow2, frame, net_h, net_w, want_outflow=True):
    """main:497-514 in one launch: (outflow, warped) = (flow_to_output_res(pf2, ...), tf_warp(frame, outflow, oh, ow)) for a
    3-channel frame [B,oh,ow,3]; bit-identical to the two calls.  With want_outflow=False the output-resolution flow is never
    written and None is returned in its place."""
    f = _f32_cuda(predict_flow2, "predict_flow2")
    img = _f32_cuda(frame, "frame")
    B, h, w, two = f.shape
    if two != 2 or img.dim() != 4 or img.shape[0] != B:
        raise ValueError(f"flow_glue_warp: flow {tuple(f.shape)} / frame {tuple(img.shape)} do not match")
    _, oh, ow, Cc = img.shape
    outflow = torch.empty((B, oh, ow, 2), dtype=torch.float32, device=f.device) if want_outflow else None
    out = torch.empty_like(img)
    with torch.cuda.device(f.device):
        _lib.check(_lib.lib().vstab_flow_glue_warp(f.data_ptr(), B, h, w, img.data_ptr(),
                                                   outflow.data_ptr() if want_outflow else None, out.data_ptr(), oh, ow, Cc,
                                                   int(net_h), int(net_w), runtime.stream_ptr()))
    return outflow, out
