"""Drop-ins for the samplers of the reference's `spatial_transformer.py` that BASELINE.json's
north_star names: `transformer` (:34-38), `AffineTransformer` (:373-452), `ProjectiveTransformer`
(:519-608), `_meshgrid` (:755-779), `_repeat` (:782-785), `_interpolate` (:787-792) and
`bilinear_interp` (:902-964).  None of them is executed by the reference's runnable scripts (they
are only imported, main:4); the arithmetic runs in HIP kernels (csrc/sampler_ops.hip)."""
from __future__ import annotations

import torch

from . import _lib, runtime


def _f32_cuda(t, name):
    if not torch.is_tensor(t) or not t.is_cuda or t.dtype != torch.float32:
        raise ValueError(f"{name} must be a float32 CUDA tensor")
    return t.contiguous()


def _meshgrid(out_size, device=None):
    """Flat [3*H*W] sampling grid: linspace(-1,1,W) as x (fastest), linspace(-1,1,H) as y, ones."""
    runtime._require_gpu()
    oh, ow = int(out_size[0]), int(out_size[1])
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    out = torch.empty(3 * oh * ow, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vstab_st_meshgrid(out.data_ptr(), oh, ow, runtime.stream_ptr()))
    return out


def _repeat(x, n_repeats):
    """tile(expand_dims(x, 1), [1, n]) flattened: every element repeated n times in place."""
    return x.reshape(-1, 1).repeat(1, int(n_repeats)).reshape(-1)


def bilinear_interp(im, x, y, out_size):
    """im [B,H,W,C]; x, y flat [B*out_h*out_w] normalised to [-1,1] -> [B*out_h*out_w, C].
    The image is zero-padded by one pixel; coordinates are clipped to [-1, W] / [-1, H]."""
    im = _f32_cuda(im, "im")
    B, H, W, Cc = im.shape
    x = _f32_cuda(x.to(torch.float32), "x").reshape(-1)
    y = _f32_cuda(y.to(torch.float32), "y").reshape(-1)
    oh, ow = int(out_size[0]), int(out_size[1])
    npix = oh * ow
    if x.numel() != B * npix or y.numel() != B * npix:
        raise ValueError(f"x/y must have B*out_h*out_w = {B * npix} elements")
    out = torch.empty((B * npix, Cc), dtype=torch.float32, device=im.device)
    with torch.cuda.device(im.device):
        _lib.check(_lib.lib().vstab_st_bilinear_interp(im.data_ptr(), B, H, W, Cc, x.data_ptr(), y.data_ptr(), oh, ow,
                                                       out.data_ptr(), runtime.stream_ptr()))
    return out


def _interpolate(im, x, y, out_size, method):
    if method == 'bilinear':
        return bilinear_interp(im, x, y, out_size)
    if method == 'bicubic':
        raise NotImplementedError("bicubic_interp is not on the path this build covers")
    return None            # the reference falls through to None for unknown methods (:792)


class _ThetaTransformer(object):
    param_dim = 0

    def __init__(self, out_size, name, interp_method='bilinear', **kwargs):
        self.name = name
        self.out_size = (int(out_size[0]), int(out_size[1]))
        self.interp_method = interp_method
        self._grid = None

    @property
    def pixel_grid(self):
        if self._grid is None:
            self._grid = _meshgrid(self.out_size)
        return self._grid

    def transform(self, inp, theta):
        if self.interp_method != 'bilinear':
            raise NotImplementedError("only interp_method='bilinear' is implemented")
        inp = _f32_cuda(inp, "inp")
        B, H, W, Cc = inp.shape
        theta = _f32_cuda(theta.to(torch.float32), "theta").reshape(-1)
        if theta.numel() != B * self.param_dim:
            raise ValueError(f"theta must have shape [{B}, {self.param_dim}]")
        oh, ow = self.out_size
        out = torch.empty((B, oh, ow, Cc), dtype=torch.float32, device=inp.device)
        with torch.cuda.device(inp.device):
            _lib.check(_lib.lib().vstab_st_transform(inp.data_ptr(), B, H, W, Cc, theta.data_ptr(), self.param_dim,
                                                     out.data_ptr(), oh, ow, runtime.stream_ptr()))
        return out


class AffineTransformer(_ThetaTransformer):
    """theta [B,6] = row-major 2x3 matrix acting on (x_t, y_t, 1), x_t, y_t in [-1,1]."""
    param_dim = 6

    def __init__(self, out_size, name='SpatialAffineTransformer', interp_method='bilinear', **kwargs):
        super().__init__(out_size, name, interp_method, **kwargs)


class ProjectiveTransformer(_ThetaTransformer):
    """theta [B,8] = first 8 entries of a 3x3 homography (last entry 1); divides by z,
    z == 0 replaced by 1e-8 (:598)."""
    param_dim = 8

    def __init__(self, out_size, name='SpatialProjectiveTransformer', interp_method='bilinear', **kwargs):
        super().__init__(out_size, name, interp_method, **kwargs)


def transformer(inp, theta, out_size, name='SpatialTransformer', **kwargs):
    """Legacy wrapper (:34-38).  The reference passes `out_size` to `transform`, which does not take
    it (a TypeError there); here the call does what the wrapper evidently means."""
    return AffineTransformer(out_size).transform(inp, theta)
