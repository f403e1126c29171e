"""Drop-in for the reference's `vgg16.py` (Vgg16.build, vgg16.py:25-64): the 13-conv / 5-pool VGG16
trunk, BASELINE.json config 5.  `vgg16.npy` is not shipped with the reference and cannot be fetched
offline, so besides a path the constructor accepts the dict itself ({layer: [W (3,3,Cin,Cout), b]},
the structure `np.load(...).item()` yields, vgg16.py:22) or a seed for synthetic weights.
All arithmetic runs in libvstab_hip.so (implicit-GEMM MFMA conv + ReLU epilogue, max-pool kernel)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, runtime

VGG_MEAN = [103.939, 116.779, 123.68]     # vgg16.py:7 (BGR order; NLDF.py:29 applies it to RGB as is)

LAYERS = (("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128),
          ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv4_1", 256, 512),
          ("conv4_2", 512, 512), ("conv4_3", 512, 512), ("conv5_1", 512, 512), ("conv5_2", 512, 512),
          ("conv5_3", 512, 512))
OUTPUTS = ("conv1_1", "conv1_2", "pool1", "conv2_1", "conv2_2", "pool2", "conv3_1", "conv3_2", "conv3_3", "pool3",
           "conv4_1", "conv4_2", "conv4_3", "pool4", "conv5_1", "conv5_2", "conv5_3", "pool5")


def synthetic_data_dict(seed: int = 7) -> Dict[str, list]:
    """He-normal filters and small biases in the vgg16.npy structure."""
    rng = np.random.default_rng(seed)
    return {name: [(rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32),
                   (rng.standard_normal(co) * 0.05).astype(np.float32)] for name, ci, co in LAYERS}


def preprocess(x: torch.Tensor) -> torch.Tensor:
    """NLDF.py:29: input_holder * 255. - vgg16.VGG_MEAN."""
    if not x.is_cuda or x.dtype != torch.float32 or x.shape[-1] != 3:
        raise ValueError("x must be a float32 CUDA tensor [..., 3]")
    x = x.contiguous()
    out = torch.empty_like(x)
    mean = (C.c_float * 4)(*VGG_MEAN, 0.0)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().vstab_scale_shift(x.data_ptr(), x.numel() // 3, 3, 255.0, mean, out.data_ptr(),
                                                runtime.stream_ptr()))
    return out


class Vgg16:
    def __init__(self, vgg16_npy_path: Optional[str] = None, data_dict: Optional[dict] = None, seed: Optional[int] = None,
                 reuse_outputs: bool = False):
        """reuse_outputs=True keeps the 18 output tensors (2.45 GB per 1080p sample) and the workspace
        between build() calls of the same shape instead of allocating them anew: the tensors a previous
        build() returned are then overwritten by the next one."""
        self.reuse_outputs = reuse_outputs
        self._cache = {}
        if data_dict is None:
            if seed is not None:
                data_dict = synthetic_data_dict(seed)
            else:
                if vgg16_npy_path is None:
                    vgg16_npy_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vgg16.npy")
                data_dict = np.load(vgg16_npy_path, allow_pickle=True, encoding="latin1").item()
        self.data_dict = data_dict
        self._ctx = None

    def _context(self, device):
        if self._ctx is None or self._ctx.device != device:
            ctx = runtime.Context(device)
            arr = (_lib.VstabTensor * 26)()
            keep = []
            i = 0
            for name, ci, co in LAYERS:
                if name not in self.data_dict:
                    raise KeyError(f"vgg16 weights lack layer {name!r}")
                W = np.ascontiguousarray(np.asarray(self.data_dict[name][0]), dtype=np.float32)
                b = np.ascontiguousarray(np.asarray(self.data_dict[name][1]), dtype=np.float32)
                if W.shape != (3, 3, ci, co) or b.shape != (co,):
                    raise ValueError(f"{name}: filter {W.shape} / biases {b.shape}, expected {(3, 3, ci, co)} / {(co,)}")
                for suffix, a in (("filter", W), ("biases", b)):
                    nb = f"{name}/{suffix}".encode()
                    keep.append((nb, a))
                    arr[i].name = nb
                    arr[i].data = a.ctypes.data_as(_lib.c_float_p)
                    arr[i].ndim = a.ndim
                    for d in range(a.ndim):
                        arr[i].shape[d] = a.shape[d]
                    i += 1
            _lib.check(_lib.lib().vstab_vgg16_load(ctx._h, arr, 26), ctx._h)
            self._ctx = ctx
        return self._ctx

    def build(self, input, train=False):
        """input [B,H,W,3] float32 CUDA -> sets self.conv1_1 ... self.pool5 (NHWC tensors)."""
        if not torch.is_tensor(input) or not input.is_cuda or input.dtype != torch.float32 or input.dim() != 4 \
                or input.shape[3] != 3:
            raise ValueError("input must be a float32 CUDA tensor [B,H,W,3]")
        x = input.contiguous()
        B, H, W, _ = x.shape
        ctx = self._context(x.device.index)
        L = _lib.lib()
        hwc = (C.c_int32 * 54)()
        _lib.check(L.vstab_vgg16_shapes(H, W, hwc))
        key = (B, H, W, x.device.index)
        cached = self._cache.get(key) if self.reuse_outputs else None
        if cached is None:
            outs = [torch.empty((B, hwc[3 * i], hwc[3 * i + 1], hwc[3 * i + 2]), dtype=torch.float32, device=x.device)
                    for i in range(18)]
            nws = L.vstab_vgg16_workspace_bytes(B, H, W)
            if nws == 0:
                raise ValueError(f"vgg16: unsupported problem {(B, H, W)}")
            ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
            if self.reuse_outputs:
                self._cache = {key: (outs, ws, nws)}
        else:
            outs, ws, nws = cached
        ptrs = (C.c_void_p * 18)(*[o.data_ptr() for o in outs])
        with torch.cuda.device(x.device):
            _lib.check(L.vstab_vgg16_forward(ctx._h, x.data_ptr(), B, H, W, ptrs, ws.data_ptr(), nws, runtime.stream_ptr()),
                       ctx._h)
        for name, t in zip(OUTPUTS, outs):
            setattr(self, name, t)
        return self
