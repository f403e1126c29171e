"""Drop-ins for the reference's `warp.py` (named by BASELINE.json's north_star; dead code in the
reference: only `import warp` at config.py:3, and `config.py` never defines the `warpType`,
`refMtrx`, `warpApprox`, `batch_size`, `height`, `width` fields these functions read).  `config` is
any object carrying those attributes; images and parameters are float32 CUDA tensors.  The
arithmetic runs in HIP kernels (csrc/sampler_ops.hip); `fit` is host numpy like the original."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, runtime


def fit(Xsrc, Xdst):
    """Least-squares affine map Xsrc -> Xdst ([N,2] point sets) as a 3x3 float32 matrix (warp.py:6-14)."""
    Xsrc = np.asarray(Xsrc, dtype=np.float64)
    Xdst = np.asarray(Xdst, dtype=np.float64)
    n = len(Xsrc)
    A = np.zeros((2 * n, 6))
    A[:n, 0:2], A[:n, 2] = Xsrc, 1.0
    A[n:, 3:5], A[n:, 5] = Xsrc, 1.0
    b = np.concatenate([Xdst[:, 0], Xdst[:, 1]])
    sol = np.linalg.lstsq(A, b, rcond=None)[0]
    return np.array([[sol[0], sol[1], sol[2]], [sol[3], sol[4], sol[5]], [0, 0, 1]], dtype=np.float32)


def compose(config, p, dp):
    return p + dp


def inverse(config, p):
    return -p


def _f32_cuda(t, name):
    if not torch.is_tensor(t):
        t = torch.as_tensor(np.asarray(t, dtype=np.float32))
    if not t.is_cuda:
        runtime._require_gpu()
        t = t.cuda()
    return t.to(torch.float32).contiguous()


def vec2mtrx(config, p):
    """p [B,8] (homography: sl(3) generator, warp.py:28-30) or [B,6] (affine, :31-34) -> [B,3,3]
    Taylor matrix exponential with config.warpApprox terms (:37-42)."""
    p = _f32_cuda(p, "p")
    B = p.shape[0]
    if config.warpType == "homography":
        dim = 8
    elif config.warpType == "affine":
        dim = 6
    else:
        raise AssertionError("warpType must be 'homography' or 'affine'")
    if p.shape[1] != dim:
        raise ValueError(f"p must be [B,{dim}] for warpType={config.warpType}")
    out = torch.empty((B, 3, 3), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(_lib.lib().vstab_vec2mtrx(p.data_ptr(), B, dim, int(config.warpApprox), out.data_ptr(),
                                             runtime.stream_ptr()))
    return out


def warpImage(image, M, oh, ow):
    """The warp of transformImage given the composed matrices M = refMtrx . pMtrx [B,3,3] (not a reference symbol: the reference
    composes inside transformImage, as `transformImage` below does inside its launch)."""
    image = _f32_cuda(image, "image")
    B, Hi, Wi, Cc = image.shape
    M = _f32_cuda(M, "matrix").reshape(B, 9)
    out = torch.empty((B, oh, ow, Cc), dtype=torch.float32, device=image.device)
    with torch.cuda.device(image.device):
        _lib.check(_lib.lib().vstab_homography_warp(image.data_ptr(), B, Hi, Wi, Cc, M.data_ptr(), out.data_ptr(),
                                                    oh, ow, runtime.stream_ptr()))
    return out


def _warp_ref(image, ref, pMtrx, oh, ow):
    """refMtrx . pMtrx (warp.py:48-49, 91-92: a tf.matmul in the reference) is composed inside the warp launch -- every product and
    sum rounded to fp32 -- not by a library GEMM in front of it."""
    image = _f32_cuda(image, "image")
    B, Hi, Wi, Cc = image.shape
    pM = _f32_cuda(pMtrx, "pMtrx").to(image.device).reshape(B, 9)
    ref = _f32_cuda(ref, "refMtrx").to(image.device).reshape(9)
    out = torch.empty((B, oh, ow, Cc), dtype=torch.float32, device=image.device)
    with torch.cuda.device(image.device):
        _lib.check(_lib.lib().vstab_transform_image(image.data_ptr(), B, Hi, Wi, Cc, ref.data_ptr(), pM.data_ptr(), out.data_ptr(),
                                                    oh, ow, runtime.stream_ptr()))
    return out


def transformImage(config, image, pMtrx):
    """image [B,H,W,3] warped by refMtrx . pMtrx on the canonical [-1,1]^2 grid (warp.py:46-86)."""
    return _warp_ref(image, config.refMtrx, pMtrx, int(config.height), int(config.width))


def transformCropImage(config, image, pMtrx):
    """As transformImage with refMtrx_b, source [B,dataH,dataW,3], output height x W (warp.py:89-129)."""
    return _warp_ref(image, config.refMtrx_b, pMtrx, int(config.height), int(config.W))
