"""Building blocks of the training step, one thin wrapper per library entry point: the objective
`lossterm` / `masked_MSE` (main:188-210, "main" = main_flownetS_pyramid_noprevloss_dataloader.py), the total-variation
terms and `loss_main` (main:213-275) with their gradient with respect to every predicted flow; filter / input gradients
of the conv and transposed-conv layers; BatchNorm(lrelu) in training mode and its backward; the resamplers' adjoints.
`train_step.Trainer` strings them into the whole step (forward, loss, backward, Adam: main:184-185, 333-335).

Everything runs in the HIP library (csrc/train_ops.hip); there is no CPU path."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import _lib, runtime
from .warp_flow import resize_images

LOSS_LEVELS = ("predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3", "predict_flow2")
TV_WEIGHTS = (2e-8 * 3, 2e-8 * 3, 2e-8 * 3, 4e-8 * 1.5, 4e-8 * 1.5)          # main:269-273


def _f32(t, name, c):
    if not torch.is_tensor(t) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4 or t.shape[3] != c:
        raise ValueError(f"{name} must be a float32 CUDA tensor [B,h,w,{c}]")
    return t.contiguous()


def lossterm(predict_flow, stab_image, unstab_image, tv_weight: float = 0.0, need_grad: bool = True):
    """main:200-210 (+ the level's TV term when tv_weight != 0).  Returns (loss [0-dim float64 CUDA tensor],
    d loss / d predict_flow [B,h,w,2] or None)."""
    pf = _f32(predict_flow, "predict_flow", 2)
    B, h, w, _ = pf.shape
    stab, unstab = _f32(stab_image, "stab_image", 3), _f32(unstab_image, "unstab_image", 3)
    if stab.shape[0] != B or unstab.shape != stab.shape:
        raise ValueError("stab_image / unstab_image must be [B,H,W,3] with the flow's batch size")
    gt = resize_images(stab, (h, w))                                   # tf.image.resize_images, main:202-203
    un = resize_images(unstab, (h, w))
    sums = torch.empty(3 * B, dtype=torch.float64, device=pf.device)
    grad = torch.empty_like(pf) if need_grad else None
    with torch.cuda.device(pf.device):
        _lib.check(_lib.lib().vstab_loss_level(pf.data_ptr(), gt.data_ptr(), un.data_ptr(), B, h, w, sums.data_ptr(), 1.0,
                                               float(tv_weight), grad.data_ptr() if need_grad else None, runtime.stream_ptr()))
    s = sums.view(B, 3)
    loss = (s[:, 0] / s[:, 1]).mean() + tv_weight * s[:, 2].sum()
    return loss, grad


def loss_main(outputs: Dict[str, torch.Tensor], gtstab_image, unstab_image, need_grad: bool = True
              ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """loss6+...+loss2 + var6+...+var2 (main:275) and its gradient w.r.t. each of the five flows."""
    total, grads = None, {}
    for name, tvw in zip(LOSS_LEVELS, TV_WEIGHTS):
        l, g = lossterm(outputs[name], gtstab_image, unstab_image, tvw, need_grad)
        total = l if total is None else total + l
        if need_grad:
            grads[name] = g
    return total, grads


def loss_main_fused(flows: Dict[str, torch.Tensor], gtstab_image, unstab_image, grads: Dict[str, torch.Tensor] = None):
    """loss_main and (optionally) its flow gradients through ONE library call (`vstab_loss_main`).  flows[name]: [B,h,w,C]
    float32 CUDA pixels whose channels 0..1 are the flow (C even; the Trainer keeps 4-channel pixels); grads[name] (if
    given): [B,h,w,C'] buffers whose channels 0..1 are overwritten.  Returns the loss as a 0-dim float64 CUDA tensor."""
    import ctypes as C
    stab, unstab = _f32(gtstab_image, "stab_image", 3), _f32(unstab_image, "unstab_image", 3)
    B, H, W, _ = stab.shape
    if unstab.shape != stab.shape:
        raise ValueError("stab_image / unstab_image must have the same shape")
    desc = (_lib.LossLevelDesc * len(LOSS_LEVELS))()
    for d, name, tvw in zip(desc, LOSS_LEVELS, TV_WEIGHTS):
        pf = flows[name]
        if not pf.is_cuda or pf.dtype != torch.float32 or pf.dim() != 4 or pf.shape[0] != B or not pf.is_contiguous():
            raise ValueError(f"{name} must be a contiguous float32 CUDA tensor [B,h,w,C]")
        d.pf, d.h, d.w, d.cs_pf, d.tv_weight = pf.data_ptr(), pf.shape[1], pf.shape[2], pf.shape[3], tvw
        if grads is not None:
            g = grads[name]
            if g.shape[:3] != pf.shape[:3] or g.dtype != torch.float32 or not g.is_contiguous() or g.device != pf.device:
                raise ValueError(f"grads[{name}] must be a contiguous float32 tensor [B,h,w,C'] beside the flow")
            d.grad, d.cs_grad = g.data_ptr(), g.shape[3]
    L = _lib.lib()
    n = L.vstab_loss_main_workspace_bytes(C.addressof(desc), len(LOSS_LEVELS), B)
    if n == 0:
        raise ValueError("loss_main: bad level descriptors (channel counts must be even)")
    ws = torch.empty(n, dtype=torch.uint8, device=stab.device)
    out = torch.empty((), dtype=torch.float64, device=stab.device)
    with torch.cuda.device(stab.device):
        _lib.check(L.vstab_loss_main(C.addressof(desc), len(LOSS_LEVELS), stab.data_ptr(), unstab.data_ptr(), B, H, W, out.data_ptr(),
                                     ws.data_ptr(), n, runtime.stream_ptr()))
    return out


# ----------------------------------------------------------------------------- backward building blocks
def conv_wgrad(x, gout, k: int, stride: int, pad: int, cx_off: int = 0, cin: int = None, cg_off: int = 0, cout: int = None,
               dW=None, db=None, accumulate: bool = False, want_db: bool = True):
    """Filter and bias gradient of PadLayer(pad) -> Conv2d(k, stride, VALID) (model.py:807-844): what TF's autodiff
    returns for tf.nn.conv2d's filter, in the reference's HWIO layout.  x [B,Hi,Wi,Cs_x] (channels cx_off..+cin),
    gout [B,Ho,Wo,Cs_g] (channels cg_off..+cout).  Returns (dW [k,k,cin,cout], db [cout] or None)."""
    for t, name in ((x, "x"), (gout, "gout")):
        if not torch.is_tensor(t) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
            raise ValueError(f"{name} must be a float32 CUDA tensor [B,H,W,C]")
    x, gout = x.contiguous(), gout.contiguous()
    B, Hi, Wi, cs_x = x.shape
    _, Ho, Wo, cs_g = gout.shape
    cin = cs_x - cx_off if cin is None else int(cin)
    cout = cs_g - cg_off if cout is None else int(cout)
    if gout.shape[0] != B:
        raise ValueError("x and gout must have the same batch size")
    if dW is None:
        dW = torch.empty((k, k, cin, cout), dtype=torch.float32, device=x.device)
        accumulate = False
    if want_db and db is None:
        db = torch.empty((cout,), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    nbytes = L.vstab_conv_wgrad_workspace_bytes(B, Ho, Wo, k, cin, cout)
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.vstab_conv_wgrad(x.data_ptr(), B, Hi, Wi, cs_x, cx_off, cin, gout.data_ptr(), Ho, Wo, cs_g, cg_off, cout,
                                      k, stride, pad, dW.data_ptr(), db.data_ptr() if want_db else None, 1 if accumulate else 0,
                                      ws.data_ptr(), ws.numel(), runtime.stream_ptr()))
    return dW, (db if want_db else None)


def conv_dgrad(gout, W, stride: int, pad: int, in_hw, cg_off: int = 0, cout: int = None, dx=None, cx_off: int = 0,
               accumulate: bool = False, bias=None):
    """Input gradient of PadLayer(pad) -> Conv2d(k, stride, VALID) with filter W [k,k,cin,cout] (CUDA tensor, HWIO):
    what TF's autodiff returns for tf.nn.conv2d's input.  gout [B,Ho,Wo,Cs_g] (channels cg_off..+cout); the result has
    the input's size in_hw and lands in channels cx_off..+cin of dx (allocated [B,Hi,Wi,cin] when None)."""
    if not torch.is_tensor(gout) or not gout.is_cuda or gout.dtype != torch.float32 or gout.dim() != 4:
        raise ValueError("gout must be a float32 CUDA tensor [B,Ho,Wo,C]")
    if not torch.is_tensor(W) or not W.is_cuda or W.dtype != torch.float32 or W.dim() != 4 or W.shape[0] != W.shape[1]:
        raise ValueError("W must be a float32 CUDA tensor [k,k,cin,cout]")
    gout, W = gout.contiguous(), W.contiguous()
    B, Ho, Wo, cs_g = gout.shape
    k, _, cin, cout_w = W.shape
    cout = cout_w if cout is None else int(cout)
    if cout != cout_w:
        raise ValueError("cout must match the filter")
    Hi, Wi = int(in_hw[0]), int(in_hw[1])
    if dx is None:
        dx = torch.empty((B, Hi, Wi, cin), dtype=torch.float32, device=gout.device)
        accumulate, cx_off = False, 0
    cs_x = dx.shape[3]
    L = _lib.lib()
    nbytes = L.vstab_conv_dgrad_workspace_bytes(B, Ho, Wo, cs_g, cout, k, stride, pad, Hi, Wi, cs_x, cx_off, cin, 1 if accumulate else 0)
    if nbytes == 0:
        raise ValueError("conv_dgrad: unsupported geometry (stride 1 or 2, channel counts multiples of 4, matching sizes)")
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=gout.device)
    with torch.cuda.device(gout.device):
        _lib.check(L.vstab_conv_dgrad(gout.data_ptr(), B, Ho, Wo, cs_g, cg_off, cout, W.data_ptr(),
                                      bias.data_ptr() if bias is not None else None, k, stride, pad, dx.data_ptr(),
                                      Hi, Wi, cs_x, cx_off, cin, 1 if accumulate else 0, ws.data_ptr(), ws.numel(),
                                      runtime.stream_ptr()))
    return dx


def _rows_view(t, c_off, C):
    if not torch.is_tensor(t) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4 or not t.is_contiguous():
        raise ValueError("expected a contiguous float32 CUDA tensor [B,H,W,C]")
    cs = t.shape[3]
    C = cs - c_off if C is None else int(C)
    return t.shape[0] * t.shape[1] * t.shape[2], cs, C


def bn_lrelu_train_forward(z, beta, moving_mean=None, moving_var=None, decay: float = 0.9, eps: float = 1e-5, c_off: int = 0, C: int = None):
    """BatchNormLayer(act=lrelu 0.1, is_train=True, gamma_init=None) IN PLACE on channels c_off..+C of z (model.py:809 etc.):
    z is overwritten by y; moving_mean / moving_var (if given) get TensorLayer's moving-average update.
    Returns (y (= z), save_mean, save_rstd)."""
    rows, cs, C = _rows_view(z, c_off, C)
    mean = torch.empty(C, dtype=torch.float32, device=z.device)
    rstd = torch.empty(C, dtype=torch.float32, device=z.device)
    L = _lib.lib()
    sc = torch.empty(int(L.vstab_bn_scratch_bytes(rows, C)), dtype=torch.uint8, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(L.vstab_bn_lrelu_train_forward(z.data_ptr(), rows, cs, c_off, C, beta.data_ptr(),
                                                  moving_mean.data_ptr() if moving_mean is not None else None,
                                                  moving_var.data_ptr() if moving_var is not None else None, float(decay), float(eps),
                                                  mean.data_ptr(), rstd.data_ptr(), sc.data_ptr(), sc.numel(), runtime.stream_ptr()))
    return z, mean, rstd


def bn_lrelu_train_backward(y, dy, beta, save_rstd, cy_off: int = 0, cg_off: int = 0, C: int = None, dbeta=None, accumulate: bool = False):
    """Backward of the layer above: dy (gradient w.r.t. y) is overwritten IN PLACE by the gradient w.r.t. the layer's input;
    returns (dz (= dy), dbeta)."""
    rows, cs_y, Cy = _rows_view(y, cy_off, C)
    rows_g, cs_g, Cg = _rows_view(dy, cg_off, C)
    if rows != rows_g or Cy != Cg:
        raise ValueError("y and dy must cover the same rows and channel count")
    if dbeta is None:
        dbeta = torch.empty(Cy, dtype=torch.float32, device=y.device)
        accumulate = False
    L = _lib.lib()
    sc = torch.empty(int(L.vstab_bn_scratch_bytes(rows, Cy)), dtype=torch.uint8, device=y.device)
    with torch.cuda.device(y.device):
        _lib.check(L.vstab_bn_lrelu_train_backward(y.data_ptr(), cs_y, cy_off, dy.data_ptr(), cs_g, cg_off, Cy, rows, beta.data_ptr(),
                                                   save_rstd.data_ptr(), dbeta.data_ptr(), 1 if accumulate else 0, sc.data_ptr(),
                                                   sc.numel(), runtime.stream_ptr()))
    return dy, dbeta


def lrelu_backward(y, dy, cy_off: int = 0, cg_off: int = 0, C: int = None):
    """dy *= (y > 0 ? 1 : 0.1) in place."""
    rows, cs_y, Cy = _rows_view(y, cy_off, C)
    _, cs_g, _ = _rows_view(dy, cg_off, C)
    with torch.cuda.device(y.device):
        _lib.check(_lib.lib().vstab_lrelu_backward(y.data_ptr(), cs_y, cy_off, dy.data_ptr(), cs_g, cg_off, Cy, rows, runtime.stream_ptr()))
    return dy


def resize_bilinear_backward(dout, in_hw, gain: float = 1.0, din=None):
    """din (+)= gain * adjoint of tf.image.resize_images(., dout's size) applied to dout [B,oh,ow,C]."""
    dout = dout.contiguous()
    B, oh, ow, C = dout.shape
    acc = din is not None
    if din is None:
        din = torch.empty((B, int(in_hw[0]), int(in_hw[1]), C), dtype=torch.float32, device=dout.device)
    with torch.cuda.device(dout.device):
        _lib.check(_lib.lib().vstab_resize_bilinear_backward(dout.data_ptr(), B, oh, ow, C, din.data_ptr(), din.shape[1], din.shape[2],
                                                             float(gain), 1 if acc else 0, runtime.stream_ptr()))
    return din


def pad_nearest_upsample(src, H: int, W: int):
    """PadLayer(1) -> nearest resize (align_corners=True) to HxW (model.py:795-802, 882-884); C % 4 == 0."""
    src = src.contiguous()
    B, h2, w2, C = src.shape
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.lib().vstab_pad_nearest_upsample(src.data_ptr(), B, h2, w2, C, out.data_ptr(), H, W, runtime.stream_ptr()))
    return out


def pad_nearest_upsample_backward(dout, src_hw, dsrc=None):
    dout = dout.contiguous()
    B, H, W, C = dout.shape
    acc = dsrc is not None
    if dsrc is None:
        dsrc = torch.empty((B, int(src_hw[0]), int(src_hw[1]), C), dtype=torch.float32, device=dout.device)
    with torch.cuda.device(dout.device):
        _lib.check(_lib.lib().vstab_pad_nearest_upsample_backward(dout.data_ptr(), B, H, W, C, dsrc.data_ptr(), dsrc.shape[1], dsrc.shape[2],
                                                                  1 if acc else 0, runtime.stream_ptr()))
    return dsrc


def conv3x3_winograd(x, W, transpose: bool = False, bias=None, y=None, cx_off: int = 0, cy_off: int = 0, act: int = 0):
    """3x3 stride-1 pad-1 convolution (transpose=False: x [..,cin] -> [..,cout]) or its input gradient (transpose=True:
    x = output gradient [..,cout] -> dx [..,cin]) in Winograd F(2x2,3x3) form; W [3,3,cin,cout] CUDA tensor.  act 0 none,
    1 leaky relu 0.1, 3 add to what is in y."""
    if not torch.is_tensor(x) or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("x must be a float32 CUDA tensor [B,H,W,C]")
    x, W = x.contiguous(), W.contiguous()
    B, H, Wd, cs_x = x.shape
    cin, cout = int(W.shape[2]), int(W.shape[3])
    n_out = cin if transpose else cout
    if y is None:
        y = torch.empty((B, H, Wd, n_out), dtype=torch.float32, device=x.device)
        cy_off = 0
    L = _lib.lib()
    nbytes = L.vstab_conv3x3_winograd_workspace_bytes(B, H, Wd, cin, cout, 1 if transpose else 0)
    if nbytes == 0:
        raise ValueError("conv3x3_winograd: unsupported geometry (reduction channels % 32, output channels % 64, < 2 GiB per tensor)")
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.vstab_conv3x3_winograd(x.data_ptr(), B, H, Wd, cs_x, cx_off, W.data_ptr(), cin, cout, 1 if transpose else 0,
                                            bias.data_ptr() if bias is not None else None, y.data_ptr(), y.shape[3], cy_off, act,
                                            ws.data_ptr(), ws.numel(), runtime.stream_ptr()))
    return y


def conv3x3_winograd_wgrad(x, gout, cx_off: int = 0, cin: int = None, cg_off: int = 0, cout: int = None, dW=None):
    """Filter gradient of PadLayer(1) -> Conv2d(3, 1, VALID) (model.py:822, 829, 836, 843) in the Winograd domain
    (`vstab_conv3x3_winograd_wgrad`): same result as `conv_wgrad(x, gout, 3, 1, 1)` up to fp32 rounding at 4/9 of its
    multiply-adds.  x [B,H,W,Cs_x] (channels cx_off..+cin), gout [B,H,W,Cs_g] (channels cg_off..+cout) -> dW [3,3,cin,cout]."""
    x, gout = x.contiguous(), gout.contiguous()
    B, H, W, cs_x = x.shape
    cs_g = gout.shape[3]
    cin = cs_x - cx_off if cin is None else int(cin)
    cout = cs_g - cg_off if cout is None else int(cout)
    if gout.shape[:3] != x.shape[:3]:
        raise ValueError("gout must have the spatial size of x (stride 1, pad 1)")
    L = _lib.lib()
    n = L.vstab_conv3x3_winograd_wgrad_workspace_bytes(B, H, W, cin, cout)
    if n == 0:
        raise ValueError("conv3x3_winograd_wgrad: channel counts must be multiples of 4 (and tensors < 2 GiB)")
    if dW is None:
        dW = torch.empty((3, 3, cin, cout), dtype=torch.float32, device=x.device)
    ws = torch.empty(n, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.vstab_conv3x3_winograd_wgrad(x.data_ptr(), B, H, W, cs_x, cx_off, cin, gout.data_ptr(), cs_g, cg_off, cout,
                                                  dW.data_ptr(), ws.data_ptr(), n, runtime.stream_ptr()))
    return dW
