"""One training step of the reference (`train()`, main:176-335 with "main" = main_flownetS_pyramid_noprevloss_dataloader.py):
forward of `flownetS_pyramid(..., is_train=True)` (BatchNorm on batch statistics, moving averages updated), `loss_main`
(five lossterms + total variation), the backward pass through the whole network and `tf.train.AdamOptimizer(lr, beta1)`.
SURVEY.md 8f rank 4.

Everything numerical runs in the HIP library through the C ABI (`vstab_conv_forward`, `vstab_conv_dgrad`, `vstab_conv_wgrad`,
`vstab_bn_lrelu_train_*`, `vstab_loss_level`, the resampler adjoints, `vstab_adam_step`, ...); this module only owns the
buffers and the order of the calls.  torch is used for allocation, zero fills and strided copies (channel padding), never for
arithmetic.

Layout: NHWC fp32 with every channel count a multiple of four.  The 27-channel input is copied into a 28-channel buffer, the
2-channel flows live in 4-channel pixels, and the concat buffers carry two zero pad channels after the flow (as in the inference
path).  Weights are stored padded the same way (zero rows / columns); a pad row only ever multiplies a zero activation and a pad
column only ever receives a zero gradient, so the padding stays zero under Adam without masking.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib, netspec, runtime
from .distributed import GradBucket
from .training import LOSS_LEVELS, TV_WEIGHTS

ENC = (("1", 7, 2, 3, 64), ("2", 5, 2, 2, 128), ("3", 5, 2, 2, 256), ("3_1", 3, 1, 1, 256), ("4", 3, 2, 1, 512),
       ("4_1", 3, 1, 1, 512), ("5", 3, 2, 1, 512), ("5_1", 3, 1, 1, 512), ("6", 3, 2, 1, 1024), ("6_1", 3, 1, 1, 1024))   # model.py:807-844
# output slice (buffer, channel offset) and input slice (buffer, offset, channels) of every encoder stage
ENC_OUT = {"1": ("conv1", 0), "2": ("concat2", 0), "3": ("conv3", 0), "3_1": ("concat3", 0), "4": ("conv4", 0), "4_1": ("concat4", 0),
           "5": ("conv5", 0), "5_1": ("concat5", 0), "6": ("conv6", 0), "6_1": ("conv6_1", 0)}
ENC_IN = {"1": ("x0", 0, 28), "2": ("conv1", 0, 64), "3": ("concat2", 0, 128), "3_1": ("conv3", 0, 256), "4": ("concat3", 0, 256),
          "4_1": ("conv4", 0, 512), "5": ("concat4", 0, 512), "5_1": ("conv5", 0, 512), "6": ("concat5", 0, 512), "6_1": ("conv6", 0, 1024)}
BUF_C = {"x0": 28, "conv1": 64, "concat2": 196, "conv3": 256, "concat3": 388, "conv4": 512, "concat4": 772, "conv5": 512,
         "concat5": 1028, "conv6": 1024, "conv6_1": 1024}
# decoder levels, coarse to fine: (deconv, its input buffer, real cin, output buffer, offset, cout, predict, upsample, flow offset)
DEC = (("deconv5", "conv6_1", 1024, "concat5", 512, 512, "predict6", "upsample6_5", 1024),
       ("deconv4", "concat5", 1026, "concat4", 512, 256, "predict5", "upsample5_4", 768),
       ("deconv3", "concat4", 770, "concat3", 256, 128, "predict4", "upsample4_3", 384),
       ("deconv2", "concat3", 386, "concat2", 128, 64, "predict3", "upsample3_2", 192))
PRED_IN = {"predict6": ("conv6_1", 1024), "predict5": ("concat5", 1026), "predict4": ("concat4", 770), "predict3": ("concat3", 386),
           "predict2": ("concat2", 194)}
BN_DECAY, BN_EPS = 0.9, 1e-5        # TensorLayer BatchNormLayer defaults
LR_INIT, LR_DECAY, DECAY_EVERY, BETA1 = 1e-4, 0.8, 20, 0.9     # config.py:19-25 (config.TRAIN.*)


def learning_rate(epoch: int, lr_init: float = LR_INIT, lr_decay: float = LR_DECAY, decay_every: int = DECAY_EVERY) -> float:
    """The reference's step schedule (main:389-393): lr_init * lr_decay ** (epoch // decay_every)."""
    return lr_init * lr_decay ** (epoch // decay_every)


def _pad_to(t: torch.Tensor, shape) -> torch.Tensor:
    out = torch.zeros(shape, dtype=torch.float32, device=t.device)
    out[tuple(slice(0, s) for s in t.shape)].copy_(t)
    return out


class Trainer:
    """Holds the (padded, device-resident) parameters, BatchNorm moving statistics and Adam state; `step()` = one
    `sess.run(optim_main)` of the reference."""

    def __init__(self, weights: Dict[str, np.ndarray], batch: int, height: int, width: int, device=None):
        runtime._require_gpu()
        self.dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.B, self.H, self.W = int(batch), int(height), int(width)
        self.sizes = list(netspec.sizes_for(self.H, self.W).enc)              # spatial size after every encoder stage
        self.L = _lib.lib()
        self.p: Dict[str, torch.Tensor] = {}
        self.shape_real: Dict[str, Tuple[int, ...]] = {}
        host = {}
        for name, v in weights.items():
            t = torch.as_tensor(np.asarray(v, dtype=np.float32))
            self.shape_real[name] = tuple(t.shape)
            host[name] = t
        self.trainable = [k for k in host if "moving_" not in k]
        # Trainable parameters, their gradients and the two Adam moments each live in ONE flat buffer with identical offsets
        # (views per tensor): Adam is a single launch over 38.7 M elements, and data-parallel replicas average the gradients with a
        # single all-reduce.  The BatchNorm moving statistics get a bucket of their own so that replicas can average them too.
        # The buffers are laid out in the order the backward pass COMPLETES the gradients (head, decoder fine to coarse, encoder last
        # stage first), so that a finished stretch of the backward pass is one contiguous range that can go on the wire (three
        # overlapped all-reduces, see _grads_ready) while the remaining layers are still being differentiated.
        order = self._backward_order()
        if set(order) != set(self.trainable):
            raise ValueError("weights do not hold the tensors of flownetS_pyramid: " + ", ".join(sorted(set(order) ^ set(self.trainable))))
        self.trainable = order
        shapes = {k: self._padded_shape(k, self.shape_real[k]) for k in self.trainable}
        self.pbucket, self.gbucket = GradBucket(shapes, self.dev), GradBucket(shapes, self.dev)
        self.mbucket, self.vbucket = GradBucket(shapes, self.dev), GradBucket(shapes, self.dev)
        self.g, self.m, self.v = self.gbucket.views, self.mbucket.views, self.vbucket.views
        for k in self.trainable:
            self.p[k] = self.pbucket.views[k]
            self.p[k][tuple(slice(0, s) for s in self.shape_real[k])].copy_(host[k])
        for k in host:
            if "moving_" in k:
                self.p[k] = _pad_to(host[k].to(self.dev), self._padded_shape(k, self.shape_real[k])).contiguous()
        self.sbucket = GradBucket({k: tuple(self.p[k].shape) for k in self.p if "moving_" in k}, self.dev)
        self.t = 0
        self._dp_group, self._dp_handles, self._dp_sent = False, [], 0      # False: no exchange in flight (None is a valid group)
        self.overlap_single_rank = False      # tests: run the overlapped exchange even when the process group has one rank
        self.wino_min_flops = 3.0e9           # a 3x3 stride-1 stage runs in Winograd form once its GEMM issues this much
        self._ws = torch.empty(1 << 20, dtype=torch.uint8, device=self.dev)
        self._alloc_buffers()

    # ------------------------------------------------------------------ parameters
    @staticmethod
    def _r4(n):
        return (n + 3) // 4 * 4

    def _padded_shape(self, name, s):
        if name.endswith("W_conv2d"):                       # [k,k,cin,cout]
            return (s[0], s[1], self._r4(s[2]), self._r4(s[3]))
        if name.endswith("W_deconv2d"):                     # [4,4,cout,cin]
            return (s[0], s[1], self._r4(s[2]), self._r4(s[3]))
        return (self._r4(s[0]),)

    def export(self, tensors: Dict[str, torch.Tensor] = None) -> Dict[str, np.ndarray]:
        """Unpadded copies (host) of the parameters, or of a dict shaped like them (gradients, Adam moments)."""
        src = self.p if tensors is None else tensors
        return {k: t[tuple(slice(0, s) for s in self.shape_real[k])].contiguous().cpu().numpy() for k, t in src.items()}

    # ------------------------------------------------------------------ buffers
    @staticmethod
    def _backward_order():
        """Trainable tensors in the order _backward finishes their gradients."""
        order = ["predict2/b_conv2d", "predict2/W_conv2d"]
        for dname, _ib, _cin, _ob, _ooff, _cout, pname, uname, _foff in reversed(DEC):
            order += [f"{dname}_bn/beta", f"{dname}/W_deconv2d", f"{dname}/b_deconv2d", f"{uname}/W_deconv2d", f"{uname}/b_deconv2d",
                      f"{pname}/W_conv2d", f"{pname}/b_conv2d"]
        for name, _k, _s, _pad, _cout in reversed(ENC):
            order += [f"{name}/beta", f"{name}/W_conv2d", f"{name}/b_conv2d"]
        return order

    # the three gradient buckets of the overlapped exchange: everything up to and including the named tensor (backward order)
    BUCKET_ENDS = ("predict6/b_conv2d", "4/b_conv2d", "1/b_conv2d")         # decoder | conv6_1..conv4 (2/3 of the weights) | conv3_1..conv1

    def _grads_ready(self, last: str):
        """Called by _backward when every gradient up to `last` is final: under data parallelism, start that range's all-reduce."""
        if self._dp_group is False:
            return
        lo = self._dp_sent
        _lo, hi = self.gbucket.span(self.trainable[0], last)
        self._dp_handles.append(self.gbucket.allreduce_range_start(lo, hi, self._dp_group, self.overlap_single_rank))
        self._dp_sent = hi

    def _alloc_buffers(self):
        B, H, W = self.B, self.H, self.W
        hw = {"x0": (H, W), "conv1": self.sizes[0], "concat2": self.sizes[1], "conv3": self.sizes[2], "concat3": self.sizes[3],
              "conv4": self.sizes[4], "concat4": self.sizes[5], "conv5": self.sizes[6], "concat5": self.sizes[7],
              "conv6": self.sizes[8], "conv6_1": self.sizes[9]}
        self.hw = hw
        z = lambda h, w, c: torch.zeros((B, h, w, c), dtype=torch.float32, device=self.dev)
        self.a = {k: z(hw[k][0], hw[k][1], BUF_C[k]) for k in BUF_C}                      # activations
        self.G = {k: z(hw[k][0], hw[k][1], BUF_C[k]) for k in BUF_C if k != "x0"}          # their gradients
        self.flow_hw = {"predict_flow6": hw["conv6_1"], "predict_flow5": hw["concat5"], "predict_flow4": hw["concat4"],
                        "predict_flow3": hw["concat3"], "predict_flow2": (H - 2, W - 2)}
        self.pf = {k: z(s[0], s[1], 4) for k, s in self.flow_hw.items()}                   # flows in 4-channel pixels
        self.dpf = {k: z(s[0], s[1], 4) for k, s in self.flow_hw.items()}
        self.pconv = {k: z(s[0], s[1], 4) for k, s in self.flow_hw.items()}                # predict conv outputs before the residual
        h2, w2 = hw["concat2"]
        self.T = z(h2, w2, 32)                                                              # F7's tap table and its gradient
        self.dT = z(h2, w2, 32)
        self.pf2c = torch.zeros((B, H - 2, W - 2, 2), dtype=torch.float32, device=self.dev)
        # the pyramid heads predict_flow6..3 run through tap tables too (round 5; see _head_forward): per level the table and its
        # gradient [B,h,w,32], the gathered conv output in 2-channel pixels, and the head's filter as the [cs_in, 32] matrix of the
        # table's 1x1 conv, its transpose and its gradient (the zero columns 18..31 / pad rows are written once, here)
        self.Th, self.dTh, self.pc2, self.WTh, self.WTth, self.dWTh = {}, {}, {}, {}, {}, {}
        for pname, (pin, _c) in PRED_IN.items():
            level = "predict_flow" + pname[-1]
            cs = BUF_C[pin]
            if pname != "predict2":
                fh, fw = self.flow_hw[level]
                self.Th[level], self.dTh[level] = z(fh, fw, 32), z(fh, fw, 32)
                self.pc2[level] = torch.zeros((B, fh, fw, 2), dtype=torch.float32, device=self.dev)
            self.WTh[level] = torch.zeros((1, 1, cs, 32), dtype=torch.float32, device=self.dev)
            self.WTth[level] = torch.zeros((1, 1, 32, cs), dtype=torch.float32, device=self.dev)
            self.dWTh[level] = torch.zeros((1, 1, cs, 32), dtype=torch.float32, device=self.dev)
        self.zero_flow = torch.zeros((B, 1, 1, 2), dtype=torch.float32, device=self.dev)
        self.save = {}                                                                       # BatchNorm (mean, rstd) per layer

    def _workspace(self, nbytes):
        if nbytes > self._ws.numel():
            self._ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.dev)
        return self._ws

    # ------------------------------------------------------------------ thin wrappers over the C ABI
    def _check(self, rc):
        _lib.check(rc)

    # ---- per-call timing of the MFMA-bound calls (bench_train.py's roofline): an event pair on the stream the library launches on
    # (torch's current stream, handed over as self.st) around every conv-family C call, with the flops of the layer it computes
    calls = None

    def profile_calls(self, on: bool):
        self.calls = [] if on else None

    def _timed(self, kind, flops, fn):
        if self.calls is None:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.calls.append((kind, float(flops), a, b))
        return r

    def profile_read(self):
        """{family: (ms summed, flops summed, calls)} of the calls recorded since profile_calls(True); synchronises."""
        torch.cuda.synchronize()
        out = {}
        for kind, fl, a, b in self.calls or []:
            ms, f, n = out.get(kind, (0.0, 0.0, 0))
            out[kind] = (ms + a.elapsed_time(b), f + fl, n + 1)
        return out

    def _conv_fwd(self, x, cx_off, cin, W, b, k, s, p, y, cy_off, cout, act=0):
        B, Hi, Wi, cs_x = x.shape
        cs_y = y.shape[3]
        Ho, Wo = y.shape[1], y.shape[2]
        n = self.L.vstab_conv_forward_workspace_bytes(B, Hi, Wi, cs_x, cin, k, s, p, cout, cs_y, cy_off, act, Ho, Wo)
        if n == 0:
            raise ValueError("conv_forward: unsupported geometry")
        ws = self._workspace(n)
        self._timed("conv_forward", 2.0 * B * Ho * Wo * k * k * cin * cout, lambda: self._check(self.L.vstab_conv_forward(
            x.data_ptr(), B, Hi, Wi, cs_x, cx_off, cin, W.data_ptr(), b.data_ptr() if b is not None else None,
            k, s, p, y.data_ptr(), Ho, Wo, cs_y, cy_off, cout, act, ws.data_ptr(), ws.numel(), self.st)))

    def _conv1_rowwin(self, x, W, b, k, s, p, y, cout):
        """model.py:807-808 on conv_rowwin_kernel (vstab_conv_rowwin_forward): x [B,H,W,27] as given, W the padded device filter
        [k,k,28,64] (its 28th input channel is never read); False when the kernel does not take the geometry."""
        B, H, Wd, cin = x.shape
        cs_w = int(W.shape[2])
        n = self.L.vstab_conv_rowwin_forward_workspace_bytes(B, H, Wd, cin, cs_w, cout, k, s, p, y.shape[3], 0, 0)
        if n == 0:
            return False
        ws = self._workspace(n)
        Ho, Wo = y.shape[1], y.shape[2]
        self._timed("conv_forward", 2.0 * B * Ho * Wo * k * k * cin * cout, lambda: self._check(self.L.vstab_conv_rowwin_forward(
            x.data_ptr(), B, H, Wd, cin, W.data_ptr(), cs_w, cout, b.data_ptr(), k, s, p, y.data_ptr(), y.shape[3], 0, 0,
            ws.data_ptr(), ws.numel(), self.st)))
        return True

    def _convT(self, g, cg_off, cout, W, b, k, s, p, dx, cx_off, cin, accumulate):
        """dx (+)= transposed conv of g with W [k,k,cin,cout] (conv input gradient; also DeConv2dLayer's forward)."""
        B, Ho, Wo, cs_g = g.shape
        _, Hi, Wi, cs_x = dx.shape
        n = self.L.vstab_conv_dgrad_workspace_bytes(B, Ho, Wo, cs_g, cout, k, s, p, Hi, Wi, cs_x, cx_off, cin, 1 if accumulate else 0)
        if n == 0:
            raise ValueError("conv_dgrad: unsupported geometry")
        ws = self._workspace(n)
        self._timed("conv_dgrad", 2.0 * B * Ho * Wo * k * k * cin * cout, lambda: self._check(self.L.vstab_conv_dgrad(
            g.data_ptr(), B, Ho, Wo, cs_g, cg_off, cout, W.data_ptr(), b.data_ptr() if b is not None else None,
            k, s, p, dx.data_ptr(), Hi, Wi, cs_x, cx_off, cin, 1 if accumulate else 0, ws.data_ptr(), ws.numel(), self.st)))

    def _wino(self, x, cx_off, W, transpose, bias, y, cy_off, act):
        """3x3 stride-1 stage (or its input gradient) in Winograd F(2x2,3x3) form; False when the geometry does not qualify."""
        B, H, Wd, cs_x = x.shape
        cin, cout = int(W.shape[2]), int(W.shape[3])
        K, N = (cout, cin) if transpose else (cin, cout)
        if 32.0 * B * ((H + 1) // 2) * ((Wd + 1) // 2) * K * N < self.wino_min_flops:   # same break-even as the inference plan
            return False
        n = self.L.vstab_conv3x3_winograd_workspace_bytes(B, H, Wd, cin, cout, 1 if transpose else 0)
        if n == 0:
            return False
        ws = self._workspace(n)
        # flops as the direct 3x3 convolution SURVEY.md 8d prices (the Winograd form issues 4/9 of them)
        self._timed("conv3x3_winograd", 2.0 * B * H * Wd * 9 * cin * cout, lambda: self._check(self.L.vstab_conv3x3_winograd(
            x.data_ptr(), B, H, Wd, cs_x, cx_off, W.data_ptr(), cin, cout, 1 if transpose else 0,
            bias.data_ptr() if bias is not None else None, y.data_ptr(), y.shape[3], cy_off, act, ws.data_ptr(), ws.numel(), self.st)))
        return True

    def _wgrad(self, x, cx_off, cin, g, cg_off, cout, k, s, p, dW, db):
        B, Hi, Wi, cs_x = x.shape
        _, Ho, Wo, cs_g = g.shape
        n = self.L.vstab_conv_wgrad_workspace_bytes(B, Ho, Wo, k, cin, cout)
        ws = self._workspace(n)
        self._timed("conv_wgrad", 2.0 * B * Ho * Wo * k * k * cin * cout, lambda: self._check(self.L.vstab_conv_wgrad(
            x.data_ptr(), B, Hi, Wi, cs_x, cx_off, cin, g.data_ptr(), Ho, Wo, cs_g, cg_off, cout, k, s, p,
            dW.data_ptr(), db.data_ptr() if db is not None else None, 0, ws.data_ptr(), ws.numel(), self.st)))

    def _colsum(self, g, c_off, C, out):
        rows = g.shape[0] * g.shape[1] * g.shape[2]
        ws = self._workspace(self.L.vstab_column_sum_scratch_bytes(rows, C))
        self._check(self.L.vstab_column_sum(g.data_ptr(), rows, g.shape[3], c_off, C, out.data_ptr(), 0, ws.data_ptr(), ws.numel(), self.st))

    def _bn_fwd(self, name, buf, c_off, C):
        rows = buf.shape[0] * buf.shape[1] * buf.shape[2]
        mean = torch.empty(C, dtype=torch.float32, device=self.dev)
        rstd = torch.empty(C, dtype=torch.float32, device=self.dev)
        ws = self._workspace(self.L.vstab_bn_scratch_bytes(rows, C))
        self._check(self.L.vstab_bn_lrelu_train_forward(buf.data_ptr(), rows, buf.shape[3], c_off, C, self.p[f"{name}/beta"].data_ptr(),
                                                        self.p[f"{name}/moving_mean"].data_ptr(), self.p[f"{name}/moving_variance"].data_ptr(),
                                                        BN_DECAY, BN_EPS, mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), ws.numel(), self.st))
        self.save[name] = (mean, rstd)

    def _bn_bwd(self, name, y, G, c_off, C):
        rows = y.shape[0] * y.shape[1] * y.shape[2]
        ws = self._workspace(self.L.vstab_bn_scratch_bytes(rows, C))
        self._check(self.L.vstab_bn_lrelu_train_backward(y.data_ptr(), y.shape[3], c_off, G.data_ptr(), G.shape[3], c_off, C, rows,
                                                         self.p[f"{name}/beta"].data_ptr(), self.save[name][1].data_ptr(),
                                                         self.g[f"{name}/beta"].data_ptr(), 0, ws.data_ptr(), ws.numel(), self.st))

    def _wino_wgrad(self, x, cx_off, cin, G, cg_off, cout, dW):
        """Filter gradient of a 3x3 stride-1 stage in the Winograd domain; False when the geometry does not qualify."""
        B, H, Wd, cs_x = x.shape
        if (cin & 31) or (cout & 63) or 32.0 * B * ((H + 1) // 2) * ((Wd + 1) // 2) * cin * cout < self.wino_min_flops:
            return False
        n = self.L.vstab_conv3x3_winograd_wgrad_workspace_bytes(B, H, Wd, cin, cout)
        if n == 0:
            return False
        ws = self._workspace(n)
        self._timed("conv3x3_winograd_wgrad", 2.0 * B * H * Wd * 9 * cin * cout, lambda: self._check(self.L.vstab_conv3x3_winograd_wgrad(
            x.data_ptr(), B, H, Wd, cs_x, cx_off, cin, G.data_ptr(), G.shape[3], cg_off, cout, dW.data_ptr(), ws.data_ptr(), ws.numel(), self.st)))
        return True

    def _resize(self, x, out):
        B, h, w, C = x.shape
        self._check(self.L.vstab_resize_bilinear(x.data_ptr(), B, h, w, C, out.data_ptr(), out.shape[1], out.shape[2], self.st))

    def _resize_bwd(self, dout, din, gain):
        B, oh, ow, C = dout.shape
        self._check(self.L.vstab_resize_bilinear_backward(dout.data_ptr(), B, oh, ow, C, din.data_ptr(), din.shape[1], din.shape[2],
                                                          float(gain), 1, self.st))

    def _axpby(self, a, x, b, y, out):
        self._check(self.L.vstab_axpby(x.data_ptr(), float(a), y.data_ptr(), float(b), out.data_ptr(), x.numel(), self.st))

    # ------------------------------------------------------------------ forward (is_train=True)
    def forward(self, feats: torch.Tensor) -> Dict[str, torch.Tensor]:
        if tuple(feats.shape) != (self.B, self.H, self.W, 27) or feats.dtype != torch.float32 or not feats.is_cuda:
            raise ValueError(f"feats must be a float32 CUDA tensor {(self.B, self.H, self.W, 27)}")
        self.st = runtime.stream_ptr()
        a, p = self.a, self.p
        a["x0"][..., :27].copy_(feats)              # the 28-channel copy the first layer's filter gradient reads
        for name, k, s, pad, cout in ENC:                                                # model.py:807-844
            ib, ioff, cin = ENC_IN[name]
            ob, ooff = ENC_OUT[name]
            if name == "1" and self._conv1_rowwin(feats.contiguous(), p["1/W_conv2d"], p["1/b_conv2d"], k, s, pad, a[ob], cout):
                pass                                 # the first layer on the inference path's row-window kernel, straight from the 27-channel input
            elif not (k == 3 and s == 1 and self._wino(a[ib], ioff, p[f"{name}/W_conv2d"], False, p[f"{name}/b_conv2d"], a[ob], ooff, 0)):
                self._conv_fwd(a[ib], ioff, cin, p[f"{name}/W_conv2d"], p[f"{name}/b_conv2d"], k, s, pad, a[ob], ooff, cout)
            self._bn_fwd(name, a[ob], ooff, cout)
        prev = None
        for dname, ib, _cin, ob, ooff, cout, pname, uname, foff in DEC:                   # model.py:847-880
            level = "predict_flow" + pname[-1]
            pin, _ = PRED_IN[pname]
            cs_in = a[pin].shape[3]
            self._head_forward(pname, level, a[pin], cs_in)
            if prev is None:
                self.pf[level].copy_(self.pconv[level])
            else:                                                                         # (conv + up) + up, model.py:857
                up = torch.empty_like(self.pf[level])
                self._resize(self.pf[prev], up)
                self._axpby(1.0, self.pconv[level], 2.0, up, self.pf[level])
            # upsample_flowN into the next concat's flow channels, deconvN + BatchNorm into its middle slice
            self._convT(self.pf[level], 0, 4, p[f"{uname}/W_deconv2d"], p[f"{uname}/b_deconv2d"], 4, 2, 1, a[ob], foff, 4, False)
            self._convT(a[ib], 0, a[ib].shape[3], p[f"{dname}/W_deconv2d"], p[f"{dname}/b_deconv2d"], 4, 2, 1, a[ob], ooff, cout, False)
            self._bn_fwd(f"{dname}_bn", a[ob], ooff, cout)
            prev = level
        # full-resolution head (model.py:882-887) through its tap table, as in the inference path: the padded, nearest-upsampled
        # concat2 (1.6 GB at B=8 512x512) never exists.  T[s][tap*2+o] = sum_c concat2[s][c] W[tap][c][o] is a 1x1 conv.
        c2 = a["concat2"]
        self.WT = self._head_matrix("predict2", "predict_flow2", 196)                     # [1,1,196,32] from [3,3,196,4] (columns 2,3 zero)
        self._conv_fwd(c2, 0, 196, self.WT, None, 1, 1, 0, self.T, 0, 32)
        pf3c = self.pf["predict_flow3"][..., :2].contiguous()
        h3, w3 = pf3c.shape[1], pf3c.shape[2]
        self._check(self.L.vstab_pf2_from_taps(self.T.data_ptr(), self.B, c2.shape[1], c2.shape[2], p["predict2/b_conv2d"].data_ptr(),
                                               pf3c.data_ptr(), h3, w3, self.pf2c.data_ptr(), self.H, self.W, self.st))
        self.pf["predict_flow2"][..., :2].copy_(self.pf2c)
        return {k: v[..., :2] for k, v in self.pf.items()}

    def _head_matrix(self, pname, level, cs):
        """The 3x3 -> 2 head's filter [3,3,cs,4] as the matrix of its tap table's 1x1 conv: WT[c][tap*2+o] = W[tap][c][o] (o < 2), into the
        level's persistent [1,1,cs,32] buffer (one strided copy; columns 18..31 stay zero)."""
        WT = self.WTh[level]
        WT[0, 0, :, :18].view(cs, 3, 3, 2).copy_(self.p[f"{pname}/W_conv2d"][..., :2].permute(2, 0, 1, 3))
        return WT

    def _head_forward(self, pname, level, x, cs):
        """predict_flowN's 3x3 pad-1 conv to 2 channels (model.py:848, 856, 865, 874) through its tap table, as the inference path and the
        full-resolution head do: T[s][tap*2+o] = sum_c x[s][c] W[tap][c][o] is a 1x1 conv to 18 (of 32) columns -- the input is read ONCE
        instead of nine times by an im2col GEMM whose N is 4 -- and the conv output gathers its nine in-image taps + bias.  The gather is
        vstab_pf2_from_taps with H = h + 2, W = w + 2: the nearest-neighbour map of the full-resolution head is then the identity, and
        its eight adds of the upsampled coarser flow add a zero field.  Result -> pconv[level] channels 0..1 (2..3 stay zero)."""
        B, h, w, _ = x.shape
        WT = self._head_matrix(pname, level, cs)
        self._conv_fwd(x, 0, cs, WT, None, 1, 1, 0, self.Th[level], 0, 32)
        self._check(self.L.vstab_pf2_from_taps(self.Th[level].data_ptr(), B, h, w, self.p[f"{pname}/b_conv2d"].data_ptr(),
                                               self.zero_flow.data_ptr(), 1, 1, self.pc2[level].data_ptr(), h + 2, w + 2, self.st))
        self.pconv[level][..., :2].copy_(self.pc2[level])

    def _head_backward(self, pname, level, x, cs, Gx, acc):
        """Its backward: dT = the adjoint of the tap gather applied to the flow gradient (vstab_pf2_taps_backward, identity map), the
        filter gradient = x^T dT (a 1x1 filter gradient: K = the pixels), the input gradient = dT W^T (1x1 conv, accumulated into the
        concat gradient when something was written there before), the bias gradient = the column sums of the flow gradient."""
        B, h, w, _ = x.shape
        d, dT, dWT = self.dpf[level], self.dTh[level], self.dWTh[level]
        self._check(self.L.vstab_pf2_taps_backward(d.data_ptr(), 4, B, h + 2, w + 2, dT.data_ptr(), h, w, self.st))
        self._wgrad(x, 0, cs, dT, 0, 32, 1, 1, 0, dWT, None)
        self.g[f"{pname}/W_conv2d"][..., :2].copy_(dWT[0, 0, :, :18].view(cs, 3, 3, 2).permute(1, 2, 0, 3))
        self._colsum(d, 0, 4, self.g[f"{pname}/b_conv2d"])
        WTt = self.WTth[level]
        WTt[0, 0].copy_(self.WTh[level][0, 0].t())
        self._conv_fwd(dT, 0, 32, WTt, None, 1, 1, 0, Gx, 0, cs, act=3 if acc else 0)

    def lrelu_masks(self) -> Dict[str, torch.Tensor]:
        """{BatchNorm layer: y > 0} of the last forward (host bool tensors): which side of the leaky relu every element is on."""
        out = {}
        for name, _k, _s, _p, cout in ENC:
            ob, ooff = ENC_OUT[name]
            out[name] = (self.a[ob][..., ooff:ooff + cout] > 0).cpu()
        for dname, _ib, _cin, ob, ooff, cout, _p, _u, _f in DEC:
            out[f"{dname}_bn"] = (self.a[ob][..., ooff:ooff + cout] > 0).cpu()
        return out

    # ------------------------------------------------------------------ loss + backward
    def loss_and_backward(self, gtstab: torch.Tensor, unstab: torch.Tensor) -> torch.Tensor:
        """loss_main (main:213-217, 269-275) and d loss_main / d every trainable tensor (into self.g)."""
        from .training import loss_main_fused
        self._zero_grads()
        total = loss_main_fused(self.pf, gtstab, unstab, self.dpf)        # all five levels, gradients straight into dpf[..., :2]
        self._backward()
        return total

    def backward_from_flow_grads(self, dflows: Dict[str, torch.Tensor]):
        """Vector-Jacobian product of the network alone: given d L / d predict_flowN [B,h,w,2] for the five flows, fills
        self.g with d L / d every trainable tensor (what tf.gradients(flows, vars, grad_ys=dflows) returns)."""
        self._zero_grads()
        for level in LOSS_LEVELS:
            self.dpf[level][..., :2].copy_(dflows[level])
        self._backward()

    def _zero_grads(self):
        # the activation gradients are NOT cleared: the first gradient written into each buffer in _backward covers all of its
        # channels and overwrites (self._fresh tracks that), later contributions accumulate
        # (nor are the flow gradients dpf: channels 0..1 are overwritten by the loss gradient before anything accumulates into
        # them, channels 2..3 only ever receive zeros)
        self._fresh = set(self.G.keys())

    def _acc(self, name) -> bool:
        """False for the first write into activation gradient `name` of this backward pass (overwrite), True afterwards."""
        first = name in self._fresh
        self._fresh.discard(name)
        return not first

    def _backward(self):
        a, p, g, G = self.a, self.p, self.g, self.G
        # full-resolution head: residual, then the adjoint of the tap gather and the 1x1 tap-table conv
        d2 = self.dpf["predict_flow2"]
        c2 = a["concat2"]
        self._resize_bwd(d2, self.dpf["predict_flow3"], 8.0)
        self._colsum(d2, 0, 4, g["predict2/b_conv2d"])
        self._check(self.L.vstab_pf2_taps_backward(d2.data_ptr(), 4, self.B, self.H, self.W, self.dT.data_ptr(), c2.shape[1], c2.shape[2],
                                                   self.st))
        dWT = self.dWTh["predict_flow2"]
        self._wgrad(c2, 0, 196, self.dT, 0, 32, 1, 1, 0, dWT, None)
        # (the padded output channels of predict2's filter gradient keep the zeros they were created with)
        g["predict2/W_conv2d"][..., :2].copy_(dWT[0, 0, :, :18].view(196, 3, 3, 2).permute(1, 2, 0, 3))
        WTt = self.WTth["predict_flow2"]
        WTt[0, 0].copy_(self.WT[0, 0].t())
        self._conv_fwd(self.dT, 0, 32, WTt, None, 1, 1, 0, G["concat2"], 0, 196, act=3 if self._acc("concat2") else 0)
        # decoder levels, fine to coarse
        levels = ["predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3"]
        for i in range(3, -1, -1):
            dname, ib, _cin, ob, ooff, cout, pname, uname, foff = DEC[i]
            level = levels[i]
            cs_in = a[ib].shape[3]
            # deconvN: BatchNorm backward in place on its slice of the concat gradient, then filter / bias / input gradients
            self._bn_bwd(f"{dname}_bn", a[ob], G[ob], ooff, cout)
            self._wgrad(G[ob], ooff, cout, a[ib], 0, cs_in, 4, 2, 1, g[f"{dname}/W_deconv2d"], None)
            # (b_deconv2d sits in front of BatchNorm: its gradient, the sum of dz, is exactly 0 -- the buffer is created zero and
            #  never written)
            self._conv_fwd(G[ob], ooff, cout, p[f"{dname}/W_deconv2d"], None, 4, 2, 1, G[ib], 0, cs_in, act=3 if self._acc(ib) else 0)
            # upsample_flowN: its input is this level's flow
            self._wgrad(G[ob], foff, 4, self.pf[level], 0, 4, 4, 2, 1, g[f"{uname}/W_deconv2d"], None)
            self._colsum(G[ob], foff, 4, g[f"{uname}/b_deconv2d"])
            self._conv_fwd(G[ob], foff, 4, p[f"{uname}/W_deconv2d"], None, 4, 2, 1, self.dpf[level], 0, 4, act=3)
            # the flow's gradient is complete: residual add, then predict_flowN
            if i > 0:
                self._resize_bwd(self.dpf[level], self.dpf[levels[i - 1]], 2.0)
            pin, _ = PRED_IN[pname]
            self._head_backward(pname, level, a[pin], a[pin].shape[3], G[pin], self._acc(pin))
        self._grads_ready(self.BUCKET_ENDS[0])
        # encoder, last stage first
        for name, k, s, pad, cout in reversed(ENC):
            ib, ioff, cin = ENC_IN[name]
            ob, ooff = ENC_OUT[name]
            self._bn_bwd(name, a[ob], G[ob], ooff, cout)
            if not (k == 3 and s == 1 and self._wino_wgrad(a[ib], ioff, cin, G[ob], ooff, cout, g[f"{name}/W_conv2d"])):
                self._wgrad(a[ib], ioff, cin, G[ob], ooff, cout, k, s, pad, g[f"{name}/W_conv2d"], None)
            # (b_conv2d: same -- the batch mean removes the bias, its gradient stays the zero it was created with)
            if ib != "x0":
                acc = self._acc(ib)
                if not (k == 3 and s == 1 and self._wino(G[ob], ooff, p[f"{name}/W_conv2d"], True, None, G[ib], ioff, 3 if acc else 0)):
                    self._convT(G[ob], ooff, cout, p[f"{name}/W_conv2d"], None, k, s, pad, G[ib], ioff, cin, acc)
            if f"{name}/b_conv2d" in self.BUCKET_ENDS:
                self._grads_ready(f"{name}/b_conv2d")

    # ------------------------------------------------------------------ Adam (main:333-335)
    def adam(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - beta2 ** self.t) / (1.0 - beta1 ** self.t)
        # one launch over the flat buffers (the alignment padding between tensors has zero gradient and stays zero)
        self._check(self.L.vstab_adam_step(self.pbucket.flat.data_ptr(), self.gbucket.flat.data_ptr(), self.mbucket.flat.data_ptr(),
                                           self.vbucket.flat.data_ptr(), self.pbucket.flat.numel(), lr_t, beta1, beta2, eps, self.st))

    def sync_replicas(self, group=None):
        """Data-parallel exchange (no-op on one rank): gradients averaged over the replicas -- inside step() in three ranges of the
        flat bucket whose all-reduces were started during the backward pass and are only waited for here; called on its own,
        with ONE all-reduce of the whole bucket -- and BatchNorm
        moving statistics averaged the same way (each replica normalises with its own batch statistics, as the reference's
        single-GPU graph would on that shard)."""
        import torch.distributed as dist
        if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not self._dp_handles):
            return
        if self._dp_handles:                       # step(): the buckets went out during the backward pass
            GradBucket.allreduce_finish(self._dp_handles)
            if self._dp_sent != self.gbucket.flat.numel():
                raise RuntimeError("overlapped gradient exchange did not cover the whole bucket")
            self._dp_handles, self._dp_sent = [], 0
        else:
            self.gbucket.allreduce_mean(group)
        for k, v in self.sbucket.views.items():
            v.copy_(self.p[k])
        self.sbucket.allreduce_mean(group)
        for k, v in self.sbucket.views.items():
            self.p[k].copy_(v)

    def save_npz(self, path: str, outer: str = "main_net", scope: str = "flownetS"):
        """tl.files.save_npz_dict(save_vars, ...) (main:424-426): every variable under main_net -- weights, biases, betas AND
        the BatchNorm moving statistics -- under the reference's key names, loadable by `load_and_assign_npz_dict`."""
        from . import weights as wts
        wts.save_npz_dict(path, self.export(), outer, scope)

    def train_epoch(self, batches, epoch: int = 0, group=None):
        """The inner loop of main:386-430 over an iterable of (feats, gtstab, unstab) CUDA batches with the epoch's learning
        rate; returns the list of per-step losses (host floats, synchronising once at the end).  Note: the reference evaluates
        `loss_main` with a SECOND training-mode forward after the update (main:409-411), which also advances its BatchNorm
        moving averages twice per iteration; here the loss is the one the update was computed from."""
        lr = learning_rate(epoch)
        losses = [self.step(f, g, u, lr, BETA1, group) for f, g, u in batches]
        return [float(l) for l in losses]

    def step(self, feats, gtstab, unstab, lr: float, beta1: float = 0.9, group=None):
        """One optimiser step; returns this rank's loss_main evaluated before the update (what
        `sess.run([loss_main, optim_main])` prints).  Under torch.distributed the replicas' gradients are averaged first."""
        import torch.distributed as dist
        parallel = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self.overlap_single_rank)
        with torch.cuda.device(self.dev):
            self.forward(feats)
            self._dp_group, self._dp_handles, self._dp_sent = (group if parallel else False), [], 0
            try:
                loss = self.loss_and_backward(gtstab, unstab)      # finished gradient buckets go on the wire as it proceeds
            finally:
                self._dp_group = False
            self.sync_replicas(group)
            self.adam(lr, beta1)
        return loss


# ---------------------------------------------------------------------- the training graph behind model.flownetS_pyramid(is_train=True)
_trainers: Dict[tuple, Trainer] = {}


def get_trainer(scope: str, batch: int, height: int, width: int) -> Trainer:
    """The scope's Trainer for this input shape (main:183-185 builds one training graph per process), created from the weights
    last assigned to the scope (initialize_global_variables / load_and_assign_npz_dict)."""
    key = (scope, int(batch), int(height), int(width))
    tr = _trainers.get(key)
    if tr is None:
        w = runtime._pending_weights.get(scope)
        if w is None:
            raise RuntimeError(f"no weights assigned to scope '{scope}': call initialize_global_variables() or "
                               "load_and_assign_npz_dict() first")
        if int(np.asarray(w["1/W_conv2d"]).shape[2]) != 27:
            raise ValueError("training is built for the 27-channel input stack of the reference's training graph (main:176-184)")
        tr = _trainers[key] = Trainer(w, batch, height, width)
    return tr


def sync_to_inference(trainer: Trainer, scope: str = "flownetS"):
    """Hand the trained variables (weights, betas, moving statistics) to the inference path of `scope` (what
    `flownetS_pyramid(..., is_train=False, reuse=True)` shares through TF variable scopes, main:185)."""
    runtime.assign_weights(trainer.export(), scope)        # re-folds BatchNorm and re-packs in every live context of the scope
