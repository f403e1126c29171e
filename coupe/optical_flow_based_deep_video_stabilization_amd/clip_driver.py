"""Autoregressive clip driver: the per-frame loop of the reference's `evaluate_originalSize`
(main:535-630, "main" = main_flownetS_pyramid_noprevloss_dataloader.py) with everything kept on the
device.  Frame i's network input stacks the previously OUTPUT (stabilised) frames at lags
31,23,15,7,4,3,2,1 (main:553), quantised through uint8 and resized to the network resolution with
cv2.resize, plus the current unstable frame; the stabilised frame is the current frame warped by the
predicted flow brought to output resolution.  Frames of one clip are therefore sequential; throughput
comes from running several clips in lockstep (the batch dimension = clips).

Frames are uint8 [n_clips, out_h, out_w, 3] CUDA tensors in cv2's BGR order, as `cap.read()` yields
them (the reference's own COLOR_BGR2RGB/RGB2BGR juggling is reproduced, main:530,550,568,625)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, runtime
from .model import flownetS_pyramid
from .pipeline import stabilise_originalsize

STAB_LAGS = (31, 23, 15, 7, 4, 3, 2, 1)          # stabidxs, main:553
RING = 32


def _overlaps(a: torch.Tensor, b: torch.Tensor) -> bool:
    """do the storages' byte ranges of two dense tensors intersect?"""
    a0, b0 = a.data_ptr(), b.data_ptr()
    return a0 < b0 + b.numel() * b.element_size() and b0 < a0 + a.numel() * a.element_size()


def _u8(t, name):
    if not torch.is_tensor(t) or not t.is_cuda or t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3:
        raise ValueError(f"{name} must be a uint8 CUDA tensor [n,H,W,3]")
    return t.contiguous()


def resize_u8(src: torch.Tensor, size_hw, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cv2.resize(src, (w, h)) with INTER_LINEAR on uint8 frames [n,H,W,3]; `out` (contiguous, same device) receives the result in place."""
    src = _u8(src, "src")
    n, sh, sw, _ = src.shape
    dh, dw = int(size_hw[0]), int(size_hw[1])
    if out is not None and (out.dtype != torch.uint8 or tuple(out.shape) != (n, dh, dw, 3) or not out.is_contiguous() or out.device != src.device):
        raise ValueError("out must be a contiguous uint8 tensor [n, h, w, 3] on src's device")
    dst = out if out is not None else torch.empty((n, dh, dw, 3), dtype=torch.uint8, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.lib().vstab_resize_u8(src.data_ptr(), n, sh, sw, dst.data_ptr(), dh, dw, runtime.stream_ptr()))
    return dst


class ClipStabiliser:
    def __init__(self, out_h: int, out_w: int, n_clips: int = 1, net_hw=(384, 512), scope: str = 'flownetS',
                 device: Optional[int] = None, homography: bool = False, ransac: Optional[dict] = None, flow_filter=None,
                 keep_outflow: bool = False):
        """homography=True is the evaluator of main:728-743: the frame WRITTEN is the unstable frame under one
        homography fitted to the dense flow (cv2.findHomography + cv2.warpPerspective), while the history later frames
        read stays the flow-warped frame (main:739).  `ransac` = keyword arguments of postfilters.find_homography.
        `flow_filter` (e.g. postfilters.MeanFlow3Filter(): the highTV evaluator, ...highTV...:629-631) maps the output-resolution flow
        to the flow that warps.  Without a `flow_filter` a frame is ONE library call (`vstab_clip_step`) into buffers allocated here,
        once: the network input assembled straight from the full-resolution frame (cv2.resize inside the launch), the network, ONE
        launch for swap(frame)/255 -> flow glue -> tf_warp -> uint8(swap(warped*255)) on the 8-bit frames, the history frame resized
        into its ring slot.  `last_flows` (and `last_outflow`) are then the SAME tensors every frame, overwritten by the next step.
        `keep_outflow` also writes the output-resolution flow (`last_outflow`; always on with `homography`)."""
        runtime._require_gpu()
        self.out_h, self.out_w, self.n = int(out_h), int(out_w), int(n_clips)
        self.net_h, self.net_w = int(net_hw[0]), int(net_hw[1])
        self.scope = scope
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.device = dev
        self.ring = torch.zeros((RING, self.n, self.net_h, self.net_w, 3), dtype=torch.uint8, device=dev)
        self.feats = torch.empty((self.n, self.net_h, self.net_w, 27), dtype=torch.float32, device=dev)
        self.frame_f = torch.empty((self.n, self.out_h, self.out_w, 3), dtype=torch.float32, device=dev)
        self.i = 0
        self.last_flows = None
        self.homography, self.ransac, self.last_homography = bool(homography), dict(ransac or {}), None
        self.last_outflow = None
        self.flow_filter = flow_filter
        self.keep_outflow = bool(keep_outflow) or self.homography
        self._one_call = None

    def _prepare_one_call(self):
        """Everything the one-call step needs, once: flow buffers, the workspace, the ring slots' addresses, the argument tail."""
        from . import netspec
        ctx = runtime.get_context(self.scope, self.device.index)
        if ctx.cin is None:
            raise RuntimeError("vstab: weights have not been loaded (initialize_global_variables / load_and_assign_npz_dict)")
        if ctx.cin != 27:
            raise ValueError("the clip driver's network input is 27 channels (8 history frames + the current one)")
        lv = netspec.sizes_for(self.net_h, self.net_w).level
        fl = [torch.empty((self.n, lv[k][0], lv[k][1], 2), dtype=torch.float32, device=self.device) for k in (6, 5, 4, 3)]
        fl.append(torch.empty((self.n, self.net_h - 2, self.net_w - 2, 2), dtype=torch.float32, device=self.device))
        outflow = torch.empty((self.n, self.out_h, self.out_w, 2), dtype=torch.float32, device=self.device) if self.keep_outflow else None
        ws = ctx.workspace(self.n, self.net_h, self.net_w, 27)
        base, slot_bytes = self.ring.data_ptr(), self.ring[0].numel()
        flows = {'predict_flow6': fl[0], 'predict_flow5': fl[1], 'predict_flow4': fl[2], 'predict_flow3': fl[3], 'predict_flow2': fl[4], 'flow': fl[4]}
        self._one_call = dict(ctx=ctx, flows=flows, outflow=outflow, ws=ws, ring_ptrs=[base + k * slot_bytes for k in range(RING)],
                              ptrs=(C.c_void_p * 8)(), fn=_lib.lib().vstab_clip_step, plan=(ctx.plan_batch, ctx.plan_flags),
                              mid=[self.n, self.net_h, self.net_w, self.out_h, self.out_w, self.feats.data_ptr()] + [f.data_ptr() for f in fl] +
                                  [outflow.data_ptr() if outflow is not None else None])

    def reset(self):
        self.i = 0
        self.last_flows = None

    def _step_u8(self, f, i, out):
        """main:550-558, 568-569, 497-514, 625/630, 556 without a flow filter, ONE library call: (flows, outflow or None, out u8).  The
        stabilised frame's history copy lands in ring slot i % RING inside the same call."""
        oc = self._one_call
        if oc is None or oc["ctx"]._h.value is None or oc["plan"] != (oc["ctx"].plan_batch, oc["ctx"].plan_flags) or \
                runtime._contexts.get((self.scope, self.device.index)) is not oc["ctx"]:
            self._prepare_one_call()
            oc = self._one_call
        ptrs, rp = oc["ptrs"], oc["ring_ptrs"]
        for j, lag in enumerate(STAB_LAGS):      # the eight history slots (null = the resized current frame: a clip's first frame)
            ptrs[j] = None if i == 0 else rp[max(i - lag, 0) % RING]
        if out is None:
            out = torch.empty_like(f)
        ws = oc["ws"]
        with torch.cuda.device(self.device):
            _lib.check(oc["fn"](oc["ctx"]._h, ptrs, f.data_ptr(), *oc["mid"], out.data_ptr(), rp[i % RING], ws.data_ptr(), ws.numel(),
                                runtime.stream_ptr()), oc["ctx"]._h)
        return oc["flows"], oc["outflow"], out

    def step(self, frame_bgr_u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One frame of every clip: uint8 [n,out_h,out_w,3] BGR -> stabilised uint8 [n,out_h,out_w,3] BGR.  `out` (same shape,
        contiguous, same device; without a flow filter) receives the result instead of a freshly allocated tensor."""
        f = _u8(frame_bgr_u8, "frame")
        if tuple(f.shape) != (self.n, self.out_h, self.out_w, 3):
            raise ValueError(f"frame must be {(self.n, self.out_h, self.out_w, 3)}, got {tuple(f.shape)}")
        if out is not None and (self.flow_filter is not None or out.dtype != torch.uint8 or tuple(out.shape) != tuple(f.shape)
                                or not out.is_contiguous() or out.device != f.device):
            raise ValueError("out must be a contiguous uint8 tensor of the frame's shape on its device (and no flow_filter)")
        if out is not None and (_overlaps(out, f) or _overlaps(out, self.ring)):
            # the warp gathers frame pixels while other workgroups already write `out`, and the history slot is resized from `out`
            raise ValueError("out must not overlap the input frame or the history ring")
        L = _lib.lib()
        i = self.i
        if self.flow_filter is None:
            flows, outflow, out = self._step_u8(f, i, out)
        else:
            cur_small = resize_u8(f, (self.net_h, self.net_w))                       # main:550
            slots = []
            for lag in STAB_LAGS:                                                    # main:553-558
                if i == 0:
                    slots.append(cur_small)      # totaloutputFrame[0] is still the raw first frame (main:548-549)
                else:
                    slots.append(self.ring[max(i - lag, 0) % RING])
            slots.append(cur_small)
            ptrs = (C.c_void_p * 9)(*[s.data_ptr() for s in slots])
            with torch.cuda.device(self.device):
                _lib.check(L.vstab_assemble_input(ptrs, self.n, self.net_h, self.net_w, self.feats.data_ptr(), runtime.stream_ptr()))
                _lib.check(L.vstab_frame_to_float(f.data_ptr(), self.n * self.out_h * self.out_w, self.frame_f.data_ptr(),
                                                  runtime.stream_ptr()))                                   # main:568
            flows, outflow, warped = stabilise_originalsize(self.feats, self.frame_f, scope=self.scope, flow_filter=self.flow_filter)    # main:569
            out = torch.empty_like(f)
            with torch.cuda.device(self.device):
                _lib.check(L.vstab_quantise_output(warped.data_ptr(), self.n * self.out_h * self.out_w, out.data_ptr(),
                                                   runtime.stream_ptr()))                                  # main:625,630
            resize_u8(out, (self.net_h, self.net_w), out=self.ring[i % RING])    # what later frames read back (main:556), written in place
        self.last_flows, self.last_outflow = flows, outflow
        self.i += 1
        if self.homography:
            from . import postfilters
            self.last_homography, _ = postfilters.find_homography(outflow, **self.ransac)               # main:728-735
            return postfilters.warp_perspective_u8(f, self.last_homography)                            # main:736,743
        return out

    def run(self, clip_bgr_u8: torch.Tensor) -> torch.Tensor:
        """clip [T, n, out_h, out_w, 3] (or [T, out_h, out_w, 3] for one clip) -> stabilised clip, same shape."""
        single = clip_bgr_u8.dim() == 4
        clip = clip_bgr_u8.unsqueeze(1) if single else clip_bgr_u8
        self.reset()
        out = torch.stack([self.step(clip[t]) for t in range(clip.shape[0])])
        return out[:, 0] if single else out


class NativeClipStabiliser:
    """The native-resolution evaluator loop (`evaluate`, main:758-866; with a `flow_filter` also `evaluate_blurNma` /
    `evaluate_medianNma` of main_flownetS_pyramid.py:582-821): every frame is brought to the network resolution (cv2.resize),
    stacked behind the previously OUTPUT frames at lags 1,2,3,4,7,15,23,31 (main:844 -- the other order than evaluate_originalSize),
    and the current frame, resized to the 382x510 flow grid, is warped by predict_flow2 (or by `flow_filter(predict_flow2)`:
    `postfilters.BlurEmaFilter()`, `MedianEmaFilter()`).  The warped frame, resized back to 384x512 and quantised, is both the result
    and the history (main:861-863).  Frames: uint8 [n_clips, H, W, 3] BGR CUDA tensors; results uint8 [n_clips, 384, 512, 3]."""
    LAGS = (1, 2, 3, 4, 7, 15, 23, 31)            # stabidxs, main:844

    def __init__(self, n_clips: int = 1, net_hw=(384, 512), scope: str = 'flownetS', device: Optional[int] = None, flow_filter=None):
        runtime._require_gpu()
        self.n, self.net_h, self.net_w = int(n_clips), int(net_hw[0]), int(net_hw[1])
        self.scope, self.flow_filter = scope, flow_filter
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.device = dev
        self.ring = torch.zeros((RING, self.n, self.net_h, self.net_w, 3), dtype=torch.uint8, device=dev)
        self.feats = torch.empty((self.n, self.net_h, self.net_w, 27), dtype=torch.float32, device=dev)
        self.i = 0
        self.last_flows = None

    def reset(self):
        self.i = 0
        self.last_flows = None

    def step(self, frame_bgr_u8: torch.Tensor) -> torch.Tensor:
        from .model import flownetS_pyramid
        from .warp_flow import resize_images_slice3, tf_warp
        f = _u8(frame_bgr_u8, "frame")
        if f.shape[0] != self.n:
            raise ValueError(f"frame must hold {self.n} clips, got {f.shape[0]}")
        L = _lib.lib()
        i = self.i
        cur_small = resize_u8(f, (self.net_h, self.net_w))                       # main:842-843
        slots = [cur_small if i == 0 else self.ring[max(i - lag, 0) % RING] for lag in self.LAGS]      # main:845-849
        slots.append(cur_small)
        ptrs = (C.c_void_p * 9)(*[s.data_ptr() for s in slots])
        with torch.cuda.device(self.device):
            _lib.check(L.vstab_assemble_input(ptrs, self.n, self.net_h, self.net_w, self.feats.data_ptr(), runtime.stream_ptr()))
        flows = flownetS_pyramid(self.feats, self.n, is_train=False, scope=self.scope)                  # main:804
        of = flows['predict_flow2']
        flow = self.flow_filter(of) if self.flow_filter is not None else of
        fh, fw = self.net_h - 2, self.net_w - 2
        unstab = resize_images_slice3(self.feats, 24, (fh, fw))                                          # main:806
        warped = tf_warp(unstab, flow, fh, fw)                                                          # main:807
        out = torch.empty((self.n, self.net_h, self.net_w, 3), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(L.vstab_resize_f32_to_u8(warped.data_ptr(), self.n, fh, fw, out.data_ptr(), self.net_h, self.net_w,
                                                runtime.stream_ptr()))                                   # main:861, 863
        self.ring[i % RING].copy_(out)
        self.last_flows = flows
        self.i += 1
        return out

    def run(self, clip_bgr_u8: torch.Tensor) -> torch.Tensor:
        """clip [T, n, H, W, 3] (or [T, H, W, 3] for one clip) -> stabilised clip [T, n, 384, 512, 3] (or [T, 384, 512, 3])."""
        single = clip_bgr_u8.dim() == 4
        clip = clip_bgr_u8.unsqueeze(1) if single else clip_bgr_u8
        self.reset()
        out = torch.stack([self.step(clip[t]) for t in range(clip.shape[0])])
        return out[:, 0] if single else out
