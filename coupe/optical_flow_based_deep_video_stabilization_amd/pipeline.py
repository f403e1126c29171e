"""The two inference graphs the reference's evaluators build around the network
(evaluate_originalSize main:491-514, evaluate main:802-807), as plain functions."""
from __future__ import annotations

import torch

from . import _lib, netspec, runtime
from .model import flownetS_pyramid
from .warp_flow import flow_glue_warp, flow_to_output_res, resize_images, resize_images_slice3, tf_warp


def _glue_warp_fusable(frame, Wn) -> bool:
    """vstab_flow_glue_warp's contract: a 3-channel frame whose base is 16-byte aligned (a batch-sliced view of an odd-sized
    frame is not), at least two flow columns.  Everything else takes the two-launch path, which has per-pixel fallbacks."""
    return frame.dim() == 4 and frame.shape[3] == 3 and Wn >= 4 and frame.is_contiguous() and frame.data_ptr() % 16 == 0


def stabilise_originalsize(feats, frame, scope='flownetS', flow_filter=None):
    """feats [B,Hn,Wn,Cin] network input, frame [B,oh,ow,3] the unstable frame at output
    resolution -> (flows dict, outflow [B,oh,ow,2], warped [B,oh,ow,3])  (main:495-514).  `flow_filter(outflow)` (e.g.
    postfilters.MeanFlow3Filter, the highTV evaluator) replaces the flow that warps; the returned outflow is the filter's input."""
    flows = flownetS_pyramid(feats, feats.shape[0], is_train=False, scope=scope)
    Hn, Wn = feats.shape[1], feats.shape[2]
    oh, ow = frame.shape[1], frame.shape[2]
    if flow_filter is None and _glue_warp_fusable(frame, Wn):            # main:497-514 is one graph: glue + warp in ONE launch
        outflow, warped = flow_glue_warp(flows['predict_flow2'], frame, Hn, Wn)
        return flows, outflow, warped
    outflow = flow_to_output_res(flows['predict_flow2'], Hn, Wn, oh, ow)
    return flows, outflow, tf_warp(frame, outflow if flow_filter is None else flow_filter(outflow), oh, ow)


def stabilise_native(feats, scope='flownetS'):
    """main:805-807: warp the current frame (channels 24:27), resized to the flow grid
    (H-2)x(W-2), by predict_flow2."""
    flows = flownetS_pyramid(feats, feats.shape[0], is_train=False, scope=scope)
    H, W = feats.shape[1], feats.shape[2]
    c_off = 24 if feats.shape[3] >= 27 else feats.shape[3] - 3
    unstab = resize_images_slice3(feats, c_off, (H - 2, W - 2))            # main:806, read in place from the 27-channel stack
    return flows, tf_warp(unstab, flows['predict_flow2'], H - 2, W - 2)


class OriginalSizeStabiliser:
    """`stabilise_originalsize` for a loop over equally shaped batches (what the reference's evaluator is: one graph, many
    sess.run calls, main:540-630): every output buffer is allocated ONCE, and a step is ONE call into the library
    (`vstab_stabilise_originalsize`) -- no allocation, no second ctypes call.  The returned tensors are the same objects every
    step and are overwritten by the next one: consume (or copy) them on the same stream before calling again."""

    def __init__(self, B, Hn, Wn, Cin, oh, ow, scope='flownetS', device=None, want_outflow=True):
        self.ctx = runtime.get_context(scope, device)
        if self.ctx.cin is None:
            raise RuntimeError("vstab: weights have not been loaded (initialize_global_variables / load_and_assign_npz_dict)")
        if Cin != self.ctx.cin:
            raise ValueError(f"feats must have {self.ctx.cin} channels")
        self.shape, self.frame_shape = (B, Hn, Wn, Cin), (B, oh, ow, 3)
        dev = torch.device("cuda", self.ctx.device)
        lv = netspec.sizes_for(Hn, Wn).level
        self.flows = [torch.empty((B, lv[k][0], lv[k][1], 2), dtype=torch.float32, device=dev) for k in (6, 5, 4, 3)]
        self.flows.append(torch.empty((B, Hn - 2, Wn - 2, 2), dtype=torch.float32, device=dev))
        self.outflow = torch.empty((B, oh, ow, 2), dtype=torch.float32, device=dev) if want_outflow else None
        self.warped = torch.empty((B, oh, ow, 3), dtype=torch.float32, device=dev)
        self.ws = self.ctx.workspace(B, Hn, Wn, Cin)
        pf6, pf5, pf4, pf3, pf2 = self.flows
        self.result = ({'predict_flow6': pf6, 'predict_flow5': pf5, 'predict_flow4': pf4, 'predict_flow3': pf3, 'predict_flow2': pf2,
                        'flow': pf2}, self.outflow, self.warped)
        self._tail = [f.data_ptr() for f in self.flows] + [self.outflow.data_ptr() if want_outflow else None, self.warped.data_ptr(),
                                                           self.ws.data_ptr(), self.ws.numel()]
        self._fn = _lib.lib().vstab_stabilise_originalsize

    def __call__(self, feats, frame):
        if tuple(feats.shape) != self.shape or tuple(frame.shape) != self.frame_shape:
            raise ValueError(f"expected feats {self.shape} and frame {self.frame_shape}")
        if feats.dtype != torch.float32 or frame.dtype != torch.float32 or not feats.is_cuda or not frame.is_cuda \
                or not feats.is_contiguous() or not frame.is_contiguous():
            raise ValueError("feats and frame must be contiguous float32 CUDA tensors")
        if (feats.data_ptr() | frame.data_ptr()) & 15:
            # a batch-sliced odd-sized frame: the one-call path ends in the fused glue + warp launch, which needs 16-byte rows
            # (stabilise_originalsize falls back to two launches for such views; here the buffers are fixed, so say so up front)
            raise ValueError("feats and frame must be 16-byte aligned (pass a contiguous copy of a batch-sliced view)")
        B, Hn, Wn, Cin = self.shape
        with torch.cuda.device(self.ctx.device):          # the stream named below is this device's current stream
            _lib.check(self._fn(self.ctx._h, feats.data_ptr(), B, Hn, Wn, Cin, frame.data_ptr(), self.frame_shape[1], self.frame_shape[2],
                                *self._tail, runtime.stream_ptr()), self.ctx._h)
        return self.result
