"""The two inference graphs the reference's evaluators build around the network
(evaluate_originalSize main:491-514, evaluate main:802-807), as plain functions."""
from __future__ import annotations

from .model import flownetS_pyramid
from .warp_flow import flow_glue_warp, flow_to_output_res, resize_images, resize_images_slice3, tf_warp


def stabilise_originalsize(feats, frame, scope='flownetS', flow_filter=None):
    """feats [B,Hn,Wn,Cin] network input, frame [B,oh,ow,3] the unstable frame at output
    resolution -> (flows dict, outflow [B,oh,ow,2], warped [B,oh,ow,3])  (main:495-514).  `flow_filter(outflow)` (e.g.
    postfilters.MeanFlow3Filter, the highTV evaluator) replaces the flow that warps; the returned outflow is the filter's input."""
    flows = flownetS_pyramid(feats, feats.shape[0], is_train=False, scope=scope)
    Hn, Wn = feats.shape[1], feats.shape[2]
    oh, ow = frame.shape[1], frame.shape[2]
    if flow_filter is None and frame.shape[3] == 3 and Wn >= 4:          # main:497-514 is one graph: glue + warp in ONE launch
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
