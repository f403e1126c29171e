"""Drop-in for the reference's `model.flownetS_pyramid` (model.py:786-893).

Same name, argument order, NHWC shapes and returned dict keys; eager torch tensors on
the current HIP device instead of TF graph tensors.  All arithmetic runs in
libvstab_hip.so (hand-written HIP for gfx950); torch only owns the buffers.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import runtime, weights as _weights


def flownetS_pyramid(feats, batch_size, is_train=False, reuse=False, scope='flownetS'):
    """feats [B,H,W,C_in] float32 in [0,1] (channels 0-23 = 8 stabilised history frames,
    24-26 = current frame; main:550-558) -> dict with 'predict_flow6' ... 'predict_flow2'
    and 'flow' (= predict_flow2), model.py:893.

    `batch_size` only sized the deconv output_shapes in the reference (model.py:850);
    it must equal feats.shape[0].  `reuse` is accepted and ignored (no variable scopes
    here).  `is_train=False` is the inference path (BatchNorm on the moving statistics,
    folded into the conv weights); `is_train=True` (main:184) runs the training-mode forward
    of the scope's `train_step.Trainer` -- BatchNorm on batch statistics, moving averages
    updated -- whose `loss_and_backward` / `adam` / `step` continue from there."""
    if not torch.is_tensor(feats):
        raise TypeError("feats must be a torch tensor")
    if feats.dim() != 4:
        raise ValueError("feats must be [B,H,W,C]")
    if batch_size is not None and int(batch_size) != feats.shape[0]:
        raise ValueError(f"batch_size={batch_size} but feats has batch {feats.shape[0]}")
    if is_train:
        from . import train_step
        tr = train_step.get_trainer(scope, feats.shape[0], feats.shape[1], feats.shape[2])
        out = {k: v.contiguous() for k, v in tr.forward(feats).items()}
        out['flow'] = out['predict_flow2']
        return out
    ctx = runtime.get_context(scope, feats.device.index if feats.is_cuda else None)
    pf6, pf5, pf4, pf3, pf2 = ctx.forward(feats)
    return {'predict_flow6': pf6, 'predict_flow5': pf5, 'predict_flow4': pf4, 'predict_flow3': pf3,
            'predict_flow2': pf2, 'flow': pf2}


# ---- variable handling: stand-ins for tl.layers.initialize_global_variables (main:517)
# ---- and tl.files.load_and_assign_npz_dict (main:520)
def initialize_global_variables(seed: int = 1, cin: int = 27, random_bn: bool = False, flow_gain: float = 1.0,
                                scope: str = 'flownetS') -> Dict[str, np.ndarray]:
    """Seeded synthetic variables with the graph's initialisers (no checkpoint is
    available offline, README.md:24)."""
    w = _weights.synthetic_weights(seed, cin, random_bn, flow_gain)
    runtime.assign_weights(w, scope)
    return w


def load_and_assign_npz_dict(name: str, sess=None, scope: str = 'flownetS') -> Dict[str, np.ndarray]:
    """Load a tensorlayer `save_npz_dict` checkpoint (keys `main_net/flownetS/<var>:0`)."""
    w = _weights.load_npz_dict(name, scope)
    runtime.assign_weights(w, scope)
    return w


def assign_weights(weights: Dict[str, np.ndarray], scope: str = 'flownetS') -> None:
    runtime.assign_weights(weights, scope)
