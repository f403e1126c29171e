"""Drop-in for the reference's `NLDF.py` (`Model.build_model`, NLDF.py:24-101): VGG16 trunk + the
non-local deep-feature saliency head at the reference's hard-wired 352x352 geometry.  The reference
never instantiates this model and the file only runs under Python 2 (`k_s / 2` as a pad width,
NLDF.py:132); it is provided because BASELINE.json's north_star names it.  Variable names follow the
TF scopes: `<layer>/W`, `<layer>/b` for Fea_Global_1, Fea_Global_2, Fea_Global, Fea_P1..5,
Fea_P2_Deconv..Fea_P5_Deconv, Local_Fea, Local_Score, Global_Score."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, runtime, vgg16

img_size = 352
label_size = img_size // 2
FEA = 128

HEAD_SHAPES = {
    "Fea_Global_1/W": (5, 5, 512, FEA), "Fea_Global_2/W": (5, 5, FEA, FEA), "Fea_Global/W": (3, 3, FEA, FEA),
    "Fea_P1/W": (3, 3, 64, FEA), "Fea_P2/W": (3, 3, 128, FEA), "Fea_P3/W": (3, 3, 256, FEA),
    "Fea_P4/W": (3, 3, 512, FEA), "Fea_P5/W": (3, 3, 512, FEA),
    "Fea_P5_Deconv/W": (5, 5, FEA, 2 * FEA), "Fea_P4_Deconv/W": (5, 5, 2 * FEA, 3 * FEA),
    "Fea_P3_Deconv/W": (5, 5, 3 * FEA, 4 * FEA), "Fea_P2_Deconv/W": (5, 5, 4 * FEA, 5 * FEA),
    "Local_Fea/W": (1, 1, 6 * FEA, 5 * FEA), "Local_Score/W": (1, 1, 5 * FEA, 2), "Global_Score/W": (1, 1, FEA, 2),
}


def synthetic_head_weights(seed: int = 11, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Variables of the head with the reference's initialisers scaled by `gain` (truncated normal
    stddev 0.01 for convs, NLDF.py:105-108; normal 0.01 for deconvs, :121-123; zero biases would make
    every bias path invisible to tests, so biases are small random numbers)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in HEAD_SHAPES.items():
        w = rng.standard_normal(shp)
        if "Deconv" not in name:
            w = np.clip(w, -2.0, 2.0)
        out[name] = (w * 0.01 * gain).astype(np.float32)
        cout = shp[2] if "Deconv" in name else shp[3]
        out[name[:-1] + "b"] = (rng.standard_normal(cout) * 0.01).astype(np.float32)
    return out


class Model:
    def __init__(self, vgg16_npy_path: Optional[str] = None, vgg_data_dict: Optional[dict] = None,
                 head_weights: Optional[Dict[str, np.ndarray]] = None, seed: Optional[int] = None):
        if vgg_data_dict is None and vgg16_npy_path is None and seed is not None:
            self.vgg = vgg16.Vgg16(seed=seed)
        else:
            self.vgg = vgg16.Vgg16(vgg16_npy_path, data_dict=vgg_data_dict)
        if head_weights is None:
            if seed is None:
                raise ValueError("NLDF.Model needs head_weights (or seed= for synthetic ones): the reference "
                                 "creates them with tf.get_variable and restores a checkpoint that is not available")
            head_weights = synthetic_head_weights(seed)
        self.head_weights = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in head_weights.items()}
        for k, shp in HEAD_SHAPES.items():
            if k not in self.head_weights or self.head_weights[k].shape != shp:
                raise ValueError(f"head weight {k}: expected shape {shp}")
        self._loaded_on = None

    def _load(self, ctx):
        if self._loaded_on is ctx:
            return
        w = self.head_weights
        arr = (_lib.VstabTensor * len(w))()
        keep = []
        for i, (name, a) in enumerate(w.items()):
            nb = name.encode()
            keep.append((nb, a))
            arr[i].name = nb
            arr[i].data = a.ctypes.data_as(_lib.c_float_p)
            arr[i].ndim = a.ndim
            for d in range(a.ndim):
                arr[i].shape[d] = a.shape[d]
        _lib.check(_lib.lib().vstab_nldf_load(ctx._h, arr, len(w)), ctx._h)
        self._loaded_on = ctx

    def build_model(self, input_holder, batch_size, reuse=False, scope='NLDF'):
        """input_holder [B,352,352,3] float32 CUDA in [0,1] -> Prob [B,176,176,1] (saliency probability).
        Also sets Fea_Global, Local_Fea, Local_Score+Global_Score = Score, Prob as attributes."""
        x = input_holder
        if not torch.is_tensor(x) or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
            raise ValueError("input_holder must be a float32 CUDA tensor [B,352,352,3]")
        if tuple(x.shape[1:]) != (img_size, img_size, 3):
            raise ValueError(f"NLDF is hard-wired to {img_size}x{img_size}x3 inputs (NLDF.py:58-64,74)")
        B = x.shape[0]
        if batch_size is not None and int(batch_size) != B:
            raise ValueError(f"batch_size={batch_size} but input has batch {B}")
        vgg = self.vgg
        vgg.build(vgg16.preprocess(x.contiguous()))                      # NLDF.py:29-31
        ctx = vgg._ctx
        self._load(ctx)
        L = _lib.lib()
        dev = x.device
        self.Prob = torch.empty((B, 176, 176, 1), dtype=torch.float32, device=dev)
        self.Score = torch.empty((B, 176, 176, 2), dtype=torch.float32, device=dev)
        self.Local_Fea = torch.empty((B, 176, 176, 5 * FEA), dtype=torch.float32, device=dev)
        self.Fea_Global = torch.empty((B, 1, 1, FEA), dtype=torch.float32, device=dev)
        nws = L.vstab_nldf_workspace_bytes(B)
        if nws == 0:
            raise ValueError(f"NLDF: unsupported batch {B}")
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        pools = (C.c_void_p * 5)(vgg.pool1.data_ptr(), vgg.pool2.data_ptr(), vgg.pool3.data_ptr(), vgg.pool4.data_ptr(),
                                 vgg.pool5.data_ptr())
        with torch.cuda.device(dev):
            _lib.check(L.vstab_nldf_forward(ctx._h, pools, B, self.Prob.data_ptr(), self.Score.data_ptr(),
                                            self.Local_Fea.data_ptr(), self.Fea_Global.data_ptr(), ws.data_ptr(), nws,
                                            runtime.stream_ptr()), ctx._h)
        return self.Prob
