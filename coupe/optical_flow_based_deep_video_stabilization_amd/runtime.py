"""Per-device contexts, registered weights and workspaces (torch is used only to own
device memory and to name the current HIP stream)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, netspec, weights as _weights


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("vstab: no HIP device visible; this path has no CPU fallback")


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def trace_ranges(on: bool):
    """roctx ranges per layer (rocprofv3 --marker-trace); raises if no roctx library is installed."""
    _lib.check(_lib.lib().vstab_trace_ranges(int(bool(on))))


HBM_SLOTS = ("warp_flow", "flow_resize_scale", "flow_glue_warp", "pf2_gather", "st_sampler", "homography_warp", "pf2_glue_warp")


def hbm_profile(mode: int):
    """Per-launch dispatch-timestamp events around the HBM-side kernels (tf_warp, flow glue, fused launch):
    0 = off (records kept), 1 = clear + on, 2 = on again."""
    _lib.check(_lib.lib().vstab_hbm_profile_enable(int(mode)))


def hbm_profile_read():
    """{slot name: (kernel ms summed, launches, algorithmic bytes summed)}; synchronise the stream first."""
    out = {}
    for i, name in enumerate(HBM_SLOTS):
        ms, n, by = C.c_double(), C.c_int(), C.c_double()
        _lib.check(_lib.lib().vstab_hbm_profile_read(i, C.byref(ms), C.byref(n), C.byref(by)))
        out[name] = (ms.value, n.value, by.value)
    return out


class Context:
    """One vstab_ctx per (device, scope): packed weights + cached workspaces."""

    def __init__(self, device: int):
        _require_gpu()
        self.device = device
        self._h = C.c_void_p()
        _lib.check(_lib.lib().vstab_create(C.byref(self._h), device))
        self.cin: Optional[int] = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self.plan_batch = 0
        self.plan_flags = 0

    def set_plan_batch(self, batch: int):
        """Pin every plan decision that changes the order of a sample's sums (split-K factors, Winograd or direct form,
        weight-stream or tiled kernel) to what a batch of `batch` samples gets: any call with B <= batch then gives each sample
        bit-identical results whatever it is batched with (ragged tails of a sharded clip).  0 unpins."""
        _lib.check(_lib.lib().vstab_set_plan_batch(self._h, int(batch)), self._h)
        self.plan_batch = int(batch)
        self._ws.clear()

    def set_plan_flags(self, flags: int):
        """VSTAB_PLAN_* bits (diagnostic; 1 = few-row layers stay on the tiled kernel + split-K combine launch, 2 = refinement levels as four
        launches, 4 = the one-call stabiliser's last two launches kept apart: A/B runs and tests)."""
        _lib.check(_lib.lib().vstab_set_plan_flags(self._h, int(flags)), self._h)
        self.plan_flags = int(flags)
        self._ws.clear()

    def close(self):
        if self._h:
            _lib.lib().vstab_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, weights: Dict[str, np.ndarray]):
        w = _weights.validate(weights)
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
        _lib.check(_lib.lib().vstab_load_weights(self._h, arr, len(w)), self._h)
        self.cin = int(w["1/W_conv2d"].shape[2])

    def workspace(self, B, H, W, Cin) -> torch.Tensor:
        if 0 < B <= self.plan_batch:
            B = self.plan_batch            # one workspace for every batch of a pinned context (it suffices for all of them)
        key = (B, H, W, Cin)
        ws = self._ws.get(key)
        if ws is None:
            n = _lib.lib().vstab_workspace_bytes_ctx(self._h, B, H, W, Cin)
            if n == 0:
                raise ValueError(f"vstab: unsupported problem size {key}: "
                                 f"{_lib.lib().vstab_last_error(None).decode()}")
            ws = torch.empty(n, dtype=torch.uint8, device=f"cuda:{self.device}")
            self._ws.clear()           # keep one workspace per context
            self._ws[key] = ws
        return ws

    def forward(self, feats: torch.Tensor):
        if self.cin is None:
            raise RuntimeError("vstab: weights have not been loaded (initialize_global_variables / "
                               "load_and_assign_npz_dict)")
        if feats.dim() != 4 or feats.dtype != torch.float32 or not feats.is_cuda:
            raise ValueError("feats must be a float32 CUDA tensor [B,H,W,C]")
        feats = feats.contiguous()
        B, H, W, Cin = feats.shape
        sz = netspec.sizes_for(H, W)
        ws = self.workspace(B, H, W, Cin)
        lv = sz.level
        dev = feats.device
        flows = [torch.empty((B, lv[k][0], lv[k][1], 2), dtype=torch.float32, device=dev) for k in (6, 5, 4, 3)]
        flows.append(torch.empty((B, H - 2, W - 2, 2), dtype=torch.float32, device=dev))
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().vstab_flownets_forward(
                self._h, feats.data_ptr(), B, H, W, Cin, *[f.data_ptr() for f in flows],
                ws.data_ptr(), ws.numel(), stream_ptr()), self._h)
        return flows

    LAUNCH_SLOTS = ("conv1", "conv2", "conv3", "conv3_1", "conv4", "conv4_1", "conv5", "conv5_1", "conv6", "conv6_1",
                    "deconv5", "deconv4", "deconv3", "deconv2", "predict2_taps")

    def profile(self, enable: bool):
        _lib.check(_lib.lib().vstab_profile_enable(self._h, int(enable)), self._h)
        _lib.check(_lib.lib().vstab_profile_reset(self._h), self._h)

    def profile_set(self, enable: bool):
        """Switch event recording on/off without clearing what has been recorded (sampling some steps)."""
        _lib.check(_lib.lib().vstab_profile_enable(self._h, int(enable)), self._h)

    def profile_read(self):
        """(ms, algorithmic flops) per launch slot, both summed over the recorded passes, and the
        number of passes.  The stream must have been synchronised."""
        ms = (C.c_double * 15)()
        fl = (C.c_double * 15)()
        n = C.c_int()
        _lib.check(_lib.lib().vstab_profile_read(self._h, ms, fl, C.byref(n)), self._h)
        return list(ms), list(fl), n.value

    def profile_read_direct(self):
        """The slots' flops counted as direct convolutions (profile_read gives what the launches issue: Winograd-form stages
        issue 4/9 of that)."""
        fl = (C.c_double * 15)()
        _lib.check(_lib.lib().vstab_profile_read_direct(self._h, fl), self._h)
        return list(fl)

    def profile_kernel_names(self):
        out = []
        for slot in range(15):
            b = C.create_string_buffer(96)
            _lib.check(_lib.lib().vstab_profile_kernel_name(self._h, slot, b, 96), self._h)
            out.append(b.value.decode())
        return out

    def internals(self, B, H, W, Cin):
        """Views of the intermediate tensors in the workspace of the last forward (tests)."""
        ws = self.workspace(B, H, W, Cin)
        ent = (_lib.VstabWsEntry * 24)()
        n = _lib.lib().vstab_workspace_layout_ctx(self._h, B, H, W, Cin, ent, 24)      # the plan THIS context runs (pinned batch, flags)
        if n < 0:
            _lib.check(n)
        out = {}
        for e in ent[:n]:
            name = e.name.decode()
            if name in ("splitk", "tickets") or e.h == 0:
                continue
            nfl = e.n * e.h * e.w * e.c_stride
            flat = ws[e.offset_bytes:e.offset_bytes + 4 * nfl].view(torch.float32)
            out[name] = flat.view(e.n, e.h, e.w, e.c_stride)[..., :e.c]
        return out


_contexts: Dict[tuple, Context] = {}
_pending_weights: Dict[str, Dict[str, np.ndarray]] = {}


def get_context(scope: str = "flownetS", device: Optional[int] = None) -> Context:
    _require_gpu()
    if device is None:
        device = torch.cuda.current_device()
    key = (scope, device)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = Context(device)
        if scope in _pending_weights:
            ctx.load_weights(_pending_weights[scope])
        _contexts[key] = ctx
    return ctx


def assign_weights(weights: Dict[str, np.ndarray], scope: str = "flownetS"):
    """Register the variables of `scope` (short names or full checkpoint keys)."""
    w = _weights.validate(weights)
    _pending_weights[scope] = w
    for (sc, _dev), ctx in _contexts.items():
        if sc == scope:
            ctx.load_weights(w)


def reset():
    for ctx in _contexts.values():
        ctx.close()
    _contexts.clear()
    _pending_weights.clear()
    import sys
    ts = sys.modules.get(__package__ + ".train_step")
    if ts is not None:
        ts._trainers.clear()
