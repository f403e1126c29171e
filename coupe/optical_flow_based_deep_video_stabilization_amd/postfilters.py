"""Flow post-filters the reference's other evaluators wrap around the same hot path:
`evaluate_blurNma` (main_flownetS_pyramid.py:582-700: 75x75 box blur of predict_flow2 blended with a temporal
EMA), `evaluate_medianNma` (:703-821: scipy.signal.medfilt of the flow, same EMA) and the mean-global-flow variant
(main_flownetS_pyramid_highTV_noBBloss.py:629-631, 679-685), and the homography evaluator of
main_flownetS_pyramid_noprevloss_dataloader.py:728-743 (cv2.findHomography(RANSAC) on the dense flow, then
cv2.warpPerspective of the unstable frame) as an on-device dense-flow RANSAC."""
from __future__ import annotations

import torch

from . import _lib, runtime


def _flow(t, name="flow"):
    if not torch.is_tensor(t) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4 or t.shape[3] != 2:
        raise ValueError(f"{name} must be a float32 CUDA tensor [B,h,w,2]")
    return t.contiguous()


def box_blur_flow(flow, k: int = 75):
    """tf.nn.conv2d(channel, constant(1/(k*k), [k,k,1,1]), SAME) on both flow channels (:634-640)."""
    f = _flow(flow)
    B, h, w, _ = f.shape
    tmp, out = torch.empty_like(f), torch.empty_like(f)
    with torch.cuda.device(f.device):
        _lib.check(_lib.lib().vstab_flow_box_blur(f.data_ptr(), B, h, w, int(k), tmp.data_ptr(), out.data_ptr(),
                                                  runtime.stream_ptr()))
    return out


def axpby(a: float, x, b: float, y):
    """a*x + b*y on equally shaped float32 CUDA tensors."""
    x, y = x.contiguous(), y.contiguous()
    if x.shape != y.shape or x.dtype != torch.float32 or not x.is_cuda or not y.is_cuda:
        raise ValueError("x and y must be equally shaped float32 CUDA tensors")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().vstab_axpby(x.data_ptr(), float(a), y.data_ptr(), float(b), out.data_ptr(), x.numel(),
                                          runtime.stream_ptr()))
    return out


def medfilt_flow(flow, kernel_size=5):
    """scipy.signal.medfilt(np.squeeze(of), kernel_size) per sample (main_flownetS_pyramid.py:809).  As in scipy a
    scalar applies to EVERY axis of the [h,w,2] field, the channel axis included -- so the reference's `5` is a
    5x5x5 window of which 75 taps are padded zeros and the result is identically zero; pass (5, 5, 1) for the
    per-channel 5x5 median its author presumably meant."""
    f = _flow(flow)
    B, h, w, _ = f.shape
    ks = (int(kernel_size),) * 3 if isinstance(kernel_size, int) else tuple(int(k) for k in kernel_size)
    if len(ks) != 3:
        raise ValueError("kernel_size must be a scalar or (kh, kw, kc)")
    out = torch.empty_like(f)
    with torch.cuda.device(f.device):
        _lib.check(_lib.lib().vstab_flow_medfilt(f.data_ptr(), B, h, w, ks[0], ks[1], ks[2], out.data_ptr(),
                                                 runtime.stream_ptr()))
    return out


def mean_flow(flow):
    """ones_like(flow_c) * reduce_mean(flow_c, axis=[1,2]) per channel: the global-translation flow (:629)."""
    f = _flow(flow)
    B, h, w, _ = f.shape
    out = torch.empty_like(f)
    with torch.cuda.device(f.device):
        _lib.check(_lib.lib().vstab_flow_mean_fill(f.data_ptr(), B, h, w, out.data_ptr(), runtime.stream_ptr()))
    return out


def find_homography(flow, K: int = 256, seed: int = 0, thresh: float = 3.0, refine: int = 2, stride: int = 1):
    """h, _ = cv2.findHomography(gridmesh, gridmesh - flow, cv2.RANSAC) for every sample of a dense flow [B,H,W,2]
    (main_flownetS_pyramid_noprevloss_dataloader.py:728-735): returns (H [B,3,3] float64 CUDA with h33 = 1, inlier
    counts int32 [B]).  The fit is the library's deterministic dense-flow RANSAC (include/vstab.h): K hash-drawn
    4-point hypotheses, 3 px consensus (cv2's default threshold), `refine` least-squares refits on the inliers."""
    f = _flow(flow)
    B, h, w, _ = f.shape
    L = _lib.lib()
    n = L.vstab_homography_workspace_bytes(B, h, w, int(K))
    if n == 0:
        raise ValueError("find_homography: need 1 <= K <= 512")
    ws = torch.empty(n, dtype=torch.uint8, device=f.device)
    Hm = torch.empty((B, 3, 3), dtype=torch.float64, device=f.device)
    inl = torch.empty((B,), dtype=torch.int32, device=f.device)
    with torch.cuda.device(f.device):
        _lib.check(L.vstab_homography_fit(f.data_ptr(), B, h, w, int(K), int(seed) & 0xffffffff, float(thresh), int(refine),
                                          int(stride), Hm.data_ptr(), inl.data_ptr(), ws.data_ptr(), n, runtime.stream_ptr()))
    return Hm, inl


def warp_perspective_u8(frames, Hm, size_hw=None):
    """cv2.warpPerspective(frame, h, (out_w, out_h)) on uint8 frames [B,H,W,3] with one src->dst matrix per frame
    (main_flownetS_pyramid_noprevloss_dataloader.py:736)."""
    if not torch.is_tensor(frames) or not frames.is_cuda or frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise ValueError("frames must be a uint8 CUDA tensor [B,H,W,3]")
    src = frames.contiguous()
    B, sh, sw, _ = src.shape
    if not torch.is_tensor(Hm) or Hm.dtype != torch.float64 or tuple(Hm.shape) != (B, 3, 3) or Hm.device != src.device:
        raise ValueError("Hm must be a float64 tensor [B,3,3] on the frames' device")
    oh, ow = (sh, sw) if size_hw is None else (int(size_hw[0]), int(size_hw[1]))
    dst = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.lib().vstab_warp_perspective_u8(src.data_ptr(), B, sh, sw, Hm.contiguous().data_ptr(), dst.data_ptr(),
                                                        oh, ow, runtime.stream_ptr()))
    return dst


class BlurEmaFilter:
    """State of evaluate_blurNma's loop: warp flow = 0.9*blur75(of) + 0.1*prevof (:643), then
    prevof = 0.9*prevof + 0.1*of (:695); prevof starts at zero (:670)."""

    def __init__(self, k: int = 75, alpha: float = 0.9):
        self.k, self.alpha, self.prev = k, alpha, None

    def __call__(self, of):
        of = _flow(of)
        if self.prev is None:
            self.prev = torch.zeros_like(of)
        out = axpby(self.alpha, box_blur_flow(of, self.k), 1.0 - self.alpha, self.prev)
        self.prev = axpby(self.alpha, self.prev, 1.0 - self.alpha, of)
        return out


class MedianEmaFilter(BlurEmaFilter):
    """evaluate_medianNma's loop (main_flownetS_pyramid.py:807-815): warp flow = 0.9*medfilt(of) + 0.1*prevof (:761),
    then prevof = 0.9*prevof + 0.1*medfilt(of) (:815) -- unlike the blur variant the EMA runs on the FILTERED flow."""

    def __init__(self, kernel_size=5, alpha: float = 0.9):
        super().__init__(k=kernel_size, alpha=alpha)

    def __call__(self, of):
        of = _flow(of)
        if self.prev is None:
            self.prev = torch.zeros_like(of)
        med = medfilt_flow(of, self.k)
        out = axpby(self.alpha, med, 1.0 - self.alpha, self.prev)
        self.prev = axpby(self.alpha, self.prev, 1.0 - self.alpha, med)
        return out


class MeanFlow3Filter:
    """The highTV evaluator's flow (main_flownetS_pyramid_highTV_noBBloss.py:629-631, 679-685): the output-resolution flow is replaced
    by its per-channel global mean (a pure translation), and the frame is warped by the average of this mean flow and the mean flows
    of the previous (up to) two frames:  outflowMean = (ofsum + outflow) / ofnum,  ofnum = min(i + 1, 3)."""

    def __init__(self, n: int = 3):
        self.n, self.hist = int(n), []

    def __call__(self, outflow):
        m = mean_flow(outflow)
        k = min(len(self.hist) + 1, self.n)
        s = m
        for j in range(k - 1):                              # curflowsum += totaloutputOF[i-j-1]  (:681-682)
            s = axpby(1.0, s, 1.0, self.hist[-1 - j])
        self.hist.append(m)
        if len(self.hist) > self.n:
            self.hist.pop(0)
        return axpby(1.0 / k, s, 0.0, s)
