"""CPU oracle for the FlowNetS-pyramid flow + bilinear flow-warp hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `coupe/` may import this file; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and
only as the checker / the timed CPU baseline -- never as the product path.

PARITY UNPINNED at the TensorFlow-1.10 / TensorLayer boundary: the reference ships
no tests, golden vectors, weights or sample data, and TF 1.10 + tensorlayer cannot
be imported in this image (plain ModuleNotFoundError), so this file
is a restatement of the graph written from the reference's Python plus the
TF-r1.10 kernel semantics recorded in SURVEY.md Appendix A.  What pins it instead:
hand-derivable known-answer tests (tests/test_oracle_kat.py), a second independent
restatement in plain C (oracle/vstab_oracle.c) that must agree with this one, and
the fixtures under tests/golden/ that this file generated.

Every function names the reference lines it follows (paths relative to
/root/reference; "main" = main_flownetS_pyramid_noprevloss_dataloader.py).
All tensors are NHWC like the reference's; `dtype` is the arithmetic type
(torch.float64 = arbiter, torch.float32 = "what TF CPU would compute").
Sample coordinates are always formed in fp32, as the TF kernels do, so the
fp64 arbiter samples exactly the same source pixels.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
FLOW_KEYS = ("predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3", "predict_flow2")


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a)).to(dtype) if not torch.is_tensor(a) else a.to(dtype)


# --------------------------------------------------------------------------- convs
def pad_conv(x, W_hwio, b, pad: int, stride: int):
    """PadLayer(pad, "constant") -> Conv2d(k, stride, VALID) + bias (model.py:807-808 etc.).
    Cross-correlation y[o] = b + sum_k x[s*o + k - p] W[k]."""
    xn = x.permute(0, 3, 1, 2)
    w = W_hwio.permute(3, 2, 0, 1)
    y = F.conv2d(xn, w, b, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1).contiguous()


def bn_lrelu(x, beta, mean, var):
    """BatchNormLayer(act=lrelu(0.1), is_train=False, gamma_init=None) (model.py:809):
    y = lrelu((x - mu) * rsqrt(var + 1e-5) + beta), no gamma."""
    y = (x - mean) * torch.rsqrt(var + BN_EPS) + beta
    return torch.maximum(y, 0.1 * y)      # tl.act.lrelu(x, 0.1) = max(x, 0.1 x), model.py:788


def bn_lrelu_train(x, beta, mask=None):
    """BatchNormLayer(act=lrelu(0.1), is_train=True, gamma_init=None): batch mean / population variance over (N,H,W)
    (tf.nn.moments), no gamma.  Returns (y, mean, var) -- the moving averages are the caller's business.
    `mask` (bool, optional) fixes which side of the leaky relu every element is on: the gradient is discontinuous at the
    kink, so a parity check of a BACKWARD pass hands the checked implementation's own activation pattern in (a pre-activation
    within rounding distance of zero would otherwise flip a whole 0.9*dy between an fp32 and an fp64 forward)."""
    mean = x.mean(dim=(0, 1, 2))
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    y = (x - mean) * torch.rsqrt(var + BN_EPS) + beta
    if mask is not None:
        return y * torch.where(mask, 1.0, 0.1).to(y.dtype), mean, var
    return torch.maximum(y, 0.1 * y), mean, var


def deconv4x4s2(x, W_hwoi, b, out_hw: Tuple[int, int]):
    """DeConv2dLayer(shape=(4,4,Cout,Cin), output_shape=(B,h,w,Cout), strides 2, SAME) + bias
    (model.py:850): y[oy,ox,co] = b + sum x[iy,ix,ci] W[ky,kx,co,ci] over oy = 2 iy + ky - 1
    (SURVEY.md A.2)."""
    xn = x.permute(0, 3, 1, 2)
    w = W_hwoi.permute(3, 2, 0, 1)            # [Cin, Cout, kh, kw], no flip
    y = F.conv_transpose2d(xn, w, b, stride=2, padding=1)
    oh, ow = out_hw
    assert (oh + 1) // 2 == x.shape[1] and (ow + 1) // 2 == x.shape[2]
    return y[:, :, :oh, :ow].permute(0, 2, 3, 1).contiguous()


# --------------------------------------------------------------------------- resizes
def _legacy_axis(n_in: int, n_out: int):
    """TF-1.10 ResizeBilinear (align_corners=False, no half-pixel centres) source
    indices and weight for one axis, computed in fp32."""
    scale = np.float32(n_in) / np.float32(n_out)
    f = np.arange(n_out, dtype=np.float32) * scale
    lo = np.floor(f)
    t = (f - lo).astype(np.float32)
    lo = lo.astype(np.int64)
    hi = np.minimum(lo + 1, n_in - 1)
    return lo, hi, t


def resize_bilinear_legacy(x, oh: int, ow: int):
    """tf.image.resize_images(x, [oh, ow]) / UpSampling2dLayer(size, is_scale=False)
    with the defaults method=BILINEAR, align_corners=False (model.py:857,866,875,886;
    main:497,806).  Returns the input unchanged when the size already matches."""
    B, h, w, C = x.shape
    if (h, w) == (oh, ow):
        return x
    ylo, yhi, ty = _legacy_axis(h, oh)
    xlo, xhi, tx = _legacy_axis(w, ow)
    ty = torch.from_numpy(ty).to(x.dtype).view(1, oh, 1, 1)
    tx = torch.from_numpy(tx).to(x.dtype).view(1, 1, ow, 1)
    top_rows, bot_rows = x[:, ylo], x[:, yhi]
    tl_, tr_ = top_rows[:, :, xlo], top_rows[:, :, xhi]
    bl_, br_ = bot_rows[:, :, xlo], bot_rows[:, :, xhi]
    top = tl_ + (tr_ - tl_) * tx
    bot = bl_ + (br_ - bl_) * tx
    return top + (bot - top) * ty


def nearest_align_corners_index(n_in: int, n_out: int) -> np.ndarray:
    """TF-1.10 ResizeNearestNeighbor(align_corners=True) source index per output index:
    min(roundf(i * (in-1)/(out-1)), in-1), product in fp32, round half away from zero
    (used at model.py:883 through the helper model.py:795-802)."""
    scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
    f = np.arange(n_out, dtype=np.float32) * scale
    r = np.floor(f + np.float32(0.5))          # f >= 0: roundf == floor(f + 0.5)
    return np.minimum(r.astype(np.int64), n_in - 1)


def predict2_fullres(concat2, W_hwio, b, H: int, W: int):
    """model.py:882-885: PadLayer(concat2, 1) -> nearest(align_corners=True) resize to the
    input's HxW -> Conv2d 3x3 VALID -> 2 channels.  Materialises the upsampled tensor
    (fine at oracle sizes)."""
    p = F.pad(concat2, (0, 0, 1, 1, 1, 1))
    iy = nearest_align_corners_index(p.shape[1], H)
    ix = nearest_align_corners_index(p.shape[2], W)
    up = p[:, iy][:, :, ix]
    return pad_conv(up, W_hwio, b, pad=0, stride=1)


# --------------------------------------------------------------------------- network
def level_sizes(H: int, W: int):
    """Spatial size of every encoder stage; generalises the 384x512 literals of
    model.py:850-886."""
    spec = ((7, 2, 3), (5, 2, 2), (5, 2, 2), (3, 1, 1), (3, 2, 1), (3, 1, 1), (3, 2, 1), (3, 1, 1),
            (3, 2, 1), (3, 1, 1))
    out, h, w = [], H, W
    for k, s, p in spec:
        h, w = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        out.append((h, w))
    return out


def flownetS_pyramid(feats, weights: Dict[str, np.ndarray], dtype=torch.float64,
                     return_internals: bool = False, is_train: bool = False, batch_stats: dict = None,
                     lrelu_masks: dict = None):
    """The graph of model.py:786-893; `is_train` selects BatchNorm's batch statistics (the training graph, main:184) --
    then `batch_stats` (if a dict) receives {layer: (mean, var)} for the moving-average update and `lrelu_masks`
    ({layer: bool tensor}) pins the leaky-relu side of every element (see bn_lrelu_train).  `weights` uses short
    names ('1/W_conv2d', 'deconv5_bn/beta', ... SURVEY.md A.7)."""
    x = _t(feats, dtype)
    Wt = {k: _t(v, dtype) for k, v in weights.items()}
    B, H, W, _ = x.shape
    internals = {}

    def enc(name, inp, k, s, p):
        y = pad_conv(inp, Wt[f"{name}/W_conv2d"], Wt[f"{name}/b_conv2d"], p, s)
        if is_train:
            y, m, v = bn_lrelu_train(y, Wt[f"{name}/beta"], None if lrelu_masks is None else lrelu_masks[name])
            if batch_stats is not None:
                batch_stats[name] = (m.detach(), v.detach())
        else:
            y = bn_lrelu(y, Wt[f"{name}/beta"], Wt[f"{name}/moving_mean"], Wt[f"{name}/moving_variance"])
        internals[f"conv{name}"] = y
        return y

    conv1 = enc("1", x, 7, 2, 3)                       # model.py:807-809
    conv2 = enc("2", conv1, 5, 2, 2)                   # :810-812
    conv3 = enc("3", conv2, 5, 2, 2)                   # :813-815
    conv3_1 = enc("3_1", conv3, 3, 1, 1)               # :818-820
    conv4 = enc("4", conv3_1, 3, 2, 1)                 # :822-824
    conv4_1 = enc("4_1", conv4, 3, 1, 1)               # :826-828
    conv5 = enc("5", conv4_1, 3, 2, 1)                 # :830-832
    conv5_1 = enc("5_1", conv5, 3, 1, 1)               # :834-836
    conv6 = enc("6", conv5_1, 3, 2, 1)                 # :838-840
    conv6_1 = enc("6_1", conv6, 3, 1, 1)               # :842-844

    def predict(name, inp):                            # :847-848 etc. (bias, no BN, no act)
        return pad_conv(inp, Wt[f"{name}/W_conv2d"], Wt[f"{name}/b_conv2d"], 1, 1)

    def dec(dname, inp, out_hw):                       # :850-851 etc.
        y = deconv4x4s2(inp, Wt[f"{dname}/W_deconv2d"], Wt[f"{dname}/b_deconv2d"], out_hw)
        if is_train:
            y, m, v = bn_lrelu_train(y, Wt[f"{dname}_bn/beta"], None if lrelu_masks is None else lrelu_masks[f"{dname}_bn"])
            if batch_stats is not None:
                batch_stats[f"{dname}_bn"] = (m.detach(), v.detach())
            return y
        return bn_lrelu(y, Wt[f"{dname}_bn/beta"], Wt[f"{dname}_bn/moving_mean"],
                        Wt[f"{dname}_bn/moving_variance"])

    def upflow(uname, flow, out_hw):                   # :852 etc.
        return deconv4x4s2(flow, Wt[f"{uname}/W_deconv2d"], Wt[f"{uname}/b_deconv2d"], out_hw)

    pf6 = predict("predict6", conv6_1)
    s5 = tuple(conv5_1.shape[1:3])
    concat5 = torch.cat([conv5_1, dec("deconv5", conv6_1, s5), upflow("upsample6_5", pf6, s5)], 3)
    up = resize_bilinear_legacy(pf6, *s5)
    pf5 = (predict("predict5", concat5) + up) + up      # ElementwiseLayer folds left, :857

    s4 = tuple(conv4_1.shape[1:3])
    concat4 = torch.cat([conv4_1, dec("deconv4", concat5, s4), upflow("upsample5_4", pf5, s4)], 3)
    up = resize_bilinear_legacy(pf5, *s4)
    pf4 = (predict("predict4", concat4) + up) + up      # :866

    s3 = tuple(conv3_1.shape[1:3])
    concat3 = torch.cat([conv3_1, dec("deconv3", concat4, s3), upflow("upsample4_3", pf4, s3)], 3)
    up = resize_bilinear_legacy(pf4, *s3)
    pf3 = (predict("predict3", concat3) + up) + up      # :875

    s2 = tuple(conv2.shape[1:3])
    concat2 = torch.cat([conv2, dec("deconv2", concat3, s2), upflow("upsample3_2", pf3, s2)], 3)
    pf2 = predict2_fullres(concat2, Wt["predict2/W_conv2d"], Wt["predict2/b_conv2d"], H, W)  # :882-885
    up = resize_bilinear_legacy(pf3, H - 2, W - 2)      # :886 literal (382, 510)
    for _ in range(8):                                  # :887, eight sequential adds
        pf2 = pf2 + up

    out = {"predict_flow6": pf6, "predict_flow5": pf5, "predict_flow4": pf4,
           "predict_flow3": pf3, "predict_flow2": pf2, "flow": pf2}
    if return_internals:
        internals.update(concat5=concat5, concat4=concat4, concat3=concat3, concat2=concat2)
        return out, internals
    return out


# --------------------------------------------------------------------------- glue
def flow_to_output_res(pf2, net_h: int, net_w: int, out_h: int, out_w: int):
    """main:497-498 with the literals generalised (384 -> net_h, 512 -> net_w), as the graph's op sequence:
        outflow = resize_images(predict_flow2*384.0/predict_flow2.shape[1], [out_h, out_w])     # (pf2 * 384.0) / 382
        outflow = concat([outflow[...,0:1]*out_w/512, outflow[...,1:2]*out_h/384], 3)           # (f * out_w) / 512, (f * out_h) / 384
    Python parses a*b/c as (a*b)/c and TF builds one op per operator: a multiply by the numerator, then a divide by the
    denominator, each rounded to the tensor's dtype -- NOT one multiply by the quotient (which differs by <= 1 ulp)."""
    dt = pf2.dtype
    c = lambda v: torch.tensor(float(v), dtype=dt)
    f = resize_bilinear_legacy((pf2 * c(net_h)) / c(pf2.shape[1]), out_h, out_w)
    return torch.stack([(f[..., 0] * c(out_w)) / c(net_w), (f[..., 1] * c(out_h)) / c(net_h)], dim=3)


# --------------------------------------------------------------------------- warp
def get_pixel_value(img, x, y):
    """main:44-68: img[b, y, x, :] for int index maps x, y of shape [B, H, W]."""
    B = img.shape[0]
    b = torch.arange(B).view(B, 1, 1).expand_as(x)
    return img[b, y.long(), x.long()]


def tf_warp(img, flow, H: int, W: int, dtype=None):
    """main:70-130, line by line: sample coordinates = pixel grid + flow (fp32, as the
    graph's tensors are), corners by truncation toward zero, all four clipped to the
    image, weights from the CLIPPED corners."""
    dtype = dtype or img.dtype
    img = _t(img, dtype)
    fl = _t(flow, torch.float32)
    gx = torch.arange(W, dtype=torch.float32).view(1, 1, W)
    gy = torch.arange(H, dtype=torch.float32).view(1, H, 1)
    x = gx + fl[..., 0]                                  # main:83,88
    y = gy + fl[..., 1]
    x0 = x.to(torch.int32)                               # tf.cast float->int32 truncates, main:92
    y0 = y.to(torch.int32)
    x1 = x0 + 1
    y1 = y0 + 1
    x0 = x0.clamp(0, W - 1); x1 = x1.clamp(0, W - 1)     # main:98-101
    y0 = y0.clamp(0, H - 1); y1 = y1.clamp(0, H - 1)
    Ia = get_pixel_value(img, x0, y0)                    # main:104-107
    Ib = get_pixel_value(img, x0, y1)
    Ic = get_pixel_value(img, x1, y0)
    Id = get_pixel_value(img, x1, y1)
    xd, yd = x.to(dtype), y.to(dtype)
    x0f, x1f, y0f, y1f = (t.to(dtype) for t in (x0, x1, y0, y1))
    wa = ((x1f - xd) * (y1f - yd)).unsqueeze(3)          # main:117-120
    wb = ((x1f - xd) * (yd - y0f)).unsqueeze(3)
    wc = ((xd - x0f) * (y1f - yd)).unsqueeze(3)
    wd = ((xd - x0f) * (yd - y0f)).unsqueeze(3)
    return wa * Ia + wb * Ib + wc * Ic + wd * Id         # tf.add_n, main:129


def warp_discontinuity_mask(flow, H: int, W: int, delta: float = 1e-2):
    """True where the sample coordinate is farther than `delta` from the lines
    x in {-1, W-1}, y in {-1, H-1}, the only places tf_warp is discontinuous
    (across x = 0 the (-1,0) extrapolation branch and the [0,1) branch
    are the same formula).  Used to compare warped frames computed from flows that
    differ by rounding."""
    fl = _t(flow, torch.float32)
    x = torch.arange(W, dtype=torch.float32).view(1, 1, W) + fl[..., 0]
    y = torch.arange(H, dtype=torch.float32).view(1, H, 1) + fl[..., 1]
    ok = torch.ones_like(x, dtype=torch.bool)
    for v, lim in ((x, W - 1), (y, H - 1)):
        ok &= (v - lim).abs() > delta
        ok &= (v + 1).abs() > delta
    return ok


# --------------------------------------------------------------------------- whole path
def stabilise_originalsize(feats, frame, weights, dtype=torch.float64, flow_filter=None):
    """The graph `evaluate_originalSize` builds (main:491-514): network on `feats`
    [B,Hn,Wn,Cin], flow brought to the output resolution of `frame` [B,oh,ow,3], warp (by `flow_filter(outflow)` if given)."""
    flows = flownetS_pyramid(feats, weights, dtype)
    Hn, Wn = feats.shape[1], feats.shape[2]
    oh, ow = frame.shape[1], frame.shape[2]
    outflow = flow_to_output_res(flows["predict_flow2"], Hn, Wn, oh, ow)
    wf = outflow if flow_filter is None else flow_filter(outflow)
    warped = tf_warp(_t(frame, dtype), wf.to(torch.float32), oh, ow, dtype)
    return flows, outflow, warped


class MeanFlow3:
    """main_flownetS_pyramid_highTV_noBBloss.py:629-631, 679-685: warp by (sum of the previous <= 2 mean flows + this mean flow) / min(i+1, 3)."""

    def __init__(self):
        self.hist = []

    def __call__(self, outflow):
        m = mean_flow(outflow)
        k = min(len(self.hist) + 1, 3)
        s = m.clone()
        for j in range(k - 1):
            s = s + self.hist[-1 - j]
        self.hist.append(m)
        return s / k


def stabilise_native(feats, weights, dtype=torch.float64):
    """The graph `evaluate` builds (main:802-807): warp of the current frame resized to
    the flow's (H-2)x(W-2) grid by predict_flow2 itself."""
    flows = flownetS_pyramid(feats, weights, dtype)
    x = _t(feats, dtype)
    H, W = x.shape[1], x.shape[2]
    unstab = resize_bilinear_legacy(x[..., 24:27] if x.shape[3] >= 27 else x[..., -3:], H - 2, W - 2)
    warped = tf_warp(unstab, flows["predict_flow2"].to(torch.float32), H - 2, W - 2, dtype)
    return flows, warped


# --------------------------------------------------------------------------- secondary samplers
# spatial_transformer.py ("ST") and warp.py of the reference; never executed by its live scripts.
def st_linspace(n: int) -> np.ndarray:
    """tf.linspace(-1.0, 1.0, n) in fp32: start + i*step, step = 2/(n-1)  (ST:766-767)."""
    step = np.float32(2.0) / np.float32(n - 1) if n > 1 else np.float32(0)
    return (np.float32(-1.0) + np.arange(n, dtype=np.float32) * step).astype(np.float32)


def st_meshgrid(out_size) -> np.ndarray:
    """_meshgrid (ST:755-779): flat [3*H*W] = x_t (x fastest), y_t, ones."""
    oh, ow = out_size
    xt, yt = np.meshgrid(st_linspace(ow), st_linspace(oh))
    return np.concatenate([xt.reshape(-1), yt.reshape(-1), np.ones(oh * ow, np.float32)]).astype(np.float32)


def st_bilinear_interp(im, x, y, out_size, dtype=torch.float32):
    """bilinear_interp (ST:902-964), line by line.  im [B,H,W,C]; x, y flat [B*oh*ow] in [-1,1]."""
    im = _t(im, dtype)
    B, H, W, C = im.shape
    imp = F.pad(im, (0, 0, 1, 1, 1, 1))                                   # edge_size = 1, zeros (:905-906)
    x = _t(x, torch.float32).reshape(-1)
    y = _t(y, torch.float32).reshape(-1)
    wf, hf = np.float32(W), np.float32(H)
    x = (x + 1.0) / 2.0 * (wf - 1.0)                                      # :916-917
    y = (y + 1.0) / 2.0 * (hf - 1.0)
    x = torch.clamp(x, -1.0, float(wf - 1 + 1)) + 1.0                     # :918-922
    y = torch.clamp(y, -1.0, float(hf - 1 + 1)) + 1.0
    x0f, y0f = torch.floor(x), torch.floor(y)
    x1f, y1f = x0f + 1, y0f + 1
    x0, y0 = x0f.long(), y0f.long()
    x1 = torch.minimum(x1f, torch.tensor(float(wf - 1 + 2))).long()      # :932-933
    y1 = torch.minimum(y1f, torch.tensor(float(hf - 1 + 2))).long()
    dim2, dim1 = W + 2, (W + 2) * (H + 2)
    npix = out_size[0] * out_size[1]
    base = torch.arange(B).repeat_interleave(npix) * dim1                 # _repeat (:938)
    flat = imp.reshape(-1, C)
    I00, I01 = flat[base + y0 * dim2 + x0], flat[base + y0 * dim2 + x1]
    I10, I11 = flat[base + y1 * dim2 + x0], flat[base + y1 * dim2 + x1]
    xd, yd = x.to(dtype), y.to(dtype)
    x0d, x1d, y0d, y1d = x0f.to(dtype), x1f.to(dtype), y0f.to(dtype), y1f.to(dtype)
    w00 = ((x1d - xd) * (y1d - yd)).unsqueeze(1)                          # :958-961 (unclipped x1_f)
    w01 = ((xd - x0d) * (y1d - yd)).unsqueeze(1)
    w10 = ((x1d - xd) * (yd - y0d)).unsqueeze(1)
    w11 = ((xd - x0d) * (yd - y0d)).unsqueeze(1)
    return w00 * I00 + w01 * I01 + w10 * I10 + w11 * I11


def st_transform(im, theta, out_size, dtype=torch.float32, matmul="blas"):
    """AffineTransformer.transform (ST:400-452) for theta [B,6], ProjectiveTransformer.transform
    (ST:539-608) for theta [B,8].  The reference's tf.matmul(theta, grid) (ST:447, 593) is a 3-term dot
    product per coordinate whose rounding sequence TF does not specify (Eigen contraction, fused
    multiply-add or not by build): matmul="blas" leaves it to torch.matmul, matmul="unfused" evaluates
    (t0*x + t1*y) + t2*1 with every product and sum rounded to fp32 -- the sequence the HIP kernels use.
    The two differ by <= 1 ulp of the source coordinate."""
    im = _t(im, dtype)
    B = im.shape[0]
    th = _t(theta, torch.float32).reshape(B, -1)
    grid = torch.from_numpy(st_meshgrid(out_size)).reshape(3, -1)

    def mm(M):                                                              # M [B,r,3] . grid [3,N]
        if matmul == "blas":
            return torch.matmul(M, grid.unsqueeze(0).expand(B, 3, -1))
        return (M[:, :, 0:1] * grid[0] + M[:, :, 1:2] * grid[1]) + M[:, :, 2:3] * grid[2]

    if th.shape[1] == 6:
        T = mm(th.reshape(B, 2, 3))
        xs, ys = T[:, 0], T[:, 1]
    else:
        th9 = torch.cat([th, torch.ones(B, 1)], 1).reshape(B, 3, 3)
        T = mm(th9)
        z = T[:, 2]
        z = torch.where(z == 0, z + np.float32(1e-8), z)                  # safe_z (:598)
        xs, ys = T[:, 0] / z, T[:, 1] / z
    out = st_bilinear_interp(im, xs.reshape(-1), ys.reshape(-1), out_size, dtype)
    return out.reshape(B, out_size[0], out_size[1], im.shape[3])


def warp_vec2mtrx(p, warp_type: str, warp_approx: int):
    """warp.vec2mtrx (warp.py:25-43) in fp32."""
    p = _t(p, torch.float32)
    B = p.shape[0]
    if warp_type == "homography":
        p1, p2, p3, p4, p5, p6, p7, p8 = p.unbind(1)
        A = torch.stack([torch.stack([p3, p2, p1], 1), torch.stack([p6, -p3 - p7, p5], 1), torch.stack([p4, p8, p7], 1)], 1)
    else:
        O = torch.zeros(B)
        p1, p2, p3, p4, p5, p6 = p.unbind(1)
        A = torch.stack([torch.stack([p1, p2, p3], 1), torch.stack([p4, p5, p6], 1), torch.stack([O, O, O], 1)], 1)
    pM = torch.eye(3).repeat(B, 1, 1)
    numer = torch.eye(3).repeat(B, 1, 1)
    denom = 1.0
    for i in range(1, warp_approx):
        numer = torch.matmul(numer, A)
        denom *= i
        pM = pM + numer / denom
    return pM


def warp_compose(ref, pM):
    """refMtrx . pMtrx of warp.py:48-49 / 91-92 (tf.matmul(refMtrx, pMtrx)) in fp32 with every product and sum rounded,
    (r0*p0 + r1*p1) + r2*p2 -- TF does not specify the rounding sequence of its 3-term dot products; this is the one the HIP
    kernels use.  ref [3,3], pM [B,3,3] -> [B,3,3]."""
    ref = _t(ref, torch.float32).reshape(3, 3)
    pM = _t(pM, torch.float32).reshape(-1, 3, 3)
    r = ref.unsqueeze(0)
    return (r[:, :, 0:1] * pM[:, 0:1, :] + r[:, :, 1:2] * pM[:, 1:2, :]) + r[:, :, 2:3] * pM[:, 2:3, :]


def warp_transform_image(image, M, oh: int, ow: int, dtype=torch.float32, matmul="blas"):
    """warp.transformImage / transformCropImage (warp.py:46-86, 89-129) given M = refMtrx . pMtrx [B,3,3];
    image [B,Hi,Wi,C] -> [B,oh,ow,C].  `matmul` as in st_transform (warp.py:55, 98 are tf.matmul too)."""
    image = _t(image, dtype)
    B, Hi, Wi, C = image.shape
    M = _t(M, torch.float32).reshape(B, 3, 3)
    X, Y = np.meshgrid(np.linspace(-1, 1, ow), np.linspace(-1, 1, oh))
    XYhom = torch.from_numpy(np.stack([X.flatten(), Y.flatten(), np.ones(oh * ow)], 0).astype(np.float32))
    if matmul == "blas":
        W3 = torch.matmul(M, XYhom.unsqueeze(0).expand(B, 3, -1))
    else:
        W3 = (M[:, :, 0:1] * XYhom[0] + M[:, :, 1:2] * XYhom[1]) + M[:, :, 2:3] * XYhom[2]
    xw = (W3[:, 0] / (W3[:, 2] + np.float32(1e-8))).reshape(B, oh, ow)
    yw = (W3[:, 1] / (W3[:, 2] + np.float32(1e-8))).reshape(B, oh, ow)
    xf, xc, yf, yc = torch.floor(xw), torch.ceil(xw), torch.floor(yw), torch.ceil(yw)
    xfi, xci, yfi, yci = (t.clamp(-1e9, 1e9).long() for t in (xf, xc, yf, yc))
    vec = torch.cat([image.reshape(-1, C), torch.zeros(1, C, dtype=dtype)], 0)
    bidx = torch.arange(B).view(B, 1, 1)
    outside = B * Hi * Wi

    def gather(xi, yi):
        inside = (xi >= 0) & (xi < Wi) & (yi >= 0) & (yi < Hi)
        idx = torch.where(inside, (bidx * Hi + yi) * Wi + xi, torch.full_like(xi, outside))
        return vec[idx]

    xr = (xw - xf).to(dtype).unsqueeze(3)
    yr = (yw - yf).to(dtype).unsqueeze(3)
    UL = gather(xfi, yfi) * (1 - xr) * (1 - yr)
    UR = gather(xci, yfi) * xr * (1 - yr)
    BL = gather(xfi, yci) * (1 - xr) * yr
    BR = gather(xci, yci) * xr * yr
    return UL + UR + BL + BR


# --------------------------------------------------------------------------- VGG16 trunk
VGG_LAYERS = (("conv1_1", "conv1_2"), ("conv2_1", "conv2_2"), ("conv3_1", "conv3_2", "conv3_3"),
              ("conv4_1", "conv4_2", "conv4_3"), ("conv5_1", "conv5_2", "conv5_3"))


def vgg16_build(x, data_dict, dtype=torch.float32):
    """Vgg16.build (vgg16.py:25-48): conv 3x3 stride 1 SAME + bias + ReLU (:55-64), max pool 2x2
    stride 2 SAME (:51-53; -inf padding on the high side for odd sizes).  Returns {name: NHWC tensor}."""
    cur = _t(x, dtype).permute(0, 3, 1, 2)
    out = {}
    for bi, block in enumerate(VGG_LAYERS, 1):
        for name in block:
            W = _t(data_dict[name][0], dtype).permute(3, 2, 0, 1)
            b = _t(data_dict[name][1], dtype)
            cur = torch.relu(F.conv2d(cur, W, b, stride=1, padding=1))
            out[name] = cur.permute(0, 2, 3, 1).contiguous()
        cur = F.max_pool2d(cur, 2, 2, ceil_mode=True)
        out[f"pool{bi}"] = cur.permute(0, 2, 3, 1).contiguous()
    return out


def vgg_preprocess(x, dtype=torch.float32):
    """NLDF.py:29: input * 255. - VGG_MEAN (vgg16.py:7)."""
    return _t(x, dtype) * 255.0 - torch.tensor([103.939, 116.779, 123.68], dtype=dtype)


# --------------------------------------------------------------------------- NLDF head (NLDF.py:24-101)
def nldf_build_model(x, vgg_dict, hw, dtype=torch.float64):
    """Model.build_model (NLDF.py:24-101) on a [B,352,352,3] input in [0,1].  `hw` = head variables
    {'<layer>/W', '<layer>/b'}.  Returns dict with Prob, Score, Local_Fea, Fea_Global."""
    v = vgg16_build(vgg_preprocess(x, dtype), vgg_dict, dtype)             # :29-31
    W = {k: _t(a, dtype) for k, a in hw.items()}

    def conv(t, name, pad):                                               # Conv_2d (:103-114)
        w = W[name + "/W"].permute(3, 2, 0, 1)
        return F.conv2d(t, w, W[name + "/b"], stride=1, padding=pad)

    def deconv(t, name, out_hw):                                          # Deconv_2d 5x5 s2 SAME (:116-129)
        w = W[name + "/W"].permute(3, 2, 0, 1)                           # [Cin, Cout, kh, kw]
        y = F.conv_transpose2d(t, w, W[name + "/b"], stride=2, padding=1)
        return y[:, :, :out_hw, :out_hw]

    def contrast(t):                                                      # Contrast_Layer (:131-134)
        return t - F.avg_pool2d(F.pad(t, (1, 1, 1, 1), mode="replicate"), 3, 1)

    nchw = lambda t: t.permute(0, 3, 1, 2)
    p1, p2, p3, p4, p5 = (nchw(v[f"pool{i}"]) for i in range(1, 6))
    g1 = torch.relu(conv(p5, "Fea_Global_1", 0))
    g2 = torch.relu(conv(g1, "Fea_Global_2", 0))
    fg = conv(g2, "Fea_Global", 0)
    fp = [torch.relu(conv(p, f"Fea_P{i}", 1)) for i, p in zip(range(1, 6), (p1, p2, p3, p4, p5))]
    lc = [contrast(t) for t in fp]
    up5 = torch.relu(deconv(torch.cat([fp[4], lc[4]], 1), "Fea_P5_Deconv", 22))
    up4 = torch.relu(deconv(torch.cat([fp[3], lc[3], up5], 1), "Fea_P4_Deconv", 44))
    up3 = torch.relu(deconv(torch.cat([fp[2], lc[2], up4], 1), "Fea_P3_Deconv", 88))
    up2 = torch.relu(deconv(torch.cat([fp[1], lc[1], up3], 1), "Fea_P2_Deconv", 176))
    lf = conv(torch.cat([fp[0], lc[0], up2], 1), "Local_Fea", 0)
    ls = conv(lf, "Local_Score", 0)
    gs = conv(fg, "Global_Score", 0)
    score = (ls + gs).permute(0, 2, 3, 1)
    prob = torch.softmax(score, dim=3)[..., 0:1]
    return {"Prob": prob, "Score": score, "Local_Fea": lf.permute(0, 2, 3, 1), "Fea_Global": fg.permute(0, 2, 3, 1)}


# --------------------------------------------------------------------------- clip driver (main:535-630)
def cv_resize_u8(src: np.ndarray, dh: int, dw: int) -> np.ndarray:
    """cv2.resize(src, (dw, dh)) INTER_LINEAR for uint8 HxWxC, restating OpenCV's 8-bit path (imgproc/resize.cpp:
    half-pixel centres, 11-bit coefficients, HResizeLinear + VResizeLinear<uchar,int,short>).  cv2 is not
    installed here; the restatement is pinned by hand-derived known answers of the published fixed-point formula and by an
    independent float bilinear at half-pixel centres (scipy.ndimage) within 1 LSB (tests/test_oracle_kat.py).  What stays
    unverifiable offline: cv2 builds that dispatch 8-bit linear resizing to a vendor library (IPP) may differ by 1 LSB."""
    sh, sw = src.shape[:2]

    def taps(dn, sn):
        scale = np.float64(1.0) / (np.float64(dn) / np.float64(sn))        # cv::resize: inv_scale = (double)dsize / ssize; hal::resize: scale = 1. / inv_scale
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)     # fx = (float)((dx+0.5)*scale_x - 0.5)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= sn - 1
        f[hi] = 0; s[hi] = sn - 1
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, np.minimum(s + 1, sn - 1), a0, a1

    x0, x1, ax0, ax1 = taps(dw, sw)
    y0, y1, ay0, ay1 = taps(dh, sh)
    S = src.astype(np.int64)
    R = S[:, x0] * ax0[None, :, None] + S[:, x1] * ax1[None, :, None]                 # [sh, dw, C]
    R0, R1 = R[y0], R[y1]
    v = (((ay0[:, None, None] * (R0 >> 4)) >> 16) + ((ay1[:, None, None] * (R1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------- homography evaluator
def _homog_hash(x):
    x = np.asarray(x, dtype=np.uint64) & np.uint64(0xffffffff)
    m = np.uint64(0xffffffff)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & m
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & m
    x ^= x >> np.uint64(16)
    return x


def homography_fit(flow, K=256, seed=0, thresh=3.0, refine=2, stride=1, sample_index=0):
    """The estimator behind `h, mask = cv2.findHomography(gridmesh, gridmesh - flow, cv2.RANSAC)` (main:728-735) as
    the HIP library defines it (include/vstab.h: cv2's RNG cannot be restated, the estimator can): K 4-point
    hypotheses from hash-drawn pixels, consensus under a `thresh` px reprojection error, arg-max with the lowest
    index on ties, `refine` least-squares refits on the inliers.  flow [H,W,2] float32 -> (3x3 float64, inliers)."""
    flow = np.asarray(flow, dtype=np.float32)
    H, W = flow.shape[:2]
    npix = H * W
    cx, cy = 0.5 * (W - 1), 0.5 * (H - 1)
    sc = 1.0 / max(cx, cy, 1.0)
    ys, xs = np.divmod(np.arange(npix), W)
    f = flow.reshape(npix, 2).astype(np.float64)
    x, y = (xs - cx) * sc, (ys - cy) * sc
    u, v = (xs - f[:, 0] - cx) * sc, (ys - f[:, 1] - cy) * sc          # gridmeshOF = gridmesh - curoutflow
    thr2 = thresh * sc * thresh * sc

    def rows(i):
        z, o = np.zeros_like(x[i]), np.ones_like(x[i])
        r0 = np.stack([x[i], y[i], o, z, z, z, -u[i] * x[i], -u[i] * y[i]], -1)
        r1 = np.stack([z, z, z, x[i], y[i], o, -v[i] * x[i], -v[i] * y[i]], -1)
        return r0, r1

    def inliers_of(h, sel):
        w = h[6] * x[sel] + h[7] * y[sel] + h[8]
        with np.errstate(all="ignore"):
            du = (h[0] * x[sel] + h[1] * y[sel] + h[2]) / w - u[sel]
            dv = (h[3] * x[sel] + h[4] * y[sel] + h[5]) / w - v[sel]
            return du * du + dv * dv <= thr2

    scored = np.arange(0, npix, stride)
    best, best_cnt = None, -1
    for k in range(K):
        t = sample_index * K + k
        ctr = (np.uint64(seed) + np.uint64(0x9E3779B9) * (np.uint64(t * 4) + np.arange(1, 5, dtype=np.uint64))) & np.uint64(0xffffffff)
        idx = ((_homog_hash(ctr) * np.uint64(npix)) >> np.uint64(32)).astype(np.int64)
        r0, r1 = rows(idx)
        A = np.empty((8, 8)); rhs = np.empty(8)
        A[0::2], A[1::2] = r0, r1
        rhs[0::2], rhs[1::2] = u[idx], v[idx]
        try:
            if abs(np.linalg.det(A)) < 1e-30:
                continue
            h = np.append(np.linalg.solve(A, rhs), 1.0)
        except np.linalg.LinAlgError:
            continue
        cnt = int(inliers_of(h, scored).sum())
        if cnt > best_cnt:
            best, best_cnt = h, cnt
    if best is None:
        return np.full((3, 3), np.nan), 0
    h, n_in = best, 0
    allp = np.arange(npix)
    for _ in range(refine):
        m = inliers_of(h, allp)
        n_in = int(m.sum())
        if n_in < 4:
            break
        r0, r1 = rows(np.nonzero(m)[0])
        A = np.concatenate([r0, r1]); rhs = np.concatenate([u[m], v[m]])
        try:
            h = np.append(np.linalg.solve(A.T @ A, A.T @ rhs), 1.0)
        except np.linalg.LinAlgError:
            break
    Hn = h.reshape(3, 3)
    T = np.array([[sc, 0, -cx * sc], [0, sc, -cy * sc], [0, 0, 1.0]])
    M = np.linalg.inv(T) @ Hn @ T
    return M / M[2, 2], n_in


def cv_warp_perspective_u8(src, M, oh, ow):
    """cv2.warpPerspective(src, M, (ow, oh)) on uint8 HxWxC (main:736): default INTER_LINEAR / BORDER_CONSTANT 0, M maps
    src -> dst.  Restates OpenCV's published 8-bit path (imgwarp.cpp: coordinates rounded to 1/32 px, 15-bit fixed-point
    bilinear weights); cv2 is not installed here: pinned by hand-derived known answers and by an exact float bilinear at the
    1/32-pixel coordinates (tests/test_oracle_kat.py), not against the library itself."""
    sh, sw = src.shape[:2]
    Mi = np.linalg.inv(np.asarray(M, dtype=np.float64))
    dx, dy = np.meshgrid(np.arange(ow, dtype=np.float64), np.arange(oh, dtype=np.float64))
    X0 = Mi[0, 0] * dx + Mi[0, 1] * dy + Mi[0, 2]
    Y0 = Mi[1, 0] * dx + Mi[1, 1] * dy + Mi[1, 2]
    Wd = Mi[2, 0] * dx + Mi[2, 1] * dy + Mi[2, 2]
    with np.errstate(all="ignore"):
        Wd = np.where(Wd != 0, 32.0 / Wd, 0.0)
    X = np.rint(np.clip(X0 * Wd, -2147483648.0, 2147483647.0)).astype(np.int64)
    Y = np.rint(np.clip(Y0 * Wd, -2147483648.0, 2147483647.0)).astype(np.int64)
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    ax, ay = X & 31, Y & 31
    S = src.astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        return np.where(ok[..., None], S[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)], 0)

    w = [(32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32]
    acc = (w[0][..., None] * tap(sy, sx) + w[1][..., None] * tap(sy, sx + 1) + w[2][..., None] * tap(sy + 1, sx)
           + w[3][..., None] * tap(sy + 1, sx + 1) + (1 << 14)) >> 15
    return np.clip(acc, 0, 255).astype(np.uint8)


def clip_loop(frames_bgr_u8: np.ndarray, weights, net_hw, dtype=torch.float32, teacher=None, flow_filter=None):
    """The loop of evaluate_originalSize (main:540-630) on a clip [T,H,W,3] uint8 BGR -> stabilised uint8 clip.
    `teacher` ([T,H,W,3] uint8, another implementation's outputs): if given, frame i is still computed here but the
    HISTORY later frames read is the teacher's frame i -- a per-frame check that cannot drift (the free-running loop
    amplifies a one-LSB difference through the warp's out-of-range mask within a few frames)."""
    T, H, W, _ = frames_bgr_u8.shape
    nh, nw = net_hw
    lags = (31, 23, 15, 7, 4, 3, 2, 1)                                             # main:553
    total = np.zeros((T, H, W, 3), np.float64)                                     # totaloutputFrame (main:534)
    outs = []
    for i in range(T):
        frame = frames_bgr_u8[i]
        if i == 0:
            total[0] = frame                                                       # main:548-549
        cur = np.zeros((1, nh, nw, 27), np.float32)
        cur[0, :, :, 24:27] = cv_resize_u8(frame, nh, nw)[..., ::-1] / 255.0      # main:550
        for j, lag in enumerate(lags):                                             # main:554-558
            src = total[0] if i - lag < 0 else total[i - lag]
            q = np.clip(np.trunc(src), 0, 255).astype(np.uint8)                    # np.uint8(...)
            cur[0, :, :, 3 * j:3 * j + 3] = np.float32(cv_resize_u8(q, nh, nw)[..., ::-1]) / 255.0
        frame_f = (frame[..., ::-1] / 255.0).astype(np.float32)[None]              # main:568
        _, _, warped = stabilise_originalsize(cur, frame_f, weights, dtype, flow_filter)        # main:569
        total[i] = (warped[0].numpy().astype(np.float64) * 255.0)[..., ::-1]       # main:625
        outs.append(np.clip(np.trunc(total[i]), 0, 255).astype(np.uint8))          # main:630
        if teacher is not None:
            total[i] = teacher[i].astype(np.float64)
    return np.stack(outs)


def cv_resize_f32(src: np.ndarray, dh: int, dw: int) -> np.ndarray:
    """cv2.resize(src, (dw, dh)) INTER_LINEAR on a float32 HxWxC image (main:861): the taps of cv_resize_u8 with float coefficients,
    horizontal pass then vertical pass in float32.  Pinned against an independent float bilinear (scipy.ndimage), not against cv2 itself."""
    sh, sw = src.shape[:2]

    def taps(dn, sn):
        scale = np.float64(1.0) / (np.float64(dn) / np.float64(sn))        # cv::resize: inv_scale = (double)dsize / ssize; hal::resize: scale = 1. / inv_scale
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)     # fx = (float)((dx+0.5)*scale_x - 0.5)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= sn - 1
        f[hi] = 0; s[hi] = sn - 1
        return s, np.minimum(s + 1, sn - 1), (np.float32(1) - f).astype(np.float32), f

    x0, x1, ax0, ax1 = taps(dw, sw)
    y0, y1, ay0, ay1 = taps(dh, sh)
    S = src.astype(np.float32)
    R = (S[:, x0] * ax0[None, :, None] + S[:, x1] * ax1[None, :, None]).astype(np.float32)
    return (ay0[:, None, None] * R[y0] + ay1[:, None, None] * R[y1]).astype(np.float32)


def native_clip_loop(frames_bgr_u8: np.ndarray, weights, net_hw, dtype=torch.float32, flow_filter=None, teacher=None):
    """The loop of evaluate() (main:831-863) on a clip [T,H,W,3] uint8 BGR -> [T,384,512,3] uint8; `flow_filter(of)` maps predict_flow2 to
    the flow that warps (evaluate_blurNma / evaluate_medianNma); `teacher` as in clip_loop."""
    T = frames_bgr_u8.shape[0]
    nh, nw = net_hw
    lags = (1, 2, 3, 4, 7, 15, 23, 31)                                             # main:844
    total = np.zeros((T, nh, nw, 3), np.float64)
    outs = []
    for i in range(T):
        small = cv_resize_u8(frames_bgr_u8[i], nh, nw)
        if i == 0:
            total[0] = small                                                       # main:841
        cur = np.zeros((1, nh, nw, 27), np.float64)
        cur[0, :, :, 24:27] = small[..., ::-1] / 255.0                             # main:842
        for j, lag in enumerate(lags):                                             # main:845-849
            src = total[0] if i - lag < 0 else total[i - lag]
            q = np.clip(np.trunc(src), 0, 255).astype(np.uint8)
            cur[0, :, :, 3 * j:3 * j + 3] = np.float32(q[..., ::-1]) / 255.0
        flows = flownetS_pyramid(cur, weights, dtype=dtype)
        of = flows["predict_flow2"]
        flow = of if flow_filter is None else flow_filter(of)
        unstab = resize_bilinear_legacy(_t(cur[..., 24:27], dtype), nh - 2, nw - 2)   # main:806
        warped = tf_warp(unstab, flow, nh - 2, nw - 2, dtype)[0].numpy().astype(np.float32)
        total[i] = (cv_resize_f32(warped, nh, nw) * np.float32(255))[..., ::-1]    # main:861
        outs.append(np.clip(np.trunc(total[i]), 0, 255).astype(np.uint8))          # main:863
        if teacher is not None:
            total[i] = teacher[i].astype(np.float64)
    return np.stack(outs)


# --------------------------------------------------------------------------- training objective
LOSS_LEVELS = ("predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3", "predict_flow2")
TV_WEIGHTS = (2e-8 * 3, 2e-8 * 3, 2e-8 * 3, 4e-8 * 1.5, 4e-8 * 1.5)          # main:269-273


def masked_mse(pred, gt, mask):
    """main:188-198."""
    mse = ((pred * mask - gt * mask) ** 2).sum(dim=(1, 2, 3))
    safe = torch.where(mask == 0, mask + 1e-8, mask)
    return (mse / safe.sum(dim=(1, 2, 3))).mean()


def lossterm(predict_flow, stab_image, unstab_image, dtype=torch.float64):
    """main:200-210: both images resized to the flow's size (tf.image.resize_images), the unstable one warped by the
    flow, compared with the stable one under the mask tf_warp(ones)."""
    h, w = predict_flow.shape[1], predict_flow.shape[2]
    ds = resize_bilinear_legacy(_t(stab_image, dtype), h, w)
    du = resize_bilinear_legacy(_t(unstab_image, dtype), h, w)
    warped = tf_warp(du, predict_flow, h, w, dtype)
    mask = tf_warp(torch.ones_like(du), predict_flow, h, w, dtype)
    return masked_mse(warped, ds, mask), warped


def total_variation(flow):
    """tf.image.total_variation summed over the batch (main:269)."""
    f = flow.double()
    return (f[:, 1:] - f[:, :-1]).abs().sum() + (f[:, :, 1:] - f[:, :, :-1]).abs().sum()


def loss_main(flows, gtstab, unstab, dtype=torch.float64):
    """loss_main of the training graph (main:213-217, 269-275): five lossterms against the ground-truth stable frame
    plus the weighted total variation of every flow.  `flows`: dict of float32 tensors (requires_grad for a backward)."""
    total = 0.0
    for name, tvw in zip(LOSS_LEVELS, TV_WEIGHTS):
        l, _ = lossterm(flows[name], gtstab, unstab, dtype)
        total = total + l + tvw * total_variation(flows[name])
    return total


# --------------------------------------------------------------------------- flow post-filters
def box_blur_flow(flow, k: int = 75, dtype=torch.float64):
    """tf.nn.conv2d(of_c, constant(1/(k*k), [k,k,1,1]), SAME) per channel (main_flownetS_pyramid.py:634-641)."""
    f = _t(flow, dtype).permute(0, 3, 1, 2)
    w = torch.full((2, 1, k, k), 1.0 / (k * k), dtype=dtype)
    return F.conv2d(f, w, padding=(k - 1) // 2, groups=2).permute(0, 2, 3, 1)


def medfilt_flow(flow, kernel_size=5):
    """scipy.signal.medfilt(np.squeeze(of), kernel_size) per sample (main_flownetS_pyramid.py:809), restated without
    scipy: zero-pad every axis of the [h,w,2] field (channels too: a scalar size applies to all three axes), sort
    each kh*kw*kc window, take element n//2.  tests/test_oracle_kat.py pins this against scipy itself."""
    f = _t(flow, torch.float64)
    ks = (int(kernel_size),) * 3 if isinstance(kernel_size, int) else tuple(int(k) for k in kernel_size)
    kh, kw, kc = ks
    B, h, w, C = f.shape
    p = F.pad(f, (kc // 2, kc // 2, kw // 2, kw // 2, kh // 2, kh // 2))          # pads last dim first: c, w, h
    win = p.unfold(1, kh, 1).unfold(2, kw, 1).unfold(3, kc, 1).reshape(B, h, w, C, kh * kw * kc)
    return win.sort(dim=-1).values[..., (kh * kw * kc) // 2]


def mean_flow(flow, dtype=torch.float64):
    """main_flownetS_pyramid_highTV_noBBloss.py:629."""
    f = _t(flow, dtype)
    return f.mean(dim=(1, 2), keepdim=True).expand_as(f)
