"""Known-answer tests that pin the oracle: every expectation here is
derivable by hand from the op definitions, none comes from running the oracle."""
import numpy as np
import pytest
import torch

from oracle import vstab_oracle as vo
from oracle import c_oracle as co

F64 = torch.float64


def test_identity_flow_zeroes_last_row_and_column():
    # main:98-120: at x = W-1 both corners clip to W-1 and the weights cancel
    img = torch.rand(2, 5, 7, 3, dtype=F64) + 1.0
    out = vo.tf_warp(img, torch.zeros(2, 5, 7, 2), 5, 7)
    assert torch.equal(out[:, :4, :6], img[:, :4, :6])
    assert torch.all(out[:, 4] == 0) and torch.all(out[:, :, 6] == 0)
    assert np.array_equal(co.tf_warp(img.numpy(), np.zeros((2, 5, 7, 2), np.float32)), out.numpy())


def test_warp_edge_semantics():
    # one row image I = [10, 20, 40, 80]; H=2 so that y stays interior on row 0
    img = torch.tensor([10., 20., 40., 80.], dtype=F64).view(1, 1, 4, 1).repeat(1, 2, 1, 1)
    def at(xflow):
        fl = torch.zeros(1, 2, 4, 2); fl[0, 0, 0, 0] = xflow      # pixel (0,0) samples x = xflow
        return float(vo.tf_warp(img, fl, 2, 4)[0, 0, 0, 0])
    assert at(0.5) == 15.0                     # plain bilinear
    assert at(2.25) == 0.75 * 40 + 0.25 * 80
    assert at(-0.5) == 1.5 * 10 - 0.5 * 20     # trunc(-0.5)=0 -> extrapolation (A.6)
    assert at(-1.0) == 0.0                     # both corners clip to 0, weights cancel
    assert at(-7.3) == 0.0
    assert at(3.0) == 0.0                      # x = W-1 exactly
    assert at(100.0) == 0.0


def test_warp_mask_trick():
    # main:206: warping ones gives the validity mask
    fl = torch.zeros(1, 4, 4, 2); fl[..., 0] = 1.0
    m = vo.tf_warp(torch.ones(1, 4, 4, 1, dtype=F64), fl, 4, 4)[0, :, :, 0]
    exp = torch.zeros(4, 4, dtype=F64); exp[:3, :2] = 1.0
    assert torch.equal(m, exp)


def test_legacy_bilinear_2x_pattern():
    # scale = 0.5 exactly: out[2k] = in[k], out[2k+1] = (in[k] + in[min(k+1,n-1)])/2
    x = torch.tensor([1., 3., 7., 15.], dtype=F64).view(1, 1, 4, 1)
    y = vo.resize_bilinear_legacy(x, 1, 8)[0, 0, :, 0]
    assert y.tolist() == [1, 2, 3, 5, 7, 11, 15, 15]
    x2 = torch.arange(6, dtype=F64).view(1, 2, 3, 1)
    y2 = vo.resize_bilinear_legacy(x2, 4, 6)[0, :, :, 0]
    assert y2[0].tolist() == [0, .5, 1, 1.5, 2, 2]
    assert y2[1].tolist() == [1.5, 2, 2.5, 3, 3.5, 3.5]
    assert y2[3].tolist() == [3, 3.5, 4, 4.5, 5, 5]
    assert np.array_equal(co.resize_bilinear(x2.numpy(), 4, 6)[0, :, :, 0], y2.numpy())


def test_legacy_bilinear_same_size_is_identity_and_downscale_samples_topleft():
    x = torch.rand(1, 4, 6, 2, dtype=F64)
    assert vo.resize_bilinear_legacy(x, 4, 6) is x
    y = vo.resize_bilinear_legacy(x, 2, 3)            # scale 2: f = 2i, t = 0 -> plain subsample
    assert torch.equal(y, x[:, ::2, ::2])


def test_nearest_align_corners_index():
    # 98 -> 384 (model.py:883 on the padded concat2): ends map to ends, roundf half away
    idx = vo.nearest_align_corners_index(98, 384)
    assert idx[0] == 0 and idx[-1] == 97 and np.all(np.diff(idx) >= 0) and np.diff(idx).max() == 1
    assert vo.nearest_align_corners_index(3, 5).tolist() == [0, 1, 1, 2, 2]   # .5 -> 1, 1.5 -> 2
    for n_in, n_out in ((98, 384), (130, 512), (66, 256), (3, 5), (182, 720), (322, 1280)):
        ref = vo.nearest_align_corners_index(n_in, n_out)
        assert [co.nearest_index(i, n_in, n_out) for i in range(n_out)] == ref.tolist()


def test_deconv_impulse_response():
    # A.2: o = 2 i + k - 1; a unit impulse at (1,1) paints W[ky,kx] at rows/cols 1..4
    x = torch.zeros(1, 3, 3, 1, dtype=F64); x[0, 1, 1, 0] = 1.0
    W = torch.arange(16, dtype=F64).view(4, 4, 1, 1) + 1
    y = vo.deconv4x4s2(x, W, torch.zeros(1, dtype=F64), (6, 6))[0, :, :, 0]
    exp = torch.zeros(6, 6, dtype=F64); exp[1:5, 1:5] = W[:, :, 0, 0]
    assert torch.equal(y, exp)
    # odd output_shape (ceil(5/2) == 3) is the same thing cropped
    y5 = vo.deconv4x4s2(x, W, torch.zeros(1, dtype=F64), (5, 5))[0, :, :, 0]
    assert torch.equal(y5, exp[:5, :5])
    # impulse at the corner: taps with o = -1 fall off
    x0 = torch.zeros(1, 3, 3, 1, dtype=F64); x0[0, 0, 0, 0] = 1.0
    y0 = vo.deconv4x4s2(x0, W, torch.zeros(1, dtype=F64), (6, 6))[0, :, :, 0]
    assert torch.equal(y0[:3, :3], W[1:, 1:, 0, 0]) and y0[3:].abs().sum() == 0


def test_deconv_channel_layout_is_hw_out_in():
    # W[ky,kx,co,ci] (model.py:850 shape=(4,4,512,1024) maps 1024 -> 512)
    x = torch.zeros(1, 1, 1, 3, dtype=F64); x[0, 0, 0, 2] = 1.0
    W = torch.zeros(4, 4, 2, 3, dtype=F64); W[1, 1, 1, 2] = 5.0
    y = vo.deconv4x4s2(x, W, torch.tensor([0.5, 0.25], dtype=F64), (2, 2))
    assert y[0, 0, 0].tolist() == [0.5, 5.25]


def test_pad_conv_is_cross_correlation_with_zero_pad():
    x = torch.zeros(1, 4, 4, 1, dtype=F64); x[0, 0, 0, 0] = 1.0
    W = torch.arange(9, dtype=F64).view(3, 3, 1, 1)
    y = vo.pad_conv(x, W, torch.zeros(1, dtype=F64), 1, 1)[0, :, :, 0]
    # output (oy,ox) sees the impulse at tap (ky,kx) = (1-oy, 1-ox)
    assert y[0, 0] == 4 and y[0, 1] == 3 and y[1, 0] == 1 and y[1, 1] == 0 and y[2:].abs().sum() == 0
    ys = vo.pad_conv(x, W, torch.zeros(1, dtype=F64), 1, 2)[0, :, :, 0]
    assert ys.shape == (2, 2) and ys[0, 0] == 4 and ys[0, 1] == 0


def test_bn_lrelu_formula():
    x = torch.tensor([[-2.0, 3.0]], dtype=F64).view(1, 1, 1, 2)
    y = vo.bn_lrelu(x, torch.tensor([0.5, -1.0], dtype=F64), torch.tensor([1.0, 1.0], dtype=F64),
                    torch.tensor([4.0 - 1e-5, 1.0 - 1e-5], dtype=F64))
    assert torch.allclose(y.flatten(), torch.tensor([0.1 * (-1.5 + 0.5), 2.0 - 1.0], dtype=F64), atol=1e-12)


def test_flow_glue_constants():
    # main:497-498 at the reference's native numbers: 384/382 pre-scale, identity post-scale
    pf2 = torch.ones(1, 382, 510, 2, dtype=F64)
    f = vo.flow_to_output_res(pf2, 384, 512, 384, 512)
    assert f.shape == (1, 384, 512, 2)
    assert torch.allclose(f, torch.full_like(f, 384.0 / 382.0), atol=1e-15)
    f2 = vo.flow_to_output_res(pf2, 384, 512, 768, 1024)
    assert torch.allclose(f2[..., 0] / f[0, 0, 0, 0], torch.full((1, 768, 1024), 2.0, dtype=F64))


def test_flow_glue_is_the_graphs_op_sequence_not_one_multiply():
    # main:497-498: `pf2*384.0/382` is (pf2 * 384.0) / 382 -- two TF ops, two fp32 roundings -- and `f*out_h/384` likewise.
    # One multiply by the fp32 quotient (what rounds 1-2 restated) differs in the last bit for about a third of all inputs;
    # these known answers tell the two forms apart.
    pf2 = torch.linspace(-40.0, 40.0, 382 * 4, dtype=torch.float32).view(1, 382, 4, 1).repeat(1, 1, 1, 2).contiguous()
    lit = vo.flow_to_output_res(pf2, 384, 512, 382, 4)   # same size: resize_images returns its input, only the arithmetic acts
    nh, dh = np.float32(384), np.float32(382)
    want_x = ((pf2[..., 0].numpy() * nh) / dh * np.float32(4)) / np.float32(512)
    want_y = ((pf2[..., 1].numpy() * nh) / dh * np.float32(382)) / np.float32(384)
    assert np.array_equal(lit[..., 0].numpy(), want_x) and np.array_equal(lit[..., 1].numpy(), want_y)
    one_mul = pf2[..., 1].numpy() * (nh / dh) * (np.float32(382) / np.float32(384))
    assert (one_mul != want_y).mean() > 0.2              # the single-multiply form is a different function
    assert np.abs(one_mul - want_y).max() <= 2 * np.spacing(np.float32(40.0))


# --------------------------------------------------------------------------- secondary samplers
def test_st_identity_theta_reproduces_image_and_meshgrid():
    # identity affine theta on the same output size samples every pixel centre exactly
    im = torch.rand(2, 5, 7, 3, dtype=F64)
    th = torch.tensor([[1., 0, 0, 0, 1, 0]]).repeat(2, 1)
    out = vo.st_transform(im, th, (5, 7), F64)
    assert torch.allclose(out, im, atol=1e-6)
    g = vo.st_meshgrid((2, 3)).reshape(3, 6)
    assert g[0].tolist() == [-1, 0, 1, -1, 0, 1] and g[1].tolist() == [-1, -1, -1, 1, 1, 1] and g[2].tolist() == [1] * 6


def test_st_zero_border_and_clipping():
    # a shift by one pixel to the right pulls zeros in from the 1-pixel border; far outside is all zero
    im = torch.ones(1, 4, 4, 1, dtype=F64)
    shift = 2.0 / 3.0                                  # one pixel in normalised units for W = 4
    out = vo.st_transform(im, torch.tensor([[1., 0, -shift, 0, 1, 0]]), (4, 4), F64)[0, :, :, 0]
    assert torch.allclose(out[:, 0], torch.zeros(4, dtype=F64), atol=1e-6) and torch.allclose(out[:, 1:], torch.ones(4, 3, dtype=F64), atol=1e-6)
    far = vo.st_transform(im, torch.tensor([[1., 0, 10, 0, 1, 0]]), (4, 4), F64)
    assert far.abs().max() == 0
    half = vo.st_transform(im, torch.tensor([[1., 0, -shift / 2, 0, 1, 0]]), (4, 4), F64)[0, :, :, 0]
    assert torch.allclose(half[:, 0], torch.full((4,), 0.5, dtype=F64), atol=1e-6)     # halfway into the zero border


def test_projective_equals_affine_when_bottom_row_is_zero():
    im = torch.rand(1, 6, 6, 2, dtype=F64)
    a = torch.tensor([[0.9, 0.1, 0.05, -0.1, 1.1, 0.0]])
    p = torch.cat([a, torch.zeros(1, 2)], 1)
    assert torch.allclose(vo.st_transform(im, a, (5, 4), F64), vo.st_transform(im, p, (5, 4), F64), atol=1e-12)


def test_vec2mtrx_is_truncated_expm():
    # homography generator A = [[p3,p2,p1],[p6,-p3-p7,p5],[p4,p8,p7]] (trace 0); one term = identity
    p = torch.tensor([[0.1, -0.2, 0.05, 0.01, 0.3, -0.1, 0.02, 0.03]])
    assert torch.equal(vo.warp_vec2mtrx(p, "homography", 1), torch.eye(3).unsqueeze(0))
    A = torch.tensor([[0.05, -0.2, 0.1], [-0.1, -0.05 - 0.02, 0.3], [0.01, 0.03, 0.02]])
    two = vo.warp_vec2mtrx(p, "homography", 3)[0]                  # I + A + A^2/2
    assert torch.allclose(two, torch.eye(3) + A + A @ A / 2, atol=1e-6)
    import scipy.linalg
    full = vo.warp_vec2mtrx(p, "homography", 12)[0].double().numpy()
    assert np.abs(full - scipy.linalg.expm(A.double().numpy())).max() < 1e-6
    aff = vo.warp_vec2mtrx(torch.tensor([[0.1, 0.2, 0.3, 0.4, 0.5, 0.6]]), "affine", 2)[0]
    assert torch.allclose(aff, torch.tensor([[1.1, 0.2, 0.3], [0.4, 1.5, 0.6], [0, 0, 1.0]]))


def test_transform_image_identity_and_outside():
    # refMtrx mapping the canonical box onto the pixel box reproduces the image; a far shift gives zeros
    im = torch.rand(1, 4, 6, 3, dtype=F64)
    M = torch.tensor([[[2.5, 0, 2.5], [0, 1.5, 1.5], [0, 0, 1.0]]])     # x: [-1,1] -> [0,5], y: -> [0,3]
    out = vo.warp_transform_image(im, M, 4, 6, F64)
    assert torch.allclose(out, im, atol=1e-5)
    Mfar = M.clone(); Mfar[0, 0, 2] = 100.0
    assert vo.warp_transform_image(im, Mfar, 4, 6, F64).abs().max() == 0
    Mhalf = M.clone(); Mhalf[0, 0, 2] = 2.0                          # half a pixel to the left
    o = vo.warp_transform_image(torch.ones(1, 4, 6, 1, dtype=F64), Mhalf, 4, 6, F64)[0, :, :, 0]
    assert torch.allclose(o[:, 0], torch.full((4,), 0.5, dtype=F64), atol=1e-5) and torch.allclose(o[:, 1:], torch.ones(4, 5, dtype=F64), atol=1e-5)


def test_medfilt_is_pinned_by_scipy():
    # third-party anchor that IS installed here: the restated order filter must equal scipy.signal.medfilt bit for bit,
    # and the reference's own call (scalar 5 on an [h,w,2] field, main_flownetS_pyramid.py:809) is identically zero
    # because 75 of the 125 taps are channel-axis padding
    import warnings
    import scipy.signal
    rng = np.random.default_rng(3)
    f = rng.normal(size=(2, 13, 17, 2)).astype(np.float32) * 4
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for ks in (5, (5, 5, 1), (3, 7, 3), (1, 1, 1), (1, 9, 1)):
            got = vo.medfilt_flow(f, ks).numpy().astype(np.float32)
            ref = np.stack([scipy.signal.medfilt(f[b], ks if isinstance(ks, int) else list(ks)) for b in range(2)])
            assert np.array_equal(got, ref), ks
    assert np.abs(vo.medfilt_flow(f, 5).numpy()).max() == 0


def test_training_loss_known_answers():
    # zero flow: tf_warp's clipped corners give weight 0 on the last row and column (x1 == x0 there), so the mask is
    # 1 on the (h-1)x(w-1) interior and 0 on the border; masked_MSE = sum_interior (U-G)^2 / (3*interior + 3*border*1e-8)
    torch.manual_seed(1)
    B, h, w = 2, 5, 7
    G, U = torch.rand(B, h, w, 3, dtype=F64), torch.rand(B, h, w, 3, dtype=F64)
    l, warped = vo.lossterm(torch.zeros(B, h, w, 2), G, U)
    num = ((U - G)[:, :-1, :-1] ** 2).sum(dim=(1, 2, 3))
    den = 3 * (h - 1) * (w - 1) + 3 * (h + w - 1) * 1e-8
    assert abs(float(l) - float((num / den).mean())) < 1e-12
    assert torch.equal(warped[:, :-1, :-1], U[:, :-1, :-1])
    # total variation of a ramp: |d/dy| = 2 on (h-1)*w edges, |d/dx| = 3 on h*(w-1) edges, second channel constant
    f = torch.zeros(1, h, w, 2)
    f[0, :, :, 0] = 2 * torch.arange(h).view(h, 1) + 3 * torch.arange(w).view(1, w)
    assert float(vo.total_variation(f)) == 2 * (h - 1) * w + 3 * h * (w - 1)


def test_training_loss_gradient_matches_finite_differences():
    # autograd through the restated graph (what TF's autodiff computes: no gradient through the integer casts) against
    # central differences at points away from the truncation discontinuities
    torch.manual_seed(2)
    B, h, w = 1, 6, 8
    G, U = torch.rand(B, h, w, 3, dtype=F64), torch.rand(B, h, w, 3, dtype=F64)
    f0 = (torch.rand(B, h, w, 2) * 0.6 + 0.2)                     # sample positions strictly inside a cell
    f = f0.clone().requires_grad_(True)
    l, _ = vo.lossterm(f, G, U)
    l.backward()
    eps = 1e-3
    for (y, x, c) in [(2, 3, 0), (2, 3, 1), (0, 0, 0), (4, 6, 1)]:
        fp, fm = f0.clone(), f0.clone()
        fp[0, y, x, c] += eps
        fm[0, y, x, c] -= eps
        num = (float(vo.lossterm(fp, G, U)[0]) - float(vo.lossterm(fm, G, U)[0])) / (2 * eps)
        assert abs(num - float(f.grad[0, y, x, c])) < 2e-4 * max(1.0, abs(num)), (y, x, c, num, float(f.grad[0, y, x, c]))


def test_homography_oracle_known_answers():
    """Known-answer checks of the homography evaluator's restatement (main:728-736)."""
    H, W = 48, 64
    Ht = np.array([[1.03, 0.02, 1.5], [-0.01, 0.98, -2.25], [2e-5, -1e-5, 1.0]])
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    d = Ht[2, 0] * xs + Ht[2, 1] * ys + Ht[2, 2]
    flow = np.stack([xs - (Ht[0, 0] * xs + Ht[0, 1] * ys + Ht[0, 2]) / d, ys - (Ht[1, 0] * xs + Ht[1, 1] * ys + Ht[1, 2]) / d], -1)
    flow = flow.astype(np.float32)
    rng = np.random.default_rng(0)
    bad = rng.random((H, W)) < 0.35
    flow[bad] += np.where(rng.random((int(bad.sum()), 2)) < 0.5, -1, 1) * rng.uniform(8, 30, (int(bad.sum()), 2)).astype(np.float32)
    M, n = vo.homography_fit(flow, K=48, seed=2)
    assert n == int((~bad).sum())                                   # every clean pixel and no corrupted one
    assert np.abs(M - Ht).max() < 1e-4
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert np.array_equal(vo.cv_warp_perspective_u8(img, np.eye(3), H, W), img)
    sh = vo.cv_warp_perspective_u8(img, np.array([[1, 0, 2.0], [0, 1, 1.0], [0, 0, 1.0]]), H, W)
    assert np.array_equal(sh[1:, 2:], img[:-1, :-2]) and sh[0].max() == 0 and sh[:, :2].max() == 0     # constant-0 border
    half = vo.cv_warp_perspective_u8(img, np.array([[1, 0, 0.5], [0, 1, 0.0], [0, 0, 1.0]]), H, W)
    want = (img[:, :-1].astype(np.int64) + img[:, 1:] + 1) >> 1                                          # a = 16: equal weights
    assert np.array_equal(half[:, 1:], want.astype(np.uint8))


# ---- cv2 restatements (missing 3): cv2.resize(..., (512,384)) main:550,556-558 and cv2.warpPerspective main:736,740-741 are
# OpenCV calls; OpenCV is not installed here.  What pins the restatements: (a) known answers derived BY HAND from OpenCV's published 8-bit
# arithmetic (imgproc/resize.cpp: half-pixel centres, 11-bit coefficients, ((b0*(R0>>4))>>16 + (b1*(R1>>4))>>16 + 2)>>2; imgwarp.cpp:
# 1/32-pixel coordinates, 15-bit weights, (sum + 2^14)>>15) -- the literal arrays below, each derived in the comment next to it; (b) an
# INDEPENDENT float bilinear (scipy.ndimage.map_coordinates, order 1) at the same sample positions: within 1 LSB everywhere.
CV_KAT_RAMP_2X = [0, 3, 8, 13, 18, 23, 28, 30]
CV_KAT_THREE_QUARTERS = [17, 150, 67]
CV_KAT_CHECKER_2X = [[0, 64, 191, 255], [64, 96, 159, 191], [191, 159, 96, 64], [255, 191, 64, 0]]


def test_cv_resize_u8_hand_derived_known_answers():
    # 1x4 ramp -> 1x8 (scale 0.5): fx = 0.5 dx - 0.25 -> dx=0: sx=-1 -> clamped to (0, f=0): 0;  dx=1: (sx=0, f=.25) a = (1536, 512):
    # R = 10*512 = 5120, (2048*(5120>>4))>>16 = 10, (10+2)>>2 = 3;  dx=2: f=.75: R = 15360 -> 30 -> 8;  dx=3: (1, .25): 25600 -> 50 -> 13;
    # dx=4: 35840 -> 70 -> 18;  dx=5: 46080 -> 90 -> 23;  dx=6: 56320 -> 110 -> 28;  dx=7: sx=3 = sw-1 -> f=0: 30
    ramp = np.array([[0, 10, 20, 30]], dtype=np.uint8)[..., None]
    assert vo.cv_resize_u8(ramp, 1, 8)[0, :, 0].tolist() == CV_KAT_RAMP_2X
    assert vo.cv_resize_u8(ramp.transpose(1, 0, 2), 8, 1)[:, 0, 0].tolist() == CV_KAT_RAMP_2X        # the vertical pass alone: same values here
    # 4 -> 3 (scale = 1/(3/4) = 1.333...): dx=0: fx = float32(0.1666..) -> a = (rint(1706.67), rint(341.33)) = (1707, 341): R = 100*341 = 34100,
    # 34100>>4 = 2131, (2048*2131)>>16 = 66, (66+2)>>2 = 17;  dx=1: fx = 1.5 -> (1024, 1024): R = 307200 -> 19200 -> 600 -> 150;
    # dx=2: fx = 2.8333 -> (341, 1707): R = 200*341 + 40*1707 = 136480 -> 8530 -> 266 -> 67
    row = np.array([[0, 100, 200, 40]], dtype=np.uint8)[..., None]
    assert vo.cv_resize_u8(row, 1, 3)[0, :, 0].tolist() == CV_KAT_THREE_QUARTERS
    # 2x2 checker -> 4x4: both passes interpolate.  Row 0 after the horizontal pass, >>4: [0, 8160, 24480, 32640], row 1 mirrored.
    # dy=0 (b = 2048, 0): (2048*8160)>>16 = 255 -> 257>>2 = 64; 24480 -> 765 -> 191; 32640 -> 1020 -> 255.
    # dy=1 (b = 1536, 512): col 1: (1536*8160)>>16 = 191, (512*24480)>>16 = 191 -> 384>>2 = 96; col 2: 573 + 63 + 2 = 638>>2 = 159; ...
    chk = np.array([[0, 255], [255, 0]], dtype=np.uint8)[..., None]
    assert vo.cv_resize_u8(chk, 4, 4)[..., 0].tolist() == CV_KAT_CHECKER_2X
    # edge replication: every output left of the first / right of the last source centre repeats the edge pixel
    edge = np.array([[7, 200]], dtype=np.uint8)[..., None]
    out = vo.cv_resize_u8(edge, 1, 16)[0, :, 0]
    assert out[:4].tolist() == [7] * 4 and out[-4:].tolist() == [200] * 4 and np.all(np.diff(out.astype(int)) >= 0)
    # an exact 2x reduction (cv2 routes INTER_LINEAR to INTER_AREA there): both coefficients are 1024 and the formula collapses to (a+b+c+d+2)>>2
    img = np.random.default_rng(0).integers(0, 256, (8, 12, 3), dtype=np.uint8).astype(np.int64)
    area = (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(vo.cv_resize_u8(img.astype(np.uint8), 4, 6), area.astype(np.uint8))


def _float_bilinear_half_pixel(src, dh, dw):
    """Independent reference: float64 bilinear at half-pixel centres, edge-replicated (scipy.ndimage), NOT the fixed-point code."""
    from scipy import ndimage
    sh, sw = src.shape[:2]
    fy = np.clip((np.arange(dh) + 0.5) * (sh / dh) - 0.5, 0, sh - 1)
    fx = np.clip((np.arange(dw) + 0.5) * (sw / dw) - 0.5, 0, sw - 1)
    yy, xx = np.meshgrid(fy, fx, indexing="ij")
    return np.stack([ndimage.map_coordinates(src[..., c].astype(np.float64), [yy, xx], order=1, mode="nearest") for c in range(src.shape[2])], -1)


@pytest.mark.parametrize("sh,sw,dh,dw", [(720, 1280, 384, 512), (1080, 1920, 384, 512), (48, 64, 96, 128), (37, 53, 80, 100), (100, 75, 75, 100)])
def test_cv_resize_u8_against_an_independent_float_bilinear(sh, sw, dh, dw):
    rng = np.random.default_rng(sh + dw)
    yy, xx = np.mgrid[0:sh, 0:sw].astype(np.float64)
    smooth = np.stack([127 + 110 * np.sin(xx / (9.0 + c)) * np.cos(yy / (13.0 - c)) for c in range(3)], -1)
    for name, img in (("noise", rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)), ("smooth", np.clip(smooth, 0, 255).astype(np.uint8))):
        got = vo.cv_resize_u8(img, dh, dw).astype(np.int64)
        ref = _float_bilinear_half_pixel(img, dh, dw)
        # never a full grey level from the exact blend (measured 0.62-0.77: 11-bit coefficients, then R>>4, >>16 and >>2 all truncate)
        assert np.abs(got - ref).max() <= 0.8, (name, float(np.abs(got - ref).max()))
        exact = np.floor(ref + 0.5).astype(np.int64)
        d = np.abs(got - exact)
        assert d.max() <= 1, name
        # ... so it equals the ROUNDED float blend on 86-91 % of the pixels and sits one level below it on the rest: that downward bias is
        # the published formula's own (the hand-derived answers above show it: 7.5 -> 8 but 66.56 -> 66 -> 17 for a true 16.67), which is
        # why "exact on 99 %" is not a property any faithful restatement can have
        assert (d == 0).mean() >= 0.85, (name, float((d == 0).mean()))


def test_cv_resize_f32_against_an_independent_float_bilinear():
    rng = np.random.default_rng(3)
    img = rng.random((45, 60, 3), dtype=np.float32)
    for dh, dw in ((384 // 8, 512 // 8), (90, 120), (33, 47)):
        got = vo.cv_resize_f32(img, dh, dw).astype(np.float64)
        assert np.abs(got - _float_bilinear_half_pixel(img, dh, dw)).max() <= 2e-6


def test_cv_warp_perspective_u8_hand_derived_and_exact_blend():
    # a 0.26-pixel shift is rounded to 8/32: X = rint(32 (dx - 0.26)) = 32 dx - 8 -> sx = dx - 1, a = 24: weights (8, 24)/32 on (sx, sx+1):
    # dst = ((8 S[dx-1] + 24 S[dx]) * 1024 + 2^14) >> 15 = (8 S[dx-1] + 24 S[dx] + 16) >> 5; the tap left of the image reads 0
    row = np.array([[10, 100, 200, 41]], dtype=np.uint8)[..., None]
    M = np.array([[1, 0, 0.26], [0, 1, 0], [0, 0, 1.0]])
    assert vo.cv_warp_perspective_u8(row, M, 1, 4)[0, :, 0].tolist() == [(24 * 10 + 16) >> 5, (80 + 2400 + 16) >> 5, (800 + 4800 + 16) >> 5,
                                                                         (1600 + 984 + 16) >> 5] == [8, 78, 175, 81]
    # general homography: the published rule quantises the source position to 1/32 px; GIVEN those positions the 15-bit weights are exact
    # ((32-a)(32-b)*32 / 2^15), so the result must EQUAL floor(exact float blend + 0.5) -- computed here in float64 with zero outside
    rng = np.random.default_rng(9)
    H, W = 40, 56
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    M = np.array([[1.02, 0.03, -1.7], [-0.02, 0.97, 2.3], [3e-4, -2e-4, 1.0]])
    got = vo.cv_warp_perspective_u8(img, M, H, W)
    Mi = np.linalg.inv(M)
    dx, dy = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    den = Mi[2, 0] * dx + Mi[2, 1] * dy + Mi[2, 2]
    qx = np.rint(32.0 * (Mi[0, 0] * dx + Mi[0, 1] * dy + Mi[0, 2]) / den) / 32.0
    qy = np.rint(32.0 * (Mi[1, 0] * dx + Mi[1, 1] * dy + Mi[1, 2]) / den) / 32.0
    P = 8                                                            # zero margin: BORDER_CONSTANT 0
    pad = np.zeros((H + 2 * P, W + 2 * P, 3))
    pad[P:-P, P:-P] = img
    x0, y0 = np.floor(qx), np.floor(qy)
    ax, ay = (qx - x0)[..., None], (qy - y0)[..., None]
    assert x0.min() >= -P and x0.max() <= W + P - 2 and y0.min() >= -P and y0.max() <= H + P - 2
    xi, yi = x0.astype(int) + P, y0.astype(int) + P
    blend = (1 - ax) * (1 - ay) * pad[yi, xi] + ax * (1 - ay) * pad[yi, xi + 1] + (1 - ax) * ay * pad[yi + 1, xi] + ax * ay * pad[yi + 1, xi + 1]
    assert np.array_equal(got, np.floor(blend + 0.5).astype(np.uint8))
    # and against the UNquantised positions on a smooth image (gradient <= ~12 grey levels per pixel: 1/64 px of position is < 0.2 levels)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    smooth = np.clip(np.stack([127 + 100 * np.sin(xx / (9.0 + c)) * np.cos(yy / (11.0 - c)) for c in range(3)], -1), 0, 255).astype(np.uint8)
    from scipy import ndimage
    ex = (Mi[0, 0] * dx + Mi[0, 1] * dy + Mi[0, 2]) / den
    ey = (Mi[1, 0] * dx + Mi[1, 1] * dy + Mi[1, 2]) / den
    ref = np.stack([ndimage.map_coordinates(smooth[..., c].astype(np.float64), [ey, ex], order=1, mode="constant", cval=0.0) for c in range(3)], -1)
    interior = (ex >= 0) & (ex <= W - 1) & (ey >= 0) & (ey <= H - 1)
    g2 = vo.cv_warp_perspective_u8(smooth, M, H, W).astype(np.float64)
    assert np.abs(g2 - ref)[interior].max() <= 1.0
