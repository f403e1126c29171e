"""Known-answer tests that pin the oracle: every expectation here is
derivable by hand from the op definitions, none comes from running the oracle."""
import numpy as np
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
