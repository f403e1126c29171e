"""GPU parity of the training objective (first slice) against the oracle: loss value in fp64
arithmetic of the restated graph, gradient from torch autograd through it."""
import pytest
import torch

from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, training
from oracle import vstab_oracle as vo

pytestmark = pytest.mark.gpu


def _case(B, H, W, sizes, seed, mag=1.5):
    g = torch.Generator().manual_seed(seed)
    gt, un = torch.rand(B, H, W, 3, generator=g), torch.rand(B, H, W, 3, generator=g)
    flows = {n: torch.randn(B, h, w, 2, generator=g) * mag for n, (h, w) in zip(vo.LOSS_LEVELS, sizes)}
    return gt, un, flows


@pytest.mark.parametrize("B,H,W,sizes", [(2, 64, 96, [(1, 2), (2, 3), (4, 6), (8, 12), (62, 94)]),
                                         (1, 48, 64, [(1, 1), (2, 2), (3, 4), (6, 8), (46, 62)]),
                                         (3, 40, 40, [(1, 1), (2, 2), (3, 3), (5, 5), (38, 38)])])
def test_loss_main_value_and_gradients(B, H, W, sizes):
    gt, un, flows = _case(B, H, W, sizes, seed=B * 7 + H)
    loss, grads = training.loss_main({k: v.cuda() for k, v in flows.items()}, gt.cuda(), un.cuda())
    ref_in = {k: v.clone().requires_grad_(True) for k, v in flows.items()}
    ref = vo.loss_main(ref_in, gt, un)
    ref.backward()
    assert abs(float(loss) - float(ref.detach())) <= 2e-5 * max(1.0, abs(float(ref.detach())))
    for k in vo.LOSS_LEVELS:
        g, r = grads[k].double().cpu(), ref_in[k].grad.double()
        assert g.shape == r.shape
        tol = 2e-5 * max(float(r.abs().max()), 1e-6) + 1e-9
        assert float((g - r).abs().max()) <= tol, (k, float((g - r).abs().max()), float(r.abs().max()))


@pytest.mark.parametrize("B,H,W,sizes,C", [(2, 64, 96, [(1, 2), (2, 3), (4, 6), (8, 12), (62, 94)], 4),
                                           (1, 512, 512, [(8, 8), (16, 16), (32, 32), (64, 64), (510, 510)], 2)])
def test_loss_main_single_call_equals_per_level_calls(B, H, W, sizes, C):
    """`vstab_loss_main` (all levels, strided C-channel flow pixels, gradients written in place) against the per-level entry
    point that the oracle test above pins."""
    gt, un, flows = _case(B, H, W, sizes, seed=31)
    ref_loss, ref_grads = training.loss_main({k: v.cuda() for k, v in flows.items()}, gt.cuda(), un.cuda())
    wide = {k: torch.zeros(*v.shape[:3], C).cuda() for k, v in flows.items()}
    for k, v in flows.items():
        wide[k][..., :2] = v.cuda()
    grads = {k: torch.full_like(v, 7.0) for k, v in wide.items()}
    loss = training.loss_main_fused(wide, gt.cuda(), un.cuda(), grads)
    assert abs(float(loss) - float(ref_loss)) <= 1e-8 * max(1.0, abs(float(ref_loss)))      # the TV weights travel as float32
    for k in vo.LOSS_LEVELS:
        assert torch.equal(grads[k][..., :2], ref_grads[k]), k
        if C > 2:
            assert float((grads[k][..., 2:] - 7.0).abs().max()) == 0.0          # the other channels are not touched
    assert abs(float(training.loss_main_fused(wide, gt.cuda(), un.cuda())) - float(loss)) <= 1e-12      # value only
    with pytest.raises(ValueError):
        training.loss_main_fused({k: v[..., :3].contiguous() for k, v in wide.items()} if C > 2 else
                                 {k: torch.zeros(*v.shape[:3], 3).cuda() for k, v in wide.items()}, gt.cuda(), un.cuda())


def test_lossterm_zero_flow_and_far_flow():
    B, h, w = 2, 9, 11
    g = torch.Generator().manual_seed(5)
    G, U = torch.rand(B, h, w, 3, generator=g), torch.rand(B, h, w, 3, generator=g)
    l, grad = training.lossterm(torch.zeros(B, h, w, 2).cuda(), G.cuda(), U.cuda())
    num = ((U - G)[:, :-1, :-1].double() ** 2).sum(dim=(1, 2, 3))
    den = 3 * (h - 1) * (w - 1) + 3 * (h + w - 1) * 1e-8
    assert abs(float(l) - float((num / den).mean())) < 1e-6
    # a flow that throws every sample far outside: all four corners collapse, mask == 0 everywhere, loss == 0, no gradient
    far = torch.full((B, h, w, 2), 1e4).cuda()
    l2, g2 = training.lossterm(far, G.cuda(), U.cuda())
    assert float(l2) == 0.0 and float(g2.abs().max()) == 0.0
    # non-finite flows do not fault
    bad = torch.full((B, h, w, 2), float("nan")).cuda()
    training.lossterm(bad, G.cuda(), U.cuda())
    torch.cuda.synchronize()


def test_tv_only_gradient_is_sign_pattern():
    f = torch.zeros(1, 4, 5, 2)
    f[0, :, :, 0] = torch.arange(5).view(1, 5).float()            # ramp in x on channel 0
    G = U = torch.zeros(1, 4, 5, 3)
    l, g = training.lossterm(f.cuda(), G.cuda(), U.cuda(), tv_weight=0.5)
    assert abs(float(l) - 0.5 * 4 * 4) < 1e-9                     # 4 rows x 4 unit steps
    g = g.cpu()
    assert torch.all(g[0, :, 0, 0] == -0.5) and torch.all(g[0, :, -1, 0] == 0.5) and torch.all(g[0, :, 1:-1, 0] == 0)
    assert float(g[..., 1].abs().max()) == 0.0


def test_argument_checks():
    with pytest.raises(ValueError):
        training.lossterm(torch.zeros(1, 4, 4, 3).cuda(), torch.zeros(1, 4, 4, 3).cuda(), torch.zeros(1, 4, 4, 3).cuda())
    with pytest.raises(ValueError):
        training.lossterm(torch.zeros(2, 4, 4, 2).cuda(), torch.zeros(1, 4, 4, 3).cuda(), torch.zeros(1, 4, 4, 3).cuda())


# ----------------------------------------------------------------------------- conv weight gradient (MFMA)
def _torch_wgrad(x, g, k, s, p):
    """tf.gradients(conv2d(pad(x), W), W) with torch autograd in float64 (cross-correlation, HWIO)."""
    import torch.nn.functional as F
    cin, cout = x.shape[3], g.shape[3]
    W = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), W, stride=s, padding=p)
    y.backward(g.double().permute(0, 3, 1, 2))
    return W.grad.permute(2, 3, 1, 0).contiguous(), g.double().sum(dim=(0, 1, 2))      # HWIO


@pytest.mark.parametrize("B,Hi,Wi,cin,cout,k,s,p", [
    (2, 12, 16, 64, 128, 3, 1, 1),        # conv3_1-like
    (2, 13, 17, 128, 256, 5, 2, 2),       # conv3-like, odd sizes
    (1, 9, 11, 256, 512, 3, 2, 1),        # conv4-like
    (3, 6, 7, 8, 20, 3, 1, 1),            # tiny channel counts: one tile, most of it padding
    (2, 8, 8, 196, 64, 1, 1, 0),          # 1x1
    (1, 20, 24, 64, 64, 7, 2, 3),         # 7x7
    (8, 16, 16, 64, 128, 5, 2, 2),        # longer reduction -> split-K
    (2, 12, 16, 32, 132, 3, 1, 1),        # 128 + 4 output columns: the 128-wide part and the narrow tail as two launches
    (1, 40, 48, 16, 388, 1, 1, 0),        # 3 x 128 + 4 columns with a long reduction (both parts split-K)
])
def test_conv_wgrad_matches_autograd(B, Hi, Wi, cin, cout, k, s, p):
    g0 = torch.Generator().manual_seed(B * 100 + cin)
    Ho, Wo = (Hi + 2 * p - k) // s + 1, (Wi + 2 * p - k) // s + 1
    x = torch.randn(B, Hi, Wi, cin, generator=g0)
    g = torch.randn(B, Ho, Wo, cout, generator=g0)
    dW, db = training.conv_wgrad(x.cuda(), g.cuda(), k, s, p)
    rW, rb = _torch_wgrad(x, g, k, s, p)
    assert dW.shape == rW.shape
    scale = float(rW.abs().max())
    assert float((dW.double().cpu() - rW).abs().max()) <= 2e-5 * scale + 1e-6
    assert float((db.double().cpu() - rb).abs().max()) <= 2e-5 * float(rb.abs().max()) + 1e-6


def test_conv_wgrad_channel_slices_and_accumulate():
    # operands living inside wider (concat) pixels, and dW += on a second call
    g0 = torch.Generator().manual_seed(9)
    B, Hi, Wi, k, s, p = 2, 10, 12, 3, 1, 1
    xw = torch.randn(B, Hi, Wi, 20, generator=g0)          # use channels 4..12
    gw = torch.randn(B, Hi, Wi, 24, generator=g0)          # use channels 8..24
    dW, db = training.conv_wgrad(xw.cuda(), gw.cuda(), k, s, p, cx_off=4, cin=8, cg_off=8, cout=16)
    rW, rb = _torch_wgrad(xw[..., 4:12], gw[..., 8:24], k, s, p)
    assert float((dW.double().cpu() - rW).abs().max()) <= 2e-5 * float(rW.abs().max())
    dW2, db2 = training.conv_wgrad(xw.cuda(), gw.cuda(), k, s, p, cx_off=4, cin=8, cg_off=8, cout=16, dW=dW.clone(), db=db.clone(),
                                   accumulate=True)
    assert float((dW2.double().cpu() - 2 * rW).abs().max()) <= 4e-5 * float(rW.abs().max())
    assert float((db2.double().cpu() - 2 * rb).abs().max()) <= 4e-5 * float(rb.abs().max())
    with pytest.raises(ValueError):
        training.conv_wgrad(torch.zeros(1, 8, 8, 6).cuda(), torch.zeros(1, 8, 8, 8).cuda(), 3, 1, 1)     # cin % 4
    with pytest.raises(ValueError):
        training.conv_wgrad(torch.zeros(1, 8, 8, 8).cuda(), torch.zeros(1, 7, 8, 8).cuda(), 3, 1, 1)     # wrong output size


def test_deconv_filter_gradient_via_swapped_roles():
    # 4x4 stride-2 SAME transposed conv y = convT(x, W[4,4,cout,cin]): dW = wgrad of the conv that maps y-shaped tensors
    # to x-shaped ones, with (input, gout) = (dy, x)
    import torch.nn.functional as F
    g0 = torch.Generator().manual_seed(4)
    B, h, w, cin, cout = 2, 5, 6, 16, 8
    x = torch.randn(B, h, w, cin, generator=g0)
    dy = torch.randn(B, 2 * h, 2 * w, cout, generator=g0)
    Wt = torch.zeros(cin, cout, 4, 4, dtype=torch.float64, requires_grad=True)          # torch conv_transpose2d layout
    y = F.conv_transpose2d(x.double().permute(0, 3, 1, 2), Wt, stride=2, padding=1)
    y.backward(dy.double().permute(0, 3, 1, 2))
    ref = Wt.grad.permute(2, 3, 1, 0)                                                    # [4,4,cout,cin] = TL's deconv filter
    dW, _ = training.conv_wgrad(dy.cuda(), x.cuda(), 4, 2, 1, want_db=False)            # [4,4,cin_of_conv=cout, cout_of_conv=cin]
    assert dW.shape == ref.shape
    assert float((dW.double().cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


# ----------------------------------------------------------------------------- conv input gradient (forward MFMA kernel)
def _torch_dgrad(g, W_hwio, s, p, in_hw):
    import torch.nn.functional as F
    B = g.shape[0]
    cin = W_hwio.shape[2]
    x = torch.zeros(B, cin, in_hw[0], in_hw[1], dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, W_hwio.double().permute(3, 2, 0, 1), stride=s, padding=p)
    y.backward(g.double().permute(0, 3, 1, 2))
    return x.grad.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,Hi,Wi,cin,cout,k,s,p", [
    (2, 12, 16, 64, 128, 3, 1, 1),        # conv3_1-like
    (2, 13, 17, 128, 256, 5, 2, 2),       # conv3-like, odd input size
    (2, 14, 18, 64, 128, 5, 2, 2),        # even input size
    (1, 9, 11, 256, 512, 3, 2, 1),        # conv4-like, odd
    (2, 10, 12, 256, 512, 3, 2, 1),       # even
    (3, 6, 7, 8, 20, 3, 1, 1),            # small channel counts
    (2, 8, 8, 196, 32, 1, 1, 0),          # 1x1
    (1, 22, 26, 64, 64, 7, 2, 3),         # 7x7 stride 2
    (2, 16, 16, 512, 512, 3, 1, 1),       # split-K
])
def test_conv_dgrad_matches_autograd(B, Hi, Wi, cin, cout, k, s, p):
    g0 = torch.Generator().manual_seed(B * 10 + cout + k)
    Ho, Wo = (Hi + 2 * p - k) // s + 1, (Wi + 2 * p - k) // s + 1
    g = torch.randn(B, Ho, Wo, cout, generator=g0)
    W = torch.randn(k, k, cin, cout, generator=g0) / (k * k * cout) ** 0.5
    dx = training.conv_dgrad(g.cuda(), W.cuda(), s, p, (Hi, Wi))
    ref = _torch_dgrad(g, W, s, p, (Hi, Wi))
    assert dx.shape == ref.shape
    assert float((dx.double().cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6


def test_conv_dgrad_slices_accumulate_and_deconv_forward():
    import torch.nn.functional as F
    g0 = torch.Generator().manual_seed(3)
    B, Hi, Wi, cin, cout, k, s, p = 2, 10, 14, 16, 24, 3, 2, 1
    Ho, Wo = (Hi + 2 * p - k) // s + 1, (Wi + 2 * p - k) // s + 1
    gw = torch.randn(B, Ho, Wo, 40, generator=g0)                  # gradient lives in channels 8..32 of a wider pixel
    W = torch.randn(k, k, cin, cout, generator=g0) * 0.1
    dxw = torch.randn(B, Hi, Wi, 28, generator=g0)                 # result accumulates into channels 4..20
    ref = dxw.double().clone()
    ref[..., 4:20] += _torch_dgrad(gw[..., 8:32], W, s, p, (Hi, Wi))
    out = training.conv_dgrad(gw.cuda(), W.cuda(), s, p, (Hi, Wi), cg_off=8, dx=dxw.cuda(), cx_off=4, accumulate=True)
    assert float((out.double().cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # DeConv2dLayer forward (4x4 s2 SAME, filter [4,4,cout,cin]) through the same entry point
    x = torch.randn(2, 5, 6, 16, generator=g0)
    Wd = torch.randn(4, 4, 8, 16, generator=g0) * 0.1                # [4,4,out,in]
    y = training.conv_dgrad(x.cuda(), Wd.cuda(), 2, 1, (10, 12))
    yref = F.conv_transpose2d(x.double().permute(0, 3, 1, 2), Wd.double().permute(3, 2, 0, 1), stride=2, padding=1).permute(0, 2, 3, 1)
    assert float((y.double().cpu() - yref).abs().max()) <= 2e-5 * float(yref.abs().max())


# ----------------------------------------------------------------------------- BatchNorm (training mode) + leaky relu
def _bn_ref(z, beta, eps=1e-5):
    z = z.double().clone().requires_grad_(True)
    mean = z.mean(dim=(0, 1, 2))
    var = ((z - mean) ** 2).mean(dim=(0, 1, 2))
    u = (z - mean) / torch.sqrt(var + eps) + beta.double()
    y = torch.maximum(u, 0.1 * u)
    return z, y, mean, var


@pytest.mark.parametrize("B,H,W,cs,c_off,C", [(2, 9, 11, 64, 0, 64), (3, 16, 16, 40, 8, 24), (1, 33, 47, 128, 0, 128), (8, 32, 32, 68, 4, 64)])
def test_bn_lrelu_train_forward_backward(B, H, W, cs, c_off, C):
    g0 = torch.Generator().manual_seed(C + H)
    zfull = torch.randn(B, H, W, cs, generator=g0) * 2 + 0.5
    beta = torch.randn(C, generator=g0) * 0.3
    mm, mv = torch.randn(C, generator=g0), torch.rand(C, generator=g0) + 0.5
    dyfull = torch.randn(B, H, W, cs, generator=g0)
    zr, yr, mean, var = _bn_ref(zfull[..., c_off:c_off + C], beta)
    yr.backward(dyfull[..., c_off:c_off + C].double())
    zg, mmg, mvg = zfull.clone().cuda(), mm.clone().cuda(), mv.clone().cuda()
    y, smean, srstd = training.bn_lrelu_train_forward(zg, beta.cuda(), mmg, mvg, decay=0.9, c_off=c_off, C=C)
    yc = y.cpu()
    assert float((yc[..., c_off:c_off + C].double() - yr.detach()).abs().max()) <= 2e-5
    if c_off:
        assert torch.equal(yc[..., :c_off], zfull[..., :c_off])                      # channels outside the slice untouched
    assert float((smean.double().cpu() - mean.detach()).abs().max()) <= 1e-5
    assert float((srstd.double().cpu() - 1 / torch.sqrt(var.detach() + 1e-5)).abs().max()) <= 1e-4
    assert float((mmg.double().cpu() - (mm.double() * 0.9 + mean.detach() * 0.1)).abs().max()) <= 1e-5
    assert float((mvg.double().cpu() - (mv.double() * 0.9 + var.detach() * 0.1)).abs().max()) <= 1e-5
    dg = dyfull.clone().cuda()
    dz, dbeta = training.bn_lrelu_train_backward(y, dg, beta.cuda(), srstd, cy_off=c_off, cg_off=c_off, C=C)
    ref = zr.grad
    assert float((dz.cpu()[..., c_off:c_off + C].double() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max()))
    gl = dyfull[..., c_off:c_off + C].double() * torch.where(yr.detach() > 0, 1.0, 0.1)
    assert float((dbeta.double().cpu() - gl.sum(dim=(0, 1, 2))).abs().max()) <= 1e-4 * max(1.0, float(gl.sum(dim=(0, 1, 2)).abs().max()))


def test_lrelu_backward():
    y = torch.randn(2, 5, 6, 12)
    dy = torch.randn(2, 5, 6, 12)
    out = training.lrelu_backward(y.cuda(), dy.clone().cuda(), cy_off=4, cg_off=4, C=8).cpu()
    ref = dy.clone()
    ref[..., 4:] = dy[..., 4:] * torch.where(y[..., 4:] > 0, 1.0, 0.1)
    assert torch.allclose(out, ref, atol=0, rtol=1e-7)


# ----------------------------------------------------------------------------- resampler adjoints
@pytest.mark.parametrize("h,w,oh,ow,C", [(6, 8, 12, 16, 4), (12, 16, 24, 32, 2), (48, 64, 382, 510, 2), (5, 7, 9, 13, 4), (24, 32, 6, 8, 3),
                                         (7, 7, 7, 7, 4)])
def test_resize_bilinear_backward_is_the_adjoint(h, w, oh, ow, C):
    g0 = torch.Generator().manual_seed(h + oh)
    x = torch.randn(2, h, w, C, generator=g0, dtype=torch.float64, requires_grad=True)
    y = vo.resize_bilinear_legacy(x, oh, ow)
    dy = torch.randn(2, oh, ow, C, generator=g0)
    y.backward(dy.double())
    got = training.resize_bilinear_backward(dy.cuda(), (h, w), gain=1.0)
    assert float((got.double().cpu() - x.grad).abs().max()) <= 1e-5 * max(1.0, float(x.grad.abs().max()))
    acc = training.resize_bilinear_backward(dy.cuda(), (h, w), gain=2.0, din=got.clone())
    assert float((acc.double().cpu() - 3 * x.grad).abs().max()) <= 3e-5 * max(1.0, float(x.grad.abs().max()))


@pytest.mark.parametrize("h2,w2,H,W", [(12, 16, 48, 64), (96, 128, 384, 512), (13, 17, 50, 70), (3, 4, 9, 9), (24, 32, 96, 128)])
def test_pad_nearest_upsample_forward_and_adjoint(h2, w2, H, W):
    g0 = torch.Generator().manual_seed(h2)
    src = torch.randn(2, h2, w2, 8, generator=g0)
    iy = torch.from_numpy(vo.nearest_align_corners_index(h2 + 2, H))
    ix = torch.from_numpy(vo.nearest_align_corners_index(w2 + 2, W))
    p = torch.nn.functional.pad(src.double().requires_grad_(True), (0, 0, 1, 1, 1, 1))
    srcd = src.double().requires_grad_(True)
    ref = torch.nn.functional.pad(srcd, (0, 0, 1, 1, 1, 1))[:, iy][:, :, ix]
    out = training.pad_nearest_upsample(src.cuda(), H, W)
    assert torch.equal(out.cpu().double(), ref.detach())
    dy = torch.randn(2, H, W, 8, generator=g0)
    ref.backward(dy.double())
    got = training.pad_nearest_upsample_backward(dy.cuda(), (h2, w2))
    assert float((got.double().cpu() - srcd.grad).abs().max()) <= 1e-5 * max(1.0, float(srcd.grad.abs().max()))


# ----------------------------------------------------------------------------- Winograd F(2x2,3x3) with device-resident weights
# (the last two: thousands of tiles with odd sizes, so tiles hang over the bottom / right edge)
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 24, 64, 128), (1, 13, 17, 256, 256), (3, 8, 8, 128, 64), (2, 33, 20, 32, 192),
                                            (2, 127, 127, 64, 256), (2, 126, 129, 256, 64)])
def test_conv3x3_winograd_forward_and_input_gradient(B, H, W, cin, cout):
    import torch.nn.functional as F
    g0 = torch.Generator().manual_seed(cin + H)
    x = torch.randn(B, H, W, cin, generator=g0)
    Wt = torch.randn(3, 3, cin, cout, generator=g0) / (9 * cin) ** 0.5
    b = torch.randn(cout, generator=g0) * 0.1
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), Wt.double().permute(3, 2, 0, 1), b.double(), padding=1).permute(0, 2, 3, 1)
    y = training.conv3x3_winograd(x.cuda(), Wt.cuda(), bias=b.cuda())
    assert float((y.double().cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    yl = training.conv3x3_winograd(x.cuda(), Wt.cuda(), bias=b.cuda(), act=1)
    assert float((yl.double().cpu() - torch.maximum(ref, 0.1 * ref)).abs().max()) <= 2e-5 * float(ref.abs().max())
    if cin % 64 == 0 and cout % 32 == 0:                      # the input gradient swaps the channel roles
        g = torch.randn(B, H, W, cout, generator=g0)
        dref = _torch_dgrad(g, Wt, 1, 1, (H, W))
        dx = training.conv3x3_winograd(g.cuda(), Wt.cuda(), transpose=True)
        assert float((dx.double().cpu() - dref).abs().max()) <= 2e-5 * float(dref.abs().max())
        base = torch.randn(B, H, W, cin + 8, generator=g0)    # accumulate into a channel slice of a wider tensor
        out = training.conv3x3_winograd(g.cuda(), Wt.cuda(), transpose=True, y=base.clone().cuda(), cy_off=4, act=3).cpu()
        exp = base.double().clone()
        exp[..., 4:4 + cin] += dref
        assert float((out.double() - exp).abs().max()) <= 2e-5 * float(exp.abs().max())
        assert torch.equal(out[..., :4], base[..., :4])


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 24, 64, 128), (1, 13, 17, 256, 256), (3, 9, 8, 128, 64), (2, 33, 20, 32, 192), (4, 64, 64, 256, 256)])
def test_conv3x3_winograd_filter_gradient(B, H, W, cin, cout):
    """dW of a 3x3 stride-1 layer through the Winograd domain against autograd (and against the direct MFMA filter gradient)."""
    import torch.nn.functional as F
    g0 = torch.Generator().manual_seed(cout + W)
    x = torch.randn(B, H, W, cin + 4, generator=g0)                 # channel slices of wider tensors on both sides
    g = torch.randn(B, H, W, cout + 8, generator=g0)
    Wt = torch.zeros(3, 3, cin, cout, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x[..., 4:].double().permute(0, 3, 1, 2), Wt.permute(3, 2, 0, 1), padding=1)
    y.backward(g[..., :cout].double().permute(0, 3, 1, 2))
    ref = Wt.grad
    dW = training.conv3x3_winograd_wgrad(x.cuda(), g.cuda(), cx_off=4, cin=cin, cg_off=0, cout=cout)
    assert tuple(dW.shape) == (3, 3, cin, cout)
    assert float((dW.double().cpu() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    direct, _ = training.conv_wgrad(x.cuda(), g.cuda(), 3, 1, 1, cx_off=4, cin=cin, cg_off=0, cout=cout, want_db=False)
    assert float((dW - direct).abs().max()) <= 3e-5 * float(ref.abs().max())


def test_new_entry_points_reject_bad_arguments():
    """Every C entry point returns a negative code (raised here as ValueError / RuntimeError) instead of launching on bad input."""
    import ctypes as C
    from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, runtime
    L = _lib.lib()
    x = torch.zeros(1, 8, 8, 32, device="cuda")
    g = torch.zeros(1, 8, 8, 64, device="cuda")
    with pytest.raises(ValueError):                                        # gout of another spatial size
        training.conv3x3_winograd_wgrad(x, torch.zeros(1, 8, 7, 64, device="cuda"))
    with pytest.raises(ValueError):                                        # channel count not a multiple of 4
        training.conv3x3_winograd_wgrad(torch.zeros(1, 8, 8, 6, device="cuda"), g)
    assert L.vstab_conv3x3_winograd_wgrad_workspace_bytes(0, 8, 8, 32, 64) == 0
    dW = torch.empty(3, 3, 32, 64, device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    code = L.vstab_conv3x3_winograd_wgrad(x.data_ptr(), 1, 8, 8, 32, 0, 32, g.data_ptr(), 64, 0, 64, dW.data_ptr(), ws.data_ptr(), 16,
                                          runtime.stream_ptr())
    assert code < 0 and b"workspace" in L.vstab_last_error(None)          # workspace too small
    code = L.vstab_conv3x3_winograd_wgrad(None, 1, 8, 8, 32, 0, 32, g.data_ptr(), 64, 0, 64, dW.data_ptr(), ws.data_ptr(), ws.numel(),
                                          runtime.stream_ptr())
    assert code < 0
    # loss_main: odd channel stride, too many levels, NULL images
    desc = (_lib.LossLevelDesc * 1)()
    pf = torch.zeros(1, 4, 4, 3, device="cuda")
    desc[0].pf, desc[0].h, desc[0].w, desc[0].cs_pf, desc[0].tv_weight = pf.data_ptr(), 4, 4, 3, 0.0
    assert L.vstab_loss_main_workspace_bytes(C.addressof(desc), 1, 1) == 0
    desc[0].cs_pf = 2
    assert L.vstab_loss_main_workspace_bytes(C.addressof(desc), 9, 1) == 0
    n = L.vstab_loss_main_workspace_bytes(C.addressof(desc), 1, 1)
    assert n > 0
    out = torch.zeros((), dtype=torch.float64, device="cuda")
    assert L.vstab_loss_main(C.addressof(desc), 1, None, None, 1, 4, 4, out.data_ptr(), ws.data_ptr(), n, runtime.stream_ptr()) < 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,H,W,act", [(2, 64, 96, 0), (4, 256, 336, 1), (8, 128, 512, 0), (1, 70, 90, 2)])
def test_first_layer_on_the_row_window_kernel_with_device_weights(B, H, W, act):
    """vstab_conv_rowwin_forward: model.py:807-808 (pad 3, 7x7 stride 2, 27 -> 64) on the inference path's first-layer kernel with the raw
    filter in device memory, padded to 28 input channels as the training buffers hold it (the 28th is never read: it is poisoned here).
    64- and 128-pixel tiles, the 128 k + (1..64) column split (W = 336 -> 168 output columns), the stream form; none / leaky / relu."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * 100 + W)
    x = torch.rand(B, H, W, 27, generator=g)
    Wf = torch.randn(7, 7, 28, 64, generator=g) / (49 * 27) ** 0.5
    Wf[:, :, 27, :] = float("nan")
    b = torch.randn(64, generator=g) * 0.1
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    L = _lib.lib()
    n = L.vstab_conv_rowwin_forward_workspace_bytes(B, H, W, 27, 28, 64, 7, 2, 3, 64, 0, act)
    if (H, W) == (70, 90):
        # 90 * 27 floats per row is not a multiple of 4: the kernel does not take it and says so
        assert n == 0
        return
    assert n > 0
    ws = torch.empty(int(n) + 256, dtype=torch.uint8, device="cuda")
    y = torch.full((B, Ho, Wo, 64), float("nan"), dtype=torch.float32, device="cuda")
    xd, Wd, bd = x.cuda(), Wf.cuda(), b.cuda()
    _lib.check(L.vstab_conv_rowwin_forward(xd.data_ptr(), B, H, W, 27, Wd.data_ptr(), 28, 64, bd.data_ptr(), 7, 2, 3, y.data_ptr(), 64, 0, act,
                                           ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), Wf[:, :, :27].double().permute(3, 2, 0, 1), b.double(), stride=2, padding=3).permute(0, 2, 3, 1)
    if act == 1:
        ref = torch.maximum(ref, 0.1 * ref)
    elif act == 2:
        ref = torch.relu(ref)
    assert float((y.double().cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
