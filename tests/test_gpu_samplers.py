"""GPU parity of the secondary samplers (spatial_transformer.py and warp.py
drop-ins) against the CPU oracle."""
import types

import numpy as np
import pytest
import torch

import coupe.optical_flow_based_deep_video_stabilization_amd as vs
from coupe.optical_flow_based_deep_video_stabilization_amd import spatial_transformer as st, warp as vwarp
from oracle import vstab_oracle as vo

pytestmark = pytest.mark.gpu


def maxabs(a, b):
    return float((a.double().cpu() - torch.as_tensor(np.asarray(b)).double()).abs().max())


def test_meshgrid_and_repeat():
    for size in ((2, 3), (5, 1), (37, 53), (1, 1)):
        g = st._meshgrid(size)
        assert np.array_equal(g.cpu().numpy(), vo.st_meshgrid(size))
    r = st._repeat(torch.arange(3), 2)
    assert r.tolist() == [0, 0, 1, 1, 2, 2]


@pytest.mark.parametrize("B,H,W,C,oh,ow", [(2, 9, 11, 3, 9, 11), (1, 32, 48, 1, 20, 30), (3, 16, 16, 4, 33, 17)])
def test_affine_and_projective_transformer(B, H, W, C, oh, ow):
    g = torch.Generator().manual_seed(H * W)
    im = torch.rand(B, H, W, C, generator=g)
    ident = torch.tensor([1., 0, 0, 0, 1, 0])
    th6 = ident + 0.3 * (torch.rand(B, 6, generator=g) - 0.5)
    out = st.AffineTransformer((oh, ow)).transform(im.cuda(), th6.cuda())
    assert out.shape == (B, oh, ow, C)
    assert maxabs(out, vo.st_transform(im, th6, (oh, ow))) <= 2e-5
    assert maxabs(st.transformer(im.cuda(), th6.cuda(), (oh, ow)), out.cpu()) == 0
    th8 = torch.cat([th6, 0.2 * (torch.rand(B, 2, generator=g) - 0.5)], 1)
    outp = st.ProjectiveTransformer((oh, ow)).transform(im.cuda(), th8.cuda())
    assert maxabs(outp, vo.st_transform(im, th8, (oh, ow))) <= 2e-5
    # identity theta at the input size reproduces the image
    same = st.AffineTransformer((H, W)).transform(im.cuda(), ident.repeat(B, 1).cuda())
    assert maxabs(same, im) <= 1e-5


def test_bilinear_interp_explicit_coordinates_and_edges():
    im = torch.rand(2, 6, 8, 3)
    x = torch.tensor([-1.0, 1.0, 0.0, -1.5, 3.0, 0.37, -1.0 - 2.0 / 7, float("nan")]).repeat(2)
    y = torch.tensor([-1.0, 1.0, 0.0, 0.2, -0.4, 1.4, 0.0, 0.0]).repeat(2)
    out = st.bilinear_interp(im.cuda(), x.cuda(), y.cuda(), (2, 4))
    ref = vo.st_bilinear_interp(im, torch.nan_to_num(x, nan=-5.0), y, (2, 4))     # NaN clips to the low edge like fmax
    assert out.shape == (16, 3)
    assert maxabs(out, ref) <= 1e-6
    assert maxabs(st._interpolate(im.cuda(), x.cuda(), y.cuda(), (2, 4), 'bilinear'), out.cpu()) == 0
    assert st._interpolate(im.cuda(), x.cuda(), y.cuda(), (2, 4), 'nearest') is None


def _cfg(**kw):
    return types.SimpleNamespace(**kw)


@pytest.mark.parametrize("warp_type,dim", [("homography", 8), ("affine", 6)])
def test_vec2mtrx(warp_type, dim):
    p = (torch.rand(5, dim) - 0.5) * 0.4
    cfg = _cfg(warpType=warp_type, warpApprox=20, batch_size=5)
    out = vwarp.vec2mtrx(cfg, p.cuda())
    assert out.shape == (5, 3, 3)
    assert maxabs(out, vo.warp_vec2mtrx(p, warp_type, 20)) <= 1e-5
    assert torch.equal(vwarp.compose(cfg, p, p), 2 * p) and torch.equal(vwarp.inverse(cfg, p), -p)


def test_transform_image_and_crop():
    B, H, W = 2, 24, 32
    g = torch.Generator().manual_seed(5)
    im = torch.rand(B, H, W, 3, generator=g)
    ref_m = vwarp.fit(np.array([[-1, -1], [1, -1], [-1, 1], [1, 1]], np.float64),
                      np.array([[0, 0], [W - 1, 0], [0, H - 1], [W - 1, H - 1]], np.float64))
    assert np.allclose(ref_m, [[(W - 1) / 2, 0, (W - 1) / 2], [0, (H - 1) / 2, (H - 1) / 2], [0, 0, 1]], atol=1e-5)
    cfg = _cfg(warpType="homography", warpApprox=20, batch_size=B, height=H, width=W, refMtrx=torch.from_numpy(ref_m))
    p = (torch.rand(B, 8, generator=g) - 0.5) * 0.2
    pM = vwarp.vec2mtrx(cfg, p.cuda())
    out = vwarp.transformImage(cfg, im.cuda(), pM)
    M = vo.warp_compose(ref_m, pM.cpu())
    assert float((M - torch.matmul(torch.from_numpy(ref_m).unsqueeze(0).expand(B, 3, 3), pM.cpu())).abs().max()) <= 1e-5      # a library GEMM's rounding
    assert maxabs(out, vo.warp_transform_image(im, M, H, W)) <= 2e-5
    # the composed matrix given directly (vstab_homography_warp) = the composition made inside the launch (vstab_transform_image)
    assert torch.equal(vwarp.warpImage(im.cuda(), M.cuda(), H, W), out)
    # identity parameters reproduce the image (pixel centres hit exactly up to rounding)
    ident = vwarp.transformImage(cfg, im.cuda(), torch.eye(3).repeat(B, 1, 1).cuda())
    assert maxabs(ident, im) <= 1e-4
    # crop variant: different source and output sizes
    cfg2 = _cfg(batch_size=B, height=10, W=14, dataH=H, dataW=W, refMtrx_b=torch.from_numpy(ref_m))
    outc = vwarp.transformCropImage(cfg2, im.cuda(), pM)
    assert outc.shape == (B, 10, 14, 3)
    assert maxabs(outc, vo.warp_transform_image(im, M, 10, 14)) <= 2e-5


def test_sampler_errors():
    with pytest.raises(ValueError):
        st.AffineTransformer((4, 4)).transform(torch.zeros(1, 4, 4, 3, device="cuda"), torch.zeros(1, 8, device="cuda"))
    with pytest.raises(ValueError):
        st.bilinear_interp(torch.zeros(1, 4, 4, 3, device="cuda"), torch.zeros(3, device="cuda"), torch.zeros(3, device="cuda"), (2, 2))
    with pytest.raises(AssertionError):
        vwarp.vec2mtrx(_cfg(warpType="similarity", warpApprox=3, batch_size=1), torch.zeros(1, 8))


# ---------------------------------------------------------------------------------------------------
# 3-channel frames run on the tiled kernel (st3_tile_kernel: 2-D tiles, 3-dword corner gathers, 16-byte row stores).
# The oracle evaluates theta . grid as the kernel does (matmul="unfused": every product and sum rounded to
# fp32), so the comparison is bit for bit; against torch.matmul's rounding sequence the source coordinate
# may differ by one ulp, which a noise image turns into <= ulp(W) * 1 of output error.
# ---------------------------------------------------------------------------------------------------
def _thetas(B, g, amp=0.15):
    ident = torch.tensor([1., 0, 0, 0, 1, 0])
    th6 = ident + amp * (torch.rand(B, 6, generator=g) - 0.5)
    th8 = torch.cat([th6, 0.1 * (torch.rand(B, 2, generator=g) - 0.5)], 1)
    return th6, th8


@pytest.mark.parametrize("B,H,W,oh,ow", [
    (2, 64, 96, 64, 96),        # whole tiles, staged 16-byte stores
    (2, 50, 70, 50, 70),        # ow % 4 != 0: 12-byte stores, ragged tiles
    (1, 48, 64, 37, 53),        # ragged output of an aligned source
    (2, 45, 66, 40, 64),        # odd source, staged stores
    (1, 240, 320, 32, 32),      # 7.5x minification
    (3, 16, 16, 80, 120),       # magnification
])
def test_tiled_sampler_three_channels_bit_exact(B, H, W, oh, ow):
    g = torch.Generator().manual_seed(H * W + oh)
    im = torch.rand(B, H, W, 3, generator=g)
    th6, th8 = _thetas(B, g)
    out6 = st.AffineTransformer((oh, ow)).transform(im.cuda(), th6.cuda())
    assert torch.equal(out6.cpu(), vo.st_transform(im, th6, (oh, ow), matmul="unfused"))
    assert maxabs(out6, vo.st_transform(im, th6, (oh, ow))) <= 1e-4
    out8 = st.ProjectiveTransformer((oh, ow)).transform(im.cuda(), th8.cuda())
    assert torch.equal(out8.cpu(), vo.st_transform(im, th8, (oh, ow), matmul="unfused"))
    # the same samples through explicit coordinates (bilinear_interp) and through the one-thread-per-pixel kernel (4 channels)
    grid = torch.from_numpy(vo.st_meshgrid((oh, ow))).reshape(3, -1)
    T = (th6.reshape(B, 2, 3)[:, :, 0:1] * grid[0] + th6.reshape(B, 2, 3)[:, :, 1:2] * grid[1]) + th6.reshape(B, 2, 3)[:, :, 2:3] * grid[2]
    outc = st.bilinear_interp(im.cuda(), T[:, 0].reshape(-1).cuda(), T[:, 1].reshape(-1).cuda(), (oh, ow))
    assert torch.equal(outc.reshape(B, oh, ow, 3), out6)
    im4 = torch.cat([im, im[..., :1]], 3)
    out4 = st.AffineTransformer((oh, ow)).transform(im4.cuda(), th6.cuda())
    assert torch.equal(out4[..., :3], out6) and torch.equal(out4[..., 3], out6[..., 0])


def test_tiled_sampler_edges_and_wild_maps():
    g = torch.Generator().manual_seed(11)
    B, H, W = 4, 40, 64
    im = torch.rand(B, H, W, 3, generator=g)
    th6 = torch.tensor([[1., 0, 0, 0, 1, 0],             # identity: reproduces the image
                        [1., 0, 1.5, 0, 1, -1.5],        # shifted out by 3/4 of the frame: mostly the zero border
                        [-1., 0, 0, 0, -1, 0],           # 180 degree turn
                        [3., 0, 0, 0, 3, 0]])            # 3x minification, everything beyond [-1,1] is border
    out = st.AffineTransformer((H, W)).transform(im.cuda(), th6.cuda())
    assert torch.equal(out.cpu(), vo.st_transform(im, th6, (H, W), matmul="unfused"))
    assert maxabs(out[0], im[0]) <= 1e-5
    # projective maps whose z changes sign inside the frame (samples scatter over the whole source), z == 0 rows
    th8 = torch.tensor([[1., 0, 0, 0, 1, 0, 2.0, 0.0], [1., 0, 0, 0, 1, 0, 0.0, -1.0], [0.5, 0.2, 0, -0.2, 0.5, 0, 1.0, 1.0],
                        [1., 0, 0, 0, 1, 0, 0.5, 0.5]])
    outp = st.ProjectiveTransformer((H, W)).transform(im.cuda(), th8.cuda())
    assert torch.equal(outp.cpu(), vo.st_transform(im, th8, (H, W), matmul="unfused"))
    # NaN / inf coordinates must not fault and clip like fmax/fmin (NaN -> low edge)
    x = torch.full((B * H * W,), float("nan")); y = torch.full((B * H * W,), float("inf"))
    o = st.bilinear_interp(im.cuda(), x.cuda(), y.cuda(), (H, W))
    assert torch.isfinite(o).all()


def test_transform_image_tiled_bit_exact():
    g = torch.Generator().manual_seed(3)
    for B, H, W, oh, ow in ((2, 64, 96, 64, 96), (2, 50, 70, 33, 45), (1, 200, 320, 24, 32)):
        im = torch.rand(B, H, W, 3, generator=g)
        ref_m = torch.tensor([[(W - 1) / 2, 0, (W - 1) / 2], [0, (H - 1) / 2, (H - 1) / 2], [0, 0, 1]])
        cfg = _cfg(warpType="homography", warpApprox=20, batch_size=B, height=oh, width=ow, refMtrx=ref_m)
        p = (torch.rand(B, 8, generator=g) - 0.5) * 0.2
        pM = vwarp.vec2mtrx(cfg, p.cuda())
        out = vwarp.transformImage(cfg, im.cuda(), pM)
        M = vo.warp_compose(ref_m, pM.cpu())          # refMtrx . pMtrx as the kernel composes it: no GEMM launch, every product and sum rounded
        assert torch.equal(out.cpu(), vo.warp_transform_image(im, M, oh, ow, matmul="unfused"))
        # integer sample points (floor == ceil): identity at the source size
        cfg_i = _cfg(warpType="homography", warpApprox=20, batch_size=B, height=H, width=W, refMtrx=ref_m)
        ident = vwarp.transformImage(cfg_i, im.cuda(), torch.eye(3).repeat(B, 1, 1).cuda())
        assert torch.equal(ident.cpu(), vo.warp_transform_image(im, ref_m.repeat(B, 1, 1), H, W, matmul="unfused"))


# ---- BASELINE configs[2]'s "spatial_transformer warp" at its stated size: batch 32, 720 x 1280 x 3.  The oracle needs seconds per
# sample there, so sample 0 is checked against it bit for bit and the rest through the copy pattern: samples are independent,
# so equal (image, theta) pairs must give equal bits wherever they sit in the batch.
def _cfg2_frames():
    g = torch.Generator().manual_seed(720)
    two = torch.rand(2, 720, 1280, 3, generator=g)
    pattern = torch.tensor([i % 2 for i in range(32)])
    pattern[-1] = 0
    return g, two, pattern


def test_cfg2_batch32_720p_spatial_transformer_warp():
    g, two, pattern = _cfg2_frames()
    im = two.cuda()[pattern.cuda()]
    th6_2, th8_2 = _thetas(2, g, amp=0.1)
    th6, th8 = th6_2[pattern], th8_2[pattern]
    for T, th, th2 in ((st.AffineTransformer, th6, th6_2), (st.ProjectiveTransformer, th8, th8_2)):
        out = T((720, 1280)).transform(im, th.cuda())
        torch.cuda.synchronize()
        assert out.shape == (32, 720, 1280, 3) and torch.isfinite(out).all()
        for i in range(32):
            assert torch.equal(out[i], out[int(pattern[i])]), i
        assert torch.equal(out[:1].cpu(), vo.st_transform(two[:1], th2[:1], (720, 1280), matmul="unfused"))
        assert maxabs(out[:1], vo.st_transform(two[:1], th2[:1], (720, 1280))) <= 5e-4      # one ulp of a coordinate <= 1280 on a noise image
        assert torch.equal(T((720, 1280)).transform(im, th.cuda()), out)                      # deterministic
        del out


def test_cfg2_batch32_720p_transform_image():
    g, two, pattern = _cfg2_frames()
    im = two.cuda()[pattern.cuda()]
    H, W = 720, 1280
    ref_m = torch.tensor([[(W - 1) / 2, 0, (W - 1) / 2], [0, (H - 1) / 2, (H - 1) / 2], [0, 0, 1]])
    cfg = _cfg(warpType="homography", warpApprox=20, batch_size=32, height=H, width=W, refMtrx=ref_m)
    p2 = (torch.rand(2, 8, generator=g) - 0.5) * 0.1
    pM = vwarp.vec2mtrx(cfg, p2[pattern].cuda())
    out = vwarp.transformImage(cfg, im, pM)
    torch.cuda.synchronize()
    assert out.shape == (32, H, W, 3) and torch.isfinite(out).all()
    for i in range(32):
        assert torch.equal(out[i], out[int(pattern[i])]), i
    M = vo.warp_compose(ref_m, pM[:1].cpu())
    assert torch.equal(out[:1].cpu(), vo.warp_transform_image(two[:1], M, H, W, matmul="unfused"))
