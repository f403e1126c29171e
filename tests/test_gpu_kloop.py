"""The assembly K loops (csrc/conv_kloop_gfx950.inc) through the C ABI's generic convolution: reductions of one, two, three ... K-tiles
take the loop's tail-only / body0 / body1 exits for both tile shapes it serves (128x128: cout >= 128, 128x64: cout = 64), rows past
the end of the GEMM, padded segment tails and out-of-image taps ride on the EXEC-narrowed validity test.  Reference: torch conv2d in
fp64 on the CPU.  (The first layer's row-window loop is reached through the network only: tests/test_gpu_parity.py, test_gpu_network_property.py.)"""
import pytest
import torch
import torch.nn.functional as F

from coupe.optical_flow_based_deep_video_stabilization_amd import _lib

pytestmark = pytest.mark.gpu


def _conv_forward(x, W, b, k, s, p, act=0):
    B, Hi, Wi, cin = x.shape
    cout = W.shape[3]
    Ho, Wo = (Hi + 2 * p - k) // s + 1, (Wi + 2 * p - k) // s + 1
    y = torch.empty(B, Ho, Wo, cout, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    n = L.vstab_conv_forward_workspace_bytes(B, Hi, Wi, cin, cin, k, s, p, cout, cout, 0, act, Ho, Wo)
    assert n > 0
    ws = torch.empty(int(n) + 256, dtype=torch.uint8, device=x.device)
    _lib.check(L.vstab_conv_forward(x.data_ptr(), B, Hi, Wi, cin, 0, cin, W.data_ptr(), b.data_ptr(), k, s, p, y.data_ptr(), Ho, Wo, cout, 0, cout, act,
                                    ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    return y


def _ref(x, W, b, k, s, p):
    y = F.conv2d(x.double().permute(0, 3, 1, 2), W.double().permute(3, 2, 0, 1), b.double(), stride=s, padding=p)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("cout", [64, 128, 256])
@pytest.mark.parametrize("B,H,W,cin,k,s,p", [
    (2, 16, 16, 32, 1, 1, 0),      # one K-tile: the loop is its tail only
    (2, 16, 16, 64, 1, 1, 0),      # two: body 0, tail 1
    (1, 20, 24, 96, 1, 1, 0),      # three: body 0, body 1, tail 0
    (3, 9, 11, 128, 1, 1, 0),      # four, rows past the end of the GEMM (297 = 2 x 128 + 41)
    (2, 13, 17, 36, 3, 1, 1),      # 3x3, padded segment tail (108 of 128 floats), out-of-image taps, odd sizes
    (1, 23, 29, 64, 5, 2, 2),      # 5x5 stride 2 (conv2 / conv3's form)
    (2, 12, 12, 260, 3, 2, 1),     # segment of 780 floats: 25 K-tiles per filter row, the last one 12 floats
])
def test_generic_conv_through_the_assembly_k_loop(B, H, W, cin, k, s, p, cout):
    g = torch.Generator().manual_seed(B * 1000 + cin + cout + k)
    x = torch.randn(B, H, W, cin, generator=g)
    Wt = torch.randn(k, k, cin, cout, generator=g) / (k * k * cin) ** 0.5
    b = torch.randn(cout, generator=g)
    y = _conv_forward(x.cuda(), Wt.cuda(), b.cuda(), k, s, p)
    ref = _ref(x, Wt, b, k, s, p)
    assert y.shape == ref.shape
    assert float((y.double().cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("cout", [132, 388, 160])
@pytest.mark.parametrize("B,H,W,cin,k,s,p", [(2, 16, 20, 64, 4, 2, 1), (1, 12, 12, 36, 3, 1, 1), (3, 8, 8, 32, 1, 1, 0)])
def test_conv_forward_with_a_few_columns_beyond_whole_tiles(B, H, W, cin, k, s, p, cout):
    """128 q + r output channels (the decoder's concat widths 388 / 772 / 1028 as the N of its input gradients: a last column tile with
    4 live columns), with a bias and with act = 3 (accumulate into y).  Round 5 ran the tail as its own 32-column launch: correct, and
    slower for 388 and 772 (the tail re-reads the whole A operand for a quarter of the MFMA work: 0.260 -> 0.292 ms, 0.256 -> 0.297 ms;
    tools/conv_forward_column_tail_r05.diff); this test stays as the guard of those widths."""
    g = torch.Generator().manual_seed(cout * 7 + cin + k)
    x = torch.randn(B, H, W, cin, generator=g)
    Wt = torch.randn(k, k, cin, cout, generator=g) / (k * k * cin) ** 0.5
    b = torch.randn(cout, generator=g)
    y = _conv_forward(x.cuda(), Wt.cuda(), b.cuda(), k, s, p)
    ref = _ref(x, Wt, b, k, s, p)
    tol = 2e-5 * float(ref.abs().max()) + 1e-6
    assert y.shape == ref.shape and float((y.double().cpu() - ref).abs().max()) <= tol
    # accumulate: y0 + conv (no bias)
    Ho, Wo = ref.shape[1], ref.shape[2]
    y0 = torch.randn(B, Ho, Wo, cout, generator=g)
    L = _lib.lib()
    yd, xd, Wd = y0.cuda(), x.cuda(), Wt.cuda()
    n = L.vstab_conv_forward_workspace_bytes(B, H, W, cin, cin, k, s, p, cout, cout, 0, 3, Ho, Wo)
    ws = torch.empty(int(n) + 256, dtype=torch.uint8, device="cuda")
    _lib.check(L.vstab_conv_forward(xd.data_ptr(), B, H, W, cin, 0, cin, Wd.data_ptr(), None, k, s, p, yd.data_ptr(), Ho, Wo, cout, 0, cout, 3,
                                    ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    ref2 = y0.double() + _ref(x, Wt, torch.zeros(cout), k, s, p)
    assert float((yd.double().cpu() - ref2).abs().max()) <= 2e-5 * float(ref2.abs().max()) + 1e-6


# ----------------------------------------------------------------------------- predict_flow2's tap table (csrc/tap_panel.hip)
def _tap_table(W):
    """[200][32] device table of vstab_predict2_tap_table from the head's filter W [3,3,194,2]"""
    t = torch.zeros(200, 32, dtype=torch.float32)
    t[:194, :18] = W.reshape(9, 194, 2).permute(1, 0, 2).reshape(194, 18)
    return t


@pytest.mark.parametrize("M", [32, 33, 1000, 8 * 128 * 128])
def test_predict2_tap_table_vs_fp64_matmul(M):
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, 196, generator=g)
    W = torch.randn(3, 3, 194, 2, generator=g) / 194 ** 0.5
    tab = _tap_table(W)
    xd, td = x.cuda(), tab.cuda()
    T = torch.full((M, 32), 7.0, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().vstab_predict2_tap_table(xd.data_ptr(), M, td.data_ptr(), T.data_ptr(), None))
    torch.cuda.synchronize()
    ref = x.double() @ tab.double()[:196]                             # rows 196..199 of the table pair with the NEXT pixel's floats: zero
    assert float((T.double().cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert float(T[:, 18:].abs().max()) == 0.0                       # the padding columns of the table rows are written as zeros


def test_predict2_tap_table_keeps_pixels_apart():
    """a workgroup multiplies straight out of the contiguous pixel rows; the last 8-float chunk of a row reaches into the NEXT pixel's
    first four floats, which must not leak into this pixel's sums -- not even as NaN * 0; nor may concat2's two padding channels"""
    g = torch.Generator().manual_seed(5)
    M = 96
    x = torch.randn(M, 196, generator=g)
    W = torch.randn(3, 3, 194, 2, generator=g) / 194 ** 0.5
    tab = _tap_table(W).cuda()
    T0 = torch.empty(M, 32, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().vstab_predict2_tap_table(x.cuda().data_ptr(), M, tab.data_ptr(), T0.data_ptr(), None))
    x2 = x.clone()
    x2[40, 0:4] = float("nan")                                       # pixel 40's first floats: only row 40 of T may change
    xd = x2.cuda()
    T1 = torch.empty(M, 32, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().vstab_predict2_tap_table(xd.data_ptr(), M, tab.data_ptr(), T1.data_ptr(), None))
    torch.cuda.synchronize()
    keep = torch.ones(M, dtype=torch.bool); keep[40] = False
    assert torch.equal(T0.cpu()[keep], T1.cpu()[keep])
    assert bool(torch.isnan(T1[40, :18]).all())


def test_predict2_tap_table_rejects_bad_arguments():
    L = _lib.lib()
    x = torch.zeros(64, 196, device="cuda"); tab = torch.zeros(200, 32, device="cuda"); T = torch.zeros(64, 32, device="cuda")
    assert L.vstab_predict2_tap_table(None, 64, tab.data_ptr(), T.data_ptr(), None) != 0
    assert L.vstab_predict2_tap_table(x.data_ptr(), 0, tab.data_ptr(), T.data_ptr(), None) != 0
    assert L.vstab_predict2_tap_table(x.data_ptr(), 3_000_000, tab.data_ptr(), T.data_ptr(), None) != 0       # 2.35e9 bytes of rows
    assert L.vstab_predict2_tap_table(x.data_ptr() + 4, 32, tab.data_ptr(), T.data_ptr(), None) != 0


# ----------------------------------------------------------------------------- the first layer's stream form (conv_rowwin.hip, stream_rows > 0)
@pytest.mark.parametrize("B,H,W", [(4, 512, 512), (2, 512, 1024), (16, 128, 512), (16, 512, 512)])
def test_first_layer_stream_form_vs_oracle(B, H, W):
    """launches whose 128-pixel tiles are a multiple of 512 run the first layer as streams of consecutive output rows per
    workgroup (4 rows each; tiles stored from registers under the next tile's first filter row, the last one by the ordinary epilogue): conv1 against
    the oracle's, top / bottom rows (filter rows outside the image) and every seam between two tiles of a stream included"""
    import numpy as np
    from coupe.optical_flow_based_deep_video_stabilization_amd import runtime, weights as wts
    import coupe.optical_flow_based_deep_video_stabilization_amd as vs
    from oracle import vstab_oracle as vo
    w = wts.synthetic_weights(seed=11, cin=27, random_bn=True, flow_gain=1.0)
    feats = np.random.default_rng(B + W).random((B, H, W, 27), dtype=np.float32)
    runtime.reset()
    vs.assign_weights(w)
    vs.flownetS_pyramid(torch.from_numpy(feats).cuda(), B, is_train=False)
    torch.cuda.synchronize()
    ints = runtime.get_context().internals(B, H, W, 27)
    _, ref_int = vo.flownetS_pyramid(feats, w, torch.float32, return_internals=True)
    # conv3 / conv4 are the outputs of the Winograd stages conv3_1 / conv4_1, whose 16 GEMMs run as streams of positions at these
    # shapes too (wino_gemm_stream.hip: 4 and 2 positions per workgroup at B=4 512x512, 8 and 4 at B=8)
    for name in ("conv1", "conv3", "conv4"):
        mine = ints[name].double().cpu()
        ref = torch.as_tensor(np.asarray(ref_int[name])).double()
        assert mine.shape == ref.shape
        assert float((mine - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max())), name
