"""GPU parity of the flow post-filters against the oracle."""
import pytest
import torch

from coupe.optical_flow_based_deep_video_stabilization_amd import postfilters as pf
from oracle import vstab_oracle as vo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,h,w,k", [(1, 382, 510, 75), (2, 40, 50, 75), (1, 30, 30, 3), (1, 9, 200, 1)])
def test_box_blur(B, h, w, k):
    f = torch.randn(B, h, w, 2) * 10
    out = pf.box_blur_flow(f.cuda(), k)
    ref = vo.box_blur_flow(f, k)
    assert float((out.double().cpu() - ref).abs().max()) <= 2e-5
    ones = pf.box_blur_flow(torch.ones(1, h, w, 2, device="cuda"), k)
    if h > k and w > k:
        assert abs(float(ones[0, h // 2, w // 2, 0]) - 1.0) < 1e-5          # interior: full window
    assert float(ones[0, 0, 0, 0]) < 1.0 or k == 1                          # corner: zero padding darkens


def test_mean_flow_and_axpby_and_ema():
    f = torch.randn(3, 382, 510, 2) * 5 + 2
    m = pf.mean_flow(f.cuda())
    assert float((m.double().cpu() - vo.mean_flow(f)).abs().max()) <= 1e-4
    a = pf.axpby(0.9, f.cuda(), 0.1, (2 * f).cuda())
    assert float((a.cpu() - (0.9 * f + 0.1 * 2 * f)).abs().max()) <= 1e-5
    filt = pf.BlurEmaFilter(k=5)
    prev = torch.zeros_like(f)
    for _ in range(3):
        cur = torch.randn_like(f)
        got = filt(cur.cuda())
        ref = 0.9 * vo.box_blur_flow(cur, 5) + 0.1 * prev.double()
        prev = 0.9 * prev + 0.1 * cur
        assert float((got.double().cpu() - ref).abs().max()) <= 1e-5
    with pytest.raises(ValueError):
        pf.box_blur_flow(torch.zeros(1, 4, 4, 2, device="cuda"), 4)         # even k


@pytest.mark.parametrize("B,h,w,ks", [(1, 382, 510, 5), (2, 45, 37, (5, 5, 1)), (1, 20, 33, (3, 7, 3)), (1, 17, 16, (1, 1, 1)),
                                      (1, 3, 4, (5, 5, 1)), (1, 40, 40, (9, 3, 1))])
def test_medfilt_is_bit_exact(B, h, w, ks):
    torch.manual_seed(5)
    f = (torch.randn(B, h, w, 2) * 6).round() / 2            # many ties
    out = pf.medfilt_flow(f.cuda(), ks)
    ref = vo.medfilt_flow(f, ks).float()
    assert torch.equal(out.cpu(), ref)
    if ks == 5:
        assert float(out.abs().max()) == 0.0                  # the reference's 5x5x5 call: always zero


def test_median_ema_filter_and_argument_checks():
    filt = pf.MedianEmaFilter(kernel_size=(5, 5, 1))
    prev = torch.zeros(1, 30, 40, 2, dtype=torch.float64)
    for _ in range(3):
        cur = torch.randn(1, 30, 40, 2)
        got = filt(cur.cuda())
        med = vo.medfilt_flow(cur, (5, 5, 1))
        ref = (0.9 * med.float() + 0.1 * prev.float())
        prev = (0.9 * prev.float() + 0.1 * med.float()).double()
        assert float((got.cpu() - ref).abs().max()) <= 1e-6
    with pytest.raises(ValueError):
        pf.medfilt_flow(torch.zeros(1, 4, 4, 2, device="cuda"), 4)
    with pytest.raises(ValueError):
        pf.medfilt_flow(torch.zeros(1, 4, 4, 2, device="cuda"), (3, 3))
