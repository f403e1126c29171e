"""The whole training step against the oracle: train-mode forward, the network's backward pass
(vector-Jacobian product against torch autograd through the fp64 restated graph), loss_main, BatchNorm moving averages
and one Adam update.

Two things make a naive "compare all gradients" test meaningless and are handled explicitly:
  * the leaky relu's gradient is discontinuous at 0: an element whose pre-activation is within rounding distance of zero
    lands on different sides in an fp32 and an fp64 forward and flips 0.9*dy.  The oracle is therefore run with the
    checked implementation's own activation pattern (`lrelu_masks`);
  * tf_warp's gradient is discontinuous across pixel-cell boundaries, so the network's backward is checked with FIXED
    upstream flow gradients (a linear functional of the flows); the loss gradient itself has its own test
    (test_gpu_training.py) on identical flows.
Sizes are chosen so that the coarsest BatchNorm still sees a few dozen rows (with 2 rows it is singular)."""
import numpy as np
import pytest
import torch

from coupe.optical_flow_based_deep_video_stabilization_amd import train_step, weights as wts
from oracle import vstab_oracle as vo

pytestmark = pytest.mark.gpu
BN_LAYERS = [e[0] for e in train_step.ENC] + ["deconv5", "deconv4", "deconv3", "deconv2"]


@pytest.mark.parametrize("winograd", [False, True])
def test_network_backward_matches_autograd(winograd):
    B, H, W = 2, 192, 256
    w = wts.synthetic_weights(seed=13, cin=27, random_bn=True, flow_gain=0.5)
    g0 = torch.Generator().manual_seed(H)
    feats = torch.rand(B, H, W, 27, generator=g0)
    tr = train_step.Trainer(w, B, H, W)
    tr.wino_min_flops = 0.0 if winograd else 1e30          # force / forbid the Winograd form of the 3x3 stride-1 stages
    flows = tr.forward(feats.cuda())

    # The activation pattern itself (the gradient check below feeds the implementation's own leaky-relu
    # sides to the oracle): against the fp64 oracle run on ITS OWN pattern, an element may sit on the other side of the kink
    # only if its value is within rounding distance of zero, and only a handful may at all.
    masks = tr.lrelu_masks()
    with torch.no_grad():
        _, own = vo.flownetS_pyramid(feats, w, torch.float64, return_internals=True, is_train=True)
    enc_names = [e[0] for e in train_step.ENC]
    for name in enc_names + ["deconv5", "deconv4", "deconv3", "deconv2"]:
        if name in enc_names:
            y = own[f"conv{name}"]
            if name in ("2", "3_1", "4_1", "5_1"):             # these are stored inside the concat buffers too; same tensor
                pass
            m = masks[name]
        else:
            cat = own["concat" + name[-1]]                         # [skip | deconvN | upsampled flow]
            skip = {"5": 512, "4": 512, "3": 256, "2": 128}[name[-1]]
            y = cat[..., skip:cat.shape[3] - 2]
            m = masks[name + "_bn"]
        flips = (y > 0) != m
        nflip = int(flips.sum())
        assert nflip <= max(4, int(2e-4 * y.numel())), (name, nflip, y.numel())
        if nflip:
            assert float(y[flips].abs().max()) <= 2e-4 * max(1.0, float(y.abs().max())), (name, float(y[flips].abs().max()))

    Wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=("moving_" not in k)) for k, v in w.items()}
    stats = {}
    out = vo.flownetS_pyramid(feats, Wt, is_train=True, batch_stats=stats, lrelu_masks=tr.lrelu_masks())
    for k in vo.LOSS_LEVELS:                                   # train-mode forward (batch statistics)
        assert float((flows[k].double().cpu() - out[k].detach()).abs().max()) <= 2e-3, k
    R = {k: torch.randn(out[k].shape, generator=g0) for k in vo.LOSS_LEVELS}
    sum((out[k] * R[k].double()).sum() for k in vo.LOSS_LEVELS).backward()
    ref = {k: v.grad for k, v in Wt.items() if v.grad is not None}

    tr.backward_from_flow_grads({k: v.cuda() for k, v in R.items()})
    got = tr.export(tr.g)
    assert set(got) == set(ref) and len(ref) == 60
    for k, r in ref.items():
        d = float((torch.from_numpy(got[k]).double() - r).abs().max())
        layer, leaf = k.split("/")
        if leaf in ("b_conv2d", "b_deconv2d") and layer in BN_LAYERS:
            # a bias in front of BatchNorm has an exactly-zero gradient (the mean subtraction removes it): the computed
            # value is rounding noise of the sum of dz, measured against the size of the layer's beta gradient
            beta = ref[(layer if leaf == "b_conv2d" else layer + "_bn") + "/beta"]
            assert float(r.abs().max()) < 1e-6 and d <= 1e-3 * float(beta.abs().max()), (k, d)
        else:
            assert d <= 1e-4 * float(r.abs().max()) + 1e-9, (k, d, float(r.abs().max()))

    # BatchNorm moving averages after this ONE forward (TensorLayer: decay 0.9)
    newp = tr.export()
    for name, (m, v) in stats.items():
        mm = w[f"{name}/moving_mean"].astype(np.float64) * 0.9 + m.numpy() * 0.1
        mv = w[f"{name}/moving_variance"].astype(np.float64) * 0.9 + v.numpy() * 0.1
        assert np.abs(newp[f"{name}/moving_mean"] - mm).max() <= 1e-4 * max(1.0, np.abs(mm).max()), name
        assert np.abs(newp[f"{name}/moving_variance"] - mv).max() <= 1e-3 * max(1.0, np.abs(mv).max()), name


def test_loss_adam_and_padding():
    B, H, W = 2, 96, 128
    w = wts.synthetic_weights(seed=12, cin=27, random_bn=True, flow_gain=0.5)
    g0 = torch.Generator().manual_seed(H)
    feats = torch.rand(B, H, W, 27, generator=g0)
    gt, un = torch.rand(B, H, W, 3, generator=g0), torch.rand(B, H, W, 3, generator=g0)
    tr = train_step.Trainer(w, B, H, W)
    flows = tr.forward(feats.cuda())
    # loss_main of the oracle on the SAME flows (the loss kernels' own parity test covers value and gradient)
    ref_loss = float(vo.loss_main({k: v.float().cpu() for k, v in flows.items()}, gt, un))
    loss = tr.loss_and_backward(gt.cuda(), un.cuda())
    assert abs(float(loss) - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss))
    got = tr.export(tr.g)
    assert all(np.isfinite(v).all() for v in got.values())
    assert max(float(np.abs(v).max()) for v in got.values()) > 0
    # one Adam update from those gradients (t = 1: lr_t = lr * sqrt(1 - b2) / (1 - b1))
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    before = tr.export()
    tr.adam(lr, b1, b2, eps)
    after = tr.export()
    for k, g in got.items():
        g64 = g.astype(np.float64)
        m, v = (1 - b1) * g64, (1 - b2) * g64 * g64
        exp = before[k].astype(np.float64) - lr * np.sqrt(1 - b2) / (1 - b1) * m / (np.sqrt(v) + eps)
        assert np.abs(after[k] - exp).max() <= 1e-6 + 1e-5 * np.abs(exp).max(), k
    for k in tr.p:                                             # moving statistics are not trained
        if "moving_" in k:
            assert k not in got
    # the zero padding of the stored parameters survives the update
    for k, t in tr.p.items():
        mask = torch.ones_like(t, dtype=torch.bool)
        mask[tuple(slice(0, s) for s in tr.shape_real[k])] = False
        if mask.any():
            assert float(t[mask].abs().max()) == 0.0, k


def test_loss_decreases_over_a_few_steps():
    B, H, W = 2, 96, 128
    w = wts.synthetic_weights(seed=5, cin=27, random_bn=False, flow_gain=0.2)
    g0 = torch.Generator().manual_seed(1)
    feats = torch.rand(B, H, W, 27, generator=g0).cuda()
    gt, un = torch.rand(B, H, W, 3, generator=g0).cuda(), torch.rand(B, H, W, 3, generator=g0).cuda()
    tr = train_step.Trainer(w, B, H, W)
    losses = [float(tr.step(feats, gt, un, lr=1e-4)) for _ in range(8)]
    assert all(np.isfinite(losses))
    assert min(losses[4:]) < losses[0], losses


def test_checkpoint_round_trip_and_schedule(tmp_path):
    from coupe.optical_flow_based_deep_video_stabilization_amd import model, runtime
    assert train_step.learning_rate(0) == 1e-4 and abs(train_step.learning_rate(45) - 1e-4 * 0.8 ** 2) < 1e-18
    B, H, W = 1, 96, 128
    w = wts.synthetic_weights(seed=8, cin=27, random_bn=True, flow_gain=0.3)
    g0 = torch.Generator().manual_seed(2)
    feats = torch.rand(B, H, W, 27, generator=g0).cuda()
    gt, un = torch.rand(B, H, W, 3, generator=g0).cuda(), torch.rand(B, H, W, 3, generator=g0).cuda()
    tr = train_step.Trainer(w, B, H, W)
    losses = tr.train_epoch([(feats, gt, un)] * 3, epoch=0)
    assert len(losses) == 3 and all(np.isfinite(losses))
    path = str(tmp_path / "ckpt.npz")
    tr.save_npz(path)
    back = wts.load_npz_dict(path)                                   # the reference's key names, validated shapes
    cur = tr.export()
    assert set(back) == set(cur) and all(np.array_equal(back[k], cur[k]) for k in cur)
    assert not np.array_equal(back["3_1/W_conv2d"], w["3_1/W_conv2d"])          # it trained
    assert not np.array_equal(back["3_1/moving_mean"], w["3_1/moving_mean"])    # and the moving statistics moved
    # the trained checkpoint drops into the inference path
    runtime.reset()
    model.load_and_assign_npz_dict(path)
    out = model.flownetS_pyramid(feats, B, is_train=False)
    ref = vo.flownetS_pyramid(feats.cpu(), back, dtype=torch.float64)
    assert float((out["predict_flow2"].double().cpu() - ref["predict_flow2"]).abs().max()) <= 1e-3


def test_flownetS_pyramid_is_train_true_and_sync_back():
    import coupe.optical_flow_based_deep_video_stabilization_amd as vs
    from coupe.optical_flow_based_deep_video_stabilization_amd import runtime
    B, H, W = 2, 96, 128
    runtime.reset()
    w = vs.initialize_global_variables(seed=4, cin=27, random_bn=True, flow_gain=0.3)
    g0 = torch.Generator().manual_seed(7)
    feats = torch.rand(B, H, W, 27, generator=g0).cuda()
    out = vs.flownetS_pyramid(feats, B, is_train=True)                      # main:184: the training graph's forward
    ref = vo.flownetS_pyramid(feats.cpu(), w, dtype=torch.float64, is_train=True)
    assert set(out) == {"predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3", "predict_flow2", "flow"}
    for k in vo.LOSS_LEVELS:
        assert out[k].shape == ref[k].shape and out[k].is_contiguous()
        assert float((out[k].double().cpu() - ref[k]).abs().max()) <= 2e-3
    tr = train_step.get_trainer("flownetS", B, H, W)
    assert tr is train_step.get_trainer("flownetS", B, H, W)              # one training graph per scope and shape
    gt, un = torch.rand(B, H, W, 3, generator=g0).cuda(), torch.rand(B, H, W, 3, generator=g0).cuda()
    tr.step(feats, gt, un, lr=1e-4)
    train_step.sync_to_inference(tr)                                        # the reuse=True inference graph sees the update
    inf = vs.flownetS_pyramid(feats, B, is_train=False)
    ref2 = vo.flownetS_pyramid(feats.cpu(), tr.export(), dtype=torch.float64)
    assert float((inf["predict_flow2"].double().cpu() - ref2["predict_flow2"]).abs().max()) <= 1e-3
    with pytest.raises(RuntimeError):
        train_step.get_trainer("no_such_scope", B, H, W)


def test_overlapped_gradient_exchange_on_one_rank():
    """step() under a (single-rank) RCCL process group with the overlapped exchange forced on: three all-reduces of bucket ranges
    are issued during the backward pass and waited for before Adam; averaging over one rank changes nothing, so the parameters
    after two steps must equal those of a Trainer stepping without a process group, bit for bit."""
    import os
    import torch.distributed as dist
    B, H, W = 1, 96, 128
    w = wts.synthetic_weights(seed=7, cin=27, random_bn=False, flow_gain=0.2)
    g0 = torch.Generator().manual_seed(2)
    feats = torch.rand(B, H, W, 27, generator=g0).cuda()
    gt, un = torch.rand(B, H, W, 3, generator=g0).cuda(), torch.rand(B, H, W, 3, generator=g0).cuda()
    ref = train_step.Trainer(w, B, H, W)
    ref_losses = [float(ref.step(feats, gt, un, lr=1e-4)) for _ in range(2)]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        tr = train_step.Trainer(w, B, H, W)
        tr.overlap_single_rank = True
        sent = []
        orig = tr.gbucket.allreduce_range_start
        tr.gbucket.allreduce_range_start = lambda lo, hi, group=None, single_rank_ok=False: (sent.append((lo, hi)), orig(lo, hi, group, single_rank_ok))[1]
        losses = [float(tr.step(feats, gt, un, lr=1e-4)) for _ in range(2)]
        torch.cuda.synchronize()
    finally:
        if own:
            dist.destroy_process_group()
    assert losses == ref_losses
    assert torch.equal(tr.pbucket.flat, ref.pbucket.flat)
    assert len(sent) == 6 and sent[0][0] == 0 and sent[2][1] == tr.gbucket.flat.numel()            # three ranges per step, gap-free
    assert sent[0][1] == sent[1][0] and sent[1][1] == sent[2][0]
    # the middle range (conv6_1 .. conv4) carries most of the 38.7 M parameters
    assert sent[1][1] - sent[1][0] > 0.5 * tr.gbucket.flat.numel()
