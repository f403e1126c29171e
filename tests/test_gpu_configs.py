"""The two BASELINE.json configs that had no `-m gpu` test as stated:

* configs[4] -- batch=16 1080p with the VGG16 trunk in the same pass (network + glue/warp + vgg16.py:25-48 on the warped frames):
  copies of an input must come out bit-identical wherever they sit in the batch -- on both sides of the 2 x 8 chunk boundary
  of the network AND of the trunk's own chunks -- and one sample is checked against the fp32 oracle.
* configs[3] on one GPU -- shard_range x micro-batches x all-gather under a world-size-1 `nccl` (= RCCL) group must reproduce the
  unsharded sequence bit for bit (uint8 frames, as the reference's writer produces them, main:625-630), and the streaming
  `FrameGatherer` must hand back what was submitted.
"""
import os
import socket

import numpy as np
import pytest
import torch

import coupe.optical_flow_based_deep_video_stabilization_amd as vs
from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, distributed as vdist, runtime, vgg16 as vvgg, weights as wts
from oracle import vstab_oracle as vo

pytestmark = pytest.mark.gpu
FLOW_TOL = 1e-3


def test_cfg4_batch16_1080p_net_warp_vgg_one_pass():
    B, H, W, cin = 16, 1080, 1920, 27
    w = wts.synthetic_weights(seed=1, cin=cin, random_bn=False)
    runtime.reset()
    vs.assign_weights(w)
    L = _lib.lib()
    assert L.vstab_workspace_bytes(16, H, W, cin) < 2 * L.vstab_workspace_bytes(8, H, W, cin)   # the batch runs as two chunks (2 GiB tensor limit)
    rng = np.random.default_rng(4)
    two = rng.random((2, H, W, cin), dtype=np.float32)
    two_fr = rng.random((2, H, W, 3), dtype=np.float32)
    pattern = np.array([i % 2 for i in range(B)])
    pattern[-1] = 0                                     # samples 0 (first chunk) and 15 (second chunk) are the same input
    idx = torch.from_numpy(pattern).cuda()
    feats = torch.from_numpy(two).cuda()[idx]
    frame = torch.from_numpy(two_fr).cuda()[idx]
    dd = vvgg.synthetic_data_dict(seed=4)
    trunk = vvgg.Vgg16(data_dict=dd, reuse_outputs=True)
    flows, outflow, warped = vs.stabilise_originalsize(feats, frame)
    net = trunk.build(vvgg.preprocess(warped))
    torch.cuda.synchronize()
    assert flows["predict_flow2"].shape == (B, H - 2, W - 2, 2) and net.pool5.shape == (B, 34, 60, 512)
    for i in range(B):                                  # copy pattern: bit identity across every chunk boundary
        j = int(pattern[i])
        for k in vo.FLOW_KEYS:
            assert torch.equal(flows[k][i], flows[k][j]), (k, i)
        assert torch.equal(outflow[i], outflow[j]) and torch.equal(warped[i], warped[j]), i
        for name in ("conv1_2", "pool2", "conv3_3", "conv4_3", "conv5_3", "pool5"):
            assert torch.equal(getattr(net, name)[i], getattr(net, name)[j]), (name, i)
    assert torch.isfinite(net.pool5).all()
    # one sample against the fp32 oracle: the network, then glue + warp and the trunk each on the GPU's own input (isolates the rows)
    ref = vo.flownetS_pyramid(two[:1], w, torch.float32)
    errs = {k: float((flows[k][0].double().cpu() - ref[k][0].double()).abs().max()) for k in vo.FLOW_KEYS}
    assert all(e <= FLOW_TOL for e in errs.values()), errs
    of_ref = vo.flow_to_output_res(flows["predict_flow2"][:1].cpu(), H, W, H, W)
    assert float((outflow[:1].cpu() - of_ref).abs().max()) <= 2e-5
    assert torch.equal(warped[:1].cpu(), vo.tf_warp(frame[:1].cpu(), outflow[:1].cpu(), H, W, torch.float32))
    vref = vo.vgg16_build(vo.vgg_preprocess(warped[:1].cpu(), torch.float32), dd, torch.float32)
    for name in vvgg.OUTPUTS:
        got, r = getattr(net, name)[:1].cpu(), vref[name]
        err = float((got - r).abs().max())
        assert err <= 3e-4 * max(1.0, float(r.abs().max())), (name, err, float(r.abs().max()))


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _stabilise_u8(feats, frame, micro):
    """The per-rank loop of bench_clip.py: micro-batches through the HIP path, uint8 frames as the reference writes them."""
    n, H, W = feats.shape[0], frame.shape[1], frame.shape[2]
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    for b0 in range(0, n, micro):
        bc = min(micro, n - b0)
        _, _, warped = vs.stabilise_originalsize(feats[b0:b0 + bc].contiguous(), frame[b0:b0 + bc].contiguous())
        _lib.check(_lib.lib().vstab_quantise_output(warped.data_ptr(), bc * H * W, out[b0:b0 + bc].data_ptr(), runtime.stream_ptr()))
    return out


def test_cfg3_sharded_clip_gather_world1_nccl(nccl_world1):
    dist = nccl_world1
    F_, H, W, cin, MB = 10, 1080, 1920, 27, 4           # ragged: micro-batches of 4, 4, 2
    runtime.reset()
    vs.initialize_global_variables(seed=1, cin=cin)
    runtime.get_context().set_plan_batch(MB)             # as bench_clip.py does
    g = torch.Generator(device="cuda").manual_seed(7)
    feats = torch.rand(F_, H, W, cin, generator=g, device="cuda")
    frame = torch.rand(F_, H, W, 3, generator=g, device="cuda")
    whole = _stabilise_u8(feats, frame, MB)              # the unsharded sequence (one rank, same micro-batching)
    lo, hi = vdist.shard_range(F_, dist.get_rank(), dist.get_world_size())
    assert (lo, hi) == (0, F_)
    shard = _stabilise_u8(feats[lo:hi], frame[lo:hi], MB)
    full = vdist.gather_sequence(shard, F_)              # all_gather_into_tensor over RCCL
    torch.cuda.synchronize()
    assert full.dtype == torch.uint8 and tuple(full.shape) == (F_, H, W, 3)
    assert torch.equal(full, whole)
    # what two and three ranks would each compute -- their shard_range blocks of the clip: 5 + 5 frames (micro-batches 4 + 1) and
    # 4 + 3 + 3, none of them batched as in the unsharded run (4 + 4 + 2) -- reassembled in rank order is bit for bit the same
    # sequence, uint8 frames and fp32 flows alike: the launch plan is pinned to the micro-batch (vstab_set_plan_batch), so split-K
    # factors, Winograd or direct form and the kernel family do not depend on what a frame is batched with
    for world in (2, 3):
        parts = []
        for r in range(world):
            a, b = vdist.shard_range(F_, r, world)
            parts.append(_stabilise_u8(feats[a:b], frame[a:b], MB))
        assert torch.equal(torch.cat(parts), whole), world
    pf_whole = torch.cat([vs.flownetS_pyramid(feats[b0:b0 + MB].contiguous(), min(MB, F_ - b0))["predict_flow2"] for b0 in range(0, F_, MB)])
    pf_ragged = torch.cat([vs.flownetS_pyramid(feats[b0:b0 + 3].contiguous(), min(3, F_ - b0))["predict_flow2"] for b0 in range(0, F_, 3)])
    assert torch.equal(pf_whole, pf_ragged)
    # overlapped reassembly (what bench_clip.py does): every micro-batch all-gathered into its place of the full clip as it finishes
    sg = vdist.SequenceGatherer(F_, (H, W, 3), torch.uint8, torch.device("cuda", 0))
    assert sg.common == F_
    for b0 in range(0, F_, MB):
        bc = min(MB, F_ - b0)
        sg.submit(whole[b0:b0 + bc], b0)
    assert torch.equal(sg.finish(), whole)
    # streaming gatherer (what bench.py uses for N > 1): three submits through two rotating buffers
    fg = vdist.FrameGatherer((MB, H, W, 3), 1, torch.device("cuda", 0), dtype=torch.uint8)
    slots = [fg.submit(whole[i:i + MB] if i + MB <= F_ else whole[F_ - MB:F_]) for i in (0, 4, 8)]
    fg.drain()
    assert torch.equal(fg.result(slots[2]).view(MB, H, W, 3), whole[F_ - MB:F_])
    assert torch.equal(fg.result(slots[1]).view(MB, H, W, 3), whole[4:8])
