#!/usr/bin/env python3
"""In-situ version of tools/conv_bench's stamps: s_memtime of wave 0 of every workgroup at entry / loop start / loop end / exit for every
conv_mfma launch of ONE forward on the network's own activations (B=8 512x512x27), from a diagnostic build of the library
(-DVSTAB_HARNESS -DVSTAB_STAMP; scripts/build_stamp_lib.sh; selected with VSTAB_LIB).  Not part of the product."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coupe.optical_flow_based_deep_video_stabilization_amd as vs   # noqa: E402
from coupe.optical_flow_based_deep_video_stabilization_amd import _lib   # noqa: E402

L = C.CDLL(os.environ["VSTAB_LIB"])
L.vstab_debug_stamp_read.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 512, 512)))
vs.initialize_global_variables(seed=1, cin=27)
g = torch.Generator().manual_seed(1000)
feats = torch.rand(B, H, W, 27, generator=g).cuda()
for _ in range(20):
    vs.flownetS_pyramid(feats, B)
torch.cuda.synchronize()
_lib.lib().vstab_debug_stamp_reset = L.vstab_debug_stamp_reset
L.vstab_debug_stamp_reset()
vs.flownetS_pyramid(feats, B)
torch.cuda.synchronize()
# launch order of conv_mfma launches in a forward (conv1 runs on the row-window kernel): see api.cpp forward_chunk
# (conv3_1 / conv4_1's GEMMs run on wino_gemm_stream_kernel and predict_flow2's taps on tap_panel_kernel at B=8 512x512: no stamps in those)
names = ["conv2", "conv3", "conv4", "conv5", "conv5_1 gemm", "conv6", "conv6_1 gemm",
         "pf6 taps", "deconv5", "pf5 taps", "deconv4", "pf4 taps", "deconv3", "pf3 taps", "deconv2"]
buf = np.zeros(8 * 8192, dtype=np.uint64)
print(f"{'launch':<14}{'wgs':>6}{'prologue':>10}{'loop':>10}{'epilogue':>10}{'total':>10}{'MHz':>7}{'span us':>9}   first-to-last start us / end us")
for slot, name in enumerate(names):
    assert L.vstab_debug_stamp_read(slot, buf.ctypes.data, buf.size) == 0
    s = buf.reshape(8192, 8).astype(np.int64)
    ok = (s[:, 3] > s[:, 0]) & (s[:, 5] > s[:, 4])
    s = s[ok]
    if len(s) == 0:
        print(name, "no stamps"); continue
    pro, loop, epi, tot = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 3] - s[:, 0]
    clk = np.median(tot / (s[:, 5] - s[:, 4]) * 100.0)
    span = (s[:, 5].max() - s[:, 4].min()) / 100.0
    print(f"{name:<14}{len(s):>6}{np.median(pro):>10.0f}{np.median(loop):>10.0f}{np.median(epi):>10.0f}{np.median(tot):>10.0f}{clk:>7.0f}{span:>9.1f}   "
          f"{(s[:, 4].max() - s[:, 4].min()) / 100.0:.1f} / {(s[:, 5].max() - s[:, 5].min()) / 100.0:.1f}")
    if name in ("conv2", "conv3", "deconv2", "conv4", "deconv3"):      # where does the span go?  start-time and lifetime distributions
        st = (s[:, 4] - s[:, 4].min()) / 100.0
        life = (s[:, 5] - s[:, 4]) / 100.0
        q = lambda v: " ".join(f"{np.percentile(v, k):7.1f}" for k in (0, 10, 25, 50, 75, 90, 100))
        print(f"      start us  (min p10 p25 p50 p75 p90 max): {q(st)}")
        print(f"      life  us  (min p10 p25 p50 p75 p90 max): {q(life)}")
        end = st + life
        print(f"      end   us  (min p10 p25 p50 p75 p90 max): {q(end)}")
        lastk = np.argsort(end)[-6:]
        print("      last to end (dispatch id, start us, life us, loop cycles): " + "  ".join(f"({int(np.nonzero(ok)[0][k])}, {st[k]:.1f}, {life[k]:.1f}, {int(loop[k])})" for k in lastk))
        srt = np.sort(st)
        print(f"      started within 1 us: {int((st < 1).sum())}; sorted start times at ranks 255/256/383/384/511/512/640/768/896: " +
              " ".join(f"{srt[min(k, len(srt) - 1)]:.1f}" for k in (255, 256, 383, 384, 511, 512, 640, 768, 896)))
        late = st > np.percentile(st, 50) + 1
        # which workgroups share a CU?  HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13; XCC_ID low bits
        hw, xcc = s[:, 6], s[:, 7] & 0xf
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
        ids = np.nonzero(ok)[0]
        first = st < 1
        by_cu = {}
        for wid, c in zip(ids[first], cu[first]):
            by_cu.setdefault(int(c), []).append(int(wid))
        pairs = [v for v in by_cu.values() if len(v) >= 2]
        diffs = sorted({abs(v[1] - v[0]) for v in pairs})
        print(f"      first-round workgroups on {len(by_cu)} distinct CUs; id differences of co-resident ones: {diffs[:12]}; xcc of id 0..15: {[int(x) for x in xcc[np.argsort(ids)][:16]]}")
        if late.any():
            for nm, m in (("first round", ~late), ("later", late)):
                print(f"      {nm:<12} prologue {np.median(pro[m]):8.0f}  loop {np.median(loop[m]):8.0f}  epilogue {np.median(epi[m]):8.0f}  (cycles, medians; loop p90 {np.percentile(loop[m], 90):.0f})")
            print(f"      workgroups starting in the second half: {int(late.sum())}, their life p50 {np.percentile(life[late], 50):.1f} us; the first half's {np.percentile(life[~late], 50):.1f} us")
buf.fill(0)
