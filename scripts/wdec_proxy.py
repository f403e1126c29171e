#!/usr/bin/env python3
"""scratch (round 6): what would the Winograd-domain GEMM of F(2x2,2x2) transposed convolutions cost?  A proxy with the same workgroup
count and reduction length as the 9-position GEMM -- a 1x1 convolution of R*128 rows, K = the concat's pixel stride, N = 4*Cout -- run through
vstab_conv_forward; read the conv_mfma_kernel rows of a rocprofv3 --kernel-trace --stats of this script.  Rows per position at B=8 512x512
(tile grid (Hin/2+1)^2, zero tiles skipped): deconv3 19 + 4*17 + 4*16 = 151 row tiles, deconv4 6 + 4*5 + 4*4 = 42, deconv2 69+4*67+4*64=593."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, runtime
L = _lib.lib()
shapes = [("deconv3_wino_gemm", 151, 772, 512), ("deconv4_wino_gemm", 42, 1028, 1024), ("deconv2_wino_gemm", 593, 388, 256),
          ("deconv3_exact_rows", 144, 772, 512), ("deconv4_exact_rows", 36, 1028, 1024)]
st = runtime.stream_ptr()
for name, R, cin, cout in shapes:
    x = torch.randn(1, R, 128, cin, device="cuda")
    W = torch.randn(1, 1, cin, cout, device="cuda") * 0.05
    y = torch.empty(1, R, 128, cout, device="cuda")
    n = L.vstab_conv_forward_workspace_bytes(1, R, 128, cin, cin, 1, 1, 0, cout, cout, 0, 0, R, 128)
    assert n > 0, name
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    def run():
        _lib.check(L.vstab_conv_forward(x.data_ptr(), 1, R, 128, cin, 0, cin, W.data_ptr(), None, 1, 1, 0, y.data_ptr(), R, 128, cout, 0, cout, 0,
                                        ws.data_ptr(), ws.numel(), st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * R * 128 * cin * cout
    print(f"{name:22s} {ms*1e3:8.1f} us per call (incl. operand pack)  GEMM {fl/1e9:6.2f} GFLOP", flush=True)
