"""Host logic of libvstab_hip.so checked on the CPU: the forward schedule's per-layer
GEMM geometry (vstab_host_layer_plan) and the weight packer (vstab_host_pack_layer) are
run through a numpy emulation of what the implicit-GEMM kernel computes
(out[m, n] = sum_k A[m, k] * Wpacked[k, n], A gathered exactly as ConvParams describes)
and compared with the oracle's convolution / transposed convolution."""
import ctypes as C

import numpy as np
import pytest
import torch

from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, netspec
from oracle import vstab_oracle as vo

FIELDS = ("B Hi Wi Cs_in KH NSEG SEG SEGP SEG_STRIDE s_in s_out Ho Wo Cs_out c_off N Npad act nphase ksplit "
          "Mmax tile vec4 in_buf out_buf reserved").split()
PH_FIELDS = "Hg Wg M off_y off_x o_y o_x".split()


def layer_plan(B, H, W, Cin, layer):
    buf = (C.c_int32 * 64)()
    n = _lib.lib().vstab_host_layer_plan(B, H, W, Cin, layer, buf, 64)
    assert n > 0
    p = dict(zip(FIELDS, buf[:26]))
    p["ph"] = [dict(zip(PH_FIELDS, buf[26 + 7 * k:33 + 7 * k])) for k in range(p["nphase"])]
    return p


def pack_layer(Cin, layer, W, scale=None):
    W = np.ascontiguousarray(W, np.float32)
    cap = 64 * 1024 * 1024
    out = np.zeros(cap, np.float32)
    sc = None if scale is None else np.ascontiguousarray(scale, np.float64).ctypes.data_as(C.POINTER(C.c_double))
    n = _lib.lib().vstab_host_pack_layer(Cin, layer, W.ctypes.data_as(_lib.c_float_p), sc,
                                         out.ctypes.data_as(_lib.c_float_p), cap)
    assert n > 0, _lib.lib().vstab_last_error(None)
    return out[:n]


def swz32(n, k):
    return (((k >> 2) ^ ((n >> 1) & 7)) << 2) | (k & 3)


def emulate(inp, p, wpk):
    """inp [B,Hi,Wi,Cs_in] float64 -> out [B,Ho,Wo,Cs_out] float64 (NaN where the layer does not write)."""
    B, Hi, Wi, Cs = p["B"], p["Hi"], p["Wi"], p["Cs_in"]
    KT = p["KH"] * p["NSEG"] * p["SEGP"] // 32
    Np = p["Npad"]
    wpk = wpk.reshape(p["nphase"], KT, Np, 32).astype(np.float64)
    unsw = np.array([[swz32(n, k) for k in range(32)] for n in range(Np)])           # [Np, 32]
    Wm = np.take_along_axis(wpk, np.broadcast_to(unsw, wpk.shape), axis=3)             # logical k order
    Wm = Wm.transpose(0, 1, 3, 2).reshape(p["nphase"], KT * 32, Np)
    rows = inp.reshape(B, Hi, Wi * Cs)
    out = np.full((B, p["Ho"], p["Wo"], p["Cs_out"]), np.nan)
    for k, ph in enumerate(p["ph"]):
        m = np.arange(ph["M"])
        n, rem = np.divmod(m, ph["Hg"] * ph["Wg"])
        j, i = np.divmod(rem, ph["Wg"])
        A = np.zeros((ph["M"], KT * 32))
        for t in range(p["KH"]):
            iy = j * p["s_in"] + ph["off_y"] + t
            yok = (iy >= 0) & (iy < Hi)
            for s in range(p["NSEG"]):
                q = np.arange(p["SEG"])
                foff = ((i * p["s_in"] + ph["off_x"]) * Cs)[:, None] + (s * p["SEG_STRIDE"] + q)[None, :]
                ok = yok[:, None] & (foff >= 0) & (foff < Wi * Cs)
                vals = rows[n[:, None], np.clip(iy, 0, Hi - 1)[:, None], np.clip(foff, 0, Wi * Cs - 1)]
                k0 = (t * p["NSEG"] + s) * p["SEGP"]
                A[:, k0:k0 + p["SEG"]] = np.where(ok, vals, 0.0)
        res = A @ Wm[k]
        out[n, j * p["s_out"] + ph["o_y"], i * p["s_out"] + ph["o_x"], p["c_off"]:p["c_off"] + p["N"]] = res[:, :p["N"]]
    return out


ENC = netspec.ENCODER


@pytest.mark.parametrize("layer", list(range(10)))
@pytest.mark.parametrize("cin", [27, 6])
def test_encoder_layer_geometry_and_packing(layer, cin):
    if cin == 6 and layer > 0:
        pytest.skip("Cin only changes layer 0")
    H, W, B = (52, 44, 2) if layer < 4 else (96, 80, 1)
    p = layer_plan(B, H, W, cin, layer)
    st = ENC[layer]
    ci = cin if layer == 0 else ENC[layer - 1].cout
    rng = np.random.default_rng(layer)
    inp = rng.standard_normal((B, p["Hi"], p["Wi"], p["Cs_in"]))          # pad channels random on purpose
    Wt = rng.standard_normal((st.k, st.k, ci, st.cout)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, p["Npad"])
    got = emulate(inp, p, pack_layer(cin, layer, Wt, scale))
    ref = vo.pad_conv(torch.from_numpy(inp[..., :ci].copy()), torch.from_numpy(Wt.astype(np.float64)), None,
                      st.pad, st.stride).numpy() * scale[:st.cout]
    assert got.shape[1:3] == ref.shape[1:3]
    sl = got[..., p["c_off"]:p["c_off"] + st.cout]
    assert np.abs(sl - ref).max() < 1e-5 * max(1, np.abs(ref).max())    # packed weights are fp32(W*scale)
    assert np.isnan(got[..., st.cout:]).all()                            # nothing written outside the slice


@pytest.mark.parametrize("l", [0, 1, 2, 3])
@pytest.mark.parametrize("HW", [(64, 64), (88, 104)])     # second size has odd output grids
def test_deconv_phase_geometry_and_packing(l, HW):
    H, W = HW
    B = 1
    p = layer_plan(B, H, W, 27, 10 + l)
    cin = (1024, 1026, 770, 386)[l]
    cout = (512, 256, 128, 64)[l]
    rng = np.random.default_rng(10 + l)
    inp = rng.standard_normal((B, p["Hi"], p["Wi"], p["Cs_in"]))
    Wt = (rng.standard_normal((4, 4, cout, cin)) * 0.1).astype(np.float32)
    got = emulate(inp, p, pack_layer(27, 10 + l, Wt))
    ref = vo.deconv4x4s2(torch.from_numpy(inp[..., :cin].copy()), torch.from_numpy(Wt.astype(np.float64)), None,
                         (p["Ho"], p["Wo"])).numpy()
    sl = got[..., p["c_off"]:p["c_off"] + cout]
    assert not np.isnan(sl).any()                                        # the 4 phases cover every output pixel
    assert np.abs(sl - ref).max() < 1e-9 * max(1, np.abs(ref).max())
    assert np.isnan(got[..., :p["c_off"]]).all() and np.isnan(got[..., p["c_off"] + cout:]).all()


def test_predict2_tap_table():
    p = layer_plan(1, 64, 64, 27, 14)
    rng = np.random.default_rng(5)
    inp = rng.standard_normal((1, p["Hi"], p["Wi"], 196))
    Wt = rng.standard_normal((3, 3, 194, 2)).astype(np.float32)
    got = emulate(inp, p, pack_layer(27, 14, Wt))
    ref = np.einsum("bhwc,tco->bhwto", inp[..., :194], Wt.reshape(9, 194, 2).astype(np.float64)).reshape(1, p["Hi"], p["Wi"], 18)
    assert np.abs(got[..., :18] - ref).max() < 1e-9
    assert np.abs(got[..., 18:32]).max() == 0


@pytest.mark.parametrize("layer,cin,cs", [(15, 1024, 1024), (16, 1026, 1028), (17, 770, 772), (18, 386, 388)])
def test_predict_head_tap_tables(layer, cin, cs):
    p = layer_plan(1, 64, 64, 27, layer)
    assert p["Cs_in"] == cs
    rng = np.random.default_rng(layer)
    inp = rng.standard_normal((1, p["Hi"], p["Wi"], cs))
    Wt = rng.standard_normal((3, 3, cin, 2)).astype(np.float32)
    got = emulate(inp, p, pack_layer(27, layer, Wt))
    ref = np.einsum("bhwc,tco->bhwto", inp[..., :cin], Wt.reshape(9, cin, 2).astype(np.float64)).reshape(1, p["Hi"], p["Wi"], 18)
    assert np.abs(got[..., :18] - ref).max() < 1e-9 * max(1, np.abs(ref).max())
    assert np.abs(got[..., 18:32]).max() == 0


def test_split_k_plan_is_consistent():
    for (B, H, W) in ((8, 512, 512), (1, 256, 256), (2, 720, 1280)):
        for layer in range(19):
            p = layer_plan(B, H, W, 27, layer)
            KT = p["KH"] * p["NSEG"] * p["SEGP"] // 32
            kts = -(-KT // p["ksplit"])
            assert (p["ksplit"] - 1) * kts < KT                         # no empty split
            if p["ksplit"] > 1:
                assert p["N"] % 4 == 0 and p["Cs_out"] % 4 == 0 and p["c_off"] % 4 == 0
            if p["vec4"]:
                assert p["Cs_in"] % 4 == 0 and p["SEG"] % 4 == 0 and p["SEG_STRIDE"] % 4 == 0
            # tile ids: 128x128, 128x64, 128x32, 64x128, 64x64 (harness builds only), 256x32, weight-stream kernel (32 columns)
            assert p["tile"] != 4 and p["Npad"] % (128, 64, 32, 128, 64, 32, 32)[p["tile"]] == 0
            if p["tile"] == 6:                                           # few rows per phase, a ticket word per (phase, row tile, column block)
                assert p["Mmax"] <= 64 and -(-p["Mmax"] // 32) * (p["Npad"] // 32) * p["nphase"] <= 4096
                assert p["ksplit"] <= 16


def test_workspace_layout_is_disjoint_and_aligned():
    L = _lib.lib()
    ent = (_lib.VstabWsEntry * 24)()
    n = L.vstab_workspace_layout(8, 512, 512, 27, ent, 24)
    assert n == 19          # 10 activation buffers, 5 tap tables, the ticket words, split-K slabs, the two Winograd-domain buffers
    names = [e.name.decode() for e in ent[:n]]
    assert names[-4:] == ["tickets", "splitk", "winograd_in", "winograd_out"]      # the plan-dependent sizes come last (vstab.h)
    assert ent[n - 4].w == 4096                                                   # one ticket word per (phase, row tile, column block)
    total = L.vstab_workspace_bytes(8, 512, 512, 27)
    spans = []
    for e in ent[:n]:
        assert e.offset_bytes % 256 == 0
        spans.append((e.offset_bytes, e.offset_bytes + 4 * e.n * e.h * e.w * e.c_stride))
    spans.sort()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert spans[-1][1] <= total
    assert L.vstab_workspace_bytes(1, 2, 2, 27) == 0                      # too small for the net


def layer_tiles(B, H, W, flags=0, plan_batch=0):
    out = (C.c_int32 * 200)()
    tiles = []
    for layer in range(19):
        n = _lib.lib().vstab_host_layer_plan_pinned(plan_batch, flags, B, H, W, 27, layer, out, 200)
        assert n > 0, (B, H, W, layer)
        tiles.append((out[21], out[19], out[25]))          # tile id, split-K factor, Winograd form
    return tiles


def test_pinned_plan_is_the_plan_of_its_batch():
    # vstab_set_plan_batch: every decision that changes the order of a sample's sums (split-K factor, Winograd or direct form, weight-stream
    # or tiled kernel) of ANY batch up to the pinned one is the pinned batch's own (of one 2 GiB chunk of it: 16 x 1080p runs as 2 x 8)
    for (P, H, W) in ((8, 512, 512), (4, 384, 512), (16, 1080, 1920)):
        rb = 8 if H == 1080 else P
        ref = layer_tiles(rb, H, W)
        for b in (1, 3, rb):
            got = layer_tiles(b, H, W, plan_batch=P)
            assert [(t[1], t[2], t[0] == 6) for t in got] == [(t[1], t[2], t[0] == 6) for t in ref], (P, b)
    assert layer_tiles(1, 512, 512) != layer_tiles(1, 512, 512, plan_batch=8)        # a lone sample plans differently by itself
    out = (C.c_int32 * 200)()
    assert _lib.lib().vstab_host_layer_plan_pinned(4, 0, 5, 384, 512, 27, 3, out, 200) < 0      # beyond the pinned batch
    # flag 1: the round-3 schedule (few-row layers on the tiled kernel with a combine launch)
    assert any(t[0] == 6 for t in layer_tiles(1, 256, 256)) and not any(t[0] == 6 for t in layer_tiles(1, 256, 256, flags=1))


# ---- Winograd F(2x2,2x2) form of the transposed convolutions (round 6; csrc/winograd_ops.hip documents the algebra).  The host side --
# tile geometry, the 9-position GEMM's plan, the packed operands -- is run through the numpy GEMM emulation between numpy statements
# of the two transforms (the device kernels' arithmetic, statement for statement) and compared with the oracle's transposed convolution.
def wdec_plan(B, H, W, l):
    buf = (C.c_int32 * 128)()
    n = _lib.lib().vstab_host_wdec_plan(B, H, W, 27, l, buf, 128)
    assert n == 26 + 63 + 8
    p = dict(zip(FIELDS, buf[:26]))
    p["ph"] = [dict(zip(PH_FIELDS, buf[26 + 7 * k:33 + 7 * k])) for k in range(9)]
    g = list(buf[89:97])
    return p, dict(NTy=g[0], NTx=g[1], nty=g[2:5], ntx=g[5:8])


def wdec_input_np(x, g):
    """x [B,Hi,Wi,Cs] -> V [B, 9*NTy, NTx, Cs]; tiles a position does not need stay NaN (the GEMM must not read them)."""
    B, Hi, Wi, Cs = x.shape
    xp = np.zeros((B, 2 * g["NTy"] + 2, 2 * g["NTx"] + 2, Cs))
    xp[:, 1:1 + Hi, 1:1 + Wi] = x                                   # xp[r] = x[r - 1]: tile t reads rows 2t-1 .. 2t+1 = xp[2t .. 2t+2]
    V = np.full((B, 9 * g["NTy"], g["NTx"], Cs), np.nan)
    d = [[xp[:, i:i + 2 * g["NTy"]:2, j:j + 2 * g["NTx"]:2] for j in range(3)] for i in range(3)]
    t = [[d[0][j] - d[1][j] for j in range(3)], [d[1][j] for j in range(3)], [d[1][j] - d[2][j] for j in range(3)]]
    for i in range(3):
        v = [t[i][0] - t[i][1], t[i][1], t[i][1] - t[i][2]]
        for j in range(3):
            pos = i * 3 + j
            V[:, pos * g["NTy"]:pos * g["NTy"] + g["nty"][i], :g["ntx"][j]] = v[j][:, :g["nty"][i], :g["ntx"][j]]
    return V


def wdec_output_np(M, g, cout, Ho, Wo):
    """M [B, 9*NTy, NTx, 4*cout] (NaN where the GEMM wrote nothing) -> y [B,Ho,Wo,cout]"""
    B = M.shape[0]
    m = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(3):
            pos = i * 3 + j
            blk = np.zeros((B, g["NTy"], g["NTx"], 4 * cout))
            blk[:, :g["nty"][i], :g["ntx"][j]] = M[:, pos * g["NTy"]:pos * g["NTy"] + g["nty"][i], :g["ntx"][j]]
            m[i][j] = blk
    y = np.full((B, Ho, Wo, cout), np.nan)
    s = [[m[0][j] + m[1][j] for j in range(3)], [m[1][j] - m[2][j] for j in range(3)]]
    for a in range(2):
        for b in range(2):
            blk = s[a][0] + s[a][1] if b == 0 else s[a][1] - s[a][2]
            for py in range(2):
                for px in range(2):
                    for ty in range(g["NTy"]):
                        oy = 4 * ty + 2 * a - py
                        if not 0 <= oy < Ho:
                            continue
                        ox = 4 * np.arange(g["NTx"]) + 2 * b - px
                        ok = (ox >= 0) & (ox < Wo)
                        ph = 2 * py + px
                        y[:, oy, ox[ok]] = blk[:, ty, ok, ph * cout:(ph + 1) * cout]
    return y


@pytest.mark.parametrize("l", [1, 2, 3])
@pytest.mark.parametrize("HW", [(64, 64), (88, 104), (96, 128)])     # (88, 104): odd output grids on some levels
def test_winograd_deconv_geometry_packing_and_transforms(l, HW):
    H, W = HW
    B = 2
    d = layer_plan(B, H, W, 27, 10 + l)                               # the direct form: sizes and channel strides
    p, g = wdec_plan(B, H, W, l)
    cin = (1024, 1026, 770, 386)[l]
    cout = (512, 256, 128, 64)[l]
    assert p["N"] == 4 * cout and p["nphase"] == 9 and p["Cs_in"] == d["Cs_in"] and p["ksplit"] == 1
    assert g["NTy"] == d["Ho"] // 4 + 1 and g["NTx"] == d["Wo"] // 4 + 1
    rng = np.random.default_rng(20 + l)
    inp = rng.standard_normal((B, d["Hi"], d["Wi"], d["Cs_in"]))
    inp[..., cin:] = 0.0                                              # the concat's pad channels are zeros (and meet zero weights)
    Wt = (rng.standard_normal((4, 4, cout, cin)) * 0.1).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout)
    cap = 9 * (p["SEGP"] // 32) * 4 * cout * 32
    wpk = np.zeros(cap, np.float32)
    n = _lib.lib().vstab_host_pack_wdec(l, Wt.ctypes.data_as(_lib.c_float_p), scale.ctypes.data_as(C.POINTER(C.c_double)),
                                        wpk.ctypes.data_as(_lib.c_float_p), cap)
    assert n == cap
    V = wdec_input_np(inp, g)
    assert p["Hi"] == V.shape[1] and p["Wi"] == V.shape[2]
    Vz = np.where(np.isnan(V), 1e30, V)                               # a tile the GEMM should skip would blow the result up
    M = emulate(Vz, p, wpk)
    got = wdec_output_np(M, g, cout, d["Ho"], d["Wo"])
    ref = vo.deconv4x4s2(torch.from_numpy(inp[..., :cin].copy()), torch.from_numpy(Wt.astype(np.float64)), None,
                         (d["Ho"], d["Wo"])).numpy() * scale
    assert not np.isnan(got).any()                                    # every output pixel is produced
    assert np.abs(got - ref).max() < 2e-6 * max(1, np.abs(ref).max())       # operands are fp32(G g G^T * scale)
    # multiply-adds per output channel and input channel: one per GEMM row and phase column against four taps per output pixel, direct
    issued = sum(ph["M"] for ph in p["ph"]) * 4
    direct = B * d["Ho"] * d["Wo"] * 4
    assert issued / direct >= 9 / 16                                  # 9 multiplies per 4x4 outputs x 4 ... / 64, plus the ragged grid
    if min(d["Hi"], d["Wi"]) >= 12:
        assert issued / direct < 0.75, issued / direct
