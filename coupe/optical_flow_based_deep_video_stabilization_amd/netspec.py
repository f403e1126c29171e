"""Static description of the FlowNetS-pyramid graph (reference model.py:786-893).

Pure host logic, no torch/HIP: layer table, the size-generalisation rule for the
literals the reference hard-wires to 384x512, checkpoint
key naming (tensorlayer `save_npz_dict`) and FLOP accounting.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

BN_EPS = 1e-5          # tensorlayer BatchNormLayer default epsilon
LRELU_SLOPE = 0.1      # model.py:788


@dataclass(frozen=True)
class EncStage:
    name: str      # tensorlayer layer name == variable scope (model.py:808-844)
    k: int
    stride: int
    pad: int
    cout: int


# model.py:807-844 -- PadLayer(p) -> Conv2d(k, s, VALID) -> BatchNorm(no gamma) -> lrelu(0.1)
ENCODER: Tuple[EncStage, ...] = (
    EncStage("1", 7, 2, 3, 64),
    EncStage("2", 5, 2, 2, 128),
    EncStage("3", 5, 2, 2, 256),
    EncStage("3_1", 3, 1, 1, 256),
    EncStage("4", 3, 2, 1, 512),
    EncStage("4_1", 3, 1, 1, 512),
    EncStage("5", 3, 2, 1, 512),
    EncStage("5_1", 3, 1, 1, 512),
    EncStage("6", 3, 2, 1, 1024),
    EncStage("6_1", 3, 1, 1, 1024),
)

# decoder step k (model.py:847-880): (deconv name, flow-deconv name, predict name of the
# level being refined, skip tensor, deconv Cout)
DECODER = (
    ("deconv5", "upsample6_5", "predict6", "5_1", 512),
    ("deconv4", "upsample5_4", "predict5", "4_1", 256),
    ("deconv3", "upsample4_3", "predict4", "3_1", 128),
    ("deconv2", "upsample3_2", "predict3", "2", 64),
)

FLOW_KEYS = ("predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3", "predict_flow2")


def conv_out(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


@dataclass(frozen=True)
class Sizes:
    """Spatial sizes of every level for an H x W input (8a-note-1)."""
    H: int
    W: int
    enc: Tuple[Tuple[int, int], ...]   # output (h, w) of the 10 encoder stages

    @property
    def level(self) -> Dict[int, Tuple[int, int]]:
        # pyramid level -> spatial size; level 2 == conv2, 3 == conv3_1 ... 6 == conv6_1
        e = self.enc
        return {1: e[0], 2: e[1], 3: e[3], 4: e[5], 5: e[7], 6: e[9]}

    @property
    def pf2(self) -> Tuple[int, int]:
        return (self.H - 2, self.W - 2)    # model.py:885-886 literal (382, 510)


def sizes_for(H: int, W: int) -> Sizes:
    h, w = H, W
    enc: List[Tuple[int, int]] = []
    for st in ENCODER:
        h, w = conv_out(h, st.k, st.stride, st.pad), conv_out(w, st.k, st.stride, st.pad)
        if h < 1 or w < 1:
            raise ValueError(f"input {H}x{W} too small for FlowNetS (stage {st.name})")
        enc.append((h, w))
    sz = Sizes(H, W, tuple(enc))
    lv = sz.level
    for k in (5, 4, 3, 2):
        # transposed conv 4x4 s2 SAME with output_shape := skip size needs ceil(out/2) == in
        for a, b in zip(lv[k], lv[k + 1]):
            if (a + 1) // 2 != b:
                raise ValueError(f"level sizes {lv[k]} / {lv[k+1]} break the deconv shape rule")
    if H < 3 or W < 3:
        raise ValueError("input must be at least 3x3")
    return sz


def concat_channels() -> Dict[int, int]:
    """Channels of concat5..concat2 (model.py:853,862,871,880)."""
    out = {}
    skip_c = {"5_1": 512, "4_1": 512, "3_1": 256, "2": 128}
    for lvl, (_, _, _, skip, dc) in zip((5, 4, 3, 2), DECODER):
        out[lvl] = skip_c[skip] + dc + 2
    return out


def weight_shapes(cin: int = 27) -> Dict[str, Tuple[int, ...]]:
    """Short-name -> shape of every variable of the graph (reference layouts:
    conv HWIO, deconv [kh,kw,Cout,Cin])."""
    shp: Dict[str, Tuple[int, ...]] = {}
    c = cin
    for st in ENCODER:
        shp[f"{st.name}/W_conv2d"] = (st.k, st.k, c, st.cout)
        shp[f"{st.name}/b_conv2d"] = (st.cout,)
        for v in ("beta", "moving_mean", "moving_variance"):
            shp[f"{st.name}/{v}"] = (st.cout,)
        c = st.cout
    cc = concat_channels()
    pred_cin = {6: 1024, 5: cc[5], 4: cc[4], 3: cc[3], 2: cc[2]}
    for lvl, ci in pred_cin.items():
        shp[f"predict{lvl}/W_conv2d"] = (3, 3, ci, 2)
        shp[f"predict{lvl}/b_conv2d"] = (2,)
    dec_cin = {"deconv5": 1024, "deconv4": cc[5], "deconv3": cc[4], "deconv2": cc[3]}
    for dname, uname, _, _, dc in DECODER:
        shp[f"{dname}/W_deconv2d"] = (4, 4, dc, dec_cin[dname])
        shp[f"{dname}/b_deconv2d"] = (dc,)
        for v in ("beta", "moving_mean", "moving_variance"):
            shp[f"{dname}_bn/{v}"] = (dc,)
        shp[f"{uname}/W_deconv2d"] = (4, 4, 2, 2)
        shp[f"{uname}/b_deconv2d"] = (2,)
    return shp


def ckpt_key(short: str, outer: str = "main_net", scope: str = "flownetS") -> str:
    """tensorlayer save_npz_dict key for a short variable name."""
    return f"{outer}/{scope}/{short}:0"


def strip_ckpt_keys(d: dict, scope: str = "flownetS") -> dict:
    """Accept either short names or full `…/flownetS/<short>:0` checkpoint keys."""
    out = {}
    tag = f"{scope}/"
    for k, v in d.items():
        s = k
        if tag in s:
            s = s.split(tag, 1)[1]
        if s.endswith(":0"):
            s = s[:-2]
        out[s] = v
    return out


def gflop_per_sample(H: int, W: int, cin: int = 27) -> float:
    """Algorithmic FLOPs (2*MAC) of one sample, deconvs at 4 taps/output."""
    sz = sizes_for(H, W)
    mac = 0
    c = cin
    for st, (h, w) in zip(ENCODER, sz.enc):
        mac += h * w * st.k * st.k * c * st.cout
        c = st.cout
    lv = sz.level
    cc = concat_channels()
    pred_cin = {6: 1024, 5: cc[5], 4: cc[4], 3: cc[3]}
    for lvl, ci in pred_cin.items():
        mac += lv[lvl][0] * lv[lvl][1] * 9 * ci * 2
    dec_cin = {5: 1024, 4: cc[5], 3: cc[4], 2: cc[3]}
    dec_cout = {5: 512, 4: 256, 3: 128, 2: 64}
    for lvl in (5, 4, 3, 2):
        mac += lv[lvl][0] * lv[lvl][1] * 4 * dec_cin[lvl] * dec_cout[lvl]
        mac += lv[lvl][0] * lv[lvl][1] * 4 * 2 * 2
    mac += (H - 2) * (W - 2) * 9 * cc[2] * 2
    return 2.0 * mac / 1e9
