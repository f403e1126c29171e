"""Weights of the FlowNetS-pyramid graph in the reference's own layouts and names.

The reference restores `fixed_ckpt/<mode>.npz` with
`tl.files.load_and_assign_npz_dict` (main:520); that file is a Google-Drive
download (README.md:24) and is not available offline, so benchmarks and tests use
seeded synthetic weights with the initialisers the graph declares (model.py:790 for
convs, tensorlayer's DeConv2dLayer default for deconvs).  Host-side numpy only.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import netspec


def synthetic_weights(seed: int = 1, cin: int = 27, random_bn: bool = False,
                      flow_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Seeded weights, short names (netspec.weight_shapes).

    conv W ~ N(0, 2/fan_in) (variance-scaling, fan-in, factor 2), conv bias 0;
    deconv W ~ N(0, 0.02^2) clipped at 2 sigma, bias 0; BN beta 0, mean 0, var 1.
    `random_bn=True` draws beta, mean ~ N(0, 0.1), var ~ U(0.5, 1.5) and small random
    biases so that BN folding and every bias path are exercised.
    `flow_gain` scales the 2-channel predict heads (larger flows for warp tests).
    """
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shp in netspec.weight_shapes(cin).items():
        leaf = name.split("/")[1]
        if leaf == "W_conv2d":
            fan_in = shp[0] * shp[1] * shp[2]
            w = rng.standard_normal(shp) * np.sqrt(2.0 / fan_in)
            if name.startswith("predict"):
                w = w * flow_gain
        elif leaf == "W_deconv2d":
            w = np.clip(rng.standard_normal(shp), -2.0, 2.0) * 0.02
        elif leaf in ("b_conv2d", "b_deconv2d"):
            w = rng.standard_normal(shp) * 0.05 if random_bn else np.zeros(shp)
        elif leaf == "beta":
            w = rng.standard_normal(shp) * 0.1 if random_bn else np.zeros(shp)
        elif leaf == "moving_mean":
            w = rng.standard_normal(shp) * 0.1 if random_bn else np.zeros(shp)
        elif leaf == "moving_variance":
            w = rng.uniform(0.5, 1.5, shp) if random_bn else np.ones(shp)
        else:  # pragma: no cover
            raise KeyError(name)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def load_npz_dict(path: str, scope: str = "flownetS") -> Dict[str, np.ndarray]:
    """Read a tensorlayer `save_npz_dict` checkpoint (main:424-426 writes it,
    main:520 reads it) into short names; validates every shape."""
    with np.load(path, allow_pickle=False) as z:
        raw = {k: z[k] for k in z.files}
    return validate(netspec.strip_ckpt_keys(raw, scope))


def save_npz_dict(path: str, weights: Dict[str, np.ndarray], outer: str = "main_net",
                  scope: str = "flownetS") -> None:
    np.savez(path, **{netspec.ckpt_key(k, outer, scope): v for k, v in weights.items()})


def validate(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    w = netspec.strip_ckpt_keys(weights)
    if "1/W_conv2d" not in w:
        raise KeyError("weights lack '1/W_conv2d'")
    cin = int(np.asarray(w["1/W_conv2d"]).shape[2])
    out = {}
    for name, shp in netspec.weight_shapes(cin).items():
        if name not in w:
            raise KeyError(f"missing variable {name!r}")
        a = np.ascontiguousarray(np.asarray(w[name]), dtype=np.float32)
        if tuple(a.shape) != tuple(shp):
            raise ValueError(f"{name}: shape {a.shape}, expected {shp}")
        out[name] = a
    return out
