#!/usr/bin/env python3
"""Write a checkpoint the way the reference's training script writes it -- without TensorFlow or TensorLayer.

What the reference does (main = main_flownetS_pyramid_noprevloss_dataloader.py):
  * main:183-185   the network is built inside `tf.variable_scope('main_net')`; model.py:786 gives it the scope 'flownetS';
  * model.py:805-887   every layer is a TensorLayer layer with an explicit `name=`; TensorLayer 1.x opens `tf.variable_scope(name)` and
    creates   Conv2d -> `W_conv2d` [kh,kw,Cin,Cout], `b_conv2d` [Cout];   DeConv2dLayer -> `W_deconv2d` with exactly the `shape=` argument
    ([kh,kw,Cout,Cin]: tf.nn.conv2d_transpose's filter layout), `b_deconv2d` [Cout];   BatchNormLayer(gamma_init=None) -> `beta`,
    `moving_mean`, `moving_variance` (no gamma).  PadLayer / ConcatLayer / UpSampling2dLayer / ElementwiseLayer / InputLayer own none;
  * main:329-330   `save_vars = tl.layers.get_variables_with_name('main_net', False, False)`: every global variable whose name contains
    'main_net', in creation order -- taken BEFORE the optimiser exists (main:333-335), so no Adam slots;
  * main:424-426   `tl.files.save_npz_dict(save_vars, name=...)` = `np.savez(name, **{var.name: value})`, var.name = '<scopes>/<leaf>:0'.

The layer list below is transcribed from the TEXT of model.py:805-887 (name= and shape= arguments, in order), independently of
`netspec.weight_shapes`, so that the loader is checked against the reference's own naming rather than against itself.  Values are seeded
noise (no trained weights exist offline: README.md:24 is a Google-Drive link).

    python tests/golden/make_tl_checkpoint.py out.npz [--cin 27] [--seed 0]
"""
import argparse

import numpy as np

# (kind, name=, filter size, n_filter / shape=)  in graph-construction order, model.py:807-885
LAYERS = [
    ("conv", "1", 7, 64), ("bn", "1"),                                  # model.py:808-809
    ("conv", "2", 5, 128), ("bn", "2"),                                 # :811-812
    ("conv", "3", 5, 256), ("bn", "3"),                                 # :814-815
    ("conv", "3_1", 3, 256), ("bn", "3_1"),                             # :819-820
    ("conv", "4", 3, 512), ("bn", "4"),                                 # :823-824
    ("conv", "4_1", 3, 512), ("bn", "4_1"),                             # :827-828
    ("conv", "5", 3, 512), ("bn", "5"),                                 # :831-832
    ("conv", "5_1", 3, 512), ("bn", "5_1"),                             # :835-836
    ("conv", "6", 3, 1024), ("bn", "6"),                                # :839-840
    ("conv", "6_1", 3, 1024), ("bn", "6_1"),                            # :843-844
    ("conv", "predict6", 3, 2),                                         # :848
    ("deconv", "deconv5", (4, 4, 512, 1024)), ("bn", "deconv5_bn"),     # :850-851
    ("deconv", "upsample6_5", (4, 4, 2, 2)),                            # :852
    ("conv", "predict5", 3, 2),                                         # :856   input = concat5: 512 + 512 + 2
    ("deconv", "deconv4", (4, 4, 256, 1026)), ("bn", "deconv4_bn"),     # :859-860
    ("deconv", "upsample5_4", (4, 4, 2, 2)),                            # :861
    ("conv", "predict4", 3, 2),                                         # :865   concat4: 512 + 256 + 2
    ("deconv", "deconv3", (4, 4, 128, 770)), ("bn", "deconv3_bn"),      # :868-869
    ("deconv", "upsample4_3", (4, 4, 2, 2)),                            # :870
    ("conv", "predict3", 3, 2),                                         # :874   concat3: 256 + 128 + 2
    ("deconv", "deconv2", (4, 4, 64, 386)), ("bn", "deconv2_bn"),       # :877-878
    ("deconv", "upsample3_2", (4, 4, 2, 2)),                            # :879
    ("conv", "predict2", 3, 2),                                         # :885   concat2 (nearest-upsampled): 128 + 64 + 2
]
# what feeds each Conv2d (its Cin is inferred from the incoming tensor): the previous stage, or a concat (model.py:853,862,871,880)
CONV_CIN = {"predict6": 1024, "predict5": 512 + 512 + 2, "predict4": 512 + 256 + 2, "predict3": 256 + 128 + 2, "predict2": 128 + 64 + 2}


def variables(cin=27, seed=0, outer="main_net", scope="flownetS"):
    """[(full variable name, array)] in creation order."""
    rng = np.random.default_rng(seed)
    out, c, last_cout = [], cin, None
    for layer in LAYERS:
        kind, name = layer[0], layer[1]
        pre = f"{outer}/{scope}/{name}/"
        if kind == "conv":
            k, nf = layer[2], layer[3]
            ci = CONV_CIN.get(name, c)
            out.append((pre + "W_conv2d:0", (rng.standard_normal((k, k, ci, nf)) * np.sqrt(2.0 / (k * k * ci))).astype(np.float32)))
            out.append((pre + "b_conv2d:0", (rng.standard_normal(nf) * 0.01).astype(np.float32)))
            if name not in CONV_CIN:
                c = nf
            last_cout = nf
        elif kind == "deconv":
            shape = layer[2]
            out.append((pre + "W_deconv2d:0", (rng.standard_normal(shape) * 0.02).astype(np.float32)))
            out.append((pre + "b_deconv2d:0", (rng.standard_normal(shape[2]) * 0.01).astype(np.float32)))
            last_cout = shape[2]
        else:       # BatchNormLayer(gamma_init=None): beta, moving_mean, moving_variance over the incoming channels
            out.append((pre + "beta:0", (rng.standard_normal(last_cout) * 0.1).astype(np.float32)))
            out.append((pre + "moving_mean:0", (rng.standard_normal(last_cout) * 0.1).astype(np.float32)))
            out.append((pre + "moving_variance:0", rng.uniform(0.5, 1.5, last_cout).astype(np.float32)))
    return out


def write(path, cin=27, seed=0):
    vs = variables(cin, seed)
    np.savez(path, **dict(vs))          # tl.files.save_npz_dict: np.savez(name, **{var.name: value})
    return vs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--cin", type=int, default=27)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    v = write(a.out, a.cin, a.seed)
    print(f"{len(v)} variables, {sum(x.size for _, x in v) / 1e6:.1f} M parameters -> {a.out}")
