"""ctypes binding of libvstab_hip.so (include/vstab.h).  No CPU fallback: if the HIP
library cannot be built or loaded every entry point raises."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class VstabTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", c_float_p), ("ndim", C.c_int32), ("shape", C.c_int32 * 4)]


class VstabWsEntry(C.Structure):
    _fields_ = [("name", C.c_char * 24), ("offset_bytes", C.c_int64), ("n", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("c", C.c_int32), ("c_stride", C.c_int32)]


EXPORTS = {
    # name: (restype, argtypes)
    "vstab_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "vstab_destroy": (None, [C.c_void_p]),
    "vstab_last_error": (C.c_char_p, [C.c_void_p]),
    "vstab_version": (C.c_char_p, []),
    "vstab_load_weights": (C.c_int, [C.c_void_p, C.POINTER(VstabTensor), C.c_int]),
    "vstab_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "vstab_workspace_layout": (C.c_int, [C.c_int] * 4 + [C.POINTER(VstabWsEntry), C.c_int]),
    "vstab_workspace_layout_ctx": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.POINTER(VstabWsEntry), C.c_int]),
    "vstab_set_plan_batch": (C.c_int, [C.c_void_p, C.c_int]),
    "vstab_set_plan_flags": (C.c_int, [C.c_void_p, C.c_uint]),
    "vstab_workspace_bytes_ctx": (C.c_size_t, [C.c_void_p] + [C.c_int] * 4),
    "vstab_flownets_forward": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 5 +
                               [C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_flow_resize_scale": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "vstab_resize_bilinear": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] + [C.c_int] * 2 + [C.c_void_p]),
    "vstab_resize_bilinear_slice3": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] + [C.c_int] * 2 + [C.c_void_p]),
    "vstab_warp_flow": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "vstab_selftest_div_const": (C.c_int, [C.c_float, C.c_uint, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "vstab_flow_glue_warp": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]),
    "vstab_stabilise_originalsize": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] + [C.c_int] * 2 + [C.c_void_p] * 8 +
                                     [C.c_size_t, C.c_void_p]),
    "vstab_trace_ranges": (C.c_int, [C.c_int]),
    "vstab_hbm_profile_enable": (C.c_int, [C.c_int]),
    "vstab_hbm_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "vstab_get_pixel_value": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]),
    "vstab_st_transform": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_st_bilinear_interp": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vstab_st_meshgrid": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_homography_warp": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_transform_image": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_int] * 2 + [C.c_void_p]),
    "vstab_vec2mtrx": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vstab_vgg16_load": (C.c_int, [C.c_void_p, C.POINTER(VstabTensor), C.c_int]),
    "vstab_vgg16_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "vstab_vgg16_shapes": (C.c_int, [C.c_int, C.c_int, c_int32_p]),
    "vstab_vgg16_forward": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_scale_shift": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_float, c_float_p, C.c_void_p, C.c_void_p]),
    "vstab_maxpool2x2": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]),
    "vstab_nldf_load": (C.c_int, [C.c_void_p, C.POINTER(VstabTensor), C.c_int]),
    "vstab_nldf_workspace_bytes": (C.c_size_t, [C.c_int]),
    "vstab_nldf_forward": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]),
    "vstab_resize_u8": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_assemble_input": (C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p]),
    "vstab_assemble_input_resized": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    "vstab_clip_step": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 9 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_flow_glue_warp_u8": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "vstab_frame_to_float": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "vstab_quantise_output": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "vstab_flow_box_blur": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vstab_axpby": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_longlong, C.c_void_p]),
    "vstab_loss_level": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "vstab_conv_wgrad_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "vstab_conv_wgrad": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_conv_dgrad_workspace_bytes": (C.c_size_t, [C.c_int] * 14),
    "vstab_conv_dgrad": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p] + [C.c_int] * 6 +
                         [C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_conv_forward_workspace_bytes": (C.c_size_t, [C.c_int] * 14),
    "vstab_conv_forward": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p] + [C.c_int] * 6 +
                           [C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_conv_rowwin_forward_workspace_bytes": (C.c_size_t, [C.c_int] * 12),
    "vstab_conv_rowwin_forward": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 3 +
                                  [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_conv3x3_winograd_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "vstab_conv3x3_winograd": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p] + [C.c_int] * 3 +
                               [C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_resize_bilinear_backward": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "vstab_pad_nearest_upsample": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_pad_nearest_upsample_backward": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vstab_column_sum_scratch_bytes": (C.c_size_t, [C.c_longlong, C.c_int]),
    "vstab_column_sum": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_pf2_from_taps": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_pf2_taps_backward": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_predict2_tap_table": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vstab_adam_step": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_float] * 4 + [C.c_void_p]),
    "vstab_bn_scratch_bytes": (C.c_size_t, [C.c_longlong, C.c_int]),
    "vstab_bn_lrelu_train_forward": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_bn_lrelu_train_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_lrelu_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_void_p]),
    "vstab_flow_medfilt": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]),
    "vstab_flow_mean_fill": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p]),
    "vstab_conv3x3_winograd_wgrad_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "vstab_conv3x3_winograd_wgrad": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p, C.c_size_t,
                                               C.c_void_p]),
    "vstab_loss_main_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "vstab_loss_main": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    "vstab_resize_f32_to_u8": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_homography_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "vstab_homography_fit": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_uint, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "vstab_warp_perspective_u8": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_host_xcd_remap": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_int32)]),
    "vstab_level_sizes": (C.c_int, [C.c_int, C.c_int, c_int32_p]),
    "vstab_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "vstab_profile_reset": (C.c_int, [C.c_void_p]),
    "vstab_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "vstab_profile_read_direct": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "vstab_profile_kernel_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "vstab_host_layer_plan": (C.c_int, [C.c_int] * 5 + [c_int32_p, C.c_int]),
    "vstab_host_layer_plan_pinned": (C.c_int, [C.c_int, C.c_uint] + [C.c_int] * 5 + [c_int32_p, C.c_int]),
    "vstab_host_pack_layer": (C.c_longlong, [C.c_int, C.c_int, c_float_p, C.POINTER(C.c_double), c_float_p,
                                             C.c_longlong]),
    "vstab_host_wdec_plan": (C.c_int, [C.c_int] * 5 + [c_int32_p, C.c_int]),
    "vstab_host_pack_wdec": (C.c_longlong, [C.c_int, c_float_p, C.POINTER(C.c_double), c_float_p, C.c_longlong]),
}

class LossLevelDesc(C.Structure):
    """vstab_loss_level_desc (include/vstab.h)."""
    _fields_ = [("pf", C.c_void_p), ("grad", C.c_void_p), ("h", C.c_int), ("w", C.c_int), ("cs_pf", C.c_int), ("cs_grad", C.c_int),
                ("tv_weight", C.c_float)]


_lib = None


def lib():
    """Load (building if needed) libvstab_hip.so.  Raises if that is impossible."""
    global _lib
    if _lib is None:
        path = _build.LIB
        if os.environ.get("VSTAB_LIB"):                 # an explicitly chosen build of the same sources (the sanitizer build of the host side)
            path = os.environ["VSTAB_LIB"]
            if not os.path.exists(path):
                raise RuntimeError(f"VSTAB_LIB={path} does not exist")
        elif os.environ.get("VSTAB_NO_BUILD") != "1":
            path = _build.build()
        elif not os.path.exists(path):
            raise RuntimeError(f"{path} is missing and VSTAB_NO_BUILD=1")
        L = C.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class VstabError(RuntimeError):
    pass


def check(code: int, ctx=None):
    if code == 0:
        return
    msg = lib().vstab_last_error(ctx)
    msg = msg.decode() if msg else "?"
    if code in (-1, -2):          # VSTAB_E_SHAPE / VSTAB_E_ALIGN
        raise ValueError(f"vstab error {code}: {msg}")
    raise VstabError(f"vstab error {code}: {msg}")
