"""MI355X-native FlowNetS-pyramid optical flow + bilinear flow warp.

Drop-in for the hot path of posgraph/coupe.optical_flow_based_deep_video_stabilization
(model.flownetS_pyramid, tf_warp, get_pixel_value and the evaluate glue) with the
arithmetic in hand-written HIP kernels for gfx950 (libvstab_hip.so, C ABI in
include/vstab.h).  See DESIGN.md and INTEGRATION.md at the repository root.
"""
from .model import (flownetS_pyramid, initialize_global_variables, load_and_assign_npz_dict,  # noqa: F401
                    assign_weights)
from .warp_flow import tf_warp, get_pixel_value, resize_images, resize_images_slice3, flow_to_output_res, flow_glue_warp  # noqa: F401
from .pipeline import stabilise_originalsize, stabilise_native, OriginalSizeStabiliser  # noqa: F401
from . import spatial_transformer, warp, vgg16, NLDF, clip_driver, postfilters, training, train_step  # noqa: F401,E402  (secondary samplers)
