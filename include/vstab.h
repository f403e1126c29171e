/*
 * vstab.h -- C ABI of libvstab_hip.so: FlowNetS-pyramid optical-flow inference and
 * bilinear flow warp for MI355X (gfx950), hand-written HIP kernels.
 *
 * The reference (posgraph/coupe.optical_flow_based_deep_video_stabilization) is pure
 * Python on TensorFlow 1.10 + TensorLayer and has no FFI of its own: its boundary for
 * this path is the pair of Python graph-builder calls
 *     flownetS_pyramid(feats, batch_size, is_train, reuse, scope)      model.py:786-893
 *     tf_warp(img, flow, H, W) / get_pixel_value(img, x, y)            main:70-130 / 44-68
 * ("main" = main_flownetS_pyramid_noprevloss_dataloader.py) plus the glue lines
 * main:497-498 and main:806 and the checkpoint restore main:520.  Each entry point below
 * cites the reference lines it replaces; INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - all tensors fp32, NHWC, contiguous, DEVICE pointers unless a parameter says "host";
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream); no hidden host<->device copies, no allocation after vstab_load_weights;
 *   - the caller owns every I/O buffer and the workspace; the context owns only the
 *     packed weights (everything a forward writes -- activations, split-K slabs, the ticket
 *     words of the in-launch reductions -- lies in the caller's workspace);
 *   - every function returns 0 or a negative VSTAB_E_* code and never throws;
 *     vstab_last_error() gives the message for the last failure on that context
 *     (or the last context-less failure when ctx == NULL);
 *   - one context per device; forwards on one context may overlap in time (different
 *     streams, also issued from different host threads) when each uses its OWN workspace:
 *     a successful forward only reads the context (a failing one writes its message under a
 *     lock; with two threads failing at once vstab_last_error(ctx) is either message, and
 *     vstab_last_error(NULL) is always the calling thread's own).  Calls that write the
 *     context (vstab_load_weights, vstab_set_plan_batch, the DIAGNOSTIC section at the end
 *     of this header -- vstab_profile_enable(ctx, 1) makes every forward write its event
 *     and name slots) must not overlap with anything on that context;
 *   - product surface first; everything under "DIAGNOSTIC AND TEST SURFACE" at the end
 *     (plan flags, per-launch profilers, roctx ranges, self-tests, host-only views of the
 *     launch plan) exists for measurements and tests, is process- or context-global state,
 *     and is not part of the drop-in contract.
 */
#ifndef VSTAB_H
#define VSTAB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSTAB_OK          0
#define VSTAB_E_SHAPE    -1   /* bad dimensions / unsupported size                      */
#define VSTAB_E_ALIGN    -2   /* pointer not 16-byte aligned where required             */
#define VSTAB_E_HIP      -3   /* a HIP runtime call or kernel launch failed             */
#define VSTAB_E_NOMEM    -4   /* workspace too small / allocation failed                */
#define VSTAB_E_WEIGHTS  -5   /* missing or mis-shaped variable in vstab_load_weights   */
#define VSTAB_E_STATE    -6   /* call order (e.g. forward before load_weights)          */

#if defined(__GNUC__)
#define VSTAB_API __attribute__((visibility("default")))
#else
#define VSTAB_API
#endif

typedef struct vstab_ctx vstab_ctx;

/* One named variable of the checkpoint, in the reference's own layout:
 * conv W [kh][kw][Cin][Cout], deconv W [kh][kw][Cout][Cin], vectors [C].
 * `name` is the short variable name ("1/W_conv2d", "deconv5_bn/beta",
 * "upsample6_5/W_deconv2d", ... = the tensorlayer key without the
 * "main_net/flownetS/" prefix and ":0" suffix).  `data` is a HOST pointer. */
typedef struct vstab_tensor {
    const char  *name;
    const float *data;
    int32_t      ndim;
    int32_t      shape[4];
} vstab_tensor;

/* A tensor living in the forward workspace (for tests / debugging). */
typedef struct vstab_ws_entry {
    char    name[24];
    int64_t offset_bytes;
    int32_t n, h, w, c;      /* logical NHWC shape                           */
    int32_t c_stride;        /* floats between consecutive pixels (>= c)     */
} vstab_ws_entry;

/* ---- context ------------------------------------------------------------------- */
VSTAB_API int  vstab_create(vstab_ctx **out, int device);
VSTAB_API void vstab_destroy(vstab_ctx *ctx);
VSTAB_API const char *vstab_last_error(const vstab_ctx *ctx);
VSTAB_API const char *vstab_version(void);

/* Replaces tl.files.load_and_assign_npz_dict (main:520) + the variable creation inside
 * flownetS_pyramid: validates every variable of the graph, folds inference BatchNorm
 * (no gamma, eps 1e-5; model.py:809) into W and b, packs the weights into the MFMA
 * operand layout and uploads them.  Synchronous.  Input channels = shape[2] of
 * "1/W_conv2d" (27 in the reference). */
VSTAB_API int vstab_load_weights(vstab_ctx *ctx, const vstab_tensor *tensors, int count);

/* Bytes of workspace vstab_flownets_forward needs for this problem (0 on bad shape).  No
 * single tensor may reach 2 GiB (32-bit buffer offsets): larger batches are processed in
 * chunks of the largest batch that fits, and the workspace is sized for one chunk. */
VSTAB_API size_t vstab_workspace_bytes(int B, int H, int W, int Cin);

/* ---- pinned launch plans.  The forward takes a few per-layer decisions that change the ORDER of a sample's floating-point sums:
 * split-K factors, Winograd or direct form of the 3x3 stride-1 stages, weight-stream or tiled kernel for few-row layers.  By default
 * they are taken for the batch of each call, so the same sample can come out a few ulp different when it is batched differently (a
 * ragged last micro-batch of a sharded clip: main:553-558's samples are independent).  vstab_set_plan_batch(ctx, P)
 * pins them to what a batch of P samples gets: every call with B <= P then gives each sample bit-identical results, whatever B is
 * (B > P is VSTAB_E_SHAPE).  P = 0 (default) unpins.  The workspace of a pinned context is sized by vstab_workspace_bytes_ctx
 * (a workspace sized for P itself always suffices); vstab_workspace_bytes / vstab_workspace_layout describe the UNPINNED plan. */
VSTAB_API int vstab_set_plan_batch(vstab_ctx *ctx, int batch);
VSTAB_API size_t vstab_workspace_bytes_ctx(const vstab_ctx *ctx, int B, int H, int W, int Cin);

/* Names/offsets of the intermediate tensors inside the workspace.  Returns the number
 * of entries written (<= max_entries) or a negative error. */
VSTAB_API int vstab_workspace_layout(int B, int H, int W, int Cin, vstab_ws_entry *entries, int max_entries);
/* The same for the plan THIS context would run (its pinned batch and plan flags): the activation offsets depend on the shape alone,
 * the sizes of "splitk", "winograd_in", "winograd_out" (always the last three entries) on the plan. */
VSTAB_API int vstab_workspace_layout_ctx(const vstab_ctx *ctx, int B, int H, int W, int Cin, vstab_ws_entry *entries, int max_entries);

/* ---- the network: flownetS_pyramid(feats, batch_size, is_train=False) model.py:786-893
 * feats [B,H,W,Cin] -> predict_flow6..3 at the pyramid levels and predict_flow2
 * [B,H-2,W-2,2] (the 'flow' key is the same tensor).  The 384x512 literals of the
 * reference are generalised by SURVEY.md 8a-note-1.  `workspace` must be 256-byte
 * aligned and at least vstab_workspace_bytes() long. */
VSTAB_API int vstab_flownets_forward(vstab_ctx *ctx, const float *feats, int B, int H, int W, int Cin,
                           float *pf6, float *pf5, float *pf4, float *pf3, float *pf2,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ---- glue: main:497-498 with the literals generalised (384 -> net_h, 512 -> net_w, 382 -> h = the flow's height).
 * out[B,oh,ow,2] = resize_images(flow*net_h/h, [oh,ow]), then x = x*ow/net_w, y = y*oh/net_h, every `*` and `/`
 * a separate fp32 operation in the reference's order ((a*b)/c) (legacy TF bilinear, align_corners=False; identity
 * resample when the size already matches). */
VSTAB_API int vstab_flow_resize_scale(const float *flow, int B, int h, int w, float *out, int oh, int ow,
                            int net_h, int net_w, void *stream);

/* ---- glue: main:806 / model.py:857 UpSampling2dLayer defaults.  Legacy TF bilinear
 * resize of an NHWC tensor. */
VSTAB_API int vstab_resize_bilinear(const float *x, int B, int h, int w, int C, float *out, int oh, int ow,
                          void *stream);

/* ---- main:806 without an intermediate copy: out [B,oh,ow,3] = tf.image.resize_images(x[..., c_off:c_off+3], [oh, ow]) for an
 * NHWC tensor x with Cs channels per pixel (the unstable frame is channels 24:27 of the 27-channel input stack).  Legacy
 * bilinear as vstab_resize_bilinear (bit-identical to it on a contiguous 3-channel copy).  out 16-byte aligned. */
VSTAB_API int vstab_resize_bilinear_slice3(const float *x, int B, int h, int w, int Cs, int c_off, float *out, int oh, int ow,
                                 void *stream);

/* ---- tf_warp(img, flow, H, W) main:70-130.  img [B,H,W,C], flow [B,H,W,2] (x, y),
 * out [B,H,W,C]; truncating corners, clipped indices, weights from the clipped corners. */
VSTAB_API int vstab_warp_flow(const float *img, const float *flow, float *out, int B, int H, int W, int C,
                    void *stream);

/* ---- main:497-514 as ONE launch (evaluate_originalSize builds them as one graph): outflow = the glue of
 * vstab_flow_resize_scale applied to flow [B,h,w,2], at [B,oh,ow,2]; warped = tf_warp(img [B,oh,ow,3], outflow).
 * `outflow` may be NULL when the caller does not need the output-resolution flow (it is then never written); when
 * given it is written once and not re-read.  Bit-identical to vstab_flow_resize_scale followed by vstab_warp_flow.
 * C must be 3; img/outflow/warped 16-byte aligned; B*oh*ow < 2^31. */
VSTAB_API int vstab_flow_glue_warp(const float *flow, int B, int h, int w, const float *img, float *outflow, float *warped,
                         int oh, int ow, int C, int net_h, int net_w, void *stream);

/* ---- evaluate_originalSize's graph (main:491-514) as ONE call: vstab_flownets_forward, then vstab_flow_glue_warp with the
 * constants of main:497-498 (net_h = H, net_w = W, the flow's height H-2) on a 3-channel frame [B,oh,ow,3].  Every
 * output buffer is the caller's (pre-allocated once, re-used every step): a step is one call and no allocation.  `outflow` may
 * be NULL.  Same results as the two calls. */
VSTAB_API int vstab_stabilise_originalsize(vstab_ctx *ctx, const float *feats, int B, int H, int W, int Cin, const float *frame,
                                           int oh, int ow, float *pf6, float *pf5, float *pf4, float *pf3, float *pf2,
                                           float *outflow, float *warped, void *workspace, size_t workspace_bytes, void *stream);

/* ---- get_pixel_value(img, x, y) main:44-68.  x, y int32 [B,H,W] -> out[b,h,w,:] =
 * img[b, y, x, :].  Indices are clamped into the image instead of faulting. */
VSTAB_API int vstab_get_pixel_value(const float *img, const int32_t *x, const int32_t *y, float *out, int B,
                          int H, int W, int C, int Hi, int Wi, void *stream);

/* ---- secondary samplers named by north_star (never executed by the reference's live scripts) ---- */
/* AffineTransformer.transform / ProjectiveTransformer.transform / transformer()
 * (spatial_transformer.py:400-452, 539-608, 34-38): theta [B,6] or [B,8] (theta_dim = 6 | 8; the
 * projective 3x3 gets a trailing 1), sampling grid linspace(-1,1) of the OUTPUT size, bilinear_interp
 * sampler with a 1-pixel zero border.  img [B,H,W,C] -> out [B,oh,ow,C]. */
VSTAB_API int vstab_st_transform(const float *img, int B, int H, int W, int C, const float *theta, int theta_dim,
                                 float *out, int oh, int ow, void *stream);
/* bilinear_interp(im, x, y, out_size) / _interpolate (spatial_transformer.py:902-964, 787-792): x, y flat
 * [B*oh*ow] normalised to [-1,1], out_size = (oh, ow) as the reference passes it; out [B*oh*ow, C]. */
VSTAB_API int vstab_st_bilinear_interp(const float *img, int B, int H, int W, int C, const float *x, const float *y,
                                       int oh, int ow, float *out, void *stream);
/* _meshgrid(out_size) (spatial_transformer.py:755-779): out[3*oh*ow] = x_t row, y_t row, ones. */
VSTAB_API int vstab_st_meshgrid(float *out, int oh, int ow, void *stream);
/* warp.transformImage / transformCropImage (warp.py:46-86, 89-129): M [B,9] = refMtrx . pMtrx maps the canonical
 * linspace(-1,1) grid of the OUTPUT size to source pixel coordinates; floor/ceil taps, zero outside. */
VSTAB_API int vstab_homography_warp(const float *img, int B, int Hi, int Wi, int C, const float *M, float *out, int oh,
                                    int ow, void *stream);
/* The same with the composition warp.py:48-49 makes in front of it (tf.matmul(refMtrx, pMtrx)) done by the kernel: ref [9] (device,
 * one matrix for the batch), pM [B,9]; M = ref . pM with every product and sum rounded to fp32, (r0*p0 + r1*p1) + r2*p2. */
VSTAB_API int vstab_transform_image(const float *img, int B, int Hi, int Wi, int C, const float *ref, const float *pM, float *out,
                                    int oh, int ow, void *stream);
/* warp.vec2mtrx (warp.py:25-43): p [B,8] (homography, sl(3) generator) or [B,6] (affine) -> [B,9]
 * Taylor matrix exponential with `warp_approx` terms. */
VSTAB_API int vstab_vec2mtrx(const float *p, int B, int dim, int warp_approx, float *out, void *stream);

/* ---- VGG16 convolutional trunk (vgg16.py:25-64; BASELINE config 5) ------------------------------
 * 13 x (conv 3x3 SAME + bias + ReLU) and 5 x (max pool 2x2 stride 2 SAME).  Weights in the layout of
 * vgg16.npy: for every layer L in conv1_1 .. conv5_3 a tensor "L/filter" [3][3][Cin][Cout] and
 * "L/biases" [Cout] (HOST pointers).  The input is whatever the caller feeds vgg.build -- NLDF.py:29
 * feeds x*255 - VGG_MEAN, see vstab_scale_shift. */
VSTAB_API int vstab_vgg16_load(vstab_ctx *ctx, const vstab_tensor *tensors, int count);
VSTAB_API size_t vstab_vgg16_workspace_bytes(int B, int H, int W);
/* Shapes of the 18 outputs in build() order (conv1_1, conv1_2, pool1, conv2_1, conv2_2, pool2, conv3_1,
 * conv3_2, conv3_3, pool3, conv4_1..3, pool4, conv5_1..3, pool5): hwc54[3*i..3*i+2] = (h, w, c). */
VSTAB_API int vstab_vgg16_shapes(int H, int W, int32_t *hwc54);
/* input [B,H,W,3] -> outputs[18] (device pointers, caller-allocated, NHWC, shapes as above).  Batches whose
 * conv1 activations would reach 2 GiB are processed in chunks. */
VSTAB_API int vstab_vgg16_forward(vstab_ctx *ctx, const float *input, int B, int H, int W, float *const *outputs18,
                                  void *workspace, size_t workspace_bytes, void *stream);
/* out = x * scale - mean[c]  (NLDF.py:29: input_holder * 255. - vgg16.VGG_MEAN), x [npix, C], C <= 4, mean HOST. */
VSTAB_API int vstab_scale_shift(const float *x, long long npix, int C, float scale, const float *mean, float *out,
                                void *stream);
/* tf.nn.max_pool(ksize 2, strides 2, SAME) on NHWC, C % 4 == 0 (vgg16.py:51-53). */
VSTAB_API int vstab_maxpool2x2(const float *x, int B, int H, int W, int C, float *out, void *stream);

/* ---- NLDF saliency head (NLDF.py:24-101; tertiary, never instantiated by the reference) ----------
 * Weights: "<L>/W", "<L>/b" for L in Fea_Global_1, Fea_Global_2, Fea_Global, Fea_P1..Fea_P5, Local_Fea,
 * Local_Score, Global_Score (conv [k][k][Cin][Cout]) and Fea_P2_Deconv..Fea_P5_Deconv ([5][5][Cout][Cin]).
 * pools5 = device pointers to the VGG16 pool1..pool5 of a 352x352 input (176/88/44/22/11, NHWC).
 * prob [B,176,176,1] (required); score [B,176,176,2], local_fea [B,176,176,640], fea_global [B,1,1,128]
 * optional (NULL to skip). */
VSTAB_API int vstab_nldf_load(vstab_ctx *ctx, const vstab_tensor *tensors, int count);
VSTAB_API size_t vstab_nldf_workspace_bytes(int B);
VSTAB_API int vstab_nldf_forward(vstab_ctx *ctx, const float *const *pools5, int B, float *prob, float *score,
                                 float *local_fea, float *fea_global, void *workspace, size_t workspace_bytes,
                                 void *stream);

/* ---- autoregressive clip driver pieces (per-frame loop of evaluate_originalSize, main:535-630) ----
 * All frames are uint8 [B,h,w,3] in cv2's BGR order, device memory. */
/* cv2.resize(src, (dw, dh)), INTER_LINEAR, 8-bit (main:550,556-558): restated fixed-point algorithm, see clip_ops.hip. */
VSTAB_API int vstab_resize_u8(const uint8_t *src, int B, int sh, int sw, uint8_t *dst, int dh, int dw, void *stream);
/* np.uint8(cv2.cvtColor(cv2.resize(img_f32, (dw, dh)) * 255, COLOR_RGB2BGR)) (main:861-863: the native evaluator's history frame):
 * float bilinear resize with cv2's half-pixel centres, * 255, channels 0 and 2 swapped, truncated to 8 bits.  src [B,sh,sw,3]. */
VSTAB_API int vstab_resize_f32_to_u8(const float *src, int B, int sh, int sw, uint8_t *dst, int dh, int dw, void *stream);
/* curinput (main:550-558): feats[B,h,w,27]; slot j < 8 = history frame of lag {31,23,15,7,4,3,2,1}[j], slot 8 = current
 * frame, each u8 [B,h,w,3] at network resolution; channels swapped (COLOR_RGB2BGR) and divided by 255.  feats 16-byte aligned. */
VSTAB_API int vstab_assemble_input(const uint8_t *const *slots9, int B, int h, int w, float *feats, void *stream);
/* The same with the current frame's cv2.resize inside (main:550 + 553-558 as one launch): frame u8 [B,sh,sw,3] at its own resolution;
 * slots8[j] == NULL reads the resized current frame (the first frame of a clip, main:548-549). */
VSTAB_API int vstab_assemble_input_resized(const uint8_t *const *slots8, const uint8_t *frame, int B, int h, int w, int sh, int sw,
                                           float *feats, void *stream);
/* The evaluator's frame path on 8-bit frames in ONE launch (main:568, 497-514, 625/630): out = uint8(swap(tf_warp(swap(frame)/255,
 * glue(flow)) * 255)); the fp32 frame and the fp32 warped frame never exist.  flow [B,h,w,2] (predict_flow2), frame / out u8
 * [B,oh,ow,3] BGR, outflow [B,oh,ow,2] or NULL.  Identical bytes to vstab_frame_to_float + vstab_flow_glue_warp + vstab_quantise_output. */
VSTAB_API int vstab_flow_glue_warp_u8(const float *flow, int B, int h, int w, const uint8_t *frame, float *outflow, uint8_t *out, int oh, int ow,
                                      int net_h, int net_w, void *stream);
/* One frame of the evaluator's loop as ONE call (main:550-558 input, 568-569 network, 497-514 + 625/630 the 8-bit glue + warp, 556 the
 * history frame): vstab_assemble_input_resized(slots8, frame) -> feats [n,net_h,net_w,27]; vstab_flownets_forward(feats) -> pf6..pf2;
 * vstab_flow_glue_warp_u8(pf2, frame) -> out u8 [n,oh,ow,3] (and outflow [n,oh,ow,2] unless NULL); vstab_resize_u8(out) -> ring_slot
 * u8 [n,net_h,net_w,3], the slot later frames read this one back from.  slots8: HOST array of 8 device pointers (lags 31,23,15,7,4,3,2,1;
 * NULL = the resized current frame).  Every buffer is the caller's, allocated once; identical bytes to the four calls.  The network
 * takes 27 input channels (8 history frames + the current one).  `out`, `ring_slot` and `frame` must not overlap (the warp gathers
 * frame pixels while `out` is being written): VSTAB_E_STATE otherwise. */
VSTAB_API int vstab_clip_step(vstab_ctx *ctx, const uint8_t *const *slots8, const uint8_t *frame, int n, int net_h, int net_w, int oh, int ow,
                              float *feats, float *pf6, float *pf5, float *pf4, float *pf3, float *pf2, float *outflow, uint8_t *out,
                              uint8_t *ring_slot, void *workspace, size_t workspace_bytes, void *stream);
/* resizedInput (main:568): swap(frame)/255 -> float [npix,3]. */
VSTAB_API int vstab_frame_to_float(const uint8_t *frame, long long npix, float *out, void *stream);
/* np.uint8(swap(warped*255)) (main:625,630,556): float [npix,3] -> u8, truncating, saturating outside [0,255]. */
VSTAB_API int vstab_quantise_output(const float *warped, long long npix, uint8_t *out, void *stream);

/* ---- flow post-filters of the reference's other evaluators ---------------------
 * k x k box blur of a flow field with zero (SAME) padding, weights 1/(k*k), k odd
 * (main_flownetS_pyramid.py:634-641, k = 75).  tmp: scratch of the same size as flow. */
VSTAB_API int vstab_flow_box_blur(const float *flow, int B, int h, int w, int k, float *tmp, float *out, void *stream);
/* out = a*x + b*y elementwise (0.9*smooth + 0.1*prev, :643; prev = 0.9*prev + 0.1*cur, :695). */
VSTAB_API int vstab_axpby(const float *x, float a, const float *y, float b, float *out, long long n, void *stream);
/* ---- training objective, first slice of SURVEY.md 8f rank 4 -------------------------------------------
 * One pyramid level of the reference's loss (main:188-210, 269-273) and its gradient w.r.t. the flow:
 *   P = tf_warp(unstab, pf), M = tf_warp(ones, pf); masked_MSE = mean_b[ sum (P*M - gt*M)^2 / sum safe(M) ];
 *   TV = sum_b tf.image.total_variation(pf[b]).
 * gt / unstab: [B,h,w,3] already resized to the flow's size (vstab_resize_bilinear = tf.image.resize_images).
 * sums: device double[3*B], overwritten with {sum sq, sum safe(M), TV} per sample -- the level's loss is
 *   scale_mse * mean_b(sums[3b]/sums[3b+1]) + scale_tv * sum_b sums[3b+2]   (the caller adds the levels up).
 * grad_pf (may be NULL): [B,h,w,2] d(that loss)/d(pf), overwritten. */
VSTAB_API int vstab_loss_level(const float *pf, const float *gt, const float *unstab, int B, int h, int w, double *sums,
                               float scale_mse, float scale_tv, float *grad_pf, void *stream);

/* loss_main (main:213-217, 269-275) over all pyramid levels in one call: per level the two tf.image.resize_images of the
 * full-size stable / unstable frames [B,H,W,3] to the flow's size (main:202-203), lossterm + the TV term, and (grad != NULL)
 * d loss_main / d flow written into channels 0..1 of the caller's [B,h,w,cs_grad] buffer.  pf: [B,h,w,cs_pf] pixels whose
 * channels 0..1 are the flow; channel strides even.  loss_out: one device double = sum over levels of
 * mean_b(masked_MSE) + tv_weight * TV. */
typedef struct vstab_loss_level_desc {
    const float *pf;
    float *grad;
    int h, w, cs_pf, cs_grad;
    float tv_weight;
} vstab_loss_level_desc;
VSTAB_API size_t vstab_loss_main_workspace_bytes(const vstab_loss_level_desc *levels, int n_levels, int B);
VSTAB_API int vstab_loss_main(const vstab_loss_level_desc *levels, int n_levels, const float *gtstab, const float *unstab, int B, int H,
                              int W, double *loss_out, void *workspace, size_t workspace_bytes, void *stream);

/* Weight and bias gradient of PadLayer(pad) -> Conv2d(k, stride, VALID) (model.py:807-844; what tf.gradients yields for
 * the filter of tf.nn.conv2d): dW[ky,kx,ci,co] = sum_{n,oy,ox} x[n, s*oy+ky-pad, s*ox+kx-pad, ci] * gout[n,oy,ox,co] in the
 * reference's HWIO layout, db[co] = sum gout (db may be NULL).  x: [B,Hi,Wi,cs_x] using channels cx_off..cx_off+cin;
 * gout: [B,Ho,Wo,cs_g] using channels cg_off..cg_off+cout; all channel counts / strides / offsets multiples of 4.
 * accumulate != 0 adds to dW / db instead of overwriting.  With x = the output gradient and gout = the input of a
 * 4x4 stride-2 SAME transposed conv (k 4, stride 2, pad 1) the result is that layer's [4,4,cout,cin] filter gradient.
 * The first call for a geometry builds a 16 B-per-output-pixel table on the device (and synchronises the stream once); it is
 * kept for the life of the process. */
VSTAB_API size_t vstab_conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int k, int cin, int cout);
VSTAB_API int vstab_conv_wgrad(const float *x, int B, int Hi, int Wi, int cs_x, int cx_off, int cin, const float *gout, int Ho,
                               int Wo, int cs_g, int cg_off, int cout, int k, int stride, int pad, float *dW, float *db,
                               int accumulate, void *workspace, size_t workspace_bytes, void *stream);

/* Input gradient of the same layer (tf.gradients w.r.t. the conv's input): dx = transposed conv of gout with
 * W [k,k,cin,cout] (DEVICE pointer, the reference's HWIO layout), stride 1 or 2.  Runs on the forward MFMA kernel:
 * stride 1 = conv over gout with the kernel flipped, stride 2 = four output-parity phases of ceil(k/2)^2 taps; the
 * packed operand is gathered on the device every call (index table cached per geometry).  accumulate != 0: dx += result.
 * With W = a 4x4 stride-2 transposed conv's [4,4,cout,cin] filter, gout = its input and (k,stride,pad) = (4,2,1) the
 * same call computes that layer's FORWARD output. */
VSTAB_API size_t vstab_conv_dgrad_workspace_bytes(int B, int Ho, int Wo, int cs_g, int cout, int k, int stride, int pad, int Hi, int Wi,
                                                  int cs_x, int cx_off, int cin, int accumulate);
VSTAB_API int vstab_conv_dgrad(const float *gout, int B, int Ho, int Wo, int cs_g, int cg_off, int cout, const float *W,
                               const float *bias /* device [cin] added to every output pixel, or NULL */, int k, int stride, int pad,
                               float *dx, int Hi, int Wi, int cs_x, int cx_off, int cin, int accumulate, void *workspace,
                               size_t workspace_bytes, void *stream);
/* PadLayer(pad) -> Conv2d(k, stride, VALID) + bias with DEVICE-resident raw weights W [k,k,cin,cout] (training: the weights
 * change every step, so the MFMA operand is gathered on the device from an index table cached per geometry).
 * act: 0 none, 1 leaky relu 0.1, 2 relu, 3 none and ADD to what is in y. */
/* Ho, Wo: the output size; 0, 0 = floor((Hi + 2 pad - k)/stride) + 1, or that + 1 (TF SAME on an odd size: the last window
 * sticks out of the image and reads zeros there). */
VSTAB_API size_t vstab_conv_forward_workspace_bytes(int B, int Hi, int Wi, int cs_x, int cin, int k, int stride, int pad, int cout, int cs_y,
                                                    int cy_off, int act, int Ho, int Wo);
VSTAB_API int vstab_conv_forward(const float *x, int B, int Hi, int Wi, int cs_x, int cx_off, int cin, const float *W, const float *bias,
                                 int k, int stride, int pad, float *y, int Ho, int Wo, int cs_y, int cy_off, int cout, int act,
                                 void *workspace, size_t workspace_bytes, void *stream);
/* PadLayer(pad) -> Conv2d(k, stride, VALID) + bias for a FEW-channel input (the network's first layer, model.py:807-808: 27 -> 64, 7x7
 * stride 2) on the inference path's row-window kernel, with DEVICE-resident raw weights Wf [k,k,cs_w,cout] of which input channels
 * 0..Cin-1 are used (cs_w >= Cin: the training buffers pad 27 to 28).  x [B,H,W,Cin] contiguous; cout <= 64; act 0 none, 1 leaky relu,
 * 2 relu; VSTAB_E_SHAPE when the kernel does not take the geometry (the caller then uses vstab_conv_forward). */
VSTAB_API size_t vstab_conv_rowwin_forward_workspace_bytes(int B, int H, int W, int Cin, int cs_w, int cout, int k, int stride, int pad, int cs_y,
                                                           int cy_off, int act);
VSTAB_API int vstab_conv_rowwin_forward(const float *x, int B, int H, int W, int Cin, const float *Wf, int cs_w, int cout, const float *bias, int k,
                                        int stride, int pad, float *y, int cs_y, int cy_off, int act, void *workspace, size_t workspace_bytes,
                                        void *stream);
/* The same for a 3x3 stride-1 pad-1 layer in Winograd F(2x2,3x3) form (4/9 of the multiply-adds): transpose = 0 is the forward
 * convolution x [.., cin] -> y [.., cout]; transpose = 1 the input gradient (x = output gradient [.., cout] -> y = dx [.., cin], kernel
 * flipped, channel roles swapped).  W: DEVICE [3,3,cin,cout]; the Winograd-domain operand is rebuilt on the device every call.
 * bias [N] or NULL; act 0 none, 1 leaky relu, 3 ADD to what is in y.  Reduction channels % 32 == 0, output channels % 64 == 0. */
VSTAB_API size_t vstab_conv3x3_winograd_workspace_bytes(int B, int H, int W, int cin, int cout, int transpose);
VSTAB_API int vstab_conv3x3_winograd(const float *x, int B, int H, int W, int cs_x, int cx_off, const float *Wf, int cin, int cout, int transpose,
                                     const float *bias, float *y, int cs_y, int cy_off, int act, void *workspace, size_t workspace_bytes,
                                     void *stream);
/* Filter gradient of the same 3x3 stride-1 pad-1 layer in the Winograd domain: V = B^T d B of the input tiles, dM = A dY A^T of the
 * output-gradient tiles, the 16 position gradients dU_xi = V_xi^T dM_xi as ONE batched reduction on the weight-gradient MFMA
 * kernel (4/9 of the direct filter gradient's multiply-adds), dW = G^T dU G.  dW: HWIO [3][3][cin][cout], overwritten.
 * x: [B,H,W,cs_x] channels cx_off..+cin; gout: [B,H,W,cs_g] channels cg_off..+cout; channel counts multiples of 4. */
VSTAB_API size_t vstab_conv3x3_winograd_wgrad_workspace_bytes(int B, int H, int W, int cin, int cout);
VSTAB_API int vstab_conv3x3_winograd_wgrad(const float *x, int B, int H, int W, int cs_x, int cx_off, int cin, const float *gout, int cs_g,
                                           int cg_off, int cout, float *dW, void *workspace, size_t workspace_bytes, void *stream);
/* din (+)= gain * (adjoint of tf.image.resize_images(., [oh,ow]))(dout): backward of the legacy bilinear resize. */
VSTAB_API int vstab_resize_bilinear_backward(const float *dout, int B, int oh, int ow, int C, float *din, int h, int w, float gain,
                                             int accumulate, void *stream);
/* PadLayer(1) -> nearest-neighbour resize (align_corners=True) to H x W (model.py:795-802, 882-884), and its adjoint. C % 4 == 0. */
VSTAB_API int vstab_pad_nearest_upsample(const float *src, int B, int h2, int w2, int C, float *out, int H, int W, void *stream);
VSTAB_API int vstab_pad_nearest_upsample_backward(const float *dout, int B, int H, int W, int C, float *dsrc, int h2, int w2, int accumulate,
                                                  void *stream);
/* The full-resolution head (model.py:882-887) through its tap table, as the inference path computes it: T [B,h2,w2,32] with
 * T[s][tap*2+o] = sum_c concat2[s][c] W[tap][c][o] (a 1x1 conv, vstab_conv_forward); pf2 [B,H-2,W-2,2] = bias + the nine
 * gathered taps + eight adds of the upsampled pf3 [B,h3,w3,2].  The backward of the gather: dT from the flow gradient
 * g [B,H-2,W-2,cs_g] (channels 0,1), columns 18..31 zeroed. */
VSTAB_API int vstab_pf2_from_taps(const float *T, int B, int h2, int w2, const float *bias2, const float *pf3, int h3, int w3, float *pf2,
                                  int H, int W, void *stream);
VSTAB_API int vstab_pf2_taps_backward(const float *g, int cs_g, int B, int H, int W, float *dT, int h2, int w2, void *stream);
/* The tap table itself as vstab_flownets_forward computes it (csrc/tap_panel.hip; model.py:882-885's 3x3 head applied per SOURCE pixel):
 * T [M,32] = concat2 [M,196] x table, M = B*h2*w2 pixel rows of 196 floats (194 channels + 2 of padding), table [200][32] on the device with
 * table[c][tap*2+o] = W[tap][c][o] for c < 194, zero elsewhere (rows 194..199 and columns 18..31).  concat2 and T 16-byte aligned,
 * M*784 < 2^31. */
VSTAB_API int vstab_predict2_tap_table(const float *concat2, long long M, const float *table, float *T, void *stream);
/* out[c] (+)= sum over the rows of g[row*cs + c_off + c] (bias gradients), deterministic two-stage reduction. */
VSTAB_API size_t vstab_column_sum_scratch_bytes(long long rows, int C);
VSTAB_API int vstab_column_sum(const float *g, long long rows, int cs, int c_off, int C, float *out, int accumulate, void *scratch,
                               size_t scratch_bytes, void *stream);
/* tf.train.AdamOptimizer update (main:333-335) on a flat tensor; lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) is the caller's. */
VSTAB_API int vstab_adam_step(float *w, const float *g, float *m, float *v, long long n, float lr_t, float beta1, float beta2, float eps,
                              void *stream);

/* BatchNormLayer(act=lrelu 0.1, is_train=True, gamma_init=None) (model.py:809...) IN PLACE on channels c_off..c_off+C of
 * an NHWC tensor viewed as [rows, cs]: batch mean / population variance (tf.nn.moments), y = lrelu((z-mean)*rsqrt(var+eps)
 * + beta), moving = moving*decay + batch*(1-decay) (moving_* may be NULL).  save_mean / save_rstd [C] feed the backward. */
VSTAB_API size_t vstab_bn_scratch_bytes(long long rows, int C);
VSTAB_API int vstab_bn_lrelu_train_forward(float *zy, long long rows, int cs, int c_off, int C, const float *beta, float *moving_mean,
                                           float *moving_var, float decay, float eps, float *save_mean, float *save_rstd, void *scratch,
                                           size_t scratch_bytes, void *stream);
/* Its backward: dy (gradient w.r.t. y, channels cg_off..+C of a [rows, cs_g] tensor) is overwritten by the gradient w.r.t.
 * the layer's input z; dbeta [C] = sum dy*lrelu'(y) (may be NULL).  y: the forward's output (xhat is recovered from it). */
VSTAB_API int vstab_bn_lrelu_train_backward(const float *y, int cs_y, int cy_off, float *dy, int cs_g, int cg_off, int C, long long rows,
                                            const float *beta, const float *save_rstd, float *dbeta, int accumulate, void *scratch,
                                            size_t scratch_bytes, void *stream);
/* dy *= (y > 0 ? 1 : 0.1): leaky-relu backward for layers without BatchNorm. */
VSTAB_API int vstab_lrelu_backward(const float *y, int cs_y, int cy_off, float *dy, int cs_g, int cg_off, int C, long long rows, void *stream);

/* scipy.signal.medfilt(np.squeeze(of), k) (evaluate_medianNma, main_flownetS_pyramid.py:809): order filter over a
 * kh x kw x kc window of each [h,w,2] field -- kc spans the channel axis; the reference's scalar 5 means 5x5x5 --
 * zero padded on all axes, output = element n/2 of the sorted window.  Odd sizes, kh,kw <= 31, kc <= 5; out != flow. */
VSTAB_API int vstab_flow_medfilt(const float *flow, int B, int h, int w, int kh, int kw, int kc, float *out, void *stream);
/* out[b,:,:,c] = mean over the image of flow[b,:,:,c] (main_flownetS_pyramid_highTV_noBBloss.py:629). */
VSTAB_API int vstab_flow_mean_fill(const float *flow, int B, int h, int w, float *out, void *stream);

/* Homography evaluator (main:728-743, main = main_flownetS_pyramid_noprevloss_dataloader.py): replaces
 *   h, mask = cv2.findHomography(gridmesh, gridmesh - flow, cv2.RANSAC)        (main:735)
 * on the dense [B,H,W,2] flow with a deterministic on-device RANSAC: K (<= 512) 4-point hypotheses per sample drawn
 * by a counter hash of `seed`, consensus = pixels (every `stride`-th) whose reprojection error is <= thresh (cv2's
 * default is 3.0), then `refine` (1..16) least-squares refits on the inlier set.  Hout[b][9] = row-major src->dst
 * matrix with h33 = 1 (float64, device), inliers[b] = size of the last consensus set (0: nothing fitted, Hout NaN).
 * cv2's own RNG cannot be reproduced, so the hypotheses differ from cv2's; the estimator is the same. */
VSTAB_API size_t vstab_homography_workspace_bytes(int B, int H, int W, int K);
VSTAB_API int vstab_homography_fit(const float *flow, int B, int H, int W, int K, unsigned seed, double thresh, int refine,
                                   int stride, double *Hout, int32_t *inliers, void *workspace, size_t workspace_bytes,
                                   void *stream);
/* cv2.warpPerspective(src, Hm[b], (ow, oh)) with default flags on uint8 [B,sh,sw,3] frames (main:736): INTER_LINEAR on
 * coordinates rounded to 1/32 px, 15-bit fixed-point blend, constant-0 border; Hm maps src -> dst (inverted inside). */
VSTAB_API int vstab_warp_perspective_u8(const uint8_t *src, int B, int sh, int sw, const double *Hm, uint8_t *dst, int oh, int ow,
                                        void *stream);

/* ==================================================================================================================
 * DIAGNOSTIC AND TEST SURFACE -- not part of the drop-in contract.  Nothing below is needed to run the path; these calls
 * exist for A/B measurements, profiles and tests, they hold context-global (plan flags, per-launch events) or PROCESS-global
 * (roctx ranges, the HBM-side profiler) state, and the plan flags change the ORDER of floating-point sums (results stay
 * within the parity tolerance but are not bit-identical across flag values).  A product caller leaves all of them alone.
 * ================================================================================================================== */

/* ---- plan flags (A/B measurements and the bit-equality tests; 3 = the round-3 schedule): VSTAB_PLAN_NO_SKINNY keeps few-row
 * layers on the tiled kernel with a split-K combine launch, VSTAB_PLAN_NO_DUAL launches a refinement level's flow head and
 * transposed convolution one after the other, VSTAB_PLAN_NO_TAIL keeps vstab_stabilise_originalsize's last two launches apart (it changes
 * no sum: A/B of the fused tail).  Context state; workspace sizes then come from vstab_workspace_bytes_ctx. */
#define VSTAB_PLAN_NO_SKINNY 1u
#define VSTAB_PLAN_NO_DUAL 2u      /* a refinement level as four launches (tap table, predict_up, transposed conv, combine) instead of two */
#define VSTAB_PLAN_NO_TAIL 4u      /* vstab_stabilise_originalsize: predict_flow2's gather and the glue + warp as two launches (bit-identical) */
#define VSTAB_PLAN_NO_WDEC 8u      /* transposed convolutions as four direct sub-pixel phases, never in Winograd F(2x2,2x2) form (round 6) */
#define VSTAB_PLAN_FORCE_WDEC 16u  /* deconv4 / deconv3 in Winograd F(2x2,2x2) form whatever their size (tests: small shapes) */
VSTAB_API int vstab_set_plan_flags(vstab_ctx *ctx, unsigned flags);

/* ---- measurement support.  With profiling enabled every conv-like launch of
 * vstab_flownets_forward (15 per forward: encoder stages 1..6_1, deconv5..2, predict2 tap
 * table; the GEMM kernel itself, not the split-K combine that may follow it) is bracketed by
 * hipEvents recorded on the forward's stream.  vstab_profile_read must be called after that stream has been
 * synchronised: it returns, both summed over the forward passes recorded since the last
 * reset (a batch split into chunks records one pass per chunk; *n_forwards counts them), the
 * elapsed milliseconds per launch slot and the ALGORITHMIC flops per slot (2*MAC of the layer
 * as SURVEY.md 8d counts it; deconvs at 4 taps/output). */
VSTAB_API int vstab_profile_enable(vstab_ctx *ctx, int enable);
VSTAB_API int vstab_profile_reset(vstab_ctx *ctx);
VSTAB_API int vstab_profile_read(vstab_ctx *ctx, double *ms_sum15, double *flops15, int *n_forwards);
/* vstab_profile_read's flops are what the launches ISSUE: the 3x3 stride-1 encoder stages run in Winograd F(2x2,3x3) form and
 * issue 4/9 of the direct convolution's multiply-adds.  This returns the same slots counted as direct convolutions. */
VSTAB_API int vstab_profile_read_direct(vstab_ctx *ctx, double *flops15);
/* Name (as rocprofv3 prints it) of the kernel instantiation launch slot `slot` used in the last forward. */
VSTAB_API int vstab_profile_kernel_name(vstab_ctx *ctx, int slot, char *buf, int cap);

/* ---- device self-test of the glue's division by a launch constant (five fused operations on a host-side reciprocal instead
 * of a run-time IEEE division; quotients that do not come out normal -- signed zeros, denormals, infinities -- take the division
 * itself): compares it BIT FOR BIT with `x / d` for the `count` fp32 bit patterns x starting at `first_bits` (NaN numerators
 * skipped), and ADDS the number of mismatches to *bad_count_dev (device memory, 8-byte aligned).
 * VSTAB_E_SHAPE for a divisor the glue would divide plainly (outside [1, 2^24], or an all-ones significand). */
VSTAB_API int vstab_selftest_div_const(float d, unsigned first_bits, unsigned long long count, unsigned long long *bad_count_dev,
                                       void *stream);

/* ---- roctx ranges: with on = 1 every layer of vstab_flownets_forward (conv1 .. conv6_1, predict_flowN+upsample, deconvN,
 * predict_flow2) and the glue/warp launches run inside a named roctx range, so a `rocprofv3 --marker-trace --kernel-trace`
 * timeline reads as the network (model.py:807-887).  The roctx library is loaded with dlopen on first use; VSTAB_E_STATE if
 * none is installed.  Process-wide, off by default. */
VSTAB_API int vstab_trace_ranges(int on);

/* ---- instrumentation of the HBM-side kernels (tf_warp, the flow glue, the fused launch): with profiling on every such launch
 * is bracketed by dispatch-timestamp events on its own stream.  Process-wide; switching it on clears earlier records.
 * Slots: 0 = vstab_warp_flow, 1 = vstab_flow_resize_scale, 2 = vstab_flow_glue_warp, 3 = the predict_flow2 gather inside
 * vstab_flownets_forward / vstab_pf2_from_taps (128 B per tap-table row + 8 B per coarser-flow and output pixel).  Read after synchronising the stream(s):
 * summed kernel milliseconds, number of launches, summed ALGORITHMIC bytes (warp 32 B/px; glue 8 B per source +
 * 8 B per output pixel; fused 8 B per source pixel + 32 (flow written) or 24 B per output pixel). */
VSTAB_API int vstab_hbm_profile_enable(int mode);   /* 0 = off (records kept), 1 = clear + on, 2 = on again (records kept) */
VSTAB_API int vstab_hbm_profile_read(int slot, double *ms_sum, int *launches, double *alg_bytes_sum);

/* ---- host-only helpers (no GPU needed; used by the CPU tests) --------------------- */
/* Level sizes of the encoder for an HxW input: hw[2*i], hw[2*i+1] = (h, w) of stage i
 * (10 stages).  Returns 0 or VSTAB_E_SHAPE. */
/* host-only: the XCD-aware workgroup -> tile map the MFMA kernels apply (dispatch-order id `lin` of a gx x gy x gz grid ->
 * tile coordinates xyz[3]); exported so that its bijectivity and banding can be tested without a GPU. */
VSTAB_API int vstab_host_xcd_remap(int gx, int gy, int gz, int lin, int32_t *xyz);
VSTAB_API int vstab_level_sizes(int H, int W, int32_t *hw20);

/* Host-side view of one conv-like launch of the forward schedule (layer 0-9 = encoder
 * stages 1..6_1, 10-13 = deconv5..2, 14 = predict2 tap table, 15-18 = predict6..3 tap tables):
 * writes 26 + 7*nphase ints
 *   B Hi Wi Cs_in KH NSEG SEG SEGP SEG_STRIDE s_in s_out Ho Wo Cs_out c_off N Npad act
 *   nphase ksplit Mmax tile vec4 in_buf out_buf reserved, then per phase
 *   Hg Wg M off_y off_x o_y o_x
 * (buffers index vstab_workspace_layout entries, -1 = feats).  Returns ints written. */
VSTAB_API int vstab_host_layer_plan(int B, int H, int W, int Cin, int layer, int32_t *out, int cap);
/* the same under a pinned plan batch / plan flags (vstab_set_plan_batch, vstab_set_plan_flags); `reserved` = 1 when the stage runs in
 * Winograd form; tile 6 = the weight-stream kernel of few-row layers (conv_skinny.hip: 32- or 64-row x 32-column workgroups) */
VSTAB_API int vstab_host_layer_plan_pinned(int plan_batch, unsigned flags, int B, int H, int W, int Cin, int layer, int32_t *out, int cap);

/* Host-side weight packing exactly as vstab_load_weights does it for `layer` of a
 * Cin-channel network: W is the reference-layout tensor, scale (may be NULL = ones) the
 * folded BatchNorm scale per output channel.  Writes the packed floats ([phase][KT][Npad][32])
 * to `wpk` (capacity `cap` floats) and returns the number of floats, or a negative error. */
VSTAB_API long long vstab_host_pack_layer(int Cin, int layer, const float *W, const double *scale,
                                          float *wpk, long long cap);

/* The transposed convolution of refinement level l (0..3 = deconv5..deconv2) in Winograd F(2x2,2x2) form (csrc/winograd_ops.hip):
 * the 9-position GEMM's plan in vstab_host_layer_plan's format (26 ints + 7 per position; `reserved` = 1 when the default plan of this
 * problem chooses the form) followed by the tile geometry {NTy, NTx, nty[3], ntx[3]}; and its packed operands as vstab_load_weights
 * builds them (W = the reference-layout filter [4][4][Cout][Cin]; 9 positions x [KT][4 Cout][32]). */
VSTAB_API int vstab_host_wdec_plan(int B, int H, int W, int Cin, int l, int32_t *out, int cap);
VSTAB_API long long vstab_host_pack_wdec(int l, const float *W, const double *scale, float *wpk, long long cap);

#ifdef __cplusplus
}
#endif
#endif /* VSTAB_H */
