// Shared between the translation units that implement the C ABI (api.cpp, nldf_api.cpp): the
// context object, error plumbing and the helpers that turn a plain convolution into a launch plan.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/vstab.h"
#include "vstab_internal.h"

struct vstab_ctx {
    int device = 0;
    bool loaded = false;
    int cin = 0;
    std::string err;                     // last failure message; written under err_mu (failure paths only)
    std::mutex err_mu;
    float *dev_weights = nullptr;        // one allocation holding every packed tensor
    size_t dev_weight_floats = 0;
    // float offsets into dev_weights
    size_t enc_w[10], enc_b[10];
    size_t enc0_rw = 0;                  // layer-1 weights in the row-window layout (conv_rowwin.hip)
    size_t zero_b = 0;                   // 1024 zeros
    size_t wino_w[10] = {0};             // Winograd-domain operands of the 3x3 stride-1 stages (winograd_ops.hip)
    size_t dec_w[4], dec_b[4];
    size_t wdec_w[4] = {0};              // Winograd F(2x2,2x2)-domain operands of deconv4 / deconv3 (winograd_ops.hip)
    size_t pred_w[4], pred_b[4];         // predict6,5,4,3
    size_t tab_wp, tab_b, pred2_b;       // predict2 tap table as plain [200][32] rows (tap_panel.hip); the zero bias the tap-table GEMMs of predict6..3 share; predict2's bias
    vstab::UpflowW up[4];
    int plan_batch = 0;                  // vstab_set_plan_batch: > 0 pins every arithmetic-changing plan decision to that batch's
    unsigned plan_flags = 0;             // vstab_set_plan_flags (VSTAB_PLAN_*)
    // profiling (vstab_profile_*): event pairs per conv-like launch, one row per forward
    bool prof = false;
    std::vector<hipEvent_t> prof_ev;     // [forward][15][2]
    int prof_forwards = 0;
    double prof_flops[15] = {0};         // flops the launches ISSUE (Winograd-form stages: 4/9 of the direct convolution)
    double prof_flops_direct[15] = {0};  // the same layers counted as direct convolutions
    const char *prof_kernel[15] = {nullptr};   // kernel instantiation each slot launched last (string literals; written only while profiling)
    // VGG16 trunk (vstab_vgg16_*)
    void *nldf = nullptr;                // NLDF head state (nldf_api.cpp)
    bool vgg_loaded = false;
    float *vgg_weights = nullptr;
    size_t vgg_w[13], vgg_b[13];
    size_t vgg_wino_w[13] = {0};         // Winograd-domain operands of conv3_2 .. conv5_3
    size_t vgg_raw0 = 0;                 // conv1_1's filter as given (HWIO), for conv3x3_rgb_kernel
    size_t vgg_zero = 0;                 // 1024 zeros (bias of the Winograd-domain GEMM)
};

int fail(vstab_ctx *ctx, int code, const char *fmt, ...);
#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(ctx, VSTAB_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)


void vstab_nldf_free(void *nldf);     // nldf_api.cpp

// ---- roctx ranges (rocprofv3 --marker-trace reads as the network).  Off by default; vstab_trace_ranges(1)
// resolves roctxRangePushA / roctxRangePop from librocprofiler-sdk-roctx.so (or libroctx64.so) with dlopen -- no link-time
// dependency -- and every layer of the forward schedules then runs inside a named range.
bool trace_ranges_enable(bool on);      // false: no roctx library could be loaded
struct TraceRange {
    explicit TraceRange(const char *name);
    ~TraceRange();
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
    bool active;
};

// ---- helpers defined in api.cpp
void choose_split(vstab::ConvParams &p, int BN, int BM = 128);
vstab::ConvTile choose_tile_split(vstab::ConvParams &p, vstab::ConvTile tile, bool vec4);
void set_layout(vstab::ConvParams &p, const vstab::KLayout &L);
void set_ranges(vstab::ConvParams &p);
const vstab_tensor *find(const vstab_tensor *t, int n, const std::string &name);
bool shape_is(const vstab_tensor *t, std::initializer_list<int> s);
// plain conv (k x k, stride, zero pad) on an NHWC tensor whose pixel stride is cs_in >= cin (run mode
// when cs_in == cin, tap mode otherwise); output slice [c_off, c_off+cout) of a cs_out-wide pixel
bool fill_plain_conv(vstab::ConvParams &p, vstab::ConvTile &tile, bool &vec4, int B, int Hi, int Wi, int cin, int cs_in, int k,
                     int stride, int pad, int cout, int cs_out, int c_off, int act);
