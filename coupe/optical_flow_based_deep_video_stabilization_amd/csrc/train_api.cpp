// C ABI of the training-side building blocks: convolution weight / bias gradients.
// The loss entry point lives in api.cpp next to the other small wrappers.
#include <cstdint>

#include "../../../include/vstab.h"
#include "api_internal.h"
#include "vstab_internal.h"

using namespace vstab;

extern "C" size_t vstab_conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int k, int cin, int cout)
{
    if (B < 1 || Ho < 1 || Wo < 1 || k < 1 || cin < 1 || cout < 1) return 0;
    WgradParams p{};
    p.M = k * k * cin; p.Cout = cout; p.K = B * Ho * Wo;
    const int ks = wgrad_choose_split(p);
    const size_t ptab = ((size_t)p.K * sizeof(int4) + 255) / 256 * 256;
    const size_t slabs = ((ks > 1 ? (size_t)ks * p.M * cout * sizeof(float) : 0) + 255) / 256 * 256;
    return ptab + slabs + (size_t)column_sum_chunks(p.K) * cout * sizeof(float) + 256;
}

extern "C" int vstab_conv_wgrad(const float *x, int B, int Hi, int Wi, int cs_x, int cx_off, int cin, const float *gout, int Ho,
                                int Wo, int cs_g, int cg_off, int cout, int k, int stride, int pad, float *dW, float *db,
                                int accumulate, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !gout || !dW || !workspace) return fail(nullptr, VSTAB_E_STATE, "conv_wgrad: NULL buffer");
    if (B < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || k < 1 || k > 7 || stride < 1 || pad < 0 || cin < 1 || cout < 1 ||
        cx_off < 0 || cg_off < 0 || cx_off + cin > cs_x || cg_off + cout > cs_g)
        return fail(nullptr, VSTAB_E_SHAPE, "conv_wgrad: bad shape");
    if ((Hi + 2 * pad - k) / stride + 1 != Ho || (Wi + 2 * pad - k) / stride + 1 != Wo)
        return fail(nullptr, VSTAB_E_SHAPE, "conv_wgrad: %dx%d input, k %d stride %d pad %d does not give a %dx%d output", Hi, Wi, k,
                    stride, pad, Ho, Wo);
    if ((cin & 3) || (cs_x & 3) || (cx_off & 3) || (cs_g & 3) || (cg_off & 3))
        return fail(nullptr, VSTAB_E_ALIGN, "conv_wgrad: channel counts, strides and offsets must be multiples of 4");
    if ((long long)B * Hi * Wi * cs_x * 4 >= 0x80000000LL || (long long)B * Ho * Wo * cs_g * 4 >= 0x80000000LL)
        return fail(nullptr, VSTAB_E_SHAPE, "conv_wgrad: tensors must stay below 2 GiB");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(gout) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(nullptr, VSTAB_E_ALIGN, "conv_wgrad: x / gout must be 16-byte, workspace 256-byte aligned");
    const size_t need = vstab_conv_wgrad_workspace_bytes(B, Ho, Wo, k, cin, cout);
    if (workspace_bytes < need) return fail(nullptr, VSTAB_E_NOMEM, "conv_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    WgradParams p{};
    p.x = x; p.g = gout; p.dW = dW;
    p.x_bytes = (unsigned)((long long)B * Hi * Wi * cs_x * 4); p.g_bytes = (unsigned)((long long)B * Ho * Wo * cs_g * 4);
    p.Hi = Hi; p.Wi = Wi; p.Cs_x = cs_x; p.cx_off = cx_off; p.Cin = cin; p.KH = k; p.KW = k;
    p.Cs_g = cs_g; p.cg_off = cg_off; p.Cout = cout; p.M = k * k * cin; p.K = B * Ho * Wo;
    p.accumulate = accumulate ? 1 : 0;
    p.ksplit = wgrad_choose_split(p);
    int4 *ptab = reinterpret_cast<int4 *>(workspace);
    p.ptab = ptab;
    p.partial = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + ((size_t)p.K * sizeof(int4) + 255) / 256 * 256);
    HIP_TRY(nullptr, launch_wgrad_pixel_table(B, Hi, Wi, cs_x, Ho, Wo, stride, pad, ptab, st));
    HIP_TRY(nullptr, launch_wgrad(p, st));
    if (db) {
        const size_t slabs = ((p.ksplit > 1 ? (size_t)p.ksplit * p.M * cout * sizeof(float) : 0) + 255) / 256 * 256;
        float *scratch = reinterpret_cast<float *>(reinterpret_cast<char *>(p.partial) + slabs);
        HIP_TRY(nullptr, launch_column_sum(gout, (long long)p.K, cs_g, cg_off, cout, db, accumulate ? 1 : 0, scratch, st));
    }
    return VSTAB_OK;
}
