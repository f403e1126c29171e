// C ABI of the NLDF saliency head (NLDF.py:24-134).  The reference never
// instantiates this model (no `NLDF.Model(` call anywhere) and the file is Python-2 only, so this is
// built last and only at the reference's hard-wired geometry: pool1..pool5 of a 352x352 input
// (176, 88, 44, 22, 11), outputs at 176x176 (NLDF.py:58-64, 74).
#include "api_internal.h"

using namespace vstab;

namespace {

constexpr int FEA = 128;                      // fea_dim, NLDF.py:33
const int POOL_C[5] = {64, 128, 256, 512, 512};        // pool1..pool5 channels
const int POOL_HW[5] = {176, 88, 44, 22, 11};
// concat buffers: cat_k = [Fea_Pk | Fea_Pk_LC | Fea_P(k+1)_Up]  (NLDF.py:57-66)
const int CAT_C[5] = {768, 640, 512, 384, 256};        // cat1..cat5
const int UP_C[4] = {512, 384, 256, 128};              // Fea_P2_Up .. Fea_P5_Up channels (into cat1..cat4)

struct NldfWeights {            // float offsets into ctx->nldf_weights
    size_t g1_w, g1_b, g2_w, g2_b, g3_w, g3_b;
    size_t p_w[5], p_b[5];      // Fea_P1..P5
    size_t d_w[4], d_b[4];      // Fea_P2_Deconv .. Fea_P5_Deconv (index 0 = P2)
    size_t lf_w, lf_b, ls_w, ls_b, gs_w, gs_b;
};

enum NBuf { N_G1, N_G2, N_G, N_CAT1, N_CAT2, N_CAT3, N_CAT4, N_CAT5, N_LF, N_LS, N_GS, N_PART, N_NBUF };

struct NldfPlan { size_t off[N_NBUF], bytes[N_NBUF], total; };

size_t fl(int B, int hw, int c) { return (size_t)B * hw * hw * c; }

bool nldf_plan(int B, NldfPlan &pl)
{
    if (B < 1 || B > 64) return false;
    size_t n[N_NBUF] = {fl(B, 7, FEA), fl(B, 3, FEA), fl(B, 1, FEA), fl(B, 176, 768), fl(B, 88, 640), fl(B, 44, 512),
                        fl(B, 22, 384), fl(B, 11, 256), fl(B, 176, 640), fl(B, 176, 2), fl(B, 1, 2), (size_t)32 << 20};
    size_t off = 0;
    for (int i = 0; i < N_NBUF; ++i) {
        if (n[i] * 4 >= 0x80000000ull) return false;
        pl.off[i] = off; pl.bytes[i] = n[i] * 4;
        off += (pl.bytes[i] + 255) / 256 * 256;
    }
    pl.total = off;
    return true;
}

}  // namespace

struct vstab_nldf {             // lives inside the context (opaque to callers)
    bool loaded = false;
    float *w = nullptr;
    NldfWeights o;
};

static vstab_nldf *nldf_of(vstab_ctx *ctx)
{
    if (!ctx->nldf) ctx->nldf = new (std::nothrow) vstab_nldf();
    return static_cast<vstab_nldf *>(ctx->nldf);
}

void vstab_nldf_free(void *p)
{
    vstab_nldf *n = static_cast<vstab_nldf *>(p);
    if (!n) return;
    if (n->w) (void)hipFree(n->w);
    delete n;
}

extern "C" size_t vstab_nldf_workspace_bytes(int B)
{
    NldfPlan pl;
    if (!nldf_plan(B, pl)) { fail(nullptr, VSTAB_E_SHAPE, "nldf: unsupported batch %d", B); return 0; }
    return pl.total;
}

extern "C" int vstab_nldf_load(vstab_ctx *ctx, const vstab_tensor *t, int count)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "nldf_load: ctx is NULL");
    if (!t || count <= 0) return fail(ctx, VSTAB_E_WEIGHTS, "nldf_load: no tensors");
    vstab_nldf *N = nldf_of(ctx);
    if (!N) return fail(ctx, VSTAB_E_NOMEM, "nldf_load: out of host memory");
    std::vector<float> host;
    auto reserve = [&](size_t n) { size_t o = (host.size() + 63) / 64 * 64; host.resize(o + n, 0.f); return o; };
    std::vector<double> ones;
    // plain conv: W [k][k][cin][cout], b [cout]
    auto conv = [&](const std::string &name, int k, int cin, int cs_in, int cout, size_t &ow, size_t &ob) -> bool {
        const vstab_tensor *W = find(t, count, name + "/W"), *b = find(t, count, name + "/b");
        if (!shape_is(W, {k, k, cin, cout}) || !shape_is(b, {cout})) { fail(ctx, VSTAB_E_WEIGHTS, "missing or mis-shaped variable %s/{W,b}", name.c_str()); return false; }
        const int BN = cout >= 128 ? 128 : (cout > 32 ? 64 : 32), npad = round_up(cout, BN);
        const KLayout L = cs_in == cin ? klayout_run(k, k, cs_in) : klayout_tap(k, k, cin, cs_in);
        ones.assign(npad, 1.0);
        ob = reserve(npad);
        fold_bn(b->data, nullptr, nullptr, nullptr, cout, npad, ones.data(), host.data() + ob);
        ow = reserve((size_t)L.ktiles() * npad * 32);
        pack_conv(W->data, ones.data(), k, k, cin, cs_in, cout, npad, L, host.data() + ow);
        return true;
    };
    NldfWeights &o = N->o;
    if (!conv("Fea_Global_1", 5, 512, 512, FEA, o.g1_w, o.g1_b)) return VSTAB_E_WEIGHTS;
    if (!conv("Fea_Global_2", 5, FEA, FEA, FEA, o.g2_w, o.g2_b)) return VSTAB_E_WEIGHTS;
    if (!conv("Fea_Global", 3, FEA, FEA, FEA, o.g3_w, o.g3_b)) return VSTAB_E_WEIGHTS;
    for (int k = 0; k < 5; ++k)
        if (!conv("Fea_P" + std::to_string(k + 1), 3, POOL_C[k], POOL_C[k], FEA, o.p_w[k], o.p_b[k])) return VSTAB_E_WEIGHTS;
    for (int k = 0; k < 4; ++k) {               // Fea_P(k+2)_Deconv: cat(k+2) -> UP_C[k]
        const std::string name = "Fea_P" + std::to_string(k + 2) + "_Deconv";
        const int cin = CAT_C[k + 1], cout = UP_C[k];
        const vstab_tensor *W = find(t, count, name + "/W"), *b = find(t, count, name + "/b");
        if (!shape_is(W, {5, 5, cout, cin}) || !shape_is(b, {cout})) return fail(ctx, VSTAB_E_WEIGHTS, "missing or mis-shaped variable %s/{W,b}", name.c_str());
        const int npad = round_up(cout, 128);
        ones.assign(npad, 1.0);
        o.d_b[k] = reserve(npad);
        fold_bn(b->data, nullptr, nullptr, nullptr, cout, npad, ones.data(), host.data() + o.d_b[k]);
        o.d_w[k] = reserve(4 * (size_t)klayout_run(3, 3, cin).ktiles() * npad * 32);
        pack_deconv5(W->data, ones.data(), cin, cin, cout, npad, host.data() + o.d_w[k]);
    }
    if (!conv("Local_Fea", 1, 768, 768, 640, o.lf_w, o.lf_b)) return VSTAB_E_WEIGHTS;
    if (!conv("Local_Score", 1, 640, 640, 2, o.ls_w, o.ls_b)) return VSTAB_E_WEIGHTS;
    if (!conv("Global_Score", 1, FEA, FEA, 2, o.gs_w, o.gs_b)) return VSTAB_E_WEIGHTS;

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (N->w) { (void)hipFree(N->w); N->w = nullptr; }
    N->loaded = false;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&N->w), host.size() * sizeof(float));
    if (e != hipSuccess) return fail(ctx, VSTAB_E_NOMEM, "hipMalloc(%zu bytes of NLDF weights): %s", host.size() * 4, hipGetErrorString(e));
    HIP_TRY(ctx, hipMemcpy(N->w, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    N->loaded = true;
    return VSTAB_OK;
}

extern "C" int vstab_nldf_forward(vstab_ctx *ctx, const float *const *pools5, int B, float *prob, float *score, float *local_fea,
                                  float *fea_global, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "nldf_forward: ctx is NULL");
    vstab_nldf *N = nldf_of(ctx);
    if (!N || !N->loaded) return fail(ctx, VSTAB_E_STATE, "nldf_forward: vstab_nldf_load has not been called");
    if (!pools5 || !prob || !workspace) return fail(ctx, VSTAB_E_STATE, "nldf_forward: NULL buffer");
    for (int k = 0; k < 5; ++k)
        if (!pools5[k] || ((uintptr_t)pools5[k] & 15)) return fail(ctx, VSTAB_E_ALIGN, "nldf_forward: pool%d NULL or not 16-byte aligned", k + 1);
    NldfPlan pl;
    if (!nldf_plan(B, pl)) return fail(ctx, VSTAB_E_SHAPE, "nldf_forward: unsupported batch %d", B);
    if (workspace_bytes < pl.total) return fail(ctx, VSTAB_E_NOMEM, "nldf_forward: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    if ((uintptr_t)workspace & 255) return fail(ctx, VSTAB_E_ALIGN, "nldf_forward: workspace must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    char *ws = (char *)workspace;
    auto buf = [&](int b) { return (float *)(ws + pl.off[b]); };
    const float *w = N->w;
    const NldfWeights &o = N->o;
    float *part = buf(N_PART);
    const size_t part_floats = pl.bytes[N_PART] / 4;

    auto run_conv = [&](const float *in, int hw_in, int cin, int cs_in, int k, int pad, int cout, float *out, int cs_out, int c_off,
                        int act, size_t ow, size_t ob) -> int {
        ConvParams p; ConvTile tile; bool vec;
        if (!fill_plain_conv(p, tile, vec, B, hw_in, hw_in, cin, cs_in, k, 1, pad, cout, cs_out, c_off, act))
            return fail(ctx, VSTAB_E_SHAPE, "nldf_forward: conv %dx%d on %d does not fit", k, k, hw_in);
        if (p.ksplit > 1 && (size_t)p.ksplit * p.Mmax * p.Npad > part_floats) p.ksplit = 1;
        if (p.ksplit > 1 && ((cout & 3) || (cs_out & 3) || (c_off & 3))) p.ksplit = 1;
        p.in = in; p.out = out; p.wpk = w + ow; p.bias = w + ob; p.partial = part;
        HIP_TRY(ctx, launch_conv(p, tile, vec, stream));
        return VSTAB_OK;
    };
#define RUN(...) do { const int rc_ = run_conv(__VA_ARGS__); if (rc_ != VSTAB_OK) return rc_; } while (0)

    // global branch (NLDF.py:36-41): 5x5 VALID, 5x5 VALID (ReLU), 3x3 VALID (linear) on pool5: 11 -> 7 -> 3 -> 1
    RUN(pools5[4], 11, 512, 512, 5, 0, FEA, buf(N_G1), FEA, 0, 2, o.g1_w, o.g1_b);
    RUN(buf(N_G1), 7, FEA, FEA, 5, 0, FEA, buf(N_G2), FEA, 0, 2, o.g2_w, o.g2_b);
    RUN(buf(N_G2), 3, FEA, FEA, 3, 0, FEA, buf(N_G), FEA, 0, 0, o.g3_w, o.g3_b);
    // local features (:44-48) + contrast (:50-54) into the first 256 channels of cat1..cat5
    const int cat_buf[5] = {N_CAT1, N_CAT2, N_CAT3, N_CAT4, N_CAT5};
    for (int k = 0; k < 5; ++k) {
        RUN(pools5[k], POOL_HW[k], POOL_C[k], POOL_C[k], 3, 1, FEA, buf(cat_buf[k]), CAT_C[k], 0, 2, o.p_w[k], o.p_b[k]);
        HIP_TRY(ctx, launch_contrast(buf(cat_buf[k]), B, POOL_HW[k], POOL_HW[k], FEA, CAT_C[k], FEA, stream));
    }
    // top-down 5x5 stride-2 transposed convs + ReLU (:57-64): cat5 -> cat4[256:], ..., cat2 -> cat1[256:]
    for (int k = 3; k >= 0; --k) {
        ConvParams p;
        std::memset(&p, 0, sizeof p);
        const int hin = POOL_HW[k + 1], hout = POOL_HW[k], cin = CAT_C[k + 1], cout = UP_C[k];
        p.B = B; p.Hi = hin; p.Wi = hin; p.Cs_in = cin;
        const KLayout L = klayout_run(3, 3, cin);
        set_layout(p, L);
        p.s_in = 1; p.s_out = 2; p.Ho = hout; p.Wo = hout; p.Cs_out = CAT_C[k]; p.c_off = 2 * FEA;
        p.N = cout; p.Npad = round_up(cout, 128); p.act = 2; p.nphase = 4;
        const size_t phase_floats = (size_t)L.ktiles() * p.Npad * 32;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                ConvPhase &ph = p.ph[py * 2 + px];
                ph.Hg = (hout - py + 1) / 2; ph.Wg = (hout - px + 1) / 2; ph.M = B * ph.Hg * ph.Wg;
                ph.off_y = -1; ph.off_x = -1; ph.o_y = py; ph.o_x = px;
                ph.w_off = (long long)(py * 2 + px) * phase_floats;
                p.Mmax = std::max(p.Mmax, ph.M);
            }
        set_ranges(p);
        choose_split(p, 128);
        if ((size_t)p.nphase * p.ksplit * p.Mmax * p.Npad > part_floats) p.ksplit = 1;
        p.in = buf(cat_buf[k + 1]); p.out = buf(cat_buf[k]); p.wpk = w + o.d_w[k]; p.bias = w + o.d_b[k]; p.partial = part;
        HIP_TRY(ctx, launch_conv(p, TILE_128x128, true, stream));
    }
    // scores (:66-77)
    float *lf = local_fea ? local_fea : buf(N_LF);
    RUN(buf(N_CAT1), 176, 768, 768, 1, 0, 640, lf, 640, 0, 0, o.lf_w, o.lf_b);
    RUN(lf, 176, 640, 640, 1, 0, 2, buf(N_LS), 2, 0, 0, o.ls_w, o.ls_b);
    RUN(buf(N_G), 1, FEA, FEA, 1, 0, 2, buf(N_GS), 2, 0, 0, o.gs_w, o.gs_b);
#undef RUN
    HIP_TRY(ctx, launch_nldf_score(buf(N_LS), buf(N_GS), B, 176 * 176, score, prob, stream));
    if (fea_global)
        HIP_TRY(ctx, hipMemcpyAsync(fea_global, buf(N_G), sizeof(float) * (size_t)B * FEA, hipMemcpyDeviceToDevice, stream));
    return VSTAB_OK;
}
