// C ABI of libvstab_hip.so (include/vstab.h): context, weight packing/upload, the
// FlowNetS-pyramid forward schedule (model.py:786-893) and the glue/warp entry points.
#include <cstdlib>
#include <map>
#include <memory>
#include <new>

#include "api_internal.h"

using namespace vstab;

// ------------------------------------------------------------------------- errors
static thread_local std::string g_last_error;

int fail(vstab_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) { std::lock_guard<std::mutex> g(ctx->err_mu); ctx->err = buf; }
    return code;
}

// a context-less callee failed: its message becomes the context's
static void adopt_last_error(vstab_ctx *ctx)
{
    if (ctx) { std::lock_guard<std::mutex> g(ctx->err_mu); ctx->err = g_last_error; }
}

// ------------------------------------------------------------------------- roctx ranges
#include <dlfcn.h>
namespace {
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)();
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop = nullptr;
bool g_trace_on = false;
}

bool trace_ranges_enable(bool on)
{
    if (on && !g_roctx_push) {
        for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_roctx_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
            g_roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
            if (g_roctx_push && g_roctx_pop) break;
            g_roctx_push = nullptr; g_roctx_pop = nullptr;
        }
    }
    g_trace_on = on && g_roctx_push != nullptr;
    return !on || g_trace_on;
}

TraceRange::TraceRange(const char *name) : active(g_trace_on)
{
    if (active) g_roctx_push(name);
}
TraceRange::~TraceRange()
{
    if (active) g_roctx_pop();
}

extern "C" int vstab_trace_ranges(int on)
{
    if (!trace_ranges_enable(on != 0)) return fail(nullptr, VSTAB_E_STATE, "trace_ranges: no roctx library (librocprofiler-sdk-roctx.so / libroctx64.so) could be loaded");
    return VSTAB_OK;
}
// ------------------------------------------------------------------------- net spec
namespace {

struct Enc { const char *name; int k, s, p, cout; };
const Enc ENC[10] = {{"1", 7, 2, 3, 64},    {"2", 5, 2, 2, 128},   {"3", 5, 2, 2, 256},  {"3_1", 3, 1, 1, 256},
                     {"4", 3, 2, 1, 512},   {"4_1", 3, 1, 1, 512}, {"5", 3, 2, 1, 512},  {"5_1", 3, 1, 1, 512},
                     {"6", 3, 2, 1, 1024},  {"6_1", 3, 1, 1, 1024}};
const char *DEC_NAME[4] = {"deconv5", "deconv4", "deconv3", "deconv2"};
const char *UP_NAME[4] = {"upsample6_5", "upsample5_4", "upsample4_3", "upsample3_2"};
const char *PRED_NAME[5] = {"predict6", "predict5", "predict4", "predict3", "predict2"};
const int DEC_COUT[4] = {512, 256, 128, 64};
const int SKIP_C[4] = {512, 512, 256, 128};                 // conv5_1, conv4_1, conv3_1, conv2
const int CONCAT_C[4] = {1026, 770, 386, 194};              // concat5..2 (model.py:853,862,871,880)
const int CONCAT_CS[4] = {1028, 772, 388, 196};             // padded pixel stride (multiple of 4)
const int DEC_CIN[4] = {1024, 1026, 770, 386};              // channels the deconv consumes
const int DEC_CS_IN[4] = {1024, 1028, 772, 388};
const int PRED_CIN[4] = {1024, 1026, 770, 386};             // predict6,5,4,3
const int PRED_CS[4] = {1024, 1028, 772, 388};

enum Buf { B_CONV1, B_CONCAT2, B_CONV3, B_CONCAT3, B_CONV4, B_CONCAT4, B_CONV5, B_CONCAT5, B_CONV6, B_CONV6_1, B_T,
           B_T6, B_T5, B_T4, B_T3, B_TICKETS, B_PARTIAL, B_WINO_V, B_WINO_M, N_BUF };
// B_PARTIAL, B_WINO_V, B_WINO_M stay LAST: their sizes depend on plan decisions (split-K factors, Winograd or direct) that a pinned
// plan may change, while every offset before them depends on the shape alone (vstab_workspace_layout relies on it)
static_assert(B_PARTIAL == N_BUF - 3 && B_WINO_V == N_BUF - 2 && B_WINO_M == N_BUF - 1, "plan-dependent buffers must come last");
const char *BUF_NAME[N_BUF] = {"conv1", "concat2", "conv3", "concat3", "conv4", "concat4", "conv5", "concat5",
                               "conv6", "conv6_1", "pf2_taps", "pf6_taps", "pf5_taps", "pf4_taps", "pf3_taps", "tickets", "splitk", "winograd_in", "winograd_out"};

// where each encoder stage reads and writes: {in buf (-1 = feats), out buf, out stride, used in channels, in stride}
struct EncIO { int in_buf, out_buf, cs_out, cs_in; };
const EncIO ENC_IO[10] = {
    {-1, B_CONV1, 64, 0},         {B_CONV1, B_CONCAT2, 196, 64},   {B_CONCAT2, B_CONV3, 256, 196},
    {B_CONV3, B_CONCAT3, 388, 256}, {B_CONCAT3, B_CONV4, 512, 388}, {B_CONV4, B_CONCAT4, 772, 512},
    {B_CONCAT4, B_CONV5, 512, 772}, {B_CONV5, B_CONCAT5, 1028, 512}, {B_CONCAT5, B_CONV6, 1024, 1028},
    {B_CONV6, B_CONV6_1, 1024, 1024}};

struct Plan {
    int B, H, W, Cin;
    int eh[10], ew[10];
    size_t off[N_BUF];      // byte offsets
    size_t bytes[N_BUF];
    int buf_h[N_BUF], buf_w[N_BUF], buf_c[N_BUF], buf_cs[N_BUF];
    size_t total;
    // conv-like launches: 10 encoder, 4 deconv, predict2 tap table, predict6..3 tap tables
    ConvParams cp[19];
    ConvTile tile[19];
    bool vec4[19];
    bool skinny[19];        // few-row layers as weight streams (conv_skinny.hip): tile[i] == TILE_SKINNY, cp[i].ksplit = its own factor
    // Winograd F(2x2,3x3) form of the 3x3 stride-1 encoder stages (cp[i] stays the direct form: host-plan tests, fallback)
    bool wino[10];
    ConvParams wcp[10];
    ConvTile wtile[10];
    // Winograd F(2x2,2x2) form of the transposed convolutions (winograd_ops.hip; cp[10 + l] stays the direct form): the 9-position GEMM
    bool wdec[4];
    ConvParams wdcp[4];
    ConvTile wdtile[4];
    WdecGeom wdg[4];
};

// What a context pins about its launch plans (vstab_set_plan_batch / vstab_set_plan_flags).  batch > 0: every per-layer decision that
// changes the ARITHMETIC of a sample -- split-K factors, Winograd or direct form, weight-stream or tiled kernel -- is taken for a
// batch of `batch` samples and reused for any smaller batch, so a sample's results do not depend on what it is batched with
// (sharded clips with ragged tails: main:553-558's samples are independent).
struct PlanPin { int batch = 0; unsigned flags = 0; };

bool level_sizes(int H, int W, int *eh, int *ew)
{
    int h = H, w = W;
    for (int i = 0; i < 10; ++i) {
        h = (h + 2 * ENC[i].p - ENC[i].k) / ENC[i].s + 1;
        w = (w + 2 * ENC[i].p - ENC[i].k) / ENC[i].s + 1;
        if (h < 1 || w < 1) return false;
        eh[i] = h; ew[i] = w;
    }
    // deconv output_shape := skip size needs ceil(out/2) == in
    const int lv[5] = {9, 7, 5, 3, 1};
    for (int l = 0; l < 4; ++l)
        if ((eh[lv[l + 1]] + 1) / 2 != eh[lv[l]] || (ew[lv[l + 1]] + 1) / 2 != ew[lv[l]]) return false;
    return H >= 3 && W >= 3;
}

}  // namespace

// Split-K factor from a small cost model instead of a fixed rule.  A launch is `tiles x ks` workgroups on 512 slots
// (256 CUs x 2 co-resident workgroups); a K-tile costs TAU2 when two workgroups share a CU and TAU1 when one has
// the CU to itself, every workgroup pays a fixed prologue/epilogue T0, and splitting adds the combine pass and the
// slab traffic.  Constants calibrated on the cfg1 / B=1 profiles (profiles/README.md, "split-K model").
static double split_cost_us(long long tiles, int KT, int ks, double slab_bytes, int BN, int BM, int *ks_eff_out)
{
    // 64-row tiles cost half a 128-row tile per K-tile when few workgroups run (measured), a little more than half
    // on a full chip (1.5x the LDS fragment reads per MFMA), so large layers keep the 128-row tile
    const double TAU2 = 4.2 * (BN >= 128 ? 1.0 : (BN == 64 ? 0.58 : 0.36)) * (BM == 64 ? 0.55 : 1.0);
    const double TAU1 = 0.525 * TAU2, T0 = 5.0;     // in-situ: 2.10 vs 4.00 us per K-tile (conv4_1), 2.20 vs 4.19 (conv4); with the assembly K loop
                                                    // 1.87 vs 3.52: same ratio, and 3.55 / 0.53 / T0 3..11 pick the same splits at B=8 512x512 (r03k sweep)
    // bytes per us for the slab traffic: slabs that stay in the L2s (32 MB across the 8 XCDs; one sample's) move at ~12 TB/s, a launch's
    // worth beyond that goes through the Infinity Cache / HBM (round 5 A/B, profiles/ab_r05t_slab_bandwidth.txt: B=8 512x512 conv5 /
    // deconv5 / deconv4 with 34 / 34 / 17 MB of slabs at split 8 / 8 / 4 are faster at 4 / 4 / 2)
    const double BW = slab_bytes * ((KT + ((KT + ks - 1) / ks) - 1) / ((KT + ks - 1) / ks)) > 16e6 ? 5.0e6 : 1.2e7;
    const int kts = (KT + ks - 1) / ks, ks_eff = (KT + kts - 1) / kts;
    const long long blocks = tiles * ks_eff, full = blocks / 512, rem = blocks % 512;
    double t = (double)full * (kts * TAU2 + T0);
    if (rem > 256) t += kts * TAU2 + T0;
    else if (rem > 0) t += kts * TAU1 + T0;
    if (ks_eff > 1) t += 6.0 + (2.0 * ks_eff + 1.0) * slab_bytes / BW;
    *ks_eff_out = ks_eff;
    return t;
}

static double best_split(const ConvParams &p, int BN, int BM, int *ks_out)
{
    const int KT = p.KH * p.NSEG * (p.SEGP / 32);
    const long long tiles = (long long)((p.Mmax + BM - 1) / BM) * (p.Npad / BN) * p.nphase;
    int best = 1, dummy;
    double best_t = split_cost_us(tiles, KT, 1, 0.0, BN, BM, &dummy);
    *ks_out = 1;
    if ((p.N & 3) != 0 || tiles >= 2048) return best_t;
    const double slab = (double)p.Mmax * p.nphase * p.Npad * 4.0;
    const int cap = std::min(64, std::max(1, KT / 3));
    for (int ks = 2; ks <= cap; ++ks) {
        int eff;
        const double t = split_cost_us(tiles, KT, ks, slab, BN, BM, &eff);
        if (eff != ks || slab * ks > 768e6) continue;               // only factors that divide the K-tiles evenly enough
        if (t < best_t * 0.995) { best_t = t; best = ks; }          // prefer the smaller factor on ties
    }
    *ks_out = best;
    return best_t;
}

void choose_split(ConvParams &p, int BN, int BM)
{
    int ks;
    best_split(p, BN, BM, &ks);
    p.ksplit = ks;
}

// Tile + split-K for a 128-column layer: small-M layers (one sample, the 1/32 and 1/64 levels) waste most of a
// 128-row tile and become fixed-overhead / weight-streaming bound; the 64x128 variant halves the MFMA work per K-tile
// there (B=1 384x512: conv5..deconv5 29-36 us -> 21-27 us each in tools/conv_bench).
ConvTile choose_tile_split(ConvParams &p, ConvTile tile, bool vec4)
{
    const int BN = tile == TILE_128x128 ? 128 : (tile == TILE_128x64 ? 64 : 32);
    int ks128;
    const double t128 = best_split(p, BN, 128, &ks128);
    p.ksplit = ks128;
    if (tile != TILE_128x128 || !vec4 || !conv_uses_lds_dma(tile, vec4)) return tile;
    int ks64;
    const double t64 = best_split(p, 128, 64, &ks64);
    if (t64 < 0.95 * t128) { p.ksplit = ks64; return TILE_64x128; }
    return tile;
}

// Few rows per phase (one sample's 1/32 and 1/64 levels, the first decoder steps): the layer is a weight stream (conv_skinny.hip).
// Returns true and sets p.ksplit to that kernel's own factor.
bool choose_skinny(ConvParams &p, bool vec4, unsigned flags)
{
    if (flags & VSTAB_PLAN_NO_SKINNY) return false;
    if (!conv_skinny_applicable(p, vec4)) return false;
    p.ksplit = conv_skinny_split(p);
    return true;
}

namespace {
KLayout enc_layout(int i, int cin_first)
{
    const Enc &e = ENC[i];
    const int cin = i == 0 ? cin_first : ENC[i - 1].cout;
    const int cs_in = i == 0 ? cin_first : ENC_IO[i].cs_in;
    if (cs_in == cin) return klayout_run(e.k, e.k, cs_in);
    return klayout_tap(e.k, e.k, cin, cs_in);
}

}  // namespace

void set_layout(ConvParams &p, const KLayout &L)
{
    p.KH = L.KH; p.NSEG = L.NSEG; p.SEG = L.SEG; p.SEGP = L.SEGP; p.SEG_STRIDE = L.SEG_STRIDE;
}

// buffer-descriptor ranges (bytes); called once Npad and the layout are final
void set_ranges(ConvParams &p)
{
    p.in_bytes = (unsigned)std::min<long long>((long long)p.B * p.Hi * p.Wi * p.Cs_in * 4, 0xFFFFFFFFLL);
    p.w_bytes = (unsigned)std::min<long long>((long long)p.KH * p.NSEG * (p.SEGP / 32) * p.Npad * 128, 0xFFFFFFFFLL);
}

// The 16-phase 1x1 GEMM over Winograd-transformed tiles (winograd_ops.hip): V [B][16][TH*TW][cin] -> M [B][16][TH*TW][cout]
void fill_wino_gemm(ConvParams &p, int B, int H, int W, int cin, int cout)
{
    std::memset(&p, 0, sizeof p);
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    p.B = B; p.Hi = 16 * TH; p.Wi = TW; p.Cs_in = cin;
    const KLayout L = klayout_run(1, 1, cin);
    set_layout(p, L);
    p.s_in = 1; p.s_out = 1; p.Ho = 16 * TH; p.Wo = TW; p.Cs_out = cout; p.c_off = 0;
    p.N = cout; p.Npad = cout; p.act = 0; p.nphase = 16;
    const size_t phase_floats = (size_t)L.ktiles() * p.Npad * 32;
    for (int xi = 0; xi < 16; ++xi) {
        ConvPhase &ph = p.ph[xi];
        ph.Hg = TH; ph.Wg = TW; ph.M = B * TH * TW;
        ph.off_y = xi * TH; ph.off_x = 0; ph.o_y = xi * TH; ph.o_x = 0;
        ph.w_off = (long long)xi * phase_floats;
    }
    p.Mmax = B * TH * TW;
    set_ranges(p);
    p.ksplit = 1;
}

// does a 3x3 stride-1 pad-1 layer run in Winograd form?  Two extra HBM passes and two more launches: pays once the Winograd-domain
// GEMM issues a GFLOP or two (measured: B=8 512x512 every encoder stage gains, 20..96 us; one 384x512 sample -- 1.61 GFLOP per stage -- lost
// 2..5 % in rounds 2-3 and GAINS 2.5 % of the frame since the stream GEMMs and one-workgroup-per-CU plans of rounds 4-5: conv3_1 / conv4_1
// 38.4 / 37.1 -> 24.2 / 25.2 us and their split-K combines gone (profiles/ab_r05k_winograd_threshold.txt); one 256x256 sample, 0.54 GFLOP per
// stage, still loses 4 %)
bool wino_applies(int B, int H, int W, int cin, int cout)
{
#ifdef VSTAB_HARNESS
    static const bool wino_on = getenv("VSTAB_NO_WINOGRAD") == nullptr;       // A/B switch of the tuning harness builds
    if (!wino_on) return false;
#endif
    if ((cin & 31) || (cout & 127)) return false;                  // whole K tiles, 128x64 output tiles
    const long long TH = (H + 1) / 2, TW = (W + 1) / 2;
    if ((long long)B * 16 * TH * TW * std::max(cin, cout) * 4 >= 0x80000000LL) return false;
#ifndef VSTAB_WINO_MIN_FLOPS
#define VSTAB_WINO_MIN_FLOPS 1.5e9       // (A/B builds: scripts/build_variant_lib.sh -DVSTAB_WINO_MIN_FLOPS=...)
#endif
    return 32.0 * B * TH * TW * cin * cout >= VSTAB_WINO_MIN_FLOPS;
}

// The 9-position 1x1 GEMM over F(2x2,2x2)-transformed tiles of a transposed convolution's input (winograd_ops.hip):
// V [B][9][NTy][NTx][cs_in] -> M [B][9][NTy][NTx][4 cout]; position (i, j) multiplies only the nty[i] x ntx[j] tiles that are not zero
void fill_wdec_gemm(ConvParams &p, int B, const WdecGeom &g, int cs_in, int cout)
{
    std::memset(&p, 0, sizeof p);
    p.B = B; p.Hi = 9 * g.NTy; p.Wi = g.NTx; p.Cs_in = cs_in;
    const KLayout L = klayout_run(1, 1, cs_in);
    set_layout(p, L);
    p.s_in = 1; p.s_out = 1; p.Ho = 9 * g.NTy; p.Wo = g.NTx; p.Cs_out = 4 * cout; p.c_off = 0;
    p.N = 4 * cout; p.Npad = 4 * cout; p.act = 0; p.nphase = 9;
    const size_t pos_floats = (size_t)L.ktiles() * p.Npad * 32;
    p.Mmax = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            ConvPhase &ph = p.ph[i * 3 + j];
            ph.Hg = g.nty[i]; ph.Wg = g.ntx[j]; ph.M = B * ph.Hg * ph.Wg;
            ph.off_y = (i * 3 + j) * g.NTy; ph.off_x = 0; ph.o_y = (i * 3 + j) * g.NTy; ph.o_x = 0;
            ph.w_off = (long long)(i * 3 + j) * pos_floats;
            p.Mmax = std::max(p.Mmax, ph.M);
        }
    set_ranges(p);
    p.ksplit = 1;
}

// does refinement level l's transposed convolution run in Winograd F(2x2,2x2) form?  9/16 of the multiply-adds, against two more HBM
// passes (V and M), one more launch, a reduction of only Cin (25 / 33 K-tiles per workgroup instead of 100 / 132) and a ragged tile
// grid (one more tile per axis than Hin/2).  Decided with the cost model that picks the split-K factors: the direct launch's modelled
// time `t_direct_us` against the 9-position GEMM's (on 128- or 64-row tiles, whichever the model prefers: *tile_out) plus the two
// transforms at the bandwidth they measure (4.5 TB/s over input + V + M + output; profiles/README.md r06).  Measured: at B=8 512x512
// deconv3 gains a little (-12 us of 200) and deconv4 would lose (336 workgroups on 512 slots: as long as the direct form) -- the model
// says the same; at 16 x 720p / 8 x 1080p per chunk deconv4 and deconv3 take 0.60 / 0.63 of their direct time and the step -3.7 % / -2.2 %.
// deconv2 (Cin 386 -> 64: 13 K-tiles, N = 256, V and M larger than the layer's own tensors) is not built.
bool wdec_applies(int l, int B, const WdecGeom &g, int Hi, int Wi, int Ho, int Wo, int cs_in, int cout, double t_direct_us, int ks_direct, bool force,
                  ConvTile *tile_out)
{
    *tile_out = TILE_128x128;
#ifdef VSTAB_HARNESS
    static const bool wdec_on = getenv("VSTAB_NO_WDEC") == nullptr;           // A/B switch of the tuning harness builds
    if (!wdec_on) return false;
#endif
    if (l < 0 || l > 2 || (cs_in & 3) || ((4 * cout) & 127) || 4 * cout > 2048) return false;     // (2048 = the zero bias the GEMMs share)
    if ((long long)B * 9 * g.NTy * g.NTx * std::max(cs_in, 4 * cout) * 4 >= 0x80000000LL) return false;
    const int KT = round_up(cs_in, 32) / 32;
    double best = 1e30;
    for (int BM : {128, 64}) {
        long long tiles = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) tiles += ((long long)B * g.nty[i] * g.ntx[j] + BM - 1) / BM;
        int eff;
        const double t = split_cost_us(tiles * (4 * cout / 128), KT, 1, 0.0, 128, BM, &eff);
        if (t < best) { best = t; *tile_out = BM == 128 ? TILE_128x128 : TILE_64x128; }
    }
    if (force) return true;                                                   // VSTAB_PLAN_FORCE_WDEC: small test shapes
    const double plane = (double)B * g.NTy * g.NTx;
    const double bytes = 4.0 * ((double)B * Hi * Wi * cs_in + 9.0 * plane * cs_in + 9.0 * plane * 4.0 * cout + (double)B * Ho * Wo * cout);
#ifndef VSTAB_WDEC_MARGIN
#define VSTAB_WDEC_MARGIN 0.92           // (A/B builds: scripts/build_variant_lib.sh -DVSTAB_WDEC_MARGIN=...)
#endif
    // a direct launch the model splits in K is a small one (B=8 512x512: deconv5 / deconv4, split 4 / 2): the model prices those 20-25 %
    // too high (measured 71 / 133 us against 89 / 163) and the ragged 9-position grid quantises badly on 512 slots -- they stay direct
    if (ks_direct > 1) return false;
    return best + bytes / 4.5e6 + 3.0 < VSTAB_WDEC_MARGIN * t_direct_us;
}

namespace {
int max_chunk(int B, int H, int W, int Cin);

bool make_plan(int B, int H, int W, int Cin, Plan &pl, const PlanPin *pin = nullptr)
{
    if (B < 1 || Cin < 1 || Cin > 4096) return false;
    pl.B = B; pl.H = H; pl.W = W; pl.Cin = Cin;
    if (!level_sizes(H, W, pl.eh, pl.ew)) return false;
    const unsigned flags = pin ? pin->flags : 0u;
    // a pinned plan batch: the decisions come from the plan of (one chunk of) that batch
    std::unique_ptr<Plan> ref;
    if (pin && pin->batch > 0) {
        if (B > pin->batch) return false;
        const int cmax = max_chunk(pin->batch, H, W, Cin);
        if (cmax < 1) return false;
        const int nch = (pin->batch + cmax - 1) / cmax, rb = (pin->batch + nch - 1) / nch;
        if (B > rb) return false;                       // callers process a pinned batch in chunks of rb (vstab_flownets_forward)
        if (B != rb) {
            ref.reset(new (std::nothrow) Plan);
            PlanPin unpinned; unpinned.flags = flags;
            if (!ref || !make_plan(rb, H, W, Cin, *ref, &unpinned)) return false;
        }
    }
    // every tensor must stay below 2^31 BYTES: the kernels address through buffer descriptors with
    // 32-bit byte offsets and use 0xC0000000 as the "reads as zero" offset (larger batches are
    // processed in chunks by vstab_flownets_forward)
    const long long lim = (1LL << 29) - 1;
    if ((long long)B * H * W * Cin > lim) return false;

    auto setbuf = [&](int b, int h, int w, int c, int cs) { pl.buf_h[b] = h; pl.buf_w[b] = w; pl.buf_c[b] = c; pl.buf_cs[b] = cs; };
    setbuf(B_CONV1, pl.eh[0], pl.ew[0], 64, 64);
    setbuf(B_CONCAT2, pl.eh[1], pl.ew[1], 194, 196);
    setbuf(B_CONV3, pl.eh[2], pl.ew[2], 256, 256);
    setbuf(B_CONCAT3, pl.eh[3], pl.ew[3], 386, 388);
    setbuf(B_CONV4, pl.eh[4], pl.ew[4], 512, 512);
    setbuf(B_CONCAT4, pl.eh[5], pl.ew[5], 770, 772);
    setbuf(B_CONV5, pl.eh[6], pl.ew[6], 512, 512);
    setbuf(B_CONCAT5, pl.eh[7], pl.ew[7], 1026, 1028);
    setbuf(B_CONV6, pl.eh[8], pl.ew[8], 1024, 1024);
    setbuf(B_CONV6_1, pl.eh[9], pl.ew[9], 1024, 1024);
    setbuf(B_T, pl.eh[1], pl.ew[1], 32, 32);
    setbuf(B_T6, pl.eh[9], pl.ew[9], 32, 32);
    setbuf(B_T5, pl.eh[7], pl.ew[7], 32, 32);
    setbuf(B_T4, pl.eh[5], pl.ew[5], 32, 32);
    setbuf(B_T3, pl.eh[3], pl.ew[3], 32, 32);
    setbuf(B_TICKETS, 0, 0, 0, 0);
    setbuf(B_PARTIAL, 0, 0, 0, 0);
    setbuf(B_WINO_V, 0, 0, 0, 0);
    setbuf(B_WINO_M, 0, 0, 0, 0);
    for (int b = 0; b < N_BUF; ++b) {
        const long long n = (long long)B * pl.buf_h[b] * pl.buf_w[b] * pl.buf_cs[b];
        if (n > lim) return false;
        pl.bytes[b] = (size_t)n * 4;
    }
    // ticket words of the in-launch split-K reductions (conv_skinny.hip): per WORKSPACE, so forwards on one context that use distinct
    // workspaces never share them; zeroed at the start of every forward that has such a layer -- by the first layer's own launch
    // (conv_rowwin's first workgroup), or by a memset node when that layer runs on another kernel
    pl.bytes[B_TICKETS] = SKINNY_MAX_TILES * sizeof(unsigned);

    // ---- encoder convs
    size_t partial_floats = 0;
    for (int i = 0; i < 10; ++i) {
        ConvParams &p = pl.cp[i];
        std::memset(&p, 0, sizeof p);
        const Enc &e = ENC[i];
        const int hi = i == 0 ? H : pl.eh[i - 1], wi = i == 0 ? W : pl.ew[i - 1];
        p.B = B; p.Hi = hi; p.Wi = wi;
        p.Cs_in = i == 0 ? Cin : ENC_IO[i].cs_in;
        set_layout(p, enc_layout(i, Cin));
        p.s_in = e.s; p.s_out = 1;
        p.Ho = pl.eh[i]; p.Wo = pl.ew[i]; p.Cs_out = ENC_IO[i].cs_out; p.c_off = 0;
        p.N = e.cout;
        const int BN = e.cout >= 128 ? 128 : 64;
        pl.tile[i] = e.cout >= 128 ? TILE_128x128 : TILE_128x64;
        p.Npad = round_up(e.cout, BN);
        p.act = 1; p.nphase = 1;
        p.ph[0].Hg = pl.eh[i]; p.ph[0].Wg = pl.ew[i]; p.ph[0].M = B * pl.eh[i] * pl.ew[i];
        p.ph[0].off_y = -e.p; p.ph[0].off_x = -e.p; p.ph[0].o_y = 0; p.ph[0].o_x = 0; p.ph[0].w_off = 0;
        p.Mmax = p.ph[0].M;
        pl.vec4[i] = (p.Cs_in % 4 == 0) && (p.SEG % 4 == 0);
        set_ranges(p);
        if (ref) { pl.tile[i] = ref->tile[i]; pl.skinny[i] = ref->skinny[i]; p.ksplit = ref->cp[i].ksplit; }
        else {
            pl.tile[i] = choose_tile_split(p, pl.tile[i], pl.vec4[i]);
            pl.skinny[i] = i > 0 && choose_skinny(p, pl.vec4[i], flags);
            if (pl.skinny[i]) pl.tile[i] = TILE_SKINNY;
        }
        if (p.ksplit > 1) partial_floats = std::max(partial_floats, (size_t)p.ksplit * p.Mmax * p.Npad);
    }
    // ---- Winograd form of the 3x3 stride-1 stages: a 16-phase 1x1 GEMM over the transformed tiles (winograd_ops.hip)
    size_t wino_v = 0, wino_m = 0;
    for (int i = 0; i < 10; ++i) {
        pl.wino[i] = false;
        const Enc &e = ENC[i];
        if (e.k != 3 || e.s != 1 || e.p != 1 || pl.skinny[i]) continue;
        const int cin_i = ENC[i - 1].cout;
        if (ref ? !ref->wino[i] : (ENC_IO[i].cs_in != cin_i || !wino_applies(B, pl.eh[i], pl.ew[i], cin_i, e.cout))) continue;      // plain input buffer
        const int TH = (pl.eh[i] + 1) / 2, TW = (pl.ew[i] + 1) / 2;
        fill_wino_gemm(pl.wcp[i], B, pl.eh[i], pl.ew[i], cin_i, e.cout);
        // The reduction is short (K = C_in: 8..32 K-tiles), so a workgroup's prologue and epilogue weigh in.  Stages with at least
        // two full rounds of 128x128 tiles (2 per CU) take those: twice the MFMA work per prologue + epilogue (in situ with the
        // assembly K loop, B=8 512x512: conv3_1 163.5 -> 152 us, conv4_1 138 -> 135.5); smaller stages keep 128x64 tiles, three
        // co-resident workgroups per CU (conv5_1 40 vs 41.3 us, conv6_1 44.5 vs 68.5)
        {
            const long long t128 = 16LL * ((pl.wcp[i].Mmax + 127) / 128) * (e.cout / 128);
            pl.wtile[i] = t128 >= 1024 ? TILE_128x128 : TILE_128x64;         // (a tile shape changes no sum: not pinned)
        }
        wino_v = std::max(wino_v, (size_t)B * 16 * TH * TW * cin_i);
        wino_m = std::max(wino_m, (size_t)B * 16 * TH * TW * e.cout);
        pl.wino[i] = true;
    }
    // ---- decoder transposed convs: 4 phases of a 2x2-tap conv
    const int dec_in[4] = {B_CONV6_1, B_CONCAT5, B_CONCAT4, B_CONCAT3};
    const int dec_out[4] = {B_CONCAT5, B_CONCAT4, B_CONCAT3, B_CONCAT2};
    for (int l = 0; l < 4; ++l) {
        ConvParams &p = pl.cp[10 + l];
        std::memset(&p, 0, sizeof p);
        const int ib = dec_in[l], ob = dec_out[l];
        p.B = B; p.Hi = pl.buf_h[ib]; p.Wi = pl.buf_w[ib]; p.Cs_in = pl.buf_cs[ib];
        const KLayout L = klayout_deconv(p.Cs_in);
        set_layout(p, L);
        p.s_in = 1; p.s_out = 2;
        p.Ho = pl.buf_h[ob]; p.Wo = pl.buf_w[ob]; p.Cs_out = pl.buf_cs[ob]; p.c_off = SKIP_C[l];
        p.N = DEC_COUT[l];
        const int BN = p.N >= 128 ? 128 : 64;
        pl.tile[10 + l] = p.N >= 128 ? TILE_128x128 : TILE_128x64;
        p.Npad = round_up(p.N, BN);
        p.act = 1; p.nphase = 4;
        const size_t phase_floats = (size_t)L.ktiles() * p.Npad * 32;
        p.Mmax = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                ConvPhase &ph = p.ph[py * 2 + px];
                ph.Hg = (p.Ho - py + 1) / 2; ph.Wg = (p.Wo - px + 1) / 2;
                ph.M = B * ph.Hg * ph.Wg;
                ph.off_y = py - 1; ph.off_x = px - 1; ph.o_y = py; ph.o_x = px;
                ph.w_off = (long long)(py * 2 + px) * phase_floats;
                p.Mmax = std::max(p.Mmax, ph.M);
            }
        pl.vec4[10 + l] = true;
        set_ranges(p);
        if (ref) { pl.tile[10 + l] = ref->tile[10 + l]; pl.skinny[10 + l] = ref->skinny[10 + l]; p.ksplit = ref->cp[10 + l].ksplit; }
        else {
            pl.tile[10 + l] = choose_tile_split(p, pl.tile[10 + l], true);
            // (with the two-problem launches a transposed convolution shares its launch with the flow head and its combine with
            // predict_up: the weight-stream kernel's one advantage -- no combine launch -- is gone, so it serves the encoder only)
            pl.skinny[10 + l] = (flags & VSTAB_PLAN_NO_DUAL) ? choose_skinny(p, true, flags) : false;
            if (pl.skinny[10 + l]) pl.tile[10 + l] = TILE_SKINNY;
        }
        if (p.ksplit > 1) partial_floats = std::max(partial_floats, (size_t)p.nphase * p.ksplit * p.Mmax * p.Npad);
        // Winograd F(2x2,2x2) form (an arithmetic-changing decision: pinned like the others); rides in the two-problem launch only
        pl.wdec[l] = false;
        pl.wdg[l] = wdec_geom(p.Hi, p.Wi, p.Ho, p.Wo);
        if (!(flags & (VSTAB_PLAN_NO_DUAL | VSTAB_PLAN_NO_WDEC)) && !pl.skinny[10 + l]) {
            const ConvTile dt = pl.tile[10 + l];
            int ks_d;
            ConvParams pd1 = p;
            const double t_direct = best_split(pd1, dt == TILE_128x64 ? 64 : 128, dt == TILE_64x128 ? 64 : 128, &ks_d);
            ConvTile wt;
            const bool on = wdec_applies(l, B, pl.wdg[l], p.Hi, p.Wi, p.Ho, p.Wo, p.Cs_in, p.N, t_direct, p.ksplit, (flags & VSTAB_PLAN_FORCE_WDEC) != 0, &wt);
            if (ref ? ref->wdec[l] : on) {
                const WdecGeom &g = pl.wdg[l];
                fill_wdec_gemm(pl.wdcp[l], B, g, p.Cs_in, p.N);
                pl.wdtile[l] = wt;                                               // (a tile shape changes no sum: not pinned)
                wino_v = std::max(wino_v, (size_t)B * 9 * g.NTy * g.NTx * p.Cs_in);
                wino_m = std::max(wino_m, (size_t)B * 9 * g.NTy * g.NTx * 4 * p.N);
                pl.wdec[l] = true;
            }
        }
    }
    pl.bytes[B_WINO_V] = wino_v * 4;
    pl.bytes[B_WINO_M] = wino_m * 4;
    // ---- predict2 tap table: 1x1 conv concat2 -> 18 (pad 32) columns
    {
        ConvParams &p = pl.cp[14];
        std::memset(&p, 0, sizeof p);
        p.B = B; p.Hi = pl.buf_h[B_CONCAT2]; p.Wi = pl.buf_w[B_CONCAT2]; p.Cs_in = 196;
        set_layout(p, klayout_run(1, 1, 196));
        p.s_in = 1; p.s_out = 1;
        p.Ho = p.Hi; p.Wo = p.Wi; p.Cs_out = 32; p.c_off = 0;
        p.N = 32; p.Npad = 32; p.act = 0; p.nphase = 1; p.ksplit = 1;
        p.ph[0].Hg = p.Hi; p.ph[0].Wg = p.Wi; p.ph[0].M = B * p.Hi * p.Wi; p.Mmax = p.ph[0].M;
        // (launched as tap_panel_kernel, tap_panel.hip: the parameters here only feed the flop / byte accounting of the reports)
        pl.tile[14] = TILE_128x32; pl.vec4[14] = true; pl.skinny[14] = false;
        set_ranges(p);
    }
    // ---- predict6..3 tap tables: 1x1 conv of the level's (concat) tensor -> 18 (pad 32) columns
    {
        const int src[4] = {B_CONV6_1, B_CONCAT5, B_CONCAT4, B_CONCAT3}, dst[4] = {B_T6, B_T5, B_T4, B_T3};
        for (int l = 0; l < 4; ++l) {
            ConvParams &p = pl.cp[15 + l];
            std::memset(&p, 0, sizeof p);
            p.B = B; p.Hi = pl.buf_h[src[l]]; p.Wi = pl.buf_w[src[l]]; p.Cs_in = pl.buf_cs[src[l]];
            set_layout(p, klayout_run(1, 1, p.Cs_in));
            p.s_in = 1; p.s_out = 1;
            p.Ho = p.Hi; p.Wo = p.Wi; p.Cs_out = 32; p.c_off = 0;
            p.N = 32; p.Npad = 32; p.act = 0; p.nphase = 1;
            p.ph[0].Hg = p.Hi; p.ph[0].Wg = p.Wi; p.ph[0].M = B * p.Hi * p.Wi; p.Mmax = p.ph[0].M;
            pl.tile[15 + l] = TILE_128x32; pl.vec4[15 + l] = true; pl.skinny[15 + l] = false;
            set_ranges(p);
            if (ref) p.ksplit = ref->cp[15 + l].ksplit;
            else choose_split(p, 32);   // A/B on one box: split-K + combine beats 4..256 long-running workgroups by ~90 us/step
            // the tap table runs in the SAME launch as the level's transposed convolution (conv_dual_kernel): their slabs sit side by side
            const ConvParams &d = pl.cp[10 + l];
            const size_t dec_slab = d.ksplit > 1 ? (size_t)d.nphase * d.ksplit * d.Mmax * d.Npad : 0;
            if (p.ksplit > 1) partial_floats = std::max(partial_floats, dec_slab + (size_t)p.ksplit * p.Mmax * p.Npad);
            (void)dst;
        }
    }
    pl.bytes[B_PARTIAL] = partial_floats * 4;
    size_t off = 0;
    for (int b = 0; b < N_BUF; ++b) {
        pl.off[b] = off;
        off += (pl.bytes[b] + 255) / 256 * 256;
    }
    pl.total = off;
    return true;
}

// Largest batch whose every tensor stays below 2 GiB (0 if even one sample does not fit).
int max_chunk(int B, int H, int W, int Cin)
{
    Plan pl;
    int lo = 0, hi = B;                     // invariant: lo fits (or 0), hi+1.. do not
    if (make_plan(B, H, W, Cin, pl)) return B;
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (make_plan(mid, H, W, Cin, pl)) lo = mid; else hi = mid - 1;
    }
    return lo;
}

}  // namespace

const vstab_tensor *find(const vstab_tensor *t, int n, const std::string &name)
{
    for (int i = 0; i < n; ++i)
        if (t[i].name && name == t[i].name) return &t[i];
    return nullptr;
}

bool shape_is(const vstab_tensor *t, std::initializer_list<int> s)
{
    if (!t || t->ndim != (int)s.size()) return false;
    int i = 0;
    for (int v : s)
        if (t->shape[i++] != v) return false;
    return t->data != nullptr;
}


// ------------------------------------------------------------------------- context
extern "C" const char *vstab_version(void) { return "vstab-hip 0.1 (gfx950)"; }

extern "C" const char *vstab_last_error(const vstab_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

extern "C" int vstab_create(vstab_ctx **out, int device)
{
    if (!out) return fail(nullptr, VSTAB_E_STATE, "vstab_create: out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(nullptr, VSTAB_E_HIP, "vstab_create: device %d of %d", device, n);
    HIP_TRY(nullptr, hipSetDevice(device));
    HIP_TRY(nullptr, conv_set_attributes());
    HIP_TRY(nullptr, rowwin_set_attributes());
    HIP_TRY(nullptr, tap_panel_set_attributes());
    HIP_TRY(nullptr, wino_gemm_stream_set_attributes());
    vstab_ctx *c = new (std::nothrow) vstab_ctx();
    if (!c) return fail(nullptr, VSTAB_E_NOMEM, "vstab_create: out of host memory");
    c->device = device;
    *out = c;
    return VSTAB_OK;
}

extern "C" void vstab_destroy(vstab_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->dev_weights) (void)hipFree(ctx->dev_weights);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    if (ctx->vgg_weights) (void)hipFree(ctx->vgg_weights);
    vstab_nldf_free(ctx->nldf);
    delete ctx;
}

extern "C" int vstab_host_xcd_remap(int gx, int gy, int gz, int lin, int32_t *xyz)
{
    if (!xyz || gx < 1 || gy < 1 || gz < 1 || lin < 0 || (long long)gx * gy * gz > 0x7fffffffLL || lin >= gx * gy * gz)
        return fail(nullptr, VSTAB_E_SHAPE, "host_xcd_remap: bad argument");
    unsigned bx, by, bz;
    xcd_remap_calc((unsigned)gx, (unsigned)gy, (unsigned)gz, (unsigned)lin, bx, by, bz);
    xyz[0] = (int32_t)bx; xyz[1] = (int32_t)by; xyz[2] = (int32_t)bz;
    return VSTAB_OK;
}

extern "C" int vstab_level_sizes(int H, int W, int32_t *hw20)
{
    int eh[10], ew[10];
    if (!hw20 || !level_sizes(H, W, eh, ew)) return fail(nullptr, VSTAB_E_SHAPE, "unsupported input size %dx%d", H, W);
    for (int i = 0; i < 10; ++i) { hw20[2 * i] = eh[i]; hw20[2 * i + 1] = ew[i]; }
    return VSTAB_OK;
}

extern "C" size_t vstab_workspace_bytes(int B, int H, int W, int Cin)
{
    Plan pl;
    const int chunk = B >= 1 ? max_chunk(B, H, W, Cin) : 0;
    if (chunk < 1 || !make_plan(chunk, H, W, Cin, pl)) { fail(nullptr, VSTAB_E_SHAPE, "unsupported problem %dx%dx%dx%d", B, H, W, Cin); return 0; }
    return pl.total;
}

// ---- pinned plans (vstab.h): decisions of a batch of `batch` samples for every smaller batch
static PlanPin pin_of(const vstab_ctx *ctx)
{
    PlanPin pin;
    if (ctx) { pin.batch = ctx->plan_batch; pin.flags = ctx->plan_flags; }
    return pin;
}

// chunk size the forward processes a batch of B in: every tensor below 2 GiB, chunks equalised; under a pinned plan batch the
// chunk size of THAT batch (ragged chunks then share its decisions)
static int chunk_size(const PlanPin &pin, int B, int H, int W, int Cin)
{
    const int ref = pin.batch > 0 ? pin.batch : B;
    const int cmax = ref >= 1 ? max_chunk(ref, H, W, Cin) : 0;
    if (cmax < 1) return 0;
    const int nch = (ref + cmax - 1) / cmax;
    return std::min(B, (ref + nch - 1) / nch);
}

extern "C" int vstab_set_plan_batch(vstab_ctx *ctx, int batch)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "set_plan_batch: ctx is NULL");
    if (batch < 0) return fail(ctx, VSTAB_E_SHAPE, "set_plan_batch: batch must be >= 0 (0 = plan for the batch of each call)");
    ctx->plan_batch = batch;
    return VSTAB_OK;
}

extern "C" int vstab_set_plan_flags(vstab_ctx *ctx, unsigned flags)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "set_plan_flags: ctx is NULL");
    if (flags & ~(unsigned)(VSTAB_PLAN_NO_SKINNY | VSTAB_PLAN_NO_DUAL | VSTAB_PLAN_NO_TAIL | VSTAB_PLAN_NO_WDEC | VSTAB_PLAN_FORCE_WDEC)) return fail(ctx, VSTAB_E_SHAPE, "set_plan_flags: unknown flag bits 0x%x", flags);
    ctx->plan_flags = flags;
    return VSTAB_OK;
}

extern "C" size_t vstab_workspace_bytes_ctx(const vstab_ctx *ctx, int B, int H, int W, int Cin)
{
    Plan pl;
    const PlanPin pin = pin_of(ctx);
    if (pin.batch > 0 && B > pin.batch) { fail(nullptr, VSTAB_E_SHAPE, "batch %d exceeds the pinned plan batch %d", B, pin.batch); return 0; }
    const int chunk = B >= 1 ? chunk_size(pin, B, H, W, Cin) : 0;
    if (chunk < 1 || !make_plan(chunk, H, W, Cin, pl, &pin)) { fail(nullptr, VSTAB_E_SHAPE, "unsupported problem %dx%dx%dx%d", B, H, W, Cin); return 0; }
    return pl.total;
}

static int workspace_layout_of(const PlanPin *pin, int chunk, int H, int W, int Cin, vstab_ws_entry *entries, int max_entries)
{
    Plan pl;
    if (!entries || chunk < 1 || !make_plan(chunk, H, W, Cin, pl, pin))
        return fail(nullptr, VSTAB_E_SHAPE, "unsupported problem %dx%dx%dx%d", chunk, H, W, Cin);
    int n = 0;
    for (int b = 0; b < N_BUF && n < max_entries; ++b) {
        vstab_ws_entry &e = entries[n++];
        std::memset(&e, 0, sizeof e);
        std::snprintf(e.name, sizeof e.name, "%s", BUF_NAME[b]);
        e.offset_bytes = (int64_t)pl.off[b];
        e.n = chunk; e.h = pl.buf_h[b]; e.w = pl.buf_w[b]; e.c = pl.buf_c[b]; e.c_stride = pl.buf_cs[b];
        if (b == B_TICKETS || b == B_PARTIAL || b == B_WINO_V || b == B_WINO_M) { e.n = 1; e.h = 1; e.w = (int32_t)std::min<size_t>(pl.bytes[b] / 4, 0x7fffffff); e.c = 1; e.c_stride = 1; }
    }
    return n;
}

extern "C" int vstab_workspace_layout(int B, int H, int W, int Cin, vstab_ws_entry *entries, int max_entries)
{
    return workspace_layout_of(nullptr, B >= 1 ? max_chunk(B, H, W, Cin) : 0, H, W, Cin, entries, max_entries);     // the workspace holds one chunk of the batch
}

extern "C" int vstab_workspace_layout_ctx(const vstab_ctx *ctx, int B, int H, int W, int Cin, vstab_ws_entry *entries, int max_entries)
{
    const PlanPin pin = pin_of(ctx);
    if (pin.batch > 0 && B > pin.batch) return fail(nullptr, VSTAB_E_SHAPE, "batch %d exceeds the pinned plan batch %d", B, pin.batch);
    return workspace_layout_of(&pin, B >= 1 ? chunk_size(pin, B, H, W, Cin) : 0, H, W, Cin, entries, max_entries);
}

// ------------------------------------------------------------------------- host-only helpers
static const int LAYER_IN[19] = {-1, B_CONV1, B_CONCAT2, B_CONV3, B_CONCAT3, B_CONV4, B_CONCAT4, B_CONV5, B_CONCAT5, B_CONV6,
                                 B_CONV6_1, B_CONCAT5, B_CONCAT4, B_CONCAT3, B_CONCAT2, B_CONV6_1, B_CONCAT5, B_CONCAT4, B_CONCAT3};
static const int LAYER_OUT[19] = {B_CONV1, B_CONCAT2, B_CONV3, B_CONCAT3, B_CONV4, B_CONCAT4, B_CONV5, B_CONCAT5, B_CONV6,
                                  B_CONV6_1, B_CONCAT5, B_CONCAT4, B_CONCAT3, B_CONCAT2, B_T, B_T6, B_T5, B_T4, B_T3};

extern "C" int vstab_host_layer_plan(int B, int H, int W, int Cin, int layer, int32_t *out, int cap)
{
    return vstab_host_layer_plan_pinned(0, 0u, B, H, W, Cin, layer, out, cap);
}

extern "C" int vstab_host_layer_plan_pinned(int plan_batch, unsigned flags, int B, int H, int W, int Cin, int layer, int32_t *out, int cap)
{
    Plan pl;
    PlanPin pin; pin.batch = plan_batch; pin.flags = flags;
    if (!out || layer < 0 || layer > 18 || plan_batch < 0 || !make_plan(B, H, W, Cin, pl, &pin))
        return fail(nullptr, VSTAB_E_SHAPE, "layer_plan: bad arguments");
    const ConvParams &p = pl.cp[layer];
    const int need = 26 + 7 * p.nphase;
    if (cap < need) return fail(nullptr, VSTAB_E_NOMEM, "layer_plan: need %d ints", need);
    const int v[26] = {p.B, p.Hi, p.Wi, p.Cs_in, p.KH, p.NSEG, p.SEG, p.SEGP, p.SEG_STRIDE, p.s_in, p.s_out, p.Ho, p.Wo,
                       p.Cs_out, p.c_off, p.N, p.Npad, p.act, p.nphase, p.ksplit, p.Mmax, (int)pl.tile[layer],
                       (int)pl.vec4[layer], LAYER_IN[layer], LAYER_OUT[layer],
                       ((layer < 10 && pl.wino[layer]) || (layer >= 10 && layer < 14 && pl.wdec[layer - 10])) ? 1 : 0};
    for (int i = 0; i < 26; ++i) out[i] = v[i];
    for (int k = 0; k < p.nphase; ++k) {
        const ConvPhase &ph = p.ph[k];
        const int q[7] = {ph.Hg, ph.Wg, ph.M, ph.off_y, ph.off_x, ph.o_y, ph.o_x};
        for (int i = 0; i < 7; ++i) out[26 + 7 * k + i] = q[i];
    }
    return need;
}

// the 9-position GEMM of refinement level l's transposed convolution in Winograd F(2x2,2x2) form, whether or not the plan would choose
// it: the fields of vstab_host_layer_plan (26 + 7 per position) followed by the tile geometry {NTy, NTx, nty[3], ntx[3]}
extern "C" int vstab_host_wdec_plan(int B, int H, int W, int Cin, int l, int32_t *out, int cap)
{
    Plan pl;
    if (!out || l < 0 || l > 3 || !make_plan(B, H, W, Cin, pl)) return fail(nullptr, VSTAB_E_SHAPE, "wdec_plan: bad arguments");
    const ConvParams &d = pl.cp[10 + l];
    ConvParams p;
    const WdecGeom g = wdec_geom(d.Hi, d.Wi, d.Ho, d.Wo);
    fill_wdec_gemm(p, B, g, d.Cs_in, d.N);
    const int need = 26 + 7 * 9 + 8;
    if (cap < need) return fail(nullptr, VSTAB_E_NOMEM, "wdec_plan: need %d ints", need);
    const int v[26] = {p.B, p.Hi, p.Wi, p.Cs_in, p.KH, p.NSEG, p.SEG, p.SEGP, p.SEG_STRIDE, p.s_in, p.s_out, p.Ho, p.Wo,
                       p.Cs_out, p.c_off, p.N, p.Npad, p.act, p.nphase, p.ksplit, p.Mmax, (int)TILE_128x128, 1, LAYER_IN[10 + l], LAYER_OUT[10 + l],
                       pl.wdec[l] ? 1 : 0};
    for (int i = 0; i < 26; ++i) out[i] = v[i];
    for (int k = 0; k < 9; ++k) {
        const ConvPhase &ph = p.ph[k];
        const int q[7] = {ph.Hg, ph.Wg, ph.M, ph.off_y, ph.off_x, ph.o_y, ph.o_x};
        for (int i = 0; i < 7; ++i) out[26 + 7 * k + i] = q[i];
    }
    const int gg[8] = {g.NTy, g.NTx, g.nty[0], g.nty[1], g.nty[2], g.ntx[0], g.ntx[1], g.ntx[2]};
    for (int i = 0; i < 8; ++i) out[26 + 63 + i] = gg[i];
    return need;
}

extern "C" long long vstab_host_pack_wdec(int l, const float *W, const double *scale, float *wpk, long long cap)
{
    if (!W || !wpk || l < 0 || l > 3) return fail(nullptr, VSTAB_E_SHAPE, "pack_wdec: bad arguments");
    const int co = DEC_COUT[l];
    const long long n = 9LL * klayout_run(1, 1, DEC_CS_IN[l]).ktiles() * 4 * co * 32;
    if (cap < n) return fail(nullptr, VSTAB_E_NOMEM, "pack_wdec: need %lld floats", n);
    std::vector<double> ones;
    if (!scale) { ones.assign(co, 1.0); scale = ones.data(); }
    pack_wdec(W, scale, DEC_CIN[l], DEC_CS_IN[l], co, wpk);
    return n;
}

extern "C" long long vstab_host_pack_layer(int Cin, int layer, const float *W, const double *scale, float *wpk,
                                           long long cap)
{
    if (!W || !wpk || layer < 0 || layer > 18 || Cin < 1) return fail(nullptr, VSTAB_E_SHAPE, "pack_layer: bad arguments");
    std::vector<double> ones;
    if (layer < 10) {
        const Enc &e = ENC[layer];
        const int ci = layer == 0 ? Cin : ENC[layer - 1].cout, cs_in = layer == 0 ? Cin : ENC_IO[layer].cs_in;
        const int npad = round_up(e.cout, e.cout >= 128 ? 128 : 64);
        const KLayout L = enc_layout(layer, Cin);
        const long long n = (long long)L.ktiles() * npad * 32;
        if (cap < n) return fail(nullptr, VSTAB_E_NOMEM, "pack_layer: need %lld floats", n);
        if (!scale) { ones.assign(npad, 1.0); scale = ones.data(); }
        pack_conv(W, scale, e.k, e.k, ci, cs_in, e.cout, npad, L, wpk);
        return n;
    }
    if (layer < 14) {
        const int l = layer - 10, co = DEC_COUT[l], npad = round_up(co, co >= 128 ? 128 : 64);
        const long long n = 4LL * klayout_deconv(DEC_CS_IN[l]).ktiles() * npad * 32;
        if (cap < n) return fail(nullptr, VSTAB_E_NOMEM, "pack_layer: need %lld floats", n);
        if (!scale) { ones.assign(npad, 1.0); scale = ones.data(); }
        pack_deconv(W, scale, DEC_CIN[l], DEC_CS_IN[l], co, npad, wpk);
        return n;
    }
    const int tcin = layer == 14 ? 194 : PRED_CIN[layer - 15], tcs = layer == 14 ? 196 : PRED_CS[layer - 15];
    const long long n = (long long)klayout_run(1, 1, tcs).ktiles() * 32 * 32;
    if (cap < n) return fail(nullptr, VSTAB_E_NOMEM, "pack_layer: need %lld floats", n);
    pack_predict2_table(W, tcin, tcs, 32, wpk);
    return n;
}

// ------------------------------------------------------------------------- weights
extern "C" int vstab_load_weights(vstab_ctx *ctx, const vstab_tensor *t, int count)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "vstab_load_weights: ctx is NULL");
    if (!t || count <= 0) return fail(ctx, VSTAB_E_WEIGHTS, "vstab_load_weights: no tensors");
    const vstab_tensor *w1 = find(t, count, "1/W_conv2d");
    if (!w1 || w1->ndim != 4) return fail(ctx, VSTAB_E_WEIGHTS, "missing variable 1/W_conv2d");
    const int cin = w1->shape[2];
    if (cin < 1 || cin > 4096) return fail(ctx, VSTAB_E_WEIGHTS, "1/W_conv2d: bad Cin %d", cin);

    std::vector<float> host;
    auto reserve = [&](size_t n) { size_t o = (host.size() + 63) / 64 * 64; host.resize(o + n, 0.f); return o; };
    std::vector<double> scale;
    auto need = [&](const std::string &name, std::initializer_list<int> s) -> const vstab_tensor * {
        const vstab_tensor *x = find(t, count, name);
        return shape_is(x, s) ? x : nullptr;
    };
#define NEED(var, name, ...)                                                                   \
    const vstab_tensor *var = need(name, {__VA_ARGS__});                                        \
    if (!var) return fail(ctx, VSTAB_E_WEIGHTS, "missing or mis-shaped variable %s", std::string(name).c_str());

    // encoder
    for (int i = 0; i < 10; ++i) {
        const Enc &e = ENC[i];
        const int ci = i == 0 ? cin : ENC[i - 1].cout;
        const int cs_in = i == 0 ? cin : ENC_IO[i].cs_in;
        const std::string n = e.name;
        NEED(W, n + "/W_conv2d", e.k, e.k, ci, e.cout)
        NEED(b, n + "/b_conv2d", e.cout)
        NEED(beta, n + "/beta", e.cout)
        NEED(mean, n + "/moving_mean", e.cout)
        NEED(var, n + "/moving_variance", e.cout)
        const int BN = e.cout >= 128 ? 128 : 64, npad = round_up(e.cout, BN);
        const KLayout L = enc_layout(i, cin);
        scale.assign(npad, 1.0);
        ctx->enc_b[i] = reserve(npad);
        fold_bn(b->data, beta->data, mean->data, var->data, e.cout, npad, scale.data(), host.data() + ctx->enc_b[i]);
        ctx->enc_w[i] = reserve((size_t)L.ktiles() * npad * 32);
        pack_conv(W->data, scale.data(), e.k, e.k, ci, cs_in, e.cout, npad, L, host.data() + ctx->enc_w[i]);
        if (e.k == 3 && e.s == 1 && (ci & 31) == 0 && (e.cout & 127) == 0) {      // Winograd-domain operands (16 positions)
            ctx->wino_w[i] = reserve((size_t)16 * (ci / 32) * e.cout * 32);
            pack_winograd(W->data, scale.data(), ci, e.cout, e.cout, host.data() + ctx->wino_w[i]);
        }
        if (i == 0) {
            const int lead = rowwin_lead(-e.p, cin), segp = rowwin_segp(-e.p, e.k, cin);
            ctx->enc0_rw = reserve((size_t)e.k * (segp / 32) * npad * 32);
            pack_conv_rowwin(W->data, scale.data(), e.k, e.k, cin, e.cout, npad, lead, segp, host.data() + ctx->enc0_rw);
        }
    }
    // decoder
    for (int l = 0; l < 4; ++l) {
        const std::string n = DEC_NAME[l];
        const int co = DEC_COUT[l], ci = DEC_CIN[l], cs = DEC_CS_IN[l];
        NEED(W, n + "/W_deconv2d", 4, 4, co, ci)
        NEED(b, n + "/b_deconv2d", co)
        NEED(beta, n + "_bn/beta", co)
        NEED(mean, n + "_bn/moving_mean", co)
        NEED(var, n + "_bn/moving_variance", co)
        const int BN = co >= 128 ? 128 : 64, npad = round_up(co, BN);
        scale.assign(npad, 1.0);
        ctx->dec_b[l] = reserve(npad);
        fold_bn(b->data, beta->data, mean->data, var->data, co, npad, scale.data(), host.data() + ctx->dec_b[l]);
        ctx->dec_w[l] = reserve(4 * (size_t)klayout_deconv(cs).ktiles() * npad * 32);
        pack_deconv(W->data, scale.data(), ci, cs, co, npad, host.data() + ctx->dec_w[l]);
        ctx->wdec_w[l] = 0;
        if (l <= 2) {                 // Winograd F(2x2,2x2)-domain operands (9 positions x 4 phases) of the levels wdec_applies() can choose
            ctx->wdec_w[l] = reserve(9 * (size_t)klayout_run(1, 1, cs).ktiles() * 4 * co * 32);
            pack_wdec(W->data, scale.data(), ci, cs, co, host.data() + ctx->wdec_w[l]);
        }

        const std::string u = UP_NAME[l];
        NEED(uw, u + "/W_deconv2d", 4, 4, 2, 2)
        NEED(ub, u + "/b_deconv2d", 2)
        std::memcpy(ctx->up[l].w, uw->data, sizeof(float) * 64);
        ctx->up[l].b[0] = ub->data[0]; ctx->up[l].b[1] = ub->data[1];
    }
    // predict heads
    for (int l = 0; l < 4; ++l) {
        const std::string n = PRED_NAME[l];
        NEED(W, n + "/W_conv2d", 3, 3, PRED_CIN[l], 2)
        NEED(b, n + "/b_conv2d", 2)
        ctx->pred_w[l] = reserve((size_t)klayout_run(1, 1, PRED_CS[l]).ktiles() * 32 * 32);
        pack_predict2_table(W->data, PRED_CIN[l], PRED_CS[l], 32, host.data() + ctx->pred_w[l]);
        ctx->pred_b[l] = reserve(4);
        host[ctx->pred_b[l]] = b->data[0]; host[ctx->pred_b[l] + 1] = b->data[1];
    }
    {
        NEED(W, "predict2/W_conv2d", 3, 3, 194, 2)
        NEED(b, "predict2/b_conv2d", 2)
        ctx->tab_b = reserve(32);
        ctx->tab_wp = reserve((size_t)200 * 32);
        pack_predict2_panel(W->data, 194, 200, host.data() + ctx->tab_wp);
        ctx->zero_b = reserve(2048);             // zero bias for the Winograd-domain GEMMs (bias is added by the inverse transform)
        ctx->pred2_b = reserve(4);
        host[ctx->pred2_b] = b->data[0]; host[ctx->pred2_b + 1] = b->data[1];
    }
#undef NEED

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->dev_weights) { (void)hipFree(ctx->dev_weights); ctx->dev_weights = nullptr; }
    ctx->loaded = false;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->dev_weights), host.size() * sizeof(float));
    if (e != hipSuccess) return fail(ctx, VSTAB_E_NOMEM, "hipMalloc(%zu bytes of packed weights): %s", host.size() * 4, hipGetErrorString(e));
    HIP_TRY(ctx, hipMemcpy(ctx->dev_weights, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->dev_weight_floats = host.size();
    ctx->cin = cin;
    ctx->loaded = true;
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- forward
// evaluate_originalSize's tail riding in the forward: when given, the last launch of a chunk computes predict_flow2, the flow glue and
// tf_warp of the chunk's frames together (launch_pf2_glue_warp); `fused` reports whether every chunk could (else the caller warps)
struct FusedTail { const float *frame; float *outflow; float *warped; int oh, ow; bool fused; const uint8_t *frame8; uint8_t *out8; };      // fp32 frames, or the clip driver's 8-bit ones (frame8 / out8)
static int forward_chunk(vstab_ctx *ctx, const float *feats, int B, int H, int W, int Cin, float *pf6, float *pf5,
                         float *pf4, float *pf3, float *pf2, void *workspace, size_t workspace_bytes, void *stream_, FusedTail *tail);
static const char *conv_kernel_name(ConvTile t, bool vec4);
static const char *dual_kernel_name(ConvTile t);
static int forward_impl(vstab_ctx *ctx, const float *feats, int B, int H, int W, int Cin, float *pf6, float *pf5, float *pf4, float *pf3,
                        float *pf2, void *workspace, size_t workspace_bytes, void *stream_, FusedTail *tail);

extern "C" int vstab_flownets_forward(vstab_ctx *ctx, const float *feats, int B, int H, int W, int Cin, float *pf6,
                                      float *pf5, float *pf4, float *pf3, float *pf2, void *workspace,
                                      size_t workspace_bytes, void *stream_)
{
    return forward_impl(ctx, feats, B, H, W, Cin, pf6, pf5, pf4, pf3, pf2, workspace, workspace_bytes, stream_, nullptr);
}

static int forward_impl(vstab_ctx *ctx, const float *feats, int B, int H, int W, int Cin, float *pf6, float *pf5, float *pf4, float *pf3,
                        float *pf2, void *workspace, size_t workspace_bytes, void *stream_, FusedTail *tail)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "forward: ctx is NULL");
    if (!ctx->loaded) return fail(ctx, VSTAB_E_STATE, "forward: vstab_load_weights has not been called");
    if (Cin != ctx->cin) return fail(ctx, VSTAB_E_SHAPE, "forward: feats has %d channels, weights expect %d", Cin, ctx->cin);
    if (!feats || !pf6 || !pf5 || !pf4 || !pf3 || !pf2 || !workspace) return fail(ctx, VSTAB_E_STATE, "forward: NULL buffer");
    int eh[10], ew[10];
    if (B < 1 || !level_sizes(H, W, eh, ew)) return fail(ctx, VSTAB_E_SHAPE, "forward: unsupported problem %dx%dx%dx%d", B, H, W, Cin);
    const PlanPin pin = pin_of(ctx);
    if (pin.batch > 0 && B > pin.batch) return fail(ctx, VSTAB_E_SHAPE, "forward: batch %d exceeds the pinned plan batch %d (vstab_set_plan_batch)", B, pin.batch);
    // samples are independent: process the batch in (equalised) chunks that keep every tensor below
    // 2 GiB; equal chunks share one launch plan, so their results are bit-identical -- and so are ragged ones under a pinned plan batch
    const int chunk = chunk_size(pin, B, H, W, Cin);
    if (chunk < 1) return fail(ctx, VSTAB_E_SHAPE, "forward: one %dx%dx%d sample exceeds the 2 GiB tensor limit", H, W, Cin);
    // a batch processed in several chunks hands every chunk its slice of the frames: the slices keep the fused launch's 16-byte alignment
    // only when a frame is a whole number of 16-byte units (else: the two launches after the last chunk, as before)
    bool all_fused = tail != nullptr && (chunk >= B || ((size_t)tail->oh * tail->ow * 4) % 16 == 0);      // (8-bit frames: 4-byte units; the same test covers them)
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int bc = std::min(chunk, B - b0);
        FusedTail t{};
        if (tail && all_fused) {
            const size_t px = (size_t)b0 * tail->oh * tail->ow;
            t = FusedTail{tail->frame ? tail->frame + px * 3 : nullptr, tail->outflow ? tail->outflow + px * 2 : nullptr,
                          tail->warped ? tail->warped + px * 3 : nullptr, tail->oh, tail->ow, false,
                          tail->frame8 ? tail->frame8 + px * 3 : nullptr, tail->out8 ? tail->out8 + px * 3 : nullptr};
        }
        const int rc = forward_chunk(ctx, feats + (size_t)b0 * H * W * Cin, bc, H, W, Cin,
                                     pf6 + (size_t)b0 * eh[9] * ew[9] * 2, pf5 + (size_t)b0 * eh[7] * ew[7] * 2,
                                     pf4 + (size_t)b0 * eh[5] * ew[5] * 2, pf3 + (size_t)b0 * eh[3] * ew[3] * 2,
                                     pf2 + (size_t)b0 * (H - 2) * (W - 2) * 2, workspace, workspace_bytes, stream_, (tail && all_fused) ? &t : nullptr);
        if (rc != VSTAB_OK) return rc;
        // the first chunk decides (the geometry is the same for every chunk; a later chunk's frame slice could only differ in alignment,
        // and a whole number of frames keeps a 16-byte aligned base 16-byte aligned when oh*ow*12 is a multiple of 16 -- checked per chunk)
        if (tail && all_fused && !t.fused) {
            if (b0 != 0) return fail(ctx, VSTAB_E_ALIGN, "stabilise: chunk %d of the batch misses the fused tail's alignment", b0 / chunk);
            all_fused = false;
        }
    }
    if (tail) tail->fused = all_fused;
    return VSTAB_OK;
}

static int forward_chunk(vstab_ctx *ctx, const float *feats, int B, int H, int W, int Cin, float *pf6, float *pf5,
                         float *pf4, float *pf3, float *pf2, void *workspace, size_t workspace_bytes, void *stream_, FusedTail *tail)
{
    Plan pl;
    const PlanPin pin = pin_of(ctx);
    if (!make_plan(B, H, W, Cin, pl, &pin)) return fail(ctx, VSTAB_E_SHAPE, "forward: unsupported problem %dx%dx%dx%d", B, H, W, Cin);
    if (workspace_bytes < pl.total) return fail(ctx, VSTAB_E_NOMEM, "forward: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    if (((uintptr_t)workspace & 255) != 0) return fail(ctx, VSTAB_E_ALIGN, "forward: workspace must be 256-byte aligned");
    if (((uintptr_t)feats & 15) || ((uintptr_t)pf6 & 7) || ((uintptr_t)pf5 & 7) || ((uintptr_t)pf4 & 7) ||
        ((uintptr_t)pf3 & 7) || ((uintptr_t)pf2 & 7))
        return fail(ctx, VSTAB_E_ALIGN, "forward: feats must be 16-byte and flows 8-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    char *ws = (char *)workspace;
    auto buf = [&](int b) { return (float *)(ws + pl.off[b]); };
    const float *dw = ctx->dev_weights;
    // the weight-stream layers finish their split-K inside the launch by tickets (conv_skinny.hip).  The words live in THIS workspace and
    // are zeroed here: a launch leaves them zero, but the workspace is the caller's (first use, reuse of freed memory) and a launch that
    // failed mid-flight leaves them dirty
    unsigned *tickets = reinterpret_cast<unsigned *>(buf(B_TICKETS));
    bool any_tickets = false;
    for (int i = 0; i < 14; ++i) any_tickets = any_tickets || (pl.skinny[i] && pl.cp[i].ksplit > 1);
    bool tickets_cleared = !any_tickets;          // the first layer's launch clears them when it is the row-window kernel; else a memset node

    // optional per-launch events
    hipEvent_t *ev = nullptr;
    if (ctx->prof) {
        const size_t need = (size_t)(ctx->prof_forwards + 1) * 30;
        while (ctx->prof_ev.size() < need) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->prof_ev.push_back(e);
        }
        ev = ctx->prof_ev.data() + (size_t)ctx->prof_forwards * 30;
        for (int i = 0; i < 15; ++i) {
            const ConvParams &p = pl.cp[i];
            double mac = 0;
            if (i < 10 && pl.wino[i]) mac = 16.0 * pl.wcp[i].Mmax * ENC[i - 1].cout * p.N;      // MACs the Winograd-domain GEMM issues (4/9 of direct)
            else if (i < 10) mac = (double)p.ph[0].M * ENC[i].k * ENC[i].k * (i == 0 ? Cin : ENC[i - 1].cout) * p.N;
            else if (i < 14 && pl.wdec[i - 10]) {                          // MACs the 9-position GEMM issues (9/16 of direct, plus the ragged tile grid)
                for (int k = 0; k < 9; ++k) mac += (double)pl.wdcp[i - 10].ph[k].M * DEC_CIN[i - 10] * 4.0 * p.N;
            }
            else if (i < 14) mac = (double)B * p.Ho * p.Wo * 4.0 * DEC_CIN[i - 10] * p.N;
            else mac = (double)p.ph[0].M * 194.0 * 18.0;
            ctx->prof_flops[i] += 2.0 * mac;
            double dmac = mac;                           // the same layer as a direct convolution (SURVEY.md 8d's accounting)
            if (i < 10 && pl.wino[i]) dmac = (double)p.ph[0].M * 9.0 * ENC[i - 1].cout * p.N;
            if (i >= 10 && i < 14 && pl.wdec[i - 10]) dmac = (double)B * p.Ho * p.Wo * 4.0 * DEC_CIN[i - 10] * p.N;
            ctx->prof_flops_direct[i] += 2.0 * dmac;
        }
    }
    // (the context is written only while profiling: plain forwards on one context may be issued from several host threads)
#define PROF_NAME(slot, name) do { if (ev) ctx->prof_kernel[slot] = (name); } while (0)
#define EV_A(slot) (ev ? ev[2 * (slot)] : nullptr)
#define EV_B(slot) (ev ? ev[2 * (slot) + 1] : nullptr)

    static const char *const ENC_RANGE[10] = {"conv1", "conv2", "conv3", "conv3_1", "conv4", "conv4_1", "conv5", "conv5_1", "conv6", "conv6_1"};
    static const char *const DEC_RANGE[4] = {"deconv5", "deconv4", "deconv3", "deconv2"};
    static const char *const HEAD_RANGE[4] = {"predict_flow6+upsample6_5", "predict_flow5+upsample5_4", "predict_flow4+upsample4_3", "predict_flow3+upsample3_2"};
    TraceRange whole_range("flownetS_pyramid");
    // encoder (model.py:807-844)
    for (int i = 0; i < 10; ++i) {
        TraceRange layer_range(ENC_RANGE[i]);
        ConvParams p = pl.cp[i];
        if (i == 0) {           // first layer: row-window kernel when its alignment conditions hold
            RowWinParams r{};
            r.in = feats; r.out = buf(B_CONV1); r.wpk = dw + ctx->enc0_rw; r.bias = dw + ctx->enc_b[0];
            const long long in_bytes = (long long)B * H * W * Cin * 4;
            r.in_bytes = (unsigned)std::min<long long>(in_bytes, 0xFFFFFFFFLL);
            r.B = B; r.Hi = H; r.Wi = W; r.Cs_in = Cin; r.KH = ENC[0].k;
            r.SEGP = rowwin_segp(-ENC[0].p, ENC[0].k, Cin);
            r.s_in = ENC[0].s; r.off_y = -ENC[0].p;
            r.e_off = -ENC[0].p * Cin - rowwin_lead(-ENC[0].p, Cin);
            r.w_a = ((r.e_off % 4) + 4) % 4;
            r.MB = rowwin_mb(B, p.Ho, p.Wo);
            r.WLEN = round_up(r.s_in * Cin * (64 * r.MB - 1) + r.w_a + r.SEGP, 4);
            r.Ho = p.Ho; r.Wo = p.Wo; r.Cs_out = p.Cs_out; r.c_off = 0; r.N = p.N; r.Npad = p.Npad; r.act = 1;
            if (in_bytes < 0x80000000LL && rowwin_applicable(r)) {
                if (!tickets_cleared) { r.clear_words = tickets; r.clear_n = SKINNY_MAX_TILES; tickets_cleared = true; }
                const int rem = p.Wo % 128;
                if (r.MB == 2 && p.Wo > 128 && rem >= 1 && rem <= 64) {
                    // 128 k + (1..64) columns: k full tiles, then the rest as ONE 64-pixel tile (second launch; events span both)
                    RowWinParams t = r;
                    r.ntile_x = p.Wo / 128;
                    t.MB = 1; t.ox_base = r.ntile_x * 128; t.ntile_x = 1; t.clear_n = 0;
                    t.WLEN = round_up(t.s_in * Cin * 63 + t.w_a + t.SEGP, 4);
                    if (rowwin_applicable(t)) {
                        HIP_TRY(ctx, launch_conv_rowwin(r, stream, EV_A(0), nullptr));
                        HIP_TRY(ctx, launch_conv_rowwin(t, stream, nullptr, EV_B(0)));
                        PROF_NAME(0, "conv_rowwin_kernel<7, 2> + <4, 1> tail");
                        continue;
                    }
                    r.ntile_x = 0;
                }
                HIP_TRY(ctx, launch_conv_rowwin(r, stream, EV_A(0), EV_B(0)));
                PROF_NAME(0, r.MB == 2 ? "conv_rowwin_kernel<7, 2>" : "conv_rowwin_kernel<4, 1>");
                continue;
            }
        }
        if (!tickets_cleared) { HIP_TRY(ctx, hipMemsetAsync(tickets, 0, pl.bytes[B_TICKETS], stream)); tickets_cleared = true; }
        if (pl.wino[i]) {       // transform, 16-position GEMM on the MFMA kernel, inverse transform (+ bias, leaky relu)
            ConvParams q = pl.wcp[i];
            const int cin_i = ENC[i - 1].cout;
            HIP_TRY(ctx, launch_wino_input(buf(ENC_IO[i].in_buf), B, pl.eh[i], pl.ew[i], ENC_IO[i].cs_in, 0, cin_i, buf(B_WINO_V), stream));
            q.in = buf(B_WINO_V); q.out = buf(B_WINO_M);
            q.wpk = dw + ctx->wino_w[i]; q.bias = dw + ctx->zero_b; q.partial = buf(B_PARTIAL);
            const int T_i = ((pl.eh[i] + 1) / 2) * ((pl.ew[i] + 1) / 2);
            const int P_i = wino_gemm_stream_positions(B, T_i, cin_i, ENC[i].cout);
            if (P_i > 0) {       // streams of positions (wino_gemm_stream.hip): B=8 512x512 conv3_1 (8 positions per workgroup), conv4_1 (4)
                HIP_TRY(ctx, launch_wino_gemm_stream(q.in, q.wpk, q.out, B, T_i, cin_i, ENC[i].cout, P_i, stream, EV_A(i), EV_B(i)));
                PROF_NAME(i, "wino_gemm_stream_kernel");
            } else {
                HIP_TRY(ctx, launch_conv(q, pl.wtile[i], true, stream, EV_A(i), EV_B(i)));
                PROF_NAME(i, conv_kernel_name(pl.wtile[i], true));
            }
            HIP_TRY(ctx, launch_wino_output(buf(B_WINO_M), B, pl.eh[i], pl.ew[i], ENC[i].cout, dw + ctx->enc_b[i], 1, buf(ENC_IO[i].out_buf),
                                            ENC_IO[i].cs_out, 0, stream));
            continue;
        }
        p.in = ENC_IO[i].in_buf < 0 ? feats : buf(ENC_IO[i].in_buf);
        p.out = buf(ENC_IO[i].out_buf);
        p.wpk = dw + ctx->enc_w[i];
        p.bias = dw + ctx->enc_b[i];
        p.partial = buf(B_PARTIAL);
        if (pl.skinny[i]) {
            HIP_TRY(ctx, launch_conv_skinny(p, tickets, stream, EV_A(i), EV_B(i)));
            PROF_NAME(i, "conv_skinny_kernel<1, 4>");
            continue;
        }
        HIP_TRY(ctx, launch_conv(p, pl.tile[i], pl.vec4[i], stream, EV_A(i), EV_B(i)));
        PROF_NAME(i, conv_kernel_name(pl.tile[i], pl.vec4[i]));
    }
    // decoder (model.py:847-880)
    float *pfs[5] = {pf6, pf5, pf4, pf3, pf2};
    const int cat_buf[4] = {B_CONCAT5, B_CONCAT4, B_CONCAT3, B_CONCAT2};
    const int lvl_enc[5] = {9, 7, 5, 3, 1};              // encoder stage giving each level's size
    const int tab_src[4] = {B_CONV6_1, B_CONCAT5, B_CONCAT4, B_CONCAT3}, tab_dst[4] = {B_T6, B_T5, B_T4, B_T3};
    // One refinement level = its flow head (model.py:847-848 ...: 3x3 -> 2 conv as a tap-table GEMM whose split-K slabs stay uncombined,
    // then predict_up: slab sum, tap gather, fold with the upsampled coarser flow, upsample_flowN into the next concat's flow channels)
    // and its transposed convolution (model.py:850-851 ...).  Both read the SAME tensor and neither needs the other, so they run as
    // TWO launches instead of four: conv_dual_kernel (deconv tiles + tap-table tiles side by side), then combine_predict_up_kernel
    // (the deconv's split-K combine + predict_up side by side).  For one sample every one of the four was little more than a
    // launch's fixed latency.  VSTAB_PLAN_NO_DUAL restores the four-launch sequence (A/B; same arithmetic, same bits).
    for (int l = 0; l < 4; ++l) {
        const float *prev = l == 0 ? nullptr : pfs[l - 1];
        const int ph_ = l == 0 ? 0 : pl.eh[lvl_enc[l - 1]], pw_ = l == 0 ? 0 : pl.ew[lvl_enc[l - 1]];      // the coarser level's size
        const int oh = pl.eh[lvl_enc[l + 1]], ow = pl.ew[lvl_enc[l + 1]];                                  // the finer level the flow is upsampled to
        ConvParams pd = pl.cp[10 + l], pt = pl.cp[15 + l];
        pd.in = buf(l == 0 ? B_CONV6_1 : cat_buf[l - 1]); pd.out = buf(cat_buf[l]);
        pd.wpk = dw + ctx->dec_w[l]; pd.bias = dw + ctx->dec_b[l]; pd.partial = buf(B_PARTIAL);
        const size_t dec_slab = pd.ksplit > 1 ? (size_t)pd.nphase * pd.ksplit * pd.Mmax * pd.Npad : 0;   // the tap table's slabs sit behind the deconv's
        pt.in = buf(tab_src[l]); pt.out = buf(tab_dst[l]);
        pt.wpk = dw + ctx->pred_w[l]; pt.bias = dw + ctx->tab_b; pt.partial = buf(B_PARTIAL) + dec_slab;
        const float *tsrc = pt.ksplit > 1 ? pt.partial : pt.out;
        const bool fuse = !(pin.flags & VSTAB_PLAN_NO_DUAL) && !pl.skinny[10 + l];
        if (fuse && pl.wdec[l]) {
            // Winograd F(2x2,2x2): input transform, the 9-position GEMM beside the level's tap-table tiles, inverse transform (+ bias, leaky
            // relu) into the concat slice; predict_up has no slabs of the transposed convolution to sum
            {
                TraceRange r2(DEC_RANGE[l]);
                const WdecGeom &g = pl.wdg[l];
                HIP_TRY(ctx, launch_wdec_input(pd.in, B, pd.Hi, pd.Wi, pd.Cs_in, buf(B_WINO_V), g, stream));
                ConvParams q = pl.wdcp[l];
                q.in = buf(B_WINO_V); q.out = buf(B_WINO_M); q.wpk = dw + ctx->wdec_w[l]; q.bias = dw + ctx->zero_b; q.partial = buf(B_PARTIAL);
                pt.partial = buf(B_PARTIAL);
                const hipError_t e = launch_conv_dual(q, pl.wdtile[l], pt, pl.tile[15 + l], stream, EV_A(10 + l), EV_B(10 + l));
                if (e == hipErrorNotSupported) {
                    HIP_TRY(ctx, launch_conv(pt, pl.tile[15 + l], true, stream, nullptr, nullptr, false));
                    HIP_TRY(ctx, launch_conv(q, pl.wdtile[l], true, stream, EV_A(10 + l), EV_B(10 + l), false));
                    PROF_NAME(10 + l, conv_kernel_name(pl.wdtile[l], true));
                } else {
                    HIP_TRY(ctx, e);
                    PROF_NAME(10 + l, dual_kernel_name(pl.wdtile[l]));
                }
            }
            // the inverse transform (+ bias, leaky relu) shares its launch with predict_up: different channel slices of the same concat
            TraceRange r3(HEAD_RANGE[l]);
            const float *tsrc2 = pt.ksplit > 1 ? pt.partial : pt.out;
            const WdecOutArgs wo{buf(B_WINO_M), pd.N / 4, dw + ctx->dec_b[l], 1, buf(cat_buf[l]), pd.Ho, pd.Wo, pd.Cs_out, pd.c_off, pl.wdg[l]};
            HIP_TRY(ctx, launch_predict_up(tsrc2, pt.ksplit, (long long)pt.Mmax * pt.Npad, B, pt.Hi, pt.Wi, dw + ctx->pred_b[l], prev, ph_, pw_,
                                           pfs[l], ctx->up[l], buf(cat_buf[l]), oh, ow, CONCAT_CS[l], CONCAT_C[l] - 2, stream, nullptr, &wo));
            continue;
        }
        if (fuse) {
            {
                TraceRange r2(DEC_RANGE[l]);
                const hipError_t e = launch_conv_dual(pd, pl.tile[10 + l], pt, pl.tile[15 + l], stream, EV_A(10 + l), EV_B(10 + l));
                if (e == hipErrorNotSupported) {        // a tile shape the two-problem kernel is not built for: one launch each, the combine still rides with predict_up
                    HIP_TRY(ctx, launch_conv(pt, pl.tile[15 + l], true, stream, nullptr, nullptr, false));
                    HIP_TRY(ctx, launch_conv(pd, pl.tile[10 + l], true, stream, EV_A(10 + l), EV_B(10 + l), false));
                    PROF_NAME(10 + l, conv_kernel_name(pl.tile[10 + l], true));
                } else {
                    HIP_TRY(ctx, e);
                    PROF_NAME(10 + l, dual_kernel_name(pl.tile[10 + l]));
                }
            }
            TraceRange r3(HEAD_RANGE[l]);
            HIP_TRY(ctx, launch_predict_up(tsrc, pt.ksplit, (long long)pt.Mmax * pt.Npad, B, pt.Hi, pt.Wi, dw + ctx->pred_b[l], prev, ph_, pw_,
                                           pfs[l], ctx->up[l], buf(cat_buf[l]), oh, ow, CONCAT_CS[l], CONCAT_C[l] - 2, stream, &pd));
            continue;
        }
        {
            TraceRange head_range(HEAD_RANGE[l]);
            HIP_TRY(ctx, launch_conv(pt, pl.tile[15 + l], true, stream, nullptr, nullptr, false));
            HIP_TRY(ctx, launch_predict_up(tsrc, pt.ksplit, (long long)pt.Mmax * pt.Npad, B, pt.Hi, pt.Wi, dw + ctx->pred_b[l], prev, ph_, pw_,
                                           pfs[l], ctx->up[l], buf(cat_buf[l]), oh, ow, CONCAT_CS[l], CONCAT_C[l] - 2, stream));
        }
        TraceRange layer_range(DEC_RANGE[l]);
        if (pl.skinny[10 + l]) {
            HIP_TRY(ctx, launch_conv_skinny(pd, tickets, stream, EV_A(10 + l), EV_B(10 + l)));
            PROF_NAME(10 + l, "conv_skinny_kernel<1, 4>");
        } else {
            HIP_TRY(ctx, launch_conv(pd, pl.tile[10 + l], true, stream, EV_A(10 + l), EV_B(10 + l)));
            PROF_NAME(10 + l, conv_kernel_name(pl.tile[10 + l], true));
        }
    }
    // full-resolution head (model.py:882-887)
    {
        TraceRange layer_range("predict_flow2");
        ConvParams p = pl.cp[14];
        p.in = buf(B_CONCAT2); p.out = buf(B_T);
        const long long M2 = (long long)B * pl.eh[1] * pl.ew[1];
        if (!tap_panel_applicable(M2, p.Cs_in, p.in, p.out)) return fail(ctx, VSTAB_E_SHAPE, "predict_flow2 tap table: unsupported geometry");
        HIP_TRY(ctx, launch_tap_panel(p.in, M2, dw + ctx->tab_wp, p.out, stream, EV_A(14), EV_B(14)));
        PROF_NAME(14, "tap_panel_kernel");
        hipError_t te = hipErrorNotSupported;
        if (tail && !(pin.flags & VSTAB_PLAN_NO_TAIL)) {       // gather + glue + warp of this chunk's frames in one launch, when the geometry allows
            TraceRange tail_range("predict_flow2 gather+flow_glue+tf_warp");
            if (tail->frame8)
                te = launch_pf2_glue_warp_u8(buf(B_T), B, pl.eh[1], pl.ew[1], dw + ctx->pred2_b, pf3, pl.eh[3], pl.ew[3], pf2, H, W, tail->frame8,
                                             tail->outflow, tail->out8, tail->oh, tail->ow, stream);
            else
                te = launch_pf2_glue_warp(buf(B_T), B, pl.eh[1], pl.ew[1], dw + ctx->pred2_b, pf3, pl.eh[3], pl.ew[3], pf2, H, W, tail->frame,
                                          tail->outflow, tail->warped, tail->oh, tail->ow, stream);
            if (te != hipSuccess && te != hipErrorNotSupported) HIP_TRY(ctx, te);
            tail->fused = te == hipSuccess;
        }
        if (te != hipSuccess) HIP_TRY(ctx, launch_pf2(buf(B_T), B, pl.eh[1], pl.ew[1], dw + ctx->pred2_b, pf3, pl.eh[3], pl.ew[3], pf2, H, W, stream));
    }
#undef EV_A
#undef EV_B
#undef PROF_NAME
    if (ev) ctx->prof_forwards++;
    return VSTAB_OK;
}

// (string literals: the profiler's name slots are plain pointers, nothing a forward does allocates)
static const char *conv_kernel_name(ConvTile t, bool vec4)
{
    const bool dma = conv_uses_lds_dma(t, vec4);
#define VSTAB_KN(shape) (vec4 ? (dma ? "conv_mfma_kernel<" shape ", true, true>" : "conv_mfma_kernel<" shape ", true, false>") \
                              : (dma ? "conv_mfma_kernel<" shape ", false, true>" : "conv_mfma_kernel<" shape ", false, false>"))
    switch (t) {
    case TILE_128x128: return VSTAB_KN("128, 128, 2, 2");
    case TILE_128x64: return VSTAB_KN("128, 64, 2, 2");
    case TILE_64x128: return VSTAB_KN("64, 128, 1, 4");
    case TILE_64x64: return VSTAB_KN("64, 64, 2, 2");
    case TILE_256x32: return VSTAB_KN("256, 32, 4, 1");
    default: return VSTAB_KN("128, 32, 4, 1");
    }
#undef VSTAB_KN
}

static const char *dual_kernel_name(ConvTile t)
{
    const bool dma = conv_uses_lds_dma(t, true);
#define VSTAB_DN(shape) (dma ? "conv_dual_kernel: conv_mfma_kernel<" shape ", true, true> + <128, 32> tap table" \
                             : "conv_dual_kernel: conv_mfma_kernel<" shape ", true, false> + <128, 32> tap table")
    switch (t) {
    case TILE_128x128: return VSTAB_DN("128, 128, 2, 2");
    case TILE_128x64: return VSTAB_DN("128, 64, 2, 2");
    case TILE_64x128: return VSTAB_DN("64, 128, 1, 4");
    case TILE_64x64: return VSTAB_DN("64, 64, 2, 2");
    case TILE_256x32: return VSTAB_DN("256, 32, 4, 1");
    default: return VSTAB_DN("128, 32, 4, 1");
    }
#undef VSTAB_DN
}

// ------------------------------------------------------------------------- profiling
extern "C" int vstab_profile_kernel_name(vstab_ctx *ctx, int slot, char *buf, int cap)
{
    if (!ctx || !buf || cap < 1 || slot < 0 || slot > 14) return fail(ctx, VSTAB_E_STATE, "profile_kernel_name: bad argument");
    std::snprintf(buf, (size_t)cap, "%s", ctx->prof_kernel[slot] ? ctx->prof_kernel[slot] : "");
    return VSTAB_OK;
}

extern "C" int vstab_profile_enable(vstab_ctx *ctx, int enable)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "profile_enable: ctx is NULL");
    ctx->prof = enable != 0;
    return VSTAB_OK;
}

extern "C" int vstab_profile_reset(vstab_ctx *ctx)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "profile_reset: ctx is NULL");
    ctx->prof_forwards = 0;
    for (double &f : ctx->prof_flops) f = 0;
    for (double &f : ctx->prof_flops_direct) f = 0;
    return VSTAB_OK;
}

extern "C" int vstab_profile_read(vstab_ctx *ctx, double *ms_sum15, double *flops15, int *n_forwards)
{
    if (!ctx || !ms_sum15 || !flops15 || !n_forwards) return fail(ctx, VSTAB_E_STATE, "profile_read: NULL argument");
    for (int i = 0; i < 15; ++i) { ms_sum15[i] = 0; flops15[i] = ctx->prof_flops[i]; }
    for (int f = 0; f < ctx->prof_forwards; ++f)
        for (int i = 0; i < 15; ++i) {
            float ms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[(size_t)f * 30 + 2 * i], ctx->prof_ev[(size_t)f * 30 + 2 * i + 1]));
            ms_sum15[i] += ms;
        }
    *n_forwards = ctx->prof_forwards;
    return VSTAB_OK;
}

extern "C" int vstab_profile_read_direct(vstab_ctx *ctx, double *flops15)
{
    if (!ctx || !flops15) return fail(ctx, VSTAB_E_STATE, "profile_read_direct: NULL argument");
    for (int i = 0; i < 15; ++i) flops15[i] = ctx->prof_flops_direct[i];
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- glue + warp
extern "C" int vstab_flow_resize_scale(const float *flow, int B, int h, int w, float *out, int oh, int ow, int net_h, int net_w,
                                       void *stream)
{
    if (!flow || !out) return fail(nullptr, VSTAB_E_STATE, "flow_resize_scale: NULL buffer");
    if (B < 1 || h < 1 || w < 1 || oh < 1 || ow < 1 || net_h < 1 || net_w < 1) return fail(nullptr, VSTAB_E_SHAPE, "flow_resize_scale: bad shape");
    if (((uintptr_t)flow & 7) || ((uintptr_t)out & 7)) return fail(nullptr, VSTAB_E_ALIGN, "flow_resize_scale: 8-byte alignment");
    HIP_TRY(nullptr, launch_flow_resize_scale(flow, B, h, w, out, oh, ow, net_h, net_w, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_selftest_div_const(float d, unsigned first_bits, unsigned long long count, unsigned long long *bad_count_dev, void *stream)
{
    if (!bad_count_dev) return fail(nullptr, VSTAB_E_STATE, "selftest_div_const: NULL counter");
    if (count < 1 || count > (1ull << 32)) return fail(nullptr, VSTAB_E_SHAPE, "selftest_div_const: 1 <= count <= 2^32");
    const hipError_t e = launch_div_const_selftest(d, first_bits, count, bad_count_dev, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return fail(nullptr, VSTAB_E_SHAPE, "selftest_div_const: the glue takes the plain division for this divisor");
    HIP_TRY(nullptr, e);
    return VSTAB_OK;
}

extern "C" int vstab_resize_bilinear(const float *x, int B, int h, int w, int C, float *out, int oh, int ow, void *stream)
{
    if (!x || !out) return fail(nullptr, VSTAB_E_STATE, "resize_bilinear: NULL buffer");
    if (B < 1 || h < 1 || w < 1 || C < 1 || oh < 1 || ow < 1) return fail(nullptr, VSTAB_E_SHAPE, "resize_bilinear: bad shape");
    HIP_TRY(nullptr, launch_resize_bilinear(x, B, h, w, C, out, oh, ow, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_resize_bilinear_slice3(const float *x, int B, int h, int w, int Cs, int c_off, float *out, int oh, int ow, void *stream)
{
    if (!x || !out) return fail(nullptr, VSTAB_E_STATE, "resize_bilinear_slice3: NULL buffer");
    if (B < 1 || h < 1 || w < 1 || oh < 1 || ow < 1 || Cs < 3 || c_off < 0 || c_off + 3 > Cs)
        return fail(nullptr, VSTAB_E_SHAPE, "resize_bilinear_slice3: bad shape");
    if ((uintptr_t)out & 15) return fail(nullptr, VSTAB_E_ALIGN, "resize_bilinear_slice3: out must be 16-byte aligned");
    const hipError_t e = launch_resize_bilinear_slice3(x, B, h, w, Cs, c_off, out, oh, ow, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return fail(nullptr, VSTAB_E_SHAPE, "resize_bilinear_slice3: problem too large");
    HIP_TRY(nullptr, e);
    return VSTAB_OK;
}

extern "C" int vstab_warp_flow(const float *img, const float *flow, float *out, int B, int H, int W, int C, void *stream)
{
    if (!img || !flow || !out) return fail(nullptr, VSTAB_E_STATE, "warp_flow: NULL buffer");
    if (B < 1 || H < 1 || W < 1 || C < 1) return fail(nullptr, VSTAB_E_SHAPE, "warp_flow: bad shape");
    if ((uintptr_t)flow & 7) return fail(nullptr, VSTAB_E_ALIGN, "warp_flow: flow must be 8-byte aligned");
    TraceRange range("tf_warp");
    HIP_TRY(nullptr, launch_warp_flow(img, flow, out, B, H, W, C, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_flow_glue_warp(const float *flow, int B, int h, int w, const float *img, float *outflow, float *warped, int oh,
                                    int ow, int C, int net_h, int net_w, void *stream)
{
    if (!flow || !img || !warped) return fail(nullptr, VSTAB_E_STATE, "flow_glue_warp: NULL buffer");
    if (B < 1 || h < 1 || w < 1 || oh < 1 || ow < 1 || net_h < 1 || net_w < 1) return fail(nullptr, VSTAB_E_SHAPE, "flow_glue_warp: bad shape");
    if (C != 3 || w < 2) return fail(nullptr, VSTAB_E_SHAPE, "flow_glue_warp: C must be 3 and w >= 2 (use flow_resize_scale + warp_flow otherwise)");
    if ((long long)B * oh * ow >= (1ll << 31)) return fail(nullptr, VSTAB_E_SHAPE, "flow_glue_warp: B*oh*ow must be < 2^31");
    if (((uintptr_t)flow & 7) || (((uintptr_t)img | (uintptr_t)outflow | (uintptr_t)warped) & 15))
        return fail(nullptr, VSTAB_E_ALIGN, "flow_glue_warp: flow 8-byte, img/outflow/warped 16-byte alignment");
    TraceRange range("flow_glue+tf_warp");
    HIP_TRY(nullptr, launch_flow_glue_warp(flow, B, h, w, img, outflow, warped, oh, ow, C, net_h, net_w, (hipStream_t)stream));
    return VSTAB_OK;
}

// evaluate_originalSize's whole graph (main:491-514) behind ONE call: the network, then the flow glue + tf_warp launch.
extern "C" int vstab_stabilise_originalsize(vstab_ctx *ctx, const float *feats, int B, int H, int W, int Cin, const float *frame, int oh,
                                            int ow, float *pf6, float *pf5, float *pf4, float *pf3, float *pf2, float *outflow,
                                            float *warped, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!frame || !warped) return fail(ctx, VSTAB_E_STATE, "stabilise_originalsize: NULL buffer");
    if (oh < 1 || ow < 1) return fail(ctx, VSTAB_E_SHAPE, "stabilise_originalsize: bad output size");
    // the tail (predict_flow2's gather, the glue, tf_warp) rides in the forward's last launch when its geometry allows (flow_ops.hip)
    FusedTail tail{frame, outflow, warped, oh, ow, false, nullptr, nullptr};
    const bool try_fused = (((uintptr_t)frame | (uintptr_t)warped | (uintptr_t)outflow) & 15) == 0;
    const int rc = forward_impl(ctx, feats, B, H, W, Cin, pf6, pf5, pf4, pf3, pf2, workspace, workspace_bytes, stream, try_fused ? &tail : nullptr);
    if (rc != VSTAB_OK) return rc;
    if (tail.fused) return VSTAB_OK;
    const int rc2 = vstab_flow_glue_warp(pf2, B, H - 2, W - 2, frame, outflow, warped, oh, ow, 3, H, W, stream);
    if (rc2 != VSTAB_OK) adopt_last_error(ctx);
    return rc2;
}

#if defined(VSTAB_HARNESS) && defined(VSTAB_STAMP)
// diagnostic build only (scripts/insitu_stamps.py): per-workgroup s_memtime stamps of the conv launches of a forward
namespace vstab { void conv_stamp_reset(); hipError_t conv_read_stamps_slot(int slot, unsigned long long *host, size_t n); }
extern "C" __attribute__((visibility("default"))) int vstab_debug_stamp_reset(void) { vstab::conv_stamp_reset(); return 0; }
extern "C" __attribute__((visibility("default"))) int vstab_debug_stamp_read(int slot, unsigned long long *host, size_t n)
{
    return vstab::conv_read_stamps_slot(slot, host, n) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int vstab_hbm_profile_enable(int mode)
{
    if (mode < 0 || mode > 2) return fail(nullptr, VSTAB_E_SHAPE, "hbm_profile_enable: mode must be 0, 1 or 2");
    hbm_profile_enable(mode);
    return VSTAB_OK;
}

extern "C" int vstab_hbm_profile_read(int slot, double *ms_sum, int *launches, double *alg_bytes_sum)
{
    if (!ms_sum || !launches || !alg_bytes_sum) return fail(nullptr, VSTAB_E_STATE, "hbm_profile_read: NULL argument");
    if (slot < 0 || slot >= HBM_SLOTS) return fail(nullptr, VSTAB_E_SHAPE, "hbm_profile_read: slot out of range");
    HIP_TRY(nullptr, hbm_profile_read(slot, ms_sum, launches, alg_bytes_sum));
    return VSTAB_OK;
}

extern "C" int vstab_get_pixel_value(const float *img, const int32_t *x, const int32_t *y, float *out, int B, int H,
                                     int W, int C, int Hi, int Wi, void *stream)
{
    if (!img || !x || !y || !out) return fail(nullptr, VSTAB_E_STATE, "get_pixel_value: NULL buffer");
    if (B < 1 || H < 1 || W < 1 || C < 1 || Hi < 1 || Wi < 1) return fail(nullptr, VSTAB_E_SHAPE, "get_pixel_value: bad shape");
    HIP_TRY(nullptr, launch_get_pixel_value(img, x, y, out, B, H, W, C, Hi, Wi, (hipStream_t)stream));
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- secondary samplers
extern "C" int vstab_st_transform(const float *img, int B, int H, int W, int C, const float *theta, int theta_dim,
                                  float *out, int oh, int ow, void *stream)
{
    if (!img || !theta || !out) return fail(nullptr, VSTAB_E_STATE, "st_transform: NULL buffer");
    if (B < 1 || H < 1 || W < 1 || C < 1 || oh < 1 || ow < 1 || (theta_dim != 6 && theta_dim != 8))
        return fail(nullptr, VSTAB_E_SHAPE, "st_transform: bad shape (theta must be [B,6] or [B,8])");
    HIP_TRY(nullptr, launch_st_transform(img, B, H, W, C, theta, theta_dim, out, oh, ow, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_st_bilinear_interp(const float *img, int B, int H, int W, int C, const float *x, const float *y,
                                        int oh, int ow, float *out, void *stream)
{
    if (!img || !x || !y || !out) return fail(nullptr, VSTAB_E_STATE, "st_bilinear_interp: NULL buffer");
    if (B < 1 || H < 1 || W < 1 || C < 1 || oh < 1 || ow < 1 || (long long)oh * ow > 0x7fffffffLL)
        return fail(nullptr, VSTAB_E_SHAPE, "st_bilinear_interp: bad shape");
    HIP_TRY(nullptr, launch_st_interp(img, B, H, W, C, x, y, oh, ow, out, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_st_meshgrid(float *out, int oh, int ow, void *stream)
{
    if (!out) return fail(nullptr, VSTAB_E_STATE, "st_meshgrid: NULL buffer");
    if (oh < 1 || ow < 1 || (long long)oh * ow > 0x7fffffffLL / 3) return fail(nullptr, VSTAB_E_SHAPE, "st_meshgrid: bad shape");
    HIP_TRY(nullptr, launch_st_meshgrid(out, oh, ow, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_homography_warp(const float *img, int B, int Hi, int Wi, int C, const float *M, float *out, int oh,
                                     int ow, void *stream)
{
    if (!img || !M || !out) return fail(nullptr, VSTAB_E_STATE, "homography_warp: NULL buffer");
    if (B < 1 || Hi < 1 || Wi < 1 || C < 1 || oh < 1 || ow < 1) return fail(nullptr, VSTAB_E_SHAPE, "homography_warp: bad shape");
    HIP_TRY(nullptr, launch_homography_warp(img, B, Hi, Wi, C, M, out, oh, ow, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_transform_image(const float *img, int B, int Hi, int Wi, int C, const float *ref, const float *pM, float *out, int oh,
                                     int ow, void *stream)
{
    if (!img || !ref || !pM || !out) return fail(nullptr, VSTAB_E_STATE, "transform_image: NULL buffer");
    if (B < 1 || Hi < 1 || Wi < 1 || C < 1 || oh < 1 || ow < 1) return fail(nullptr, VSTAB_E_SHAPE, "transform_image: bad shape");
    HIP_TRY(nullptr, launch_homography_warp(img, B, Hi, Wi, C, pM, out, oh, ow, (hipStream_t)stream, ref));
    return VSTAB_OK;
}

extern "C" int vstab_vec2mtrx(const float *p, int B, int dim, int warp_approx, float *out, void *stream)
{
    if (!p || !out) return fail(nullptr, VSTAB_E_STATE, "vec2mtrx: NULL buffer");
    if (B < 1 || (dim != 8 && dim != 6) || warp_approx < 1 || warp_approx > 64)
        return fail(nullptr, VSTAB_E_SHAPE, "vec2mtrx: p must be [B,8] or [B,6], 1 <= warpApprox <= 64");
    HIP_TRY(nullptr, launch_vec2mtrx(p, B, dim, warp_approx, out, (hipStream_t)stream));
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- VGG16 trunk (vgg16.py)
namespace {
struct VggLayer { const char *name; int cin, cout; bool pool_after; };
const VggLayer VGG[13] = {{"conv1_1", 3, 64, false},   {"conv1_2", 64, 64, true},    {"conv2_1", 64, 128, false},
                          {"conv2_2", 128, 128, true}, {"conv3_1", 128, 256, false}, {"conv3_2", 256, 256, false},
                          {"conv3_3", 256, 256, true}, {"conv4_1", 256, 512, false}, {"conv4_2", 512, 512, false},
                          {"conv4_3", 512, 512, true}, {"conv5_1", 512, 512, false}, {"conv5_2", 512, 512, false},
                          {"conv5_3", 512, 512, true}};

}  // namespace

bool fill_plain_conv(ConvParams &p, ConvTile &tile, bool &vec4, int B, int Hi, int Wi, int cin, int cs_in, int k, int stride,
                     int pad, int cout, int cs_out, int c_off, int act)
{
    std::memset(&p, 0, sizeof p);
    const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
    if (Ho < 1 || Wo < 1 || cs_in < cin) return false;
    p.B = B; p.Hi = Hi; p.Wi = Wi; p.Cs_in = cs_in;
    set_layout(p, cs_in == cin ? klayout_run(k, k, cs_in) : klayout_tap(k, k, cin, cs_in));
    p.s_in = stride; p.s_out = 1; p.Ho = Ho; p.Wo = Wo; p.Cs_out = cs_out; p.c_off = c_off;
    p.N = cout;
    const int BN = cout >= 128 ? 128 : (cout > 32 ? 64 : 32);
    tile = cout >= 128 ? TILE_128x128 : (cout > 32 ? TILE_128x64 : TILE_128x32);
    p.Npad = round_up(cout, BN);
    p.act = act; p.nphase = 1;
    p.ph[0].Hg = Ho; p.ph[0].Wg = Wo; p.ph[0].M = B * Ho * Wo; p.ph[0].off_y = -pad; p.ph[0].off_x = -pad;
    p.Mmax = p.ph[0].M;
    vec4 = (cs_in % 4 == 0) && (p.SEG % 4 == 0);
    if (!vec4 && tile != TILE_128x64) return false;        // the dword-gather variant exists for 128x64 only
    if ((long long)B * Hi * Wi * cs_in * 4 >= 0x80000000LL || (long long)B * Ho * Wo * cs_out * 4 >= 0x80000000LL) return false;
    set_ranges(p);
    tile = choose_tile_split(p, tile, vec4);
    return true;
}

namespace {
struct VggPlan { int h[18], w[18], c[18]; size_t partial_floats, wino_v, wino_m; bool wino[13]; };

bool vgg_plan(int B, int H, int W, VggPlan &v)
{
    if (B < 1 || H < 1 || W < 1) return false;
    int h = H, w = W, o = 0;
    v.partial_floats = v.wino_v = v.wino_m = 0;
    for (int l = 0; l < 13; ++l) {
        ConvParams p; ConvTile t; bool vec;
        if (!fill_plain_conv(p, t, vec, B, h, w, VGG[l].cin, VGG[l].cin, 3, 1, 1, VGG[l].cout, VGG[l].cout, 0, 2)) return false;
        v.wino[l] = VGG[l].cin >= 256 && wino_applies(B, h, w, VGG[l].cin, VGG[l].cout);     // conv3_2 .. conv5_3 when the level is large enough
        if (v.wino[l]) {
            const size_t tiles = (size_t)B * 16 * ((h + 1) / 2) * ((w + 1) / 2);
            v.wino_v = std::max(v.wino_v, tiles * VGG[l].cin);
            v.wino_m = std::max(v.wino_m, tiles * VGG[l].cout);
        } else if (p.ksplit > 1) v.partial_floats = std::max(v.partial_floats, (size_t)p.ksplit * p.Mmax * p.Npad);
        v.h[o] = h; v.w[o] = w; v.c[o] = VGG[l].cout; ++o;
        if (VGG[l].pool_after) {
            h = (h + 1) / 2; w = (w + 1) / 2;
            v.h[o] = h; v.w[o] = w; v.c[o] = VGG[l].cout; ++o;
        }
    }
    return true;
}

size_t vgg_ws_bytes(const VggPlan &v)       // [split-K slabs | Winograd V | Winograd M], each 256-byte aligned
{
    auto a256 = [](size_t n) { return (n + 255) / 256 * 256; };
    return a256(std::max<size_t>(v.partial_floats * 4, 256)) + a256(v.wino_v * 4) + a256(v.wino_m * 4);
}

int vgg_max_chunk(int B, int H, int W)
{
    VggPlan v;
    if (vgg_plan(B, H, W, v)) return B;
    int lo = 0, hi = B;
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (vgg_plan(mid, H, W, v)) lo = mid; else hi = mid - 1;
    }
    return lo;
}
}  // namespace

extern "C" int vstab_vgg16_shapes(int H, int W, int32_t *hwc54)
{
    VggPlan v;
    if (!hwc54 || !vgg_plan(1, H, W, v)) return fail(nullptr, VSTAB_E_SHAPE, "vgg16: unsupported input %dx%d", H, W);
    for (int i = 0; i < 18; ++i) { hwc54[3 * i] = v.h[i]; hwc54[3 * i + 1] = v.w[i]; hwc54[3 * i + 2] = v.c[i]; }
    return VSTAB_OK;
}

extern "C" size_t vstab_vgg16_workspace_bytes(int B, int H, int W)
{
    const int chunk = B >= 1 ? vgg_max_chunk(B, H, W) : 0;
    VggPlan v;
    if (chunk < 1 || !vgg_plan(chunk, H, W, v)) { fail(nullptr, VSTAB_E_SHAPE, "vgg16: unsupported problem %dx%dx%d", B, H, W); return 0; }
    return vgg_ws_bytes(v);
}

extern "C" int vstab_vgg16_load(vstab_ctx *ctx, const vstab_tensor *t, int count)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "vgg16_load: ctx is NULL");
    if (!t || count <= 0) return fail(ctx, VSTAB_E_WEIGHTS, "vgg16_load: no tensors");
    std::vector<float> host;
    auto reserve = [&](size_t n) { size_t o = (host.size() + 63) / 64 * 64; host.resize(o + n, 0.f); return o; };
    std::vector<double> ones;
    ctx->vgg_zero = reserve(1024);              // zero bias for the Winograd-domain GEMMs (the inverse transform adds the real one)
    for (int l = 0; l < 13; ++l) {
        const std::string n = VGG[l].name;
        const vstab_tensor *W = find(t, count, n + "/filter"), *b = find(t, count, n + "/biases");
        if (!shape_is(W, {3, 3, VGG[l].cin, VGG[l].cout}) || !shape_is(b, {VGG[l].cout}))
            return fail(ctx, VSTAB_E_WEIGHTS, "missing or mis-shaped variable %s/{filter,biases}", n.c_str());
        const int BN = VGG[l].cout >= 128 ? 128 : 64, npad = round_up(VGG[l].cout, BN);
        const KLayout L = klayout_run(3, 3, VGG[l].cin);
        ones.assign(npad, 1.0);
        ctx->vgg_b[l] = reserve(npad);
        fold_bn(b->data, nullptr, nullptr, nullptr, VGG[l].cout, npad, ones.data(), host.data() + ctx->vgg_b[l]);
        ctx->vgg_w[l] = reserve((size_t)L.ktiles() * npad * 32);
        pack_conv(W->data, ones.data(), 3, 3, VGG[l].cin, VGG[l].cin, VGG[l].cout, npad, L, host.data() + ctx->vgg_w[l]);
        if (l == 0) {                                  // conv1_1 also unpacked (HWIO as given) for its store-shaped kernel
            ctx->vgg_raw0 = reserve((size_t)27 * VGG[0].cout);
            std::memcpy(host.data() + ctx->vgg_raw0, W->data, sizeof(float) * 27 * VGG[0].cout);
        }
        ctx->vgg_wino_w[l] = 0;
        if (VGG[l].cin >= 256) {                       // Winograd-domain operand for the layers that may run in that form
            ctx->vgg_wino_w[l] = reserve((size_t)16 * (VGG[l].cin / 32) * VGG[l].cout * 32);
            pack_winograd(W->data, ones.data(), VGG[l].cin, VGG[l].cout, VGG[l].cout, host.data() + ctx->vgg_wino_w[l]);
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->vgg_weights) { (void)hipFree(ctx->vgg_weights); ctx->vgg_weights = nullptr; }
    ctx->vgg_loaded = false;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->vgg_weights), host.size() * sizeof(float));
    if (e != hipSuccess) return fail(ctx, VSTAB_E_NOMEM, "hipMalloc(%zu bytes of VGG16 weights): %s", host.size() * 4, hipGetErrorString(e));
    HIP_TRY(ctx, hipMemcpy(ctx->vgg_weights, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->vgg_loaded = true;
    return VSTAB_OK;
}

extern "C" int vstab_vgg16_forward(vstab_ctx *ctx, const float *input, int B, int H, int W, float *const *outs, void *workspace,
                                   size_t workspace_bytes, void *stream_)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "vgg16_forward: ctx is NULL");
    if (!ctx->vgg_loaded) return fail(ctx, VSTAB_E_STATE, "vgg16_forward: vstab_vgg16_load has not been called");
    if (!input || !outs || !workspace) return fail(ctx, VSTAB_E_STATE, "vgg16_forward: NULL buffer");
    for (int i = 0; i < 18; ++i)
        if (!outs[i] || ((uintptr_t)outs[i] & 15)) return fail(ctx, VSTAB_E_ALIGN, "vgg16_forward: output %d NULL or not 16-byte aligned", i);
    const int cmax = B >= 1 ? vgg_max_chunk(B, H, W) : 0;
    if (cmax < 1) return fail(ctx, VSTAB_E_SHAPE, "vgg16_forward: unsupported problem %dx%dx%d", B, H, W);
    const int nchunks = (B + cmax - 1) / cmax, chunk = (B + nchunks - 1) / nchunks;
    VggPlan v;
    if (!vgg_plan(chunk, H, W, v)) return fail(ctx, VSTAB_E_SHAPE, "vgg16_forward: plan failed");
    if (workspace_bytes < vgg_ws_bytes(v)) return fail(ctx, VSTAB_E_NOMEM, "vgg16_forward: workspace %zu < %zu bytes", workspace_bytes, vgg_ws_bytes(v));
    if ((uintptr_t)workspace & 255) return fail(ctx, VSTAB_E_ALIGN, "vgg16_forward: workspace must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    float *wsV = reinterpret_cast<float *>(static_cast<char *>(workspace) + (std::max<size_t>(v.partial_floats * 4, 256) + 255) / 256 * 256);
    float *wsM = reinterpret_cast<float *>(reinterpret_cast<char *>(wsV) + (v.wino_v * 4 + 255) / 256 * 256);
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int bc = std::min(chunk, B - b0);
        const float *cur = input + (size_t)b0 * H * W * 3;
        int h = H, w = W, o = 0;
        for (int l = 0; l < 13; ++l) {
            ConvParams p; ConvTile tile; bool vec;
            if (!fill_plain_conv(p, tile, vec, bc, h, w, VGG[l].cin, VGG[l].cin, 3, 1, 1, VGG[l].cout, VGG[l].cout, 0, 2))
                return fail(ctx, VSTAB_E_SHAPE, "vgg16_forward: layer %s does not fit", VGG[l].name);
            float *dst = outs[o] + (size_t)b0 * v.h[o] * v.w[o] * v.c[o];
            if (l == 0) {
                HIP_TRY(ctx, launch_conv3x3_rgb(cur, bc, h, w, ctx->vgg_weights + ctx->vgg_raw0, ctx->vgg_weights + ctx->vgg_b[0], VGG[0].cout, 1, dst, stream));
            } else if (v.wino[l] && wino_applies(bc, h, w, VGG[l].cin, VGG[l].cout)) {      // (a short last chunk may fall below the break-even)
                ConvParams q;
                fill_wino_gemm(q, bc, h, w, VGG[l].cin, VGG[l].cout);
                HIP_TRY(ctx, launch_wino_input(cur, bc, h, w, VGG[l].cin, 0, VGG[l].cin, wsV, stream));
                q.in = wsV; q.out = wsM; q.wpk = ctx->vgg_weights + ctx->vgg_wino_w[l]; q.bias = ctx->vgg_weights + ctx->vgg_zero;
                HIP_TRY(ctx, launch_conv(q, TILE_128x64, true, stream));
                HIP_TRY(ctx, launch_wino_output(wsM, bc, h, w, VGG[l].cout, ctx->vgg_weights + ctx->vgg_b[l], 2, dst, VGG[l].cout, 0, stream));
            } else {
                p.in = cur; p.out = dst; p.wpk = ctx->vgg_weights + ctx->vgg_w[l]; p.bias = ctx->vgg_weights + ctx->vgg_b[l];
                p.partial = (float *)workspace;
                HIP_TRY(ctx, launch_conv(p, tile, vec, stream));
            }
            cur = dst; ++o;
            if (VGG[l].pool_after) {
                float *pd = outs[o] + (size_t)b0 * v.h[o] * v.w[o] * v.c[o];
                HIP_TRY(ctx, launch_maxpool2x2(cur, bc, h, w, VGG[l].cout, pd, stream));
                h = (h + 1) / 2; w = (w + 1) / 2;
                cur = pd; ++o;
            }
        }
    }
    return VSTAB_OK;
}

extern "C" int vstab_scale_shift(const float *x, long long npix, int C, float scale, const float *mean, float *out, void *stream)
{
    if (!x || !mean || !out) return fail(nullptr, VSTAB_E_STATE, "scale_shift: NULL buffer");
    if (npix < 1 || C < 1 || C > 4) return fail(nullptr, VSTAB_E_SHAPE, "scale_shift: bad shape");
    HIP_TRY(nullptr, launch_scale_shift(x, npix, C, scale, mean, out, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_maxpool2x2(const float *x, int B, int H, int W, int C, float *out, void *stream)
{
    if (!x || !out) return fail(nullptr, VSTAB_E_STATE, "maxpool2x2: NULL buffer");
    if (B < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return fail(nullptr, VSTAB_E_SHAPE, "maxpool2x2: bad shape (C must be a multiple of 4)");
    HIP_TRY(nullptr, launch_maxpool2x2(x, B, H, W, C, out, (hipStream_t)stream));
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- clip driver pieces
extern "C" int vstab_resize_u8(const uint8_t *src, int B, int sh, int sw, uint8_t *dst, int dh, int dw, void *stream)
{
    if (!src || !dst) return fail(nullptr, VSTAB_E_STATE, "resize_u8: NULL buffer");
    if (B < 1 || sh < 1 || sw < 1 || dh < 1 || dw < 1) return fail(nullptr, VSTAB_E_SHAPE, "resize_u8: bad shape");
    HIP_TRY(nullptr, launch_resize_u8(src, B, sh, sw, dst, dh, dw, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_resize_f32_to_u8(const float *src, int B, int sh, int sw, uint8_t *dst, int dh, int dw, void *stream)
{
    if (!src || !dst) return fail(nullptr, VSTAB_E_STATE, "resize_f32_to_u8: NULL buffer");
    if (B < 1 || sh < 1 || sw < 1 || dh < 1 || dw < 1) return fail(nullptr, VSTAB_E_SHAPE, "resize_f32_to_u8: bad shape");
    HIP_TRY(nullptr, launch_resize_f32_to_u8(src, B, sh, sw, dst, dh, dw, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_assemble_input(const uint8_t *const *slots9, int B, int h, int w, float *feats, void *stream)
{
    if (!slots9 || !feats) return fail(nullptr, VSTAB_E_STATE, "assemble_input: NULL buffer");
    for (int j = 0; j < 9; ++j)
        if (!slots9[j]) return fail(nullptr, VSTAB_E_STATE, "assemble_input: slot %d is NULL", j);
    if (B < 1 || h < 1 || w < 1) return fail(nullptr, VSTAB_E_SHAPE, "assemble_input: bad shape");
    if ((uintptr_t)feats & 15) return fail(nullptr, VSTAB_E_ALIGN, "assemble_input: feats must be 16-byte aligned");
    HIP_TRY(nullptr, launch_assemble_input(slots9, B, h, w, feats, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_assemble_input_resized(const uint8_t *const *slots8, const uint8_t *frame, int B, int h, int w, int sh, int sw, float *feats,
                                            void *stream)
{
    if (!slots8 || !frame || !feats) return fail(nullptr, VSTAB_E_STATE, "assemble_input_resized: NULL buffer");
    if (B < 1 || h < 1 || w < 1 || sh < 1 || sw < 1) return fail(nullptr, VSTAB_E_SHAPE, "assemble_input_resized: bad shape");
    if ((uintptr_t)feats & 15) return fail(nullptr, VSTAB_E_ALIGN, "assemble_input_resized: feats must be 16-byte aligned");
    HIP_TRY(nullptr, launch_assemble_input_resized(slots8, frame, B, h, w, sh, sw, feats, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_flow_glue_warp_u8(const float *flow, int B, int h, int w, const uint8_t *frame, float *outflow, uint8_t *out, int oh, int ow,
                                       int net_h, int net_w, void *stream)
{
    if (!flow || !frame || !out) return fail(nullptr, VSTAB_E_STATE, "flow_glue_warp_u8: NULL buffer");
    if (B < 1 || h < 1 || w < 2 || oh < 1 || ow < 1 || net_h < 1 || net_w < 1) return fail(nullptr, VSTAB_E_SHAPE, "flow_glue_warp_u8: bad shape (w >= 2)");
    if ((long long)B * oh * ow >= (1ll << 31) / 3) return fail(nullptr, VSTAB_E_SHAPE, "flow_glue_warp_u8: 3*B*oh*ow must be < 2^31");
    if (((uintptr_t)flow & 7) || ((uintptr_t)outflow & 7) || ((uintptr_t)out & 3))
        return fail(nullptr, VSTAB_E_ALIGN, "flow_glue_warp_u8: flow / outflow 8-byte, out 4-byte alignment");
    TraceRange range("frame_to_float+flow_glue+tf_warp+quantise");
    HIP_TRY(nullptr, launch_flow_glue_warp_u8(flow, B, h, w, frame, outflow, out, oh, ow, net_h, net_w, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_frame_to_float(const uint8_t *frame, long long npix, float *out, void *stream)
{
    if (!frame || !out || npix < 1) return fail(nullptr, VSTAB_E_STATE, "frame_to_float: bad argument");
    HIP_TRY(nullptr, launch_frame_to_float(frame, npix, out, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_quantise_output(const float *warped, long long npix, uint8_t *out, void *stream)
{
    if (!warped || !out || npix < 1) return fail(nullptr, VSTAB_E_STATE, "quantise_output: bad argument");
    HIP_TRY(nullptr, launch_quantise_output(warped, npix, out, (hipStream_t)stream));
    return VSTAB_OK;
}

// One frame of the evaluator's loop (main:550-558, 568-569, 497-514, 625/630, 556) as ONE call: network input from the history slots + the
// frame (cv2.resize inside the launch), the network, the 8-bit glue + warp launch, the stabilised frame resized into its history slot.
extern "C" int vstab_clip_step(vstab_ctx *ctx, const uint8_t *const *slots8, const uint8_t *frame, int n, int net_h, int net_w, int oh, int ow,
                               float *feats, float *pf6, float *pf5, float *pf4, float *pf3, float *pf2, float *outflow, uint8_t *out,
                               uint8_t *ring_slot, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!ctx) return fail(nullptr, VSTAB_E_STATE, "clip_step: ctx is NULL");
    if (!slots8 || !frame || !feats || !out || !ring_slot) return fail(ctx, VSTAB_E_STATE, "clip_step: NULL buffer");
    if (n < 1 || net_h < 3 || net_w < 4 || oh < 1 || ow < 1) return fail(ctx, VSTAB_E_SHAPE, "clip_step: bad shape");
    {   // the warp GATHERS frame pixels while other workgroups already write `out`, and the history slot is resized from `out`:
        // neither may overlap the frame, nor each other
        const size_t fb = (size_t)n * oh * ow * 3, sb = (size_t)n * net_h * net_w * 3;
        auto overlap = [](const void *a, size_t na, const void *b, size_t nb) {
            const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
            return x < y + nb && y < x + na;
        };
        if (overlap(out, fb, frame, fb) || overlap(ring_slot, sb, frame, fb) || overlap(ring_slot, sb, out, fb))
            return fail(ctx, VSTAB_E_STATE, "clip_step: out / ring_slot / frame must not overlap");
    }
    int rc = vstab_assemble_input_resized(slots8, frame, n, net_h, net_w, oh, ow, feats, stream);
    // the network; its last launch also does the 8-bit glue + warp of the frame when the geometry allows (flow_ops.hip, pf2_glue_warp_kernel)
    FusedTail tail{nullptr, outflow, nullptr, oh, ow, false, frame, out};
    const bool try_fused = (((uintptr_t)outflow & 7) | ((uintptr_t)out & 3)) == 0 && (long long)n * oh * ow < (1ll << 31) / 3;
    if (rc == VSTAB_OK) rc = forward_impl(ctx, feats, n, net_h, net_w, 27, pf6, pf5, pf4, pf3, pf2, workspace, workspace_bytes, stream, try_fused ? &tail : nullptr);
    else adopt_last_error(ctx);
    if (rc != VSTAB_OK) return rc;
    if (!tail.fused) rc = vstab_flow_glue_warp_u8(pf2, n, net_h - 2, net_w - 2, frame, outflow, out, oh, ow, net_h, net_w, stream);
    if (rc == VSTAB_OK) rc = vstab_resize_u8(out, n, oh, ow, ring_slot, net_h, net_w, stream);
    if (rc != VSTAB_OK) adopt_last_error(ctx);
    return rc;
}

// ------------------------------------------------------------------------- flow post-filters
extern "C" int vstab_flow_box_blur(const float *flow, int B, int h, int w, int k, float *tmp, float *out, void *stream)
{
    if (!flow || !tmp || !out) return fail(nullptr, VSTAB_E_STATE, "flow_box_blur: NULL buffer");
    if (B < 1 || h < 1 || w < 1 || k < 1 || !(k & 1)) return fail(nullptr, VSTAB_E_SHAPE, "flow_box_blur: bad shape (k must be odd)");
    HIP_TRY(nullptr, launch_flow_box_blur(flow, B, h, w, k, tmp, out, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_axpby(const float *x, float a, const float *y, float b, float *out, long long n, void *stream)
{
    if (!x || !y || !out || n < 1) return fail(nullptr, VSTAB_E_STATE, "axpby: bad argument");
    HIP_TRY(nullptr, launch_axpby(x, a, y, b, out, n, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_loss_level(const float *pf, const float *gt, const float *unstab, int B, int h, int w, double *sums,
                                float scale_mse, float scale_tv, float *grad_pf, void *stream)
{
    if (!pf || !gt || !unstab || !sums) return fail(nullptr, VSTAB_E_STATE, "loss_level: NULL buffer");
    if (B < 1 || h < 1 || w < 1 || (long long)h * w > (1LL << 30)) return fail(nullptr, VSTAB_E_SHAPE, "loss_level: bad shape");
    if ((reinterpret_cast<uintptr_t>(pf) & 7) || (grad_pf && (reinterpret_cast<uintptr_t>(grad_pf) & 7)) ||
        (reinterpret_cast<uintptr_t>(sums) & 7))
        return fail(nullptr, VSTAB_E_ALIGN, "loss_level: flow / gradient / sums must be 8-byte aligned");
    HIP_TRY(nullptr, launch_loss_level(pf, gt, unstab, B, h, w, sums, scale_mse, scale_tv, grad_pf, (hipStream_t)stream));
    return VSTAB_OK;
}

static int loss_main_check(const vstab_loss_level_desc *lv, int n, int B)
{
    if (!lv || n < 1 || n > 8 || B < 1) return VSTAB_E_SHAPE;
    for (int l = 0; l < n; ++l) {
        if (!lv[l].pf || lv[l].h < 1 || lv[l].w < 1 || lv[l].cs_pf < 2 || (lv[l].cs_pf & 1)) return VSTAB_E_SHAPE;
        if (lv[l].grad && (lv[l].cs_grad < 2 || (lv[l].cs_grad & 1))) return VSTAB_E_SHAPE;
        if (((uintptr_t)lv[l].pf | (uintptr_t)lv[l].grad) & 7) return VSTAB_E_ALIGN;
    }
    return VSTAB_OK;
}

extern "C" size_t vstab_loss_main_workspace_bytes(const vstab_loss_level_desc *levels, int n_levels, int B)
{
    if (loss_main_check(levels, n_levels, B) != VSTAB_OK) return 0;
    static_assert(sizeof(vstab_loss_level_desc) == sizeof(LossLevel), "descriptor layout");
    return loss_main_workspace_bytes(B, reinterpret_cast<const LossLevel *>(levels), n_levels);
}

extern "C" int vstab_loss_main(const vstab_loss_level_desc *levels, int n_levels, const float *gtstab, const float *unstab, int B, int H,
                               int W, double *loss_out, void *workspace, size_t workspace_bytes, void *stream)
{
    const int c = loss_main_check(levels, n_levels, B);
    if (c != VSTAB_OK) return fail(nullptr, c, "loss_main: bad level descriptor (1..8 levels, even channel strides >= 2, 8-byte aligned buffers)");
    if (!gtstab || !unstab || !loss_out || !workspace) return fail(nullptr, VSTAB_E_STATE, "loss_main: NULL buffer");
    if (H < 1 || W < 1) return fail(nullptr, VSTAB_E_SHAPE, "loss_main: bad image shape");
    if (((uintptr_t)loss_out | (uintptr_t)workspace) & 7) return fail(nullptr, VSTAB_E_ALIGN, "loss_main: loss_out / workspace must be 8-byte aligned");
    const LossLevel *lv = reinterpret_cast<const LossLevel *>(levels);
    if (workspace_bytes < loss_main_workspace_bytes(B, lv, n_levels)) return fail(nullptr, VSTAB_E_NOMEM, "loss_main: workspace too small");
    void *ws = (void *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    HIP_TRY(nullptr, launch_loss_main(lv, n_levels, gtstab, unstab, B, H, W, loss_out, ws, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_flow_medfilt(const float *flow, int B, int h, int w, int kh, int kw, int kc, float *out, void *stream)
{
    if (!flow || !out) return fail(nullptr, VSTAB_E_STATE, "flow_medfilt: NULL buffer");
    if (flow == out) return fail(nullptr, VSTAB_E_STATE, "flow_medfilt: in-place filtering is not supported");
    if (B < 1 || h < 1 || w < 1) return fail(nullptr, VSTAB_E_SHAPE, "flow_medfilt: bad shape");
    if (kh < 1 || kw < 1 || kc < 1 || !(kh & 1) || !(kw & 1) || !(kc & 1) || kh > 31 || kw > 31 || kc > 5)
        return fail(nullptr, VSTAB_E_SHAPE, "flow_medfilt: kernel sizes must be odd, kh,kw <= 31, kc <= 5");
    HIP_TRY(nullptr, launch_flow_medfilt(flow, B, h, w, kh, kw, kc, out, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" size_t vstab_homography_workspace_bytes(int B, int H, int W, int K)
{
    if (B < 1 || H < 1 || W < 1 || K < 1 || K > 512) return 0;
    return homography_workspace_bytes(B, H, W, K);
}

extern "C" int vstab_homography_fit(const float *flow, int B, int H, int W, int K, unsigned seed, double thresh, int refine,
                                    int stride, double *Hout, int32_t *inliers, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    if (!flow || !Hout || !inliers || !workspace) return fail(nullptr, VSTAB_E_STATE, "homography_fit: NULL buffer");
    if (B < 1 || H < 2 || W < 2 || (long long)H * W > 0x7fffffffLL) return fail(nullptr, VSTAB_E_SHAPE, "homography_fit: bad shape");
    if (K < 1 || K > 512 || refine < 1 || refine > 16 || stride < 1 || !(thresh > 0.0))
        return fail(nullptr, VSTAB_E_SHAPE, "homography_fit: need 1 <= K <= 512, 1 <= refine <= 16, stride >= 1, thresh > 0");
    if (workspace_bytes < homography_workspace_bytes(B, H, W, K)) return fail(nullptr, VSTAB_E_NOMEM, "homography_fit: workspace too small");
    if (((uintptr_t)flow | (uintptr_t)Hout | (uintptr_t)workspace) & 7)
        return fail(nullptr, VSTAB_E_ALIGN, "homography_fit: flow / Hout / workspace must be 8-byte aligned");
    HIP_TRY(nullptr, launch_homography_fit(flow, B, H, W, K, seed, thresh, refine, stride, Hout, inliers, workspace, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_warp_perspective_u8(const uint8_t *src, int B, int sh, int sw, const double *Hm, uint8_t *dst, int oh, int ow,
                                         void *stream)
{
    if (!src || !Hm || !dst) return fail(nullptr, VSTAB_E_STATE, "warp_perspective_u8: NULL buffer");
    if (src == dst) return fail(nullptr, VSTAB_E_STATE, "warp_perspective_u8: in-place warping is not supported");
    if (B < 1 || sh < 1 || sw < 1 || oh < 1 || ow < 1 || sh > 32767 || sw > 32767)
        return fail(nullptr, VSTAB_E_SHAPE, "warp_perspective_u8: bad shape");
    HIP_TRY(nullptr, launch_warp_perspective_u8(src, B, sh, sw, Hm, dst, oh, ow, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_flow_mean_fill(const float *flow, int B, int h, int w, float *out, void *stream)
{
    if (!flow || !out) return fail(nullptr, VSTAB_E_STATE, "flow_mean_fill: NULL buffer");
    if (B < 1 || h < 1 || w < 1) return fail(nullptr, VSTAB_E_SHAPE, "flow_mean_fill: bad shape");
    HIP_TRY(nullptr, launch_flow_mean_fill(flow, B, h, w, out, (hipStream_t)stream));
    return VSTAB_OK;
}
