// C ABI of the training-side building blocks: convolution weight / bias gradients.
// The loss entry point lives in api.cpp next to the other small wrappers.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../../include/vstab.h"
#include "api_internal.h"
#include "vstab_internal.h"

using namespace vstab;

namespace {
// The weight-gradient kernel's per-output-pixel table depends on the geometry only: built once per (device, geometry) and kept
// for the life of the process (16 B per output pixel; a training step used to rebuild 23 of them).
const int4 *wgrad_pixel_table(int B, int Hi, int Wi, int cs_x, int Ho, int Wo, int stride, int pad, hipStream_t st)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int, int, int, int, int>, int4 *> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const auto key = std::make_tuple(dev, B, Hi, Wi, cs_x, Ho, Wo, stride, pad);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int4 *t = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&t), (size_t)B * Ho * Wo * sizeof(int4)) != hipSuccess) return nullptr;
    // built on the caller's stream and completed before anyone (on any stream) can be handed the cached pointer
    if (launch_wgrad_pixel_table(B, Hi, Wi, cs_x, Ho, Wo, stride, pad, t, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        (void)hipFree(t);
        return nullptr;
    }
    cache[key] = t;
    return t;
}
}  // namespace

namespace {
// How a filter gradient with `cout` columns is launched: as one launch, or -- 128 k + a few columns (the decoder's concat inputs:
// 1026 / 770 / 386 channels + padding) -- as its 128-wide part plus a narrow tail, instead of a whole 128-column tile row for the tail.
struct WgradSplit { int n_main, n_tail, ks_main, ks_tail; size_t slab_floats; };
WgradSplit wgrad_split(int M, int cout, int K)
{
    WgradSplit s{};
    WgradParams p{};
    p.M = M; p.K = K;
    s.n_main = cout / 128 * 128; s.n_tail = cout - s.n_main;
    if (!(s.n_main >= 128 && s.n_tail > 0 && s.n_tail <= 32)) { s.n_main = cout; s.n_tail = 0; }
    p.Cout = s.n_main; s.ks_main = wgrad_choose_split(p);
    s.slab_floats = s.ks_main > 1 ? (size_t)s.ks_main * M * s.n_main : 0;
    if (s.n_tail) {
        p.Cout = s.n_tail; s.ks_tail = wgrad_choose_split(p);
        s.slab_floats = std::max(s.slab_floats, s.ks_tail > 1 ? (size_t)s.ks_tail * M * s.n_tail : 0);      // the launches run one after the other
    }
    return s;
}
}  // namespace

extern "C" size_t vstab_conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int k, int cin, int cout)
{
    if (B < 1 || Ho < 1 || Wo < 1 || k < 1 || cin < 1 || cout < 1) return 0;
    const WgradSplit s = wgrad_split(k * k * cin, cout, B * Ho * Wo);
    const size_t slabs = (s.slab_floats * sizeof(float) + 255) / 256 * 256;
    return slabs + (size_t)column_sum_chunks((long long)B * Ho * Wo, cout) * cout * sizeof(float) + 256;
}

extern "C" int vstab_conv_wgrad(const float *x, int B, int Hi, int Wi, int cs_x, int cx_off, int cin, const float *gout, int Ho,
                                int Wo, int cs_g, int cg_off, int cout, int k, int stride, int pad, float *dW, float *db,
                                int accumulate, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !gout || !dW || !workspace) return fail(nullptr, VSTAB_E_STATE, "conv_wgrad: NULL buffer");
    if (B < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || k < 1 || k > 7 || stride < 1 || pad < 0 || cin < 1 || cout < 1 ||
        cx_off < 0 || cg_off < 0 || cx_off + cin > cs_x || cg_off + cout > cs_g)
        return fail(nullptr, VSTAB_E_SHAPE, "conv_wgrad: bad shape");
    {   // one extra output row/col is allowed (TF SAME on odd sizes / cropped transposed conv); its out-of-image taps read zero
        const int Ho_min = (Hi + 2 * pad - k) / stride + 1, Wo_min = (Wi + 2 * pad - k) / stride + 1;
        if (Ho < Ho_min || Wo < Wo_min || Ho > Ho_min + 1 || Wo > Wo_min + 1)
            return fail(nullptr, VSTAB_E_SHAPE, "conv_wgrad: %dx%d input, k %d stride %d pad %d does not match a %dx%d output", Hi, Wi, k,
                        stride, pad, Ho, Wo);
    }
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
    const WgradSplit sp = wgrad_split(p.M, cout, p.K);
    p.ptab = wgrad_pixel_table(B, Hi, Wi, cs_x, Ho, Wo, stride, pad, st);
    if (!p.ptab) return fail(nullptr, VSTAB_E_NOMEM, "conv_wgrad: cannot build the pixel table");
    p.partial = reinterpret_cast<float *>(workspace);
    p.Cout = sp.n_main; p.ldw = cout; p.col0 = 0; p.ksplit = sp.ks_main;
    HIP_TRY(nullptr, launch_wgrad(p, st));
    if (sp.n_tail) {
        p.Cout = sp.n_tail; p.cg_off = cg_off + sp.n_main; p.col0 = sp.n_main; p.ksplit = sp.ks_tail;
        HIP_TRY(nullptr, launch_wgrad(p, st));
    }
    if (db) {
        const size_t slabs = (sp.slab_floats * sizeof(float) + 255) / 256 * 256;
        float *scratch = reinterpret_cast<float *>(reinterpret_cast<char *>(p.partial) + slabs);
        HIP_TRY(nullptr, launch_column_sum(gout, (long long)p.K, cs_g, cg_off, cout, db, accumulate ? 1 : 0, scratch, st));
    }
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- input gradient of a convolution
namespace {
// a zero vector every bias-less GEMM can point at (instead of a memset in front of each launch); one per device, never freed
const float *zero_bias()
{
    static std::mutex mu;
    static std::map<int, float *> buf;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = buf.find(dev);
    if (it != buf.end()) return it->second;
    float *p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), 8192 * sizeof(float)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 8192 * sizeof(float)) != hipSuccess) { (void)hipFree(p); return nullptr; }
    buf[dev] = p;
    return p;
}

// Does a gather table read its source contiguously along the output column?  Sampled: for logical k, the entries of columns n and n + 1
// (both live) differ by one.  Rows of the packed operand are [K-tile][Npad][32] with the 16-byte chunks of column n swizzled by (n >> 1) & 7.
bool table_is_n_fast(const std::vector<int32_t> &tbl, int npad)
{
    long long hit = 0, seen = 0;
    const size_t rows = tbl.size() / 32;
    for (size_t r = 0; r + 1 < rows && seen < 4096; r += 7) {
        const int n = (int)(r % npad);
        if (n + 1 >= npad) continue;
        for (int k = 0; k < 32; k += 5) {
            const int32_t a = tbl[r * 32 + swz32(n, k)], b = tbl[(r + 1) * 32 + swz32(n + 1, k)];
            if (a > 0 && b > 0) { ++seen; hit += (b == a + 1); }
        }
    }
    return seen >= 16 && hit * 10 >= seen * 9;
}

struct DgradPlan {
    ConvParams p;
    ConvTile tile;
    bool vec4;
    size_t packed_floats;       // all phases
    int32_t *tbl;               // device index table, owned by the cache (NULL when `blocked`)
    bool tbl_nfast;             // the table's sources run contiguously along the output column: the LDS-transposing replay (train_ops.hip)
    int blocked_K;              // > 0: the operand is a plain [K][N] matrix -> launch_pack_blocked instead of the table
    // odd-k stride-2 input gradients: the four output-parity phases have DIFFERENT tap counts ((k+1)/2 or (k-1)/2 per axis), so each
    // is its own launch with exactly its taps instead of one 4-phase launch padded to ceil(k/2)^2 (44 % / 31 % of the MACs of a
    // 3x3 / 5x5 layer were zero taps).  nsub == 0: the single launch `p` is the whole plan.
    int nsub;
    ConvParams sp[4];
    ConvTile stile[4];
    size_t soff[4];             // float offset of each phase's operand inside the packed buffer / the table
    size_t part_floats;         // split-K slab space (max over the launches)
};

// geometry -> plan + device index table (built once: the host packer's gather, replayed on the GPU every call because the
// weights of a training run change every step)
bool dgrad_plan(int B, int Ho, int Wo, int cs_g, int cout, int k, int stride, int pad, int Hi, int Wi, int cs_x, int cx_off, int cin,
                int accumulate, DgradPlan &out)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int, int, int, int, int, int, int, int, int, int, int>, DgradPlan> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;             // the plan owns an index table in THIS device's memory
    const auto key = std::make_tuple(dev, B, Ho, Wo, cs_g, cout, k, stride, pad, Hi, Wi, cs_x, cx_off, cin, accumulate);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) { out = it->second; return true; }
    DgradPlan d{};
    ConvParams &p = d.p;
    std::vector<int32_t> tbl;
    const int act = accumulate ? 3 : 0;
    if (stride == 1) {
        // conv over gout with pad k-1-pad, flipped kernel, channel roles swapped
        if (!fill_plain_conv(p, d.tile, d.vec4, B, Ho, Wo, cout, cs_g, k, 1, k - 1 - pad, cin, cs_x, cx_off, act)) return false;
        if (p.Ho != Hi || p.Wo != Wi || !d.vec4) return false;
        const KLayout L{p.KH, p.NSEG, p.SEG, p.SEGP, p.SEG_STRIDE};
        d.packed_floats = (size_t)L.ktiles() * p.Npad * 32;
        tbl.resize(d.packed_floats);
        pack_index_dgrad_s1(k, cin, cout, cs_g, L, p.Npad, tbl.data());
    } else if (k & 1) {
        // one launch per output-parity phase, each with exactly its taps: row taps ky = t0y + 2u (u < nty), input row j + c - u
        if (cs_g & 3) return false;
        const int BN = cin >= 128 ? 128 : (cin > 32 ? 64 : 32);
        const ConvTile base = cin >= 128 ? TILE_128x128 : (cin > 32 ? TILE_128x64 : TILE_128x32);
        const int npad = (cin + BN - 1) / BN * BN;
        d.vec4 = true;
        d.packed_floats = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                const int t0y = (py + pad) & 1, t0x = (px + pad) & 1;
                const int nty = (k - t0y + 1) / 2, ntx = (k - t0x + 1) / 2;
                const int Hg = (Hi - py + 1) / 2, Wg = (Wi - px + 1) / 2;
                if (Hg < 1 || Wg < 1 || nty < 1 || ntx < 1) continue;
                ConvParams &q = d.sp[d.nsub];
                std::memset(&q, 0, sizeof q);
                q.B = B; q.Hi = Ho; q.Wi = Wo; q.Cs_in = cs_g;
                const KLayout L = cs_g == cout ? klayout_run(nty, ntx, cs_g) : klayout_tap(nty, ntx, cout, cs_g);
                set_layout(q, L);
                if (q.SEG & 3) return false;
                q.s_in = 1; q.s_out = 2; q.Ho = Hi; q.Wo = Wi; q.Cs_out = cs_x; q.c_off = cx_off;
                q.N = cin; q.Npad = npad; q.act = act; q.nphase = 1;
                ConvPhase &ph = q.ph[0];
                ph.Hg = Hg; ph.Wg = Wg; ph.M = B * Hg * Wg;
                ph.off_y = (py + pad - t0y) / 2 - nty + 1; ph.off_x = (px + pad - t0x) / 2 - ntx + 1;
                ph.o_y = py; ph.o_x = px; ph.w_off = 0;
                q.Mmax = ph.M;
                set_ranges(q);
                d.stile[d.nsub] = choose_tile_split(q, base, true);
                d.soff[d.nsub] = d.packed_floats;
                const size_t pf = (size_t)L.ktiles() * npad * 32;
                tbl.resize(d.packed_floats + pf);
                // row tap tt <-> ky = t0y + 2 (nty-1-tt): increasing input row; GEMM columns n = ci, reduction channel = co
                pack_index_phase(k, cin, cout, cs_g, L, npad, t0y, nty, t0x, ntx, tbl.data() + d.packed_floats);
                d.packed_floats += pf;
                if (q.ksplit > 1) d.part_floats = std::max(d.part_floats, (size_t)q.ksplit * q.Mmax * q.Npad);
                ++d.nsub;
            }
        if (d.nsub == 0) return false;
#ifdef VSTAB_HARNESS
        static const bool sublaunch = getenv("VSTAB_DGRAD_SUBLAUNCH") != nullptr;       // A/B switch of the tuning harness: always one launch per phase
#else
        constexpr bool sublaunch = false;
#endif
        long long tiles_all = 0;
        for (int s = 0; s < d.nsub; ++s) tiles_all += (long long)((d.sp[s].ph[0].M + 127) / 128) * (npad / BN);
        if (!sublaunch && tiles_all <= 256) {
            // Small layers: ONE launch in which every phase carries its own K layout (ConvPhase::KH ...), split-K sized for the four
            // together -- instead of four launches that each fill a fraction of the chip (conv6 at B=8 512x512: 112 -> 85 us; the
            // whole B=1 step 3.45 -> 3.37 ms).  Large layers stay one launch per phase: merged they were SLOWER (conv3 523 -> 585 us,
            // conv2 557 -> 633 us): a launch that works on one phase keeps a quarter of the filter hot in L2.
            p = d.sp[0];
            p.nphase = d.nsub;
            p.Mmax = 0;
            unsigned wmax = 0;
            long long tiles = 0;
            int kt_min = 1 << 30;
            for (int s = 0; s < d.nsub; ++s) {
                const ConvParams &q = d.sp[s];
                ConvPhase &ph = p.ph[s];
                ph = q.ph[0];
                ph.w_off = (long long)d.soff[s];
                ph.KH = q.KH; ph.NSEG = q.NSEG; ph.SEG = q.SEG; ph.SEGP = q.SEGP; ph.SEG_STRIDE = q.SEG_STRIDE;
                p.Mmax = std::max(p.Mmax, ph.M);
                wmax = std::max(wmax, q.w_bytes);
                tiles += (long long)((ph.M + 127) / 128) * (npad / BN);
                kt_min = std::min(kt_min, q.KH * q.NSEG * (q.SEGP / 32));
            }
            p.w_bytes = wmax;
            int ks = (int)(512 / std::max<long long>(tiles, 1));                  // two co-resident workgroups per CU
            ks = std::max(1, std::min(ks, std::max(1, kt_min / 4)));
            p.ksplit = ks;
            d.tile = base;
            d.nsub = 0;
        } else {
            d.p = d.sp[0];
            d.tile = d.stile[0];
        }
    } else {
        const int kt2 = (k + 1) / 2;
        std::memset(&p, 0, sizeof p);
        p.B = B; p.Hi = Ho; p.Wi = Wo; p.Cs_in = cs_g;
        const KLayout L = cs_g == cout ? klayout_run(kt2, kt2, cs_g) : klayout_tap(kt2, kt2, cout, cs_g);
        set_layout(p, L);
        p.s_in = 1; p.s_out = 2; p.Ho = Hi; p.Wo = Wi; p.Cs_out = cs_x; p.c_off = cx_off;
        p.N = cin;
        const int BN = cin >= 128 ? 128 : (cin > 32 ? 64 : 32);
        d.tile = cin >= 128 ? TILE_128x128 : (cin > 32 ? TILE_128x64 : TILE_128x32);
        p.Npad = (cin + BN - 1) / BN * BN;
        p.act = act; p.nphase = 4;
        const size_t phase_floats = (size_t)L.ktiles() * p.Npad * 32;
        p.Mmax = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                ConvPhase &ph = p.ph[py * 2 + px];
                const int t0y = (py + pad) & 1, t0x = (px + pad) & 1;
                ph.Hg = (Hi - py + 1) / 2; ph.Wg = (Wi - px + 1) / 2;
                ph.M = B * ph.Hg * ph.Wg;
                ph.off_y = (py + pad - t0y) / 2 - kt2 + 1; ph.off_x = (px + pad - t0x) / 2 - kt2 + 1;
                ph.o_y = py; ph.o_x = px;
                ph.w_off = (long long)(py * 2 + px) * phase_floats;
                p.Mmax = std::max(p.Mmax, ph.M);
            }
        d.vec4 = true;
        if ((cs_g & 3) || (p.SEG & 3)) return false;
        set_ranges(p);
        d.tile = choose_tile_split(p, d.tile, true);
        d.packed_floats = 4 * phase_floats;
        tbl.resize(d.packed_floats);
        pack_index_dgrad_s2(k, pad, cin, cout, cs_g, L, p.Npad, tbl.data());
    }
    if (d.nsub == 0 && d.p.ksplit > 1) d.part_floats = (size_t)d.p.nphase * d.p.ksplit * d.p.Mmax * d.p.Npad;
    if (hipMalloc(reinterpret_cast<void **>(&d.tbl), tbl.size() * sizeof(int32_t)) != hipSuccess) return false;
    if (hipMemcpy(d.tbl, tbl.data(), tbl.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d.tbl); return false; }
    cache[key] = d;
    out = d;
    return true;
}

size_t dgrad_ws(const DgradPlan &d, size_t *bias_off, size_t *part_off)
{
    const ConvParams &p = d.p;
    size_t off = (d.packed_floats * sizeof(float) + 255) / 256 * 256;
    *bias_off = off;
    off += ((size_t)p.Npad * sizeof(float) + 255) / 256 * 256;
    *part_off = off;
    off += (d.part_floats * sizeof(float) + 255) / 256 * 256;
    return off + 256;
}
}  // namespace

extern "C" size_t vstab_conv_dgrad_workspace_bytes(int B, int Ho, int Wo, int cs_g, int cout, int k, int stride, int pad, int Hi, int Wi,
                                                   int cs_x, int cx_off, int cin, int accumulate)
{
    DgradPlan d;
    if (!dgrad_plan(B, Ho, Wo, cs_g, cout, k, stride, pad, Hi, Wi, cs_x, cx_off, cin, accumulate ? 1 : 0, d)) return 0;
    size_t a, b;
    return dgrad_ws(d, &a, &b);
}

extern "C" int vstab_conv_dgrad(const float *gout, int B, int Ho, int Wo, int cs_g, int cg_off, int cout, const float *W, const float *bias_in,
                                int k, int stride, int pad, float *dx, int Hi, int Wi, int cs_x, int cx_off, int cin, int accumulate,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    if (!gout || !W || !dx || !workspace) return fail(nullptr, VSTAB_E_STATE, "conv_dgrad: NULL buffer");
    if (B < 1 || Ho < 1 || Wo < 1 || Hi < 1 || Wi < 1 || k < 1 || k > 7 || (stride != 1 && stride != 2) || pad < 0 || pad > k - 1 ||
        cin < 1 || cout < 1 || cg_off < 0 || cx_off < 0 || cg_off + cout > cs_g || cx_off + cin > cs_x)
        return fail(nullptr, VSTAB_E_SHAPE, "conv_dgrad: bad shape (stride must be 1 or 2)");
    // gout may have MORE rows/cols than a symmetric-pad conv of dx's size would give: TF's SAME conv of an odd size pads one more
    // at the end, and a transposed conv with a cropped output_shape (model.py:850: 23 <- 12) is exactly that case.  Windows that
    // stick out past dx simply lose those taps (range checks).
    const int Ho_min = (Hi + 2 * pad - k) / stride + 1, Wo_min = (Wi + 2 * pad - k) / stride + 1;
    if (Ho < Ho_min || Wo < Wo_min || Ho > Ho_min + 1 || Wo > Wo_min + 1 || (stride == 1 && (Ho != Ho_min || Wo != Wo_min)))
        return fail(nullptr, VSTAB_E_SHAPE, "conv_dgrad: %dx%d input, k %d stride %d pad %d does not match a %dx%d output", Hi, Wi, k,
                    stride, pad, Ho, Wo);
    if ((cin & 3) || (cout & 3) || (cs_x & 3) || (cx_off & 3) || (cs_g & 3) || (cg_off & 3))
        return fail(nullptr, VSTAB_E_ALIGN, "conv_dgrad: channel counts, strides and offsets must be multiples of 4");
    if ((long long)B * Hi * Wi * cs_x * 4 >= 0x80000000LL || (long long)B * Ho * Wo * cs_g * 4 >= 0x80000000LL)
        return fail(nullptr, VSTAB_E_SHAPE, "conv_dgrad: tensors must stay below 2 GiB");
    if ((reinterpret_cast<uintptr_t>(gout) & 15) || (reinterpret_cast<uintptr_t>(dx) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(nullptr, VSTAB_E_ALIGN, "conv_dgrad: gout / dx must be 16-byte, workspace 256-byte aligned");
    DgradPlan d;
    // the plan reads gout from its first used channel: cs_g stays the pixel stride, the pointer moves by cg_off
    if (!dgrad_plan(B, Ho, Wo, cs_g, cout, k, stride, pad, Hi, Wi, cs_x, cx_off, cin, accumulate ? 1 : 0, d))
        return fail(nullptr, VSTAB_E_SHAPE, "conv_dgrad: unsupported geometry");
    size_t bias_off, part_off;
    const size_t need = dgrad_ws(d, &bias_off, &part_off);
    if (workspace_bytes < need) return fail(nullptr, VSTAB_E_NOMEM, "conv_dgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(nullptr, conv_set_attributes());
    char *ws = reinterpret_cast<char *>(workspace);
    float *wpk = reinterpret_cast<float *>(ws);
    float *bias = reinterpret_cast<float *>(ws + bias_off);
    HIP_TRY(nullptr, launch_pack_apply(W, d.tbl, (long long)d.packed_floats, wpk, st));
    // the bias vector the kernel reads has Npad entries: a shared zero vector when there is none, the caller's own when it is
    // already that long, a padded copy otherwise
    if (!bias_in && d.p.Npad <= 8192 && zero_bias()) bias = const_cast<float *>(zero_bias());
    else if (bias_in && cin == d.p.Npad) bias = const_cast<float *>(bias_in);
    else {
        HIP_TRY(nullptr, hipMemsetAsync(bias, 0, (size_t)d.p.Npad * sizeof(float), st));
        if (bias_in) HIP_TRY(nullptr, hipMemcpyAsync(bias, bias_in, (size_t)cin * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    const int nl = d.nsub ? d.nsub : 1;
    for (int s = 0; s < nl; ++s) {
        ConvParams p = d.nsub ? d.sp[s] : d.p;
        p.in = gout + cg_off;
        p.in_bytes = (unsigned)((long long)B * Ho * Wo * cs_g * 4 - (long long)cg_off * 4);
        p.wpk = wpk + (d.nsub ? d.soff[s] : 0); p.bias = bias; p.out = dx;
        p.partial = reinterpret_cast<float *>(ws + part_off);
        HIP_TRY(nullptr, launch_conv(p, d.nsub ? d.stile[s] : d.tile, d.vec4, st));
    }
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- BatchNorm (training mode) + leaky relu
extern "C" size_t vstab_bn_scratch_bytes(long long rows, int C)
{
    if (rows < 1 || C < 1) return 0;
    return ((size_t)4 * bn_chunks(rows, C) + 2) * C * sizeof(float) + 256;
}

extern "C" int vstab_bn_lrelu_train_forward(float *zy, long long rows, int cs, int c_off, int C, const float *beta, float *moving_mean,
                                            float *moving_var, float decay, float eps, float *save_mean, float *save_rstd, void *scratch,
                                            size_t scratch_bytes, void *stream)
{
    if (!zy || !beta || !save_mean || !save_rstd || !scratch) return fail(nullptr, VSTAB_E_STATE, "bn_lrelu_train_forward: NULL buffer");
    if (rows < 1 || C < 1 || c_off < 0 || c_off + C > cs) return fail(nullptr, VSTAB_E_SHAPE, "bn_lrelu_train_forward: bad shape");
    if ((C & 3) || (cs & 3) || (c_off & 3) || (reinterpret_cast<uintptr_t>(zy) & 15) || (reinterpret_cast<uintptr_t>(beta) & 15) ||
        (reinterpret_cast<uintptr_t>(save_mean) & 15) || (reinterpret_cast<uintptr_t>(save_rstd) & 15))
        return fail(nullptr, VSTAB_E_ALIGN, "bn_lrelu_train_forward: channels multiples of 4, 16-byte aligned buffers");
    if (scratch_bytes < vstab_bn_scratch_bytes(rows, C)) return fail(nullptr, VSTAB_E_NOMEM, "bn_lrelu_train_forward: scratch too small");
    HIP_TRY(nullptr, launch_bn_lrelu_train_forward(zy, rows, cs, c_off, C, beta, moving_mean, moving_var, decay, eps, save_mean, save_rstd,
                                                   reinterpret_cast<float *>(scratch), (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_bn_lrelu_train_backward(const float *y, int cs_y, int cy_off, float *dy, int cs_g, int cg_off, int C, long long rows,
                                             const float *beta, const float *save_rstd, float *dbeta, int accumulate, void *scratch,
                                             size_t scratch_bytes, void *stream)
{
    if (!y || !dy || !beta || !save_rstd || !scratch) return fail(nullptr, VSTAB_E_STATE, "bn_lrelu_train_backward: NULL buffer");
    if (rows < 1 || C < 1 || cy_off < 0 || cg_off < 0 || cy_off + C > cs_y || cg_off + C > cs_g)
        return fail(nullptr, VSTAB_E_SHAPE, "bn_lrelu_train_backward: bad shape");
    if (scratch_bytes < vstab_bn_scratch_bytes(rows, C)) return fail(nullptr, VSTAB_E_NOMEM, "bn_lrelu_train_backward: scratch too small");
    HIP_TRY(nullptr, launch_bn_lrelu_train_backward(y, cs_y, cy_off, dy, cs_g, cg_off, C, rows, beta, save_rstd, dbeta, accumulate ? 1 : 0,
                                                    reinterpret_cast<float *>(scratch), (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_lrelu_backward(const float *y, int cs_y, int cy_off, float *dy, int cs_g, int cg_off, int C, long long rows, void *stream)
{
    if (!y || !dy) return fail(nullptr, VSTAB_E_STATE, "lrelu_backward: NULL buffer");
    if (rows < 1 || C < 1 || cy_off < 0 || cg_off < 0 || cy_off + C > cs_y || cg_off + C > cs_g)
        return fail(nullptr, VSTAB_E_SHAPE, "lrelu_backward: bad shape");
    HIP_TRY(nullptr, launch_lrelu_backward(y, cs_y, cy_off, dy, cs_g, cg_off, C, rows, (hipStream_t)stream));
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- forward conv with device-resident raw weights
namespace {
bool fwd_plan(int B, int Hi, int Wi, int cs_x, int cin, int k, int stride, int pad, int cout, int cs_y, int cy_off, int act, int Ho, int Wo,
              DgradPlan &out)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int, int, int, int, int, int, int, int, int, int, int>, DgradPlan> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;             // (a table-packed plan owns device memory)
    const auto key = std::make_tuple(dev, B, Hi, Wi, cs_x, cin, k, stride, pad, cout, cs_y, cy_off, act, Ho, Wo);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) { out = it->second; return true; }
    DgradPlan d{};
    if (!fill_plain_conv(d.p, d.tile, d.vec4, B, Hi, Wi, cin, cs_x, k, stride, pad, cout, cs_y, cy_off, act)) return false;
    if (Ho > 0 && Wo > 0 && (Ho != d.p.Ho || Wo != d.p.Wo)) {
        // one more output row / column than the symmetric-pad formula gives (TF SAME on an odd size): its window sticks out
        // of the image and the range checks zero those taps
        if (Ho < d.p.Ho || Wo < d.p.Wo || Ho > d.p.Ho + 1 || Wo > d.p.Wo + 1) return false;
        if ((long long)B * Ho * Wo * cs_y * 4 >= 0x80000000LL) return false;
        d.p.Ho = Ho; d.p.Wo = Wo;
        d.p.ph[0].Hg = Ho; d.p.ph[0].Wg = Wo; d.p.ph[0].M = B * Ho * Wo; d.p.Mmax = d.p.ph[0].M;
        const ConvTile base = cout >= 128 ? TILE_128x128 : (cout > 32 ? TILE_128x64 : TILE_128x32);
        d.tile = choose_tile_split(d.p, base, d.vec4);
    }
    const KLayout L{d.p.KH, d.p.NSEG, d.p.SEG, d.p.SEGP, d.p.SEG_STRIDE};
    d.packed_floats = (size_t)L.ktiles() * d.p.Npad * 32;
    if (d.p.ksplit > 1) d.part_floats = (size_t)d.p.nphase * d.p.ksplit * d.p.Mmax * d.p.Npad;
    if (L.NSEG == 1 && L.SEG == L.SEGP && cs_x == cin && (cout & 3) == 0 && (d.p.Npad & 63) == 0) {
        d.blocked_K = L.KH * L.SEG;              // HWIO with its first three axes flattened IS the [K][N] matrix
    } else {
        std::vector<int32_t> tbl(d.packed_floats);
        pack_index_conv(k, k, cin, cs_x, cout, d.p.Npad, L, tbl.data());
        d.tbl_nfast = table_is_n_fast(tbl, d.p.Npad);
        if (hipMalloc(reinterpret_cast<void **>(&d.tbl), tbl.size() * sizeof(int32_t)) != hipSuccess) return false;
        if (hipMemcpy(d.tbl, tbl.data(), tbl.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d.tbl); return false; }
    }
    cache[key] = d;
    out = d;
    return true;
}
}  // namespace

extern "C" size_t vstab_conv_forward_workspace_bytes(int B, int Hi, int Wi, int cs_x, int cin, int k, int stride, int pad, int cout, int cs_y,
                                                     int cy_off, int act, int Ho, int Wo)
{
    DgradPlan d;
    if (!fwd_plan(B, Hi, Wi, cs_x, cin, k, stride, pad, cout, cs_y, cy_off, act, Ho, Wo, d)) return 0;
    size_t a, b;
    return dgrad_ws(d, &a, &b);
}

extern "C" int vstab_conv_forward(const float *x, int B, int Hi, int Wi, int cs_x, int cx_off, int cin, const float *W, const float *bias_in,
                                  int k, int stride, int pad, float *y, int Ho, int Wo, int cs_y, int cy_off, int cout, int act,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !W || !y || !workspace) return fail(nullptr, VSTAB_E_STATE, "conv_forward: NULL buffer");
    if (B < 1 || Hi < 1 || Wi < 1 || k < 1 || k > 7 || stride < 1 || pad < 0 || cin < 1 || cout < 1 || cx_off < 0 || cy_off < 0 ||
        cx_off + cin > cs_x || cy_off + cout > cs_y || act < 0 || act > 3)
        return fail(nullptr, VSTAB_E_SHAPE, "conv_forward: bad shape");
    if ((cx_off & 3) || (cy_off & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(nullptr, VSTAB_E_ALIGN, "conv_forward: channel offsets multiples of 4, x 16-byte, workspace 256-byte aligned");
    DgradPlan d;
    if (!fwd_plan(B, Hi, Wi, cs_x, cin, k, stride, pad, cout, cs_y, cy_off, act, Ho, Wo, d))
        return fail(nullptr, VSTAB_E_SHAPE, "conv_forward: unsupported geometry");
    size_t bias_off, part_off;
    const size_t need = dgrad_ws(d, &bias_off, &part_off);
    if (workspace_bytes < need) return fail(nullptr, VSTAB_E_NOMEM, "conv_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(nullptr, conv_set_attributes());
    char *ws = reinterpret_cast<char *>(workspace);
    float *wpk = reinterpret_cast<float *>(ws);
    float *bias = reinterpret_cast<float *>(ws + bias_off);
    if (d.blocked_K) HIP_TRY(nullptr, launch_pack_blocked(W, d.blocked_K, cout, d.p.Npad, 1, wpk, st));
    else HIP_TRY(nullptr, launch_pack_apply(W, d.tbl, (long long)d.packed_floats, wpk, st, d.tbl_nfast));
    if (!bias_in && d.p.Npad <= 8192 && zero_bias()) bias = const_cast<float *>(zero_bias());
    else if (bias_in && cout == d.p.Npad) bias = const_cast<float *>(bias_in);
    else {
        HIP_TRY(nullptr, hipMemsetAsync(bias, 0, (size_t)d.p.Npad * sizeof(float), st));
        if (bias_in) HIP_TRY(nullptr, hipMemcpyAsync(bias, bias_in, (size_t)cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    ConvParams p = d.p;
    p.in = x + cx_off;
    p.in_bytes = (unsigned)((long long)B * Hi * Wi * cs_x * 4 - (long long)cx_off * 4);
    p.wpk = wpk; p.bias = bias; p.out = y;
    p.partial = reinterpret_cast<float *>(ws + part_off);
    HIP_TRY(nullptr, launch_conv(p, d.tile, d.vec4, st));
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- the first layer on the inference path's row-window kernel
// model.py:807-808 (PadLayer(3) -> Conv2d 7x7 stride 2 VALID, 27 -> 64) with DEVICE-resident raw weights: conv_rowwin_kernel (stream form,
// assembly K loop: 0.92 of the MFMA peak at B=8 512x512, where the generic implicit GEMM reaches 0.73) fed by an operand gathered on
// the device every call -- the host packer's layout, recorded once per geometry as an index table by packing a tensor of indices.
namespace {
struct RowWinPlan { RowWinParams r; int32_t *tbl; size_t packed_floats; bool two; RowWinParams t; };

bool rowwin_plan(int B, int H, int W, int Cin, int cs_w, int cout, int k, int stride, int pad, int cs_y, int cy_off, int act, RowWinPlan &out)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int, int, int, int, int, int, int, int, int>, RowWinPlan> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const auto key = std::make_tuple(dev, B, H, W, Cin, cs_w, cout, k, stride, pad, cs_y, cy_off, act);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) { out = it->second; return true; }
    if (cout < 1 || cout > 64 || cs_w < Cin || k < 1 || k > 7 || stride < 1 || pad < 0 || (cy_off & 3) || (cs_y & 3)) return false;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Ho < 1 || Wo < 1 || (long long)B * H * W * Cin * 4 >= 0x80000000LL || (long long)B * Ho * Wo * cs_y * 4 >= 0x80000000LL) return false;
    RowWinPlan d{};
    RowWinParams &r = d.r;
    r.in_bytes = (unsigned)((long long)B * H * W * Cin * 4);
    r.B = B; r.Hi = H; r.Wi = W; r.Cs_in = Cin; r.KH = k;
    r.SEGP = rowwin_segp(-pad, k, Cin);
    r.s_in = stride; r.off_y = -pad;
    r.e_off = -pad * Cin - rowwin_lead(-pad, Cin);
    r.w_a = ((r.e_off % 4) + 4) % 4;
    r.MB = rowwin_mb(B, Ho, Wo);
    r.WLEN = round_up(r.s_in * Cin * (64 * r.MB - 1) + r.w_a + r.SEGP, 4);
    r.Ho = Ho; r.Wo = Wo; r.Cs_out = cs_y; r.c_off = cy_off; r.N = cout; r.Npad = 64; r.act = act;
    r.in = reinterpret_cast<const float *>(16);       // (rowwin_applicable tests the alignment of the real pointer at launch time)
    if (!rowwin_applicable(r)) return false;
    const int rem = Wo % 128;
    if (r.MB == 2 && Wo > 128 && rem >= 1 && rem <= 64) {      // 128 k + (1..64) columns: k full tiles, then ONE 64-pixel tile (api.cpp's rule)
        RowWinParams t = r;
        t.MB = 1; t.ox_base = (Wo / 128) * 128; t.ntile_x = 1;
        t.WLEN = round_up(t.s_in * Cin * 63 + t.w_a + t.SEGP, 4);
        if (rowwin_applicable(t)) { r.ntile_x = Wo / 128; d.two = true; d.t = t; }
    }
    // the packer's layout as a gather table: pack a filter whose value at (ky, kx, ci, n) is 1 + its index in the [k,k,cs_w,cout] tensor
    const int lead = rowwin_lead(-pad, Cin);
    d.packed_floats = (size_t)k * (r.SEGP / 32) * 64 * 32;
    std::vector<float> idx((size_t)k * k * Cin * cout), packed(d.packed_floats);
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx)
            for (int ci = 0; ci < Cin; ++ci)
                for (int n = 0; n < cout; ++n)
                    idx[(((size_t)ky * k + kx) * Cin + ci) * cout + n] = (float)(1 + (((size_t)ky * k + kx) * cs_w + ci) * cout + n);
    if ((size_t)k * k * cs_w * cout >= (1u << 24)) return false;      // indices must be exact in fp32
    std::vector<double> ones(64, 1.0);
    pack_conv_rowwin(idx.data(), ones.data(), k, k, Cin, cout, 64, lead, r.SEGP, packed.data());
    std::vector<int32_t> tbl(d.packed_floats);
    for (size_t i = 0; i < tbl.size(); ++i) tbl[i] = (int32_t)packed[i];
    if (hipMalloc(reinterpret_cast<void **>(&d.tbl), tbl.size() * sizeof(int32_t)) != hipSuccess) return false;
    if (hipMemcpy(d.tbl, tbl.data(), tbl.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d.tbl); return false; }
    cache[key] = d;
    out = d;
    return true;
}
}  // namespace

extern "C" size_t vstab_conv_rowwin_forward_workspace_bytes(int B, int H, int W, int Cin, int cs_w, int cout, int k, int stride, int pad, int cs_y,
                                                            int cy_off, int act)
{
    RowWinPlan d;
    if (!rowwin_plan(B, H, W, Cin, cs_w, cout, k, stride, pad, cs_y, cy_off, act, d)) return 0;
    return (d.packed_floats * sizeof(float) + 255) / 256 * 256 + 512;
}

extern "C" int vstab_conv_rowwin_forward(const float *x, int B, int H, int W, int Cin, const float *Wf, int cs_w, int cout, const float *bias, int k,
                                         int stride, int pad, float *y, int cs_y, int cy_off, int act, void *workspace, size_t workspace_bytes,
                                         void *stream)
{
    if (!x || !Wf || !y || !workspace) return fail(nullptr, VSTAB_E_STATE, "conv_rowwin_forward: NULL buffer");
    if (act < 0 || act > 2) return fail(nullptr, VSTAB_E_SHAPE, "conv_rowwin_forward: act must be 0, 1 or 2");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(nullptr, VSTAB_E_ALIGN, "conv_rowwin_forward: x and y 16-byte, workspace 256-byte aligned");
    RowWinPlan d;
    if (!rowwin_plan(B, H, W, Cin, cs_w, cout, k, stride, pad, cs_y, cy_off, act, d))
        return fail(nullptr, VSTAB_E_SHAPE, "conv_rowwin_forward: the row-window kernel does not take this geometry");
    const size_t poff = (d.packed_floats * sizeof(float) + 255) / 256 * 256;
    if (workspace_bytes < poff + 512) return fail(nullptr, VSTAB_E_NOMEM, "conv_rowwin_forward: workspace %zu < %zu bytes", workspace_bytes, poff + 512);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(nullptr, rowwin_set_attributes());
    float *wpk = reinterpret_cast<float *>(workspace);
    float *b64 = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + poff);
    HIP_TRY(nullptr, launch_pack_apply(Wf, d.tbl, (long long)d.packed_floats, wpk, st));
    HIP_TRY(nullptr, hipMemsetAsync(b64, 0, 64 * sizeof(float), st));
    if (bias) HIP_TRY(nullptr, hipMemcpyAsync(b64, bias, (size_t)cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    RowWinParams r = d.r;
    r.in = x; r.out = y; r.wpk = wpk; r.bias = b64;
    if (d.two) {
        RowWinParams t = d.t;
        t.in = x; t.out = y; t.wpk = wpk; t.bias = b64;
        HIP_TRY(nullptr, launch_conv_rowwin(r, st));
        HIP_TRY(nullptr, launch_conv_rowwin(t, st));
    } else HIP_TRY(nullptr, launch_conv_rowwin(r, st));
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- resampler adjoints, full-res upsampler, Adam
extern "C" int vstab_resize_bilinear_backward(const float *dout, int B, int oh, int ow, int C, float *din, int h, int w, float gain,
                                              int accumulate, void *stream)
{
    if (!dout || !din) return fail(nullptr, VSTAB_E_STATE, "resize_bilinear_backward: NULL buffer");
    if (B < 1 || oh < 1 || ow < 1 || h < 1 || w < 1 || C < 1) return fail(nullptr, VSTAB_E_SHAPE, "resize_bilinear_backward: bad shape");
    HIP_TRY(nullptr, launch_resize_bilinear_backward(dout, B, oh, ow, C, din, h, w, gain, accumulate ? 1 : 0, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_pad_nearest_upsample(const float *src, int B, int h2, int w2, int C, float *out, int H, int W, void *stream)
{
    if (!src || !out) return fail(nullptr, VSTAB_E_STATE, "pad_nearest_upsample: NULL buffer");
    if (B < 1 || h2 < 1 || w2 < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return fail(nullptr, VSTAB_E_SHAPE, "pad_nearest_upsample: bad shape");
    if ((long long)B * H * W * C * 4 >= (1LL << 40)) return fail(nullptr, VSTAB_E_SHAPE, "pad_nearest_upsample: too large");
    HIP_TRY(nullptr, launch_pad_nearest_up(src, B, h2, w2, C, out, H, W, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_pad_nearest_upsample_backward(const float *dout, int B, int H, int W, int C, float *dsrc, int h2, int w2, int accumulate,
                                                   void *stream)
{
    if (!dout || !dsrc) return fail(nullptr, VSTAB_E_STATE, "pad_nearest_upsample_backward: NULL buffer");
    if (B < 1 || h2 < 1 || w2 < 1 || H < 1 || W < 1 || C < 4 || (C & 3))
        return fail(nullptr, VSTAB_E_SHAPE, "pad_nearest_upsample_backward: bad shape");
    HIP_TRY(nullptr, launch_pad_nearest_up_backward(dout, B, H, W, C, dsrc, h2, w2, accumulate ? 1 : 0, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_adam_step(float *w, const float *g, float *m, float *v, long long n, float lr_t, float beta1, float beta2, float eps,
                               void *stream)
{
    if (!w || !g || !m || !v) return fail(nullptr, VSTAB_E_STATE, "adam_step: NULL buffer");
    if (n < 1) return fail(nullptr, VSTAB_E_SHAPE, "adam_step: bad size");
    HIP_TRY(nullptr, launch_adam(w, g, m, v, n, lr_t, beta1, beta2, eps, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" size_t vstab_column_sum_scratch_bytes(long long rows, int C) { return rows < 1 || C < 1 ? 0 : (size_t)column_sum_chunks(rows, C) * C * sizeof(float) + 256; }

extern "C" int vstab_column_sum(const float *g, long long rows, int cs, int c_off, int C, float *out, int accumulate, void *scratch,
                                size_t scratch_bytes, void *stream)
{
    if (!g || !out || !scratch) return fail(nullptr, VSTAB_E_STATE, "column_sum: NULL buffer");
    if (rows < 1 || C < 1 || c_off < 0 || c_off + C > cs) return fail(nullptr, VSTAB_E_SHAPE, "column_sum: bad shape");
    if (scratch_bytes < vstab_column_sum_scratch_bytes(rows, C)) return fail(nullptr, VSTAB_E_NOMEM, "column_sum: scratch too small");
    HIP_TRY(nullptr, launch_column_sum(g, rows, cs, c_off, C, out, accumulate ? 1 : 0, reinterpret_cast<float *>(scratch), (hipStream_t)stream));
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- full-resolution head through its tap table
extern "C" int vstab_pf2_from_taps(const float *T, int B, int h2, int w2, const float *bias2, const float *pf3, int h3, int w3, float *pf2,
                                   int H, int W, void *stream)
{
    if (!T || !bias2 || !pf3 || !pf2) return fail(nullptr, VSTAB_E_STATE, "pf2_from_taps: NULL buffer");
    if (B < 1 || h2 < 1 || w2 < 1 || h3 < 1 || w3 < 1 || H < 3 || W < 3) return fail(nullptr, VSTAB_E_SHAPE, "pf2_from_taps: bad shape");
    HIP_TRY(nullptr, launch_pf2(T, B, h2, w2, bias2, pf3, h3, w3, pf2, H, W, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_predict2_tap_table(const float *concat2, long long M, const float *table, float *T, void *stream)
{
    if (!concat2 || !table || !T) return fail(nullptr, VSTAB_E_STATE, "predict2_tap_table: NULL buffer");
    if (M < 1 || M * 784 >= 0x80000000LL) return fail(nullptr, VSTAB_E_SHAPE, "predict2_tap_table: 1 <= M, M * 784 < 2^31");
    if (((uintptr_t)concat2 | (uintptr_t)T | (uintptr_t)table) & 15) return fail(nullptr, VSTAB_E_ALIGN, "predict2_tap_table: buffers must be 16-byte aligned");
    static const hipError_t attr = tap_panel_set_attributes();
    HIP_TRY(nullptr, attr);
    HIP_TRY(nullptr, launch_tap_panel(concat2, M, table, T, (hipStream_t)stream));
    return VSTAB_OK;
}

extern "C" int vstab_pf2_taps_backward(const float *g, int cs_g, int B, int H, int W, float *dT, int h2, int w2, void *stream)
{
    if (!g || !dT) return fail(nullptr, VSTAB_E_STATE, "pf2_taps_backward: NULL buffer");
    if (B < 1 || h2 < 1 || w2 < 1 || H < 3 || W < 3 || cs_g < 2) return fail(nullptr, VSTAB_E_SHAPE, "pf2_taps_backward: bad shape");
    HIP_TRY(nullptr, launch_pf2_taps_backward(g, cs_g, B, H, W, dT, h2, w2, (hipStream_t)stream));
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- 3x3 stride-1 conv / its input gradient in Winograd form
namespace {
struct WinoPlan {
    ConvParams p;               // the 16-phase 1x1 GEMM over the transformed tiles
    size_t phase_floats;        // packed floats per position
    int K, N, TH, TW;
};

bool wino_plan(int B, int H, int W, int K, int N, WinoPlan &out)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int>, WinoPlan> cache;
    const auto key = std::make_tuple(B, H, W, K, N);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) { out = it->second; return true; }
    if ((K & 31) || (N & 63)) return false;
    WinoPlan d{};
    d.K = K; d.N = N; d.TH = (H + 1) / 2; d.TW = (W + 1) / 2;
    if ((long long)B * 16 * d.TH * d.TW * std::max(K, N) * 4 >= 0x80000000LL) return false;
    ConvParams &p = d.p;
    std::memset(&p, 0, sizeof p);
    p.B = B; p.Hi = 16 * d.TH; p.Wi = d.TW; p.Cs_in = K;
    const KLayout L = klayout_run(1, 1, K);
    set_layout(p, L);
    p.s_in = 1; p.s_out = 1; p.Ho = 16 * d.TH; p.Wo = d.TW; p.Cs_out = N; p.c_off = 0;
    p.N = N; p.Npad = N; p.act = 0; p.nphase = 16; p.ksplit = 1;
    d.phase_floats = (size_t)L.ktiles() * p.Npad * 32;
    for (int xi = 0; xi < 16; ++xi) {
        ConvPhase &ph = p.ph[xi];
        ph.Hg = d.TH; ph.Wg = d.TW; ph.M = B * d.TH * d.TW;
        ph.off_y = xi * d.TH; ph.o_y = xi * d.TH;
        ph.w_off = (long long)xi * d.phase_floats;
    }
    p.Mmax = B * d.TH * d.TW;
    set_ranges(p);
    cache[key] = d;
    out = d;
    return true;
}

size_t a256(size_t n) { return (n + 255) / 256 * 256; }
}  // namespace

extern "C" size_t vstab_conv3x3_winograd_workspace_bytes(int B, int H, int W, int cin, int cout, int transpose)
{
    WinoPlan d;
    const int K = transpose ? cout : cin, N = transpose ? cin : cout;
    if (B < 1 || H < 1 || W < 1 || !wino_plan(B, H, W, K, N, d)) return 0;
    const size_t tiles = (size_t)B * 16 * d.TH * d.TW;
    return a256((size_t)16 * K * N * 4) + a256(16 * d.phase_floats * 4) + a256(tiles * K * 4) + a256(tiles * N * 4) + a256((size_t)N * 4) + 256;
}

extern "C" int vstab_conv3x3_winograd(const float *x, int B, int H, int W, int cs_x, int cx_off, const float *Wf, int cin, int cout, int transpose,
                                      const float *bias, float *y, int cs_y, int cy_off, int act, void *workspace, size_t workspace_bytes,
                                      void *stream)
{
    if (!x || !Wf || !y || !workspace) return fail(nullptr, VSTAB_E_STATE, "conv3x3_winograd: NULL buffer");
    const int K = transpose ? cout : cin, N = transpose ? cin : cout;             // reduction / output channels of this call
    if (B < 1 || H < 1 || W < 1 || cin < 1 || cout < 1 || cx_off < 0 || cy_off < 0 || cx_off + K > cs_x || cy_off + N > cs_y ||
        (act != 0 && act != 1 && act != 3))
        return fail(nullptr, VSTAB_E_SHAPE, "conv3x3_winograd: bad shape");
    if ((cs_x & 3) || (cx_off & 3) || (cs_y & 3) || (cy_off & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
        (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(nullptr, VSTAB_E_ALIGN, "conv3x3_winograd: strides / offsets multiples of 4, 16-byte tensors, 256-byte workspace");
    WinoPlan d;
    if (!wino_plan(B, H, W, K, N, d)) return fail(nullptr, VSTAB_E_SHAPE, "conv3x3_winograd: needs K %% 32 == 0, N %% 64 == 0 and < 2 GiB per tensor");
    if (workspace_bytes < vstab_conv3x3_winograd_workspace_bytes(B, H, W, cin, cout, transpose))
        return fail(nullptr, VSTAB_E_NOMEM, "conv3x3_winograd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(nullptr, conv_set_attributes());
    char *ws = reinterpret_cast<char *>(workspace);
    const size_t tiles = (size_t)B * 16 * d.TH * d.TW;
    float *Wt = reinterpret_cast<float *>(ws); ws += a256((size_t)16 * K * N * 4);
    float *wpk = reinterpret_cast<float *>(ws); ws += a256(16 * d.phase_floats * 4);
    float *V = reinterpret_cast<float *>(ws); ws += a256(tiles * K * 4);
    float *M = reinterpret_cast<float *>(ws); ws += a256(tiles * N * 4);
    float *bz = reinterpret_cast<float *>(ws);
    HIP_TRY(nullptr, launch_wino_weights(Wf, cin, cout, transpose, Wt, st));
    HIP_TRY(nullptr, launch_pack_blocked(Wt, K, N, N, 16, wpk, st));           // 16 x [K][N] -> 16 x [K/32][N][32]
    HIP_TRY(nullptr, hipMemsetAsync(bz, 0, (size_t)N * sizeof(float), st));
    HIP_TRY(nullptr, launch_wino_input(x, B, H, W, cs_x, cx_off, K, V, st));
    ConvParams p = d.p;
    p.in = V; p.out = M; p.wpk = wpk; p.bias = bz; p.partial = nullptr;
    HIP_TRY(nullptr, launch_conv(p, TILE_128x64, true, st));
    HIP_TRY(nullptr, launch_wino_output(M, B, H, W, N, bias ? bias : bz, act, y, cs_y, cy_off, st));
    return VSTAB_OK;
}

// ------------------------------------------------------------------------- filter gradient of a 3x3 stride-1 conv in Winograd form
namespace {
bool wino_wgrad_shape(int B, int H, int W, int cin, int cout, WgradParams &p, int &TH, int &TW)
{
    if (B < 1 || H < 1 || W < 1 || cin < 4 || cout < 4 || (cin & 3) || (cout & 3)) return false;
    TH = (H + 1) / 2; TW = (W + 1) / 2;
    const long long T = (long long)B * TH * TW;
    if (T * 16 * std::max(cin, cout) * 4 >= 0x80000000LL || T * std::max(cin, cout) * 4 >= 0x80000000LL) return false;
    p = WgradParams{};
    p.Hi = (int)T; p.Wi = 1; p.Cs_x = cin; p.cx_off = 0; p.Cin = cin; p.KH = 1; p.KW = 1;
    p.Cs_g = cout; p.cg_off = 0; p.Cout = cout; p.M = cin; p.K = (int)T;
    p.nbatch = 16;
    p.ksplit = wgrad_choose_split(p);
    return true;
}
}  // namespace

extern "C" size_t vstab_conv3x3_winograd_wgrad_workspace_bytes(int B, int H, int W, int cin, int cout)
{
    WgradParams p; int TH, TW;
    if (!wino_wgrad_shape(B, H, W, cin, cout, p, TH, TW)) return 0;
    const size_t T = (size_t)B * TH * TW;
    return a256(T * 16 * cin * 4) + a256(T * 16 * cout * 4) + a256((size_t)16 * cin * cout * 4) +
           a256(p.ksplit > 1 ? (size_t)16 * p.ksplit * cin * cout * 4 : 0) + 256;
}

extern "C" int vstab_conv3x3_winograd_wgrad(const float *x, int B, int H, int W, int cs_x, int cx_off, int cin, const float *gout, int cs_g,
                                            int cg_off, int cout, float *dW, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !gout || !dW || !workspace) return fail(nullptr, VSTAB_E_STATE, "conv3x3_winograd_wgrad: NULL buffer");
    if (cx_off < 0 || cg_off < 0 || cx_off + cin > cs_x || cg_off + cout > cs_g)
        return fail(nullptr, VSTAB_E_SHAPE, "conv3x3_winograd_wgrad: bad channel slices");
    WgradParams p; int TH, TW;
    if (!wino_wgrad_shape(B, H, W, cin, cout, p, TH, TW))
        return fail(nullptr, VSTAB_E_SHAPE, "conv3x3_winograd_wgrad: needs channel counts that are multiples of 4 and < 2 GiB per tensor");
    if ((cs_x & 3) || (cx_off & 3) || (cs_g & 3) || (cg_off & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(gout) & 15) ||
        (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(nullptr, VSTAB_E_ALIGN, "conv3x3_winograd_wgrad: strides / offsets multiples of 4, 16-byte tensors, 256-byte workspace");
    if (workspace_bytes < vstab_conv3x3_winograd_wgrad_workspace_bytes(B, H, W, cin, cout))
        return fail(nullptr, VSTAB_E_NOMEM, "conv3x3_winograd_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t T = (size_t)B * TH * TW;
    char *ws = reinterpret_cast<char *>(workspace);
    float *V = reinterpret_cast<float *>(ws); ws += a256(T * 16 * cin * 4);
    float *dM = reinterpret_cast<float *>(ws); ws += a256(T * 16 * cout * 4);
    float *dU = reinterpret_cast<float *>(ws); ws += a256((size_t)16 * cin * cout * 4);
    p.partial = reinterpret_cast<float *>(ws);
    HIP_TRY(nullptr, launch_wino_input(x, B, H, W, cs_x, cx_off, cin, V, st, true));           // V  [16][T][cin]
    HIP_TRY(nullptr, launch_wino_outgrad(gout, B, H, W, cs_g, cg_off, cout, dM, st));            // dM [16][T][cout]
    // 16 independent reductions dU_xi [cin x cout] = V_xi^T [cin x T] . dM_xi [T x cout]: the weight-gradient kernel's batched form over
    // a 1x1 "image" of T pixels
    p.ptab = wgrad_pixel_table(1, (int)T, 1, cin, (int)T, 1, 1, 0, st);
    if (!p.ptab) return fail(nullptr, VSTAB_E_NOMEM, "conv3x3_winograd_wgrad: cannot build the pixel table");
    p.x = V; p.g = dM; p.dW = dU;
    p.x_bytes = (unsigned)(T * cin * 4); p.g_bytes = (unsigned)(T * cout * 4);
    p.x_bstride = (long long)(T * cin); p.g_bstride = (long long)(T * cout);
    p.accumulate = 0;
    HIP_TRY(nullptr, launch_wgrad(p, st));
    HIP_TRY(nullptr, launch_wino_filter_grad(dU, cin, cout, dW, st));
    return VSTAB_OK;
}
