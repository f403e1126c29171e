// VALU kernels of the flow pyramid and the warp: everything on the hot path that is not
// a wide-output convolution.  All are HBM/latency-bound gathers and lerps (~1 FLOP/byte).
// Built with -ffp-contract=off so that the lerp / weight arithmetic is the same sequence
// of fp32 roundings the reference's TF graph performs; dot products use explicit fmaf.
#include "vstab_internal.h"
#include "hbm_profile.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace vstab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- records of the optional per-launch timing of the HBM-side kernels (hbm_profile.h has the launch wrapper)
struct HbmProfState {
    std::mutex mu;
    bool on = false;
    struct Rec { hipEvent_t a, b; };
    std::vector<Rec> recs[HBM_SLOTS];
    double bytes[HBM_SLOTS] = {0};
};
static HbmProfState &hbm_prof() { static HbmProfState s; return s; }

void hbm_profile_enable(int mode)      // 0 = off (records kept), 1 = clear the records and switch on, 2 = switch on again, records kept
{
    HbmProfState &P = hbm_prof();
    std::lock_guard<std::mutex> g(P.mu);
    const bool on = mode != 0;
    if (mode == 1)
        for (int i = 0; i < HBM_SLOTS; ++i) {
            for (auto &r : P.recs[i]) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
            P.recs[i].clear(); P.bytes[i] = 0;
        }
    P.on = on;
}

hipError_t hbm_profile_read(int slot, double *ms_sum, int *launches, double *alg_bytes_sum)
{
    HbmProfState &P = hbm_prof();
    std::lock_guard<std::mutex> g(P.mu);
    if (slot < 0 || slot >= HBM_SLOTS) return hipErrorInvalidValue;
    double ms = 0;
    for (auto &r : P.recs[slot]) {
        float m = 0.f;
        const hipError_t e = hipEventElapsedTime(&m, r.a, r.b);      // the stream must have been synchronised
        if (e != hipSuccess) return e;
        ms += m;
    }
    *ms_sum = ms; *launches = (int)P.recs[slot].size(); *alg_bytes_sum = P.bytes[slot];
    return hipSuccess;
}

hipError_t hbm_profile_begin(int slot, double alg_bytes, hipEvent_t *a, hipEvent_t *b)
{
    HbmProfState &P = hbm_prof();
    *a = *b = nullptr;
    if (!P.on) return hipSuccess;
    HbmProfState::Rec r;
    hipError_t e = hipEventCreate(&r.a);
    if (e != hipSuccess) return e;
    e = hipEventCreate(&r.b);
    if (e != hipSuccess) { (void)hipEventDestroy(r.a); return e; }
    std::lock_guard<std::mutex> g(P.mu);
    P.recs[slot].push_back(r); P.bytes[slot] += alg_bytes;
    *a = r.a; *b = r.b;
    return hipSuccess;
}

// ---- legacy TF bilinear (ResizeBilinear, align_corners=False, no half-pixel centres):
// f = i * (in/out) in fp32, lo = floor(f), hi = min(lo+1, in-1), t = f - lo
struct Lerp { int lo, hi; float t; };
__device__ __forceinline__ Lerp legacy_coord(int o, float scale, int n_in)
{
    const float f = (float)o * scale;
    const float fl = floorf(f);
    Lerp L;
    L.lo = min((int)fl, n_in - 1);
    L.hi = min(L.lo + 1, n_in - 1);
    L.t = f - fl;
    return L;
}
__device__ __forceinline__ float lerp2(float tl, float tr, float bl, float br, float tx, float ty)
{
    const float top = tl + (tr - tl) * tx;
    const float bot = bl + (br - bl) * tx;
    return top + (bot - top) * ty;
}

__device__ __forceinline__ f32x2 sample_flow_legacy(const float *f, int n, int h, int w, int oy, int ox,
                                                    float sy, float sx)
{
    const Lerp Y = legacy_coord(oy, sy, h), X = legacy_coord(ox, sx, w);
    const f32x2 *b = reinterpret_cast<const f32x2 *>(f) + (long long)n * h * w;
    const f32x2 tl = b[Y.lo * w + X.lo], tr = b[Y.lo * w + X.hi];
    const f32x2 bl = b[Y.hi * w + X.lo], br = b[Y.hi * w + X.hi];
    f32x2 r;
    r.x = lerp2(tl.x, tr.x, bl.x, br.x, X.t, Y.t);
    r.y = lerp2(tl.y, tr.y, bl.y, br.y, X.t, Y.t);
    return r;
}

// ---------------------------------------------------------------------------------
// predict_flowN (model.py:848,856,865,874): 3x3 pad-1 conv to 2 channels (+bias), then the
// ElementwiseLayer left fold (conv + u) + u with u = legacy-bilinear upsample of the coarser flow
// (model.py:857).  N = 2 cannot feed a 32-wide MFMA tile directly, and a wave-per-pixel dot product
// re-reads every input pixel nine times through L2; instead the MFMA kernel computes, once per
// SOURCE pixel s, the tap table T[s][tap][o] = sum_c x[s][c] W[tap][c][o] (a 1x1 conv to 18
// columns), and this kernel gathers out[y,x,o] = b[o] + sum_{dy,dx} T[(y+dy-1, x+dx-1)][3dy+dx][o]
// over the in-image taps: the same sum in a different association.
//
// One launch per pyramid level does the whole head: it also sums the split-K slabs of the tap table (the MFMA
// launch skips its combine pass: `src` = `ks` slabs `slab_stride` floats apart, or the finished table with ks = 1)
// and applies upsample_flowN -- the 2->2 channel 4x4 stride-2 SAME transposed conv + bias (model.py:852;
// o = 2*i + k - 1  =>  for output o the taps are k = (o+1)&1 and k+2, i = (o+1-k)/2) -- to the flow it has just
// produced, writing (u, v, 0, 0) into the next concat's flow channels so its two pad channels are zero every call.
// A workgroup owns a 16x16 tile of flow pixels, computes them plus a one-pixel halo into LDS (the halo is
// recomputed, not exchanged), stores the tile, then emits the 32x32 upsampled outputs that depend on it: three
// dependent launches (combine, gather, upsample) become one.
// ---------------------------------------------------------------------------------
// PU_T = flow pixels per workgroup edge (template parameter: 8, or 4 for the small grids of one sample -- a level of 48 x 64 flow pixels
// is 48 workgroups of 8 x 8 on 256 CUs and a chain of dependent phases each; 192 workgroups of 4 x 4 run the same phases on four
// times the CUs with a quarter of the items per phase.  Same sums per output either way: identical bits).
struct PredictUpArgs {
    const float *src; int ks; long long slab_stride; int h, w; const float *bias2; const float *prev; int ph, pw; float sy, sx;
    float *out; float *concat; int oh, ow, Cs, c_off;
};
// the workgroup program of tile (bx, by) of sample bz (predict_up_kernel; the later workgroups of combine_predict_up_kernel)
template <int PU_T>
__device__ __forceinline__ void predict_up_body(const PredictUpArgs &A, const UpflowW &W, const int bx, const int by, const int bz)
{
    constexpr int PU_H = PU_T + 2;                  // + halo
    constexpr int PU_S = PU_T + 4;                  // + the source pixels the halo's 3x3 taps reach
    const float *__restrict__ src = A.src; const int ks = A.ks; const long long slab_stride = A.slab_stride; const int h = A.h, w = A.w;
    const float *__restrict__ bias2 = A.bias2; const float *__restrict__ prev = A.prev; const int ph = A.ph, pw = A.pw;
    const float sy = A.sy, sx = A.sx; float *__restrict__ out = A.out; float *__restrict__ concat = A.concat;
    const int oh = A.oh, ow = A.ow, Cs = A.Cs, c_off = A.c_off;
    __shared__ float tab[PU_S * PU_S * 18];      // combined tap-table entries of the source pixels (18 used columns)
    __shared__ f32x2 tile[PU_H * PU_H];
    const int n = bz, y0 = by * PU_T, x0 = bx * PU_T;
    const float *Tn = src + (long long)n * h * w * 32;
    // stage A: sum the split-K slabs, one (source pixel, column pair) per work item -- PU_S^2 * 9 independent items,
    // four slab loads in flight each, added in slab order (the combine pass's association)
    for (int i = threadIdx.x; i < PU_S * PU_S * 9; i += 256) {
        const int s = i / 9, t = i - s * 9;
        const int y = y0 - 2 + s / PU_S, x = x0 - 2 + s % PU_S;
        float t0 = 0.f, t1 = 0.f;
        if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) {
            const float *tp = Tn + ((long long)y * w + x) * 32 + t * 2;
            int k = 0;
            for (; k + 4 <= ks; k += 4) {
                const f32x2 v0 = *reinterpret_cast<const f32x2 *>(tp + (k + 0) * slab_stride);
                const f32x2 v1 = *reinterpret_cast<const f32x2 *>(tp + (k + 1) * slab_stride);
                const f32x2 v2 = *reinterpret_cast<const f32x2 *>(tp + (k + 2) * slab_stride);
                const f32x2 v3 = *reinterpret_cast<const f32x2 *>(tp + (k + 3) * slab_stride);
                t0 += v0.x; t1 += v0.y; t0 += v1.x; t1 += v1.y; t0 += v2.x; t1 += v2.y; t0 += v3.x; t1 += v3.y;
            }
            for (; k < ks; ++k) {
                const f32x2 v = *reinterpret_cast<const f32x2 *>(tp + k * slab_stride);
                t0 += v.x; t1 += v.y;
            }
        }
        tab[s * 18 + t * 2] = t0;
        tab[s * 18 + t * 2 + 1] = t1;
    }
    __syncthreads();
    // stage B: predict_flowN of the tile + halo: bias + the in-image taps in (dy, dx) order, then the fold with the
    // upsampled coarser flow
    for (int i = threadIdx.x; i < PU_H * PU_H; i += 256) {
        const int ty = i / PU_H, tx = i - ty * PU_H;
        const int y = y0 - 1 + ty, x = x0 - 1 + tx;
        f32x2 o = {0.f, 0.f};
        if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) {
            float a0 = bias2[0], a1 = bias2[1];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int iy = y + dy - 1;
                if (iy < 0 || iy >= h) continue;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int ix = x + dx - 1;
                    if (ix < 0 || ix >= w) continue;
                    const float *e = tab + ((ty + dy) * PU_S + tx + dx) * 18 + (dy * 3 + dx) * 2;
                    a0 += e[0];
                    a1 += e[1];
                }
            }
            if (prev) {
                const f32x2 u = sample_flow_legacy(prev, n, ph, pw, y, x, sy, sx);
                a0 = (a0 + u.x) + u.x;
                a1 = (a1 + u.y) + u.y;
            }
            o.x = a0; o.y = a1;
            if (ty >= 1 && ty <= PU_T && tx >= 1 && tx <= PU_T) reinterpret_cast<f32x2 *>(out)[((long long)n * h + y) * w + x] = o;
        }
        tile[i] = o;
    }
    __syncthreads();
    // stage C: upsample_flowN of the tile
    for (int i = threadIdx.x; i < 4 * PU_T * PU_T; i += 256) {
        const int oy = 2 * y0 + i / (2 * PU_T), ox = 2 * x0 + (i & (2 * PU_T - 1));
        if (oy >= oh || ox >= ow) continue;
        float u = W.b[0], v = W.b[1];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ky = ((oy + 1) & 1) + 2 * a;
            const int iy = (oy + 1 - ky) >> 1;
            if (iy < 0 || iy >= h) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int kx = ((ox + 1) & 1) + 2 * b;
                const int ix = (ox + 1 - kx) >> 1;
                if (ix < 0 || ix >= w) continue;
                const f32x2 f = tile[(iy - y0 + 1) * PU_H + (ix - x0 + 1)];
                const float *wk = W.w + (ky * 4 + kx) * 4;     // [co][ci]
                u = fmaf(f.x, wk[0], u); u = fmaf(f.y, wk[1], u);
                v = fmaf(f.x, wk[2], v); v = fmaf(f.y, wk[3], v);
            }
        }
        f32x4 o = {u, v, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(concat + (((long long)n * oh + oy) * ow + ox) * Cs + c_off) = o;
    }
}

template <int PU_T>
__global__ __launch_bounds__(256) void predict_up_kernel(const PredictUpArgs A, const UpflowW W)
{
    predict_up_body<PU_T>(A, W, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// A refinement level's two small dependent-free passes in ONE launch: the split-K combine of the level's transposed convolution
// (the first nC workgroups; splitk_combine_kernel's work items) and predict_flowN + upsample_flowN (the rest).  They write different
// channel slices of the same concat buffer and read nothing of each other; for one sample each is little more than a launch's fixed
// latency (5.5 us and 8.8 us at 384x512), so sharing the launch hides the shorter one.
template <int PU_T>
__global__ __launch_bounds__(256) void combine_predict_up_kernel(const ConvParams pc, const unsigned nC, const PredictUpArgs A, const UpflowW W,
                                                                 const unsigned tiles_x, const unsigned tiles_y)
{
    if (blockIdx.x < nC) {                          // workgroup uniform
        splitk_combine_item(pc, (long long)blockIdx.x * 256 + threadIdx.x);
        return;
    }
    const unsigned t = blockIdx.x - nC, per = tiles_x * tiles_y;
    const unsigned bz = t / per, r = t - bz * per;
    predict_up_body<PU_T>(A, W, (int)(r % tiles_x), (int)(r / tiles_x), (int)bz);
}

// The same pairing for a transposed convolution in Winograd F(2x2,2x2) form (winograd_ops.hip): its inverse transform (the first nC
// workgroups; wdec_output_kernel's work items, sample-major) beside predict_flowN + upsample_flowN.
template <int PU_T>
__global__ __launch_bounds__(256) void wdec_output_predict_up_kernel(const WdecOutArgs O, const unsigned nC, const unsigned per_sample, const PredictUpArgs A,
                                                                     const UpflowW W, const unsigned tiles_x, const unsigned tiles_y)
{
    if (blockIdx.x < nC) {                          // workgroup uniform
        const unsigned n = blockIdx.x / per_sample, b = blockIdx.x - n * per_sample;
        wdec_output_item(O, (long long)b * 256 + threadIdx.x, (int)n);
        return;
    }
    const unsigned t = blockIdx.x - nC, per = tiles_x * tiles_y;
    const unsigned bz = t / per, r = t - bz * per;
    predict_up_body<PU_T>(A, W, (int)(r % tiles_x), (int)(r / tiles_x), (int)bz);
}

hipError_t launch_predict_up(const float *src, int ks, long long slab_stride, int B, int h, int w, const float *bias2,
                             const float *prev, int ph, int pw, float *out, const UpflowW &W, float *concat, int oh, int ow,
                             int Cs, int c_off, hipStream_t stream, const ConvParams *combine, const WdecOutArgs *wdec)
{
    if ((Cs & 3) || (c_off & 3) || c_off + 4 > Cs || ks < 1 || oh > 2 * h || ow > 2 * w) return hipErrorInvalidValue;
    PredictUpArgs A{src, ks, slab_stride, h, w, bias2, prev, ph, pw, prev ? (float)ph / (float)h : 0.f, prev ? (float)pw / (float)w : 0.f,
                    out, concat, oh, ow, Cs, c_off};
    const int T = (long long)((w + 7) / 8) * ((h + 7) / 8) * B < 256 ? 4 : 8;       // 4 x 4 tiles while 8 x 8 ones would leave CUs idle
    const unsigned tx = (unsigned)((w + T - 1) / T), ty = (unsigned)((h + T - 1) / T);
    if (wdec) {
        const unsigned long long per_sample = (unsigned long long)(((long long)wdec->g.NTy * wdec->g.NTx * 16 * wdec->C4 + 255) / 256);
        const unsigned long long nC = per_sample * (unsigned)B, nP = (unsigned long long)tx * ty * (unsigned)B;
        if (nC + nP >= 0x7fffffffull) return hipErrorInvalidValue;
        if (T == 4) wdec_output_predict_up_kernel<4><<<dim3((unsigned)(nC + nP)), dim3(256), 0, stream>>>(*wdec, (unsigned)nC, (unsigned)per_sample, A, W, tx, ty);
        else wdec_output_predict_up_kernel<8><<<dim3((unsigned)(nC + nP)), dim3(256), 0, stream>>>(*wdec, (unsigned)nC, (unsigned)per_sample, A, W, tx, ty);
        return hipGetLastError();
    }
    if (combine && combine->ksplit > 1) {
        const long long total = (long long)combine->Mmax * (combine->N >> 2) * combine->nphase;
        const unsigned long long nC = (unsigned long long)((total + 255) / 256), nP = (unsigned long long)tx * ty * (unsigned)B;
        if ((combine->N & 3) || nC + nP >= 0x7fffffffull) return hipErrorInvalidValue;
        if (T == 4) combine_predict_up_kernel<4><<<dim3((unsigned)(nC + nP)), dim3(256), 0, stream>>>(*combine, (unsigned)nC, A, W, tx, ty);
        else combine_predict_up_kernel<8><<<dim3((unsigned)(nC + nP)), dim3(256), 0, stream>>>(*combine, (unsigned)nC, A, W, tx, ty);
        return hipGetLastError();
    }
    if (T == 4) predict_up_kernel<4><<<dim3(tx, ty, (unsigned)B), dim3(256), 0, stream>>>(A, W);
    else predict_up_kernel<8><<<dim3(tx, ty, (unsigned)B), dim3(256), 0, stream>>>(A, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// predict_flow2 at full resolution (model.py:882-887).  The reference pads concat2 by one
// pixel, nearest-upsamples it (align_corners=True) to the input's HxW and runs a 3x3 VALID
// conv.  Every tap of that conv reads some source pixel s = (ny(y+dy), nx(x+dx)), so
//   pf2_raw[y,x,o] = b[o] + sum_{dy,dx} T[s][dy*3+dx][o],  T[s][tap][o] = sum_c P[s][c] W[tap][c][o]
// T is a 1x1 conv to 18 channels computed once per source pixel by the MFMA kernel; this
// kernel does the 9 gathers, then pf2 = pf2_raw + 8 sequential adds of up(pf3).
// Index map: src = min((int)roundf(i * (in-1)/(out-1)), in-1) on the padded grid.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int nearest_ac(int i, float scale, int n_in)
{
    return min((int)roundf((float)i * scale), n_in - 1);
}

__global__ __launch_bounds__(256) void pf2_kernel(const float *__restrict__ T, int B, int h2, int w2,
                                                  const float *__restrict__ bias2, const float *__restrict__ pf3,
                                                  int h3, int w3, float *__restrict__ pf2, int H, int W,
                                                  float nsy, float nsx, float usy, float usx)
{
    const int oh = H - 2, ow = W - 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * oh * ow;
    if (idx >= total) return;
    const int n = (int)(idx / (oh * ow));
    const int rem = (int)(idx - (long long)n * oh * ow);
    const int y = rem / ow, x = rem - y * ow;
    int sy[3], sx[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        sy[d] = nearest_ac(y + d, nsy, h2 + 2) - 1;     // index into the unpadded concat2
        sx[d] = nearest_ac(x + d, nsx, w2 + 2) - 1;
    }
    float a0 = bias2[0], a1 = bias2[1];
    const float *Tn = T + (long long)n * h2 * w2 * 32;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        if (sy[dy] < 0 || sy[dy] >= h2) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            if (sx[dx] < 0 || sx[dx] >= w2) continue;
            const f32x2 t = *reinterpret_cast<const f32x2 *>(Tn + ((long long)sy[dy] * w2 + sx[dx]) * 32 + (dy * 3 + dx) * 2);
            a0 += t.x;
            a1 += t.y;
        }
    }
    const f32x2 u = sample_flow_legacy(pf3, n, h3, w3, y, x, usy, usx);
#pragma unroll
    for (int r = 0; r < 8; ++r) { a0 += u.x; a1 += u.y; }
    f32x2 o; o.x = a0; o.y = a1;
    reinterpret_cast<f32x2 *>(pf2)[idx] = o;
}

// Tiled form: a workgroup owns a 16 x 64 tile of output pixels.  The nearest-neighbour map is monotone, so the source pixels the
// tile's 3x3 taps can reach are a small rectangle of the tap table (about (16+2)/4+2 rows x (64+2)/4+2 columns): it is copied
// into LDS once (the 18 used floats of each 128-byte table row; the zero-padding ring as zeros) and every output pixel then takes
// its nine taps as 8-byte LDS reads instead of nine 8-byte global gathers -- the direct kernel spent its time in the L1 tag pipe
// (nine lookups per pixel) with the table itself L2-resident.  Same sums in the same (dy, dx) order; a skipped out-of-image tap
// is now an added +0.0f.
constexpr int PF2_TH = 16, PF2_TW = 64, PF2_PPT = 4, PF2_CAP = 320;      // CAP: window pixels held in LDS (23 KB)
__global__ __launch_bounds__(256) void pf2_tile_kernel(const float *__restrict__ T, int B, int h2, int w2,
                                                       const float *__restrict__ bias2, const float *__restrict__ pf3,
                                                       int h3, int w3, float *__restrict__ pf2, int H, int W,
                                                       float nsy, float nsx, float usy, float usx, int tiles_x, int tiles_y)
{
    __shared__ __attribute__((aligned(8))) float tab[PF2_CAP * 18];
    const int oh = H - 2, ow = W - 2;
    unsigned bx, by, bz;
    xcd_remap_calc(gridDim.x, 1, 1, blockIdx.x, bx, by, bz);
    const int tpi = tiles_x * tiles_y;
    const int n = (int)bx / tpi, trem = (int)bx - n * tpi;
    const int ty0 = (trem / tiles_x) * PF2_TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * PF2_TW;
    // source window (indices into the UNPADDED concat2 grid, so -1 and h2 / w2 are the zero ring)
    const int r_lo = nearest_ac(ty0, nsy, h2 + 2) - 1, r_hi = nearest_ac(min(ty0 + PF2_TH - 1, oh - 1) + 2, nsy, h2 + 2) - 1;
    const int c_lo = nearest_ac(tx0, nsx, w2 + 2) - 1, c_hi = nearest_ac(min(tx0 + PF2_TW - 1, ow - 1) + 2, nsx, w2 + 2) - 1;
    const int wcols = c_hi - c_lo + 1, npx = (r_hi - r_lo + 1) * wcols;       // npx <= PF2_CAP: checked on the host
    const float *Tn = T + (size_t)n * h2 * w2 * 32;
    const float inv = 1.0f / (float)wcols;
    for (int u = threadIdx.x; u < npx * 9; u += 256) {
        const int px = u / 9, t = u - px * 9;
        const int wr = (int)(((float)px + 0.5f) * inv), wc = px - wr * wcols;
        const int sy = r_lo + wr, sx = c_lo + wc;
        f32x2 v = {0.f, 0.f};
        if ((unsigned)sy < (unsigned)h2 && (unsigned)sx < (unsigned)w2) v = *reinterpret_cast<const f32x2 *>(Tn + ((size_t)sy * w2 + sx) * 32 + t * 2);
        *reinterpret_cast<f32x2 *>(tab + px * 18 + t * 2) = v;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float b0 = bias2[0], b1 = bias2[1];
#pragma unroll
    for (int j = 0; j < PF2_PPT; ++j) {
        const int y = ty0 + j * 4 + (lane >> 4), x = tx0 + wave * 16 + (lane & 15);
        if (y >= oh || x >= ow) continue;
        int ry[3], rx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ry[d] = (nearest_ac(y + d, nsy, h2 + 2) - 1 - r_lo) * wcols;
            rx[d] = nearest_ac(x + d, nsx, w2 + 2) - 1 - c_lo;
        }
        float a0 = b0, a1 = b1;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const f32x2 t = *reinterpret_cast<const f32x2 *>(tab + (ry[dy] + rx[dx]) * 18 + (dy * 3 + dx) * 2);
                a0 += t.x;
                a1 += t.y;
            }
        const f32x2 u = sample_flow_legacy(pf3, n, h3, w3, y, x, usy, usx);
#pragma unroll
        for (int r = 0; r < 8; ++r) { a0 += u.x; a1 += u.y; }
        f32x2 o; o.x = a0; o.y = a1;
        reinterpret_cast<f32x2 *>(pf2)[((size_t)n * oh + y) * ow + x] = o;
    }
}

static int nearest_ac_host(int i, float scale, int n_in)
{
    return std::min((int)roundf((float)i * scale), n_in - 1);
}

hipError_t launch_pf2(const float *T, int B, int h2, int w2, const float *bias2, const float *pf3, int h3, int w3,
                      float *pf2, int H, int W, hipStream_t stream)
{
    const long long total = (long long)B * (H - 2) * (W - 2);
    const float nsy = H > 1 ? (float)(h2 + 2 - 1) / (float)(H - 1) : 0.f;
    const float nsx = W > 1 ? (float)(w2 + 2 - 1) / (float)(W - 1) : 0.f;
    const float usy = (float)h3 / (float)(H - 2), usx = (float)w3 / (float)(W - 2);
    const int oh = H - 2, ow = W - 2;
    const int tiles_x = (ow + PF2_TW - 1) / PF2_TW, tiles_y = (oh + PF2_TH - 1) / PF2_TH;
    // largest source window of any tile (the device evaluates the same fp32 map): rows and columns separately
    int rows_max = 0, cols_max = 0;
    for (int t = 0; t < tiles_y; ++t)
        rows_max = std::max(rows_max, nearest_ac_host(std::min(t * PF2_TH + PF2_TH - 1, oh - 1) + 2, nsy, h2 + 2) - nearest_ac_host(t * PF2_TH, nsy, h2 + 2) + 1);
    for (int t = 0; t < tiles_x; ++t)
        cols_max = std::max(cols_max, nearest_ac_host(std::min(t * PF2_TW + PF2_TW - 1, ow - 1) + 2, nsx, w2 + 2) - nearest_ac_host(t * PF2_TW, nsx, w2 + 2) + 1);
    // compulsory traffic (K9 is LDS / L2-bound, reported against the HBM bound of these bytes): the 128-byte table rows
    // and the coarser flow read once, 8 bytes written per output pixel
    const double alg_bytes = 128.0 * B * h2 * w2 + 8.0 * B * h3 * w3 + 8.0 * total;
    if (rows_max * cols_max <= PF2_CAP && (long long)tiles_x * tiles_y * B < (1ll << 31))
        return launch_timed(HBM_SLOT_PF2, alg_bytes, pf2_tile_kernel, dim3((unsigned)(tiles_x * tiles_y * B)), dim3(256), stream, T, B, h2, w2,
                            bias2, pf3, h3, w3, pf2, H, W, nsy, nsx, usy, usx, tiles_x, tiles_y);
    return launch_timed(HBM_SLOT_PF2, alg_bytes, pf2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, T, B, h2, w2, bias2, pf3,
                        h3, w3, pf2, H, W, nsy, nsx, usy, usx);
}

// ---------------------------------------------------------------------------------
// main:497-498: out = resize_images(flow*net_h/h, [oh, ow]); x = x*ow/net_w; y = y*oh/net_h -- every `*` and `/` of the reference's
// expression is its own TF op, i.e. its own fp32 rounding (a*b/c parses as (a*b)/c), and is evaluated as such here.
// ---------------------------------------------------------------------------------
// source flow grid + the constants of main:497-498 as the reference's graph applies them, one TF op (one fp32 rounding) each:
//   predict_flow2*384.0/382   ->  (t * nh) / dh     nh = net_h, dh = the flow's own height
//   outflow[...,0:1]*out_w/512 -> (v * mx) / dx     mx = out_w, dx = net_w;   y: (v * my) / dy, my = out_h, dy = net_h
// The three divisors are constants of the launch, and an IEEE fp32 division by a run-time value costs ~11 instructions (scale,
// quarter-rate reciprocal, refinement, fix-up) ten times per pixel -- it made the stand-alone glue ALU-bound (0.67 -> 0.40 of
// 8 TB/s at 1080p).  With r = RN(1/d) from the host the same correctly rounded quotient takes five: q = x*r, then two
// residual corrections q += (x - d*q)*r with fused multiply-adds -- the tail of the hardware's own sequence (Markstein: the last
// correction of a quotient already within one ulp rounds correctly unless d's significand is all ones).  A quotient that does not
// come out as a normal number -- zero (the corrections would turn -0 into +0), a denormal (their residuals underflow), x = +-inf
// (they make NaN) -- takes the plain division, a branch that flows never take.  Checked bit for bit against `/` on the device for
// EVERY non-NaN fp32 x, signed zeros, denormal quotients and infinities included (vstab_selftest_div_const,
// tests/test_gpu_parity.py); divisors outside 1 <= d <= 2^24 or with an all-ones significand take the plain division throughout.
struct GlueParams { int h, w; float nh, dh, mx, dx, my, dy, ry, rx; float rdh, rdx, rdy; int fast; };
__device__ __forceinline__ float div_const(float x, float d, float r, bool fast)
{
    if (!fast) return x / d;
    float q = x * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, x), r, q);
    q = __builtin_fmaf(__builtin_fmaf(-d, q, x), r, q);
    if (__builtin_expect(!(fabsf(q) >= 1.17549435e-38f), 0)) q = x / d;      // zero, denormal, NaN (x = +-inf or NaN): the IEEE division itself
    return q;
}
__device__ __forceinline__ float glue_pre(float t, const GlueParams &G) { return div_const(t * G.nh, G.dh, G.rdh, G.fast); }
__device__ __forceinline__ float glue_post_x(float v, const GlueParams &G) { return div_const(v * G.mx, G.dx, G.rdx, G.fast); }
__device__ __forceinline__ float glue_post_y(float v, const GlueParams &G) { return div_const(v * G.my, G.dy, G.rdy, G.fast); }
__host__ inline bool div_const_ok(float d)
{
    uint32_t u;
    memcpy(&u, &d, 4);
    return d >= 1.0f && d <= 16777216.0f && (u & 0x7fffffu) != 0x7fffffu;
}
__host__ inline GlueParams glue_params(int h, int w, int oh, int ow, int net_h, int net_w)
{
    GlueParams G{h, w, (float)net_h, (float)h, (float)ow, (float)net_w, (float)oh, (float)net_h, (float)h / (float)oh, (float)w / (float)ow, 0.f, 0.f, 0.f, 0};
    G.rdh = (float)(1.0 / (double)G.dh); G.rdx = (float)(1.0 / (double)G.dx); G.rdy = (float)(1.0 / (double)G.dy);
    G.fast = div_const_ok(G.dh) && div_const_ok(G.dx) && div_const_ok(G.dy);
    return G;
}

// device self-test of div_const against the IEEE division: every non-NaN fp32 bit pattern x in [first, first + count) -- zero, denormal
// and infinite quotients included; counts the bit mismatches
__global__ __launch_bounds__(256) void div_const_selftest_kernel(float d, float r, unsigned first, unsigned long long count, unsigned long long *bad)
{
    unsigned long long n = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
        const float x = __builtin_bit_cast(float, (unsigned)(first + i));
        const float want = x / d;
        if (x != x) continue;                                                    // NaN numerators: both are NaN, payloads are not compared
        const float got = div_const(x, d, r, true);
        if (__builtin_bit_cast(unsigned, got) != __builtin_bit_cast(unsigned, want)) ++n;
    }
    if (n) atomicAdd(bad, n);
}
hipError_t launch_div_const_selftest(float d, unsigned first, unsigned long long count, unsigned long long *bad, hipStream_t stream)
{
    if (!div_const_ok(d)) return hipErrorInvalidValue;
    div_const_selftest_kernel<<<dim3(4096), dim3(256), 0, stream>>>(d, (float)(1.0 / (double)d), first, count, bad);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void flow_resize_scale_kernel(const float *__restrict__ flow, int B, float *__restrict__ out, int oh, int ow,
                                                                GlueParams G)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * oh * ow;
    if (idx >= total) return;
    const int h = G.h, w = G.w;
    const int n = (int)(idx / (oh * ow));
    const int rem = (int)(idx - (long long)n * oh * ow);
    const int oy = rem / ow, ox = rem - oy * ow;
    const Lerp Y = legacy_coord(oy, G.ry, h), X = legacy_coord(ox, G.rx, w);
    const f32x2 *b = reinterpret_cast<const f32x2 *>(flow) + (long long)n * h * w;
    f32x2 tl = b[Y.lo * w + X.lo], tr = b[Y.lo * w + X.hi], bl = b[Y.hi * w + X.lo], br = b[Y.hi * w + X.hi];
    f32x2 o;
    o.x = glue_post_x(lerp2(glue_pre(tl.x, G), glue_pre(tr.x, G), glue_pre(bl.x, G), glue_pre(br.x, G), X.t, Y.t), G);
    o.y = glue_post_y(lerp2(glue_pre(tl.y, G), glue_pre(tr.y, G), glue_pre(bl.y, G), glue_pre(br.y, G), X.t, Y.t), G);
    reinterpret_cast<f32x2 *>(out)[idx] = o;
}

// tiled forms (defined next to the warp's tile kernel below); they return hipErrorNotSupported when a shape is not theirs
static hipError_t launch_glue_tile(const float *flow, int B, float *out, int oh, int ow, const GlueParams &G, hipStream_t stream);
static hipError_t launch_resize3_tile(const float *x, int B, int h, int w, int Cs, int c_off, float *out, int oh, int ow, hipStream_t stream);

hipError_t launch_flow_resize_scale(const float *flow, int B, int h, int w, float *out, int oh, int ow, int net_h, int net_w, hipStream_t stream)
{
    const long long total = (long long)B * oh * ow;
    const GlueParams G = glue_params(h, w, oh, ow, net_h, net_w);
    {
        const hipError_t e = launch_glue_tile(flow, B, out, oh, ow, G, stream);
        if (e != hipErrorNotSupported) return e;
    }
    // same size: TF returns the tensor unchanged; scale 1.0 gives lo = i, t = 0 -> identical values
    return launch_timed(HBM_SLOT_GLUE, 8.0 * B * h * w + 8.0 * total, flow_resize_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                        stream, flow, B, out, oh, ow, G);
}

// generic NHWC legacy-bilinear resize (main:806; one thread per output element)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float *__restrict__ x, int B, int h, int w, int C,
                                                              float *__restrict__ out, int oh, int ow, float ry, float rx)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * oh * ow * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    const int n = (int)(pix / (oh * ow));
    const int rem = (int)(pix - (long long)n * oh * ow);
    const int oy = rem / ow, ox = rem - oy * ow;
    const Lerp Y = legacy_coord(oy, ry, h), X = legacy_coord(ox, rx, w);
    const float *b = x + (long long)n * h * w * C + c;
    const float tl = b[((long long)Y.lo * w + X.lo) * C], tr = b[((long long)Y.lo * w + X.hi) * C];
    const float bl = b[((long long)Y.hi * w + X.lo) * C], br = b[((long long)Y.hi * w + X.hi) * C];
    out[idx] = lerp2(tl, tr, bl, br, X.t, Y.t);
}

hipError_t launch_resize_bilinear(const float *x, int B, int h, int w, int C, float *out, int oh, int ow, hipStream_t stream)
{
    const long long total = (long long)B * oh * ow * C;
    if (h == oh && w == ow)
        return hipMemcpyAsync(out, x, sizeof(float) * total, hipMemcpyDeviceToDevice, stream);
    if (C == 3) {
        const hipError_t e = launch_resize3_tile(x, B, h, w, 3, 0, out, oh, ow, stream);
        if (e != hipErrorNotSupported) return e;
    }
    resize_bilinear_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(
        x, B, h, w, C, out, oh, ow, (float)h / (float)oh, (float)w / (float)ow);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// tf_warp (main:70-130): dense backward warp.  One thread per output pixel: one 8-byte
// flow read, four corner gathers, C stores.  Corners by float->int truncation toward
// zero, all four clipped into the image, weights from the CLIPPED corners (A.6).
// The saturating v_cvt_i32_f32 keeps NaN/inf flows inside the image (no fault).
// ---------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void warp_flow_kernel(const float *__restrict__ img, const float *__restrict__ flow,
                                                        float *__restrict__ out, int B, int H, int W, int Cdyn)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * W;
    if (idx >= total) return;
    const int n = (int)(idx / (H * W));
    const int rem = (int)(idx - (long long)n * H * W);
    const int yy = rem / W, xx = rem - yy * W;
    const f32x2 f = reinterpret_cast<const f32x2 *>(flow)[idx];
    const float x = (float)xx + f.x, y = (float)yy + f.y;
    // Clamp BEFORE the conversion: float->int of an out-of-range value and x0 + 1 at INT_MAX are
    // undefined in C++ (the optimiser folds the clips below around them and a NaN/inf flow then
    // indexes out of bounds).  Inside [-2, W] nothing changes; outside, both corners still end
    // up clipped to the same border pixel exactly as in the reference.
    int x0 = (int)fminf(fmaxf(x, -2.f), (float)W), y0 = (int)fminf(fmaxf(y, -2.f), (float)H);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
    y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
    const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
    const float wa = (x1f - x) * (y1f - y), wb = (x1f - x) * (y - y0f);
    const float wc = (x - x0f) * (y1f - y), wd = (x - x0f) * (y - y0f);
    const int Cc = C > 0 ? C : Cdyn;
    const float *base = img + (long long)n * H * W * Cc;
    const float *Ia = base + ((long long)y0 * W + x0) * Cc, *Ib = base + ((long long)y1 * W + x0) * Cc;
    const float *Ic = base + ((long long)y0 * W + x1) * Cc, *Id = base + ((long long)y1 * W + x1) * Cc;
    float *o = out + idx * Cc;
    for (int c = 0; c < Cc; ++c) o[c] = ((wa * Ia[c] + wb * Ib[c]) + wc * Ic[c]) + wd * Id[c];   // tf.add_n order
}

// ---------------------------------------------------------------------------------
// W1 for 3-channel frames, shaped for the memory system (32 B/px algorithmic: 12 gathered + 8 flow + 12 written), optionally
// with the flow glue G1 (main:497-498) fused in front so that the output-resolution flow is produced in registers, written
// once (the evaluator returns it) and never re-read -- main:497-514 is one graph in the reference (40 B/px with the flow
// written: 8 source flow + 8 output flow + 12 + 12).
// What bounds it (profiles/README.md, "r02 warp study"; PMC TCP_TOTAL_CACHE_ACCESSES): the L1 looks up one 64-byte piece per
// four lanes per cycle, and a gather of 12-byte pixels costs 1.75 lookups per four lanes even when they are neighbours -- the
// L1 tag pipe saturates before HBM does.  So the kernel minimises lookups per pixel:
//   * a workgroup owns a 16 x 32 tile of ONE sample's output pixels and every wave instruction works on a 4 x 16 patch, so the
//     lines a gather instruction touches are a compact 2-D footprint ((4 + spread) rows x (16 + spread) pixels) whether or not
//     the flow is smooth (a 1-D run of 64 pixels touches 64 x `spread` lines once neighbouring flows differ: 0.37 -> 0.55 of
//     8 TB/s on the benchmark's flows, which move neighbouring pixels' sample points 0.6 px apart per pixel);
//   * one 3-dword load per corner (never three scalar ones), both pixels of a thread's loads issued back to back;
//   * results leave through LDS as 16-byte stores of whole 384-byte tile rows (W % 4 == 0; otherwise 12-byte stores);
//   * workgroups are numbered through the XCD map: one XCD's L2 sees a contiguous band of tile rows;
//   * no 64-bit divisions.
// The arithmetic is the sequence of warp_flow_kernel / flow_resize_scale_kernel above, statement for statement: results are
// bit-identical to the two-launch path (tests: test_warp_tiled_kernel_bit_exact_vs_fp32_oracle, test_fused_glue_warp_bit_identical).
// Rejected after measurement (tools/warp_variants.inc): 1-D 1024-pixel tiles with LDS-staged stores, an LDS-staged source
// window (bounding box of the tile's corners, coalesced fill, gathers from LDS) and its persistent, flow-prefetching form.
// ---------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) rgb3 { float r, g, b; };
struct __attribute__((packed, aligned(8))) flow2 { f32x2 a, b; };     // two neighbouring flow pixels (8-byte aligned, 16 bytes)

// a workgroup owns a TH x TW tile of ONE sample's output pixels; a wave instruction works on a WH x WW patch (WH * WW = 64)
template <bool FUSED, bool WRITE_FLOW, int WH, int WW, int TW, int PPT, bool REMAP = true, bool NT = false, bool STAGE = false>
__global__ __launch_bounds__(256) void warp3_tile_kernel(const float *__restrict__ img, const float *__restrict__ flow,
                                                         float *__restrict__ out, float *__restrict__ outflow, int B,
                                                         int H, int W, int tiles_x, int tiles_y, GlueParams G)
{
    static_assert(WH * WW == 64 && TW % WW == 0 && (4 * PPT) % (TW / WW) == 0, "patch / tile shapes");
    constexpr int PPR = TW / WW;                    // patches per tile row
    constexpr int TH = WH * (4 * PPT) / PPR;
    __shared__ __attribute__((aligned(16))) float stage[STAGE ? TH * TW * 3 : 4];
    unsigned bx = blockIdx.x, by, bz;
    if (REMAP) xcd_remap_calc(gridDim.x, 1, 1, blockIdx.x, bx, by, bz);
    const int tpi = tiles_x * tiles_y;
    const int n = (int)bx / tpi, trem = (int)bx - n * tpi;
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * TW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long HW = (long long)H * W;

    int yy[PPT], xx[PPT];
    bool ok[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int q = j * 4 + wave;
        yy[j] = ty0 + (q / PPR) * WH + lane / WW;
        xx[j] = tx0 + (q % PPR) * WW + lane % WW;
        ok[j] = yy[j] < H && xx[j] < W;
        if (!ok[j]) { yy[j] = 0; xx[j] = 0; }
    }
    f32x2 f[PPT];
    if (FUSED) {
        f32x2 tl[PPT], tr[PPT], bl[PPT], br[PPT];
        Lerp Y[PPT], X[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            Y[j] = legacy_coord(yy[j], G.ry, G.h); X[j] = legacy_coord(xx[j], G.rx, G.w);
            const f32x2 *b = reinterpret_cast<const f32x2 *>(flow) + (long long)n * G.h * G.w;
            // the left/right taps of a row are neighbours (hi = lo + 1, or both the last column): ONE 16-byte load per row
            // -- half the L1 lookups of four 8-byte gathers -- of the source pixels (xb, xb + 1), xb = min(lo, w - 2) (w >= 2: host)
            const int xb = min(X[j].lo, G.w - 2);
            const flow2 top = *reinterpret_cast<const flow2 *>(b + Y[j].lo * G.w + xb), bot = *reinterpret_cast<const flow2 *>(b + Y[j].hi * G.w + xb);
            const bool l1 = X[j].lo != xb, h1 = X[j].hi != xb;
            tl[j] = l1 ? top.b : top.a; tr[j] = h1 ? top.b : top.a;
            bl[j] = l1 ? bot.b : bot.a; br[j] = h1 ? bot.b : bot.a;
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            f[j].x = glue_post_x(lerp2(glue_pre(tl[j].x, G), glue_pre(tr[j].x, G), glue_pre(bl[j].x, G), glue_pre(br[j].x, G), X[j].t, Y[j].t), G);
            f[j].y = glue_post_y(lerp2(glue_pre(tl[j].y, G), glue_pre(tr[j].y, G), glue_pre(bl[j].y, G), glue_pre(br[j].y, G), X[j].t, Y[j].t), G);
            if (WRITE_FLOW && ok[j]) reinterpret_cast<f32x2 *>(outflow)[n * HW + (long long)yy[j] * W + xx[j]] = f[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const f32x2 *fp = reinterpret_cast<const f32x2 *>(flow) + n * HW + (long long)yy[j] * W + xx[j];
            f[j] = NT ? __builtin_nontemporal_load(fp) : *fp;
        }
    }
    float wa[PPT], wb[PPT], wc[PPT], wd[PPT];
    rgb3 Ia[PPT], Ib[PPT], Ic[PPT], Id[PPT];
    const rgb3 *b = reinterpret_cast<const rgb3 *>(img) + n * HW;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const float x = (float)xx[j] + f[j].x, y = (float)yy[j] + f[j].y;
        int x0 = (int)fminf(fmaxf(x, -2.f), (float)W), y0 = (int)fminf(fmaxf(y, -2.f), (float)H);
        int x1 = x0 + 1, y1 = y0 + 1;
        x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
        y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
        const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
        wa[j] = (x1f - x) * (y1f - y); wb[j] = (x1f - x) * (y - y0f);
        wc[j] = (x - x0f) * (y1f - y); wd[j] = (x - x0f) * (y - y0f);
        Ia[j] = b[y0 * W + x0]; Ib[j] = b[y1 * W + x0]; Ic[j] = b[y0 * W + x1]; Id[j] = b[y1 * W + x1];
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        rgb3 r;
        r.r = ((wa[j] * Ia[j].r + wb[j] * Ib[j].r) + wc[j] * Ic[j].r) + wd[j] * Id[j].r;      // tf.add_n order
        r.g = ((wa[j] * Ia[j].g + wb[j] * Ib[j].g) + wc[j] * Ic[j].g) + wd[j] * Id[j].g;
        r.b = ((wa[j] * Ia[j].b + wb[j] * Ib[j].b) + wc[j] * Ic[j].b) + wd[j] * Id[j].b;
        if (STAGE) {
            const int q = j * 4 + wave;
            *reinterpret_cast<rgb3 *>(stage + (((q / PPR) * WH + lane / WW) * TW + (q % PPR) * WW + lane % WW) * 3) = r;
        } else if (ok[j]) {
            float *o = out + (n * HW + (long long)yy[j] * W + xx[j]) * 3;
            if (NT) { __builtin_nontemporal_store(r.r, o); __builtin_nontemporal_store(r.g, o + 1); __builtin_nontemporal_store(r.b, o + 2); }
            else *reinterpret_cast<rgb3 *>(o) = r;
        }
    }
    if (STAGE) {       // W % 4 == 0 (host): a tile row is TW*12 bytes from a 16-byte aligned address; 16-byte stores, dwords at a ragged right edge
        __syncthreads();
        constexpr int R4 = TW * 3 / 4;
        const int vw3 = min(TW, W - tx0) * 3;
        for (int e = threadIdx.x; e < TH * R4; e += 256) {
            const int row = e / R4, c4 = e - row * R4;
            if (ty0 + row >= H || c4 * 4 >= vw3) continue;
            float *o = out + (n * HW + (long long)(ty0 + row) * W + tx0) * 3 + c4 * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + row * TW * 3 + c4 * 4);
            if (c4 * 4 + 4 <= vw3) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(o)); else *reinterpret_cast<f32x4 *>(o) = v; }
            else for (int i = 0; c4 * 4 + i < vw3; ++i) o[i] = v[i];
        }
    }
}

constexpr int WT_WH = 4, WT_WW = 16, WT_TW = 32, WT_PPT = 2, WT_TH = WT_WH * (4 * WT_PPT) / (WT_TW / WT_WW);      // 16 x 32 tile

static bool warp3_ok(const void *img, const void *out, const void *outflow, int B, int H, int W, int C)
{
    return C == 3 && (long long)B * H * W < (1ll << 31) && (((uintptr_t)img | (uintptr_t)out | (uintptr_t)outflow) & 15) == 0;
}

template <bool FUSED, bool WRITE_FLOW>
static hipError_t launch_warp3(int slot, double alg_bytes, const float *img, const float *flow, float *out, float *outflow, int B, int H,
                               int W, const GlueParams &G, hipStream_t stream)
{
    const int tx = (W + WT_TW - 1) / WT_TW, ty = (H + WT_TH - 1) / WT_TH;
    const dim3 grid((unsigned)((long long)tx * ty * B)), block(256);
    if ((W & 3) == 0)
        return launch_timed(slot, alg_bytes, warp3_tile_kernel<FUSED, WRITE_FLOW, WT_WH, WT_WW, WT_TW, WT_PPT, true, false, true>, grid, block,
                            stream, img, flow, out, outflow, B, H, W, tx, ty, G);
    return launch_timed(slot, alg_bytes, warp3_tile_kernel<FUSED, WRITE_FLOW, WT_WH, WT_WW, WT_TW, WT_PPT, true, false, false>, grid, block,
                        stream, img, flow, out, outflow, B, H, W, tx, ty, G);
}

hipError_t launch_warp_flow(const float *img, const float *flow, float *out, int B, int H, int W, int C, hipStream_t stream)
{
    const long long total = (long long)B * H * W;
    if (total == 0) return hipSuccess;
    if (warp3_ok(img, out, nullptr, B, H, W, C) && (long long)((W + WT_TW - 1) / WT_TW) * ((H + WT_TH - 1) / WT_TH) * B < (1ll << 31))
        return launch_warp3<false, false>(HBM_SLOT_WARP, 32.0 * total, img, flow, out, nullptr, B, H, W, GlueParams{}, stream);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    const double bytes = (8.0 + 8.0 * C) * total;
    if (C == 3) return launch_timed(HBM_SLOT_WARP, bytes, warp_flow_kernel<3>, grid, block, stream, img, flow, out, B, H, W, C);
    if (C == 1) return launch_timed(HBM_SLOT_WARP, bytes, warp_flow_kernel<1>, grid, block, stream, img, flow, out, B, H, W, C);
    return launch_timed(HBM_SLOT_WARP, bytes, warp_flow_kernel<0>, grid, block, stream, img, flow, out, B, H, W, C);
}

// main:497-514 as one launch: outflow = glue(flow [B,h,w,2]) at [B,oh,ow,2] (written if `outflow` is given), warped =
// tf_warp(img [B,oh,ow,3], outflow).  C must be 3 (callers fall back to the two kernels otherwise).
hipError_t launch_flow_glue_warp(const float *flow, int B, int h, int w, const float *img, float *outflow, float *out, int oh, int ow,
                                 int C, int net_h, int net_w, hipStream_t stream)
{
    const long long total = (long long)B * oh * ow;
    if (total == 0) return hipSuccess;
    if (!warp3_ok(img, out, outflow, B, oh, ow, C) || w < 2) return hipErrorInvalidValue;
    const GlueParams G = glue_params(h, w, oh, ow, net_h, net_w);
    const double src = 8.0 * B * h * w;                       // the source flow is read once from HBM (its 4 taps per pixel hit in cache)
    if (outflow) return launch_warp3<true, true>(HBM_SLOT_GLUE_WARP, src + 32.0 * total, img, flow, out, outflow, B, oh, ow, G, stream);
    return launch_warp3<true, false>(HBM_SLOT_GLUE_WARP, src + 24.0 * total, img, flow, out, nullptr, B, oh, ow, G, stream);
}

// ---------------------------------------------------------------------------------
// The tail of evaluate_originalSize's graph in ONE launch (round 5): predict_flow2's gather (model.py:882-887: nine taps of the tap
// table + eight adds of the upsampled predict_flow3, pf2_tile_kernel above), the glue of main:497-498 and tf_warp (main:514,
// warp3_tile_kernel<FUSED>).  A workgroup owns the same 16 x 32 tile of OUTPUT pixels the warp kernel does.  The glue's bilinear taps
// of that tile reach a small rectangle of predict_flow2 (<= 18 x 34 pixels when the output is no smaller than the flow grid); the
// workgroup computes exactly that rectangle -- statement for statement pf2_tile_kernel's sums, out of a tap-table window in LDS -- keeps
// it in LDS, writes the part it OWNS to the returned predict_flow2 tensor (rows [lo(ty0), lo(ty0 + TH)), the last tile to the end:
// neighbouring tiles' rectangles overlap by a pixel or two, every pixel is written once), and glue + warp read their four flow taps
// from LDS.  predict_flow2 is written (it is a returned tensor) but never re-read: 33 MB less traffic at B=8 512x512, one launch
// boundary less, and the warp's L1 tag pipe -- its limit -- loses the two flow lookups per pixel.  Same fp32 operations in the same
// order as the two launches: bit-identical (tests/test_gpu_parity.py::test_fused_tail_bit_identical).
// ---------------------------------------------------------------------------------
constexpr int FT_PF_CAP = 640, FT_T_CAP = 96;       // predict_flow2 window pixels (18 x 34 = 612) and tap-table window pixels (7 x 11 = 77) held in LDS
struct TailParams {
    int h2, w2, h3, w3, H, W;                       // tap table grid, predict_flow3 grid, network input size (predict_flow2 is (H-2) x (W-2))
    float nsy, nsx, usy, usx;
};
// U8 = the clip driver's frame path (warp3_u8_tile_kernel's statements: BGR bytes -> swap / 255 through the 256-entry table, the warp,
// truncating quantiser, bytes out); otherwise fp32 frames in and out (warp3_tile_kernel's)
template <bool WRITE_FLOW, bool STAGE, bool U8>
__global__ __launch_bounds__(256) void pf2_glue_warp_kernel(const float *__restrict__ T, const float *__restrict__ bias2, const float *__restrict__ pf3,
                                                            float *__restrict__ pf2, const void *__restrict__ img_, void *__restrict__ out_,
                                                            float *__restrict__ outflow, int B, int OH, int OW, int tiles_x, int tiles_y,
                                                            TailParams P, GlueParams G)
{
    const float *__restrict__ img = reinterpret_cast<const float *>(img_);
    float *__restrict__ out = reinterpret_cast<float *>(out_);
    constexpr int TH = WT_TH, TW = WT_TW, WH = WT_WH, WW = WT_WW, PPT = WT_PPT, PPR = TW / WW;
    // the tap-table window is dead once predict_flow2's rectangle is in `pfw` (second barrier): the output staging tile takes its place
    // (12 KB of LDS per workgroup instead of 23: the wave limit of the register file, not LDS, bounds the occupancy)
    constexpr int TAB_FLOATS = FT_T_CAP * 18, STAGE_FLOATS = STAGE ? TH * TW * 3 : 4;
    __shared__ __attribute__((aligned(16))) float tab[TAB_FLOATS > STAGE_FLOATS ? TAB_FLOATS : STAGE_FLOATS];
    __shared__ __attribute__((aligned(8))) f32x2 pfw[FT_PF_CAP];
    float *stage = tab;
    __shared__ float lut[U8 ? 256 : 1];
    if (U8) lut[threadIdx.x] = (float)threadIdx.x / 255.0f;          // 256 threads; read after the barriers below
    unsigned bx, by, bz;
    xcd_remap_calc(gridDim.x, 1, 1, blockIdx.x, bx, by, bz);
    const int tpi = tiles_x * tiles_y;
    const int n = (int)bx / tpi, trem = (int)bx - n * tpi;
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int h = G.h, w = G.w;                       // predict_flow2's grid
    // ---- the predict_flow2 rectangle: what the tile's glue taps reach, extended to what the tile owns
    const int yl = min(ty0 + TH - 1, OH - 1), xl = min(tx0 + TW - 1, OW - 1);
    const int wy0 = legacy_coord(ty0, G.ry, h).lo, wx0 = legacy_coord(tx0, G.rx, w).lo;
    const int own_y1 = tyi == tiles_y - 1 ? h : legacy_coord(ty0 + TH, G.ry, h).lo;
    const int own_x1 = txi == tiles_x - 1 ? w : legacy_coord(tx0 + TW, G.rx, w).lo;
    const int wy1 = max(legacy_coord(yl, G.ry, h).hi, own_y1 - 1), wx1 = max(legacy_coord(xl, G.rx, w).hi, own_x1 - 1);
    const int WC = wx1 - wx0 + 1, npf = (wy1 - wy0 + 1) * WC;          // <= FT_PF_CAP: checked on the host
    // ---- its tap-table window (indices into the UNPADDED concat2 grid: -1 and h2 / w2 are the zero ring), pf2_tile_kernel's way
    const int r_lo = nearest_ac(wy0, P.nsy, P.h2 + 2) - 1, r_hi = nearest_ac(wy1 + 2, P.nsy, P.h2 + 2) - 1;
    const int c_lo = nearest_ac(wx0, P.nsx, P.w2 + 2) - 1, c_hi = nearest_ac(wx1 + 2, P.nsx, P.w2 + 2) - 1;
    const int wcols = c_hi - c_lo + 1, npx = (r_hi - r_lo + 1) * wcols;  // <= FT_T_CAP: checked on the host
    const float *Tn = T + (size_t)n * P.h2 * P.w2 * 32;
    {
        const float inv = 1.0f / (float)wcols;
        for (int u = threadIdx.x; u < npx * 9; u += 256) {
            const int px = u / 9, t = u - px * 9;
            const int wr = (int)(((float)px + 0.5f) * inv), wc = px - wr * wcols;
            const int sy = r_lo + wr, sx = c_lo + wc;
            f32x2 v = {0.f, 0.f};
            if ((unsigned)sy < (unsigned)P.h2 && (unsigned)sx < (unsigned)P.w2) v = *reinterpret_cast<const f32x2 *>(Tn + ((size_t)sy * P.w2 + sx) * 32 + t * 2);
            *reinterpret_cast<f32x2 *>(tab + px * 18 + t * 2) = v;
        }
    }
    __syncthreads();
    // ---- predict_flow2 over the rectangle (pf2_tile_kernel's statements), kept in LDS; the owned part leaves for HBM
    {
        const float b0 = bias2[0], b1 = bias2[1];
        const float invc = 1.0f / (float)WC;
        for (int u = threadIdx.x; u < npf; u += 256) {
            const int wr = (int)(((float)u + 0.5f) * invc), wc = u - wr * WC;
            const int y = wy0 + wr, x = wx0 + wc;
            int ry[3], rx[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                ry[d] = (nearest_ac(y + d, P.nsy, P.h2 + 2) - 1 - r_lo) * wcols;
                rx[d] = nearest_ac(x + d, P.nsx, P.w2 + 2) - 1 - c_lo;
            }
            float a0 = b0, a1 = b1;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x2 t = *reinterpret_cast<const f32x2 *>(tab + (ry[dy] + rx[dx]) * 18 + (dy * 3 + dx) * 2);
                    a0 += t.x;
                    a1 += t.y;
                }
            const f32x2 uu = sample_flow_legacy(pf3, n, P.h3, P.w3, y, x, P.usy, P.usx);
#pragma unroll
            for (int r = 0; r < 8; ++r) { a0 += uu.x; a1 += uu.y; }
            f32x2 o; o.x = a0; o.y = a1;
            pfw[u] = o;
            if (y < own_y1 && x < own_x1) reinterpret_cast<f32x2 *>(pf2)[((size_t)n * h + y) * w + x] = o;
        }
    }
    __syncthreads();
    // ---- glue + warp of the tile's output pixels (warp3_tile_kernel<FUSED>'s statements; the four flow taps come from LDS)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long HW = (long long)OH * OW;
    int yy[PPT], xx[PPT];
    bool ok[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int q = j * 4 + wave;
        yy[j] = ty0 + (q / PPR) * WH + lane / WW;
        xx[j] = tx0 + (q % PPR) * WW + lane % WW;
        ok[j] = yy[j] < OH && xx[j] < OW;
        if (!ok[j]) { yy[j] = ty0; xx[j] = tx0; }              // a pixel of this tile (its taps are inside the LDS rectangle); never stored
    }
    f32x2 f[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const Lerp Y = legacy_coord(yy[j], G.ry, h), X = legacy_coord(xx[j], G.rx, w);
        const f32x2 tl = pfw[(Y.lo - wy0) * WC + (X.lo - wx0)], tr = pfw[(Y.lo - wy0) * WC + (X.hi - wx0)];
        const f32x2 bl = pfw[(Y.hi - wy0) * WC + (X.lo - wx0)], br = pfw[(Y.hi - wy0) * WC + (X.hi - wx0)];
        f[j].x = glue_post_x(lerp2(glue_pre(tl.x, G), glue_pre(tr.x, G), glue_pre(bl.x, G), glue_pre(br.x, G), X.t, Y.t), G);
        f[j].y = glue_post_y(lerp2(glue_pre(tl.y, G), glue_pre(tr.y, G), glue_pre(bl.y, G), glue_pre(br.y, G), X.t, Y.t), G);
        if (WRITE_FLOW && ok[j]) reinterpret_cast<f32x2 *>(outflow)[n * HW + (long long)yy[j] * OW + xx[j]] = f[j];
    }
    float wa[PPT], wb[PPT], wc_[PPT], wd[PPT];
    rgb3 Ia[PPT], Ib[PPT], Ic[PPT], Id[PPT];
    const rgb3 *b = reinterpret_cast<const rgb3 *>(img) + n * HW;
    const unsigned char *b8 = reinterpret_cast<const unsigned char *>(img_) + n * HW * 3;
    auto px = [&](int y, int x) {                                     // fp32 frame, or swap(frame) / 255: channels 2, 1, 0 of the BGR pixel
        if constexpr (U8) {
            const unsigned char *q = b8 + ((long long)y * OW + x) * 3;
            rgb3 r; r.r = lut[q[2]]; r.g = lut[q[1]]; r.b = lut[q[0]];
            return r;
        } else {
            return b[y * OW + x];
        }
    };
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const float x = (float)xx[j] + f[j].x, y = (float)yy[j] + f[j].y;
        int x0 = (int)fminf(fmaxf(x, -2.f), (float)OW), y0 = (int)fminf(fmaxf(y, -2.f), (float)OH);
        int x1 = x0 + 1, y1 = y0 + 1;
        x0 = min(max(x0, 0), OW - 1); x1 = min(max(x1, 0), OW - 1);
        y0 = min(max(y0, 0), OH - 1); y1 = min(max(y1, 0), OH - 1);
        const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
        wa[j] = (x1f - x) * (y1f - y); wb[j] = (x1f - x) * (y - y0f);
        wc_[j] = (x - x0f) * (y1f - y); wd[j] = (x - x0f) * (y - y0f);
        Ia[j] = px(y0, x0); Ib[j] = px(y1, x0); Ic[j] = px(y0, x1); Id[j] = px(y1, x1);
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        rgb3 r;
        r.r = ((wa[j] * Ia[j].r + wb[j] * Ib[j].r) + wc_[j] * Ic[j].r) + wd[j] * Id[j].r;      // tf.add_n order
        r.g = ((wa[j] * Ia[j].g + wb[j] * Ib[j].g) + wc_[j] * Ic[j].g) + wd[j] * Id[j].g;
        r.b = ((wa[j] * Ia[j].b + wb[j] * Ib[j].b) + wc_[j] * Ic[j].b) + wd[j] * Id[j].b;
        const int q = j * 4 + wave;
        const int spix = ((q / PPR) * WH + lane / WW) * TW + (q % PPR) * WW + lane % WW;
        if constexpr (U8) {                           // quantise_output_kernel's: truncate, saturate, channels swapped back
            unsigned char *o = reinterpret_cast<unsigned char *>(stage) + spix * 3;
            const float rr[3] = {r.r, r.g, r.b};
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = (unsigned char)fminf(fmaxf(truncf(rr[2 - c] * 255.0f), 0.f), 255.f);
        } else if (STAGE) {
            *reinterpret_cast<rgb3 *>(stage + spix * 3) = r;
        } else if (ok[j]) {
            *reinterpret_cast<rgb3 *>(out + (n * HW + (long long)yy[j] * OW + xx[j]) * 3) = r;
        }
    }
    if constexpr (U8) {
        __syncthreads();
        const unsigned char *sb = reinterpret_cast<const unsigned char *>(stage);
        unsigned char *out8 = reinterpret_cast<unsigned char *>(out_);
        // a tile row is TW*3 = 96 bytes; 4-byte stores where rows start on a 4-byte boundary (OW % 4 == 0), bytes otherwise / at a ragged edge
        const int vw3 = min(TW, OW - tx0) * 3;
        if ((OW & 3) == 0) {
            constexpr int R4 = TW * 3 / 4;
            for (int e = threadIdx.x; e < TH * R4; e += 256) {
                const int row = e / R4, c4 = e - row * R4;
                if (ty0 + row >= OH || c4 * 4 >= vw3) continue;
                unsigned char *o = out8 + (n * HW + (long long)(ty0 + row) * OW + tx0) * 3 + c4 * 4;
                if (c4 * 4 + 4 <= vw3) *reinterpret_cast<unsigned *>(o) = *reinterpret_cast<const unsigned *>(sb + row * TW * 3 + c4 * 4);
                else for (int i = 0; c4 * 4 + i < vw3; ++i) o[i] = sb[row * TW * 3 + c4 * 4 + i];
            }
        } else {
            for (int e = threadIdx.x; e < TH * TW * 3; e += 256) {
                const int row = e / (TW * 3), c = e - row * (TW * 3);
                if (ty0 + row < OH && c < vw3) out8[(n * HW + (long long)(ty0 + row) * OW + tx0) * 3 + c] = sb[e];
            }
        }
    } else if (STAGE) {       // OW % 4 == 0 (host): a tile row is TW*12 bytes from a 16-byte aligned address
        __syncthreads();
        constexpr int R4 = TW * 3 / 4;
        const int vw3 = min(TW, OW - tx0) * 3;
        for (int e = threadIdx.x; e < TH * R4; e += 256) {
            const int row = e / R4, c4 = e - row * R4;
            if (ty0 + row >= OH || c4 * 4 >= vw3) continue;
            float *o = out + (n * HW + (long long)(ty0 + row) * OW + tx0) * 3 + c4 * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + row * TW * 3 + c4 * 4);
            if (c4 * 4 + 4 <= vw3) *reinterpret_cast<f32x4 *>(o) = v;
            else for (int i = 0; c4 * 4 + i < vw3; ++i) o[i] = v[i];
        }
    }
}

static Lerp legacy_coord_host(int o, float scale, int n_in)
{
    const float f = (float)o * scale;
    const float fl = floorf(f);
    Lerp L;
    L.lo = std::min((int)fl, n_in - 1);
    L.hi = std::min(L.lo + 1, n_in - 1);
    L.t = f - fl;
    return L;
}

// predict_flow2 gather + glue + warp as one launch; hipErrorNotSupported when a tile's rectangles would not fit the LDS windows (an output
// much smaller than the flow grid) or the buffers miss the warp kernel's alignment: the caller then runs the two launches
static hipError_t launch_tail(const float *T, int B, int h2, int w2, const float *bias2, const float *pf3, int h3, int w3, float *pf2, int H, int W,
                              const void *img, float *outflow, void *out, int oh, int ow, bool u8, hipStream_t stream)
{
    const int h = H - 2, w = W - 2;
    if (B < 1 || h < 1 || w < 2 || oh < 1 || ow < 1) return hipErrorNotSupported;
    if ((uintptr_t)pf2 & 7) return hipErrorNotSupported;
    if (u8) {
        if ((long long)B * oh * ow >= (1ll << 31) / 3 || ((uintptr_t)outflow & 7) || ((uintptr_t)out & 3)) return hipErrorNotSupported;
    } else if (!warp3_ok(img, out, outflow, B, oh, ow, 3)) return hipErrorNotSupported;
    const GlueParams G = glue_params(h, w, oh, ow, H, W);
    TailParams P{h2, w2, h3, w3, H, W, H > 1 ? (float)(h2 + 2 - 1) / (float)(H - 1) : 0.f, W > 1 ? (float)(w2 + 2 - 1) / (float)(W - 1) : 0.f,
                 (float)h3 / (float)h, (float)w3 / (float)w};
    const int tx = (ow + WT_TW - 1) / WT_TW, ty = (oh + WT_TH - 1) / WT_TH;
    if ((long long)tx * ty * B >= (1ll << 31)) return hipErrorNotSupported;
    // the largest rectangles of any tile, rows and columns separately (the device evaluates the same fp32 maps); the tiles' owned ranges
    // must tile [0, h) x [0, w) without gaps: lo(0) = 0 and the ranges are consecutive by construction
    auto span = [&](int tiles, int T_, int OUT, float r, int nflow, float ns, int ntab, int &pf_max, int &t_max) {
        for (int t = 0; t < tiles; ++t) {
            const int o0 = t * T_, ol = std::min(o0 + T_ - 1, OUT - 1);
            const int w0 = legacy_coord_host(o0, r, nflow).lo;
            const int own1 = t == tiles - 1 ? nflow : legacy_coord_host(o0 + T_, r, nflow).lo;
            const int w1 = std::max(legacy_coord_host(ol, r, nflow).hi, own1 - 1);
            pf_max = std::max(pf_max, w1 - w0 + 1);
            t_max = std::max(t_max, nearest_ac_host(w1 + 2, ns, ntab + 2) - nearest_ac_host(w0, ns, ntab + 2) + 1);
        }
    };
    if (legacy_coord_host(0, G.ry, h).lo != 0 || legacy_coord_host(0, G.rx, w).lo != 0) return hipErrorNotSupported;
    int pr = 0, pc = 0, tr = 0, tc = 0;
    span(ty, WT_TH, oh, G.ry, h, P.nsy, h2, pr, tr);
    span(tx, WT_TW, ow, G.rx, w, P.nsx, w2, pc, tc);
    if (pr * pc > FT_PF_CAP || tr * tc > FT_T_CAP) return hipErrorNotSupported;
    const long long total = (long long)B * oh * ow;
    // compulsory traffic: tap-table rows, the coarser flow, predict_flow2 WRITTEN once (never re-read), frame read, warped (and the
    // output-resolution flow) written
    const double px_bytes = u8 ? (outflow ? 14.0 : 6.0) : (outflow ? 32.0 : 24.0);
    const double alg_bytes = 128.0 * B * h2 * w2 + 8.0 * B * h3 * w3 + 8.0 * B * h * w + px_bytes * total;
    const dim3 grid((unsigned)((long long)tx * ty * B)), block(256);
#define VSTAB_TAIL(WF, ST, U) launch_timed(HBM_SLOT_TAIL, alg_bytes, pf2_glue_warp_kernel<WF, ST, U>, grid, block, stream, T, bias2, pf3, pf2, img, out, outflow, B, oh, ow, tx, ty, P, G)
    if (u8) return outflow ? VSTAB_TAIL(true, true, true) : VSTAB_TAIL(false, true, true);
    if (outflow) return (ow & 3) == 0 ? VSTAB_TAIL(true, true, false) : VSTAB_TAIL(true, false, false);
    return (ow & 3) == 0 ? VSTAB_TAIL(false, true, false) : VSTAB_TAIL(false, false, false);
#undef VSTAB_TAIL
}

hipError_t launch_pf2_glue_warp(const float *T, int B, int h2, int w2, const float *bias2, const float *pf3, int h3, int w3, float *pf2, int H, int W,
                                const float *img, float *outflow, float *out, int oh, int ow, hipStream_t stream)
{
    return launch_tail(T, B, h2, w2, bias2, pf3, h3, w3, pf2, H, W, img, outflow, out, oh, ow, false, stream);
}

// the same on the clip driver's 8-bit frames: identical bytes to launch_pf2 + launch_flow_glue_warp_u8
hipError_t launch_pf2_glue_warp_u8(const float *T, int B, int h2, int w2, const float *bias2, const float *pf3, int h3, int w3, float *pf2, int H, int W,
                                   const unsigned char *frame, float *outflow, unsigned char *out, int oh, int ow, hipStream_t stream)
{
    return launch_tail(T, B, h2, w2, bias2, pf3, h3, w3, pf2, H, W, frame, outflow, out, oh, ow, true, stream);
}

// ---------------------------------------------------------------------------------
// The evaluator's frame path in ONE launch, 8-bit in and out (main:568, 497-514, 625/630): frame_f = swap(frame)/255, outflow =
// glue(flow), warped = tf_warp(frame_f, outflow), out = uint8(swap(warped*255)) -- without frame_f and warped ever existing in HBM
// (the clip driver ran four launches here: frame_to_float, the fused glue + warp, quantise_output; 2 x 12 B/px of fp32 traffic more).
// Statement for statement those kernels' arithmetic: u8 -> float through a 256-entry table of the IEEE quotients i / 255.0f (what
// frame_to_float_kernel divides per pixel), the warp as warp3_tile_kernel, the quantiser as quantise_output_kernel (truncation,
// saturating where numpy is undefined, channels swapped back): identical bytes (tests/test_gpu_clip.py).
// ---------------------------------------------------------------------------------
template <bool WRITE_FLOW>
__global__ __launch_bounds__(256) void warp3_u8_tile_kernel(const unsigned char *__restrict__ img, const float *__restrict__ flow,
                                                            unsigned char *__restrict__ out, float *__restrict__ outflow, int B, int H, int W,
                                                            int tiles_x, int tiles_y, GlueParams G)
{
    constexpr int WH = WT_WH, WW = WT_WW, TW = WT_TW, PPT = WT_PPT, PPR = TW / WW, TH = WT_TH;
    __shared__ float lut[256];
    __shared__ __attribute__((aligned(16))) unsigned char stage[TH * TW * 3];
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;                  // 256 threads
    unsigned bx, by, bz;
    xcd_remap_calc(gridDim.x, 1, 1, blockIdx.x, bx, by, bz);
    const int tpi = tiles_x * tiles_y;
    const int n = (int)bx / tpi, trem = (int)bx - n * tpi;
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * TW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long HW = (long long)H * W;
    int yy[PPT], xx[PPT];
    bool ok[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int q = j * 4 + wave;
        yy[j] = ty0 + (q / PPR) * WH + lane / WW;
        xx[j] = tx0 + (q % PPR) * WW + lane % WW;
        ok[j] = yy[j] < H && xx[j] < W;
        if (!ok[j]) { yy[j] = 0; xx[j] = 0; }
    }
    f32x2 f[PPT];
    {
        f32x2 tl[PPT], tr[PPT], bl[PPT], br[PPT];
        Lerp Y[PPT], X[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            Y[j] = legacy_coord(yy[j], G.ry, G.h); X[j] = legacy_coord(xx[j], G.rx, G.w);
            const f32x2 *b = reinterpret_cast<const f32x2 *>(flow) + (long long)n * G.h * G.w;
            const int xb = min(X[j].lo, G.w - 2);
            const flow2 top = *reinterpret_cast<const flow2 *>(b + Y[j].lo * G.w + xb), bot = *reinterpret_cast<const flow2 *>(b + Y[j].hi * G.w + xb);
            const bool l1 = X[j].lo != xb, h1 = X[j].hi != xb;
            tl[j] = l1 ? top.b : top.a; tr[j] = h1 ? top.b : top.a;
            bl[j] = l1 ? bot.b : bot.a; br[j] = h1 ? bot.b : bot.a;
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            f[j].x = glue_post_x(lerp2(glue_pre(tl[j].x, G), glue_pre(tr[j].x, G), glue_pre(bl[j].x, G), glue_pre(br[j].x, G), X[j].t, Y[j].t), G);
            f[j].y = glue_post_y(lerp2(glue_pre(tl[j].y, G), glue_pre(tr[j].y, G), glue_pre(bl[j].y, G), glue_pre(br[j].y, G), X[j].t, Y[j].t), G);
            if (WRITE_FLOW && ok[j]) reinterpret_cast<f32x2 *>(outflow)[n * HW + (long long)yy[j] * W + xx[j]] = f[j];
        }
    }
    __syncthreads();                                                  // the table
    const unsigned char *b8 = img + n * HW * 3;
    float wa[PPT], wb[PPT], wc[PPT], wd[PPT];
    rgb3 Ia[PPT], Ib[PPT], Ic[PPT], Id[PPT];
    auto px = [&](int y, int x) {                                     // swap(frame)/255: channels 2, 1, 0 of the BGR pixel
        const unsigned char *q = b8 + ((long long)y * W + x) * 3;
        rgb3 r; r.r = lut[q[2]]; r.g = lut[q[1]]; r.b = lut[q[0]];
        return r;
    };
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const float x = (float)xx[j] + f[j].x, y = (float)yy[j] + f[j].y;
        int x0 = (int)fminf(fmaxf(x, -2.f), (float)W), y0 = (int)fminf(fmaxf(y, -2.f), (float)H);
        int x1 = x0 + 1, y1 = y0 + 1;
        x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
        y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
        const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
        wa[j] = (x1f - x) * (y1f - y); wb[j] = (x1f - x) * (y - y0f);
        wc[j] = (x - x0f) * (y1f - y); wd[j] = (x - x0f) * (y - y0f);
        Ia[j] = px(y0, x0); Ib[j] = px(y1, x0); Ic[j] = px(y0, x1); Id[j] = px(y1, x1);
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        float r[3];
        r[0] = ((wa[j] * Ia[j].r + wb[j] * Ib[j].r) + wc[j] * Ic[j].r) + wd[j] * Id[j].r;      // tf.add_n order
        r[1] = ((wa[j] * Ia[j].g + wb[j] * Ib[j].g) + wc[j] * Ic[j].g) + wd[j] * Id[j].g;
        r[2] = ((wa[j] * Ia[j].b + wb[j] * Ib[j].b) + wc[j] * Ic[j].b) + wd[j] * Id[j].b;
        const int q = j * 4 + wave;
        unsigned char *o = stage + (((q / PPR) * WH + lane / WW) * TW + (q % PPR) * WW + lane % WW) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (unsigned char)fminf(fmaxf(truncf(r[2 - c] * 255.0f), 0.f), 255.f);   // quantise_output_kernel's
    }
    __syncthreads();
    // a tile row is TW*3 = 96 bytes; 4-byte stores where rows start on a 4-byte boundary (W % 4 == 0), bytes otherwise / at a ragged edge
    const int vw3 = min(TW, W - tx0) * 3;
    if ((W & 3) == 0) {
        constexpr int R4 = TW * 3 / 4;
        for (int e = threadIdx.x; e < TH * R4; e += 256) {
            const int row = e / R4, c4 = e - row * R4;
            if (ty0 + row >= H || c4 * 4 >= vw3) continue;
            unsigned char *o = out + (n * HW + (long long)(ty0 + row) * W + tx0) * 3 + c4 * 4;
            if (c4 * 4 + 4 <= vw3) *reinterpret_cast<unsigned *>(o) = *reinterpret_cast<const unsigned *>(stage + row * TW * 3 + c4 * 4);
            else for (int i = 0; c4 * 4 + i < vw3; ++i) o[i] = stage[row * TW * 3 + c4 * 4 + i];
        }
    } else {
        for (int e = threadIdx.x; e < TH * TW * 3; e += 256) {
            const int row = e / (TW * 3), c = e - row * (TW * 3);
            if (ty0 + row < H && c < vw3) out[(n * HW + (long long)(ty0 + row) * W + tx0) * 3 + c] = stage[e];
        }
    }
}

hipError_t launch_flow_glue_warp_u8(const float *flow, int B, int h, int w, const unsigned char *frame, float *outflow, unsigned char *out, int oh,
                                    int ow, int net_h, int net_w, hipStream_t stream)
{
    const long long total = (long long)B * oh * ow;
    if (total == 0) return hipSuccess;
    const int tx = (ow + WT_TW - 1) / WT_TW, ty = (oh + WT_TH - 1) / WT_TH;
    if (w < 2 || total >= (1ll << 31) / 3 || (long long)tx * ty * B >= (1ll << 31) || ((uintptr_t)flow & 7) || ((uintptr_t)outflow & 7) || ((uintptr_t)out & 3))
        return hipErrorInvalidValue;
    const GlueParams G = glue_params(h, w, oh, ow, net_h, net_w);
    const dim3 grid((unsigned)((long long)tx * ty * B)), block(256);
    const double src = 8.0 * B * h * w;                       // 3 B/px frame + 3 B/px out (+ 8 B/px flow written)
    if (outflow) return launch_timed(HBM_SLOT_GLUE_WARP, src + 14.0 * total, warp3_u8_tile_kernel<true>, grid, block, stream, frame, flow, out, outflow, B, oh, ow, tx, ty, G);
    return launch_timed(HBM_SLOT_GLUE_WARP, src + 6.0 * total, warp3_u8_tile_kernel<false>, grid, block, stream, frame, flow, out, (float *)nullptr, B, oh, ow, tx, ty, G);
}

// ---------------------------------------------------------------------------------
// The stand-alone glue (main:497-498, for callers that filter the flow between glue and warp) and the 3-channel legacy-bilinear
// resize (main:806: the unstable frame -- channels 24:27 of the 27-channel stack, read in place through `Cs` / `c_off` -- to
// the flow grid; main:202-203) on the warp's tile mapping: 16 x 32 pixel tiles, 4 x 16 wave patches, the left/right taps of a
// source row fetched as neighbours, rows leaving through LDS as 16-byte stores.  Same statements as flow_resize_scale_kernel /
// resize_bilinear_kernel: bit-identical (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------
template <bool STAGE>
__global__ __launch_bounds__(256) void glue_tile_kernel(const float *__restrict__ flow, float *__restrict__ out, int B, int oh, int ow,
                                                        int tiles_x, int tiles_y, GlueParams G)
{
    constexpr int TW = WT_TW, TH = WT_TH, PPT = WT_PPT, WW = WT_WW, WH = WT_WH, PPR = TW / WW;
    __shared__ __attribute__((aligned(16))) float stage[STAGE ? TH * TW * 2 : 4];
    unsigned bx, by, bz;
    xcd_remap_calc(gridDim.x, 1, 1, blockIdx.x, bx, by, bz);
    const int tpi = tiles_x * tiles_y;
    const int n = (int)bx / tpi, trem = (int)bx - n * tpi;
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * TW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f32x2 *b = reinterpret_cast<const f32x2 *>(flow) + (size_t)n * G.h * G.w;
    flow2 top[PPT], bot[PPT];
    Lerp Y[PPT], X[PPT];
    int yy[PPT], xx[PPT], xb[PPT];
    bool ok[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int q = j * 4 + wave;
        const int y = ty0 + (q / PPR) * WH + lane / WW, x = tx0 + (q % PPR) * WW + lane % WW;
        ok[j] = y < oh && x < ow;
        yy[j] = min(y, oh - 1); xx[j] = min(x, ow - 1);
        Y[j] = legacy_coord(yy[j], G.ry, G.h); X[j] = legacy_coord(xx[j], G.rx, G.w);
        xb[j] = min(X[j].lo, G.w - 2);
        top[j] = *reinterpret_cast<const flow2 *>(b + Y[j].lo * G.w + xb[j]);
        bot[j] = *reinterpret_cast<const flow2 *>(b + Y[j].hi * G.w + xb[j]);
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const bool l1 = X[j].lo != xb[j], h1 = X[j].hi != xb[j];
        const f32x2 tl = l1 ? top[j].b : top[j].a, tr = h1 ? top[j].b : top[j].a, bl = l1 ? bot[j].b : bot[j].a, br = h1 ? bot[j].b : bot[j].a;
        f32x2 o;
        o.x = glue_post_x(lerp2(glue_pre(tl.x, G), glue_pre(tr.x, G), glue_pre(bl.x, G), glue_pre(br.x, G), X[j].t, Y[j].t), G);
        o.y = glue_post_y(lerp2(glue_pre(tl.y, G), glue_pre(tr.y, G), glue_pre(bl.y, G), glue_pre(br.y, G), X[j].t, Y[j].t), G);
        if (STAGE) {
            const int q = j * 4 + wave;
            *reinterpret_cast<f32x2 *>(stage + (((q / PPR) * WH + lane / WW) * TW + (q % PPR) * WW + lane % WW) * 2) = o;
        } else if (ok[j]) {
            reinterpret_cast<f32x2 *>(out)[((size_t)n * oh + yy[j]) * ow + xx[j]] = o;
        }
    }
    if (STAGE) {           // ow even (host): a tile row is 256 bytes from a 16-byte aligned address
        __syncthreads();
        constexpr int R4 = TW * 2 / 4;
        const int vw2 = min(TW, ow - tx0) * 2;
        for (int e = threadIdx.x; e < TH * R4; e += 256) {
            const int row = e / R4, c4 = e - row * R4;
            if (ty0 + row >= oh || c4 * 4 >= vw2) continue;
            float *o = out + (((size_t)n * oh + ty0 + row) * ow + tx0) * 2 + c4 * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + row * TW * 2 + c4 * 4);
            if (c4 * 4 + 4 <= vw2) *reinterpret_cast<f32x4 *>(o) = v;
            else { o[0] = v[0]; o[1] = v[1]; }
        }
    }
}

static hipError_t launch_glue_tile(const float *flow, int B, float *out, int oh, int ow, const GlueParams &G, hipStream_t stream)
{
    const int h = G.h, w = G.w;
    const long long tiles = (long long)((ow + WT_TW - 1) / WT_TW) * ((oh + WT_TH - 1) / WT_TH) * B;
    if (w < 2 || tiles >= (1ll << 31) || (long long)B * oh * ow >= (1ll << 31) || ((uintptr_t)out & 15)) return hipErrorNotSupported;
    const int tx = (ow + WT_TW - 1) / WT_TW, ty = (oh + WT_TH - 1) / WT_TH;
    const double bytes = 8.0 * B * h * w + 8.0 * B * oh * ow;
    if ((ow & 1) == 0)
        return launch_timed(HBM_SLOT_GLUE, bytes, glue_tile_kernel<true>, dim3((unsigned)tiles), dim3(256), stream, flow, out, B, oh, ow, tx, ty, G);
    return launch_timed(HBM_SLOT_GLUE, bytes, glue_tile_kernel<false>, dim3((unsigned)tiles), dim3(256), stream, flow, out, B, oh, ow, tx, ty, G);
}

template <bool STAGE>
__global__ __launch_bounds__(256) void resize3_tile_kernel(const float *__restrict__ x, int B, int h, int w, int Cs, int c_off,
                                                           float *__restrict__ out, int oh, int ow, float ry, float rx, int tiles_x,
                                                           int tiles_y)
{
    constexpr int TW = WT_TW, TH = WT_TH, PPT = WT_PPT, WW = WT_WW, WH = WT_WH, PPR = TW / WW;
    __shared__ __attribute__((aligned(16))) float stage[STAGE ? TH * TW * 3 : 4];
    unsigned bx, by, bz;
    xcd_remap_calc(gridDim.x, 1, 1, blockIdx.x, bx, by, bz);
    const int tpi = tiles_x * tiles_y;
    const int n = (int)bx / tpi, trem = (int)bx - n * tpi;
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * TW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *b = x + (size_t)n * h * w * Cs + c_off;
    rgb3 tl[PPT], tr[PPT], bl[PPT], br[PPT];
    Lerp Y[PPT], X[PPT];
    int yy[PPT], xx[PPT];
    bool ok[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int q = j * 4 + wave;
        const int y = ty0 + (q / PPR) * WH + lane / WW, xo = tx0 + (q % PPR) * WW + lane % WW;
        ok[j] = y < oh && xo < ow;
        yy[j] = min(y, oh - 1); xx[j] = min(xo, ow - 1);
        Y[j] = legacy_coord(yy[j], ry, h); X[j] = legacy_coord(xx[j], rx, w);
        tl[j] = *reinterpret_cast<const rgb3 *>(b + ((size_t)Y[j].lo * w + X[j].lo) * Cs); tr[j] = *reinterpret_cast<const rgb3 *>(b + ((size_t)Y[j].lo * w + X[j].hi) * Cs);
        bl[j] = *reinterpret_cast<const rgb3 *>(b + ((size_t)Y[j].hi * w + X[j].lo) * Cs); br[j] = *reinterpret_cast<const rgb3 *>(b + ((size_t)Y[j].hi * w + X[j].hi) * Cs);
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        rgb3 o;
        o.r = lerp2(tl[j].r, tr[j].r, bl[j].r, br[j].r, X[j].t, Y[j].t);
        o.g = lerp2(tl[j].g, tr[j].g, bl[j].g, br[j].g, X[j].t, Y[j].t);
        o.b = lerp2(tl[j].b, tr[j].b, bl[j].b, br[j].b, X[j].t, Y[j].t);
        if (STAGE) {
            const int q = j * 4 + wave;
            *reinterpret_cast<rgb3 *>(stage + (((q / PPR) * WH + lane / WW) * TW + (q % PPR) * WW + lane % WW) * 3) = o;
        } else if (ok[j]) {
            reinterpret_cast<rgb3 *>(out)[((size_t)n * oh + yy[j]) * ow + xx[j]] = o;
        }
    }
    if (STAGE) {           // ow % 4 == 0 (host)
        __syncthreads();
        constexpr int R4 = TW * 3 / 4;
        const int vw3 = min(TW, ow - tx0) * 3;
        for (int e = threadIdx.x; e < TH * R4; e += 256) {
            const int row = e / R4, c4 = e - row * R4;
            if (ty0 + row >= oh || c4 * 4 >= vw3) continue;
            float *o = out + (((size_t)n * oh + ty0 + row) * ow + tx0) * 3 + c4 * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + row * TW * 3 + c4 * 4);
            if (c4 * 4 + 4 <= vw3) *reinterpret_cast<f32x4 *>(o) = v;
            else for (int i = 0; c4 * 4 + i < vw3; ++i) o[i] = v[i];
        }
    }
}

static hipError_t launch_resize3_tile(const float *x, int B, int h, int w, int Cs, int c_off, float *out, int oh, int ow, hipStream_t stream)
{
    const long long tiles = (long long)((ow + WT_TW - 1) / WT_TW) * ((oh + WT_TH - 1) / WT_TH) * B;
    if (tiles >= (1ll << 31) || (long long)B * oh * ow >= (1ll << 31) || ((uintptr_t)out & 15) || c_off < 0 || c_off + 3 > Cs)
        return hipErrorNotSupported;
    const int tx = (ow + WT_TW - 1) / WT_TW, ty = (oh + WT_TH - 1) / WT_TH;
    const float ry = (float)h / (float)oh, rx = (float)w / (float)ow;
    if ((ow & 3) == 0) resize3_tile_kernel<true><<<dim3((unsigned)tiles), dim3(256), 0, stream>>>(x, B, h, w, Cs, c_off, out, oh, ow, ry, rx, tx, ty);
    else resize3_tile_kernel<false><<<dim3((unsigned)tiles), dim3(256), 0, stream>>>(x, B, h, w, Cs, c_off, out, oh, ow, ry, rx, tx, ty);
    return hipGetLastError();
}

// main:806 without the intermediate copy of the three channels: out [B,oh,ow,3] = resize_images(x[..., c_off:c_off+3], [oh, ow])
hipError_t launch_resize_bilinear_slice3(const float *x, int B, int h, int w, int Cs, int c_off, float *out, int oh, int ow, hipStream_t stream)
{
    return launch_resize3_tile(x, B, h, w, Cs, c_off, out, oh, ow, stream);
}

// tf.nn.max_pool(ksize 2, strides 2, SAME) (vgg16.py:51-53): out = ceil(n/2); the window's taps beyond
// the bottom/right edge do not take part.  One thread per 4 output channels.
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float *__restrict__ x, int B, int H, int W, int C4,
                                                         float *__restrict__ out, int oh, int ow)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * oh * ow * C4;
    if (idx >= total) return;
    const int c = (int)(idx % C4);
    const long long pix = idx / C4;
    const int n = (int)(pix / (oh * ow));
    const int rem = (int)(pix - (long long)n * oh * ow);
    const int oy = rem / ow, ox = rem - oy * ow;
    const int y0 = 2 * oy, x0 = 2 * ox;
    const bool y1ok = y0 + 1 < H, x1ok = x0 + 1 < W;
    const f32x4 *b = reinterpret_cast<const f32x4 *>(x) + (long long)n * H * W * C4 + c;
    f32x4 m = b[((long long)y0 * W + x0) * C4];
    if (x1ok) { const f32x4 v = b[((long long)y0 * W + x0 + 1) * C4]; m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w); }
    if (y1ok) {
        const f32x4 v = b[((long long)(y0 + 1) * W + x0) * C4]; m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        if (x1ok) { const f32x4 u = b[((long long)(y0 + 1) * W + x0 + 1) * C4]; m.x = fmaxf(m.x, u.x); m.y = fmaxf(m.y, u.y); m.z = fmaxf(m.z, u.z); m.w = fmaxf(m.w, u.w); }
    }
    reinterpret_cast<f32x4 *>(out)[idx] = m;
}

hipError_t launch_maxpool2x2(const float *x, int B, int H, int W, int C, float *out, hipStream_t stream)
{
    if (C & 3) return hipErrorInvalidValue;
    const int oh = (H + 1) / 2, ow = (W + 1) / 2;
    const long long total = (long long)B * oh * ow * (C / 4);
    maxpool2x2_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(x, B, H, W, C / 4, out, oh, ow);
    return hipGetLastError();
}

// NLDF.py:29: input * 255 - VGG_MEAN (per channel)
struct Mean4 { float m[4]; };
__global__ __launch_bounds__(256) void scale_shift_kernel(const float *__restrict__ x, long long n, int C, float scale, Mean4 mean,
                                                          float *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % C);
    out[idx] = x[idx] * scale - mean.m[c];
}

// VGG16's first layer (conv1_1, vgg16.py:29: 3x3 SAME on the 3-channel image -> 64 channels, ReLU): K = 27 is a single padded
// K-tile of the MFMA kernel, whose time is then its 4-byte scattered epilogue (1.2 TB/s on the 2.1 GB it writes at 4 x 1080p).  This
// layer is an HBM-bound WRITE, so it gets a plain kernel shaped for the store: 16 lanes per pixel, one float4 of output channels
// each (256 contiguous bytes per pixel), four pixels of a row per thread so that each filter chunk is read from LDS once for the
// four; the 3x6x3 input window comes through L1 (the 16 lanes of a pixel group hit the same addresses), filter + bias sit in LDS.  W: HWIO [3][3][3][cout], cout % 4 == 0, cout <= 64.
__global__ __launch_bounds__(256) void conv3x3_rgb_kernel(const float *__restrict__ x, int B, int H, int W, const float *__restrict__ Wf,
                                                          const float *__restrict__ bias, int cout, int relu, float *__restrict__ out)
{
    __shared__ f32x4 sW[27 * 16];
    __shared__ f32x4 sB[16];
    const int q4 = cout >> 2;
    for (int i = threadIdx.x; i < 27 * q4; i += 256) sW[(i / q4) * 16 + (i % q4)] = *reinterpret_cast<const f32x4 *>(Wf + (long long)(i / q4) * cout + (i % q4) * 4);
    if (threadIdx.x < q4) sB[threadIdx.x] = *reinterpret_cast<const f32x4 *>(bias + threadIdx.x * 4);
    __syncthreads();
    // a thread: 4 consecutive pixels of a row x one float4 of output channels (the filter chunk is read once for the four)
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int q = (int)(idx & 15);
    const long long grp = idx >> 4;
    const int WQ = (W + 3) >> 2;
    if (grp >= (long long)B * H * WQ || q >= q4) return;
    const int n = (int)(grp / ((long long)H * WQ));
    const int rem = (int)(grp - (long long)n * H * WQ);
    const int y = rem / WQ, x0 = (rem - y * WQ) * 4;
    const float *xb = x + (long long)n * H * W * 3;
    f32x4 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = sB[q];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = y + ky - 1;
        const bool yok = (unsigned)iy < (unsigned)H;
        float v[6][3];                                   // input columns x0-1 .. x0+4 of this row
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ix = x0 + j - 1;
            const bool ok = yok && (unsigned)ix < (unsigned)W;
            const float *px = xb + ((long long)iy * W + ix) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[j][c] = ok ? px[c] : 0.f;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 w = sW[((ky * 3 + kx) * 3 + c) * 16 + q];
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[p] += v[p + kx][c] * w;
            }
    }
    float *o = out + (((long long)n * H + y) * W + x0) * cout + q * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (x0 + p >= W) break;
        f32x4 r = acc[p];
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], 0.f);
        }
        *reinterpret_cast<f32x4 *>(o + (long long)p * cout) = r;
    }
}

hipError_t launch_conv3x3_rgb(const float *x, int B, int H, int W, const float *Wf, const float *bias, int cout, int relu, float *out,
                              hipStream_t stream)
{
    if ((cout & 3) || cout > 64 || cout < 4) return hipErrorInvalidValue;
    const long long n = (long long)B * H * ((W + 3) / 4) * 16;
    conv3x3_rgb_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(x, B, H, W, Wf, bias, cout, relu, out);
    return hipGetLastError();
}

hipError_t launch_scale_shift(const float *x, long long npix, int C, float scale, const float *mean4, float *out, hipStream_t stream)
{
    if (C < 1 || C > 4) return hipErrorInvalidValue;
    Mean4 m{};
    for (int c = 0; c < C; ++c) m.m[c] = mean4[c];
    const long long n = npix * C;
    scale_shift_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(x, n, C, scale, m, out);
    return hipGetLastError();
}

// get_pixel_value (main:44-68): out[b,h,w,:] = img[b, y[b,h,w], x[b,h,w], :]
__global__ __launch_bounds__(256) void get_pixel_value_kernel(const float *__restrict__ img, const int32_t *__restrict__ x,
                                                              const int32_t *__restrict__ y, float *__restrict__ out,
                                                              int B, int H, int W, int C, int Hi, int Wi)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * W;
    if (idx >= total) return;
    const int n = (int)(idx / (H * W));
    const int xi = min(max(x[idx], 0), Wi - 1), yi = min(max(y[idx], 0), Hi - 1);
    const float *s = img + (((long long)n * Hi + yi) * Wi + xi) * C;
    float *o = out + idx * C;
    for (int c = 0; c < C; ++c) o[c] = s[c];
}

hipError_t launch_get_pixel_value(const float *img, const int32_t *x, const int32_t *y, float *out, int B, int H, int W,
                                  int C, int Hi, int Wi, hipStream_t stream)
{
    const long long total = (long long)B * H * W;
    get_pixel_value_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(img, x, y, out, B, H, W, C, Hi, Wi);
    return hipGetLastError();
}

}  // namespace vstab
