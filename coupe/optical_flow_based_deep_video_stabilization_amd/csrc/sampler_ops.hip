// Samplers named by the reference's spatial_transformer.py and warp.py (BASELINE configs[2]'s
// "spatial_transformer warp").  HBM-bound 4-tap gathers, coordinates generated in-kernel (no grid tensor is materialised).
// 3-channel frames run on st3_tile_kernel (2-D tiles, 3-dword corner gathers, rows leaving as 16-byte stores); other channel
// counts on the one-thread-per-pixel kernels.
// -ffp-contract=off keeps the weight arithmetic the reference's op-by-op fp32 sequence.
#include "vstab_internal.h"
#include "hbm_profile.h"

namespace vstab {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// tf.linspace(-1, 1, n)[i] in fp32 (start + i*step, step = 2/(n-1); a single point is -1)
__device__ __forceinline__ float lin11(int i, int n)
{
    const float step = n > 1 ? 2.0f / (float)(n - 1) : 0.0f;
    return -1.0f + (float)i * step;
}

// ---------------------------------------------------------------------------------
// bilinear_interp (spatial_transformer.py:902-964): normalised coordinates in [-1,1] against an image
// zero-padded by one pixel; x = (x+1)/2*(W-1), clipped to [-1, W], shifted by the pad; x0 = floor,
// x1 = min(x0+1, W+1) as index but weights use the UNclipped x0+1.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void st_sample_pixel(const float *__restrict__ img, int n, int H, int W, int C, float xn, float yn,
                                                float *__restrict__ o)
{
    const float wf = (float)W, hf = (float)H;
    float x = (xn + 1.0f) / 2.0f * (wf - 1.0f);
    float y = (yn + 1.0f) / 2.0f * (hf - 1.0f);
    x = fminf(fmaxf(x, -1.0f), wf - 1.0f + 1.0f);       // clip_by_value(x, -edge, W-1+edge); NaN -> -1
    y = fminf(fmaxf(y, -1.0f), hf - 1.0f + 1.0f);
    x += 1.0f;
    y += 1.0f;
    const float x0f = floorf(x), y0f = floorf(y);
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const int x0 = (int)x0f, y0 = (int)y0f;              // in [0, W+1] after the clip
    const int x1 = (int)fminf(x1f, wf - 1.0f + 2.0f), y1 = (int)fminf(y1f, hf - 1.0f + 2.0f);
    const float w00 = (x1f - x) * (y1f - y), w01 = (x - x0f) * (y1f - y);
    const float w10 = (x1f - x) * (y - y0f), w11 = (x - x0f) * (y - y0f);
    // padded index p in [0, W+1]: image column p-1, zero on the border
    const bool vx0 = x0 >= 1 && x0 <= W, vx1 = x1 >= 1 && x1 <= W, vy0 = y0 >= 1 && y0 <= H, vy1 = y1 >= 1 && y1 <= H;
    const float *b = img + (long long)n * H * W * C;
    const long long i00 = ((long long)(y0 - 1) * W + (x0 - 1)) * C, i01 = ((long long)(y0 - 1) * W + (x1 - 1)) * C;
    const long long i10 = ((long long)(y1 - 1) * W + (x0 - 1)) * C, i11 = ((long long)(y1 - 1) * W + (x1 - 1)) * C;
    for (int c = 0; c < C; ++c) {
        const float I00 = (vx0 && vy0) ? b[i00 + c] : 0.f, I01 = (vx1 && vy0) ? b[i01 + c] : 0.f;
        const float I10 = (vx0 && vy1) ? b[i10 + c] : 0.f, I11 = (vx1 && vy1) ? b[i11 + c] : 0.f;
        o[c] = ((w00 * I00 + w01 * I01) + w10 * I10) + w11 * I11;      // tf.add_n order
    }
}

// explicit coordinates: x, y flat [B*npix] (spatial_transformer.py:902)
__global__ __launch_bounds__(256) void st_interp_kernel(const float *__restrict__ img, int B, int H, int W, int C,
                                                        const float *__restrict__ x, const float *__restrict__ y,
                                                        int npix, float *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * npix) return;
    const int n = (int)(idx / npix);
    st_sample_pixel(img, n, H, W, C, x[idx], y[idx], out + idx * C);
}

// AffineTransformer / ProjectiveTransformer .transform (spatial_transformer.py:400-452, 539-608):
// T_g = theta . (x_t, y_t, 1) on the linspace(-1,1) grid of the OUTPUT size; projective divides by
// z with z == 0 replaced by z + 1e-8 (:598).
__global__ __launch_bounds__(256) void st_transform_kernel(const float *__restrict__ img, int B, int H, int W, int C,
                                                           const float *__restrict__ theta, int tdim,
                                                           float *__restrict__ out, int oh, int ow)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * oh * ow) return;
    const int n = (int)(idx / (oh * ow));
    const int rem = (int)(idx - (long long)n * oh * ow);
    const int oy = rem / ow, ox = rem - oy * ow;
    const float xt = lin11(ox, ow), yt = lin11(oy, oh);
    const float *t = theta + (long long)n * tdim;
    float xs = (t[0] * xt + t[1] * yt) + t[2];
    float ys = (t[3] * xt + t[4] * yt) + t[5];
    if (tdim == 8) {
        float zs = (t[6] * xt + t[7] * yt) + 1.0f;
        if (zs == 0.0f) zs = zs + 1e-8f;
        xs = xs / zs;
        ys = ys / zs;
    }
    st_sample_pixel(img, n, H, W, C, xs, ys, out + idx * C);
}

// _meshgrid(out_size) (spatial_transformer.py:755-779): flat [3*oh*ow] = x_t row, y_t row, ones
__global__ __launch_bounds__(256) void st_meshgrid_kernel(float *__restrict__ out, int oh, int ow)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int np = oh * ow;
    if (idx >= np) return;
    const int oy = idx / ow, ox = idx - oy * ow;
    out[idx] = lin11(ox, ow);
    out[np + idx] = lin11(oy, oh);
    out[2 * np + idx] = 1.0f;
}

// ---------------------------------------------------------------------------------
// warp.transformImage / transformCropImage (warp.py:46-129): homography from the canonical
// [-1,1]^2 grid (np.linspace in float64, cast to fp32) straight to source PIXEL coordinates,
// /(z+1e-8), floor/ceil taps, taps outside the image read an appended zero row.
// M = refMtrx . pMtrx, row-major [B,9].
// ---------------------------------------------------------------------------------
// M = refMtrx . pMtrx (warp.py:48-49's tf.matmul) composed here when `ref` is given: every product and sum its own fp32 operation,
// (r0*p0 + r1*p1) + r2*p2, like the grid products below -- no library GEMM in front of the launch
__device__ __forceinline__ void compose3(const float *__restrict__ ref, const float *__restrict__ p, float *m)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) m[3 * i + j] = (ref[3 * i] * p[j] + ref[3 * i + 1] * p[3 + j]) + ref[3 * i + 2] * p[6 + j];
}

__global__ __launch_bounds__(256) void homography_warp_kernel(const float *__restrict__ img, int B, int Hi, int Wi, int C,
                                                              const float *__restrict__ M, const float *__restrict__ ref,
                                                              float *__restrict__ out, int oh, int ow)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * oh * ow) return;
    const int n = (int)(idx / (oh * ow));
    const int rem = (int)(idx - (long long)n * oh * ow);
    const int oy = rem / ow, ox = rem - oy * ow;
    const float X = ow > 1 ? (float)(-1.0 + (double)ox * (2.0 / (double)(ow - 1))) : -1.0f;
    const float Y = oh > 1 ? (float)(-1.0 + (double)oy * (2.0 / (double)(oh - 1))) : -1.0f;
    float m[9];
    if (ref) compose3(ref, M + (long long)n * 9, m);
    else {
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = M[(long long)n * 9 + k];
    }
    const float xh = (m[0] * X + m[1] * Y) + m[2];
    const float yh = (m[3] * X + m[4] * Y) + m[5];
    const float zh = (m[6] * X + m[7] * Y) + m[8];
    const float xw = xh / (zh + 1e-8f), yw = yh / (zh + 1e-8f);
    const float xf = floorf(xw), xc = ceilf(xw), yf = floorf(yw), yc = ceilf(yw);
    // clamp before the int conversion (out-of-range float->int is undefined); anything outside is "outside"
    const float lim = 1.0e9f;
    const int xfi = (int)fminf(fmaxf(xf, -lim), lim), xci = (int)fminf(fmaxf(xc, -lim), lim);
    const int yfi = (int)fminf(fmaxf(yf, -lim), lim), yci = (int)fminf(fmaxf(yc, -lim), lim);
    const float xr = xw - xf, yr = yw - yf;
    const bool fx = xfi >= 0 && xfi < Wi, cx = xci >= 0 && xci < Wi, fy = yfi >= 0 && yfi < Hi, cy = yci >= 0 && yci < Hi;
    const float *b = img + (long long)n * Hi * Wi * C;
    const float wUL = (1.0f - xr) * (1.0f - yr), wUR = xr * (1.0f - yr), wBL = (1.0f - xr) * yr, wBR = xr * yr;
    float *o = out + idx * C;
    for (int c = 0; c < C; ++c) {
        const float UL = (fx && fy) ? b[((long long)yfi * Wi + xfi) * C + c] : 0.f;
        const float UR = (cx && fy) ? b[((long long)yfi * Wi + xci) * C + c] : 0.f;
        const float BL = (fx && cy) ? b[((long long)yci * Wi + xfi) * C + c] : 0.f;
        const float BR = (cx && cy) ? b[((long long)yci * Wi + xci) * C + c] : 0.f;
        // image*(1-Xratio)*(1-Yratio) evaluates left to right: (I*(1-xr))*(1-yr)
        o[c] = (((UL * (1.0f - xr)) * (1.0f - yr) + (UR * xr) * (1.0f - yr)) + (BL * (1.0f - xr)) * yr) + (BR * xr) * yr;
    }
    (void)wUL; (void)wUR; (void)wBL; (void)wBR;
}

// ---------------------------------------------------------------------------------
// 3-channel frames: all three sampler families on one tile kernel (24 B/px algorithmic: 12 gathered + 12 written, + 8 with
// explicit coordinates).
//   FAM_ST_THETA  Affine/ProjectiveTransformer.transform   (spatial_transformer.py:400-452, 539-608)
//   FAM_ST_COORDS bilinear_interp with explicit x, y       (spatial_transformer.py:902-964)
//   FAM_HOMOG     warp.transformImage / transformCropImage (warp.py:46-129)
// Shaped like tf_warp's tile kernel (flow_ops.hip, warp3_tile_kernel; profiles/README.md "r02 warp study"): a workgroup owns a
// 16 x 32 tile of ONE sample's output pixels and a wave instruction works on a 4 x 16 patch, so the lines a gather touches are a
// compact 2-D footprint under any rotation; one 3-dword load per corner; results leave through LDS as 16-byte stores of whole
// 384-byte tile rows (ow % 4 == 0; 12-byte stores otherwise); XCD-contiguous tile order.
// Measured and rejected twice (profiles/README.md "r03 sampler study"): staging the source window in LDS -- bounding box of the taps
// by DPP reductions, aligned 16-byte fill, corners from LDS -- cuts the L1 lookups 3x (0.50 instead of 1.56 per pixel) and is
// SLOWER both per workgroup tile (commit 2076632: four barriers, three dependent phases; 0.34-0.39 of 8 TB/s against 0.53-0.65)
// and per wave patch with no barrier at all (0.41-0.49): what bounds these kernels is requests in flight, not tag lookups.
// The arithmetic is st_sample_pixel's / homography_warp_kernel's statement for statement: bit-identical results.
// ---------------------------------------------------------------------------------
enum { FAM_ST_THETA = 0, FAM_ST_COORDS = 1, FAM_HOMOG = 2 };
struct __attribute__((packed, aligned(4))) rgb3 { float r, g, b; };
// theta [B,tdim] (M [B,9] for FAM_HOMOG) or x, y [B*oh*ow]; the grid steps 2/(n-1) are divided once on the host (the same IEEE
// quotient lin11 / homography_warp_kernel compute per pixel): fp32 for tf.linspace, fp64 for np.linspace
struct StSrc { const float *theta; const float *x; const float *y; int tdim; float sx, sy; double dsx, dsy; const float *ref; };      // ref: FAM_HOMOG's refMtrx (theta = pMtrx then) or null

constexpr int ST_TW = 32, ST_TH = 16, ST_PPT = 2, ST_WW = 16, ST_WH = 4, ST_PPR = ST_TW / ST_WW;
static_assert(ST_WH * (4 * ST_PPT) / ST_PPR == ST_TH, "tile shape");

// the four taps of one output pixel: image coordinates clamped into the image (what is addressed), validity per axis (what
// counts: an invalid tap reads as zero) and the blend weights (ST: w00, w01, w10, w11; homography: xr, yr)
struct Taps { int xa, xb, ya, yb; bool vxa, vxb, vya, vyb; float w0, w1, w2, w3; };

__device__ __forceinline__ Taps st_taps(float xn, float yn, int H, int W)          // st_sample_pixel's arithmetic
{
    const float wf = (float)W, hf = (float)H;
    float x = (xn + 1.0f) / 2.0f * (wf - 1.0f);
    float y = (yn + 1.0f) / 2.0f * (hf - 1.0f);
    x = fminf(fmaxf(x, -1.0f), wf - 1.0f + 1.0f);
    y = fminf(fmaxf(y, -1.0f), hf - 1.0f + 1.0f);
    x += 1.0f;
    y += 1.0f;
    const float x0f = floorf(x), y0f = floorf(y);
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = (int)fminf(x1f, wf - 1.0f + 2.0f), y1 = (int)fminf(y1f, hf - 1.0f + 2.0f);
    Taps t;
    t.w0 = (x1f - x) * (y1f - y); t.w1 = (x - x0f) * (y1f - y);
    t.w2 = (x1f - x) * (y - y0f); t.w3 = (x - x0f) * (y - y0f);
    t.vxa = x0 >= 1 && x0 <= W; t.vxb = x1 >= 1 && x1 <= W; t.vya = y0 >= 1 && y0 <= H; t.vyb = y1 >= 1 && y1 <= H;
    t.xa = min(max(x0 - 1, 0), W - 1); t.xb = min(max(x1 - 1, 0), W - 1);
    t.ya = min(max(y0 - 1, 0), H - 1); t.yb = min(max(y1 - 1, 0), H - 1);
    return t;
}

__device__ __forceinline__ Taps homog_taps(const float *__restrict__ m, int ox, int oy, double dsx, double dsy, int Hi, int Wi)   // homography_warp_kernel's
{
    const float X = (float)(-1.0 + (double)ox * dsx);
    const float Y = (float)(-1.0 + (double)oy * dsy);
    const float xh = (m[0] * X + m[1] * Y) + m[2];
    const float yh = (m[3] * X + m[4] * Y) + m[5];
    const float zh = (m[6] * X + m[7] * Y) + m[8];
    const float xw = xh / (zh + 1e-8f), yw = yh / (zh + 1e-8f);
    const float xf = floorf(xw), xc = ceilf(xw), yf = floorf(yw), yc = ceilf(yw);
    const float lim = 1.0e9f;
    const int xfi = (int)fminf(fmaxf(xf, -lim), lim), xci = (int)fminf(fmaxf(xc, -lim), lim);
    const int yfi = (int)fminf(fmaxf(yf, -lim), lim), yci = (int)fminf(fmaxf(yc, -lim), lim);
    Taps t;
    t.w0 = xw - xf; t.w1 = yw - yf; t.w2 = 0.f; t.w3 = 0.f;
    t.vxa = xfi >= 0 && xfi < Wi; t.vxb = xci >= 0 && xci < Wi; t.vya = yfi >= 0 && yfi < Hi; t.vyb = yci >= 0 && yci < Hi;
    t.xa = min(max(xfi, 0), Wi - 1); t.xb = min(max(xci, 0), Wi - 1);
    t.ya = min(max(yfi, 0), Hi - 1); t.yb = min(max(yci, 0), Hi - 1);
    return t;
}

template <int FAM>
__device__ __forceinline__ float st_blend(const Taps &t, float I00, float I01, float I10, float I11)
{
    if (FAM == FAM_HOMOG) {
        const float xr = t.w0, yr = t.w1;        // image*(1-Xratio)*(1-Yratio) evaluates left to right
        return (((I00 * (1.0f - xr)) * (1.0f - yr) + (I01 * xr) * (1.0f - yr)) + (I10 * (1.0f - xr)) * yr) + (I11 * xr) * yr;
    }
    return ((t.w0 * I00 + t.w1 * I01) + t.w2 * I10) + t.w3 * I11;      // tf.add_n order
}

template <int FAM, bool STAGE>
__global__ __launch_bounds__(256) void st3_tile_kernel(const float *__restrict__ img, int B, int H, int W, StSrc S,
                                                       float *__restrict__ out, int oh, int ow, int tiles_x, int tiles_y)
{
    constexpr int TW = ST_TW, TH = ST_TH, PPT = ST_PPT, WW = ST_WW, WH = ST_WH, PPR = ST_PPR;
    __shared__ __attribute__((aligned(16))) float lds[STAGE ? ST_TH * ST_TW * 3 : 4];
    unsigned bx, by, bz;
    xcd_remap_calc(gridDim.x, 1, 1, blockIdx.x, bx, by, bz);
    const int tpi = tiles_x * tiles_y;
    const int n = (int)bx / tpi, trem = (int)bx - n * tpi;
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * TW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int HW = H * W;                              // 3 B H W < 2^31 (host)

    Taps t[PPT];
    int yy[PPT], xx[PPT];
    bool ok[PPT];
    float th[9];
    if (FAM != FAM_ST_COORDS) {
        const float *tp = S.theta + (long long)n * S.tdim;       // wave-uniform: scalar loads
#pragma unroll
        for (int k = 0; k < 9; ++k) th[k] = k < S.tdim ? tp[k] : 1.0f;
        if (FAM == FAM_HOMOG && S.ref) {                           // M = refMtrx . pMtrx, here instead of a GEMM launch in front
            float pm[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) pm[k] = th[k];
            compose3(S.ref, pm, th);
        }
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int q = j * 4 + wave;
        const int y = ty0 + (q / PPR) * WH + lane / WW, x = tx0 + (q % PPR) * WW + lane % WW;
        ok[j] = y < oh && x < ow;
        yy[j] = min(y, oh - 1); xx[j] = min(x, ow - 1);          // a pixel beyond the output repeats an edge pixel of this tile (not stored)
        if (FAM == FAM_ST_THETA) {
            const float xt = -1.0f + (float)xx[j] * S.sx, yt = -1.0f + (float)yy[j] * S.sy;
            float xs = (th[0] * xt + th[1] * yt) + th[2];
            float ys = (th[3] * xt + th[4] * yt) + th[5];
            if (S.tdim == 8) {
                float zs = (th[6] * xt + th[7] * yt) + 1.0f;
                if (zs == 0.0f) zs = zs + 1e-8f;
                xs = xs / zs;
                ys = ys / zs;
            }
            t[j] = st_taps(xs, ys, H, W);
        } else if (FAM == FAM_ST_COORDS) {
            const long long i = ((long long)n * oh + yy[j]) * ow + xx[j];
            t[j] = st_taps(S.x[i], S.y[i], H, W);
        } else {
            t[j] = homog_taps(th, xx[j], yy[j], S.dsx, S.dsy, H, W);
        }
    }
    const rgb3 *b = reinterpret_cast<const rgb3 *>(img) + (long long)n * HW;
    rgb3 I00[PPT], I01[PPT], I10[PPT], I11[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        I00[j] = b[t[j].ya * W + t[j].xa]; I01[j] = b[t[j].ya * W + t[j].xb];
        I10[j] = b[t[j].yb * W + t[j].xa]; I11[j] = b[t[j].yb * W + t[j].xb];
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const bool v00 = t[j].vxa && t[j].vya, v01 = t[j].vxb && t[j].vya, v10 = t[j].vxa && t[j].vyb, v11 = t[j].vxb && t[j].vyb;
        rgb3 r;
        r.r = st_blend<FAM>(t[j], v00 ? I00[j].r : 0.f, v01 ? I01[j].r : 0.f, v10 ? I10[j].r : 0.f, v11 ? I11[j].r : 0.f);
        r.g = st_blend<FAM>(t[j], v00 ? I00[j].g : 0.f, v01 ? I01[j].g : 0.f, v10 ? I10[j].g : 0.f, v11 ? I11[j].g : 0.f);
        r.b = st_blend<FAM>(t[j], v00 ? I00[j].b : 0.f, v01 ? I01[j].b : 0.f, v10 ? I10[j].b : 0.f, v11 ? I11[j].b : 0.f);
        if (STAGE) {
            const int q = j * 4 + wave;
            *reinterpret_cast<rgb3 *>(lds + (((q / PPR) * WH + lane / WW) * TW + (q % PPR) * WW + lane % WW) * 3) = r;
        } else if (ok[j]) {
            reinterpret_cast<rgb3 *>(out)[((long long)n * oh + yy[j]) * ow + xx[j]] = r;
        }
    }
    if (STAGE) {       // ow % 4 == 0 (host): a tile row is TW*12 bytes from a 16-byte aligned address
        __syncthreads();
        constexpr int R4 = TW * 3 / 4;
        const int vw3 = min(TW, ow - tx0) * 3;
        for (int e = threadIdx.x; e < TH * R4; e += 256) {
            const int row = e / R4, c4 = e - row * R4;
            if (ty0 + row >= oh || c4 * 4 >= vw3) continue;
            float *o = out + (((long long)n * oh + ty0 + row) * ow + tx0) * 3 + c4 * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(lds + row * TW * 3 + c4 * 4);
            if (c4 * 4 + 4 <= vw3) *reinterpret_cast<f32x4 *>(o) = v;
            else for (int i = 0; c4 * 4 + i < vw3; ++i) o[i] = v[i];
        }
    }
}

// picks the instantiation for a 3-channel launch; hipErrorNotSupported when the shape is not the tile kernel's
template <int FAM>
static hipError_t launch_st3(int slot, const float *img, int B, int H, int W, StSrc S, float *out, int oh, int ow, hipStream_t stream)
{
    S.sx = ow > 1 ? 2.0f / (float)(ow - 1) : 0.0f; S.sy = oh > 1 ? 2.0f / (float)(oh - 1) : 0.0f;
    S.dsx = ow > 1 ? 2.0 / (double)(ow - 1) : 0.0; S.dsy = oh > 1 ? 2.0 / (double)(oh - 1) : 0.0;
    const long long tx = (ow + ST_TW - 1) / ST_TW, ty = (oh + ST_TH - 1) / ST_TH, tiles = tx * ty * B;
    if (tiles >= (1ll << 31) || (long long)B * H * W * 3 >= (1ll << 31) || (long long)B * oh * ow * 3 >= (1ll << 31)) return hipErrorNotSupported;
    const bool stage = (ow & 3) == 0 && ((uintptr_t)out & 15) == 0;
    const double bytes = (FAM == FAM_ST_COORDS ? 32.0 : 24.0) * B * oh * ow;       // every output pixel reads ~one source pixel, writes one (+ x, y)
    const dim3 grid((unsigned)tiles), block(256);
    if (stage) return launch_timed(slot, bytes, st3_tile_kernel<FAM, true>, grid, block, stream, img, B, H, W, S, out, oh, ow, (int)tx, (int)ty);
    return launch_timed(slot, bytes, st3_tile_kernel<FAM, false>, grid, block, stream, img, B, H, W, S, out, oh, ow, (int)tx, (int)ty);
}

// warp.vec2mtrx (warp.py:25-43): sl(3) / affine generator -> matrix exponential by Taylor series,
// pMtrx = sum_{i=0}^{warpApprox-1} A^i / i!   (fp32, one thread per batch element)
__global__ void vec2mtrx_kernel(const float *__restrict__ p, int B, int dim, int approx, float *__restrict__ out)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= B) return;
    const float *q = p + (long long)n * dim;
    float A[9];
    if (dim == 8) {
        A[0] = q[2]; A[1] = q[1]; A[2] = q[0];
        A[3] = q[5]; A[4] = -q[2] - q[6]; A[5] = q[4];
        A[6] = q[3]; A[7] = q[7]; A[8] = q[6];
    } else {
        A[0] = q[0]; A[1] = q[1]; A[2] = q[2];
        A[3] = q[3]; A[4] = q[4]; A[5] = q[5];
        A[6] = 0.f; A[7] = 0.f; A[8] = 0.f;
    }
    float P[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Nm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float denom = 1.0f;
    for (int i = 1; i < approx; ++i) {
        float T[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) T[r * 3 + c] = (Nm[r * 3] * A[c] + Nm[r * 3 + 1] * A[3 + c]) + Nm[r * 3 + 2] * A[6 + c];
        denom *= (float)i;
        for (int k = 0; k < 9; ++k) { Nm[k] = T[k]; P[k] += T[k] / denom; }
    }
    for (int k = 0; k < 9; ++k) out[(long long)n * 9 + k] = P[k];
}

hipError_t launch_st_interp(const float *img, int B, int H, int W, int C, const float *x, const float *y, int oh, int ow, float *out,
                            hipStream_t stream)
{
    const int npix = oh * ow;
    const long long total = (long long)B * npix;
    if (C == 3) {
        const hipError_t e = launch_st3<FAM_ST_COORDS>(HBM_SLOT_ST, img, B, H, W, StSrc{nullptr, x, y, 0, 0.f, 0.f, 0., 0., nullptr}, out, oh, ow, stream);
        if (e != hipErrorNotSupported) return e;
    }
    st_interp_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(img, B, H, W, C, x, y, npix, out);
    return hipGetLastError();
}

hipError_t launch_st_transform(const float *img, int B, int H, int W, int C, const float *theta, int tdim, float *out, int oh,
                               int ow, hipStream_t stream)
{
    const long long total = (long long)B * oh * ow;
    if (C == 3) {
        const hipError_t e = launch_st3<FAM_ST_THETA>(HBM_SLOT_ST, img, B, H, W, StSrc{theta, nullptr, nullptr, tdim, 0.f, 0.f, 0., 0., nullptr}, out, oh, ow, stream);
        if (e != hipErrorNotSupported) return e;
    }
    st_transform_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(img, B, H, W, C, theta, tdim, out, oh, ow);
    return hipGetLastError();
}

hipError_t launch_st_meshgrid(float *out, int oh, int ow, hipStream_t stream)
{
    st_meshgrid_kernel<<<dim3((unsigned)((oh * ow + 255) / 256)), dim3(256), 0, stream>>>(out, oh, ow);
    return hipGetLastError();
}

hipError_t launch_homography_warp(const float *img, int B, int Hi, int Wi, int C, const float *M, float *out, int oh, int ow,
                                  hipStream_t stream, const float *ref)
{
    const long long total = (long long)B * oh * ow;
    if (C == 3) {
        const hipError_t e = launch_st3<FAM_HOMOG>(HBM_SLOT_HOMOG, img, B, Hi, Wi, StSrc{M, nullptr, nullptr, 9, 0.f, 0.f, 0., 0., ref}, out, oh, ow, stream);
        if (e != hipErrorNotSupported) return e;
    }
    homography_warp_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(img, B, Hi, Wi, C, M, ref, out, oh, ow);
    return hipGetLastError();
}

hipError_t launch_vec2mtrx(const float *p, int B, int dim, int approx, float *out, hipStream_t stream)
{
    vec2mtrx_kernel<<<dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream>>>(p, B, dim, approx, out);
    return hipGetLastError();
}

}  // namespace vstab
