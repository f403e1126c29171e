// Secondary samplers named by the reference's spatial_transformer.py and warp.py (SURVEY.md 8a rows
// S1-S3).  All are HBM/latency-bound 4-tap gathers: one thread per output pixel, coordinates
// generated in-kernel (no grid tensor is materialised), coalesced stores.
// -ffp-contract=off keeps the weight arithmetic the reference's op-by-op fp32 sequence.
#include "vstab_internal.h"

namespace vstab {

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
__global__ __launch_bounds__(256) void homography_warp_kernel(const float *__restrict__ img, int B, int Hi, int Wi, int C,
                                                              const float *__restrict__ M, float *__restrict__ out, int oh,
                                                              int ow)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * oh * ow) return;
    const int n = (int)(idx / (oh * ow));
    const int rem = (int)(idx - (long long)n * oh * ow);
    const int oy = rem / ow, ox = rem - oy * ow;
    const float X = ow > 1 ? (float)(-1.0 + (double)ox * (2.0 / (double)(ow - 1))) : -1.0f;
    const float Y = oh > 1 ? (float)(-1.0 + (double)oy * (2.0 / (double)(oh - 1))) : -1.0f;
    const float *m = M + (long long)n * 9;
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

hipError_t launch_st_interp(const float *img, int B, int H, int W, int C, const float *x, const float *y, int npix, float *out,
                            hipStream_t stream)
{
    const long long total = (long long)B * npix;
    st_interp_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(img, B, H, W, C, x, y, npix, out);
    return hipGetLastError();
}

hipError_t launch_st_transform(const float *img, int B, int H, int W, int C, const float *theta, int tdim, float *out, int oh,
                               int ow, hipStream_t stream)
{
    const long long total = (long long)B * oh * ow;
    st_transform_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(img, B, H, W, C, theta, tdim, out, oh, ow);
    return hipGetLastError();
}

hipError_t launch_st_meshgrid(float *out, int oh, int ow, hipStream_t stream)
{
    st_meshgrid_kernel<<<dim3((unsigned)((oh * ow + 255) / 256)), dim3(256), 0, stream>>>(out, oh, ow);
    return hipGetLastError();
}

hipError_t launch_homography_warp(const float *img, int B, int Hi, int Wi, int C, const float *M, float *out, int oh, int ow,
                                  hipStream_t stream)
{
    const long long total = (long long)B * oh * ow;
    homography_warp_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(img, B, Hi, Wi, C, M, out, oh, ow);
    return hipGetLastError();
}

hipError_t launch_vec2mtrx(const float *p, int B, int dim, int approx, float *out, hipStream_t stream)
{
    vec2mtrx_kernel<<<dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream>>>(p, B, dim, approx, out);
    return hipGetLastError();
}

}  // namespace vstab
