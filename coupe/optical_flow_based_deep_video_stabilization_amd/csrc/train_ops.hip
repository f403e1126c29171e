// Training objective of the reference (first slice): the per-level photometric term
// `lossterm` / `masked_MSE` (main:188-210, "main" = main_flownetS_pyramid_noprevloss_dataloader.py) and the
// total-variation regulariser (main:269-273), forward AND the gradient with respect to the predicted flow --
// the tensor the network's backward pass starts from.
//
//   lossterm(pf, stab, unstab):  G = resize(stab, pf.size), U = resize(unstab, pf.size)      (done by the caller)
//                                P = tf_warp(U, pf), M = tf_warp(ones, pf)
//                                masked_MSE = mean_b [ sum (P*M - G*M)^2 / sum safe(M) ],  safe(0) = 1e-8
//   total_variation(pf) = sum_b sum |pf[y+1,x]-pf[y,x]| + |pf[y,x+1]-pf[y,x]|  over both channels
//
// Gradient (what TF's autodiff yields): tf_warp's corner indices come from integer casts and carry no gradient;
// the flow enters through the four bilinear weights only.  M = (x1-x0)(y1-y0) of the CLIPPED corners does not
// depend on the fractional position, so its derivative vanishes (TF's four terms cancel) and
//   dL/dfx = 2 M^2 / (B den_b) * sum_c (P_c - G_c) * [ (y1-y)(Ic-Ia) + (y-y0)(Id-Ib) ]_c        (same for fy)
// Two HBM-bound passes: (1) per-sample sums num_b, den_b and the TV sum (double atomics), (2) gradient.
// Hand-written for gfx950 (64-lane wavefront reductions).
#include <hip/hip_runtime.h>

#include "vstab_internal.h"

namespace vstab {

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
struct Corners {
    int x0, x1, y0, y1;
    float x, y, x0f, x1f, y0f, y1f;
};

// tf_warp's sampling geometry (main:83-101), identical to warp_flow_kernel
__device__ __forceinline__ Corners corners(int xx, int yy, f32x2 f, int H, int W)
{
    Corners c;
    c.x = (float)xx + f.x;
    c.y = (float)yy + f.y;
    int x0 = (int)fminf(fmaxf(c.x, -2.f), (float)W), y0 = (int)fminf(fmaxf(c.y, -2.f), (float)H);
    int x1 = x0 + 1, y1 = y0 + 1;
    c.x0 = min(max(x0, 0), W - 1); c.x1 = min(max(x1, 0), W - 1);
    c.y0 = min(max(y0, 0), H - 1); c.y1 = min(max(y1, 0), H - 1);
    c.x0f = (float)c.x0; c.x1f = (float)c.x1; c.y0f = (float)c.y0; c.y1f = (float)c.y1;
    return c;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
}  // namespace

// pass 1: sums[3*b + 0] += sum (P*M - G*M)^2, sums[3*b + 1] += sum safe(M) (3 channels), sums[3*b + 2] += TV
__global__ __launch_bounds__(256) void loss_sums_kernel(const float *__restrict__ pf, const float *__restrict__ G,
                                                        const float *__restrict__ U, int h, int w, double *__restrict__ sums)
{
    const int n = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    double num = 0.0, den = 0.0, tv = 0.0;
    if (idx < h * w) {
        const int yy = idx / w, xx = idx - yy * w;
        const f32x2 *fb = reinterpret_cast<const f32x2 *>(pf) + (long long)n * h * w;
        const f32x2 f = fb[idx];
        const Corners c = corners(xx, yy, f, h, w);
        const float wa = (c.x1f - c.x) * (c.y1f - c.y), wb = (c.x1f - c.x) * (c.y - c.y0f);
        const float wc = (c.x - c.x0f) * (c.y1f - c.y), wd = (c.x - c.x0f) * (c.y - c.y0f);
        const float M = ((wa + wb) + wc) + wd;                       // tf_warp(ones): add_n of the four weights
        const float *ub = U + (long long)n * h * w * 3;
        const float *Ia = ub + (c.y0 * w + c.x0) * 3, *Ib = ub + (c.y1 * w + c.x0) * 3;
        const float *Ic = ub + (c.y0 * w + c.x1) * 3, *Id = ub + (c.y1 * w + c.x1) * 3;
        const float *g = G + ((long long)n * h * w + idx) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float P = ((wa * Ia[ch] + wb * Ib[ch]) + wc * Ic[ch]) + wd * Id[ch];
            const float d = P * M - g[ch] * M;
            num += (double)(d * d);
        }
        den = 3.0 * (double)(M == 0.f ? M + 1e-8f : M);
        if (yy + 1 < h) { const f32x2 q = fb[idx + w]; tv += (double)fabsf(q.x - f.x) + (double)fabsf(q.y - f.y); }
        if (xx + 1 < w) { const f32x2 q = fb[idx + 1]; tv += (double)fabsf(q.x - f.x) + (double)fabsf(q.y - f.y); }
    }
    __shared__ double red[3][4];
    num = wave_sum(num); den = wave_sum(den); tv = wave_sum(tv);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = num; red[1][wave] = den; red[2][wave] = tv; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
        atomicAdd(&sums[3 * n + threadIdx.x], s);
    }
}

// pass 2: grad[b,y,x,:] = d( scale_mse * mean_b(num_b/den_b) + scale_tv * TV ) / d pf[b,y,x,:]
__global__ __launch_bounds__(256) void loss_grad_kernel(const float *__restrict__ pf, const float *__restrict__ G,
                                                        const float *__restrict__ U, int B, int h, int w,
                                                        const double *__restrict__ sums, float scale_mse, float scale_tv,
                                                        float *__restrict__ grad)
{
    const int n = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h * w) return;
    const int yy = idx / w, xx = idx - yy * w;
    const f32x2 *fb = reinterpret_cast<const f32x2 *>(pf) + (long long)n * h * w;
    const f32x2 f = fb[idx];
    const Corners c = corners(xx, yy, f, h, w);
    const float wa = (c.x1f - c.x) * (c.y1f - c.y), wb = (c.x1f - c.x) * (c.y - c.y0f);
    const float wc = (c.x - c.x0f) * (c.y1f - c.y), wd = (c.x - c.x0f) * (c.y - c.y0f);
    const float M = ((wa + wb) + wc) + wd;
    const float *ub = U + (long long)n * h * w * 3;
    const float *Ia = ub + (c.y0 * w + c.x0) * 3, *Ib = ub + (c.y1 * w + c.x0) * 3;
    const float *Ic = ub + (c.y0 * w + c.x1) * 3, *Id = ub + (c.y1 * w + c.x1) * 3;
    const float *g = G + ((long long)n * h * w + idx) * 3;
    const float k = scale_mse * 2.f * M * M / ((float)B * (float)sums[3 * n + 1]);
    float gx = 0.f, gy = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float a = Ia[ch], b = Ib[ch], cc = Ic[ch], d = Id[ch];
        const float P = ((wa * a + wb * b) + wc * cc) + wd * d;
        const float e = P - g[ch];
        gx += e * ((c.y1f - c.y) * (cc - a) + (c.y - c.y0f) * (d - b));
        gy += e * ((c.x1f - c.x) * (b - a) + (c.x - c.x0f) * (d - cc));
    }
    gx *= k; gy *= k;
    // total variation: d|q - f| / df = -sign(q - f) at this pixel, +sign(f - p) from the neighbour above / to the left
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    float tx = 0.f, ty = 0.f;
    if (yy + 1 < h) { const f32x2 q = fb[idx + w]; tx -= sgn(q.x - f.x); ty -= sgn(q.y - f.y); }
    if (xx + 1 < w) { const f32x2 q = fb[idx + 1]; tx -= sgn(q.x - f.x); ty -= sgn(q.y - f.y); }
    if (yy > 0) { const f32x2 q = fb[idx - w]; tx += sgn(f.x - q.x); ty += sgn(f.y - q.y); }
    if (xx > 0) { const f32x2 q = fb[idx - 1]; tx += sgn(f.x - q.x); ty += sgn(f.y - q.y); }
    f32x2 o;
    o.x = gx + scale_tv * tx;
    o.y = gy + scale_tv * ty;
    reinterpret_cast<f32x2 *>(grad)[(long long)n * h * w + idx] = o;
}

hipError_t launch_loss_level(const float *pf, const float *G, const float *U, int B, int h, int w, double *sums, float scale_mse,
                             float scale_tv, float *grad, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 3 * (size_t)B, stream);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)((h * w + 255) / 256), (unsigned)B);
    loss_sums_kernel<<<grid, dim3(256), 0, stream>>>(pf, G, U, h, w, sums);
    if (grad) loss_grad_kernel<<<grid, dim3(256), 0, stream>>>(pf, G, U, B, h, w, sums, scale_mse, scale_tv, grad);
    return hipGetLastError();
}

// device-side weight packing: replay the host packer's gather from its index table
__global__ __launch_bounds__(256) void pack_apply_kernel(const float *__restrict__ W, const int32_t *__restrict__ tbl, long long n,
                                                         float *__restrict__ wpk)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t t = tbl[i];
    wpk[i] = t ? W[t - 1] : 0.f;
}

hipError_t launch_pack_apply(const float *W, const int32_t *tbl, long long n, float *wpk, hipStream_t stream)
{
    pack_apply_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(W, tbl, n, wpk);
    return hipGetLastError();
}

}  // namespace vstab
