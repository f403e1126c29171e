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
typedef float f32x4v __attribute__((ext_vector_type(4)));

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

// pass 1: sums[3*b + 0] += sum (P*M - G*M)^2, sums[3*b + 1] += sum safe(M) (3 channels), sums[3*b + 2] += TV.
// pf: [B,h,w,cs_pf] pixels whose first two channels are the flow (cs_pf even).  A workgroup covers LOSS_PPT * 256 pixels, so
// a full-resolution level issues ~64 atomics per sample and sum instead of ~1000 (they serialise on one address).
constexpr int LOSS_PPT = 16;
__global__ __launch_bounds__(256) void loss_sums_kernel(const float *__restrict__ pf, int cs_pf, const float *__restrict__ G,
                                                        const float *__restrict__ U, int h, int w, double *__restrict__ sums)
{
    const int n = blockIdx.y;
    double num = 0.0, den = 0.0, tv = 0.0;
    const float *fb = pf + (long long)n * h * w * cs_pf;
    auto flow = [&](int i) { return *reinterpret_cast<const f32x2 *>(fb + (long long)i * cs_pf); };
    for (int it = 0; it < LOSS_PPT; ++it) {
        const int idx = (blockIdx.x * LOSS_PPT + it) * 256 + threadIdx.x;
        if (idx >= h * w) break;
        const int yy = idx / w, xx = idx - yy * w;
        const f32x2 f = flow(idx);
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
        den += 3.0 * (double)(M == 0.f ? M + 1e-8f : M);
        if (yy + 1 < h) { const f32x2 q = flow(idx + w); tv += (double)fabsf(q.x - f.x) + (double)fabsf(q.y - f.y); }
        if (xx + 1 < w) { const f32x2 q = flow(idx + 1); tv += (double)fabsf(q.x - f.x) + (double)fabsf(q.y - f.y); }
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

// pass 2: grad[b,y,x,0:2] = d( scale_mse * mean_b(num_b/den_b) + scale_tv * TV ) / d pf[b,y,x,:]   (grad: [B,h,w,cs_g] pixels)
__global__ __launch_bounds__(256) void loss_grad_kernel(const float *__restrict__ pf, int cs_pf, const float *__restrict__ G,
                                                        const float *__restrict__ U, int B, int h, int w,
                                                        const double *__restrict__ sums, float scale_mse, float scale_tv,
                                                        float *__restrict__ grad, int cs_g)
{
    const int n = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h * w) return;
    const int yy = idx / w, xx = idx - yy * w;
    const float *fb = pf + (long long)n * h * w * cs_pf;
    auto flow = [&](int i) { return *reinterpret_cast<const f32x2 *>(fb + (long long)i * cs_pf); };
    const f32x2 f = flow(idx);
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
    if (yy + 1 < h) { const f32x2 q = flow(idx + w); tx -= sgn(q.x - f.x); ty -= sgn(q.y - f.y); }
    if (xx + 1 < w) { const f32x2 q = flow(idx + 1); tx -= sgn(q.x - f.x); ty -= sgn(q.y - f.y); }
    if (yy > 0) { const f32x2 q = flow(idx - w); tx += sgn(f.x - q.x); ty += sgn(f.y - q.y); }
    if (xx > 0) { const f32x2 q = flow(idx - 1); tx += sgn(f.x - q.x); ty += sgn(f.y - q.y); }
    f32x2 o;
    o.x = gx + scale_tv * tx;
    o.y = gy + scale_tv * ty;
    *reinterpret_cast<f32x2 *>(grad + ((long long)n * h * w + idx) * cs_g) = o;
}

// loss_main = sum over levels of [ mean_b(num/den) + tv_weight * sum_b TV ]  (main:213-217, 269-275)
struct LossTotalArgs { float tvw[8]; int n; };
__global__ void loss_total_kernel(const double *__restrict__ sums, int B, LossTotalArgs a, double *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double total = 0.0;
    for (int l = 0; l < a.n; ++l) {
        const double *s = sums + (long long)l * 3 * B;
        double mse = 0.0, tv = 0.0;
        for (int b = 0; b < B; ++b) { mse += s[3 * b] / s[3 * b + 1]; tv += s[3 * b + 2]; }
        total += mse / (double)B + (double)a.tvw[l] * tv;
    }
    out[0] = total;
}

static hipError_t loss_level_kernels(const float *pf, int cs_pf, const float *G, const float *U, int B, int h, int w, double *sums,
                                     float scale_mse, float scale_tv, float *grad, int cs_g, hipStream_t stream)
{
    const int npix = h * w;
    loss_sums_kernel<<<dim3((unsigned)((npix + 256 * LOSS_PPT - 1) / (256 * LOSS_PPT)), (unsigned)B), dim3(256), 0, stream>>>(pf, cs_pf, G, U, h, w,
                                                                                                                          sums);
    if (grad)
        loss_grad_kernel<<<dim3((unsigned)((npix + 255) / 256), (unsigned)B), dim3(256), 0, stream>>>(pf, cs_pf, G, U, B, h, w, sums, scale_mse,
                                                                                                    scale_tv, grad, cs_g);
    return hipGetLastError();
}

hipError_t launch_loss_level(const float *pf, const float *G, const float *U, int B, int h, int w, double *sums, float scale_mse,
                             float scale_tv, float *grad, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 3 * (size_t)B, stream);
    if (e != hipSuccess) return e;
    return loss_level_kernels(pf, 2, G, U, B, h, w, sums, scale_mse, scale_tv, grad, 2, stream);
}

size_t loss_main_workspace_bytes(int B, const LossLevel *lv, int n)
{
    size_t img = 0;
    for (int l = 0; l < n; ++l) img = std::max(img, (size_t)B * lv[l].h * lv[l].w * 3 * sizeof(float));
    img = (img + 255) & ~(size_t)255;
    return 256 + (((size_t)n * 3 * B * sizeof(double) + 255) & ~(size_t)255) + 2 * img;
}

// All levels of loss_main in one call: per level the two tf.image.resize_images (main:202-203), the sums and the gradient written
// straight into the caller's (strided) flow-gradient buffer; then the scalar.  22 launches instead of ~80 through the per-level API.
hipError_t launch_loss_main(const LossLevel *lv, int n, const float *gtstab, const float *unstab, int B, int H, int W, double *loss_out,
                            void *workspace, hipStream_t stream)
{
    char *p = static_cast<char *>(workspace);
    double *sums = reinterpret_cast<double *>(p);
    const size_t sbytes = ((size_t)n * 3 * B * sizeof(double) + 255) & ~(size_t)255;
    size_t img = 0;
    for (int l = 0; l < n; ++l) img = std::max(img, (size_t)B * lv[l].h * lv[l].w * 3 * sizeof(float));
    img = (img + 255) & ~(size_t)255;
    float *G = reinterpret_cast<float *>(p + sbytes), *U = reinterpret_cast<float *>(p + sbytes + img);
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)n * 3 * B * sizeof(double), stream);
    if (e != hipSuccess) return e;
    LossTotalArgs a;
    a.n = n;
    for (int l = 0; l < n; ++l) {
        a.tvw[l] = lv[l].tv_weight;
        if ((e = launch_resize_bilinear(gtstab, B, H, W, 3, G, lv[l].h, lv[l].w, stream)) != hipSuccess) return e;
        if ((e = launch_resize_bilinear(unstab, B, H, W, 3, U, lv[l].h, lv[l].w, stream)) != hipSuccess) return e;
        if ((e = loss_level_kernels(lv[l].pf, lv[l].cs_pf, G, U, B, lv[l].h, lv[l].w, sums + (size_t)l * 3 * B, 1.0f, lv[l].tv_weight, lv[l].grad,
                                    lv[l].cs_grad, stream)) != hipSuccess)
            return e;
    }
    loss_total_kernel<<<1, 64, 0, stream>>>(sums, B, a, loss_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// BatchNormLayer(act = lrelu 0.1, is_train = True, gamma_init = None) (model.py:809 etc.; TensorLayer 1.x):
//   mean, var = tf.nn.moments(z, [0,1,2])  (population variance);  y = lrelu((z - mean) * rsqrt(var + eps) + beta);
//   moving = moving * decay + batch * (1 - decay)   (assign_moving_average, zero_debias = False)
// and its backward.  Column reductions over the [rows, C] view of an NHWC channel slice run in two deterministic stages
// (row-chunk partials, then the chunks in order), like column_sum_kernel; the forward's mean and variance come from ONE
// pass (sum and sum of squares in double precision).  The layer works IN PLACE: the conv's output z
// is overwritten by y, and the backward recovers what it needs from y (lrelu is invertible: u = y > 0 ? y : y / 0.1,
// xhat = u - beta), so no second activation copy is kept.
// ---------------------------------------------------------------------------------
namespace {
// the two sums of the backward: sum g and sum g * xhat with g = dy * lrelu'(y), xhat = lrelu^-1(y) - beta; part = [chunk][2][C]
__global__ __launch_bounds__(256) void bn_bwd_colsum_kernel(const float *__restrict__ y, int cs_y, int cy_off, const float *__restrict__ dyp,
                                                            int cs_g, int cg_off, const float *__restrict__ beta, long long rows,
                                                            int rows_per_chunk, int C, float *__restrict__ part)
{
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float bt = beta[c];
        long long r = r0 + wave;
        for (; r + 12 < r1; r += 16) {                               // four rows in flight per wave (HBM-bound pass)
            float yy[4], dy[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { yy[u] = y[(r + 4 * u) * cs_y + cy_off + c]; dy[u] = dyp[(r + 4 * u) * cs_g + cg_off + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float g = yy[u] > 0.f ? dy[u] : 0.1f * dy[u];
                const float xhat = (yy[u] > 0.f ? yy[u] : yy[u] / 0.1f) - bt;
                s0 += g;
                s1 += g * xhat;
            }
        }
        for (; r < r1; r += 4) {
            const float yy = y[r * cs_y + cy_off + c], dy = dyp[r * cs_g + cg_off + c];
            const float g = yy > 0.f ? dy : 0.1f * dy;
            const float xhat = (yy > 0.f ? yy : yy / 0.1f) - bt;
            s0 += g;
            s1 += g * xhat;
        }
    }
    red[0][wave][threadIdx.x & 63] = s0;
    red[1][wave][threadIdx.x & 63] = s1;
    __syncthreads();
    if (wave == 0 && c < C) {
        const int l = threadIdx.x;
        part[((long long)blockIdx.y * 2 + 0) * C + c] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        part[((long long)blockIdx.y * 2 + 1) * C + c] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
    }
}

// forward statistics in ONE pass over z: per-chunk sum and sum of squares in double precision (full-rate fp64 adds on this
// chip; E[z^2] - mean^2 in double loses nothing that matters for fp32 data), part = [chunk][2][C] doubles
__global__ __launch_bounds__(256) void bn_moments_kernel(const float *__restrict__ z, int cs, int c_off, long long rows, int rows_per_chunk,
                                                         int C, double *__restrict__ part)
{
    __shared__ double red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    double s0 = 0.0, s1 = 0.0;
    if (c < C)
    {
        long long r = r0 + wave;
        for (; r + 12 < r1; r += 16) {                               // four rows in flight per wave (HBM-bound pass)
            float x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = z[(r + 4 * u) * cs + c_off + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s0 += (double)x[u]; s1 += (double)x[u] * (double)x[u]; }
        }
        for (; r < r1; r += 4) {
            const double x = (double)z[r * cs + c_off + c];
            s0 += x;
            s1 += x * x;
        }
    }
    red[0][wave][threadIdx.x & 63] = s0;
    red[1][wave][threadIdx.x & 63] = s1;
    __syncthreads();
    if (wave == 0 && c < C) {
        const int l = threadIdx.x;
        part[((long long)blockIdx.y * 2 + 0) * C + c] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        part[((long long)blockIdx.y * 2 + 1) * C + c] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
    }
}

// second stage (one workgroup of 16 waves per 64 channels): wave w adds chunks w, w+16, ... with four loads in flight;
// the 16 partial sums are added in wave order, so the statistics are reproducible.
__global__ __launch_bounds__(1024) void bn_moments_final_kernel(const double *__restrict__ part, int chunks, int C, double inv_rows, float eps,
                                                                float decay, float *__restrict__ mean, float *__restrict__ rstd,
                                                                float *__restrict__ mov_mean, float *__restrict__ mov_var)
{
    __shared__ double red[2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s = 0.0, q = 0.0;
    if (c < C) {
        int k = wave;
        for (; k + 16 < chunks; k += 32) {
            const double a0 = part[((long long)k * 2 + 0) * C + c], b0 = part[((long long)k * 2 + 1) * C + c];
            const double a1 = part[((long long)(k + 16) * 2 + 0) * C + c], b1 = part[((long long)(k + 16) * 2 + 1) * C + c];
            s += a0; q += b0; s += a1; q += b1;
        }
        for (; k < chunks; k += 16) { s += part[((long long)k * 2 + 0) * C + c]; q += part[((long long)k * 2 + 1) * C + c]; }
    }
    red[0][wave][lane] = s;
    red[1][wave][lane] = q;
    __syncthreads();
    if (wave != 0 || c >= C) return;
    s = 0.0; q = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { s += red[0][w][lane]; q += red[1][w][lane]; }
    const double m = s * inv_rows;
    double var = q * inv_rows - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (mov_mean) mov_mean[c] = mov_mean[c] * decay + (float)m * (1.f - decay);
    if (mov_var) mov_var[c] = mov_var[c] * decay + (float)var * (1.f - decay);
}

__global__ __launch_bounds__(256) void bn_lrelu_apply_kernel(float *__restrict__ zy, int cs, int c_off, int C4, long long rows,
                                                             const float *__restrict__ mean, const float *__restrict__ rstd,
                                                             const float *__restrict__ beta)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C4) return;
    const long long r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    float *p = zy + r * cs + c_off + c;
    f32x4v v = *reinterpret_cast<f32x4v *>(p);
    const f32x4v m = *reinterpret_cast<const f32x4v *>(mean + c), s = *reinterpret_cast<const f32x4v *>(rstd + c);
    const f32x4v b = *reinterpret_cast<const f32x4v *>(beta + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float u = (v[e] - m[e]) * s[e] + b[e];
        v[e] = fmaxf(u, 0.1f * u);
    }
    *reinterpret_cast<f32x4v *>(p) = v;
}

// backward sums: sums[c] = sum g, sums[C + c] = sum g * xhat (chunks added in order); dbeta = sum g
__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(const float *__restrict__ part, int chunks, int C, float *__restrict__ sums,
                                                            float *__restrict__ dbeta, int accumulate)
{
    __shared__ float red[2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a = 0.f, b = 0.f;
    if (c < C) {
        int k = wave;
        for (; k + 16 < chunks; k += 32) {
            const float a0 = part[((long long)k * 2 + 0) * C + c], b0 = part[((long long)k * 2 + 1) * C + c];
            const float a1 = part[((long long)(k + 16) * 2 + 0) * C + c], b1 = part[((long long)(k + 16) * 2 + 1) * C + c];
            a += a0; b += b0; a += a1; b += b1;
        }
        for (; k < chunks; k += 16) { a += part[((long long)k * 2 + 0) * C + c]; b += part[((long long)k * 2 + 1) * C + c]; }
    }
    red[0][wave][lane] = a;
    red[1][wave][lane] = b;
    __syncthreads();
    if (wave != 0 || c >= C) return;
    a = 0.f; b = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { a += red[0][w][lane]; b += red[1][w][lane]; }
    sums[c] = a;
    sums[C + c] = b;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + a : a;
}

// dz = rstd * (g - sum_g / R - xhat * sum_gx / R), written over dy
__global__ __launch_bounds__(256) void bn_lrelu_bwd_apply_kernel(const float *__restrict__ y, int cs_y, int cy_off, float *__restrict__ dy,
                                                                 int cs_g, int cg_off, int C, long long rows,
                                                                 const float *__restrict__ sums, const float *__restrict__ beta,
                                                                 const float *__restrict__ rstd, float inv_rows)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const long long r = idx / C;
    const int c = (int)(idx - r * C);
    const float yy = y[r * cs_y + cy_off + c], d = dy[r * cs_g + cg_off + c];
    const float g = yy > 0.f ? d : 0.1f * d;
    const float xhat = (yy > 0.f ? yy : yy / 0.1f) - beta[c];
    dy[r * cs_g + cg_off + c] = rstd[c] * (g - sums[c] * inv_rows - xhat * (sums[C + c] * inv_rows));
}

// leaky-relu backward alone (layers without BatchNorm): dy *= (y > 0 ? 1 : 0.1)
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float *__restrict__ y, int cs_y, int cy_off, float *__restrict__ dy, int cs_g,
                                                        int cg_off, int C, long long rows)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const long long r = idx / C;
    const int c = (int)(idx - r * C);
    if (!(y[r * cs_y + cy_off + c] > 0.f)) dy[r * cs_g + cg_off + c] *= 0.1f;
}
}  // namespace

int bn_chunks(long long rows, int C) { return reduce_chunks(rows, C); }

hipError_t launch_bn_lrelu_train_forward(float *zy, long long rows, int cs, int c_off, int C, const float *beta, float *mov_mean,
                                         float *mov_var, float decay, float eps, float *save_mean, float *save_rstd, float *scratch,
                                         hipStream_t stream)
{
    if ((C & 3) || (cs & 3) || (c_off & 3)) return hipErrorInvalidValue;
    const int chunks = bn_chunks(rows, C);                                 // scratch: 2 * chunks * C doubles
    const int rpc = (int)((rows + chunks - 1) / chunks);
    double *part = reinterpret_cast<double *>(scratch);
    bn_moments_kernel<<<dim3((unsigned)((C + 63) / 64), (unsigned)chunks), dim3(256), 0, stream>>>(zy, cs, c_off, rows, rpc, C, part);
    bn_moments_final_kernel<<<dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, stream>>>(part, chunks, C, 1.0 / (double)rows, eps, decay,
                                                                                       save_mean, save_rstd, mov_mean, mov_var);
    const long long n4 = rows * (C / 4);
    bn_lrelu_apply_kernel<<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream>>>(zy, cs, c_off, C / 4, rows, save_mean, save_rstd, beta);
    return hipGetLastError();
}

hipError_t launch_bn_lrelu_train_backward(const float *y, int cs_y, int cy_off, float *dy, int cs_g, int cg_off, int C, long long rows,
                                          const float *beta, const float *save_rstd, float *dbeta, int accumulate, float *scratch,
                                          hipStream_t stream)
{
    const int chunks = bn_chunks(rows, C);
    const int rpc = (int)((rows + chunks - 1) / chunks);
    bn_bwd_colsum_kernel<<<dim3((unsigned)((C + 63) / 64), (unsigned)chunks), dim3(256), 0, stream>>>(y, cs_y, cy_off, dy, cs_g, cg_off, beta,
                                                                                                   rows, rpc, C, scratch);
    float *sums = scratch + (size_t)2 * chunks * C;                        // scratch: (2 * chunks + 2) * C floats
    bn_bwd_final_kernel<<<dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, stream>>>(scratch, chunks, C, sums, dbeta, accumulate);
    const long long n = rows * C;
    bn_lrelu_bwd_apply_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(y, cs_y, cy_off, dy, cs_g, cg_off, C, rows, sums,
                                                                                         beta, save_rstd, 1.0f / (float)rows);
    return hipGetLastError();
}

hipError_t launch_lrelu_backward(const float *y, int cs_y, int cy_off, float *dy, int cs_g, int cg_off, int C, long long rows,
                                 hipStream_t stream)
{
    const long long n = rows * C;
    lrelu_bwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(y, cs_y, cy_off, dy, cs_g, cg_off, C, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Adjoints of the two resamplers in the decoder, and the full-resolution head's upsampler itself.
// Both backward kernels GATHER (one thread per input element loops over the few output positions that read it and
// re-derives the forward's index / weight with the forward's own float arithmetic), so sums have a fixed order.
// ---------------------------------------------------------------------------------
namespace {
struct LerpT { int lo, hi; float t; };
__device__ __forceinline__ LerpT legacy_coord_t(int o, float scale, int n_in)        // = flow_ops.hip legacy_coord
{
    const float f = (float)o * scale;
    const float fl = floorf(f);
    LerpT L;
    L.lo = min((int)fl, n_in - 1);
    L.hi = min(L.lo + 1, n_in - 1);
    L.t = f - fl;
    return L;
}
__device__ __forceinline__ int nearest_ac_t(int i, float scale, int n_in) { return min((int)roundf((float)i * scale), n_in - 1); }

// backward of tf.image.resize_images / UpSampling2dLayer (legacy bilinear): out = top + (bot - top) * ty with
// top = tl + (tr - tl) * tx  =>  d out / d tl = (1-tx)(1-ty), tr: tx(1-ty), bl: (1-tx)ty, br: tx*ty
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float *__restrict__ dout, int oh, int ow, int C,
                                                                  float *__restrict__ din, int h, int w, float ry, float rx,
                                                                  float gain, int accumulate)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)gridDim.y * 0 + (long long)h * w * C;
    const int n = blockIdx.y;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int pix = (int)(idx / C);
    const int iy = pix / w, ix = pix - iy * w;
    const int oy0 = max(0, (int)floorf((float)(iy - 1) / ry) - 1), oy1 = min(oh - 1, (int)ceilf((float)(iy + 1) / ry) + 1);
    const int ox0 = max(0, (int)floorf((float)(ix - 1) / rx) - 1), ox1 = min(ow - 1, (int)ceilf((float)(ix + 1) / rx) + 1);
    const float *g = dout + (long long)n * oh * ow * C + c;
    float s = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
        const LerpT Y = legacy_coord_t(oy, ry, h);
        const float wy = (Y.lo == iy ? 1.f - Y.t : 0.f) + (Y.hi == iy ? Y.t : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            const LerpT X = legacy_coord_t(ox, rx, w);
            const float wx = (X.lo == ix ? 1.f - X.t : 0.f) + (X.hi == ix ? X.t : 0.f);
            if (wx != 0.f) s += wy * wx * g[((long long)oy * ow + ox) * C];
        }
    }
    float *o = din + ((long long)n * h * w) * C + idx;
    *o = accumulate ? *o + gain * s : gain * s;
}

// F7 of the network: PadLayer(1) then nearest-neighbour resize with align_corners=True to H x W (model.py:795-802, 882-884)
__global__ __launch_bounds__(256) void pad_nearest_up_kernel(const float *__restrict__ src, int h2, int w2, int C4, float *__restrict__ out,
                                                             int H, int W, float sy, float sx)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)H * W * C4) return;
    const int n = blockIdx.y;
    const int c = (int)(idx % C4);
    const int pix = (int)(idx / C4);
    const int y = pix / W, x = pix - y * W;
    const int syi = nearest_ac_t(y, sy, h2 + 2) - 1, sxi = nearest_ac_t(x, sx, w2 + 2) - 1;
    f32x4v v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)syi < (unsigned)h2 && (unsigned)sxi < (unsigned)w2)
        v = reinterpret_cast<const f32x4v *>(src)[(((long long)n * h2 + syi) * w2 + sxi) * C4 + c];
    reinterpret_cast<f32x4v *>(out)[(long long)n * H * W * C4 + idx] = v;
}

__global__ __launch_bounds__(256) void pad_nearest_up_bwd_kernel(const float *__restrict__ dout, int H, int W, int C4, float *__restrict__ dsrc,
                                                                 int h2, int w2, float sy, float sx, int accumulate)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)h2 * w2 * C4) return;
    const int n = blockIdx.y;
    const int c = (int)(idx % C4);
    const int pix = (int)(idx / C4);
    const int syi = pix / w2, sxi = pix - syi * w2;
    // output rows y with nearest(y) == syi + 1: a contiguous range around (syi + 1) / sy
    const int yc = sy > 0.f ? (int)((float)(syi + 1) / sy) : 0, xc = sx > 0.f ? (int)((float)(sxi + 1) / sx) : 0;
    const int span_y = sy > 0.f ? (int)(1.f / sy) + 2 : H, span_x = sx > 0.f ? (int)(1.f / sx) + 2 : W;
    f32x4v s = {0.f, 0.f, 0.f, 0.f};
    for (int y = max(0, yc - span_y); y <= min(H - 1, yc + span_y); ++y) {
        if (nearest_ac_t(y, sy, h2 + 2) != syi + 1) continue;
        for (int x = max(0, xc - span_x); x <= min(W - 1, xc + span_x); ++x) {
            if (nearest_ac_t(x, sx, w2 + 2) != sxi + 1) continue;
            s += reinterpret_cast<const f32x4v *>(dout)[(((long long)n * H + y) * W + x) * C4 + c];
        }
    }
    f32x4v *o = reinterpret_cast<f32x4v *>(dsrc) + (long long)n * h2 * w2 * C4 + idx;
    if (accumulate) s += *o;
    *o = s;
}

// Adjoint of pf2_kernel's gather (flow_ops.hip): the full-resolution head reads, for output pixel (y,x) and tap (dy,dx), the
// tap-table entry of source pixel s = (nearest(y+dy) - 1, nearest(x+dx) - 1) of the UNPADDED concat2, so
//   dT[n, s, tap*2 + o] = sum over the output pixels whose tap (dy,dx) maps to s of g[n, y, x, o].
// One thread per (source pixel, tap); the output rows / columns that map to a source index form a contiguous range
// (found by re-evaluating the forward's index function), so the sum has a fixed order.  Columns 18..31 are zeroed.
__global__ __launch_bounds__(256) void pf2_taps_bwd_kernel(const float *__restrict__ g, int cs_g, int H, int W, float *__restrict__ dT,
                                                           int h2, int w2, float sy, float sx)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h2 * w2 * 16) return;
    const int n = blockIdx.y;
    const int tap = idx & 15;
    const int pix = idx >> 4;
    f32x2 *out = reinterpret_cast<f32x2 *>(dT + ((long long)n * h2 * w2 + pix) * 32) + tap;
    if (tap >= 9) { f32x2 z = {0.f, 0.f}; *out = z; return; }
    const int syi = pix / w2, sxi = pix - syi * w2;
    const int dy = tap / 3, dx = tap - dy * 3;
    const int oh = H - 2, ow = W - 2;
    const int rc = sy > 0.f ? (int)((float)(syi + 1) / sy) : 0, cc = sx > 0.f ? (int)((float)(sxi + 1) / sx) : 0;
    const int span_y = sy > 0.f ? (int)(1.f / sy) + 2 : H, span_x = sx > 0.f ? (int)(1.f / sx) + 2 : W;
    float a0 = 0.f, a1 = 0.f;
    for (int r = max(dy, rc - span_y); r <= min(oh - 1 + dy, rc + span_y); ++r) {          // r = y + dy: row of the padded grid
        if (nearest_ac_t(r, sy, h2 + 2) != syi + 1) continue;
        const int y = r - dy;
        for (int c = max(dx, cc - span_x); c <= min(ow - 1 + dx, cc + span_x); ++c) {
            if (nearest_ac_t(c, sx, w2 + 2) != sxi + 1) continue;
            const float *p = g + (((long long)n * oh + y) * ow + (c - dx)) * cs_g;
            a0 += p[0];
            a1 += p[1];
        }
    }
    f32x2 o; o.x = a0; o.y = a1;
    *out = o;
}

// tf.train.AdamOptimizer (main:333-335): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t m / (sqrt(v) + eps),
// lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) computed by the caller
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, long long n, float lr_t, float b1, float b2, float eps)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] -= lr_t * mi / (sqrtf(vi) + eps);
}
}  // namespace

hipError_t launch_resize_bilinear_backward(const float *dout, int B, int oh, int ow, int C, float *din, int h, int w, float gain,
                                           int accumulate, hipStream_t stream)
{
    const long long per = (long long)h * w * C;
    resize_bilinear_bwd_kernel<<<dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, stream>>>(
        dout, oh, ow, C, din, h, w, (float)h / (float)oh, (float)w / (float)ow, gain, accumulate);
    return hipGetLastError();
}

hipError_t launch_pad_nearest_up(const float *src, int B, int h2, int w2, int C, float *out, int H, int W, hipStream_t stream)
{
    if (C & 3) return hipErrorInvalidValue;
    const float sy = H > 1 ? (float)(h2 + 2 - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w2 + 2 - 1) / (float)(W - 1) : 0.f;
    const long long per = (long long)H * W * (C / 4);
    pad_nearest_up_kernel<<<dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, stream>>>(src, h2, w2, C / 4, out, H, W, sy, sx);
    return hipGetLastError();
}

hipError_t launch_pad_nearest_up_backward(const float *dout, int B, int H, int W, int C, float *dsrc, int h2, int w2, int accumulate,
                                          hipStream_t stream)
{
    if (C & 3) return hipErrorInvalidValue;
    const float sy = H > 1 ? (float)(h2 + 2 - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w2 + 2 - 1) / (float)(W - 1) : 0.f;
    const long long per = (long long)h2 * w2 * (C / 4);
    pad_nearest_up_bwd_kernel<<<dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, stream>>>(dout, H, W, C / 4, dsrc, h2, w2, sy,
                                                                                                      sx, accumulate);
    return hipGetLastError();
}

hipError_t launch_pf2_taps_backward(const float *g, int cs_g, int B, int H, int W, float *dT, int h2, int w2, hipStream_t stream)
{
    const float sy = H > 1 ? (float)(h2 + 2 - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w2 + 2 - 1) / (float)(W - 1) : 0.f;
    const int per = h2 * w2 * 16;
    pf2_taps_bwd_kernel<<<dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, stream>>>(g, cs_g, H, W, dT, h2, w2, sy, sx);
    return hipGetLastError();
}

hipError_t launch_adam(float *w, const float *g, float *m, float *v, long long n, float lr_t, float b1, float b2, float eps,
                       hipStream_t stream)
{
    adam_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(w, g, m, v, n, lr_t, b1, b2, eps);
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

// The same packing without a table for the common case: a plain [K][N] matrix (HWIO with its first three axes flattened -- a layer
// whose K runs are whole 32-float chunks -- or one Winograd position) -> [K/32][Npad][32] with the 16-byte chunks of row n XOR-swizzled
// by ((n >> 1) & 7).  A 32 x 64 block goes through LDS so that both the read (rows of N) and the write (128-byte rows) are coalesced;
// the table replay reads 4 scattered bytes per element plus its 4-byte index.  grid (K/32, Npad/64, batch); N % 4 == 0.
__global__ __launch_bounds__(256) void pack_blocked_kernel(const float *__restrict__ W, int K, int N, int Npad, float *__restrict__ wpk)
{
    __shared__ float tile[32][65];
    const int kt = blockIdx.x, n0 = blockIdx.y * 64;
    const float *Wb = W + (long long)blockIdx.z * K * N;
    float *ob = wpk + (long long)blockIdx.z * (K / 32) * Npad * 32;
    {
        const int r = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
        const float *src = Wb + (long long)(kt * 32 + r) * N + n0 + c8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            if (n0 + c8 + 4 * h < N) v = *reinterpret_cast<const f32x4v *>(src + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r][c8 + 4 * h + e] = v[e];
        }
    }
    __syncthreads();
    const int nl = threadIdx.x >> 2, n = n0 + nl;
    if (n >= Npad) return;
    float *orow = ob + ((long long)kt * Npad + n) * 32;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = (threadIdx.x & 3) * 2 + h;
        f32x4v v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[4 * c + e][nl];
        *reinterpret_cast<f32x4v *>(orow + ((c ^ ((n >> 1) & 7)) << 2)) = v;
    }
}

hipError_t launch_pack_blocked(const float *W, int K, int N, int Npad, int batch, float *wpk, hipStream_t stream)
{
    if ((K & 31) || (N & 3) || (Npad & 63) || N > Npad || batch < 1) return hipErrorInvalidValue;
    pack_blocked_kernel<<<dim3((unsigned)(K / 32), (unsigned)(Npad / 64), (unsigned)batch), dim3(256), 0, stream>>>(W, K, N, Npad, wpk);
    return hipGetLastError();
}

// The table replay for operands whose SOURCE is contiguous along the output column n (a forward convolution's HWIO filter read in tap
// mode: conv3 ... conv6 over the concat buffers -- whatever table table_is_n_fast() accepts in train_api.cpp's forward plans; the
// input-gradient plans build their operands with their own kernels): pack_apply_kernel walks a packed row --
// 32 consecutive k of one column -- and so gathers 4-byte words a whole filter row (N floats) apart.  Here a workgroup owns 64 consecutive
// packed rows (one K-tile x 64 columns, 8 KB): the table block is read coalesced into LDS, thread (k, n) fetches W[index of (n, k)] with
// n running fastest -- neighbouring lanes read neighbouring words -- and the block leaves as coalesced 16-byte stores.
__global__ __launch_bounds__(256) void pack_apply_nfast_kernel(const float *__restrict__ W, const int32_t *__restrict__ tbl, long long rows,
                                                               float *__restrict__ wpk)
{
    __shared__ int32_t idx[64][33];
    __shared__ float val[64][33];
    const long long r0 = (long long)blockIdx.x * 64;
    const int nr = (int)min((long long)64, rows - r0);
    for (int e = threadIdx.x; e < nr * 32; e += 256) idx[e >> 5][e & 31] = tbl[r0 * 32 + e];
    __syncthreads();
    const int nl = threadIdx.x & 63;
    if (nl < nr) {
        const int n = (int)((r0 + nl) & 0x7fffffff);                        // only bits 1..3 of the row number enter the swizzle
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int k = (threadIdx.x >> 6) + 4 * it;
            const int pos = ((((k >> 2) ^ ((n >> 1) & 7)) << 2) | (k & 3));
            const int32_t t = idx[nl][pos];
            val[nl][pos] = t ? W[t - 1] : 0.f;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nr * 32; e += 256) wpk[r0 * 32 + e] = val[e >> 5][e & 31];
}

hipError_t launch_pack_apply(const float *W, const int32_t *tbl, long long n, float *wpk, hipStream_t stream, bool n_fast)
{
    if (n_fast && (n & 31) == 0) {
        const long long rows = n >> 5;
        pack_apply_nfast_kernel<<<dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, stream>>>(W, tbl, rows, wpk);
        return hipGetLastError();
    }
    pack_apply_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(W, tbl, n, wpk);
    return hipGetLastError();
}

}  // namespace vstab
