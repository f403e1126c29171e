// Winograd F(2x2, 3x3) around the MFMA kernel for the four 3x3 stride-1 stages of the encoder (conv3_1, conv4_1, conv5_1,
// conv6_1; model.py:818-844): Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 2x2 output tile, summed over the input channels.
// The element-wise product summed over channels is, for each of the 16 positions xi of the 4x4 transformed tile, a plain
// GEMM [tiles x Cin] x [Cin x Cout] -- it runs on conv_mfma_kernel as a 16-phase 1x1 convolution (phase = xi) -- and costs
// 16 * 2 * tiles * Cin * Cout = 4/9 of the direct convolution's multiply-adds.  The two transforms are HBM-bound passes:
//   wino_input_kernel :  x [B,H,W,Cs] (channels c_off..+C) -> V [B,16,TH,TW,C],   V_xi = (B^T d B)_xi, d = 4x4 patch at
//                        rows 2ty-1.., columns 2tx-1.. (zero outside the image: the layer's pad 1)
//   wino_output_kernel:  M [B,16,TH,TW,C] -> out [B,Ho,Wo,Cs_out] (channels c_off..+C), Y = A^T M A + bias, leaky relu
// One thread per (tile, 4 channels): 16 float4 loads, the transform in registers, 16 (or 4) float4 stores; consecutive
// threads are consecutive channel quads, so every access is a full 16-byte-per-lane row segment.
// Same result as the direct convolution up to fp32 rounding (the transforms only add and halve).
#include <hip/hip_runtime.h>

#include "vstab_internal.h"

namespace vstab {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The transformed planes of a LARGE stage leave as non-temporal stores (NT = true): V is 4x (9/4 x) the stage's input and, from about the
// Infinity Cache's 256 MB down to half of it, writing it through the caches only evicts what the GEMM that follows would still find there
// (round 6, one box: conv3_1's transform at B=8 512x512, V = 134 MB, 33.9 -> 26.4 us and its GEMM 133.2 -> 130.7 us; the SMALLER V of
// deconv3, 64 MB, is better left cached: its GEMM 155.5 -> 171.9 us with non-temporal stores) -- launch_wino_input / launch_wdec_input decide
#ifndef VSTAB_NT_MIN_BYTES
#define VSTAB_NT_MIN_BYTES (100ll << 20)          // (A/B builds: scripts/build_variant_lib.sh -DVSTAB_NT_MIN_BYTES=...)
#endif
template <bool NT>
__device__ __forceinline__ void vstore(float *p, const f32x4 v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
    else *reinterpret_cast<f32x4 *>(p) = v;
}

// (sample_stride, pos_stride) in floats: (16*tiles*C, tiles*C) = [B][16][tiles][C] for the forward GEMM's phases,
// (tiles*C, B*tiles*C) = [16][B*tiles][C] when the reduction runs over all tiles of the batch (filter gradient)
template <bool NT>
__global__ __launch_bounds__(256) void wino_input_kernel(const float *__restrict__ x, int H, int W, int Cs, int c_off, int C4,
                                                         float *__restrict__ V, int TH, int TW, long long sample_stride, long long pos_stride)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)TH * TW * C4) return;
    const int n = blockIdx.y;
    const int c = (int)(idx % C4);
    const int tile = (int)(idx / C4);
    const int ty = tile / TW, tx = tile - ty * TW;
    const float *xb = x + (long long)n * H * W * Cs + c_off + c * 4;
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 2 * ty - 1 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = 2 * tx - 1 + j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) v = *reinterpret_cast<const f32x4 *>(xb + ((long long)y * W + xx) * Cs);
            d[i][j] = v;
        }
    }
    // B^T d (rows), then (.) B (columns): B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    f32x4 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
    }
    float *vb = V + (long long)n * sample_stride + (long long)tile * (C4 * 4) + c * 4;
    const long long xs = pos_stride;                                         // stride between the 16 positions
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        vstore<NT>(vb + (i * 4 + 0) * xs, t[i][0] - t[i][2]);
        vstore<NT>(vb + (i * 4 + 1) * xs, t[i][1] + t[i][2]);
        vstore<NT>(vb + (i * 4 + 2) * xs, t[i][2] - t[i][1]);
        vstore<NT>(vb + (i * 4 + 3) * xs, t[i][1] - t[i][3]);
    }
}

__global__ __launch_bounds__(256) void wino_output_kernel(const float *__restrict__ M, int TH, int TW, int C4, const float *__restrict__ bias,
                                                          int act, float *__restrict__ out, int Ho, int Wo, int Cs_out, int c_off)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)TH * TW * C4) return;
    const int n = blockIdx.y;
    const int c = (int)(idx % C4);
    const int tile = (int)(idx / C4);
    const int ty = tile / TW, tx = tile - ty * TW;
    const float *mb = M + (((long long)n * 16) * TH * TW + tile) * (C4 * 4) + c * 4;
    const long long xs = (long long)TH * TW * (C4 * 4);
    f32x4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const f32x4 *>(mb + (i * 4 + j) * xs);
    // A^T m (rows), then (.) A (columns): A^T = [1 1 1 0; 0 1 -1 -1]
    f32x4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = (m[0][j] + m[1][j]) + m[2][j];
        s[1][j] = (m[1][j] - m[2][j]) - m[3][j];
    }
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + c * 4);
    const float slope = act == 1 ? 0.1f : 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = 2 * ty + a;
        if (oy >= Ho) continue;
        f32x4 y0 = ((s[a][0] + s[a][1]) + s[a][2]) + bv;
        f32x4 y1 = ((s[a][1] - s[a][2]) - s[a][3]) + bv;
        float *ob = out + (((long long)n * Ho + oy) * Wo + 2 * tx) * Cs_out + c_off + c * 4;
        if (act == 3) {                                         // accumulate (gradient sums)
            y0 += *reinterpret_cast<const f32x4 *>(ob);
            if (2 * tx + 1 < Wo) y1 += *reinterpret_cast<const f32x4 *>(ob + Cs_out);
        } else if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { y0[e] = fmaxf(y0[e], slope * y0[e]); y1[e] = fmaxf(y1[e], slope * y1[e]); }
        }
        *reinterpret_cast<f32x4 *>(ob) = y0;
        if (2 * tx + 1 < Wo) *reinterpret_cast<f32x4 *>(ob + Cs_out) = y1;
    }
}

// Winograd-domain weights on the device (training: the filter changes every step): Wt[xi = 4 i + j][k][n] = sum_{a,b} G[i][a] G[j][b] w(a,b,k,n)
// with w(a,b,k,n) = W[a][b][k][n] (forward: k = input channel, n = output channel) or, for the input gradient, the flipped filter
// with the channel roles swapped, W[2-a][2-b][n][k] (k = output channel of the layer, n = its input channel)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float *__restrict__ W, int cin, int cout, int transpose, float *__restrict__ Wt)
{
    const int K = transpose ? cout : cin, N = transpose ? cin : cout;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)K * N) return;
    const int k = (int)(idx / N), n = (int)(idx - (long long)k * N);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            g[a][b] = transpose ? W[(((long long)(2 - a) * 3 + (2 - b)) * cin + n) * cout + k] : W[(((long long)a * 3 + b) * cin + k) * cout + n];
    // G g (rows): G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], then (.) G^T (columns)
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * ((g[0][b] + g[1][b]) + g[2][b]);
        t[2][b] = 0.5f * ((g[0][b] - g[1][b]) + g[2][b]);
        t[3][b] = g[2][b];
    }
    const long long xs = (long long)K * N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        Wt[(i * 4 + 0) * xs + idx] = t[i][0];
        Wt[(i * 4 + 1) * xs + idx] = 0.5f * ((t[i][0] + t[i][1]) + t[i][2]);
        Wt[(i * 4 + 2) * xs + idx] = 0.5f * ((t[i][0] - t[i][1]) + t[i][2]);
        Wt[(i * 4 + 3) * xs + idx] = t[i][2];
    }
}

// Output-gradient transform of the Winograd-domain filter gradient: dM = A dY A^T per 2x2 tile (A = [1 0; 1 1; 1 -1; 0 -1], the
// transpose of the output transform's A^T; pixels of a tile that lie outside an odd-sized image count as zero).
// dy [B,H,W,Cs] (channels c_off..+C) -> dM [B,16,TH,TW,C], one thread per (tile, 4 channels) like the input transform.
__global__ __launch_bounds__(256) void wino_outgrad_kernel(const float *__restrict__ dy, int H, int W, int Cs, int c_off, int C4,
                                                           float *__restrict__ dM, int TH, int TW, long long sample_stride, long long pos_stride)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)TH * TW * C4) return;
    const int n = blockIdx.y;
    const int c = (int)(idx % C4);
    const int tile = (int)(idx / C4);
    const int ty = tile / TW, tx = tile - ty * TW;
    const float *yb = dy + (long long)n * H * W * Cs + c_off + c * 4;
    f32x4 d[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int y = 2 * ty + a, xx = 2 * tx + b;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y < H && xx < W) v = *reinterpret_cast<const f32x4 *>(yb + ((long long)y * W + xx) * Cs);
            d[a][b] = v;
        }
    f32x4 t[4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        t[0][b] = d[0][b];
        t[1][b] = d[0][b] + d[1][b];
        t[2][b] = d[0][b] - d[1][b];
        t[3][b] = -d[1][b];
    }
    float *mb = dM + (long long)n * sample_stride + (long long)tile * (C4 * 4) + c * 4;
    const long long xs = pos_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4 *>(mb + (i * 4 + 0) * xs) = t[i][0];
        *reinterpret_cast<f32x4 *>(mb + (i * 4 + 1) * xs) = t[i][0] + t[i][1];
        *reinterpret_cast<f32x4 *>(mb + (i * 4 + 2) * xs) = t[i][0] - t[i][1];
        *reinterpret_cast<f32x4 *>(mb + (i * 4 + 3) * xs) = -t[i][1];
    }
}

// dg[a][b][ci][co] = sum_{i,j} G[i][a] G[j][b] dU[4 i + j][ci][co]  (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]): the 16 position
// gradients back to the 3x3 filter, HWIO
__global__ __launch_bounds__(256) void wino_filter_grad_kernel(const float *__restrict__ dU, long long n, float *__restrict__ dW)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[i][j] = dU[(long long)(i * 4 + j) * n + idx];
    float t[3][4];                                   // G^T u: rows a
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
        t[1][j] = 0.5f * (u[1][j] - u[2][j]);
        t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        dW[(long long)(a * 3 + 0) * n + idx] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
        dW[(long long)(a * 3 + 1) * n + idx] = 0.5f * (t[a][1] - t[a][2]);
        dW[(long long)(a * 3 + 2) * n + idx] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Winograd F(2x2, 2x2) for the 4x4 stride-2 SAME transposed convolutions of the decoder (model.py:850-851, 859-860, 868-869, 877-878:
// y[oy,ox,co] = sum x[iy,ix,ci] W[ky,kx,co,ci] over oy = 2 iy + ky - 1).  Per axis, output parity p is a 2-tap stride-1 correlation of the
// input:  y[2i] = x[i-1] W[3] + x[i] W[1],  y[2i+1] = x[i] W[2] + x[i+1] W[0].  Tile t takes the three inputs d = (x[2t-1], x[2t], x[2t+1])
// and yields FOUR outputs: the even pair y[4t], y[4t+2] (taps g = (W[3], W[1])) and the odd pair y[4t-1], y[4t+1] (g = (W[2], W[0])) --
// both pairs read the SAME d, so one transformed tile V feeds all four output phases:
//     v = (d0 - d1, d1, d1 - d2),  u = (g0, g0 + g1, g1),  m_i = v_i u_i,  pair = (m0 + m1, m1 - m2)       3 multiplies for 2 outputs
// In 2-D: 9 positions, each a GEMM [tiles x Cin] x [Cin x 4 Cout] (the four phases side by side in N): 36 multiply-adds per 4x4 output
// block instead of 64.  The transforms have coefficients 0, +-1 only.  Tile grid per axis: NT = Ho/4 + 1 (the odd pair of tile 0 starts at
// output -1, so an even Ho needs one more tile than Hin/2: its d is (x[Hin-1], 0, 0) and only v0 is non-zero); position row i needs the
// tiles t < nt[i] with nt[0] = min(NT, Hin/2 + 1), nt[1] = nt[2] = min(NT, (Hin+1)/2) -- the others are identically zero and are
// neither written here nor multiplied (the GEMM's phase grids are nt_y[i] x nt_x[j], the inverse transform reads them as zeros).
//   wdec_input_kernel :  x [B,Hi,Wi,Cs] (all Cs floats of a pixel: the concat's pad channels are zeros and meet zero weights)
//                        -> V [B][9][NTy][NTx][Cs]
//   wdec_output_kernel:  M [B][9][NTy][NTx][4*Cout] (column = phase * Cout + co, phase = 2 py + px) -> out [B,Ho,Wo,Cs_out] channels
//                        c_off..+Cout, + bias, leaky relu
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(256) void wdec_input_kernel(const float *__restrict__ x, int Hi, int Wi, int C4, float *__restrict__ V, const WdecGeom g)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)g.NTy * g.NTx * C4) return;
    const int n = blockIdx.y;
    const int c = (int)(idx % C4);
    const int tile = (int)(idx / C4);
    const int ty = tile / g.NTx, tx = tile - ty * g.NTx;
    const int Cs = C4 * 4;
    const float *xb = x + (long long)n * Hi * Wi * Cs + c * 4;
    f32x4 d[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int y = 2 * ty - 1 + i;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int xx = 2 * tx - 1 + j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)Hi && (unsigned)xx < (unsigned)Wi) v = *reinterpret_cast<const f32x4 *>(xb + ((long long)y * Wi + xx) * Cs);
            d[i][j] = v;
        }
    }
    f32x4 t[3][3];                                  // rows: (d0 - d1, d1, d1 - d2)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0][j] = d[0][j] - d[1][j];
        t[1][j] = d[1][j];
        t[2][j] = d[1][j] - d[2][j];
    }
    const long long plane = (long long)g.NTy * g.NTx * Cs;
    float *vb = V + (long long)n * 9 * plane + (long long)tile * Cs + c * 4;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (ty >= g.nty[i]) continue;
        if (tx < g.ntx[0]) vstore<NT>(vb + (i * 3 + 0) * plane, t[i][0] - t[i][1]);
        if (tx < g.ntx[1]) vstore<NT>(vb + (i * 3 + 1) * plane, t[i][1]);
        if (tx < g.ntx[2]) vstore<NT>(vb + (i * 3 + 2) * plane, t[i][1] - t[i][2]);
    }
}

__global__ __launch_bounds__(256) void wdec_output_kernel(const WdecOutArgs A)
{
    wdec_output_item(A, (long long)blockIdx.x * 256 + threadIdx.x, (int)blockIdx.y);
}

hipError_t launch_wdec_input(const float *x, int B, int Hi, int Wi, int Cs, float *V, const WdecGeom &g, hipStream_t stream)
{
    if ((Cs & 3) || B < 1) return hipErrorInvalidValue;
    const long long per = (long long)g.NTy * g.NTx * (Cs / 4);
    const dim3 grid((unsigned)((per + 255) / 256), (unsigned)B);
    if ((long long)B * 9 * g.NTy * g.NTx * Cs * 4 >= VSTAB_NT_MIN_BYTES) wdec_input_kernel<true><<<grid, dim3(256), 0, stream>>>(x, Hi, Wi, Cs / 4, V, g);
    else wdec_input_kernel<false><<<grid, dim3(256), 0, stream>>>(x, Hi, Wi, Cs / 4, V, g);
    return hipGetLastError();
}

hipError_t launch_wdec_output(const float *M, int B, int Ho, int Wo, int cout, const float *bias, int act, float *out, int Cs_out, int c_off,
                              const WdecGeom &g, hipStream_t stream)
{
    if ((cout & 3) || (Cs_out & 3) || (c_off & 3) || B < 1) return hipErrorInvalidValue;
    const WdecOutArgs A{M, cout / 4, bias, act, out, Ho, Wo, Cs_out, c_off, g};
    const long long per = (long long)g.NTy * g.NTx * cout;
    wdec_output_kernel<<<dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, stream>>>(A);
    return hipGetLastError();
}

// dM: [16][B*tiles][C] (position-major over the whole batch)
hipError_t launch_wino_outgrad(const float *dy, int B, int H, int W, int Cs, int c_off, int C, float *dM, hipStream_t stream)
{
    if ((C & 3) || (Cs & 3) || (c_off & 3)) return hipErrorInvalidValue;
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    const long long per = (long long)TH * TW * (C / 4);
    wino_outgrad_kernel<<<dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, stream>>>(dy, H, W, Cs, c_off, C / 4, dM, TH, TW,
                                                                                                (long long)TH * TW * C, (long long)B * TH * TW * C);
    return hipGetLastError();
}

hipError_t launch_wino_filter_grad(const float *dU, int cin, int cout, float *dW, hipStream_t stream)
{
    const long long n = (long long)cin * cout;
    wino_filter_grad_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(dU, n, dW);
    return hipGetLastError();
}

hipError_t launch_wino_weights(const float *W, int cin, int cout, int transpose, float *Wt, hipStream_t stream)
{
    const long long n = (long long)cin * cout;
    wino_weight_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(W, cin, cout, transpose, Wt);
    return hipGetLastError();
}

hipError_t launch_wino_input(const float *x, int B, int H, int W, int Cs, int c_off, int C, float *V, hipStream_t stream, bool pos_major)
{
    if ((C & 3) || (Cs & 3) || (c_off & 3)) return hipErrorInvalidValue;
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    const long long per = (long long)TH * TW * (C / 4), tc = (long long)TH * TW * C;
    const dim3 grid((unsigned)((per + 255) / 256), (unsigned)B);
    if ((long long)B * 16 * tc * 4 >= VSTAB_NT_MIN_BYTES)
        wino_input_kernel<true><<<grid, dim3(256), 0, stream>>>(x, H, W, Cs, c_off, C / 4, V, TH, TW, pos_major ? tc : 16 * tc, pos_major ? B * tc : tc);
    else
        wino_input_kernel<false><<<grid, dim3(256), 0, stream>>>(x, H, W, Cs, c_off, C / 4, V, TH, TW, pos_major ? tc : 16 * tc, pos_major ? B * tc : tc);
    return hipGetLastError();
}

hipError_t launch_wino_output(const float *M, int B, int Ho, int Wo, int C, const float *bias, int act, float *out, int Cs_out, int c_off,
                              hipStream_t stream)
{
    if ((C & 3) || (Cs_out & 3) || (c_off & 3)) return hipErrorInvalidValue;
    const int TH = (Ho + 1) / 2, TW = (Wo + 1) / 2;
    const long long per = (long long)TH * TW * (C / 4);
    wino_output_kernel<<<dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, stream>>>(M, TH, TW, C / 4, bias, act, out, Ho, Wo,
                                                                                              Cs_out, c_off);
    return hipGetLastError();
}

}  // namespace vstab
