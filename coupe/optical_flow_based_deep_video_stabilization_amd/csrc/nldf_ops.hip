// Small kernels of the NLDF saliency head (NLDF.py:24-134; SURVEY.md 8a row N1, tertiary: the
// reference never instantiates it).  The convolutions run on the MFMA implicit-GEMM kernel.
#include "vstab_internal.h"

namespace vstab {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Contrast_Layer (NLDF.py:131-134): x - avg_pool3x3(VALID) of x padded by 1 with tf.pad 'SYMMETRIC'
// (for a pad of 1: the edge pixel repeated).  Reads channels [0,C) and writes [c_dst, c_dst+C) of the same
// Cs-wide pixels (the [Fea | Fea_LC] half of a concat buffer).  One thread per 4 channels.
__global__ __launch_bounds__(256) void contrast_kernel(float *__restrict__ buf, int B, int H, int W, int C4, int Cs, int c_dst)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * W * C4;
    if (idx >= total) return;
    const int c = (int)(idx % C4);
    const long long pix = idx / C4;
    const int n = (int)(pix / (H * W));
    const int rem = (int)(pix - (long long)n * H * W);
    const int y = rem / W, x = rem - y * W;
    const float *b = buf + (long long)n * H * W * Cs + c * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = min(max(y + dy, 0), H - 1);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = min(max(x + dx, 0), W - 1);
            s += *reinterpret_cast<const f32x4 *>(b + ((long long)yy * W + xx) * Cs);
        }
    }
    const f32x4 ctr = *reinterpret_cast<const f32x4 *>(b + ((long long)y * W + x) * Cs);
    f32x4 o;
    o.x = ctr.x - s.x / 9.0f; o.y = ctr.y - s.y / 9.0f; o.z = ctr.z - s.z / 9.0f; o.w = ctr.w - s.w / 9.0f;
    *reinterpret_cast<f32x4 *>(buf + (long long)pix * Cs + c_dst + c * 4) = o;
}

hipError_t launch_contrast(float *buf, int B, int H, int W, int C, int Cs, int c_dst, hipStream_t stream)
{
    if ((C & 3) || (Cs & 3) || (c_dst & 3) || c_dst < C) return hipErrorInvalidValue;
    const long long total = (long long)B * H * W * (C / 4);
    contrast_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(buf, B, H, W, C / 4, Cs, c_dst);
    return hipGetLastError();
}

// Score = Local_Score + Global_Score (broadcast over the image), Prob = softmax(Score)[..., 0] (NLDF.py:73-77)
__global__ __launch_bounds__(256) void nldf_score_kernel(const float *__restrict__ local2, const float *__restrict__ global2,
                                                         int B, int npix, float *__restrict__ score, float *__restrict__ prob)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * npix) return;
    const int n = (int)(idx / npix);
    const float s0 = local2[2 * idx] + global2[2 * n], s1 = local2[2 * idx + 1] + global2[2 * n + 1];
    if (score) { score[2 * idx] = s0; score[2 * idx + 1] = s1; }
    const float m = fmaxf(s0, s1);
    const float e0 = expf(s0 - m), e1 = expf(s1 - m);
    prob[idx] = e0 / (e0 + e1);
}

hipError_t launch_nldf_score(const float *local2, const float *global2, int B, int npix, float *score, float *prob, hipStream_t stream)
{
    const long long total = (long long)B * npix;
    nldf_score_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(local2, global2, B, npix, score, prob);
    return hipGetLastError();
}

}  // namespace vstab
