// Micro-benchmark (not part of the product): does the chip hold a different clock on v_mfma_f32_16x16x4_f32 than on v_mfma_f32_32x32x2_f32
// (MI355X_MICROARCH.md 'DVFS give-back' item 7 measured this for the bf16 shapes; cdna_hip_programming.md 5.4 rule 28)?  Same output tile per
// wave (64 x 64, 64 accumulator registers), every operand re-read from LDS by ds_read_b128, RANDOM full-range operands, one or two waves
// per SIMD, >= 2 s of back-to-back launches; reports TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
//   mfma_shape <mode 0: 32x32x2 | 1: 16x16x4> <workgroups> <fill: r random | z zero>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float *src, float *out, unsigned long long *stamps, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float s = 0;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (MODE == 0) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        const float *base = lds + (lane & 31) * 32 + (lane >> 5) * 4;
        for (int it = 0; it < iters; ++it) {          // 8 k per trip: 4 ds_read_b128, 16 MFMAs of 64 cycles
            const int o = (it & 3) * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(base + o), a1 = *reinterpret_cast<const f32x4 *>(base + 4096 + o);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(base + 8192 + o), b1 = *reinterpret_cast<const f32x4 *>(base + 12288 + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[3], 0, 0, 0);
            }
        }
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    } else {
        f32x4 acc[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
        const float *base = lds + (lane & 15) * 32 + (lane >> 4) * 4;
        for (int it = 0; it < iters; it += 2) {       // 16 k per trip: 8 ds_read_b128, 64 MFMAs of 32 cycles (the work of two MODE 0 trips)
            const int o = (it & 2) * 8;
            f32x4 a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = *reinterpret_cast<const f32x4 *>(base + q * 512 + o);
                b[q] = *reinterpret_cast<const f32x4 *>(base + 8192 + q * 512 + o);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x][j], b[y][j], acc[x][y], 0, 0, 0);
        }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
    }
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0, blocks = argc > 2 ? atoi(argv[2]) : 512, iters = 6400;
    const bool zero = argc > 3 && argv[3][0] == 'z';
    std::vector<float> h(16384);
    srand(1);
    for (auto &x : h) x = zero ? 0.f : ((float)rand() / RAND_MAX - 0.5f) * 4.f;
    float *o, *src; unsigned long long *st;
    hipMalloc(&o, (size_t)blocks * 256 * 4); hipMalloc(&src, 16384 * 4); hipMalloc(&st, (size_t)blocks * 16);
    hipMemcpy(src, h.data(), 16384 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double fl = (double)blocks * 4 * iters * 16 * 4096.0;
    std::vector<double> tf;
    std::vector<unsigned long long> hs((size_t)blocks * 2);
    double clk = 0;
    for (int rep = 0; rep < 400; ++rep) {            // >= 2 s of back-to-back launches
        hipEventRecord(e0, 0);
        if (mode == 0) k<0><<<blocks, 256>>>(src, o, st, iters); else k<1><<<blocks, 256>>>(src, o, st, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 300) tf.push_back(fl / (ms * 1e-3) / 1e12);
    }
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < blocks; ++b) if (hs[2 * b + 1]) c.push_back((double)hs[2 * b] / (double)hs[2 * b + 1] * 100.0);
    std::sort(c.begin(), c.end()); std::sort(tf.begin(), tf.end());
    clk = c.empty() ? 0 : c[c.size() / 2];
    printf("%s blocks %d fill %s: median %.1f TF (%.3f of 157.3), min %.1f max %.1f; in-kernel clock %.0f MHz -> %.3f of the peak at that clock\n",
           mode == 0 ? "32x32x2" : "16x16x4", blocks, zero ? "zero" : "random", tf[tf.size() / 2], tf[tf.size() / 2] / 157.3, tf.front(), tf.back(), clk,
           tf[tf.size() / 2] / (157.3 * clk / 2400.0));
    return 0;
}
