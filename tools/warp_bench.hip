// Tuning harness for the HBM-side kernels (not part of the product): times variants of the 3-channel warp at one shape with a
// smooth synthetic flow, next to a float4 copy of the same byte count (what "achievable" means on this box at this size).
//   warp_bench [B H W iters]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
#include "../coupe/optical_flow_based_deep_video_stabilization_amd/csrc/flow_ops.hip"
#include "warp_variants.inc"
using namespace vstab;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy4_kernel(const f32x4 *__restrict__ a, f32x4 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

template <typename F> static float time_us(F f, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
}

template <int PPT, bool STAGED, bool REMAP>
static void run_variant(const char *name, const float *img, const float *flow, float *out, unsigned total, int H, int W, int iters, double bytes)
{
    const unsigned nb = (total + 256 * PPT - 1) / (256 * PPT);
    const float us = time_us([&] { warp3_kernel<false, false, PPT, STAGED, REMAP><<<dim3(nb), dim3(256)>>>(img, flow, out, nullptr, total, H, W, GlueParams{}); }, iters);
    printf("%-34s %9.2f us  %8.1f GB/s  %.3f of 8 TB/s\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
}

template <int WH, int WW, int TW, int PPT, bool REMAP, bool NT = false, bool STAGE = false>
static void run_tile(const char *name, const float *img, const float *flow, float *out, const float *ref, int B, int H, int W, int iters, double bytes)
{
    constexpr int TH = WH * (4 * PPT) / (TW / WW);
    const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
    const size_t px = (size_t)B * H * W;
    hipMemset(out, 0xff, px * 12);
    const float us = time_us([&] { warp3_tile_kernel<false, false, WH, WW, TW, PPT, REMAP, NT, STAGE><<<dim3((unsigned)(tx * ty * B)), dim3(256)>>>(img, flow, out, nullptr, B, H, W, tx, ty, GlueParams{}); }, iters);
    std::vector<float> h1(px * 3), h2(px * 3);
    hipMemcpy(h1.data(), out, px * 12, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), ref, px * 12, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < px * 3; ++i) bad += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("%-34s %9.2f us  %8.1f GB/s  %.3f of 8 TB/s  (tile %dx%d, mismatches %zu)\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0, TH, TW, bad);
}

template <int CAP>
static void run_lds(const char *name, const float *img, const float *flow, float *out, const float *ref, int B, int H, int W, int iters, double bytes)
{
    const int tx = (W + WLDS_TW - 1) / WLDS_TW, ty = (H + WLDS_TH - 1) / WLDS_TH;
    const size_t px = (size_t)B * H * W;
    hipMemset(out, 0xff, px * 12);
    const float us = time_us([&] { warp3_lds_kernel<false, false, CAP><<<dim3((unsigned)(tx * ty * B)), dim3(256)>>>(img, flow, out, nullptr, B, H, W, tx, ty, GlueParams{}); }, iters);
    std::vector<float> h1(px * 3), h2(px * 3);
    hipMemcpy(h1.data(), out, px * 12, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), ref, px * 12, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < px * 3; ++i) bad += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("%-34s %9.2f us  %8.1f GB/s  %.3f of 8 TB/s  (cap %d px, mismatches %zu)\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0, CAP, bad);
}

template <int CAPF>
static void run_pipe(const char *name, const float *img, const float *flow, float *out, const float *ref, int B, int H, int W, int iters, double bytes, int wgs)
{
    const int tx = (W + WLDS_TW - 1) / WLDS_TW, ty = (H + WLDS_TH - 1) / WLDS_TH;
    const size_t px = (size_t)B * H * W;
    hipMemset(out, 0xff, px * 12);
    const float us = time_us([&] { warp3_pipe_kernel<false, false, CAPF><<<dim3((unsigned)wgs), dim3(256)>>>(img, flow, out, nullptr, B, H, W, tx, ty, GlueParams{}); }, iters);
    std::vector<float> h1(px * 3), h2(px * 3);
    hipMemcpy(h1.data(), out, px * 12, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), ref, px * 12, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < px * 3; ++i) bad += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("%-34s %9.2f us  %8.1f GB/s  %.3f of 8 TB/s  (LDS %d KB, %d wgs, mismatches %zu)\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0, CAPF * 4 / 1024, wgs, bad);
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 16, H = argc > 2 ? atoi(argv[2]) : 1080, W = argc > 3 ? atoi(argv[3]) : 1920;
    const int iters = argc > 4 ? atoi(argv[4]) : 20;
    const char *flow_file = argc > 5 ? argv[5] : nullptr;       // raw fp32 [B,H,W,2] written by scripts/flow_stats.py
    const size_t px = (size_t)B * H * W;
    std::vector<float> himg(px * 3), hflow(px * 2);
    for (size_t i = 0; i < px * 3; ++i) himg[i] = (float)rand() / RAND_MAX;
    for (int n = 0; n < B; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t i = ((size_t)n * H + y) * W + x;
                hflow[2 * i] = 6.f * sinf(x / 97.f + n) + 3.f * cosf(y / 61.f);
                hflow[2 * i + 1] = 5.f * cosf(x / 83.f) + 4.f * sinf(y / 71.f + n);
            }
    if (flow_file) {
        FILE *fp = fopen(flow_file, "rb");
        if (!fp || fread(hflow.data(), 4, px * 2, fp) != px * 2) { printf("cannot read %s\n", flow_file); return 1; }
        fclose(fp);
        printf("flow from %s\n", flow_file);
    }
    float *img, *flow, *out, *out2;
    CK(hipMalloc(&img, px * 12)); CK(hipMalloc(&flow, px * 8)); CK(hipMalloc(&out, px * 12)); CK(hipMalloc(&out2, px * 12));
    CK(hipMemcpy(img, himg.data(), px * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(flow, hflow.data(), px * 8, hipMemcpyHostToDevice));
    const double bytes = (double)px * 32;
    printf("shape %dx%dx%d: %.1f MB algorithmic per launch\n", B, H, W, bytes / 1e6);
    {   // float4 copy moving the same number of bytes (half read, half written)
        const size_t n4 = px;      // px float4 = 16 B read + 16 B written per pixel
        float *a, *b; CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16));
        CK(hipMemset(a, 0, n4 * 16));
        for (int nb : {2048, 8192, 65536}) {
            const float us = time_us([&] { copy4_kernel<<<dim3(nb), dim3(256)>>>((const f32x4 *)a, (f32x4 *)b, n4); }, iters);
            printf("float4 copy, %6d blocks           %9.2f us  %8.1f GB/s  %.3f of 8 TB/s\n", nb, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    {
        const float us = time_us([&] { launch_warp_flow(img, flow, out, B, H, W, 3, 0); }, iters);
        printf("%-34s %9.2f us  %8.1f GB/s  %.3f of 8 TB/s\n", "product launch_warp_flow", us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
        const unsigned nb = (unsigned)((px + 255) / 256);
        const float us0 = time_us([&] { warp_flow_kernel<3><<<dim3(nb), dim3(256)>>>(img, flow, out2, B, H, W, 3); }, iters);
        printf("%-34s %9.2f us  %8.1f GB/s  %.3f of 8 TB/s\n", "round-1 kernel (1 px/thread)", us0, bytes / us0 * 1e-3, bytes / us0 * 1e-3 / 8000.0);
        std::vector<float> h1(px * 3), h2(px * 3);
        CK(hipMemcpy(h1.data(), out, px * 12, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), out2, px * 12, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < px * 3; ++i) bad += h1[i] != h2[i];
        printf("bitwise mismatches vs round-1 kernel: %zu\n", bad);
    }
    const unsigned total = (unsigned)px;
    if (getenv("WARP_BENCH_ONLY")) {           // PMC runs: the product kernel, one tile variant, the copy
        run_tile<4, 16, 32, 2, true>("tile wave4x16 blk16x32 ppt2", img, flow, out, out2, B, H, W, iters, bytes);
        run_variant<4, false, true>("ppt4 direct remap", img, flow, out, total, H, W, iters, bytes);
        return 0;
    }
    run_variant<4, true, true>("ppt4 staged remap", img, flow, out, total, H, W, iters, bytes);
    run_variant<8, true, true>("ppt8 staged remap", img, flow, out, total, H, W, iters, bytes);
    run_variant<4, true, false>("ppt4 staged noremap", img, flow, out, total, H, W, iters, bytes);
    run_variant<1, false, true>("ppt1 direct remap", img, flow, out, total, H, W, iters, bytes);
    run_variant<2, false, true>("ppt2 direct remap", img, flow, out, total, H, W, iters, bytes);
    run_variant<4, false, true>("ppt4 direct remap", img, flow, out, total, H, W, iters, bytes);
    run_variant<4, false, false>("ppt4 direct noremap", img, flow, out, total, H, W, iters, bytes);
    run_tile<4, 16, 32, 2, true, false, true>("tile 4x16 16x32 ppt2 STAGE", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 32, 2, true, true, true>("tile 4x16 16x32 ppt2 STAGE NT", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 64, 4, true, false, true>("tile 4x16 16x64 ppt4 STAGE", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 64, 4, true, true, true>("tile 4x16 16x64 ppt4 STAGE NT", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 64, 2, true, true, true>("tile 4x16 8x64 ppt2 STAGE NT", img, flow, out, out2, B, H, W, iters, bytes);
    run_pipe<9216>("pipe 36KB x1024", img, flow, out, out2, B, H, W, iters, bytes, 1024);
    run_pipe<9216>("pipe 36KB x2048", img, flow, out, out2, B, H, W, iters, bytes, 2048);
    run_pipe<7168>("pipe 28KB x1280", img, flow, out, out2, B, H, W, iters, bytes, 1280);
    run_pipe<7168>("pipe 28KB x2560", img, flow, out, out2, B, H, W, iters, bytes, 2560);
    run_pipe<6144>("pipe 24KB x1536", img, flow, out, out2, B, H, W, iters, bytes, 1536);
    run_pipe<12288>("pipe 48KB x768", img, flow, out, out2, B, H, W, iters, bytes, 768);
    run_pipe<4>("pipe, all tiles fall back x2048", img, flow, out, out2, B, H, W, iters, bytes, 2048);
    run_lds<3072>("lds window cap 3072", img, flow, out, out2, B, H, W, iters, bytes);
    run_lds<2048>("lds window cap 2048", img, flow, out, out2, B, H, W, iters, bytes);
    run_lds<4096>("lds window cap 4096", img, flow, out, out2, B, H, W, iters, bytes);
    run_lds<1>("lds kernel, all tiles fall back", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<8, 8, 16, 1, true>("tile wave8x8 blk16x16", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<8, 8, 32, 1, true>("tile wave8x8 blk8x32", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<8, 8, 32, 4, true>("tile wave8x8 blk32x32 ppt4", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<8, 8, 32, 2, true>("tile wave8x8 blk16x32 ppt2", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<8, 8, 64, 2, true>("tile wave8x8 blk8x64 ppt2", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 32, 2, true>("tile wave4x16 blk16x32 ppt2", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 64, 4, true>("tile wave4x16 blk16x64 ppt4", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 64, 2, true>("tile wave4x16 blk8x64 ppt2", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<2, 32, 64, 4, true>("tile wave2x32 blk16x64 ppt4", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<2, 32, 64, 2, true>("tile wave2x32 blk8x64 ppt2", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<16, 4, 16, 1, true>("tile wave16x4 blk16x16", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 64, 4, true, true>("tile wave4x16 blk16x64 ppt4 NT", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 32, 2, true, true>("tile wave4x16 blk16x32 ppt2 NT", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 16, 1, true>("tile wave4x16 blk16x16 ppt1", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 32, 4, true>("tile wave4x16 blk32x32 ppt4", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<4, 16, 16, 4, true>("tile wave4x16 blk64x16 ppt4", img, flow, out, out2, B, H, W, iters, bytes);
    run_tile<8, 8, 32, 2, false>("tile wave8x8 blk16x32 ppt2 noremap", img, flow, out, out2, B, H, W, iters, bytes);
    return 0;
}
