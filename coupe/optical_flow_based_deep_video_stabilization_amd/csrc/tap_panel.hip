// predict_flow2's tap table (model.py:882-885 as restated in DESIGN.md K9): T[m][n] = sum_c concat2[m][c] * Wt[c][n] for the 18 (tap, component)
// columns of the 3x3 head, one row per quarter-resolution pixel -- a [M x 196] x [196 x 18] product, 7.8 FLOP per byte: HBM-bound.
//
// As a launch of the implicit-GEMM kernel (round 1-3: conv_mfma_kernel<256,32>) it read the 784-byte pixel rows in seven 128-byte K-tile
// pieces, one piece per loop trip with one trip of prefetch: a chain of seven memory latencies per workgroup, every piece straddling two
// cache lines (3.0 TB/s, 0.41 of the roofline in situ).  But concat2 is ONE contiguous array of M x 196 floats and the whole reduction
// (196) fits LDS, so here a workgroup takes 32 consecutive pixel rows = 25 088 contiguous bytes and requests ALL of them -- plus the 25 KB
// weight table -- in one burst of `buffer_load_dwordx4 ... lds` (1 KB per wave instruction, perfectly coalesced, 50 KB in flight per
// workgroup, three workgroups per CU), waits once, and multiplies out of the panel as it lies:
//   * A operand of v_mfma_f32_32x32x2_f32 straight from the panel: lane (i, h) reads 16 bytes of row i at floats 8q + 4h .. +3 for the
//     four k-steps (q, j) -- the K permutation conv_mfma.hip uses (k = 8q + 4h + j; B uses the same map, the sum is unchanged);
//   * the reduction is split over the four waves (7 + 6 + 6 + 6 chunks of 8 floats), the four partial 32x32 blocks are summed through
//     LDS in wave order -- deterministic;
//   * floats 196..199 of the last chunk belong to the next pixel: their lanes multiply zeros (and the packed weights of rows >= 194 are
//     zero, which also covers concat2's two padding channels).
// Rows past the end of the tensor arrive as zeros through the descriptor's range check; T rows are 32 floats (18 used) as pf2_tile_kernel
// reads them.  In situ (B=8 512x512): 34.1 -> 26.0 us, (103 + 17) MB = 4.6 TB/s; the 100 MFMAs per workgroup (N padded 18 -> 32) are 10 us
// of matrix pipe chip-wide, overlapped with the loads only across workgroups.
#include <hip/hip_ext.h>

#include "vstab_internal.h"

namespace vstab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int TP_CS = 196;                         // floats per pixel row of concat2
constexpr int TP_KPAD = 200;                       // rows of the packed weight table (25 chunks of 8)
#ifndef VSTAB_TP_ROWS
#define VSTAB_TP_ROWS 32                           // in situ B=8 512x512: 26.0 us with 32-row workgroups (three per CU), 28.7 with 64-row ones (two)
#endif
constexpr int TP_ROWS = VSTAB_TP_ROWS;             // pixel rows per workgroup
constexpr int TP_NP = TP_ROWS / 32;                // 32-row panels
constexpr int TP_W_BYTES = TP_KPAD * 32 * 4;       // 25 600
constexpr int TP_X_BYTES = TP_ROWS * TP_CS * 4;    // 50 176 = 49 x 1024
constexpr int TP_W_CHUNKS = TP_W_BYTES / 1024;     // 25 wave-sized DMA pieces
constexpr int TP_X_CHUNKS = (TP_X_BYTES + 1023) / 1024;     // 49
constexpr int TP_LDS = TP_W_BYTES + TP_X_CHUNKS * 1024;
static_assert(TP_W_BYTES % 1024 == 0, "whole wave-sized DMA pieces");
static_assert(4 * TP_NP * 32 * 32 * 4 <= TP_X_CHUNKS * 1024, "the partial blocks fit the panel buffer");
}  // namespace

__global__ __launch_bounds__(256) void tap_panel_kernel(const float *__restrict__ x, unsigned x_bytes, int M, const float *__restrict__ wp,
                                                        float *__restrict__ T)
{
    extern __shared__ __attribute__((aligned(16))) char tp_smem[];
    float *sW = reinterpret_cast<float *>(tp_smem);                   // [200][32]
    float *sX = reinterpret_cast<float *>(tp_smem + TP_W_BYTES);      // [64][196], later the partial blocks [4 waves][2 panels][32][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int m0 = blockIdx.x * TP_ROWS;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wp), 0, TP_W_BYTES, 0x00020000);
    // every request of the workgroup goes out before anything is waited for: the panel first (HBM), then the table (L2)
    const unsigned xbase = (unsigned)m0 * (unsigned)(TP_CS * 4);      // < 2^31: checked on the host
#pragma unroll
    for (int j = 0; j < (TP_X_CHUNKS + 3) / 4; ++j) {
        const int t = j * 4 + wave;
        if (t < TP_X_CHUNKS) {                                        // wave uniform
            __attribute__((address_space(3))) void *d = (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(sX) + t * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, d, 16, xbase + (unsigned)(t * 1024 + lane * 16), 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < (TP_W_CHUNKS + 3) / 4; ++j) {
        const int t = j * 4 + wave;
        if (t < TP_W_CHUNKS) {
            __attribute__((address_space(3))) void *d = (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(sW) + t * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, d, 16, (unsigned)(t * 1024 + lane * 16), 0, 0, 0);
        }
    }
    __syncthreads();                                                  // vmcnt(0) + barrier: panel and table are in LDS

    const int li = lane & 31, lh = lane >> 5;
    const int q0 = wave == 0 ? 0 : 1 + 6 * wave, q1 = 7 + 6 * wave;   // chunks of 8 floats: [0,7) [7,13) [13,19) [19,25)
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float *a0p = sX + li * TP_CS + 4 * lh;
    const float *a1p = a0p + (TP_NP > 1 ? 32 : 0) * TP_CS;
    const float *bp = sW + (4 * lh) * 32 + li;
    for (int q = q0; q < q1; ++q) {
        f32x4 a0 = *reinterpret_cast<const f32x4 *>(a0p + 8 * q);
        f32x4 a1 = *reinterpret_cast<const f32x4 *>(a1p + 8 * q);
        if (q == TP_KPAD / 8 - 1 && lh) { a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0; }        // floats 196..199: the next pixel's
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = bp[(8 * q + j) * 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b, acc0, 0, 0, 0);
            if (TP_NP > 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b, acc1, 0, 0, 0);
        }
    }
    __syncthreads();                                                  // every wave is done with the panel
    float *sR = sX;                                                   // [wave][panel][row][col]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        sR[((wave * TP_NP + 0) * 32 + row) * 32 + li] = acc0[r];
        if (TP_NP > 1) sR[((wave * TP_NP + TP_NP - 1) * 32 + row) * 32 + li] = acc1[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TP_NP; ++i) {
        const int e = tid + 256 * i;
        const int row = e >> 3, c4 = (e & 7) * 4;                     // row 0..63 of the workgroup, 16-byte column group
        f32x4 s = *reinterpret_cast<const f32x4 *>(sR + ((0 * TP_NP + (row >> 5)) * 32 + (row & 31)) * 32 + c4);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4 *>(sR + ((w * TP_NP + (row >> 5)) * 32 + (row & 31)) * 32 + c4);
        if (m0 + row < M) *reinterpret_cast<f32x4 *>(T + (size_t)(m0 + row) * 32 + c4) = s;
    }
}

hipError_t tap_panel_set_attributes()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(tap_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TP_LDS);
}

bool tap_panel_applicable(long long M, int cs_in, const void *x, const void *T)
{
    return cs_in == TP_CS && M >= 1 && M * TP_CS * 4 < 0x80000000LL && ((uintptr_t)x & 15) == 0 && ((uintptr_t)T & 15) == 0;
}

// x [M][196] (concat2), wp [200][32] (pack_predict2_panel), T [M][32]
hipError_t launch_tap_panel(const float *x, long long M, const float *wp, float *T, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (!tap_panel_applicable(M, TP_CS, x, T)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((M + TP_ROWS - 1) / TP_ROWS)), block(256);
    const unsigned x_bytes = (unsigned)(M * TP_CS * 4);
    if (ev_start && ev_stop) hipExtLaunchKernelGGL(tap_panel_kernel, grid, block, TP_LDS, stream, ev_start, ev_stop, 0, x, x_bytes, (int)M, wp, T);
    else tap_panel_kernel<<<grid, block, TP_LDS, stream>>>(x, x_bytes, (int)M, wp, T);
    return hipGetLastError();
}

}  // namespace vstab
