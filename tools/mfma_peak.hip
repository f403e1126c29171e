// Micro-benchmark (not part of the product): what does a bare fp32 MFMA loop reach on this box, with and without
// the ds_read_b128 operand traffic of the conv kernel?  Run under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = seed * (float)((i * 7 + threadIdx.x) % 13 - 6) * 0.01f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {seed, seed * 2, seed * 3, seed * 4}, a1 = a0 * 0.5f, b0 = a0 * 0.25f, b1 = a0 * 0.125f;
    const float *base = lds + (lane & 31) * 32 + (lane >> 5) * 4;
    if (MODE == 3) {       // inline-asm ds_read one iteration ahead, counted waits by hand (hipcc cannot re-place them)
        f32x4 n0, n1, n2, n3;
        const unsigned ab = (unsigned)(size_t)(base);           // LDS byte address (address space 3 -> 32 bit)
        auto issue = [&](int it, f32x4 &r0, f32x4 &r1, f32x4 &r2, f32x4 &r3) {
            const unsigned a = (unsigned)(__builtin_amdgcn_readfirstlane(0) + 0) + (unsigned)((it & 3) * 32);
            const unsigned addr = (unsigned)(reinterpret_cast<size_t>(base) & 0xffffffffu) + a;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16384\n\tds_read_b128 %2, %4 offset:32768\n\tds_read_b128 %3, %4 offset:49152"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr) : "memory");
        };
        (void)ab;
        issue(0, a0, a1, b0, b1);
        for (int it = 0; it < iters; it += 2) {
            issue(it + 1, n0, n1, n2, n3);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)::"memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            issue(it + 2, a0, a1, b0, b1);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3)::"memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(n0[j], n2[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(n0[j], n3[j], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(n1[j], n2[j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(n1[j], n3[j], acc[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)::"memory");
    } else if (MODE == 2) {       // software-pipelined: fragments of iteration it+1 are read before the MFMAs of iteration it
        f32x4 n0, n1, n2, n3;
        for (int it = 0; it < iters; ++it) {
            const int o = ((it + 1) & 3) * 8;
            n0 = *reinterpret_cast<const f32x4 *>(base + o);
            n1 = *reinterpret_cast<const f32x4 *>(base + 4096 + o);
            n2 = *reinterpret_cast<const f32x4 *>(base + 8192 + o);
            n3 = *reinterpret_cast<const f32x4 *>(base + 12288 + o);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[3], 0, 0, 0);
            }
            a0 = n0; a1 = n1; b0 = n2; b1 = n3;
        }
    } else
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            const int o = (it & 3) * 8;
            a0 = *reinterpret_cast<const f32x4 *>(base + o);
            a1 = *reinterpret_cast<const f32x4 *>(base + 4096 + o);
            b0 = *reinterpret_cast<const f32x4 *>(base + 8192 + o);
            b1 = *reinterpret_cast<const f32x4 *>(base + 12288 + o);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0, blocks = argc > 2 ? atoi(argv[2]) : 512, iters = 6400;
    float *o; hipMalloc(&o, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0, 0);
        if (mode == 0) k<0><<<blocks, 256>>>(o, iters, 0.37f); else if (mode == 1) k<1><<<blocks, 256>>>(o, iters, 0.37f); else if (mode == 2) k<2><<<blocks, 256>>>(o, iters, 0.37f); else k<3><<<blocks, 256>>>(o, iters, 0.37f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 4 * iters * 16 * 4096.0;
        if (rep >= 3) printf("mode %d blocks %d: %.1f us  %.1f TF (%.3f of 157.3)\n", mode, blocks, ms * 1e3, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 1e12 / 157.3);
    }
    return 0;
}
