// Micro-test (not part of the product): does an out-of-range `buffer_load_dwordx4 ... lds` lane write
// zeros to LDS or leave the old bytes?  (decides whether LDS-DMA can replace the predicated gathers)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *p, float *o, int nbytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s = (float *)smem;
    for (int i = threadIdx.x; i < 1024; i += 256) s[i] = -7.0f;        // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, nbytes, 0x00020000);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x) >> 6;
    const unsigned off = (threadIdx.x & 1) ? 0xC0000000u : threadIdx.x * 16u;   // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(s + w * 256), 16, off, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) o[i] = s[i];
}
int main()
{
    float *p, *o, h[1024];
    hipMalloc(&p, 4096); hipMalloc(&o, 4096);
    for (int i = 0; i < 1024; ++i) h[i] = (float)(i + 1);
    hipMemcpy(p, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 256, 4096>>>(p, o, 4096);
    hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
    int ok_in = 0, zero_oob = 0, stale_oob = 0;
    for (int t = 0; t < 256; ++t)
        for (int e = 0; e < 4; ++e) {
            const float v = h[t * 4 + e];
            if (t & 1) { if (v == 0.f) ++zero_oob; else if (v == -7.f) ++stale_oob; }
            else if (v == (float)(t * 4 + e + 1)) ++ok_in;
        }
    printf("in-range correct %d/512, out-of-range zero %d/512, out-of-range stale %d/512\n", ok_in, zero_oob, stale_oob);
    return 0;
}
