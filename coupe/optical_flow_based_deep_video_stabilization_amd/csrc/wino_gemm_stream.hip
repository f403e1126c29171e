// The 16 Winograd-domain GEMMs of a 3x3 stride-1 stage ("Winograd F(2x2,3x3)"; model.py:816-844's conv3_1 ... conv6_1) as
// STREAMS: M[xi] = V[xi] . U[xi] for the positions xi = 0..15 of the transformed 4x4 tile, V [B][16][T][C_in] (wino_input_kernel),
// U packed per position like any 1x1 filter (pack_winograd), M [B][16][T][C_out] (read by wino_output_kernel).
//
// As one workgroup per (position, 128-row tile, column tile) of the implicit-GEMM kernel a workgroup lives for 8 or 16 K-tiles between a
// prologue (first operands' memory latency) and an epilogue (LDS transpose + store burst) that together are 40 % of its life.  Here a
// workgroup keeps its 128 x 64 output tile and walks through P consecutive positions: the K-tile stream runs on across positions (the
// operand cursor jumps by one position plane, the packed weights of consecutive positions are contiguous), a finished position's 32
// accumulators are copied to registers, the next position starts with srcC = 0, and the finished tile leaves as four-byte stores
// between the MFMAs of the next position's first K-tile (tools/gen_conv_kloop.py, GemmStreamGen, documents the schedule).  The last
// position's tile goes through the LDS-transposing epilogue below.  Same K-tile order, same MFMA order per accumulator, same `+ 0.0`
// of the zero bias as the implicit-GEMM launch it replaces: bit-identical to it.
//
// Needs T % 128 == 0 (a tile never straddles two samples: row offsets are linear in the tile), C_in % 64 == 0 (an even number >= 4 of
// 32-float K-tiles, all whole), C_out % 64 == 0: the launcher checks and the plan falls back to launch_conv otherwise.
#include <hip/hip_ext.h>

#include "vstab_internal.h"

namespace vstab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#include "conv_kloop_gfx950.inc"

struct WinoGemmStreamArgs {
    const float *V, *wpk;
    float *M;
    unsigned v_bytes, w_bytes, m_bytes;
    int T, Cin, Cout, kps, P;
};

namespace {
constexpr int GS_BM = 128, GS_BN = 64;
// 48 KB of operand ring (the epilogue's [128][64] staging fits it); the request is 56 KB so that exactly TWO workgroups fit a CU: the
// launch is 512 long-lived workgroups, and three on one CU with one on another would stay that way for the whole launch
constexpr int GS_LDS = 56 * 1024;
static_assert((2 * GS_BM * 32 + 2 * GS_BN * 32) * 4 <= GS_LDS, "operand ring");
}  // namespace

__global__ __launch_bounds__(256) void wino_gemm_stream_kernel(const WinoGemmStreamArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    float *sA = reinterpret_cast<float *>(gs_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    unsigned bx_, by_, bz_;
    xcd_remap(bx_, by_, bz_);                       // column tiles of one row tile (same A rows), then position groups, then row tiles
    const int m0 = (int)bx_ * GS_BM, n0 = (int)by_ * GS_BN, xi0 = (int)bz_ * p.P;
    const int smp = m0 / p.T, t0 = m0 - smp * p.T;  // sample and first row inside its position plane
    const int wave_u = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int li = lane & 31, lh = lane >> 5, sw = (li >> 1) & 7;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)sA;
    unsigned la[4], lb[4];
    {
        const int a_row0 = (wm * 64 + li) * 32, b_row0 = (wn * 32 + li) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int chunk = ((2 * q + lh) ^ sw) * 4;
            la[q] = lds0 + (unsigned)(a_row0 + chunk) * 4u;
            lb[q] = lds0 + (unsigned)(2 * GS_BM * 32 + b_row0 + chunk) * 4u;
        }
    }
    // operand fetch: lane's 16 bytes of gathered row (tid >> 3) + 32 j, source chunk XOR-swizzled like conv_mfma.hip's LDS layout
    unsigned x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (tid >> 3) + 32 * j;
        const int c4 = ((tid & 7) ^ ((row >> 1) & 7)) * 4;
        x[j] = (unsigned)(((smp * 16) * p.T + t0 + row) * p.Cin + c4) * 4u;
    }
    const unsigned wv = (unsigned)(n0 * 32 + tid * 4) * 4u;
    const int plane = p.T * p.Cin * 4;              // bytes between the planes of two positions
    const int wstep = p.Cout * 128;                 // bytes per K-tile of the packed weights

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.V), 0, p.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wpk), 0, p.w_bytes, 0x00020000);
    {   // K-tile 0 of the first position into buffer 0
        const unsigned so = (unsigned)(xi0 * plane), sw0 = (unsigned)(xi0 * p.kps) * (unsigned)wstep;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __attribute__((address_space(3))) void *d = (__attribute__((address_space(3))) void *)(sA + (32 * j + 8 * wave_u) * 32);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, d, 16, x[j], so, 0, 0);
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            __attribute__((address_space(3))) void *d = (__attribute__((address_space(3))) void *)(sA + 2 * GS_BM * 32 + jb * 1024 + wave_u * 256);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwt, d, 16, wv, sw0 + jb * 4096, 0, 0);
        }
    }
    __syncthreads();

    const unsigned long long ain = (unsigned long long)(size_t)p.V, awt = (unsigned long long)(size_t)p.wpk, aout = (unsigned long long)(size_t)p.M;
    i32x4 din, dwt, dout;
    din.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ain);  din.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ain >> 32) & 0xffffu));
    din.z = __builtin_amdgcn_readfirstlane((int)p.v_bytes);      din.w = 0x00020000;
    dwt.x = __builtin_amdgcn_readfirstlane((int)(unsigned)awt);  dwt.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(awt >> 32) & 0xffffu));
    dwt.z = __builtin_amdgcn_readfirstlane((int)p.w_bytes);      dwt.w = 0x00020000;
    dout.x = __builtin_amdgcn_readfirstlane((int)(unsigned)aout); dout.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(aout >> 32) & 0xffffu));
    dout.z = __builtin_amdgcn_readfirstlane((int)p.m_bytes);     dout.w = 0x00020000;
    const int m_a = __builtin_amdgcn_readfirstlane((int)lds0 + wave_u * 1024);
    const int m_b = m_a + 2 * GS_BM * 128;
    const unsigned vout = (unsigned)((wm * 64 + 4 * lh) * p.Cout + n0 + wn * 32 + li) * 4u;
    int s_soff = xi0 * plane + 128, s_soffw = (xi0 * p.kps + 1) * wstep, s_kc = 1, s_npos = p.P;
    int s_obase = (((smp * 16 + xi0) * p.T + t0) * p.Cout) * 4, s_n, s_t;
    float o[32];
    asm volatile(VSTAB_GEMM_STREAM_ASM
                 : [soff] "+s"(s_soff), [soffw] "+s"(s_soffw), [kc] "+s"(s_kc), [npos] "+s"(s_npos), [obase] "+s"(s_obase), [n] "=&s"(s_n), [t] "=&s"(s_t),
                   [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5]), [o6] "=&v"(o[6]),
                   [o7] "=&v"(o[7]), [o8] "=&v"(o[8]), [o9] "=&v"(o[9]), [o10] "=&v"(o[10]), [o11] "=&v"(o[11]), [o12] "=&v"(o[12]),
                   [o13] "=&v"(o[13]), [o14] "=&v"(o[14]), [o15] "=&v"(o[15]), [o16] "=&v"(o[16]), [o17] "=&v"(o[17]), [o18] "=&v"(o[18]),
                   [o19] "=&v"(o[19]), [o20] "=&v"(o[20]), [o21] "=&v"(o[21]), [o22] "=&v"(o[22]), [o23] "=&v"(o[23]), [o24] "=&v"(o[24]),
                   [o25] "=&v"(o[25]), [o26] "=&v"(o[26]), [o27] "=&v"(o[27]), [o28] "=&v"(o[28]), [o29] "=&v"(o[29]), [o30] "=&v"(o[30]),
                   [o31] "=&v"(o[31])
                 : [la0] "v"(la[0]), [la1] "v"(la[1]), [la2] "v"(la[2]), [la3] "v"(la[3]), [lb0] "v"(lb[0]), [lb1] "v"(lb[1]), [lb2] "v"(lb[2]),
                   [lb3] "v"(lb[3]), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [wv] "v"(wv), [vout] "v"(vout),
                   [rin] "s"(din), [rwt] "s"(dwt), [dout] "s"(dout), [ma] "s"(m_a), [mb] "s"(m_b), [kps] "s"(p.kps),
                   [posjump] "s"(plane - p.kps * 128), [wstep] "s"(wstep), [npairs] "s"((p.kps - 2) / 2), [cs4] "s"(p.Cout * 4),
                   [ostep] "s"(p.T * p.Cout * 4)
                 : "memory", "scc", VSTAB_GEMM_STREAM_CLOBBERS);

    // the stream's last position: transposed through LDS (the operand ring is free: the block's last barrier is behind every wave's last
    // fragment read ... of the previous K-tile; one more barrier for the last one), 16-byte stores of whole rows
    __syncthreads();
    float *sC = sA;                                   // [128][64]
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            sC[(wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * GS_BN + wn * 32 + li] = o[mb * 16 + r] + 0.0f;
    __syncthreads();
    float *dst = p.M + ((long long)((smp * 16 + xi0 + p.P - 1) * p.T + t0)) * p.Cout + n0;
#pragma unroll 4
    for (int e = tid; e < GS_BM * (GS_BN / 4); e += 256) {
        const int row = e >> 4, c4 = (e & 15) * 4;
        *reinterpret_cast<f32x4 *>(dst + (long long)row * p.Cout + c4) = *reinterpret_cast<const f32x4 *>(sC + row * GS_BN + c4);
    }
}

hipError_t wino_gemm_stream_set_attributes()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(wino_gemm_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS);
}

// positions per workgroup for a stage of B samples x T tiles, or 0 when the stage does not qualify (launch_conv then)
int wino_gemm_stream_positions(int B, int T, int cin, int cout)
{
#ifdef VSTAB_NO_ASM_KLOOP
    return 0;                                       // A/B builds only (scripts/build_variant_lib.sh): the 16-phase launch_conv
#endif
    if (T < 128 || T % 128 || cin % 64 || cin < 128 || cout % 64) return 0;
    const long long cols = (long long)B * (T / 128) * (cout / 64);            // output tiles per position
    for (int P = 16; P >= 2; P >>= 1)
        if (cols * (16 / P) == 512) return P;                                 // exactly two workgroups per CU
    return 0;
}

// V [B][16][T][cin], wpk = the stage's packed weights (16 positions back to back, pack_winograd), M [B][16][T][cout]
hipError_t launch_wino_gemm_stream(const float *V, const float *wpk, float *M, int B, int T, int cin, int cout, int P, hipStream_t stream,
                                   hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (P < 2 || 16 % P || wino_gemm_stream_positions(B, T, cin, cout) != P) return hipErrorInvalidValue;
    const long long vb = (long long)B * 16 * T * cin * 4, mb = (long long)B * 16 * T * cout * 4, wb = 16LL * (cin / 32) * cout * 128;
    if (vb >= 0x80000000LL || mb >= 0x80000000LL || wb >= 0x80000000LL) return hipErrorInvalidValue;
    if (((uintptr_t)V | (uintptr_t)wpk | (uintptr_t)M) & 15) return hipErrorInvalidValue;
    WinoGemmStreamArgs a;
    a.V = V; a.wpk = wpk; a.M = M; a.v_bytes = (unsigned)vb; a.w_bytes = (unsigned)wb; a.m_bytes = (unsigned)mb;
    a.T = T; a.Cin = cin; a.Cout = cout; a.kps = cin / 32; a.P = P;
    const dim3 grid((unsigned)(B * (T / 128)), (unsigned)(cout / 64), (unsigned)(16 / P)), block(256);
    if (ev_start && ev_stop) hipExtLaunchKernelGGL(wino_gemm_stream_kernel, grid, block, GS_LDS, stream, ev_start, ev_stop, 0, a);
    else wino_gemm_stream_kernel<<<grid, block, GS_LDS, stream>>>(a);
    return hipGetLastError();
}

}  // namespace vstab
