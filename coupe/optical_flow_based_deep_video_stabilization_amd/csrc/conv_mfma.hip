// Implicit-GEMM convolution / transposed convolution on the gfx950 fp32 matrix cores.
//
// Replaces, for inference, the TensorLayer stacks of model.py:807-844 (PadLayer + Conv2d
// VALID + BatchNormLayer + lrelu) and model.py:850-851 etc. (DeConv2dLayer 4x4 s2 SAME +
// BatchNormLayer + lrelu): BatchNorm is folded into W/b by the packer, the activation is
// fused into the epilogue, and outputs go straight into their channel slice of the concat
// buffers (model.py:853,862,871,880 need no copy).
//
// Numerics: v_mfma_f32_32x32x2_f32 is a k-ordered chain of fp32 FMAs (exact fp32, no
// reduced-precision path exists on gfx950), so results match an fp32 CPU convolution up
// to summation order.
//
// Structure (one workgroup = 4 waves, one wave per SIMD; 2-3 workgroups per CU):
//   * GEMM rows = output pixels, cols = output channels, K = (row tap, segment, 32-float
//     chunk of the segment): a segment is a contiguous piece of the NHWC input row --
//     either the whole KW*Cs run ("run mode", x and c are adjacent in NHWC) or, when only
//     the first channels of a wider concat pixel are consumed, one tap's channels
//     ("tap mode", NSEG = KW, stride Cs);
//   * A tile (BM x 32) gathered from the input with 16-byte loads (dword loads for the
//     27-channel network input), B tile (BN x 32) is a linear 16-byte copy of the
//     pre-tiled, pre-swizzled packed weights;
//   * both land in double-buffered LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, no staging registers, no
//     ds_write; issued for tile t+1 before the MFMAs of tile t, one barrier per tile).  The 27-channel
//     dword-gather variant and the 128x32 tile still stage through registers;
//   * LDS rows are 128 B with the 16-byte chunk index XORed by ((row>>1)&7): the
//     ds_read_b128 operand reads of a wave are bank-conflict free;
//   * K order inside a tile is permuted so that ONE ds_read_b128 feeds four MFMA k-steps:
//     lane (i, h) holds k = 8q + 4h + j for step (q, j) -- A and B use the same map;
//   * split-K (grid.z) with fp32 slabs + a combine kernel for the small late layers.
#include <algorithm>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "vstab_internal.h"

namespace vstab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define VSTAB_COMMA ,
#include "conv_kloop_gfx950.inc"

#ifndef VSTAB_ABL
#define VSTAB_ABL 0        // tuning-harness ablations (tools/conv_bench): 1 no operand fetch, 2 no LDS stores, 4 no barrier;
#endif                     // never defined in the product build

#if defined(VSTAB_HARNESS) && defined(VSTAB_STAMP)
// diagnostic build of tools/conv_bench only: s_memtime / s_memrealtime of wave 0 of every workgroup at four points (entry, loop
// start, loop end, exit) go to a buffer nothing else reads (MI355X_MICROARCH.md, DVFS item 6; cdna_hip_programming.md, In-kernel stamps)
// 32 launch slots (launch_conv numbers its launches; conv_stamp_reset() restarts the count) so that a whole forward can be read back
__device__ unsigned long long g_conv_stamps[32][8 * 8192];
#define STAMP(i) do { if (threadIdx.x == 0 && sid < 8192) { g_conv_stamps[p.stamp_slot & 31][sid * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
                                                          if ((i) == 0 || (i) == 3) g_conv_stamps[p.stamp_slot & 31][sid * 8 + 4 + (i) / 3] = __builtin_amdgcn_s_memrealtime(); \
                                                          if ((i) == 0) { g_conv_stamps[p.stamp_slot & 31][sid * 8 + 6] = __builtin_amdgcn_s_getreg(63492);      /* HW_ID */ \
                                                                          g_conv_stamps[p.stamp_slot & 31][sid * 8 + 7] = __builtin_amdgcn_s_getreg(63508); } } } while (0)   /* XCC_ID */
static int g_stamp_next = 0;
void conv_stamp_reset()
{
    g_stamp_next = 0;
    void *d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_conv_stamps)) == hipSuccess) (void)hipMemset(d, 0, sizeof(unsigned long long) * 32 * 8 * 8192);
}
hipError_t conv_read_stamps_slot(int slot, unsigned long long *host, size_t n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_stamps), n * sizeof(unsigned long long), (size_t)(slot & 31) * 8 * 8192 * sizeof(unsigned long long));
}
hipError_t conv_read_stamps(unsigned long long *host, size_t n) { return conv_read_stamps_slot((g_stamp_next - 1) & 31, host, n); }
#else
#define STAMP(i) do { } while (0)
#endif

// the whole workgroup program for tile (bx_, by_, bz_) of the launch described by p; a __global__ wrapper (one launch = one
// problem, or conv_dual_kernel's two) supplies the tile coordinates
template <int BM, int BN, int WM, int WN, bool VEC, bool DMA = false>
__device__ __forceinline__ void conv_mfma_body(const ConvParams &p, const unsigned bx_, const unsigned by_, const unsigned bz_)
{
#if defined(VSTAB_HARNESS) && defined(VSTAB_STAMP)
    const unsigned sid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#endif
    STAMP(0);
    static_assert(WM * WN == 4, "four waves per workgroup");
    static_assert(!DMA || VEC, "LDS-DMA staging needs the 16-byte operand path");
    constexpr int MB = BM / WM / 32, NB = BN / WN / 32;
    static_assert(MB >= 1 && NB >= 1, "wave tile");
    constexpr int A_ROWS_V = BM / 32;   // float4 loads per thread per tile (VEC)
    constexpr int A_ELEMS_S = BM / 8;   // dword loads per thread per tile (!VEC)
    constexpr int B_PASS = BN / 32;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *sA = reinterpret_cast<float *>(smem);
    float *sB = sA + 2 * BM * 32;
    int4 *rinfo = reinterpret_cast<int4 *>(sB + 2 * BN * 32);
    int *ooff = reinterpret_cast<int *>(rinfo + BM);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int z = (int)bz_;
    const int phase = z / p.ksplit, split = z - phase * p.ksplit;
    const ConvPhase ph = p.ph[phase];
    const int m0 = (int)bx_ * BM;
    if (m0 >= ph.M) return;                       // uniform for the whole workgroup
    const int n0 = (int)by_ * BN;

    // ---- per-row gather table: {input element offset, iy0, q_lo, q_hi} and output offset
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        int4 ri = make_int4(0, -(1 << 28), 0, 0);
        int oo = -1;
        if (m < ph.M) {
            const int hw = ph.Hg * ph.Wg;
            const int n = m / hw, rem = m - n * hw;
            const int j = rem / ph.Wg, i = rem - j * ph.Wg;
            const int iy0 = j * p.s_in + ph.off_y, ix0 = i * p.s_in + ph.off_x;
            ri.x = ((n * p.Hi + iy0) * p.Wi + ix0) * p.Cs_in;
            ri.y = iy0;
            ri.z = -ix0 * p.Cs_in;
            ri.w = (p.Wi - ix0) * p.Cs_in;
            oo = ((n * p.Ho + j * p.s_out + ph.o_y) * p.Wo + i * p.s_out + ph.o_x) * p.Cs_out + p.c_off;
        }
        rinfo[r] = ri;
        ooff[r] = oo;
    }
    __syncthreads();

    // this phase's K layout (its own, or the launch-wide one)
    const bool own = ph.KH != 0;
    const int L_KH = own ? ph.KH : p.KH, L_NSEG = own ? ph.NSEG : p.NSEG, L_SEG = own ? ph.SEG : p.SEG;
    const int L_SEGP = own ? ph.SEGP : p.SEGP, L_STRIDE = own ? ph.SEG_STRIDE : p.SEG_STRIDE;
    const int kps = L_SEGP >> 5;                  // K-tiles per segment
    const int KT = L_KH * L_NSEG * kps;
    const int kts = (KT + p.ksplit - 1) / p.ksplit;
    const int kt0 = split * kts;
    const int kt1 = min(KT, kt0 + kts);

    // ---- staging registers
    f32x4 ra[VEC ? A_ROWS_V : 1];
    float ras[VEC ? 1 : A_ELEMS_S];
    f32x4 rb[B_PASS];
    int4 R[VEC ? A_ROWS_V : 1];
    if constexpr (VEC) {
#pragma unroll
        for (int j = 0; j < A_ROWS_V; ++j) R[j] = rinfo[(tid >> 3) + 32 * j];
    }
    // Operands are fetched with raw buffer loads: an out-of-image tap, the padded tail of a segment
    // or a tile past the end of this split simply gets an offset beyond the descriptor's range and
    // the hardware returns zeros -- the loop body has no branches and the compiler can interleave
    // the address arithmetic and the loads with the MFMAs.
    const int row_pitch = p.Wi * p.Cs_in;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wpk + ph.w_off), 0, p.w_bytes, 0x00020000);
    const unsigned OOB = 0xC0000000u;                 // every tensor is < 2^31 bytes (checked on the host)
    const unsigned wvoff0 = (unsigned)(n0 * 32 + tid * 4) * 4u;

    // K-tile cursor (row tap, segment, chunk) of the NEXT tile to load
    int c_ky, c_sg, c_kc;
    {
        const int per_row = L_NSEG * kps;
        c_ky = kt0 / per_row;
        const int r = kt0 - c_ky * per_row;
        c_sg = r / kps;
        c_kc = r - c_sg * kps;
    }

    auto load_tile = [&](int kt) {
        const int ky = c_ky;
        const int qseg = c_kc * 32;                       // float offset inside the segment
        const int qabs0 = c_sg * L_STRIDE + qseg;     // float offset from the run start
        {   // advance the cursor without branches (keeps the loop body one scheduling region)
            const int kc1 = c_kc + 1;
            const bool wrap_kc = kc1 == kps;
            const int sg1 = c_sg + (wrap_kc ? 1 : 0);
            const bool wrap_sg = sg1 == L_NSEG;
            c_kc = wrap_kc ? 0 : kc1;
            c_sg = wrap_sg ? 0 : sg1;
            c_ky += wrap_sg ? 1 : 0;
        }
        if constexpr (VEC) {
            const int qs = qseg + (tid & 7) * 4;
            const int qa = qabs0 + (tid & 7) * 4;
            const int rowoff = ky * row_pitch + qa;
            const bool segok = qs < L_SEG;
#pragma unroll
            for (int j = 0; j < A_ROWS_V; ++j) {
                const bool ok = segok & ((unsigned)(R[j].y + ky) < (unsigned)p.Hi) & (qa >= R[j].z) & (qa < R[j].w);
                const unsigned off = ok ? (unsigned)(R[j].x + rowoff) * 4u : OOB;
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0));
            }
        } else {
            const int qs = qseg + (tid & 31);
            const int qa = qabs0 + (tid & 31);
            const int rowoff = ky * row_pitch + qa;
            const bool segok = qs < L_SEG;
#pragma unroll
            for (int j = 0; j < A_ELEMS_S; ++j) {
                const int4 ri = rinfo[(tid >> 5) + 8 * j];
                const bool ok = segok & ((unsigned)(ri.y + ky) < (unsigned)p.Hi) & (qa >= ri.z) & (qa < ri.w);
                const unsigned off = ok ? (unsigned)(ri.x + rowoff) * 4u : OOB;
                ras[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, off, 0, 0));
            }
        }
        const unsigned woff = wvoff0 + (unsigned)kt * (unsigned)(p.Npad * 128);
#pragma unroll
        for (int jb = 0; jb < B_PASS; ++jb)
            rb[jb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rwt, woff + jb * 4096, 0, 0));
    };

    auto store_tile = [&](int buf) {
        float *dA = sA + buf * (BM * 32);
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < A_ROWS_V; ++j) {
                const int row = (tid >> 3) + 32 * j;
                const int chunk = (tid & 7) ^ ((row >> 1) & 7);
                *reinterpret_cast<f32x4 *>(dA + row * 32 + chunk * 4) = ra[j];
            }
        } else {
            const int k = tid & 31;
#pragma unroll
            for (int j = 0; j < A_ELEMS_S; ++j) {
                const int row = (tid >> 5) + 8 * j;
                dA[row * 32 + ((((k >> 2) ^ ((row >> 1) & 7)) << 2) | (k & 3))] = ras[j];
            }
        }
        float *dB = sB + buf * (BN * 32) + tid * 4;
#pragma unroll
        for (int jb = 0; jb < B_PASS; ++jb) *reinterpret_cast<f32x4 *>(dB + jb * 1024) = rb[jb];
    };

    // LDS-DMA staging (DMA = true): `buffer_load_dwordx4 ... lds` writes the tile straight into LDS -- no
    // staging registers, no ds_write.  One wave instruction lands 64 x 16 B contiguously (8 rows x 128 B) at
    // a wave-uniform base, so the XOR swizzle moves to the SOURCE: the lane that fills LDS chunk c' of a row
    // fetches global chunk c' ^ ((row>>1)&7).  Out-of-range lanes (offset 0xC0000000) land as zeros.
    //
    // Address arithmetic is the one thing this path still spends VALU issue slots on (measured: the fetch's address
    // math costs ~3.5 % of the kernel, its memory traffic nothing), so everything that does not change from tile to
    // tile is folded into per-thread constants up front: the lane's byte offset inside the input (relative to a
    // descriptor whose base is moved back by the padding, so it is never negative), the three comparison limits of
    // its validity test.  The tile's own displacement (ky, segment, chunk) is wave
    // uniform and rides in the instruction's scalar offset.
    const int wave_u = __builtin_amdgcn_readfirstlane(tid) >> 6;
    int4 RA[VEC ? A_ROWS_V : 1];                  // {byte offset, x-range low, x-range high, segment limit} per gathered row
    const int bias_el = ((ph.off_y < 0 ? -ph.off_y : 0) * p.Wi + (ph.off_x < 0 ? -ph.off_x : 0)) * p.Cs_in;      // elements the base moves back
    const __amdgpu_buffer_rsrc_t rin_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in) - bias_el, 0, p.in_bytes + 4u * (unsigned)bias_el, 0x00020000);
    if constexpr (DMA) {
#pragma unroll
        for (int j = 0; j < A_ROWS_V; ++j) {
            const int row = (tid >> 3) + 32 * j;
            const int c4 = ((tid & 7) ^ ((row >> 1) & 7)) * 4;            // source chunk (floats) for this LDS slot
            RA[j] = make_int4((R[j].x + c4 + bias_el) * 4, R[j].z - c4, R[j].w - c4, L_SEG - c4);
        }
    }
    auto dma_tile = [&](int kt, int buf) {
        const int ky = c_ky;
        const int qseg = c_kc * 32;
        const int qabs0 = c_sg * L_STRIDE + qseg;
        {
            const int kc1 = c_kc + 1;
            const bool wrap_kc = kc1 == kps;
            const int sg1 = c_sg + (wrap_kc ? 1 : 0);
            const bool wrap_sg = sg1 == L_NSEG;
            c_kc = wrap_kc ? 0 : kc1;
            c_sg = wrap_sg ? 0 : sg1;
            c_ky += wrap_sg ? 1 : 0;
        }
        if constexpr (VEC) {
            const unsigned soff = (unsigned)(ky * row_pitch + qabs0) * 4u;       // wave uniform
#pragma unroll
            for (int j = 0; j < A_ROWS_V; ++j) {
                const bool ok = (qseg < RA[j].w) & ((unsigned)(R[j].y + ky) < (unsigned)p.Hi) & (qabs0 >= RA[j].y) & (qabs0 < RA[j].z);
                const unsigned off = ok ? (unsigned)RA[j].x : OOB;
                __attribute__((address_space(3))) void *dst =
                    (__attribute__((address_space(3))) void *)(sA + buf * (BM * 32) + (32 * j + 8 * wave_u) * 32);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin_b, dst, 16, off, soff, 0, 0);
            }
        }
        const unsigned soffw = (unsigned)kt * (unsigned)(p.Npad * 128);          // wave uniform, like the 4 KB pass stride
#pragma unroll
        for (int jb = 0; jb < B_PASS; ++jb) {
            __attribute__((address_space(3))) void *dst =
                (__attribute__((address_space(3))) void *)(sB + buf * (BN * 32) + jb * 1024 + wave_u * 256);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwt, dst, 16, wvoff0, soffw + jb * 4096, 0, 0);
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;
    const int a_row0 = (wm * MB * 32 + li) * 32;
    const int b_row0 = (wn * NB * 32 + li) * 32;

    // operand fragments of one k-group (8 k): two named sets so that the LDS reads of group g+1
    // are in flight while the 4*MB*NB MFMAs of group g issue -- also across the tile barrier
    f32x4 fa0[MB], fb0[NB], fa1[MB], fb1[NB];
    auto rd = [&](int buf, int q, f32x4 (&a)[MB], f32x4 (&b)[NB]) {
        const float *cA = sA + buf * (BM * 32) + a_row0;
        const float *cB = sB + buf * (BN * 32) + b_row0;
        const int chunk = ((2 * q + lh) ^ sw) * 4;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[mb] = *reinterpret_cast<const f32x4 *>(cA + mb * 1024 + chunk);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const f32x4 *>(cB + nb * 1024 + chunk);
    };
    auto mm = [&](const f32x4 (&a)[MB], const f32x4 (&b)[NB]) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][jj], b[nb][jj], acc[mb][nb], 0, 0, 0);
    };

    // Issue order of one steady-state tile, pinned with sched_group_barrier (hipcc left alone sinks
    // the weight loads to their ds_write and waits for the L2 round trip, and issues the fragment
    // reads only after the previous group's MFMAs):
    //   k-group 0: reads of group 1, then one global load (+ its address arithmetic) per MFMA
    //   k-group 1: reads of group 2, MFMAs
    //   k-group 2: reads of group 3, MFMAs with the LDS writes of tile t+1 between them
    //   barrier;   reads of group 0 of tile t+1, MFMAs of group 3
    constexpr int G = 4 * MB * NB;                                    // MFMAs per k-group
    constexpr int NRD = MB + NB;                                      // ds_read_b128 per k-group
    constexpr int NLD = (VEC ? A_ROWS_V : A_ELEMS_S) + B_PASS;        // global loads per tile
    constexpr int NST = NLD;                                          // LDS writes per tile
    constexpr bool PIN = (NLD <= G) && (NST <= G);
#ifdef VSTAB_NO_ASM_KLOOP
    constexpr bool ASM_KLOOP = false;                                 // A/B builds only (scripts/build_variant_lib.sh)
#else
#ifdef VSTAB_NO_ASM_KLOOP_64
    constexpr bool ASM_64 = false;                                    // A/B builds only: the 64 x 128 tile on hipcc's loop (rounds 1-4)
#else
    constexpr bool ASM_64 = BM == 64 && BN == 128 && WM == 1 && WN == 4;
#endif
    constexpr bool ASM_KLOOP = DMA && VEC && VSTAB_ABL == 0 && ((BM == 128 && WM == 2 && WN == 2 && (BN == 128 || BN == 64)) || ASM_64);
#endif

    if constexpr (ASM_KLOOP) {
        // The whole K loop as one assembly block (conv_kloop_gfx950.inc, written by tools/gen_conv_kloop.py, which documents the
        // schedule): same tiles, same buffers, same MFMA order per accumulator as the C++ loop below -- bit-identical results --
        // with every fragment read issued a full k-group ahead of its use, the fetch of tile t+1 (validity test per gathered row
        // by EXEC narrowing, `buffer_load_dwordx4 ... lds`) between the MFMAs of k-group 0, one barrier per tile inside k-group 3.
        if (kt0 < kt1) {
            dma_tile(kt0, 0);                      // leaves the cursor on tile kt0 + 1
            __syncthreads();
            STAMP(1);
            const unsigned ldsA = (unsigned)(size_t)(__attribute__((address_space(3))) void *)sA;
            unsigned la[4], lb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int chunk = ((2 * q + lh) ^ sw) * 4;
                la[q] = ldsA + (unsigned)(a_row0 + chunk) * 4u;
                lb[q] = ldsA + (unsigned)(2 * BM * 32 + b_row0 + chunk) * 4u;
            }
            // per gathered row of this thread (four for the 128-row tiles, two for the 64-row one: the rest are unused operands)
            int span[4] = {0, 0, 0, 0}, ax[4] = {0, 0, 0, 0}, alo[4] = {0, 0, 0, 0}, aw[4] = {0, 0, 0, 0}, ay[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < A_ROWS_V; ++j) {
                span[j] = max(RA[j].z - RA[j].y, 0);
                ax[j] = RA[j].x; alo[j] = RA[j].y; aw[j] = RA[j].w; ay[j] = R[j].y;
            }
            const unsigned long long ain = (unsigned long long)(size_t)(p.in - bias_el);
            const unsigned long long awt = (unsigned long long)(size_t)(p.wpk + ph.w_off);
            i32x4 din, dwt;                        // the two buffer descriptors (stride 0, raw, the ranges of rin_b / rwt)
            din.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ain);
            din.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ain >> 32) & 0xffffu));
            din.z = __builtin_amdgcn_readfirstlane((int)(p.in_bytes + 4u * (unsigned)bias_el));
            din.w = 0x00020000;
            dwt.x = __builtin_amdgcn_readfirstlane((int)(unsigned)awt);
            dwt.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(awt >> 32) & 0xffffu));
            dwt.z = __builtin_amdgcn_readfirstlane((int)p.w_bytes);
            dwt.w = 0x00020000;
            const int m_a = __builtin_amdgcn_readfirstlane((int)ldsA + wave_u * 1024);
            const int m_b = __builtin_amdgcn_readfirstlane((int)ldsA + 2 * BM * 128 + wave_u * 1024);
            const int wstep = p.Npad * 128;
            int s_ky = c_ky, s_sg = c_sg, s_kc = c_kc, s_soffw = (kt0 + 1) * wstep, s_n = kt1 - kt0 - 1;
            int t_qseg, t_qabs, t_soff, t_t, t_kyc;
            unsigned v_t, v_o0, v_o1;
#define VSTAB_KLOOP_IO(ACCS)                                                                                                                \
            : ACCS, [ky] "+s"(s_ky), [sg] "+s"(s_sg), [kc] "+s"(s_kc), [soffw] "+s"(s_soffw), [n] "+s"(s_n),                              \
              [qseg] "=&s"(t_qseg), [qabs] "=&s"(t_qabs), [soff] "=&s"(t_soff), [t] "=&s"(t_t), [kyc] "=&s"(t_kyc),                        \
              [vt] "=&v"(v_t), [vo0] "=&v"(v_o0), [vo1] "=&v"(v_o1)                                                                        \
            : [la0] "v"(la[0]), [la1] "v"(la[1]), [la2] "v"(la[2]), [la3] "v"(la[3]),                                                     \
              [lb0] "v"(lb[0]), [lb1] "v"(lb[1]), [lb2] "v"(lb[2]), [lb3] "v"(lb[3]),                                                     \
              [x0] "v"(ax[0]), [x1] "v"(ax[1]), [x2] "v"(ax[2]), [x3] "v"(ax[3]),                                                         \
              [lo0] "v"(alo[0]), [lo1] "v"(alo[1]), [lo2] "v"(alo[2]), [lo3] "v"(alo[3]),                                                 \
              [span0] "v"(span[0]), [span1] "v"(span[1]), [span2] "v"(span[2]), [span3] "v"(span[3]),                                     \
              [w0] "v"(aw[0]), [w1] "v"(aw[1]), [w2] "v"(aw[2]), [w3] "v"(aw[3]),                                                         \
              [y0] "v"(ay[0]), [y1] "v"(ay[1]), [y2] "v"(ay[2]), [y3] "v"(ay[3]),                                                         \
              [wv] "v"(wvoff0), [rin] "s"(din), [rwt] "s"(dwt), [ma] "s"(m_a), [mb] "s"(m_b),                                            \
              [kps] "s"(kps), [nseg] "s"(L_NSEG), [lstride] "s"(L_STRIDE), [pitch] "s"(row_pitch), [hi] "s"(p.Hi), [wstep] "s"(wstep)     \
            : "memory", "vcc", "scc", VSTAB_KLOOP_CLOBBERS
            if constexpr (BM == 64) {
                asm volatile(VSTAB_KLOOP_ASM_64x128
                             VSTAB_KLOOP_IO([c00] "+a"(acc[0][0]) VSTAB_COMMA [c10] "+a"(acc[1][0])));
            } else if constexpr (BN == 128) {
                asm volatile(VSTAB_KLOOP_ASM_128x128
                             VSTAB_KLOOP_IO([c00] "+a"(acc[0][0]) VSTAB_COMMA [c01] "+a"(acc[0][1]) VSTAB_COMMA [c10] "+a"(acc[1][0]) VSTAB_COMMA [c11] "+a"(acc[1][1])));
            } else {
                asm volatile(VSTAB_KLOOP_ASM_128x64
                             VSTAB_KLOOP_IO([c00] "+a"(acc[0][0]) VSTAB_COMMA [c10] "+a"(acc[1][0])));
            }
#undef VSTAB_KLOOP_IO
            STAMP(2);
        }
    } else if constexpr (DMA) {
        if (kt0 < kt1) {
            dma_tile(kt0, 0);
            __syncthreads();                       // waits vmcnt(0) for the DMA, then the barrier
            int buf = 0;
            rd(0, 0, fa0, fb0);
            STAMP(1);
            for (int kt = kt0; kt + 1 < kt1; ++kt) {
                if (!(VSTAB_ABL & 1)) dma_tile(kt + 1, buf ^ 1);         // lands in the idle buffer while this tile computes
                rd(buf, 1, fa1, fb1);
                mm(fa0, fb0);
                rd(buf, 2, fa0, fb0);
                mm(fa1, fb1);
                rd(buf, 3, fa1, fb1);
                mm(fa0, fb0);
                if (!(VSTAB_ABL & 4)) __syncthreads();
                rd(buf ^ 1, 0, fa0, fb0);
                mm(fa1, fb1);
                buf ^= 1;
            }
            rd(buf, 1, fa1, fb1);
            mm(fa0, fb0);
            rd(buf, 2, fa0, fb0);
            mm(fa1, fb1);
            rd(buf, 3, fa1, fb1);
            mm(fa0, fb0);
            mm(fa1, fb1);
            STAMP(2);
        }
    } else if (kt0 < kt1) {
        load_tile(kt0);
        store_tile(0);
        __syncthreads();
        int buf = 0;
        rd(0, 0, fa0, fb0);
        for (int kt = kt0; kt + 1 < kt1; ++kt) {
            if (!(VSTAB_ABL & 1)) load_tile(kt + 1);                 // global -> registers (tile t+1)
            rd(buf, 1, fa1, fb1);
            mm(fa0, fb0);                      // k-group 0
            rd(buf, 2, fa0, fb0);
            mm(fa1, fb1);                      // k-group 1
            rd(buf, 3, fa1, fb1);
            mm(fa0, fb0);                      // k-group 2
            if (!(VSTAB_ABL & 2)) store_tile(buf ^ 1);               // registers -> the other LDS buffer
            if constexpr (PIN && VSTAB_ABL == 0) {
                __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, G - NLD, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, G - NST, 0);
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            if (!(VSTAB_ABL & 4)) __syncthreads();
            rd(buf ^ 1, 0, fa0, fb0);          // first fragments of tile t+1 ...
            mm(fa1, fb1);                      // ... fly under k-group 3 of tile t
            if constexpr (PIN) {
                __builtin_amdgcn_sched_group_barrier(0x100, NRD, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, G, 1);
            }
            buf ^= 1;
        }
        // last tile of this split: nothing left to fetch
        rd(buf, 1, fa1, fb1);
        mm(fa0, fb0);
        rd(buf, 2, fa0, fb0);
        mm(fa1, fb1);
        rd(buf, 3, fa1, fb1);
        mm(fa0, fb0);
        mm(fa1, fb1);
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5): a lane's 16 registers are 16
    // ROWS of one column.  Stored as they stand that is 64 four-byte store instructions per wave and tile, and the tile's end is
    // store-ISSUE bound: in-kernel stamps (tools/conv_bench -DVSTAB_STAMP, profiles/README.md "r03 stamps") put it at 23.5 k
    // cycles for a 128x128 tile -- 5-6 % of a layer at one workgroup per CU, with the matrix pipe idle.  So the tile is
    // transposed through LDS (the operand buffers are free by now and exactly as large as the tile) and leaves as 16-byte
    // stores of whole rows.  Same values: bias is added on the way in, the activation applied on the way out.
    const bool to_slab = p.ksplit > 1;
    if (to_slab || p.out_vec4) {
        static_assert(BM * BN <= 2 * (BM + BN) * 32, "the output tile fits the operand buffers");
        __syncthreads();                              // every wave has read its last fragments
        float *sC = sA;                               // [BM][BN]
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int cl = (wn * NB + nb) * 32 + li;
            const float bv = (!to_slab && n0 + cl < p.N) ? p.bias[n0 + cl] : 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    sC[row * BN + cl] = to_slab ? acc[mb][nb][r] : acc[mb][nb][r] + bv;
                }
            }
        }
        __syncthreads();
        constexpr int C4 = BN / 4;
        float *pz = p.partial + ((long long)z * p.Mmax + m0) * p.Npad;
        const float slope = p.act == 1 ? 0.1f : 0.0f;
#pragma unroll 4
        for (int e = tid; e < BM * C4; e += 256) {
            const int row = e / C4, c4 = e - row * C4;
            const int col = n0 + c4 * 4;
            f32x4 v = *reinterpret_cast<const f32x4 *>(sC + row * BN + c4 * 4);
            if (to_slab) {
                if (m0 + row < ph.M) *reinterpret_cast<f32x4 *>(pz + (long long)row * p.Npad + col) = v;
                continue;
            }
            const int oo = ooff[row];
            if (oo < 0 || col >= p.N) continue;
            float *o = p.out + (long long)oo + col;
            if (col + 4 <= p.N) {
                if (p.act == 3) v += *reinterpret_cast<const f32x4 *>(o);                 // accumulate (gradient sums)
                else if (p.act) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], slope * v[i]);
                }
                *reinterpret_cast<f32x4 *>(o) = v;
            } else {
                for (int i = 0; col + i < p.N; ++i) {
                    float x = v[i];
                    if (p.act == 3) x += o[i];
                    else if (p.act) x = fmaxf(x, slope * x);
                    o[i] = x;
                }
            }
        }
    } else {                                          // an output that is not 16-byte friendly (arbitrary caller tensors): 4-byte stores
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = n0 + (wn * NB + nb) * 32 + li;
            const bool cok = col < p.N;
            const float bv = cok ? p.bias[col] : 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int oo = ooff[row];
                    if (cok && oo >= 0) {
                        float v = acc[mb][nb][r] + bv;
                        if (p.act == 3) v += p.out[(long long)oo + col];                 // accumulate (gradient sums)
                        else if (p.act) v = fmaxf(v, (p.act == 1 ? 0.1f : 0.0f) * v);
                        p.out[(long long)oo + col] = v;
                    }
                }
            }
        }
    }
    STAMP(3);
}

template <int BM, int BN, int WM, int WN, bool VEC, bool DMA = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p)
{
    unsigned bx_, by_, bz_;
    xcd_remap(bx_, by_, bz_, p.no_remap != 0);      // each XCD works on a contiguous band of tiles (shared halos stay in its L2)
    conv_mfma_body<BM, BN, WM, WN, VEC, DMA>(p, bx_, by_, bz_);
}

// Two independent problems in ONE launch ("horizontal fusion"): the first nA workgroups run problem A with its tile shape, the rest
// problem B with its own.  Used for a refinement level's transposed convolution and the tap-table GEMM of the flow head that reads
// the same tensor (model.py:850 and :847-848 both consume conv6_1; :859 / :855-856 concat5; ...): neither fills the chip for one
// sample, both are a launch's fixed latency, and one launch instead of two lets them share it (and the input's L2 lines).
template <int BM1, int BN1, int WM1, int WN1, int BM2, int BN2, int WM2, int WN2>
__global__ __launch_bounds__(256) void conv_dual_kernel(const ConvParams pa, const ConvParams pb, const unsigned nA, const unsigned nB, const int b_first,
                                                        const uint3 gA, const uint3 gB)
{
    // Dispatch order.  Both problems co-resident (one sample): A first -- its tiles keep the XCD banding of a launch of their own
    // (measured: B first costs a 384x512 frame +10 us).  A alone fills whole rounds (B=8 512x512: deconv2's 1024 workgroups are two
    // rounds exactly): B's short workgroups first -- they delay half the slots by their own short life instead of forming a ragged
    // last round of their own (headline shape 2.858 -> 2.839 ms).
    // (B first: B's workgroups are padded to a multiple of 8 -- the pad exits at once -- so that A's ids keep their residue mod 8, i.e. the
    // XCD the hardware deals them to is the one xcd_remap_calc assumes and A's row-tile bands stay whole per L2)
    unsigned bx_, by_, bz_;
    const unsigned nBp = (nB + 7u) & ~7u;
    const unsigned idA = b_first ? blockIdx.x - nBp : blockIdx.x;
    const bool isA = b_first ? blockIdx.x >= nBp : blockIdx.x < nA;     // workgroup uniform
    if (b_first && !isA && blockIdx.x >= nB) return;                    // the pad
    if (isA) {
        xcd_remap_calc(gA.x, gA.y, gA.z, idA, bx_, by_, bz_);
        conv_mfma_body<BM1, BN1, WM1, WN1, true, true>(pa, bx_, by_, bz_);
    } else {
        xcd_remap_calc(gB.x, gB.y, gB.z, b_first ? blockIdx.x : blockIdx.x - nA, bx_, by_, bz_);
        conv_mfma_body<BM2, BN2, WM2, WN2, true, true>(pb, bx_, by_, bz_);
    }
}

// Sum the split-K slabs, add bias, activate, scatter to the output tensor.
__global__ __launch_bounds__(256) void splitk_combine_kernel(const ConvParams p)
{
    splitk_combine_item(p, (long long)blockIdx.x * 256 + threadIdx.x);
}

constexpr int CONV_CUS = 256;                           // MI355X: 8 XCDs x 32 CUs
constexpr size_t CONV_LDS_TWO_PER_CU = 56 * 1024;       // an LDS request of which only two fit a CU's 160 KB

template <int BM, int BN>
static size_t conv_lds_bytes()
{
    size_t n = (size_t)(2 * BM * 32 + 2 * BN * 32) * 4 + (size_t)BM * 20;
#ifdef VSTAB_HARNESS
    if (const char *e = getenv("VSTAB_LDS_PAD")) n += (size_t)atoi(e);     // occupancy experiments (tools/conv_bench only)
#endif
    return n;
}

hipError_t conv_set_attributes()
{
    hipError_t e;
#define VSTAB_SET(K, BM, BN)                                                                           \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)conv_lds_bytes<BM, BN>());                                            \
    if (e != hipSuccess) return e;
#ifdef VSTAB_HARNESS      // register-staged forms of the 16-byte-gather kernels: A/B material of tools/conv_bench, not in the product library
    VSTAB_SET((conv_mfma_kernel<128, 128, 2, 2, true>), 128, 128)
    VSTAB_SET((conv_mfma_kernel<128, 64, 2, 2, true>), 128, 64)
    VSTAB_SET((conv_mfma_kernel<128, 32, 4, 1, true>), 128, 32)
#endif
    VSTAB_SET((conv_mfma_kernel<128, 64, 2, 2, false>), 128, 64)
    VSTAB_SET((conv_mfma_kernel<128, 128, 2, 2, true, true>), 128, 128)
    VSTAB_SET((conv_mfma_kernel<128, 64, 2, 2, true, true>), 128, 64)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mfma_kernel<128, 64, 2, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)CONV_LDS_TWO_PER_CU);
    if (e != hipSuccess) return e;
    VSTAB_SET((conv_mfma_kernel<128, 32, 4, 1, true, true>), 128, 32)
    VSTAB_SET((conv_mfma_kernel<64, 128, 1, 4, true, true>), 64, 128)
#ifdef VSTAB_HARNESS
    VSTAB_SET((conv_mfma_kernel<256, 32, 4, 1, true, true>), 256, 32)
    VSTAB_SET((conv_mfma_kernel<64, 64, 2, 2, true, true>), 64, 64)
#endif
#undef VSTAB_SET
    // the two-problem launches: a refinement level's transposed convolution beside the 128x32 tap-table GEMM of its flow head
#define VSTAB_SET2(BM, BN, WM, WN)                                                                                                 \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dual_kernel<BM, BN, WM, WN, 128, 32, 4, 1>),                        \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(CONV_LDS_TWO_PER_CU, conv_lds_bytes<BM, BN>())); \
    if (e != hipSuccess) return e;
    VSTAB_SET2(128, 128, 2, 2)
    VSTAB_SET2(128, 64, 2, 2)
    VSTAB_SET2(64, 128, 1, 4)
#undef VSTAB_SET2
    return hipSuccess;
}

static bool lds_dma_enabled()
{
#ifdef VSTAB_HARNESS
    static const bool on = getenv("VSTAB_NO_LDS_DMA") == nullptr;     // A/B switch of tools/conv_bench
    return on;
#else
    return true;          // the product library has one operand-staging path per tile: LDS-DMA
#endif
}

bool conv_uses_lds_dma(ConvTile tile, bool vec4)
{
    return vec4 && lds_dma_enabled();
}

hipError_t launch_conv(const ConvParams &p_in, ConvTile tile, bool vec4, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop,
                       bool combine)
{
    ConvParams p = p_in;
#ifdef VSTAB_HARNESS
    static const int no_remap = getenv("VSTAB_NO_XCD_REMAP") != nullptr;      // A/B switch of tools/conv_bench
    p.no_remap = no_remap;
#else
    p.no_remap = 0;
#endif
#if defined(VSTAB_HARNESS) && defined(VSTAB_STAMP)
    p.stamp_slot = g_stamp_next++;
#endif
    p.out_vec4 = (((uintptr_t)p.out & 15) == 0 && (p.Cs_out & 3) == 0 && (p.c_off & 3) == 0) ? 1 : 0;
#ifdef VSTAB_HARNESS
    if (getenv("VSTAB_NO_VEC_EPILOGUE")) p.out_vec4 = 0;                                 // A/B switch of tools/conv_bench
#endif
    const int BM = (tile == TILE_64x128 || tile == TILE_64x64) ? 64 : (tile == TILE_256x32 ? 256 : 128);
    const int BN = (tile == TILE_128x128 || tile == TILE_64x128) ? 128 : ((tile == TILE_128x64 || tile == TILE_64x64) ? 64 : 32);
    if (p.in_bytes >= 0x80000000u || p.w_bytes >= 0x80000000u) return hipErrorInvalidValue;
    if (p.Npad % BN != 0 || p.SEGP % 32 != 0 || p.SEGP < p.SEG || p.NSEG < 1 || p.ksplit < 1 || p.nphase < 1 || p.nphase > 16)
        return hipErrorInvalidValue;
    if (vec4 && ((p.Cs_in & 3) || (p.SEG & 3) || (p.SEG_STRIDE & 3))) return hipErrorInvalidValue;
    if (p.ksplit > 1 && ((p.N & 3) || (p.Cs_out & 3) || (p.c_off & 3) || p.partial == nullptr || ((uintptr_t)p.partial & 15)))
        return hipErrorInvalidValue;
    dim3 grid((p.Mmax + BM - 1) / BM, p.Npad / BN, p.nphase * p.ksplit), block(256);
    // With events the kernel is dispatched through hipExtLaunchKernelGGL, which timestamps the kernel's own
    // dispatch packet: the events bracket exactly this kernel (not the split-K combine) and add no marker
    // packets to the stream (plain hipEventRecord pairs cost ~3.5 us each here, ~3 % of a step).
    const bool timed = ev_start != nullptr && ev_stop != nullptr;
#define VSTAB_LAUNCH(KERNEL, LDS)                                                                        \
    do {                                                                                                  \
        if (timed) hipExtLaunchKernelGGL(KERNEL, grid, block, LDS, stream, ev_start, ev_stop, 0, p);      \
        else KERNEL<<<grid, block, LDS, stream>>>(p);                                                     \
    } while (0)
    const bool use_dma = lds_dma_enabled();
    if (tile == TILE_128x128 && vec4 && use_dma)
        VSTAB_LAUNCH((conv_mfma_kernel<128, 128, 2, 2, true, true>), (conv_lds_bytes<128, 128>()));
    else if (tile == TILE_128x64 && vec4 && use_dma) {
        // Three of these workgroups fit a CU (48 KB of LDS each).  A launch of 513 ... 1024 of them that is a whole number of
        // two-per-CU rounds but not of three-per-CU rounds runs its last quarter one workgroup per CU -- or, as the dispatcher
        // hands the stragglers to whichever CU frees a slot first, two on some CUs and none on others (in-situ stamps of
        // deconv2 at B=8 512x512: 1024 workgroups, the last 256 take 63 us alone and 105 us where two share a CU, 238 us in
        // all).  Asking for 56 KB makes it two even rounds: 202 us.  Longer launches (the Winograd-domain GEMMs, 2048 / 4096
        // workgroups) measure the same or slower that way and keep three.
        size_t lds = conv_lds_bytes<128, 64>();
        {
            const long long wgs = (long long)grid.x * grid.y * grid.z;
            if (wgs > 2 * CONV_CUS && wgs <= 4 * CONV_CUS && wgs % (2 * CONV_CUS) == 0) lds = CONV_LDS_TWO_PER_CU;
        }
        VSTAB_LAUNCH((conv_mfma_kernel<128, 64, 2, 2, true, true>), lds);
    } else if (tile == TILE_128x32 && vec4 && use_dma)
        VSTAB_LAUNCH((conv_mfma_kernel<128, 32, 4, 1, true, true>), (conv_lds_bytes<128, 32>()));
    else if (tile == TILE_64x128 && vec4)
        VSTAB_LAUNCH((conv_mfma_kernel<64, 128, 1, 4, true, true>), (conv_lds_bytes<64, 128>()));
#ifdef VSTAB_HARNESS      // predict_flow2's tap table ran on this shape through round 3; it is tap_panel_kernel's now (tap_panel.hip)
    else if (tile == TILE_256x32 && vec4)
        VSTAB_LAUNCH((conv_mfma_kernel<256, 32, 4, 1, true, true>), (conv_lds_bytes<256, 32>()));
#endif
#ifdef VSTAB_HARNESS      // measured for one-sample launches (scripts/sweep_b1.sh: 3-8 % per layer), not worth a fifth product instantiation
    else if (tile == TILE_64x64 && vec4)
        VSTAB_LAUNCH((conv_mfma_kernel<64, 64, 2, 2, true, true>), (conv_lds_bytes<64, 64>()));
#endif
#ifdef VSTAB_HARNESS
    else if (tile == TILE_128x128 && vec4)
        VSTAB_LAUNCH((conv_mfma_kernel<128, 128, 2, 2, true>), (conv_lds_bytes<128, 128>()));
    else if (tile == TILE_128x64 && vec4)
        VSTAB_LAUNCH((conv_mfma_kernel<128, 64, 2, 2, true>), (conv_lds_bytes<128, 64>()));
    else if (tile == TILE_128x32 && vec4)
        VSTAB_LAUNCH((conv_mfma_kernel<128, 32, 4, 1, true>), (conv_lds_bytes<128, 32>()));
#endif
    else if (tile == TILE_128x64 && !vec4)       // dword gathers (27-channel rows that are not 16-byte friendly): tests/test_gpu_parity.py, C_in = 27 at odd widths
        VSTAB_LAUNCH((conv_mfma_kernel<128, 64, 2, 2, false>), (conv_lds_bytes<128, 64>()));
    else
        return hipErrorInvalidValue;
#undef VSTAB_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (p.ksplit > 1 && combine) {
        const long long total = (long long)p.Mmax * (p.N >> 2) * p.nphase;
        splitk_combine_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(p);
        e = hipGetLastError();
    }
    return e;
}

// Problem A (a transposed convolution on a 128x128, 128x64 or 64x128 tile; its split-K slabs, if any, are left for the caller's
// combine) and problem B (a 128x32 tap-table GEMM, slabs left for predict_up) in one launch.  hipErrorNotSupported: not a pair this
// kernel is built for -- launch them one after the other.
hipError_t launch_conv_dual(const ConvParams &pa_in, ConvTile tile_a, const ConvParams &pb_in, ConvTile tile_b, hipStream_t stream,
                            hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (!lds_dma_enabled() || tile_b != TILE_128x32 || (tile_a != TILE_128x128 && tile_a != TILE_128x64 && tile_a != TILE_64x128))
        return hipErrorNotSupported;
    ConvParams pa = pa_in, pb = pb_in;
    for (ConvParams *q : {&pa, &pb}) {
        q->no_remap = 0;
        q->out_vec4 = (((uintptr_t)q->out & 15) == 0 && (q->Cs_out & 3) == 0 && (q->c_off & 3) == 0) ? 1 : 0;
        if (q->in_bytes >= 0x80000000u || q->w_bytes >= 0x80000000u) return hipErrorInvalidValue;
        if (q->SEGP % 32 != 0 || q->SEGP < q->SEG || q->NSEG < 1 || q->ksplit < 1 || q->nphase < 1 || q->nphase > 16) return hipErrorInvalidValue;
        if ((q->Cs_in & 3) || (q->SEG & 3) || (q->SEG_STRIDE & 3)) return hipErrorInvalidValue;
        if (q->ksplit > 1 && ((q->N & 3) || (q->Cs_out & 3) || (q->c_off & 3) || q->partial == nullptr || ((uintptr_t)q->partial & 15)))
            return hipErrorInvalidValue;
    }
    const int BMa = tile_a == TILE_64x128 ? 64 : 128, BNa = tile_a == TILE_128x64 ? 64 : 128;
    if (pa.Npad % BNa != 0 || pb.Npad % 32 != 0) return hipErrorInvalidValue;
    const uint3 gA = make_uint3((unsigned)((pa.Mmax + BMa - 1) / BMa), (unsigned)(pa.Npad / BNa), (unsigned)(pa.nphase * pa.ksplit));
    const uint3 gB = make_uint3((unsigned)((pb.Mmax + 127) / 128), (unsigned)(pb.Npad / 32), (unsigned)(pb.nphase * pb.ksplit));
    const unsigned long long nA = (unsigned long long)gA.x * gA.y * gA.z, nB = (unsigned long long)gB.x * gB.y * gB.z;
    // Co-resident (two workgroups per CU hold both problems): the launch takes as long as its longer member.  Beyond that B's tiles go
    // first (see the kernel); appended BEHIND a transposed convolution that fills whole rounds they were a ragged round of their own
    // (+43 us on deconv2 at B=8 512x512, profiles/README.md r04).
    const int b_first = nA + nB > 2ull * CONV_CUS ? 1 : 0;
    const dim3 grid((unsigned)(nA + (b_first ? ((nB + 7ull) & ~7ull) : nB))), block(256);      // B first: B padded to whole groups of 8 (the kernel's XCD note)
    const bool timed = ev_start != nullptr && ev_stop != nullptr;
#define VSTAB_LAUNCH2(BM, BN, WM, WN)                                                                                                   \
    do {                                                                                                                                \
        size_t lds = std::max(conv_lds_bytes<BM, BN>(), conv_lds_bytes<128, 32>());                                                     \
        if (BN == 64 && nA > 2 * CONV_CUS && nA <= 4 * CONV_CUS && nA % (2 * CONV_CUS) == 0) lds = CONV_LDS_TWO_PER_CU;  /* as launch_conv */ \
        if (timed) hipExtLaunchKernelGGL((conv_dual_kernel<BM, BN, WM, WN, 128, 32, 4, 1>), grid, block, lds, stream, ev_start, ev_stop, 0, \
                                         pa, pb, (unsigned)nA, (unsigned)nB, b_first, gA, gB);                                          \
        else conv_dual_kernel<BM, BN, WM, WN, 128, 32, 4, 1><<<grid, block, lds, stream>>>(pa, pb, (unsigned)nA, (unsigned)nB, b_first, gA, gB); \
    } while (0)
    if (tile_a == TILE_128x128) VSTAB_LAUNCH2(128, 128, 2, 2);
    else if (tile_a == TILE_128x64) VSTAB_LAUNCH2(128, 64, 2, 2);
    else VSTAB_LAUNCH2(64, 128, 1, 4);
#undef VSTAB_LAUNCH2
    return hipGetLastError();
}

}  // namespace vstab
