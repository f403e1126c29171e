// Weight gradient of a convolution on the fp32 MFMA (backward of F1/F4 in model.py:786-893).
//
//   y[n,oy,ox,co] = b[co] + sum_{ky,kx,ci} x[n, s*oy+ky-p, s*ox+kx-p, ci] * W[ky,kx,ci,co]
//   dW[ky,kx,ci,co] = sum_{n,oy,ox} x[n, s*oy+ky-p, s*ox+kx-p, ci] * g[n,oy,ox,co]
//
// GEMM view: rows m = (ky, kx, ci) -- exactly the HWIO row-major order of dW -- columns = co, and the REDUCTION runs over
// the output pixels k = (n, oy, ox).  Unlike the forward kernel both operands have their reduction index as the SLOW memory
// index (a pixel's channels are contiguous in NHWC), which is what the MFMA wants here: lane (i, h) of
// v_mfma_f32_32x32x2_f32 needs A[m = i][k = h], i.e. 32 consecutive channels of ONE pixel per half-wave -- a conflict-free
// ds_read_b32 from a k-major LDS tile that LDS-DMA can fill with 16-byte loads.  No transposes, no im2col.
//
//   * workgroup tile 128 (m) x 128 (co), K-tile = 32 output pixels, 4 waves with 64x64 register tiles (AGPRs)
//   * LDS [2 stages][32 pixels][128 floats] for both operands, filled by `buffer_load_dwordx4 ... lds`; a pixel whose tap
//     falls outside the image, a pixel past the end of the batch or a channel past Cin/Cout reads as zero through the
//     descriptor's range check (offset 0xC0000000), so the loop has no branches
//   * a per-pixel table {element offset of the pixel's window origin, s*oy - p, s*ox - p} (built once per geometry by
//     wgrad_pixel_table_kernel) replaces the n/oy/ox integer divisions; a lane's tap and channel are per-thread constants
//   * split-K over blockIdx.z (the reduction is B*Ho*Wo long, the output only k*k*Cin x Cout): fp32 slabs summed in
//     slab order by wgrad_combine_kernel -- deterministic
// The same kernel gives the gradient of a 4x4 stride-2 transposed conv's weights (roles of x and g swapped by the caller).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "vstab_internal.h"

namespace vstab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define VSTAB_COMMA ,
#include "conv_kloop_gfx950.inc"

// ptab[k] = {x element offset of (n, s*oy - p, s*ox - p, 0), s*oy - p, s*ox - p, 0}
__global__ __launch_bounds__(256) void wgrad_pixel_table_kernel(int B, int Hi, int Wi, int Cs, int Ho, int Wo, int s, int p,
                                                                int4 *__restrict__ ptab)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= B * Ho * Wo) return;
    const int n = k / (Ho * Wo), r = k - n * Ho * Wo;
    const int oy = r / Wo, ox = r - oy * Wo;
    const int y0 = s * oy - p, x0 = s * ox - p;
    ptab[k] = make_int4(((n * Hi + y0) * Wi + x0) * Cs, y0, x0, 0);
}

// BN = 128 (wave tile 64x64) or 64 (wave tile 64x32, for layers with at most 64 output channels)
template <int BN>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const WgradParams p)
{
    constexpr int BM = 128, KT = 32;
    constexpr int NBW = BN / 64;                 // 32-column MFMA blocks per wave
    constexpr int BL = BN / 4;                   // lanes per pixel row of the B tile (16-byte chunks)
    constexpr int BR = 256 / BL;                 // B rows per DMA pass
    constexpr int BP = KT / BR;                  // DMA passes for B
    // dynamic LDS like the forward kernel: with static arrays hipcc orders every fragment read after the DMA that was
    // just issued into the OTHER stage (it sees one object) and the prefetch is lost
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];
    float *sA = reinterpret_cast<float *>(wg_smem);                 // [2][KT * BM]
    float *sB = sA + 2 * KT * BM;                                   // [2][KT * BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    unsigned bx_, by_, bz_;
    xcd_remap(bx_, by_, bz_);                       // the tiles of one K split (which all read the same pixels) share an XCD
    const int m0 = (int)bx_ * BM, n0 = (int)by_ * BN;
    const int batch = (int)bz_ / p.ksplit, split = (int)bz_ - batch * p.ksplit;
    const int ktiles = (p.K + KT - 1) / KT;
    const int kts = (ktiles + p.ksplit - 1) / p.ksplit;
    const int kt0 = split * kts, kt1 = min(ktiles, kt0 + kts);

    const unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x) + (long long)batch * p.x_bstride, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.g) + (long long)batch * p.g_bstride, 0, p.g_bytes, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(tid) >> 6;

    // this thread's column chunk (4 floats) inside the 128-float tile rows: constant over the whole K loop
    const int cc = (tid & 31) * 4;
    // A: row m = m0 + cc .. +3 -> tap and input channel (Cin % 4 == 0, so a chunk never straddles a tap)
    const int m = m0 + cc;
    const bool m_ok = m < p.M;
    const int tap = m_ok ? m / p.Cin : 0, ci = m - tap * p.Cin;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int a_delta = (ky * p.Wi + kx) * p.Cs_x + p.cx_off + ci;          // added to the pixel's window origin
    // B: column co = n0 + ccb .. +3
    const int ccb = (tid % BL) * 4;
    const bool n_ok = n0 + ccb < p.Cout;
    const int b_delta = p.cg_off + n0 + ccb;

    // one K-tile: 32 pixels x 128 floats per operand = 4 DMA instructions per thread per operand (8 pixel rows per pass).
    // The pixel-table entries of a tile are fetched one tile AHEAD of its DMA (plain loads into registers, issued behind
    // the previous tile's DMAs): the loop never waits on a table load, and there is no branch in it.
    int4 pt[4];
    auto load_pt = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pt[j] = p.ptab[min(kt * KT + (tid >> 5) + 8 * j, p.K - 1)];
    };
    auto dma_tile = [&](int kt, int buf) {
        const bool t_ok = kt < kt1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kt * KT + (tid >> 5) + 8 * j;
            const bool a_ok = t_ok & (k < p.K) & m_ok & ((unsigned)(pt[j].y + ky) < (unsigned)p.Hi) & ((unsigned)(pt[j].z + kx) < (unsigned)p.Wi);
            const unsigned aoff = a_ok ? (unsigned)(pt[j].x + a_delta) * 4u : OOB;
            __attribute__((address_space(3))) void *da =
                (__attribute__((address_space(3))) void *)(sA + buf * (KT * BM) + (8 * j + 2 * wave_u) * BM);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, da, 16, aoff, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int k = kt * KT + tid / BL + BR * j;
            const unsigned boff = (t_ok & (k < p.K) & n_ok) ? (unsigned)(k * p.Cs_g + b_delta) * 4u : OOB;
            __attribute__((address_space(3))) void *db =
                (__attribute__((address_space(3))) void *)(sB + buf * (KT * BN) + (BR * j + (64 / BL) * wave_u) * BN);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, db, 16, boff, 0, 0, 0);
        }
    };

    f32x16 acc[2][NBW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    const int a_col = wm * 64 + li, b_col = wn * (BN / 2) + li;

#ifdef VSTAB_NO_ASM_KLOOP
    constexpr bool ASM_KLOOP = false;               // A/B builds only (scripts/build_variant_lib.sh)
#else
    constexpr bool ASM_KLOOP = true;
#endif
    if (ASM_KLOOP && kt0 < kt1) {
        // The K loop as one assembly block (conv_kloop_gfx950.inc; tools/gen_conv_kloop.py documents the schedule): same tiles, same
        // MFMA order per accumulator as the C++ loop below (bit-identical), fragment reads a group of four k-steps ahead of their use.
        load_pt(kt0);
        dma_tile(kt0, 0);
        load_pt(kt0 + 1);
        __syncthreads();
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)sA;
        const unsigned ra0 = lds0 + (unsigned)(lh * BM + a_col) * 4u, ra1 = ra0 + 128u;
        const unsigned rb0 = lds0 + (unsigned)(2 * KT * BM + lh * BN + b_col) * 4u, rb1 = rb0 + 128u;
        const unsigned long long ax = (unsigned long long)(size_t)(p.x + (long long)batch * p.x_bstride);
        const unsigned long long ag = (unsigned long long)(size_t)(p.g + (long long)batch * p.g_bstride);
        i32x4 dx, dg;
        dx.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ax);
        dx.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ax >> 32) & 0xffffu));
        dx.z = __builtin_amdgcn_readfirstlane((int)p.x_bytes);
        dx.w = 0x00020000;
        dg.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ag);
        dg.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ag >> 32) & 0xffffu));
        dg.z = __builtin_amdgcn_readfirstlane((int)p.g_bytes);
        dg.w = 0x00020000;
        const int m_a = __builtin_amdgcn_readfirstlane((int)lds0 + wave_u * 1024);
        const int m_b = m_a + 2 * KT * BM * 4;
        const unsigned long long mok = __builtin_amdgcn_ballot_w64(m_ok), nok = __builtin_amdgcn_ballot_w64(n_ok);
        int v_k = (kt0 + 1) * KT + (tid >> 5), v_kb = (kt0 + 1) * KT + tid / BL;
        int v_bo = (v_kb * p.Cs_g + b_delta) * 4;
        int s_n = kt1 - kt0 - 1, s_t;
        unsigned v_t, v_o0, v_o1;
#define VSTAB_WG_IO(ACCS)                                                                                                                    \
        : ACCS, [vk] "+v"(v_k), [vkb] "+v"(v_kb), [vbo] "+v"(v_bo), [n] "+s"(s_n), [t] "=&s"(s_t), [vt] "=&v"(v_t), [vo0] "=&v"(v_o0),       \
          [vo1] "=&v"(v_o1)                                                                                                                 \
        : [ra0] "v"(ra0), [ra1] "v"(ra1), [rb0] "v"(rb0), [rb1] "v"(rb1), [vky] "v"(ky), [vkx] "v"(kx), [vad] "v"(a_delta),                \
          [px0] "v"(pt[0].x), [py0] "v"(pt[0].y), [pz0] "v"(pt[0].z), [px1] "v"(pt[1].x), [py1] "v"(pt[1].y), [pz1] "v"(pt[1].z),          \
          [px2] "v"(pt[2].x), [py2] "v"(pt[2].y), [pz2] "v"(pt[2].z), [px3] "v"(pt[3].x), [py3] "v"(pt[3].y), [pz3] "v"(pt[3].z),          \
          [rx] "s"(dx), [rg] "s"(dg), [ptab] "s"(p.ptab), [ma] "s"(m_a), [mb] "s"(m_b), [mok] "s"(mok), [nok] "s"(nok),                    \
          [kk] "s"(p.K), [km1] "s"(p.K - 1), [hi] "s"(p.Hi), [wi] "s"(p.Wi), [brow] "s"(BR * p.Cs_g * 4), [btile] "s"(KT * p.Cs_g * 4)    \
        : "memory", "vcc", "scc", VSTAB_WGRAD_CLOBBERS
        if constexpr (BN == 128) {
            asm volatile(VSTAB_WGRAD_ASM_128 VSTAB_WG_IO([c00] "+a"(acc[0][0]) VSTAB_COMMA [c01] "+a"(acc[0][NBW - 1]) VSTAB_COMMA [c10] "+a"(acc[1][0])
                                                        VSTAB_COMMA [c11] "+a"(acc[1][NBW - 1])));
        } else {
            asm volatile(VSTAB_WGRAD_ASM_64 VSTAB_WG_IO([c00] "+a"(acc[0][0]) VSTAB_COMMA [c10] "+a"(acc[1][0])));
        }
#undef VSTAB_WG_IO
        __syncthreads();                            // as the C++ loop's last barrier: every wave is past its last fragment read
    } else if (kt0 < kt1) {
        load_pt(kt0);
        dma_tile(kt0, 0);
        load_pt(kt0 + 1);
        __syncthreads();
        int buf = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            dma_tile(kt + 1, buf ^ 1);                                     // past the last tile: every lane out of range
            load_pt(kt + 2);
            const float *cA = sA + buf * (KT * BM) + lh * BM + a_col;
            const float *cB = sB + buf * (KT * BN) + lh * BN + b_col;
            // MFMA k-step ks covers pixels 2*ks + {0, 1}; the fragments of step ks+1 are read before the MFMAs of step ks
            float a0 = cA[0], a1 = cA[32], b0 = cB[0], b1 = NBW > 1 ? cB[32] : 0.f;
#pragma unroll
            for (int ks = 0; ks < KT / 2; ++ks) {
                float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
                if (ks + 1 < KT / 2) {
                    na0 = cA[(ks + 1) * 2 * BM]; na1 = cA[(ks + 1) * 2 * BM + 32];
                    nb0 = cB[(ks + 1) * 2 * BN];
                    if (NBW > 1) nb1 = cB[(ks + 1) * 2 * BN + 32];
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                if (NBW > 1) acc[0][NBW - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][NBW - 1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                if (NBW > 1) acc[1][NBW - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][NBW - 1], 0, 0, 0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
            __syncthreads();                                               // tile kt+1 landed, tile kt's slot is free
            buf ^= 1;
        }
    }

    // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  As in conv_mfma.hip the tile leaves through
    // LDS (the operand ring is free now and as large as the tile) as 16-byte stores of whole rows when the destination allows it,
    // instead of 64 four-byte store instructions per wave.
    const bool slab = p.ksplit > 1;
    const int ld = slab ? p.Cout : (p.ldw ? p.ldw : p.Cout);
    float *dst = slab ? p.partial + (size_t)(batch * p.ksplit + split) * p.M * p.Cout : p.dW + (size_t)batch * p.M * p.Cout + p.col0;
    const bool vec = ((ld & 3) == 0) && ((p.Cout & 3) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    if (vec) {
        static_assert(BM * BN <= 2 * KT * (BM + BN), "the output tile fits the operand ring");
        float *sC = sA;                                   // [BM][BN]; the loop's last barrier has every wave past its last read
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int cl = wn * (BN / 2) + nb * 32 + li;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sC[(wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + cl] = acc[mb][nb][r];
        }
        __syncthreads();
        constexpr int C4 = BN / 4;
        const bool rmw = p.ksplit == 1 && p.accumulate;
#pragma unroll 4
        for (int e = tid; e < BM * C4; e += 256) {
            const int rl = e / C4, c4 = e - rl * C4;
            const int row = m0 + rl, col = n0 + c4 * 4;
            if (row >= p.M || col >= p.Cout) continue;    // Cout % 4 == 0: a float4 is inside or outside as a whole
            f32x4 v = *reinterpret_cast<const f32x4 *>(sC + rl * BN + c4 * 4);
            float *o = dst + (size_t)row * ld + col;
            if (rmw) v += *reinterpret_cast<const f32x4 *>(o);
            *reinterpret_cast<f32x4 *>(o) = v;
        }
        return;
    }
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int col = n0 + wn * (BN / 2) + nb * 32 + li;
        if (col >= p.Cout) continue;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < p.M) {
                    float v = acc[mb][nb][r];
                    if (p.ksplit == 1 && p.accumulate) v += dst[(size_t)row * ld + col];
                    dst[(size_t)row * ld + col] = v;
                }
            }
    }
}

// n = M*Cout elements per batch; blockIdx.y = batch (slabs [batch][ks][n], result [batch][n]); the result's rows are ld floats apart
// and start at column col0 (a column window of a wider dW)
__global__ __launch_bounds__(256) void wgrad_combine_kernel(const float *__restrict__ partial, int ks, long long n,
                                                            int accumulate, float *__restrict__ dW, int cout, int ld, int col0)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    partial += (long long)blockIdx.y * ks * n;
    const long long row = i / cout;
    dW += (long long)blockIdx.y * n + row * (ld - cout) + col0;
    float s = 0.f;
    int k = 0;
    for (; k + 4 <= ks; k += 4) {
        const float v0 = partial[(k + 0) * n + i], v1 = partial[(k + 1) * n + i];
        const float v2 = partial[(k + 2) * n + i], v3 = partial[(k + 3) * n + i];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < ks; ++k) s += partial[k * n + i];
    if (accumulate) s += dW[i];
    dW[i] = s;
}

// out[c] (+)= sum over rows of g[row][c_off + c] (bias / beta gradients, BatchNorm sums).  Two deterministic stages: a
// (64-channel block) x (row chunk) grid of workgroups -- 4 waves striding over the chunk's rows, coalesced 256-byte row
// reads -- writes one partial per chunk; the second stage adds the chunks in order.
__global__ __launch_bounds__(256) void column_sum_kernel(const float *__restrict__ g, long long rows, int rows_per_chunk, int Cs,
                                                         int c_off, int C, int Cp, float *__restrict__ part)
{
    // Cp = min(64, next power of two >= C): a wave covers 64 / Cp rows at once, so narrow slices (the 2-channel flow
    // gradients) still keep every lane busy
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RW = 64 / Cp, sub = lane / Cp;
    const int c = blockIdx.x * 64 + (lane & (Cp - 1));
    const long long r0 = (long long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const long long st = 4LL * RW;
        long long r = r0 + wave * RW + sub;
        for (; r + 3 * st < r1; r += 4 * st) {                       // four rows in flight per lane
            const float v0 = g[r * Cs + c_off + c], v1 = g[(r + st) * Cs + c_off + c];
            const float v2 = g[(r + 2 * st) * Cs + c_off + c], v3 = g[(r + 3 * st) * Cs + c_off + c];
            s0 += v0; s1 += v1; s0 += v2; s1 += v3;
        }
        for (; r < r1; r += st) s0 += g[r * Cs + c_off + c];
    }
    float s = s0 + s1;
    for (int off = Cp; off < 64; off <<= 1) s += __shfl_xor(s, off);
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && lane < Cp && c < C)
        part[(long long)blockIdx.y * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// second stage: one workgroup of 16 waves per 64 channels; wave w adds chunks w, w+16, ... (four loads in flight), the
// 16 partial sums are then added in wave order -- a fixed order, so the result is reproducible.
__global__ __launch_bounds__(1024) void column_sum_final_kernel(const float *__restrict__ part, int chunks, int C, float *__restrict__ out,
                                                                int accumulate)
{
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int k = wave;
        for (; k + 48 < chunks; k += 64) {
            const float v0 = part[(long long)k * C + c], v1 = part[(long long)(k + 16) * C + c];
            const float v2 = part[(long long)(k + 32) * C + c], v3 = part[(long long)(k + 48) * C + c];
            s0 += v0; s1 += v1; s0 += v2; s1 += v3;
        }
        for (; k < chunks; k += 16) s0 += part[(long long)k * C + c];
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += red[w][lane];
        out[c] = accumulate ? out[c] + s : s;
    }
}

int wgrad_choose_split(const WgradParams &p)
{
    const int bn = p.Cout <= 64 ? 64 : 128;
    const long long tiles = (long long)((p.M + 127) / 128) * ((p.Cout + bn - 1) / bn) * (p.nbatch > 1 ? p.nbatch : 1);
    const int ktiles = (p.K + 31) / 32;
    int ks = (int)(512 / (tiles > 0 ? tiles : 1));                           // two 64 KB workgroups fit a CU
    if (ks < 1) ks = 1;
    if (ks > 256) ks = 256;                                                  // (a couple of tiles over a million pixels: the tap-table head)
    const int cap = ktiles / 4 > 1 ? ktiles / 4 : 1;
    if (ks > cap) ks = cap;
    const int kts = (ktiles + ks - 1) / ks;
    return (ktiles + kts - 1) / kts;
}

hipError_t launch_wgrad_pixel_table(int B, int Hi, int Wi, int Cs, int Ho, int Wo, int s, int pad, int4 *ptab, hipStream_t stream)
{
    const int K = B * Ho * Wo;
    wgrad_pixel_table_kernel<<<dim3((unsigned)((K + 255) / 256)), dim3(256), 0, stream>>>(B, Hi, Wi, Cs, Ho, Wo, s, pad, ptab);
    return hipGetLastError();
}

hipError_t launch_wgrad(const WgradParams &p, hipStream_t stream)
{
    if ((p.Cin & 3) || (p.Cs_x & 3) || (p.cx_off & 3) || (p.Cs_g & 3) || (p.cg_off & 3) || p.ksplit < 1) return hipErrorInvalidValue;
    if (p.x_bytes >= 0x80000000u || p.g_bytes >= 0x80000000u) return hipErrorInvalidValue;
    if (p.ksplit > 1 && !p.partial) return hipErrorInvalidValue;
    const unsigned nb = (unsigned)(p.nbatch > 1 ? p.nbatch : 1);
    if (p.Cout <= 64) {
        dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)((p.Cout + 63) / 64), (unsigned)p.ksplit * nb);
        wgrad_mfma_kernel<64><<<grid, dim3(256), 2 * 32 * (128 + 64) * sizeof(float), stream>>>(p);
    } else {
        dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)((p.Cout + 127) / 128), (unsigned)p.ksplit * nb);
        wgrad_mfma_kernel<128><<<grid, dim3(256), 2 * 32 * (128 + 128) * sizeof(float), stream>>>(p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (p.ksplit > 1) {
        const long long n = (long long)p.M * p.Cout;
        wgrad_combine_kernel<<<dim3((unsigned)((n + 255) / 256), nb), dim3(256), 0, stream>>>(p.partial, p.ksplit, n, p.accumulate, p.dW, p.Cout,
                                                                                            p.ldw ? p.ldw : p.Cout, p.col0);
        e = hipGetLastError();
    }
    return e;
}

int column_sum_chunks(long long rows, int C) { return reduce_chunks(rows, C); }

hipError_t launch_column_sum(const float *g, long long rows, int Cs, int c_off, int C, float *out, int accumulate, float *scratch,
                             hipStream_t stream)
{
    const int chunks = column_sum_chunks(rows, C);                       // scratch: chunks * C floats
    const int rpc = (int)((rows + chunks - 1) / chunks);
    int Cp = 64;
    while (Cp / 2 >= C && Cp > 1) Cp /= 2;
    column_sum_kernel<<<dim3((unsigned)((C + 63) / 64), (unsigned)chunks), dim3(256), 0, stream>>>(g, rows, rpc, Cs, c_off, C, Cp, scratch);
    column_sum_final_kernel<<<dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, stream>>>(scratch, chunks, C, out, accumulate);
    return hipGetLastError();
}

}  // namespace vstab
