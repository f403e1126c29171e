// "Row-window" implicit-GEMM convolution for the network's first layer (model.py:807-809:
// PadLayer(3) + Conv2d 7x7 s2 VALID + BatchNorm + lrelu on the 27-channel frame stack).
//
// Why a second conv kernel: with C_in = 27 a pixel is 108 bytes, so the generic kernel has to
// gather its im2col tile with predicated dword loads (16 per thread per 32-wide K-tile) and
// runs at ~50 % of the fp32 MFMA rate.  But in NHWC the K run of one output pixel and one
// filter row -- KW*C_in = 189 consecutive floats -- is contiguous, and neighbouring output
// pixels start s*C_in = 54 floats apart.  So for 128 consecutive output pixels of one output
// row, ALL operands of one filter row live in ONE contiguous window of 54*127+192 floats of
// the input row.  The workgroup copies that window to LDS with aligned 16-byte loads (7 per
// thread per filter row instead of 96 dword gathers) and the MFMA A operands are read
// straight out of it (lane i reads at 54*i + k): the im2col matrix is never materialised,
// not even in LDS.
//   * A reads are ds_read_b64 (two k-steps each); the run is extended by one leading dummy
//     float (zero weight) so that 54*i + k is 8-byte aligned; 32 lanes hit 32 distinct even
//     banks -> conflict free.
//   * B (weights, 64 x 1344, packed per K-tile) is read by every wave straight from
//     global/L1 into registers (4 x 16 B per lane per K-tile, prefetched one tile ahead):
//     no LDS staging, hence no barrier per K-tile -- only one per filter row (7 per block).
//   * window double-buffered: the loads of filter row ky+1 are in flight during the 192
//     MFMAs of row ky.
// Zero padding: window floats left of / right of the image row and whole rows above/below
// the image are loaded as zeros via the buffer descriptor's range check (no branches).
// Requires (s*C_in) even, (W*C_in) % 4 == 0, 16-byte aligned input; otherwise the generic
// scalar-gather kernel runs instead.
#include <hip/hip_ext.h>

#include "vstab_internal.h"

namespace vstab {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#include "conv_kloop_gfx950.inc"

// MB = 32-pixel blocks per wave: 2 -> 128 output pixels per workgroup (wave tile 64 x 32); 1 -> 64 pixels (wave tile 32 x 32) for
// launches that would otherwise put fewer than two workgroups on a CU (one sample at 384x512: 384 workgroups on 256 CUs run as
// two uneven rounds, 91 us; 768 half-size ones are all resident at once)
template <int NWIN4, int MB>   // float4 loads per thread per window
__global__ __launch_bounds__(256) void conv_rowwin_kernel(const RowWinParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *win = reinterpret_cast<float *>(smem);           // [2][WLEN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                // 2x2 waves, wave tile 32*MB (pixels) x 32 (channels)
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware order: grid = (row, x tile, sample) so that a remapped XCD range is a band of consecutive output rows, whose
    // 7-row input windows overlap by five rows
    unsigned bx_, by_, bz_;
    xcd_remap(bx_, by_, bz_);
    // a chore for the launches that FOLLOW this one in the forward: the first workgroup zeroes the ticket words of the in-launch split-K
    // reductions (conv_skinny.hip).  As the forward's first launch this kernel finishes before any of them starts (stream order), and the
    // words ride here instead of in a memset node of their own (4.7 us per forward: 1 ... 1.6 % of a one-sample frame)
    if (p.clear_n > 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int i = tid; i < p.clear_n; i += 256) p.clear_words[i] = 0u;
    const int stream_r = (MB == 2 && p.asm_loop) ? p.stream_rows : 0;          // > 0: this workgroup computes stream_r output rows (oy, oy + S, oy + 2 S ...)
    // rows between a stream's tiles.  1: the stream walks down CONSECUTIVE rows.  (Interleaving a column's streams -- stride = their
    // number, so that they work on adjacent rows at any time and share input rows in L2 -- brings the launch's HBM reads from 3.2x to 1.9x
    // the one-tile-per-workgroup launch's, and measures the same at B=8 512x512 but 2.5 % SLOWER at 720p / 1080p: profiles/README.md r03p)
    const int stream_s = 1;
    int oy = stream_r > 0 ? (int)bx_ * stream_r : (int)bx_;
    const int ox0 = p.ox_base + (int)by_ * (64 * MB), n = (int)bz_;
    const int pix_step = p.s_in * p.Cs_in;
    const int row_floats = p.Wi * p.Cs_in;
    const int g0 = pix_step * ox0 + p.e_off - p.w_a;       // window start, floats from the row start (multiple of 4)

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
    const unsigned OOB = 0xC0000000u;

    f32x4 wv[NWIN4];
    auto load_window = [&](int ky) {
        const int iy = oy * p.s_in + p.off_y + ky;
        const bool yok = (unsigned)iy < (unsigned)p.Hi;
        const int rowbase = ((n * p.Hi + iy) * p.Wi) * p.Cs_in;      // element offset (< 2^29, checked on the host)
#pragma unroll
        for (int j = 0; j < NWIN4; ++j) {
            const int c4 = tid + 256 * j;
            const int g = g0 + 4 * c4;
            const bool ok = yok && g >= 0 && g < row_floats && 4 * c4 < p.WLEN;
            const unsigned off = ok ? (unsigned)(rowbase + g) * 4u : OOB;
            wv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0));
        }
    };
    // byte stride of the two window buffers: the assembly loop fetches windows by LDS-DMA, whole 4 KB wave chunks (lanes past the
    // window's end land zeros), so its buffers are a whole number of chunks apart
    constexpr int ASM_BUF = MB == 2 ? VSTAB_ROWWIN_BUF_BYTES : VSTAB_ROWWIN1_BUF_BYTES;
    const int wstride = p.asm_loop ? ASM_BUF / 4 : p.WLEN;
    auto store_window = [&](int buf) {
        float *d = win + buf * wstride;
#pragma unroll
        for (int j = 0; j < NWIN4; ++j) {
            const int c4 = tid + 256 * j;
            if (4 * c4 < p.WLEN) *reinterpret_cast<f32x4 *>(d + 4 * c4) = wv[j];
        }
    };

    // B fragments: packed [kt][Npad][32]; lane (i,h) of n-block wn owns floats (2q+h)*4 .. +3 of row wn*32+i
    const float *bbase = p.wpk + (long long)(wn * 32 + li) * 32 + lh * 4;
    const int ktile_stride = p.Npad * 32;
    f32x4 b0[4], b1[4];
    const int KT = p.KH * (p.SEGP >> 5);
    auto load_b = [&](f32x4 (&dst)[4], int kt) {
        const float *s = bbase + (long long)(kt < KT ? kt : KT - 1) * ktile_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f32x4 *>(s + q * 8);
    };

    f32x16 acc[MB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int kpr = p.SEGP >> 5;                       // K-tiles per filter row
    // A operand address inside the window: pix_step*m + w_a + 32*kc + 4*r + 2*h
    const int a_off0 = pix_step * (wm * 32 * MB + li) + p.w_a + 2 * lh;
    const int a_off1 = a_off0 + pix_step * 32;

    // one K-tile: 16 ds_read_b64 + 32 MFMAs; the B fragments are already in registers
    auto compute = [&](const float *w, int kc, const f32x4 (&b)[4]) {
        const float *wa0 = w + a_off0 + 32 * kc, *wa1 = w + a_off1 + 32 * kc;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const f32x2 a0 = *reinterpret_cast<const f32x2 *>(wa0 + 4 * r);
            f32x2 a1 = a0;
            if (MB == 2) a1 = *reinterpret_cast<const f32x2 *>(wa1 + 4 * r);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int s = 2 * r + j;
                const float bv = b[s >> 2][s & 3];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bv, acc[0], 0, 0, 0);
                if (MB == 2) acc[MB - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bv, acc[MB - 1], 0, 0, 0);
            }
        }
    };

    load_window(0);
    store_window(0);
    if (p.asm_loop) {
        // The K loop as one assembly block (conv_kloop_gfx950.inc; tools/gen_conv_kloop.py documents the schedule): same MFMA order per
        // accumulator as the C++ loop below, so the same bits; fragments of K-tile t+1 are requested under the MFMAs of tile t,
        // the next filter row's window arrives by LDS-DMA, one barrier per filter row.
        {
            __syncthreads();
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)win;
            unsigned a0c = lds0 + (unsigned)a_off0 * 4u, a1c = lds0 + (unsigned)a_off1 * 4u;
            unsigned a0n = a0c + ASM_BUF, a1n = a1c + ASM_BUF;
            const unsigned vb = (unsigned)((wn * 32 + li) * 32 + lh * 4) * 4u;
            unsigned w[7];                            // per-lane byte offset of its 16 bytes of every window chunk inside the input row, or out of range
#pragma unroll
            for (int j = 0; j < 7; ++j) w[j] = OOB;
#pragma unroll
            for (int j = 0; j < NWIN4; ++j) {
                const int c4 = tid + 256 * j;
                const int g = g0 + 4 * c4;
                w[j] = (g >= 0 && g < row_floats && 4 * c4 < p.WLEN) ? (unsigned)g * 4u : OOB;
            }
            const unsigned long long ain = (unsigned long long)(size_t)p.in, awt = (unsigned long long)(size_t)p.wpk;
            i32x4 din, dwt;
            din.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ain);
            din.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ain >> 32) & 0xffffu));
            din.z = __builtin_amdgcn_readfirstlane((int)p.in_bytes);
            din.w = 0x00020000;
            dwt.x = __builtin_amdgcn_readfirstlane((int)(unsigned)awt);
            dwt.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(awt >> 32) & 0xffffu));
            dwt.z = __builtin_amdgcn_readfirstlane(KT * ktile_stride * 4);
            dwt.w = 0x00020000;
            const int wave_u = __builtin_amdgcn_readfirstlane(tid) >> 6;
            const int m_cur = __builtin_amdgcn_readfirstlane((int)lds0 + wave_u * 1024);
            int m_n = m_cur + ASM_BUF;                                    // the first fetch goes to buffer 1
            const int m_x = m_cur ^ m_n;
            int s_iy = oy * p.s_in + p.off_y + 1;                         // the next filter row to fetch
            int s_soff = ((n * p.Hi + s_iy) * p.Wi) * p.Cs_in * 4;
            int s_koff = 0, s_nrows = p.KH - 1;
            unsigned v_t0, v_t1;
            long long s_mask;
            if (MB == 2 && stream_r > 0) {
                // the stream form (tools/gen_conv_kloop.py, RowWinStreamGen): stream_r tiles back to back, every tile but the last stored from
                // registers under the next tile's first filter row; the last tile's accumulators come back for the epilogue below
                const unsigned long long aout = (unsigned long long)(size_t)p.out;
                i32x4 dout;
                dout.x = __builtin_amdgcn_readfirstlane((int)(unsigned)aout);
                dout.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(aout >> 32) & 0xffffu));
                dout.z = __builtin_amdgcn_readfirstlane((int)((unsigned)p.B * (unsigned)p.Ho * (unsigned)p.Wo * (unsigned)p.Cs_out * 4u));
                dout.w = 0x00020000;
                const unsigned vout = (unsigned)((wm * 64 + 4 * lh) * p.Cs_out + p.c_off + wn * 32 + li) * 4u;
                const float bias_l = p.bias[wn * 32 + li];
                const float slope = p.act == 1 ? 0.1f : (p.act == 0 ? 1.0f : 0.0f);      // max(v, slope * v): leaky relu / identity / relu
                int s_ntiles = stream_r, s_obase = (((n * p.Ho + oy) * p.Wo + ox0) * p.Cs_out) * 4, s_st;
                float o[32];
                asm volatile(VSTAB_ROWWIN_STREAM_ASM_KPR6
                             : [a0c] "+v"(a0c), [a1c] "+v"(a1c), [a0n] "+v"(a0n), [a1n] "+v"(a1n),
                               [mn] "+s"(m_n), [iy] "+s"(s_iy), [soff] "+s"(s_soff), [koff] "+s"(s_koff), [nrows] "+s"(s_nrows), [ntiles] "+s"(s_ntiles),
                               [obase] "+s"(s_obase), [st] "=&s"(s_st), [vt0] "=&v"(v_t0), [vt1] "=&v"(v_t1), [mask] "=&s"(s_mask),
                               [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5]), [o6] "=&v"(o[6]),
                               [o7] "=&v"(o[7]), [o8] "=&v"(o[8]), [o9] "=&v"(o[9]), [o10] "=&v"(o[10]), [o11] "=&v"(o[11]), [o12] "=&v"(o[12]),
                               [o13] "=&v"(o[13]), [o14] "=&v"(o[14]), [o15] "=&v"(o[15]), [o16] "=&v"(o[16]), [o17] "=&v"(o[17]), [o18] "=&v"(o[18]),
                               [o19] "=&v"(o[19]), [o20] "=&v"(o[20]), [o21] "=&v"(o[21]), [o22] "=&v"(o[22]), [o23] "=&v"(o[23]), [o24] "=&v"(o[24]),
                               [o25] "=&v"(o[25]), [o26] "=&v"(o[26]), [o27] "=&v"(o[27]), [o28] "=&v"(o[28]), [o29] "=&v"(o[29]), [o30] "=&v"(o[30]),
                               [o31] "=&v"(o[31])
                             : [vb] "v"(vb), [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]),
                               [din] "s"(din), [dwt] "s"(dwt), [mx] "s"(m_x), [hi] "s"(p.Hi), [rowbytes] "s"(row_floats * 4), [kstride] "s"(ktile_stride * 4),
                               [bias] "v"(bias_l), [slope] "s"(slope), [cs4] "s"(p.Cs_out * 4), [ostep] "s"(stream_s * p.Wo * p.Cs_out * 4), [vout] "v"(vout),
                               [dout] "s"(dout), [fwd] "s"(stream_s * p.s_in - p.KH)
                             : "memory", "scc", VSTAB_ROWWIN_STREAM_CLOBBERS);
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[0][r] = o[r]; acc[MB - 1][r] = o[16 + r]; }
                oy += (stream_r - 1) * stream_s;      // the epilogue below stores the stream's last tile
            } else if constexpr (MB == 1) {
                asm volatile(VSTAB_ROWWIN1_ASM_KPR6
                             : [c0] "+a"(acc[0]), [a0c] "+v"(a0c), [a0n] "+v"(a0n),
                               [mn] "+s"(m_n), [iy] "+s"(s_iy), [soff] "+s"(s_soff), [koff] "+s"(s_koff), [nrows] "+s"(s_nrows),
                               [vt0] "=&v"(v_t0), [vt1] "=&v"(v_t1), [mask] "=&s"(s_mask)
                             : [vb] "v"(vb), [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]),
                               [din] "s"(din), [dwt] "s"(dwt), [mx] "s"(m_x), [hi] "s"(p.Hi), [rowbytes] "s"(row_floats * 4), [kstride] "s"(ktile_stride * 4)
                             : "memory", "scc", VSTAB_ROWWIN1_CLOBBERS);
                (void)a1c; (void)a1n;
            } else
            asm volatile(VSTAB_ROWWIN_ASM_KPR6
                         : [c0] "+a"(acc[0]), [c1] "+a"(acc[MB - 1]), [a0c] "+v"(a0c), [a1c] "+v"(a1c), [a0n] "+v"(a0n), [a1n] "+v"(a1n),
                           [mn] "+s"(m_n), [iy] "+s"(s_iy), [soff] "+s"(s_soff), [koff] "+s"(s_koff), [nrows] "+s"(s_nrows),
                           [vt0] "=&v"(v_t0), [vt1] "=&v"(v_t1), [mask] "=&s"(s_mask)
                         : [vb] "v"(vb), [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]),
                           [din] "s"(din), [dwt] "s"(dwt), [mx] "s"(m_x), [hi] "s"(p.Hi), [rowbytes] "s"(row_floats * 4), [kstride] "s"(ktile_stride * 4)
                         : "memory", "scc", VSTAB_ROWWIN_CLOBBERS);
        }
    } else {
    load_b(b0, 0);
    __syncthreads();
    int buf = 0, kt = 0;
    for (int ky = 0; ky < p.KH; ++ky) {
        const bool more = ky + 1 < p.KH;
        if (more) load_window(ky + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float *w = win + buf * p.WLEN;
        int kc = 0;
        // two K-tiles per trip with two named fragment sets: the loads of tile t+1 are issued before
        // the MFMAs of tile t and first used 32 MFMAs later (no register copies, no early wait)
        // (sched_barrier pins the loads here: hipcc otherwise sinks them to their first use and
        // waits for the L2 round trip four times per K-tile)
        for (; kc + 1 < kpr; kc += 2, kt += 2) {
            load_b(b1, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(w, kc, b0);
            load_b(b0, kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            compute(w, kc + 1, b1);
        }
        if (kc < kpr) {                                  // odd number of K-tiles per filter row
            load_b(b1, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(w, kc, b0);
#pragma unroll
            for (int q = 0; q < 4; ++q) b0[q] = b1[q];
            ++kt;
        }
        if (more) {
            store_window(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    }

    // epilogue: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  As in conv_mfma.hip the tile leaves through LDS (the window
    // buffers are free now) as 16-byte stores of whole 256-byte pixel rows instead of 32 four-byte store instructions per wave.
    const int col = wn * 32 + li;
    if (p.out_vec4) {
        constexpr int TP = 64 * MB;                   // pixels of the tile; sC [TP][64]
        __syncthreads();                              // every wave has read its last operands out of the window
        float *sC = win;
        const float bv = col < p.N ? p.bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sC[(wm * 32 * MB + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + col] = acc[mb][r] + bv;
        __syncthreads();
        const float slope = p.act == 1 ? 0.1f : 0.0f;
        float *orow = p.out + ((long long)(n * p.Ho + oy) * p.Wo + ox0) * p.Cs_out + p.c_off;
#pragma unroll 4
        for (int e = tid; e < TP * 16; e += 256) {
            const int px = e >> 4, c4 = (e & 15) * 4;
            if (ox0 + px >= p.Wo || c4 >= p.N) continue;
            f32x4 v = *reinterpret_cast<const f32x4 *>(sC + px * 64 + c4);
            if (p.act) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], slope * v[i]);
            }
            float *o = orow + (long long)px * p.Cs_out + c4;
            if (c4 + 4 <= p.N) *reinterpret_cast<f32x4 *>(o) = v;
            else for (int i = 0; c4 + i < p.N; ++i) o[i] = v[i];
        }
    } else if (col < p.N) {
        const float bv = p.bias[col];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = ox0 + wm * 32 * MB + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ox < p.Wo) {
                    float v = acc[mb][r] + bv;
                    if (p.act) v = fmaxf(v, (p.act == 1 ? 0.1f : 0.0f) * v);
                    p.out[((long long)(n * p.Ho + oy) * p.Wo + ox) * p.Cs_out + p.c_off + col] = v;
                }
            }
    }
}

bool rowwin_applicable(const RowWinParams &p)
{
    const int pix_step = p.s_in * p.Cs_in;
    return (pix_step % 2 == 0) && ((p.Wi * p.Cs_in) % 4 == 0) && (((uintptr_t)p.in & 15) == 0) && p.N <= 64 &&
           p.Npad == 64 && (p.MB == 1 || p.MB == 2) && p.WLEN <= (p.MB == 2 ? 7 : 4) * 1024 && (p.WLEN % 4) == 0 &&
           p.in_bytes < 0x80000000u;
}

hipError_t rowwin_set_attributes()
{
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_rowwin_kernel<7, 2>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 7 * 1024 * 4);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(conv_rowwin_kernel<4, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 1024 * 4);
}

hipError_t launch_conv_rowwin(const RowWinParams &p, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (!rowwin_applicable(p)) return hipErrorInvalidValue;
    const int tile = 64 * p.MB;
    if (p.ox_base < 0 || p.ox_base >= p.Wo || (p.ox_base & 1) || p.ntile_x < 0) return hipErrorInvalidValue;
    const int ntx = p.ntile_x > 0 ? p.ntile_x : (p.Wo - p.ox_base + tile - 1) / tile;       // x tiles of THIS launch
    dim3 grid(p.Ho, ntx, p.B), block(256);       // (row, x tile, sample): see the XCD remap in the kernel
    RowWinParams q = p;
    // the staged epilogue needs 16-byte friendly output rows and the [tile][64] staging area inside the two window buffers
    q.out_vec4 = (((uintptr_t)p.out & 15) == 0 && (p.Cs_out & 3) == 0 && (p.c_off & 3) == 0 && 2 * p.WLEN >= tile * 64) ? 1 : 0;
    q.asm_loop = ((p.SEGP >> 5) == 6 && p.KH >= 1 && 4 * p.WLEN <= (p.MB == 2 ? VSTAB_ROWWIN_BUF_BYTES : VSTAB_ROWWIN1_BUF_BYTES)) ? 1 : 0;
#ifdef VSTAB_NO_ASM_KLOOP
    q.asm_loop = 0;                              // A/B builds only (scripts/build_variant_lib.sh)
#endif
    // stream form (a workgroup walks down R consecutive output rows of one x tile of one sample; prologue and epilogue once per R tiles):
    // needs whole 128-pixel tiles, all 64 output channels, a plain activation, 7 filter rows.  R = the divisor of Ho that minimises a
    // small cost model in units of one tile's time with two workgroups per CU: rounds of 512 workgroups x (R + F), F = the fixed
    // prologue + epilogue (0.08: 13 k of 172 k cycles); a launch of at most 256 workgroups has a CU to each (0.53 per tile).  B=8
    // 512x512: R = 8, 512 workgroups, one round; B=8 1080p: R = 30, 1008 workgroups.
    q.stream_rows = 0;
    if (q.asm_loop && p.MB == 2 && p.KH == 7 && p.N == 64 && p.act >= 0 && p.act <= 2 && p.ox_base + ntx * 128 <= p.Wo &&
        (unsigned long long)p.B * p.Ho * p.Wo * p.Cs_out * 4ull < 0x80000000ull) {       // its byte offsets are signed 32-bit, like the input side's
        const double F = 0.08;
        const long long cols = (long long)ntx * p.B;
        double best = (double)((cols * p.Ho + 511) / 512) * (1.0 + F);          // one tile per workgroup
        for (int R = 2; R <= 32 && R <= p.Ho; ++R) {
            if (p.Ho % R) continue;
            const long long wgs = cols * (p.Ho / R);
            const double c = wgs <= 256 ? 0.53 * R + F : (double)((wgs + 511) / 512) * (R + F);
            if (c < best * 0.995) { best = c; q.stream_rows = R; }
        }
    }
#ifdef VSTAB_NO_ROWWIN_STREAM
    q.stream_rows = 0;                           // A/B builds only
#endif
    if (q.stream_rows > 0) grid.x = (unsigned)(p.Ho / q.stream_rows);
    const size_t lds2 = q.asm_loop ? (size_t)2 * (p.MB == 2 ? VSTAB_ROWWIN_BUF_BYTES : VSTAB_ROWWIN1_BUF_BYTES) : (size_t)2 * p.WLEN * 4;
    const bool timed = ev_start || ev_stop;      // timestamps of the kernel's own dispatch packet, no marker packets (see conv_mfma.hip); a
                                                 // launch that is one half of a pair carries only the start or only the stop event
    if (p.MB == 2) {
        if (timed) hipExtLaunchKernelGGL((conv_rowwin_kernel<7, 2>), grid, block, lds2, stream, ev_start, ev_stop, 0, q);
        else conv_rowwin_kernel<7, 2><<<grid, block, lds2, stream>>>(q);
    } else {
        if (timed) hipExtLaunchKernelGGL((conv_rowwin_kernel<4, 1>), grid, block, lds2, stream, ev_start, ev_stop, 0, q);
        else conv_rowwin_kernel<4, 1><<<grid, block, lds2, stream>>>(q);
    }
    return hipGetLastError();
}

}  // namespace vstab
