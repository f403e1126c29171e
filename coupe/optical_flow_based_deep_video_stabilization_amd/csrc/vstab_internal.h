// Internal declarations shared by the HIP kernels, the host packer and the C ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vstab {

inline int round_up_c(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------
// XCD-aware workgroup -> tile mapping (device side).  The dispatcher deals workgroups round-robin over the 8 XCDs (each with
// a private 4 MiB L2), so neighbouring tiles -- which share input halo rows, whole A tiles (column blocks of one row tile) or
// whole operand panels -- would land on different L2s and every one of them would fetch its operands from HBM again.  The
// remap hands each XCD a CONTIGUOUS range of a linear tile order (y fastest, then z, then x -- the column blocks and the
// transposed-conv phases of one row tile stay together, then come its neighbours): linear id l -> label l % 8 (which
// workgroups share an XCD) -> that label's range.  Bijective for any grid size (cdna_hip_programming.md T1); a speed tool only,
// nothing depends on the placement being as observed.
// ---------------------------------------------------------------------------------
#ifdef __HIPCC__
#define VSTAB_HD __host__ __device__
#else
#define VSTAB_HD
#endif
// the mapping itself (shared with the host-side test hook vstab_host_xcd_remap): dispatch-order id `lin` of a grid Nx x Ny x Nz
// -> tile coordinates
VSTAB_HD inline void xcd_remap_calc(unsigned Nx, unsigned Ny, unsigned Nz, unsigned lin, unsigned &bx, unsigned &by, unsigned &bz)
{
    const unsigned T = Nx * Ny * Nz;
    const unsigned q = T >> 3, r = T & 7u, c = lin & 7u, idx = lin >> 3;
    const unsigned nl = (c < r ? c * (q + 1) : r * (q + 1) + (c - r) * q) + idx;
    // grid.y and grid.z are almost always 1, 2, 4 or 8 here: shifts instead of the ~40-instruction runtime division
    unsigned t2;
    if ((Ny & (Ny - 1)) == 0) { const unsigned s = 31u - (unsigned)__builtin_clz(Ny); by = nl & (Ny - 1); t2 = nl >> s; }
    else { by = nl % Ny; t2 = nl / Ny; }
    if ((Nz & (Nz - 1)) == 0) { const unsigned s = 31u - (unsigned)__builtin_clz(Nz); bz = t2 & (Nz - 1); bx = t2 >> s; }
    else { bz = t2 % Nz; bx = t2 / Nz; }
}
#ifdef __HIPCC__
__device__ __forceinline__ void xcd_remap(unsigned &bx, unsigned &by, unsigned &bz, bool identity = false)
{
    if (identity) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return; }      // A/B switch for tuning runs (VSTAB_NO_XCD_REMAP)
    xcd_remap_calc(gridDim.x, gridDim.y, gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), bx, by, bz);
}
#endif

// ---------------------------------------------------------------------------------
// Implicit-GEMM convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// GEMM view: rows m = output pixels of one "phase" grid (n, j, i), columns = output
// channels, K = KH row-taps x NSEG segments x SEG floats.  A segment is a contiguous
// piece of the NHWC input row starting SEG_STRIDE*s floats after the run start: either
// the whole KW*Cs run (run mode, NSEG = 1; x and c are adjacent in NHWC) or the first
// Cin channels of tap s of a wider pixel (tap mode, NSEG = KW, SEG_STRIDE = Cs).
//   input row   iy  = j*s_in + off_y + t          (t = row tap 0..KH-1)
//   run start   ix0 = i*s_in + off_x              (float offset q in the run -> element
//                                                  ((n*Hi+iy)*Wi+ix0)*Cs_in + q)
//   output      (j*s_out + o_y, i*s_out + o_x), channel c_off + col, pixel stride Cs_out
// A plain conv is one phase (s_in = stride, off = -pad, s_out = 1); a 4x4 stride-2
// transposed conv is four phases of a 2x2-tap conv (s_in = 1, s_out = 2).
// Out-of-image taps and the padded tail of the run read as zero.
// ---------------------------------------------------------------------------------
struct ConvPhase {
    int Hg, Wg, M;          // phase grid and its number of GEMM rows (B*Hg*Wg)
    int off_y, off_x;       // input offsets
    int o_y, o_x;           // output offsets
    int pad_;
    long long w_off;        // float offset of this phase's packed weights inside wpk
    // K layout of THIS phase when it differs from ConvParams' (KH == 0: use the launch-wide one).  The output-parity phases of an
    // odd-k stride-2 input gradient have different tap counts ((k+1)/2 or (k-1)/2 per axis); giving each phase its own reduction
    // length lets all four run in ONE launch with exactly their taps.
    int KH, NSEG, SEG, SEGP, SEG_STRIDE;
    int pad2_;
};

struct ConvParams {
    const float *in;
    const float *wpk;       // packed weights [KT][Npad][32], 16B chunks XOR-swizzled
    const float *bias;      // [Npad] (BatchNorm already folded in)
    float *out;
    float *partial;         // split-K slabs [(phase*ksplit+split)][Mmax][Npad]
    unsigned in_bytes;      // size of the input tensor   (< 2^31: buffer-descriptor range checks)
    unsigned w_bytes;       // size of ONE phase's packed weights
    int B, Hi, Wi, Cs_in;
    int KH;                 // row taps
    int NSEG, SEG, SEGP;    // segments per row tap, floats per segment, SEG rounded up to 32
    int SEG_STRIDE;         // floats between segment starts (run mode: NSEG = 1)
    int s_in, s_out;
    int Ho, Wo, Cs_out, c_off;
    int N, Npad;
    int act;                // 0 = none, 1 = leaky relu 0.1 (max(v, 0.1 v)), 2 = relu, 3 = none, ADD to what is in out
    int nphase, ksplit, Mmax;
    int no_remap;           // tuning switch: keep the dispatch order (VSTAB_NO_XCD_REMAP)
    int stamp_slot;         // -DVSTAB_STAMP diagnostic builds: which stamp buffer this launch writes (launch_conv counts)
    int out_vec4;           // set by launch_conv: out, Cs_out and c_off are 16-byte friendly -> the tile leaves through LDS as 16-byte stores
    ConvPhase ph[16];       // 4 transposed-conv phases, or the 16 positions of a Winograd-domain GEMM
};

#ifdef __HIPCC__
// One work item of the split-K combine (splitk_combine_kernel; also the first workgroups of combine_predict_up_kernel): item idx sums
// the slabs of four consecutive output columns of one GEMM row in slab order, adds the bias, applies the activation, scatters.
__device__ __forceinline__ void splitk_combine_item(const ConvParams &p, const long long idx)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));

    const int n4 = p.N >> 2;
    const long long per_phase = (long long)p.Mmax * n4;
    if (idx >= per_phase * p.nphase) return;
    const int phase = (int)(idx / per_phase);
    const long long rem = idx - phase * per_phase;
    const int m = (int)(rem / n4), c4 = (int)(rem - (long long)m * n4);
    const ConvPhase ph = p.ph[phase];
    if (m >= ph.M) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float *src = p.partial + (((long long)(phase * p.ksplit) * p.Mmax + m) * p.Npad + c4 * 4);
    const long long slab = (long long)p.Mmax * p.Npad;
    int k = 0;
    for (; k + 8 <= p.ksplit; k += 8) {           // eight slab loads in flight, added in slab order
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + (k + u) * slab);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    if (k + 4 <= p.ksplit) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + (k + u) * slab);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u];
        k += 4;
    }
    if (k < p.ksplit) {                           // up to three left: load all, add in order
        f32x4 v[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) v[u] = k + u < p.ksplit ? *reinterpret_cast<const f32x4 *>(src + (k + u) * slab) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (k + u < p.ksplit) s += v[u];
    }
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + c4 * 4);
    s += bv;
    if (p.act == 1 || p.act == 2) {
        const float slope = p.act == 1 ? 0.1f : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = fmaxf(s[e], slope * s[e]);
    }
    const int hw = ph.Hg * ph.Wg;
    const int n = m / hw, r2 = m - n * hw;
    const int j = r2 / ph.Wg, i = r2 - j * ph.Wg;
    const long long oo = ((long long)(n * p.Ho + j * p.s_out + ph.o_y) * p.Wo + i * p.s_out + ph.o_x) * p.Cs_out + p.c_off;
    if (p.act == 3) s += *reinterpret_cast<const f32x4 *>(p.out + oo + c4 * 4);         // accumulate (gradient sums)
    *reinterpret_cast<f32x4 *>(p.out + oo + c4 * 4) = s;
}
#endif

enum ConvTile { TILE_128x128 = 0, TILE_128x64 = 1, TILE_128x32 = 2, TILE_64x128 = 3, TILE_64x64 = 4, TILE_256x32 = 5,
                TILE_SKINNY = 6 };     // conv_skinny.hip: 32/64 rows x 32 columns per workgroup, weights streamed through registers

// Launch the implicit-GEMM kernel (and the split-K combine when p.ksplit > 1).
// ev_start/ev_stop (optional) are recorded immediately around the GEMM kernel itself.
// combine = false: a split-K launch leaves its slabs in p.partial for the consumer to sum (launch_predict_up)
hipError_t launch_conv(const ConvParams &p, ConvTile tile, bool vec4, hipStream_t stream,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, bool combine = true);
// two problems in one launch (conv_dual_kernel): A on a 128x128 / 128x64 / 64x128 tile, B a 128x32 tap-table GEMM; neither's slabs are
// combined.  hipErrorNotSupported when the pair is not one the kernel is built for.
hipError_t launch_conv_dual(const ConvParams &pa, ConvTile tile_a, const ConvParams &pb, ConvTile tile_b, hipStream_t stream,
                            hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t conv_set_attributes();   // raises the dynamic-LDS limit once per process
bool conv_uses_lds_dma(ConvTile tile, bool vec4);   // which instantiation launch_conv picks (for reports)

// ---------------------------------------------------------------------------------
// Few-row layers as weight streams (conv_skinny.hip): the same ConvParams (phases, K layout, packed weights, slabs), a
// 32 x 32 tile per workgroup whose four waves split the workgroup's K slice, fragments loaded straight into registers,
// and the split-K reduction finished inside the launch by the workgroup that draws a tile's last ticket.
// ---------------------------------------------------------------------------------
constexpr int SKINNY_MAX_ROWS = 64;         // GEMM rows per phase up to which a layer is a weight stream rather than a tiled GEMM (measured: profiles/README.md r04)
constexpr int SKINNY_MAX_TILES = 4096;      // ticket words a context holds (vstab_create)
bool conv_skinny_applicable(const ConvParams &p, bool vec4);
long long skinny_tiles(const ConvParams &p);
int conv_skinny_split(const ConvParams &p, int cap = 16);
// counters: SKINNY_MAX_TILES zeroed words that are zero again when the launch has finished (needed when p.ksplit > 1)
hipError_t launch_conv_skinny(const ConvParams &p, unsigned *counters, hipStream_t stream, hipEvent_t ev_start = nullptr,
                              hipEvent_t ev_stop = nullptr);

// ---------------------------------------------------------------------------------
// Row-window convolution (conv_rowwin.hip): first layer, one output-row segment of 128
// pixels per workgroup, operands read from a contiguous LDS copy of the input row window.
// ---------------------------------------------------------------------------------
struct RowWinParams {
    const float *in;
    const float *wpk;       // packed [KH*SEGP/32][Npad][32], rowwin k-permutation, no swizzle
    const float *bias;
    float *out;
    unsigned in_bytes;      // size of the input tensor (buffer descriptor range)
    int B, Hi, Wi, Cs_in;
    int KH, SEGP;           // filter rows; extended run (lead dummies + KW*Cs) padded to 32
    int s_in, off_y;
    int e_off;              // extended run start = ox*s_in*Cs + e_off (floats from the row start), even
    int w_a;                // window lead (0 or 2): window start = s_in*Cs*ox0 + e_off - w_a, multiple of 4
    int WLEN;               // window length in floats (multiple of 4)
    int Ho, Wo, Cs_out, c_off;
    int N, Npad, act;
    int MB;                 // 2: 128-pixel tiles; 1: 64-pixel tiles (small launches)
    int out_vec4;           // set by launch_conv_rowwin: the tile leaves through LDS as 16-byte stores
    int asm_loop;           // set by launch_conv_rowwin: the K loop runs as the assembly block of conv_kloop_gfx950.inc (128-pixel tiles, 6 K-tiles per filter row)
    int stream_rows;        // set by launch_conv_rowwin: > 0 = a workgroup walks down this many consecutive output rows as one seamless stream of tiles (grid.x = Ho / stream_rows)
    unsigned *clear_words;  // clear_n > 0: the launch's first workgroup zeroes these words (the forward's split-K tickets; the launches that
    int clear_n;            // use them come later in the stream)
    int ox_base, ntile_x;   // first output column and number of x tiles of this launch (0 tiles = up to the row end): a row whose length
                            // is 128 k + (1..64) runs as k 128-pixel tiles plus ONE 64-pixel tile in a second launch instead of a
                            // half-empty 128-pixel one (Wo = 960 at 1080p: 6 % of the first layer's MFMA work)
};
// tile height of the row-window kernel for a launch of Ho x Wo x B output pixels: 64-pixel tiles while 128-pixel ones would not
// give every CU two workgroups
inline int rowwin_mb(int B, int Ho, int Wo) { return (long long)B * Ho * ((Wo + 127) / 128) < 512 ? 1 : 2; }
bool rowwin_applicable(const RowWinParams &p);
hipError_t rowwin_set_attributes();
hipError_t launch_conv_rowwin(const RowWinParams &p, hipStream_t stream, hipEvent_t ev_start = nullptr,
                              hipEvent_t ev_stop = nullptr);
// lead dummy floats d (0/1) making off_x*Cs - d even, and the padded extended run length
inline int rowwin_lead(int off_x, int cs) { return ((off_x * cs) % 2 != 0) ? 1 : 0; }
inline int rowwin_segp(int off_x, int kw, int cs) { return round_up_c(rowwin_lead(off_x, cs) + kw * cs, 32); }
// Conv weights W[kh][kw][Cin][Cout] -> rowwin packed layout
void pack_conv_rowwin(const float *W, const double *scale, int kh, int kw, int cin, int cout, int npad,
                      int lead, int segp, float *wpk);

// ---------------------------------------------------------------------------------
// Winograd F(2x2,3x3) transforms around the MFMA kernel (winograd_ops.hip) and the weight transform + packing:
// U_xi = (G g G^T)_xi with the BatchNorm scale folded in, 16 blocks of a 1x1-conv operand (klayout_run(1,1,cin)).
// ---------------------------------------------------------------------------------
// filter gradient of a 3x3 stride-1 layer in the Winograd domain (winograd_ops.hip): dM = A dY A^T per 2x2 tile of the output
// gradient -> [B][16][TH*TW][C]; dg = G^T dU G from the 16 position gradients dU [16][cin][cout] -> HWIO [3][3][cin][cout]
hipError_t launch_wino_outgrad(const float *dy, int B, int H, int W, int Cs, int c_off, int C, float *dM, hipStream_t stream);
hipError_t launch_wino_filter_grad(const float *dU, int cin, int cout, float *dW, hipStream_t stream);
// the 16 Winograd-domain GEMMs of a stage as streams of positions (wino_gemm_stream.hip): positions per workgroup for B samples x T
// tiles, or 0 when the stage does not qualify (the 16-phase launch_conv then)
int wino_gemm_stream_positions(int B, int T, int cin, int cout);
hipError_t launch_wino_gemm_stream(const float *V, const float *wpk, float *M, int B, int T, int cin, int cout, int P, hipStream_t stream,
                                   hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t wino_gemm_stream_set_attributes();
hipError_t launch_wino_input(const float *x, int B, int H, int W, int Cs, int c_off, int C, float *V, hipStream_t stream,
                             bool pos_major = false);       // pos_major: V as [16][B*tiles][C] instead of [B][16][tiles][C]
// device-side weight transform for training: Wt [16][K][N] from W [3,3,cin,cout]; transpose = the input gradient's operand
hipError_t launch_wino_weights(const float *W, int cin, int cout, int transpose, float *Wt, hipStream_t stream);
hipError_t launch_wino_output(const float *M, int B, int Ho, int Wo, int C, const float *bias, int act, float *out, int Cs_out, int c_off,
                              hipStream_t stream);

// Winograd F(2x2,2x2) form of a 4x4 stride-2 SAME transposed convolution (winograd_ops.hip documents the algebra): tile grid and, per
// position row / column, how many tiles are not identically zero
struct WdecGeom {
    int NTy, NTx;           // tiles per axis: Ho/4 + 1, Wo/4 + 1
    int nty[3], ntx[3];     // tiles position row i / column j needs (the rest is zero)
};
inline WdecGeom wdec_geom(int Hi, int Wi, int Ho, int Wo)
{
    WdecGeom g;
    g.NTy = Ho / 4 + 1; g.NTx = Wo / 4 + 1;
    g.nty[0] = g.NTy < Hi / 2 + 1 ? g.NTy : Hi / 2 + 1; g.nty[1] = g.nty[2] = g.NTy < (Hi + 1) / 2 ? g.NTy : (Hi + 1) / 2;
    g.ntx[0] = g.NTx < Wi / 2 + 1 ? g.NTx : Wi / 2 + 1; g.ntx[1] = g.ntx[2] = g.NTx < (Wi + 1) / 2 ? g.NTx : (Wi + 1) / 2;
    return g;
}
struct WdecOutArgs { const float *M; int C4; const float *bias; int act; float *out; int Ho, Wo, Cs_out, c_off; WdecGeom g; };
#ifdef __HIPCC__
// One work item of the inverse transform (wdec_output_kernel; also the first workgroups of wdec_output_predict_up_kernel): item idx of
// sample n = (tile, phase, four output channels): nine M values -> the phase's 2 x 2 outputs, + bias, activation, stores
__device__ __forceinline__ void wdec_output_item(const WdecOutArgs &A, const long long idx, const int n)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const WdecGeom &g = A.g;
    const int C4 = A.C4;
    if (idx >= (long long)g.NTy * g.NTx * 4 * C4) return;
    const int c = (int)(idx % C4);
    const int ph = (int)((idx / C4) & 3);
    const int tile = (int)(idx / (4 * C4));
    const int ty = tile / g.NTx, tx = tile - ty * g.NTx;
    const int py = ph >> 1, px = ph & 1;
    const int N = 16 * C4;                           // 4 phases x Cout floats per GEMM row
    const long long plane = (long long)g.NTy * g.NTx * N;
    const float *mb = A.M + (long long)n * 9 * plane + (long long)tile * N + ph * (4 * C4) + c * 4;
    f32x4 m[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ty < g.nty[i] && tx < g.ntx[j]) v = *reinterpret_cast<const f32x4 *>(mb + (i * 3 + j) * plane);
            m[i][j] = v;
        }
    f32x4 s[2][3];                                  // rows: (m0 + m1, m1 - m2)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        s[0][j] = m[0][j] + m[1][j];
        s[1][j] = m[1][j] - m[2][j];
    }
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(A.bias + c * 4);
    const float slope = A.act == 1 ? 0.1f : 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = 4 * ty + 2 * a - py;          // even phase: 4t, 4t+2; odd phase: 4t-1, 4t+1
        if ((unsigned)oy >= (unsigned)A.Ho) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ox = 4 * tx + 2 * b - px;
            if ((unsigned)ox >= (unsigned)A.Wo) continue;
            f32x4 y = (b == 0 ? s[a][0] + s[a][1] : s[a][1] - s[a][2]) + bv;
            if (A.act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], slope * y[e]);
            }
            *reinterpret_cast<f32x4 *>(A.out + (((long long)n * A.Ho + oy) * A.Wo + ox) * A.Cs_out + A.c_off + c * 4) = y;
        }
    }
}
#endif
hipError_t launch_wdec_input(const float *x, int B, int Hi, int Wi, int Cs, float *V, const WdecGeom &g, hipStream_t stream);
hipError_t launch_wdec_output(const float *M, int B, int Ho, int Wo, int cout, const float *bias, int act, float *out, int Cs_out, int c_off,
                              const WdecGeom &g, hipStream_t stream);

// ---------------------------------------------------------------------------------
// Small VALU kernels of the flow pyramid and the warp.
// ---------------------------------------------------------------------------------
// predict_flowN + upsample_flowN in one launch.  src: the tap table T[B,h,w,32] (T[.., tap*2 + o] = sum_c x[.., c]
// W[tap][c][o], computed by the MFMA kernel as a 1x1 conv) or its `ks` split-K slabs `slab_stride` floats apart;
// out = gather(T) + bias [+ (.. + u) + u, u = legacy-bilinear upsample of the coarser flow `prev`]
// (model.py:848,856-857,865-866,874-875); then the 4x4 s2 SAME transposed conv 2->2 + bias of `out`, written as
// (u, v, 0, 0) into the four floats at c_off of the next level's concat pixel (model.py:852,861,870,879).
struct UpflowW { float w[64]; float b[2]; };   // w[ky][kx][co][ci]
hipError_t launch_predict_up(const float *src, int ks, long long slab_stride, int B, int h, int w, const float *bias2,
                             const float *prev, int ph, int pw, float *out, const UpflowW &W, float *concat, int oh, int ow,
                             int Cs, int c_off, hipStream_t stream, const ConvParams *combine = nullptr, const WdecOutArgs *wdec = nullptr);
// wdec != null: the same launch also runs the inverse Winograd transform of the level's transposed convolution (wdec_output_item)
// combine != null (with ksplit > 1): the same launch also sums that launch's split-K slabs into its output (the level's transposed
// convolution, launched with combine = false): combine_predict_up_kernel

// predict_flow2 (model.py:882-887) from the per-source-pixel tap table T[B,h2,w2,32]
// (T[.., tap*2 + o] = sum_c concat2[.., c] * W[tap][c][o]) and pf3.
// predict_flow2's tap table as one burst-loaded panel per 64 pixel rows (tap_panel.hip): x [M][196], wp [200][32], T [M][32]
bool tap_panel_applicable(long long M, int cs_in, const void *x, const void *T);
hipError_t launch_tap_panel(const float *x, long long M, const float *wp, float *T, hipStream_t stream, hipEvent_t ev_start = nullptr,
                            hipEvent_t ev_stop = nullptr);
hipError_t tap_panel_set_attributes();
hipError_t launch_pf2(const float *T, int B, int h2, int w2, const float *bias2, const float *pf3,
                      int h3, int w3, float *pf2, int H, int W, hipStream_t stream);

hipError_t launch_flow_resize_scale(const float *flow, int B, int h, int w, float *out, int oh,
                                    int ow, int net_h, int net_w, hipStream_t stream);
hipError_t launch_resize_bilinear(const float *x, int B, int h, int w, int C, float *out, int oh,
                                  int ow, hipStream_t stream);
hipError_t launch_resize_bilinear_slice3(const float *x, int B, int h, int w, int Cs, int c_off, float *out, int oh, int ow,
                                         hipStream_t stream);
hipError_t launch_warp_flow(const float *img, const float *flow, float *out, int B, int H, int W,
                            int C, hipStream_t stream);
// per-launch timing of the HBM-side kernels (flow_ops.hip): slots of vstab_hbm_profile_read
enum { HBM_SLOT_WARP = 0, HBM_SLOT_GLUE = 1, HBM_SLOT_GLUE_WARP = 2, HBM_SLOT_PF2 = 3, HBM_SLOT_ST = 4, HBM_SLOT_HOMOG = 5, HBM_SLOT_TAIL = 6, HBM_SLOTS = 7 };
void hbm_profile_enable(int mode);
hipError_t hbm_profile_read(int slot, double *ms_sum, int *launches, double *alg_bytes_sum);
hipError_t launch_div_const_selftest(float d, unsigned first, unsigned long long count, unsigned long long *bad, hipStream_t stream);
// predict_flow2 gather + flow glue + tf_warp in ONE launch (flow_ops.hip); hipErrorNotSupported = run the two launches instead
hipError_t launch_pf2_glue_warp(const float *T, int B, int h2, int w2, const float *bias2, const float *pf3, int h3, int w3, float *pf2, int H, int W,
                                const float *img, float *outflow, float *out, int oh, int ow, hipStream_t stream);
hipError_t launch_pf2_glue_warp_u8(const float *T, int B, int h2, int w2, const float *bias2, const float *pf3, int h3, int w3, float *pf2, int H, int W,
                                   const unsigned char *frame, float *outflow, unsigned char *out, int oh, int ow, hipStream_t stream);
hipError_t launch_flow_glue_warp(const float *flow, int B, int h, int w, const float *img, float *outflow, float *out, int oh, int ow,
                                 int C, int net_h, int net_w, hipStream_t stream);
// 2x2 stride-2 SAME max pool (vgg16.py:51-53), NHWC, C % 4 == 0
hipError_t launch_maxpool2x2(const float *x, int B, int H, int W, int C, float *out, hipStream_t stream);
// 3x3 SAME conv on a 3-channel image (VGG16 conv1_1): W HWIO [3][3][3][cout] unpacked, cout % 4 == 0, cout <= 64
hipError_t launch_conv3x3_rgb(const float *x, int B, int H, int W, const float *Wf, const float *bias, int cout, int relu, float *out,
                              hipStream_t stream);
// y = x * scale - mean[c]  (NLDF.py:29 preprocessing), C <= 4
hipError_t launch_scale_shift(const float *x, long long npix, int C, float scale, const float *mean4, float *out,
                              hipStream_t stream);
hipError_t launch_get_pixel_value(const float *img, const int32_t *x, const int32_t *y, float *out,
                                  int B, int H, int W, int C, int Hi, int Wi, hipStream_t stream);

// ---------------------------------------------------------------------------------
// Secondary samplers (sampler_ops.hip): spatial_transformer.py / warp.py rows S1-S3.
// ---------------------------------------------------------------------------------
hipError_t launch_st_interp(const float *img, int B, int H, int W, int C, const float *x, const float *y,
                            int oh, int ow, float *out, hipStream_t stream);
hipError_t launch_st_transform(const float *img, int B, int H, int W, int C, const float *theta, int tdim,
                               float *out, int oh, int ow, hipStream_t stream);
hipError_t launch_st_meshgrid(float *out, int oh, int ow, hipStream_t stream);
// ref != null: M holds pMtrx [B,9] and the kernels compose refMtrx . pMtrx themselves (warp.py:48-49)
hipError_t launch_homography_warp(const float *img, int B, int Hi, int Wi, int C, const float *M, float *out,
                                  int oh, int ow, hipStream_t stream, const float *ref = nullptr);
hipError_t launch_vec2mtrx(const float *p, int B, int dim, int approx, float *out, hipStream_t stream);

// clip driver helpers (clip_ops.hip)
hipError_t launch_resize_u8(const unsigned char *src, int B, int sh, int sw, unsigned char *dst, int dh, int dw, hipStream_t stream);
hipError_t launch_resize_f32_to_u8(const float *src, int B, int sh, int sw, unsigned char *dst, int dh, int dw, hipStream_t stream);
hipError_t launch_assemble_input(const unsigned char *const *slots9, int B, int h, int w, float *feats, hipStream_t stream);
// the same with the current frame's cv2.resize inside: slots8[j] == null reads the resized current frame (a clip's first frame)
hipError_t launch_assemble_input_resized(const unsigned char *const *slots8, const unsigned char *frame, int B, int h, int w, int sh, int sw,
                                         float *feats, hipStream_t stream);
// frame_to_float + flow glue + tf_warp + quantise_output in one launch on 8-bit frames (flow_ops.hip)
hipError_t launch_flow_glue_warp_u8(const float *flow, int B, int h, int w, const unsigned char *frame, float *outflow, unsigned char *out, int oh,
                                    int ow, int net_h, int net_w, hipStream_t stream);
hipError_t launch_frame_to_float(const unsigned char *f, long long npix, float *out, hipStream_t stream);
hipError_t launch_quantise_output(const float *warped, long long npix, unsigned char *out, hipStream_t stream);

hipError_t launch_flow_box_blur(const float *flow, int B, int h, int w, int k, float *tmp, float *out, hipStream_t stream);
hipError_t launch_axpby(const float *x, float a, const float *y, float b, float *out, long long n, hipStream_t stream);
hipError_t launch_flow_mean_fill(const float *flow, int B, int h, int w, float *out, hipStream_t stream);
// ---------------------------------------------------------------------------------
// Convolution weight gradient on the MFMA (wgrad_mfma.hip): dW[(ky,kx,ci)][co] = sum over output pixels of
// x[n, s*oy+ky-p, s*ox+kx-p, ci] * g[n,oy,ox,co]; reduction index = output pixel.
// ---------------------------------------------------------------------------------
struct WgradParams {
    const float *x;         // [B,Hi,Wi,Cs_x], channels cx_off .. cx_off+Cin
    const float *g;         // [B,Ho,Wo,Cs_g] = [K][Cs_g], channels cg_off .. cg_off+Cout
    const int4 *ptab;       // [K] pixel table (launch_wgrad_pixel_table)
    float *dW;              // [KH*KW*Cin][Cout] (HWIO row-major)
    float *partial;         // split-K slabs [ksplit][M][Cout]
    unsigned x_bytes, g_bytes;
    int Hi, Wi, Cs_x, cx_off, Cin;
    int KH, KW;
    int Cs_g, cg_off, Cout;
    int M, K;               // KH*KW*Cin, B*Ho*Wo
    int ksplit, accumulate; // accumulate: dW += result
    // batched form (nbatch > 1): independent reductions z = 0..nbatch-1 over the same pixel table, operands x + z*x_bstride and
    // g + z*g_bstride (elements; x_bytes / g_bytes then size ONE batch), result dW + z*M*Cout.  The 16 positions of a
    // Winograd-domain filter gradient are such a batch.  0 or 1: the plain form.
    int nbatch;
    long long x_bstride, g_bstride;
    // column window: this launch computes columns col0 .. col0+Cout of a dW whose rows are ldw floats apart (0: ldw = Cout, col0 = 0).
    // Lets a layer with 128 k + few output columns (the concat inputs: 1026, 770, 386 channels) run its 128-wide part and its narrow
    // tail as two launches instead of paying a whole 128-column tile for the tail.
    int ldw, col0;
};
hipError_t launch_wgrad_pixel_table(int B, int Hi, int Wi, int Cs, int Ho, int Wo, int s, int pad, int4 *ptab, hipStream_t stream);
int wgrad_choose_split(const WgradParams &p);
hipError_t launch_wgrad(const WgradParams &p, hipStream_t stream);
// out[c] (+)= sum over rows of g[row*Cs + c_off + c]
// chunk count of the two-stage column reductions: enough (64-channel block) x (row chunk) workgroups to fill the chip
// (about three per CU), at least 128 rows per chunk
inline int reduce_chunks(long long rows, int C)
{
    const long long colblocks = C >= 64 ? (C + 63) / 64 : 1;
    long long n = 768 / colblocks;
    if (n < 8) n = 8;
    const long long by_rows = rows / 128;
    if (n > by_rows) n = by_rows;
    return (int)(n < 1 ? 1 : n);
}
int column_sum_chunks(long long rows, int C);  // scratch floats needed = column_sum_chunks(rows, C) * C
hipError_t launch_column_sum(const float *g, long long rows, int Cs, int c_off, int C, float *out, int accumulate, float *scratch,
                             hipStream_t stream);

// lossterm / masked_MSE / total_variation of one pyramid level and their gradient w.r.t. the flow (train_ops.hip)
hipError_t launch_loss_level(const float *pf, const float *G, const float *U, int B, int h, int w, double *sums, float scale_mse,
                             float scale_tv, float *grad, hipStream_t stream);
struct LossLevel { const float *pf; float *grad; int h, w, cs_pf, cs_grad; float tv_weight; };     // = vstab_loss_level_desc
size_t loss_main_workspace_bytes(int B, const LossLevel *lv, int n);
hipError_t launch_loss_main(const LossLevel *lv, int n, const float *gtstab, const float *unstab, int B, int H, int W, double *loss_out,
                            void *workspace, hipStream_t stream);
hipError_t launch_flow_medfilt(const float *flow, int B, int h, int w, int kh, int kw, int kc, float *out, hipStream_t stream);

// dense-flow homography fit + cv2.warpPerspective on 8-bit frames (homography_ops.hip)
int homography_moment_blocks(int H, int W);
size_t homography_workspace_bytes(int B, int H, int W, int K);
hipError_t launch_homography_fit(const float *flow, int B, int H, int W, int K, unsigned seed, double thresh, int refine,
                                 int stride, double *Hout, int *inliers, void *ws, hipStream_t stream);
hipError_t launch_warp_perspective_u8(const unsigned char *src, int B, int sh, int sw, const double *Hm, unsigned char *dst,
                                      int oh, int ow, hipStream_t stream);

// NLDF head helpers (nldf_ops.hip)
hipError_t launch_contrast(float *buf, int B, int H, int W, int C, int Cs, int c_dst, hipStream_t stream);
hipError_t launch_nldf_score(const float *local2, const float *global2, int B, int npix, float *score, float *prob,
                             hipStream_t stream);

// ---------------------------------------------------------------------------------
// Host-side weight packing (pack.cpp): pure CPU code.
// ---------------------------------------------------------------------------------
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// Physical position of logical float k (0..31) of row n inside a 32-float LDS/packed row:
// 16-byte chunks are XOR-swizzled with ((n>>1)&7) so that ds_read_b128 of 16 different
// rows is bank-conflict free.
inline int swz32(int n, int k) { return ((((k >> 2) ^ ((n >> 1) & 7)) << 2) | (k & 3)); }

// K layout of one conv GEMM (mirrors the fields of ConvParams)
struct KLayout {
    int KH, NSEG, SEG, SEGP, SEG_STRIDE;
    int ktiles() const { return KH * NSEG * (SEGP / 32); }
};
KLayout klayout_run(int kh, int kw, int cs_in);            // whole KW*Cs run per row tap
KLayout klayout_tap(int kh, int kw, int cin, int cs_in);   // first cin channels of each tap

// BatchNorm fold (inference, no gamma): scale[n] = rsqrt(var+eps), bias' = (b-mean)*scale+beta.
// beta == nullptr means "no BatchNorm".  Outputs have npad entries (zero padded).
void fold_bn(const float *b, const float *beta, const float *mean, const float *var, int cout,
             int npad, double *scale, float *bias_out);

// Conv weights W[kh][kw][Cin][Cout] -> packed [KT][Npad][32] for the given K layout.
// cs_in = channel stride of the input buffer (>= cin; extra channels get zero weights).
void pack_conv(const float *W, const double *scale, int kh, int kw, int cin, int cs_in, int cout,
               int npad, const KLayout &L, float *wpk);

// Transposed-conv weights W[4][4][Cout][Cin] -> 4 phase matrices (phase = py*2+px), each a
// 2-row-tap conv in run mode (run = 2*cs_in); phase p starts at p*phase_floats.
void pack_deconv(const float *W, const double *scale, int cin, int cs_in, int cout, int npad,
                 float *wpk);
KLayout klayout_deconv(int cs_in);

// 5x5 stride-2 SAME transposed conv W[5][5][Cout][Cin] (NLDF.py:57-64) -> 4 phase matrices of a 3-row-tap conv
// in run mode (run = 3*cs_in); the even phases use 2 of the 3 taps (the third gets zero weights).
void pack_deconv5(const float *W, const double *scale, int cin, int cs_in, int cout, int npad, float *wpk);

// 3x3 conv W[3][3][Cin][Cout] -> 16 Winograd-domain 1x1 operands (position xi = 4 i + j), phase xi starts at xi * ktiles * npad * 32
void pack_winograd(const float *W, const double *scale, int cin, int cout, int npad, float *wpk);

// 4x4 stride-2 transposed conv W[4][4][Cout][Cin] -> 9 Winograd F(2x2,2x2)-domain 1x1 operands (position 3 i + j), each [cs_in -> 4 cout]
// with column = phase * cout + co (phase = 2 py + px); position p starts at p * ktiles * (4 cout) * 32
void pack_wdec(const float *W, const double *scale, int cin, int cs_in, int cout, float *wpk);

// predict head W[3][3][Cin][2] -> tap-table weights: 1x1 conv (run mode over cs_in) with 18 (pad npad)
// output columns, col = tap*2 + o
void pack_predict2_table(const float *W, int cin, int cs_in, int npad, float *wpk);
void pack_predict2_panel(const float *W, int cin, int kpad, float *wp);        // [kpad][32] for tap_panel_kernel

// BatchNormLayer(lrelu 0.1, no gamma) in training mode, in place on an NHWC channel slice, and its backward (train_ops.hip).
// scratch: 2 * bn_chunks(rows, C) * C floats.
int bn_chunks(long long rows, int C);
hipError_t launch_bn_lrelu_train_forward(float *zy, long long rows, int cs, int c_off, int C, const float *beta, float *mov_mean,
                                         float *mov_var, float decay, float eps, float *save_mean, float *save_rstd, float *scratch,
                                         hipStream_t stream);
hipError_t launch_bn_lrelu_train_backward(const float *y, int cs_y, int cy_off, float *dy, int cs_g, int cg_off, int C, long long rows,
                                          const float *beta, const float *save_rstd, float *dbeta, int accumulate, float *scratch,
                                          hipStream_t stream);
hipError_t launch_lrelu_backward(const float *y, int cs_y, int cy_off, float *dy, int cs_g, int cg_off, int C, long long rows,
                                 hipStream_t stream);

// adjoint of the legacy bilinear resize (din (+)= gain * J^T dout), the full-resolution head's pad + nearest upsampler and its
// adjoint, Adam (train_ops.hip)
hipError_t launch_resize_bilinear_backward(const float *dout, int B, int oh, int ow, int C, float *din, int h, int w, float gain,
                                           int accumulate, hipStream_t stream);
hipError_t launch_pad_nearest_up(const float *src, int B, int h2, int w2, int C, float *out, int H, int W, hipStream_t stream);
hipError_t launch_pad_nearest_up_backward(const float *dout, int B, int H, int W, int C, float *dsrc, int h2, int w2, int accumulate,
                                          hipStream_t stream);
// adjoint of pf2_kernel's tap gather: dT [B,h2,w2,32] (col = tap*2+o) from the flow gradient g [B,H-2,W-2,cs_g]
hipError_t launch_pf2_taps_backward(const float *g, int cs_g, int B, int H, int W, float *dT, int h2, int w2, hipStream_t stream);
hipError_t launch_adam(float *w, const float *g, float *m, float *v, long long n, float lr_t, float b1, float b2, float eps,
                       hipStream_t stream);

// Index tables for device-side packing (training): tbl[i] = 1 + raw-weight index of packed element i, 0 = zero.
void pack_index_conv(int kh, int kw, int cin, int cs_in, int cout, int npad, const KLayout &L, int32_t *tbl);
void pack_index_dgrad_s1(int k, int cin, int cout, int cs_g, const KLayout &L, int npad, int32_t *tbl);
void pack_index_dgrad_s2(int k, int pad, int cin, int cout, int cs_g, const KLayout &L, int npad, int32_t *tbl);   // 4 phases of ceil(k/2)^2 taps
void pack_index_phase(int k, int cin, int cout, int cs_g, const KLayout &L, int npad, int t0y, int nty, int t0x, int ntx, int32_t *tbl);
// wpk[i] = tbl[i] ? W[tbl[i]-1] : 0
// n_fast: the table's sources are contiguous along the output column (see pack_apply_nfast_kernel); rows of the packed operand are numbered
// (K-tile, column), so "row number mod 16" is "column mod 16" only when Npad is a multiple of 16 (it is a multiple of 32 everywhere)
hipError_t launch_pack_apply(const float *W, const int32_t *tbl, long long n, float *wpk, hipStream_t stream, bool n_fast = false);
// table-free packing of a plain [batch][K][N] matrix (K % 32 == 0, N % 4 == 0, Npad % 64 == 0) into [batch][K/32][Npad][32]
hipError_t launch_pack_blocked(const float *W, int K, int N, int Npad, int batch, float *wpk, hipStream_t stream);

}  // namespace vstab
