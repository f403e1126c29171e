// Host-side weight packing: reference layouts -> the K-tiled, swizzled operand layout the
// implicit-GEMM kernel streams.  Pure CPU code (unit-tested without a GPU).
//
// Packed layout: [KT][Npad][32] floats.  K-tile kt = ((t*NSEG + s)*SEGP/32 + kc) holds, for
// output column n, the 32 weights multiplying input floats q = kc*32 .. +31 of segment s of
// row tap t; inside the 32-float row the 16-byte chunks are XOR-swizzled by ((n>>1)&7)
// (swz32) so that the kernel's global->LDS copy of a B tile is linear.
#include <cmath>
#include <cstring>

#include "vstab_internal.h"

namespace vstab {

KLayout klayout_run(int kh, int kw, int cs_in)
{
    KLayout L{kh, 1, kw * cs_in, round_up(kw * cs_in, 32), 0};
    return L;
}
KLayout klayout_tap(int kh, int kw, int cin, int cs_in)
{
    KLayout L{kh, kw, cin, round_up(cin, 32), cs_in};
    return L;
}
KLayout klayout_deconv(int cs_in) { return klayout_run(2, 2, cs_in); }

void fold_bn(const float *b, const float *beta, const float *mean, const float *var, int cout, int npad,
             double *scale, float *bias_out)
{
    for (int n = 0; n < npad; ++n) {
        double s = 1.0, bb = 0.0;
        if (n < cout) {
            bb = b ? (double)b[n] : 0.0;
            if (beta) {
                s = 1.0 / std::sqrt((double)var[n] + 1e-5);
                bb = (bb - (double)mean[n]) * s + (double)beta[n];
            }
        }
        scale[n] = s;
        bias_out[n] = (float)bb;
    }
}

// generic fill: wfn(t, kx, ci, n) -> weight, kx/ci derived from the absolute run offset
template <class F>
static void pack_generic(const KLayout &L, int cs_in, int cin, int kw, int cout, int npad, F wfn, float *wpk)
{
    const int kps = L.SEGP / 32;
    std::memset(wpk, 0, sizeof(float) * (size_t)L.ktiles() * npad * 32);
    for (int t = 0; t < L.KH; ++t)
        for (int s = 0; s < L.NSEG; ++s)
            for (int q = 0; q < L.SEG; ++q) {
                const int qa = s * L.SEG_STRIDE + q;
                const int kx = qa / cs_in, ci = qa % cs_in;
                if (kx >= kw || ci >= cin) continue;
                const size_t kt = (size_t)(t * L.NSEG + s) * kps + q / 32;
                float *row = wpk + kt * npad * 32;
                for (int n = 0; n < cout; ++n) row[(size_t)n * 32 + swz32(n, q & 31)] = wfn(t, kx, ci, n);
            }
}

void pack_conv(const float *W, const double *scale, int kh, int kw, int cin, int cs_in, int cout, int npad,
               const KLayout &L, float *wpk)
{
    (void)kh;
    pack_generic(L, cs_in, cin, kw, cout, npad,
                 [&](int t, int kx, int ci, int n) {
                     return (float)((double)W[(((size_t)t * kw + kx) * cin + ci) * cout + n] * scale[n]);
                 },
                 wpk);
}

// rowwin layout: physical position of logical k (0..31) inside a 32-float row so that lane half h's
// float4 number q, element e feeds MFMA step 4q+e with k = 8q + 4(e>>1) + 2h + (e&1)  (conv_rowwin.hip)
static inline int perm_rowwin(int k)
{
    const int q = k >> 3, rem = k & 7, h = (rem >> 1) & 1, e = ((rem >> 2) << 1) | (rem & 1);
    return (2 * q + h) * 4 + e;
}

void pack_conv_rowwin(const float *W, const double *scale, int kh, int kw, int cin, int cout, int npad, int lead,
                      int segp, float *wpk)
{
    const int kpr = segp / 32;
    std::memset(wpk, 0, sizeof(float) * (size_t)kh * kpr * npad * 32);
    for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx)
            for (int ci = 0; ci < cin; ++ci) {
                const int k = lead + kx * cin + ci;
                float *row = wpk + ((size_t)ky * kpr + k / 32) * npad * 32;
                for (int n = 0; n < cout; ++n)
                    row[(size_t)n * 32 + perm_rowwin(k & 31)] =
                        (float)((double)W[(((size_t)ky * kw + kx) * cin + ci) * cout + n] * scale[n]);
            }
}

void pack_deconv(const float *W, const double *scale, int cin, int cs_in, int cout, int npad, float *wpk)
{
    const KLayout L = klayout_deconv(cs_in);
    const size_t phase_floats = (size_t)L.ktiles() * npad * 32;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            // input row iy = j + py - 1 + t  <->  ky = 3 - py - 2t   (oy = 2*iy + ky - 1 = 2j + py)
            pack_generic(L, cs_in, cin, 2, cout, npad,
                         [&](int t, int bx, int ci, int n) {
                             const int ky = 3 - py - 2 * t, kx = 3 - px - 2 * bx;
                             return (float)((double)W[(((size_t)ky * 4 + kx) * cout + n) * cin + ci] * scale[n]);
                         },
                         wpk + (size_t)(py * 2 + px) * phase_floats);
        }
}

void pack_deconv5(const float *W, const double *scale, int cin, int cs_in, int cout, int npad, float *wpk)
{
    const KLayout L = klayout_run(3, 3, cs_in);
    const size_t phase_floats = (size_t)L.ktiles() * npad * 32;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            // oy = 2*iy + ky - 1, input row iy = j - 1 + t  <->  ky = (py ? 4 : 3) - 2t  (ky < 0: tap absent)
            std::memset(wpk + (size_t)(py * 2 + px) * phase_floats, 0, sizeof(float) * phase_floats);
            const int kps = L.SEGP / 32;
            float *dst = wpk + (size_t)(py * 2 + px) * phase_floats;
            for (int t = 0; t < 3; ++t) {
                const int ky = (py ? 4 : 3) - 2 * t;
                if (ky < 0) continue;
                for (int bx = 0; bx < 3; ++bx) {
                    const int kx = (px ? 4 : 3) - 2 * bx;
                    if (kx < 0) continue;
                    for (int ci = 0; ci < cin; ++ci) {
                        const int q = bx * cs_in + ci;
                        float *row = dst + ((size_t)t * kps + q / 32) * npad * 32;
                        for (int n = 0; n < cout; ++n)
                            row[(size_t)n * 32 + swz32(n, q & 31)] =
                                (float)((double)W[(((size_t)ky * 5 + kx) * cout + n) * cin + ci] * scale[n]);
                    }
                }
            }
        }
}

void pack_winograd(const float *W, const double *scale, int cin, int cout, int npad, float *wpk)
{
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const KLayout L = klayout_run(1, 1, cin);
    const size_t phase_floats = (size_t)L.ktiles() * npad * 32;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            pack_generic(L, cin, cin, 1, cout, npad,
                         [&](int, int, int ci, int n) {
                             double u = 0.0;                                   // (G g G^T)[i][j] in double, BatchNorm scale folded in
                             for (int a = 0; a < 3; ++a)
                                 for (int b = 0; b < 3; ++b)
                                     u += G[i][a] * G[j][b] * (double)W[(((size_t)a * 3 + b) * cin + ci) * cout + n];
                             return (float)(u * scale[n]);
                         },
                         wpk + (size_t)(i * 4 + j) * phase_floats);
}

// Winograd F(2x2,2x2) operands of a 4x4 stride-2 transposed convolution (winograd_ops.hip): per axis the 2-tap filter of output parity p is
// g = (W[3 - p], W[1 - p]) (taps on inputs d0, d1 resp. d1, d2 of a tile) and its transform u = (g0, g0 + g1, g1) = G g, G = [1 0; 1 1; 0 1]
void pack_wdec(const float *W, const double *scale, int cin, int cs_in, int cout, float *wpk)
{
    static const double G[3][2] = {{1.0, 0.0}, {1.0, 1.0}, {0.0, 1.0}};
    const KLayout L = klayout_run(1, 1, cs_in);
    const int npad = 4 * cout;
    const size_t pos_floats = (size_t)L.ktiles() * npad * 32;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            pack_generic(L, cs_in, cin, 1, npad, npad,
                         [&](int, int, int ci, int n) {
                             const int ph = n / cout, co = n - ph * cout, py = ph >> 1, px = ph & 1;
                             double u = 0.0;                               // (G g G^T)[i][j] in double, BatchNorm scale folded in
                             for (int a = 0; a < 2; ++a)
                                 for (int b = 0; b < 2; ++b)
                                     u += G[i][a] * G[j][b] * (double)W[(((size_t)(3 - py - 2 * a) * 4 + (3 - px - 2 * b)) * cout + co) * cin + ci];
                             return (float)(u * scale[co]);
                         },
                         wpk + (size_t)(i * 3 + j) * pos_floats);
}

void pack_predict2_table(const float *W, int cin, int cs_in, int npad, float *wpk)
{
    const KLayout L = klayout_run(1, 1, cs_in);
    pack_generic(L, cs_in, cin, 1, 18, npad,
                 [&](int, int, int ci, int n) { return W[((size_t)(n >> 1) * cin + ci) * 2 + (n & 1)]; }, wpk);
}

// the same table for tap_panel_kernel (tap_panel.hip): plain [kpad][32] rows, row c = input channel, column n = 2 * tap + component;
// rows >= cin (concat2's padding channels, the tail of the last 8-float chunk) and columns >= 18 are zero
void pack_predict2_panel(const float *W, int cin, int kpad, float *wp)
{
    std::memset(wp, 0, sizeof(float) * (size_t)kpad * 32);
    for (int c = 0; c < cin; ++c)
        for (int n = 0; n < 18; ++n) wp[(size_t)c * 32 + n] = W[((size_t)(n >> 1) * cin + c) * 2 + (n & 1)];
}

// ---------------------------------------------------------------------------------
// Index tables for DEVICE-side packing (training: the weights change every step, so the gather is replayed on the GPU
// by pack_apply_kernel): tbl[i] = 1 + index into the raw weight tensor of packed element i, 0 = structural zero.
// ---------------------------------------------------------------------------------
template <class F>
static void pack_index_generic(const KLayout &L, int cs_in, int cin, int kw, int cout, int npad, F ifn, int32_t *tbl)
{
    const int kps = L.SEGP / 32;
    std::memset(tbl, 0, sizeof(int32_t) * (size_t)L.ktiles() * npad * 32);
    for (int t = 0; t < L.KH; ++t)
        for (int s = 0; s < L.NSEG; ++s)
            for (int q = 0; q < L.SEG; ++q) {
                const int qa = s * L.SEG_STRIDE + q;
                const int kx = qa / cs_in, ci = qa % cs_in;
                if (kx >= kw || ci >= cin) continue;
                const size_t kt = (size_t)(t * L.NSEG + s) * kps + q / 32;
                int32_t *row = tbl + kt * npad * 32;
                for (int n = 0; n < cout; ++n) {
                    const long long src = ifn(t, kx, ci, n);
                    row[(size_t)n * 32 + swz32(n, q & 31)] = src < 0 ? 0 : (int32_t)(src + 1);
                }
            }
}

void pack_index_conv(int kh, int kw, int cin, int cs_in, int cout, int npad, const KLayout &L, int32_t *tbl)
{
    (void)kh;
    pack_index_generic(L, cs_in, cin, kw, cout, npad,
                       [&](int t, int kx, int ci, int n) { return (long long)(((size_t)t * kw + kx) * cin + ci) * cout + n; }, tbl);
}

// Input gradient of a stride-1 conv W[k][k][Cin][Cout] (pad p) = conv over the output gradient with pad k-1-p, the kernel
// flipped and the channel roles swapped: GEMM columns n = ci, reduction channel = co.
void pack_index_dgrad_s1(int k, int cin, int cout, int cs_g, const KLayout &L, int npad, int32_t *tbl)
{
    pack_index_generic(L, cs_g, cout, k, cin, npad,
                       [&](int t, int kx, int co, int n) {
                           return (long long)(((size_t)(k - 1 - t) * k + (k - 1 - kx)) * cin + n) * cout + co;
                       },
                       tbl);
}

// Input gradient of a stride-2 conv W[k][k][Cin][Cout] (pad p) = stride-2 transposed conv of the output gradient:
// dx[i] = sum over taps t = t0 + 2u (t0 = (i+p)&1) of g[(i + p - t) / 2] W[t].  Four phases (parity of the output pixel) of
// a conv with KT2 = ceil(k/2) row and column taps in run mode; row tap tt <-> u = KT2-1-tt, i.e. increasing input row.
void pack_index_dgrad_s2(int k, int pad, int cin, int cout, int cs_g, const KLayout &L, int npad, int32_t *tbl)
{
    const int kt2 = (k + 1) / 2;
    const size_t phase_elems = (size_t)L.ktiles() * npad * 32;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int t0y = (py + pad) & 1, t0x = (px + pad) & 1;
            pack_index_generic(L, cs_g, cout, kt2, cin, npad,
                               [&](int tt, int bx, int co, int n) -> long long {
                                   const int ky = t0y + 2 * (kt2 - 1 - tt), kx = t0x + 2 * (kt2 - 1 - bx);
                                   if (ky >= k || kx >= k) return -1;
                                   return (long long)(((size_t)ky * k + kx) * cin + n) * cout + co;
                               },
                               tbl + (size_t)(py * 2 + px) * phase_elems);
        }
}

// One output-parity phase of that transposed conv with EXACTLY its taps: nty x ntx taps, row tap tt <-> ky = t0y + 2 (nty-1-tt),
// column tap bx <-> kx = t0x + 2 (ntx-1-bx) (increasing input row / column); L = klayout_run/tap(nty, ntx, ...).
void pack_index_phase(int k, int cin, int cout, int cs_g, const KLayout &L, int npad, int t0y, int nty, int t0x, int ntx, int32_t *tbl)
{
    pack_index_generic(L, cs_g, cout, ntx, cin, npad,
                       [&](int tt, int bx, int co, int n) -> long long {
                           const int ky = t0y + 2 * (nty - 1 - tt), kx = t0x + 2 * (ntx - 1 - bx);
                           if (ky >= k || kx >= k) return -1;
                           return (long long)(((size_t)ky * k + kx) * cin + n) * cout + co;
                       },
                       tbl);
}

}  // namespace vstab
