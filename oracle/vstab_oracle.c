/*
 * Plain-C restatement of the FlowNetS-pyramid + tf_warp hot path, double precision.
 *
 * TEST INFRASTRUCTURE ONLY -- second, independent restatement used to cross-check
 * oracle/vstab_oracle.py (which leans on torch's conv kernels).  Direct loops, no
 * library calls, so it is only run at small sizes.  PARITY UNPINNED at the TF-1.10 /
 * TensorLayer boundary (no reference tests/goldens exist; see vstab_oracle.py header).
 *
 * Reference lines followed ("main" = main_flownetS_pyramid_noprevloss_dataloader.py):
 *   vo_pad_conv            model.py:807-808 (PadLayer + Conv2d VALID + bias)
 *   vo_bn_lrelu            model.py:809     (BatchNormLayer, no gamma, lrelu 0.1)
 *   vo_deconv4x4s2         model.py:850     (DeConv2dLayer 4x4 s2 SAME + bias)
 *   vo_resize_bilinear     model.py:857,886 / main:497,806 (legacy TF bilinear)
 *   vo_nearest_index       model.py:883     (nearest, align_corners=True)
 *   vo_predict2_fullres    model.py:882-885
 *   vo_flownetS_pyramid    model.py:786-893
 *   vo_flow_to_output_res  main:497-498
 *   vo_tf_warp             main:70-130 (+ get_pixel_value main:44-68)
 * Layouts: activations NHWC, conv W [kh][kw][Cin][Cout], deconv W [kh][kw][Cout][Cin].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BN_EPS 1e-5

void vo_pad_conv(const double *x, int B, int H, int W, int Cin, const double *Wt, int k, int Cout,
                 const double *b, int pad, int stride, double *y)
{
    int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    for (int n = 0; n < B; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                double *o = y + (((size_t)n * Ho + oy) * Wo + ox) * Cout;
                for (int co = 0; co < Cout; ++co) o[co] = b ? b[co] : 0.0;
                for (int ky = 0; ky < k; ++ky) {
                    int iy = oy * stride + ky - pad;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = 0; kx < k; ++kx) {
                        int ix = ox * stride + kx - pad;
                        if (ix < 0 || ix >= W) continue;
                        const double *xi = x + (((size_t)n * H + iy) * W + ix) * Cin;
                        const double *w = Wt + ((size_t)(ky * k + kx) * Cin) * Cout;
                        for (int ci = 0; ci < Cin; ++ci) {
                            double v = xi[ci];
                            const double *wr = w + (size_t)ci * Cout;
                            for (int co = 0; co < Cout; ++co) o[co] += v * wr[co];
                        }
                    }
                }
            }
}

void vo_bn_lrelu(double *x, size_t npix, int C, const double *beta, const double *mean,
                 const double *var)
{
    for (size_t p = 0; p < npix; ++p)
        for (int c = 0; c < C; ++c) {
            double v = (x[p * C + c] - mean[c]) / sqrt(var[c] + BN_EPS) + beta[c];
            x[p * C + c] = v > 0.1 * v ? v : 0.1 * v;
        }
}

/* scatter form of conv2d_transpose: every input pixel adds its 4x4 footprint at
 * oy = 2*iy + ky - 1; outputs beyond output_shape are dropped. */
void vo_deconv4x4s2(const double *x, int B, int h, int w, int Cin, const double *Wt, int Cout,
                    const double *b, int oh, int ow, double *y)
{
    for (size_t i = 0; i < (size_t)B * oh * ow; ++i)
        for (int co = 0; co < Cout; ++co) y[i * Cout + co] = b ? b[co] : 0.0;
    for (int n = 0; n < B; ++n)
        for (int iy = 0; iy < h; ++iy)
            for (int ix = 0; ix < w; ++ix) {
                const double *xi = x + (((size_t)n * h + iy) * w + ix) * Cin;
                for (int ky = 0; ky < 4; ++ky) {
                    int oy = 2 * iy + ky - 1;
                    if (oy < 0 || oy >= oh) continue;
                    for (int kx = 0; kx < 4; ++kx) {
                        int ox = 2 * ix + kx - 1;
                        if (ox < 0 || ox >= ow) continue;
                        double *o = y + (((size_t)n * oh + oy) * ow + ox) * Cout;
                        const double *wk = Wt + (size_t)(ky * 4 + kx) * Cout * Cin;
                        for (int co = 0; co < Cout; ++co) {
                            const double *wr = wk + (size_t)co * Cin;
                            double s = 0.0;
                            for (int ci = 0; ci < Cin; ++ci) s += xi[ci] * wr[ci];
                            o[co] += s;
                        }
                    }
                }
            }
}

void vo_resize_bilinear(const double *x, int B, int h, int w, int C, int oh, int ow, double *y)
{
    if (h == oh && w == ow) { memcpy(y, x, sizeof(double) * (size_t)B * h * w * C); return; }
    float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
    for (int n = 0; n < B; ++n)
        for (int oy = 0; oy < oh; ++oy) {
            float fy = (float)oy * sy;
            int y0 = (int)floorf(fy), y1 = y0 + 1 < h ? y0 + 1 : h - 1;
            double ty = (double)(fy - (float)y0);
            for (int ox = 0; ox < ow; ++ox) {
                float fx = (float)ox * sx;
                int x0 = (int)floorf(fx), x1 = x0 + 1 < w ? x0 + 1 : w - 1;
                double tx = (double)(fx - (float)x0);
                const double *tl = x + (((size_t)n * h + y0) * w + x0) * C;
                const double *tr = x + (((size_t)n * h + y0) * w + x1) * C;
                const double *bl = x + (((size_t)n * h + y1) * w + x0) * C;
                const double *br = x + (((size_t)n * h + y1) * w + x1) * C;
                double *o = y + (((size_t)n * oh + oy) * ow + ox) * C;
                for (int c = 0; c < C; ++c) {
                    double top = tl[c] + (tr[c] - tl[c]) * tx;
                    double bot = bl[c] + (br[c] - bl[c]) * tx;
                    o[c] = top + (bot - top) * ty;
                }
            }
        }
}

int vo_nearest_index(int i, int n_in, int n_out)
{
    float scale = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f;
    int s = (int)roundf((float)i * scale);
    return s < n_in - 1 ? s : n_in - 1;
}

/* concat2 [B,h2,w2,C] -> pf2_raw [B,H-2,W-2,2]; the upsampled tensor is never built */
void vo_predict2_fullres(const double *c2, int B, int h2, int w2, int C, const double *Wt,
                         const double *b, int H, int W, double *y)
{
    int ph = h2 + 2, pw = w2 + 2, oh = H - 2, ow = W - 2;
    for (int n = 0; n < B; ++n)
        for (int oy = 0; oy < oh; ++oy)
            for (int ox = 0; ox < ow; ++ox) {
                double a0 = b[0], a1 = b[1];
                for (int dy = 0; dy < 3; ++dy) {
                    int py = vo_nearest_index(oy + dy, ph, H) - 1;   /* index into unpadded */
                    if (py < 0 || py >= h2) continue;
                    for (int dx = 0; dx < 3; ++dx) {
                        int px = vo_nearest_index(ox + dx, pw, W) - 1;
                        if (px < 0 || px >= w2) continue;
                        const double *xi = c2 + (((size_t)n * h2 + py) * w2 + px) * C;
                        const double *wk = Wt + (size_t)(dy * 3 + dx) * C * 2;
                        for (int c = 0; c < C; ++c) { a0 += xi[c] * wk[2 * c]; a1 += xi[c] * wk[2 * c + 1]; }
                    }
                }
                double *o = y + (((size_t)n * oh + oy) * ow + ox) * 2;
                o[0] = a0; o[1] = a1;
            }
}

void vo_tf_warp(const double *img, const float *flow, int B, int H, int W, int C, double *out)
{
    for (int n = 0; n < B; ++n)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx) {
                size_t p = ((size_t)n * H + yy) * W + xx;
                float x = (float)xx + flow[2 * p], y = (float)yy + flow[2 * p + 1];
                int x0 = (int)x, y0 = (int)y;            /* C cast truncates like tf.cast */
                int x1 = x0 + 1, y1 = y0 + 1;
                x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0);
                x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
                y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
                y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
                double wa = ((double)x1 - x) * ((double)y1 - y), wb = ((double)x1 - x) * (y - (double)y0);
                double wc = (x - (double)x0) * ((double)y1 - y), wd = (x - (double)x0) * (y - (double)y0);
                const double *Ia = img + (((size_t)n * H + y0) * W + x0) * C;
                const double *Ib = img + (((size_t)n * H + y1) * W + x0) * C;
                const double *Ic = img + (((size_t)n * H + y0) * W + x1) * C;
                const double *Id = img + (((size_t)n * H + y1) * W + x1) * C;
                for (int c = 0; c < C; ++c)
                    out[p * C + c] = wa * Ia[c] + wb * Ib[c] + wc * Ic[c] + wd * Id[c];
            }
}

void vo_flow_to_output_res(const double *pf2, int B, int h, int w, int net_h, int net_w, int oh,
                           int ow, double *out)
{
    size_t n = (size_t)B * h * w * 2;
    double *tmp = (double *)malloc(sizeof(double) * n);
    /* the graph's op sequence: (pf2 * net_h) / h, resize, (x * ow) / net_w, (y * oh) / net_h */
    for (size_t i = 0; i < n; ++i) tmp[i] = (pf2[i] * (double)net_h) / (double)h;
    vo_resize_bilinear(tmp, B, h, w, 2, oh, ow, out);
    free(tmp);
    for (size_t i = 0; i < (size_t)B * oh * ow; ++i) {
        out[2 * i] = (out[2 * i] * (double)ow) / (double)net_w;
        out[2 * i + 1] = (out[2 * i + 1] * (double)oh) / (double)net_h;
    }
}

/* ---- whole network.  wts: 88 pointers in this order:
 *  10 x {W,b,beta,mean,var} for stages 1,2,3,3_1,4,4_1,5,5_1,6,6_1
 *   5 x {W,b} for predict6,5,4,3,2
 *   4 x {W,b,beta,mean,var} for deconv5,4,3,2 (+ their _bn)
 *   4 x {W,b} for upsample6_5, 5_4, 4_3, 3_2
 * outputs pf6..pf2 preallocated by the caller.                                   */
static const int ENC[10][4] = { {7,2,3,64},{5,2,2,128},{5,2,2,256},{3,1,1,256},{3,2,1,512},
    {3,1,1,512},{3,2,1,512},{3,1,1,512},{3,2,1,1024},{3,1,1,1024} };

static double *dalloc(size_t n) { return (double *)malloc(sizeof(double) * (n ? n : 1)); }

static void concat3(const double *a, int ca, const double *b, int cb, const double *c, int cc,
                    size_t npix, double *out)
{
    int ct = ca + cb + cc;
    for (size_t p = 0; p < npix; ++p) {
        memcpy(out + p * ct, a + p * ca, sizeof(double) * ca);
        memcpy(out + p * ct + ca, b + p * cb, sizeof(double) * cb);
        memcpy(out + p * ct + ca + cb, c + p * cc, sizeof(double) * cc);
    }
}

void vo_flownetS_pyramid(const double *feats, int B, int H, int W, int Cin, const double **wts,
                         double *pf6, double *pf5, double *pf4, double *pf3, double *pf2)
{
    double *act[10]; int hh[10], ww[10];
    const double *cur = feats; int h = H, w = W, c = Cin;
    for (int i = 0; i < 10; ++i) {
        int k = ENC[i][0], s = ENC[i][1], p = ENC[i][2], co = ENC[i][3];
        int ho = (h + 2 * p - k) / s + 1, wo = (w + 2 * p - k) / s + 1;
        act[i] = dalloc((size_t)B * ho * wo * co);
        vo_pad_conv(cur, B, h, w, c, wts[5 * i], k, co, wts[5 * i + 1], p, s, act[i]);
        vo_bn_lrelu(act[i], (size_t)B * ho * wo, co, wts[5 * i + 2], wts[5 * i + 3], wts[5 * i + 4]);
        cur = act[i]; h = ho; w = wo; c = co; hh[i] = ho; ww[i] = wo;
    }
    const double **pw = wts + 50, **dw = wts + 60, **uw = wts + 80;
    /* level 6 */
    vo_pad_conv(act[9], B, hh[9], ww[9], 1024, pw[0], 3, 2, pw[1], 1, 1, pf6);
    const int skip_idx[4] = {7, 5, 3, 1}, skip_c[4] = {512, 512, 256, 128}, dec_c[4] = {512, 256, 128, 64};
    double *pfs[5] = {pf6, pf5, pf4, pf3, pf2};
    const double *prev = act[9]; int pc = 1024, ph = hh[9], pwid = ww[9];
    double *concat_prev = NULL;
    for (int l = 0; l < 4; ++l) {
        int sh = hh[skip_idx[l]], sw = ww[skip_idx[l]], dc = dec_c[l], ct = skip_c[l] + dc + 2;
        size_t npix = (size_t)B * sh * sw;
        double *d = dalloc(npix * dc), *uf = dalloc(npix * 2), *cat = dalloc(npix * ct);
        vo_deconv4x4s2(prev, B, ph, pwid, pc, dw[5 * l], dc, dw[5 * l + 1], sh, sw, d);
        vo_bn_lrelu(d, npix, dc, dw[5 * l + 2], dw[5 * l + 3], dw[5 * l + 4]);
        vo_deconv4x4s2(pfs[l], B, ph, pwid, 2, uw[2 * l], 2, uw[2 * l + 1], sh, sw, uf);
        concat3(act[skip_idx[l]], skip_c[l], d, dc, uf, 2, npix, cat);
        free(d); free(uf);
        if (l < 3) {
            double *up = dalloc(npix * 2);
            vo_pad_conv(cat, B, sh, sw, ct, pw[2 * (l + 1)], 3, 2, pw[2 * (l + 1) + 1], 1, 1, pfs[l + 1]);
            vo_resize_bilinear(pfs[l], B, ph, pwid, 2, sh, sw, up);
            for (size_t i = 0; i < npix * 2; ++i) pfs[l + 1][i] = (pfs[l + 1][i] + up[i]) + up[i];
            free(up);
        } else {
            size_t n2 = (size_t)B * (H - 2) * (W - 2) * 2;
            double *up = dalloc(n2);
            vo_predict2_fullres(cat, B, sh, sw, ct, pw[8], pw[9], H, W, pf2);
            vo_resize_bilinear(pf3, B, ph, pwid, 2, H - 2, W - 2, up);
            for (size_t i = 0; i < n2; ++i) { double v = pf2[i]; for (int r = 0; r < 8; ++r) v += up[i]; pf2[i] = v; }
            free(up);
        }
        free(concat_prev); concat_prev = cat;
        prev = cat; pc = ct; ph = sh; pwid = sw;
    }
    free(concat_prev);
    for (int i = 0; i < 10; ++i) free(act[i]);
}
