// Homography evaluator (the loop body at main:728-743, "main" =
// main_flownetS_pyramid_noprevloss_dataloader.py): the reference takes the dense output-resolution flow, forms the
// out_h*out_w correspondences  (x, y) -> (x, y) - flow[y, x]  (main:728-731), fits ONE homography to them with
// cv2.findHomography(..., cv2.RANSAC) (main:735) and writes cv2.warpPerspective(frame_unstab, h) (main:736).
//
// OpenCV is an un-vendored dependency that is not installed here and its RANSAC draws from its own RNG, so its
// exact hypotheses cannot be reproduced; what is built is the same estimator (4-point hypotheses, 3 px
// reprojection threshold = cv2's default ransacReprojThreshold, consensus = inlier count, least-squares refit on
// the consensus set) as a deterministic dense-flow RANSAC that never leaves the device:
//   1. K hypotheses per sample from 4 pixels picked by a counter-based hash (restated in the oracle),
//      8x8 solve in fp64 in centre/scale-normalised coordinates;
//   2. ONE pass over the flow scores all K hypotheses (HBM: the flow is read once; the candidates sit in LDS);
//   3. arg-max (lowest index wins ties);
//   4. `refine` rounds of: 23 fp64 moments of the current inlier set -> 8x8 normal equations -> solve.
// All reductions are two-stage with a fixed summation order, so a run is bit-reproducible.
#include "vstab_internal.h"

namespace vstab {

__host__ __device__ inline unsigned homog_hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// Solve the n x n system held in a[n][n+1] (augmented) by Gaussian elimination with partial pivoting.
// Returns false when a pivot is below `eps` in magnitude.
template <int N>
__device__ bool solve_aug(double (&a)[N][N + 1], double eps)
{
    for (int c = 0; c < N; ++c) {
        int piv = c;
        double best = fabs(a[c][c]);
        for (int r = c + 1; r < N; ++r) {
            const double v = fabs(a[r][c]);
            if (v > best) { best = v; piv = r; }
        }
        if (!(best > eps)) return false;
        if (piv != c)
            for (int k = c; k <= N; ++k) { const double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
        const double inv = 1.0 / a[c][c];
        for (int r = c + 1; r < N; ++r) {
            const double f = a[r][c] * inv;
            for (int k = c; k <= N; ++k) a[r][k] -= f * a[c][k];
        }
    }
    for (int c = N - 1; c >= 0; --c) {
        double s = a[c][N];
        for (int k = c + 1; k < N; ++k) s -= a[c][k] * a[k][N];
        a[c][N] = s / a[c][c];
    }
    return true;
}

struct HomogGeom { int H, W; double cx, cy, sc; };

__device__ __forceinline__ void homog_point(const float *__restrict__ flow, const HomogGeom &g, long long pix, double &x, double &y,
                                            double &u, double &v)
{
    const int py = (int)(pix / g.W), px = (int)(pix - (long long)py * g.W);
    const float2 f = *reinterpret_cast<const float2 *>(flow + pix * 2);
    x = ((double)px - g.cx) * g.sc;
    y = ((double)py - g.cy) * g.sc;
    u = ((double)px - (double)f.x - g.cx) * g.sc;                       // gridmeshOF = gridmesh - curoutflow (main:731)
    v = ((double)py - (double)f.y - g.cy) * g.sc;
}

// cand[b][k][9]: hypothesis k of sample b in normalised coordinates, h33 = 1 (all-NaN when the 4 points are degenerate).
__global__ __launch_bounds__(64) void homog_hypotheses_kernel(const float *__restrict__ flow, int B, HomogGeom g, int K,
                                                              unsigned seed, double *__restrict__ cand)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= B * K) return;
    const int b = t / K;
    const long long npix = (long long)g.H * g.W;
    const float *fb = flow + (long long)b * npix * 2;
    double a[8][9];
    for (int j = 0; j < 4; ++j) {
        const unsigned h = homog_hash(seed + 0x9E3779B9U * (unsigned)(t * 4 + j + 1));
        const long long pix = (long long)(((unsigned long long)h * (unsigned long long)npix) >> 32);
        double x, y, u, v;
        homog_point(fb, g, pix, x, y, u, v);
        double *r0 = a[2 * j], *r1 = a[2 * j + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
    }
    const bool ok = solve_aug<8>(a, 1e-10);
    double *o = cand + (long long)t * 9;
    for (int i = 0; i < 8; ++i) o[i] = ok ? a[i][8] : __longlong_as_double(0x7ff8000000000000LL);
    o[8] = 1.0;
}

__device__ __forceinline__ bool homog_inlier(const double *h, double x, double y, double u, double v, double thr2)
{
    const double w = h[6] * x + h[7] * y + h[8];
    const double iw = 1.0 / w;
    const double du = (h[0] * x + h[1] * y + h[2]) * iw - u;
    const double dv = (h[3] * x + h[4] * y + h[5]) * iw - v;
    return du * du + dv * dv <= thr2;                                    // false for NaN hypotheses
}

// counts[b][k] += #{scored pixels whose reprojection error under hypothesis k is <= thr}.  grid (nblk, B), 256 threads;
// every thread keeps SCORE_P pixels in registers and walks the K hypotheses (LDS broadcast reads).
constexpr int SCORE_P = 4;
__global__ __launch_bounds__(256) void homog_score_kernel(const float *__restrict__ flow, HomogGeom g, int K, int stride,
                                                          double thr2, const double *__restrict__ cand, int *__restrict__ counts)
{
    extern __shared__ double s_h[];                                     // K*9 doubles, then K ints
    int *s_cnt = reinterpret_cast<int *>(s_h + (size_t)K * 9);
    const int b = blockIdx.y;
    const long long npix = (long long)g.H * g.W;
    const long long nscore = (npix + stride - 1) / stride;
    const float *fb = flow + (long long)b * npix * 2;
    for (int i = threadIdx.x; i < K * 9; i += 256) s_h[i] = cand[(long long)b * K * 9 + i];
    for (int i = threadIdx.x; i < K; i += 256) s_cnt[i] = 0;
    double x[SCORE_P], y[SCORE_P], u[SCORE_P], v[SCORE_P];
    bool live[SCORE_P];
#pragma unroll
    for (int p = 0; p < SCORE_P; ++p) {
        const long long s = ((long long)blockIdx.x * SCORE_P + p) * 256 + threadIdx.x;
        live[p] = s < nscore;
        x[p] = y[p] = u[p] = v[p] = 0.0;
        if (live[p]) homog_point(fb, g, s * stride, x[p], y[p], u[p], v[p]);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < K; ++k) {
        const double *h = s_h + k * 9;
        int c = 0;
#pragma unroll
        for (int p = 0; p < SCORE_P; ++p)
            c += __popcll(__ballot(live[p] && homog_inlier(h, x[p], y[p], u[p], v[p], thr2)));
        if (lane == 0 && c) atomicAdd(&s_cnt[k], c);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256)
        if (s_cnt[i]) atomicAdd(&counts[b * K + i], s_cnt[i]);
}

// cur[b][9] <- the hypothesis with the largest consensus (lowest index on ties); best_cnt[b] <- its count.
__global__ __launch_bounds__(64) void homog_select_kernel(const int *__restrict__ counts, const double *__restrict__ cand, int K,
                                                          double *__restrict__ cur, int *__restrict__ best_cnt)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    int bc = -1, bk = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
        const int c = counts[b * K + k];
        if (c > bc) { bc = c; bk = k; }
    }
    for (int off = 32; off; off >>= 1) {
        const int oc = __shfl_xor(bc, off), ok = __shfl_xor(bk, off);
        if (oc > bc || (oc == bc && ok < bk)) { bc = oc; bk = ok; }
    }
    if (lane < 9) cur[b * 9 + lane] = cand[((long long)b * K + bk) * 9 + lane];
    if (lane == 0) best_cnt[b] = bc;
}

// 23 moments (+ count) of the pixels that are inliers of cur[b]; part[b][blk][24].
//   m: S = {x2, xy, y2, x, y, 1}; U = u*S (6); V = v*S (6); Q = (u2+v2)*{x2, xy, y2, x, y} (5)
constexpr int NMOM = 24;
__global__ __launch_bounds__(256) void homog_moments_kernel(const float *__restrict__ flow, HomogGeom g, double thr2,
                                                            const double *__restrict__ cur, double *__restrict__ part)
{
    __shared__ double s_red[4][NMOM];
    const int b = blockIdx.y;
    const long long npix = (long long)g.H * g.W;
    const float *fb = flow + (long long)b * npix * 2;
    double h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = cur[b * 9 + i];
    double m[NMOM];
#pragma unroll
    for (int i = 0; i < NMOM; ++i) m[i] = 0.0;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long long)gridDim.x * 256) {
        double x, y, u, v;
        homog_point(fb, g, pix, x, y, u, v);
        if (!homog_inlier(h, x, y, u, v, thr2)) continue;
        const double s[6] = {x * x, x * y, y * y, x, y, 1.0};
        const double q = u * u + v * v;
#pragma unroll
        for (int i = 0; i < 6; ++i) { m[i] += s[i]; m[6 + i] += u * s[i]; m[12 + i] += v * s[i]; }
#pragma unroll
        for (int i = 0; i < 5; ++i) m[18 + i] += q * s[i];
        m[23] += 1.0;
    }
#pragma unroll
    for (int i = 0; i < NMOM; ++i) {
        double t = m[i];
        for (int off = 32; off; off >>= 1) t += __shfl_xor(t, off);
        m[i] = t;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < NMOM; ++i) s_red[wave][i] = m[i];
    __syncthreads();
    if (threadIdx.x < NMOM)
        part[((long long)b * gridDim.x + blockIdx.x) * NMOM + threadIdx.x] =
            ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
}

// Sum the partial moments in block order, solve the normal equations of
//   [x y 1 0 0 0 -ux -uy] h = u,   [0 0 0 x y 1 -vx -vy] h = v
// and replace cur[b] (kept when fewer than 4 inliers or a singular system).  With `final` the de-normalised
// matrix (h33 = 1) goes to Hout[b][9] and the inlier count of this fit to inliers[b].
__global__ __launch_bounds__(64) void homog_solve_kernel(const double *__restrict__ part, int nblk, HomogGeom g,
                                                         double *__restrict__ cur, int final, double *__restrict__ Hout,
                                                         int *__restrict__ inliers)
{
    __shared__ double s_m[NMOM];
    const int b = blockIdx.x;
    if (threadIdx.x < NMOM) {
        double t = 0.0;
        for (int i = 0; i < nblk; ++i) t += part[((long long)b * nblk + i) * NMOM + threadIdx.x];
        s_m[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double *S = s_m, *U = s_m + 6, *V = s_m + 12, *Q = s_m + 18;
    // index into S-like blocks: 0 x2, 1 xy, 2 y2, 3 x, 4 y, 5 one
    double a[8][9];
    const double P[3][3] = {{S[0], S[1], S[3]}, {S[1], S[2], S[4]}, {S[3], S[4], S[5]}};          // sum p p^T, p = (x, y, 1)
    const double PU[3][2] = {{U[0], U[1]}, {U[1], U[2]}, {U[3], U[4]}};                          // sum u p (x, y)
    const double PV[3][2] = {{V[0], V[1]}, {V[1], V[2]}, {V[3], V[4]}};
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 9; ++j) a[i][j] = 0.0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { a[i][j] = P[i][j]; a[3 + i][3 + j] = P[i][j]; }
        for (int j = 0; j < 2; ++j) {
            a[i][6 + j] = -PU[i][j]; a[6 + j][i] = -PU[i][j];
            a[3 + i][6 + j] = -PV[i][j]; a[6 + j][3 + i] = -PV[i][j];
        }
    }
    a[6][6] = Q[0]; a[6][7] = Q[1]; a[7][6] = Q[1]; a[7][7] = Q[2];
    a[0][8] = U[3]; a[1][8] = U[4]; a[2][8] = U[5];
    a[3][8] = V[3]; a[4][8] = V[4]; a[5][8] = V[5];
    a[6][8] = -Q[3]; a[7][8] = -Q[4];
    const bool ok = s_m[23] >= 4.0 && solve_aug<8>(a, 1e-14 * (fabs(S[5]) + 1.0));
    double h[9];
    for (int i = 0; i < 8; ++i) h[i] = ok ? a[i][8] : cur[b * 9 + i];
    h[8] = 1.0;
    for (int i = 0; i < 9; ++i) cur[b * 9 + i] = h[i];
    if (!final) return;
    // H = T^-1 Hn T,  T = [[sc,0,-cx*sc],[0,sc,-cy*sc],[0,0,1]]
    const double sc = g.sc, cx = g.cx, cy = g.cy, is = 1.0 / sc;
    double A[9];                                                         // Hn T
    for (int r = 0; r < 3; ++r) {
        A[r * 3 + 0] = h[r * 3 + 0] * sc;
        A[r * 3 + 1] = h[r * 3 + 1] * sc;
        A[r * 3 + 2] = h[r * 3 + 2] - (h[r * 3 + 0] * cx + h[r * 3 + 1] * cy) * sc;
    }
    double M[9];
    for (int c = 0; c < 3; ++c) {
        M[0 + c] = A[0 + c] * is + cx * A[6 + c];
        M[3 + c] = A[3 + c] * is + cy * A[6 + c];
        M[6 + c] = A[6 + c];
    }
    const double n = 1.0 / M[8];
    for (int i = 0; i < 9; ++i) Hout[b * 9 + i] = M[i] * n;
    inliers[b] = (int)s_m[23];
}

hipError_t launch_homography_fit(const float *flow, int B, int H, int W, int K, unsigned seed, double thresh, int refine,
                                 int stride, double *Hout, int *inliers, void *ws, hipStream_t stream)
{
    HomogGeom g;
    g.H = H; g.W = W;
    g.cx = 0.5 * (W - 1); g.cy = 0.5 * (H - 1);
    const double ext = g.cx > g.cy ? g.cx : g.cy;
    g.sc = 1.0 / (ext > 1.0 ? ext : 1.0);
    const double thr2 = thresh * g.sc * thresh * g.sc;
    const int nblk = homography_moment_blocks(H, W);
    char *w = static_cast<char *>(ws);
    double *cand = reinterpret_cast<double *>(w);                  w += (size_t)B * K * 9 * sizeof(double);
    double *cur = reinterpret_cast<double *>(w);                   w += (size_t)B * 9 * sizeof(double);
    double *part = reinterpret_cast<double *>(w);                  w += (size_t)B * nblk * NMOM * sizeof(double);
    int *counts = reinterpret_cast<int *>(w);                      w += (size_t)B * K * sizeof(int);
    int *best = reinterpret_cast<int *>(w);
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)B * K * sizeof(int), stream);
    if (e != hipSuccess) return e;
    homog_hypotheses_kernel<<<(B * K + 63) / 64, 64, 0, stream>>>(flow, B, g, K, seed, cand);
    const long long npix = (long long)H * W, nscore = (npix + stride - 1) / stride;
    const unsigned sblk = (unsigned)((nscore + 256 * SCORE_P - 1) / (256 * SCORE_P));
    homog_score_kernel<<<dim3(sblk, B), 256, (size_t)K * 9 * sizeof(double) + (size_t)K * sizeof(int), stream>>>(
        flow, g, K, stride, thr2, cand, counts);
    homog_select_kernel<<<B, 64, 0, stream>>>(counts, cand, K, cur, best);
    for (int it = 0; it < refine; ++it) {
        homog_moments_kernel<<<dim3(nblk, B), 256, 0, stream>>>(flow, g, thr2, cur, part);
        homog_solve_kernel<<<B, 64, 0, stream>>>(part, nblk, g, cur, it == refine - 1, Hout, inliers);
    }
    return hipGetLastError();
}

int homography_moment_blocks(int H, int W)
{
    const long long npix = (long long)H * W;
    long long n = (npix + 256 * 8 - 1) / (256 * 8);
    return (int)(n < 1 ? 1 : (n > 512 ? 512 : n));
}

size_t homography_workspace_bytes(int B, int H, int W, int K)
{
    return (size_t)B * K * 9 * sizeof(double) + (size_t)B * 9 * sizeof(double) +
           (size_t)B * homography_moment_blocks(H, W) * NMOM * sizeof(double) + (size_t)B * K * sizeof(int) + (size_t)B * sizeof(int) + 64;
}

// cv2.warpPerspective(src, M, (ow, oh)) with the default flags (INTER_LINEAR, BORDER_CONSTANT 0, M maps src -> dst
// so the sampler uses M^-1; main:736) on 8-bit frames.  Restated from OpenCV's published behaviour
// (imgproc/imgwarp.cpp: source coordinates rounded to 1/32 px -- INTER_BITS = 5 -- and a 15-bit fixed-point bilinear
// blend); cv2 is not installed here: pinned by hand-derived known answers and an exact float bilinear at the 1/32-pixel coordinates
// (tests/test_oracle_kat.py, tests/test_gpu_postfilters.py), not against the library itself (the oracle restates the same arithmetic):
//   X = rint(32 * X0 / W0), sx = X >> 5, a = X & 31 (same for Y);   w = {(32-a)(32-b), a(32-b), (32-a)b, ab} * 32
//   dst = (sum w_i * src_i + 2^14) >> 15,  taps outside the image read 0.
__global__ __launch_bounds__(256) void warp_perspective_u8_kernel(const unsigned char *__restrict__ src, int B, int sh, int sw,
                                                                  const double *__restrict__ Hm, unsigned char *__restrict__ dst,
                                                                  int oh, int ow)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * oh * ow) return;
    const int n = (int)(idx / ((long long)oh * ow));
    const int rem = (int)(idx - (long long)n * oh * ow);
    const int dy = rem / ow, dx = rem - dy * ow;
    const double *m = Hm + n * 9;
    // inverse by the adjugate
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double A0 = e * i - f * h, A1 = c * h - b * i, A2 = b * f - c * e;
    const double A3 = f * g - d * i, A4 = a * i - c * g, A5 = c * d - a * f;
    const double A6 = d * h - e * g, A7 = b * g - a * h, A8 = a * e - b * d;
    const double det = a * A0 + b * A3 + c * A6;
    const double id = det != 0.0 ? 1.0 / det : 0.0;
    const double X0 = (A0 * dx + A1 * dy + A2) * id, Y0 = (A3 * dx + A4 * dy + A5) * id;
    double Wd = (A6 * dx + A7 * dy + A8) * id;
    Wd = Wd != 0.0 ? 32.0 / Wd : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, X0 * Wd));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, Y0 * Wd));
    const int X = (int)rint(fX), Y = (int)rint(fY);
    int sx = X >> 5, sy = Y >> 5;
    sx = min(max(sx, -32768), 32767); sy = min(max(sy, -32768), 32767);                          // the short map of cv2
    const int ax = X & 31, ay = Y & 31;
    const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
    const unsigned char *base = src + (long long)n * sh * sw * 3;
    const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw;
    const bool y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int p00 = (x0 && y0) ? base[((long long)sy * sw + sx) * 3 + ch] : 0;
        const int p01 = (x1 && y0) ? base[((long long)sy * sw + sx + 1) * 3 + ch] : 0;
        const int p10 = (x0 && y1) ? base[((long long)(sy + 1) * sw + sx) * 3 + ch] : 0;
        const int p11 = (x1 && y1) ? base[((long long)(sy + 1) * sw + sx + 1) * 3 + ch] : 0;
        const int v = (w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11 + (1 << 14)) >> 15;
        dst[idx * 3 + ch] = (unsigned char)min(max(v, 0), 255);
    }
}

hipError_t launch_warp_perspective_u8(const unsigned char *src, int B, int sh, int sw, const double *Hm, unsigned char *dst,
                                      int oh, int ow, hipStream_t stream)
{
    const long long n = (long long)B * oh * ow;
    warp_perspective_u8_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(src, B, sh, sw, Hm, dst, oh, ow);
    return hipGetLastError();
}

}  // namespace vstab
