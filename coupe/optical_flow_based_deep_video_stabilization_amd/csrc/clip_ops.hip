// Kernels of the autoregressive clip driver (the per-frame loop of evaluate_originalSize,
// main:535-630): everything the reference does on the host with cv2/numpy between
// two sess.run calls, kept on the device so a clip never leaves HBM.
#include "vstab_internal.h"

namespace vstab {

// cv2.resize(src, (dw, dh)) with the default INTER_LINEAR on 8-bit images (main:550,556-558).  OpenCV is an
// un-vendored dependency of the reference and is not installed here, so this restates its published 8u
// path (imgproc/resize.cpp: half-pixel centres, 11-bit fixed-point coefficients, HResizeLinear then
// VResizeLinear<uchar,int,short>):
//   scale_x = 1. / ((double)dw / sw)   (cv::resize: inv_scale_x = (double)dsize.width / ssize.width; hal::resize: scale_x = 1./inv_scale_x)
//   fx = (float)((dx+0.5)*scale_x - 0.5)  [the product and difference in double]; sx = floor(fx); fx -= sx;
//   sx<0 -> (0, fx=0); sx>=sw-1 -> (sw-1, fx=0)
//   a = {sat16(round((1-fx)*2048)), sat16(round(fx*2048))};   row value  R = S[sx]*a0 + S[sx+1]*a1
//   dst = ( ((b0*(R0>>4))>>16) + ((b1*(R1>>4))>>16) + 2 ) >> 2
// (exact x2 reductions, which cv2 routes to INTER_AREA, give the same bytes: both coefficients are 1024 and the formula collapses to
// (a+b+c+d+2)>>2).  cv2 itself cannot run here; what pins this restatement: hand-derived known answers of the formula above and an
// independent float bilinear at half-pixel centres (scipy.ndimage) within 1 LSB -- tests/test_oracle_kat.py, tests/test_gpu_clip.py.
// Builds of cv2 that dispatch 8-bit linear resizing to a vendor library (IPP) may differ from the generic path by 1 LSB.
struct Tap { int i0, i1; int a0, a1; };
static inline double cv_scale(int dn, int sn) { return 1.0 / ((double)dn / (double)sn); }     // host side: one division pair per launch
__device__ __forceinline__ Tap cv_tap(int d, double scale, int sn)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= sn - 1) { f = 0.f; s = sn - 1; }
    Tap t;
    t.i0 = s; t.i1 = min(s + 1, sn - 1);
    t.a0 = (int)rintf((1.f - f) * 2048.f);
    t.a1 = (int)rintf(f * 2048.f);
    return t;
}

// src u8 [B,sh,sw,3] -> dst u8 [B,dh,dw,3]
__global__ __launch_bounds__(256) void resize_u8_kernel(const unsigned char *__restrict__ src, int B, int sh, int sw,
                                                        unsigned char *__restrict__ dst, int dh, int dw, double scale_x, double scale_y)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * dh * dw) return;
    const int n = (int)(idx / (dh * dw));
    const int rem = (int)(idx - (long long)n * dh * dw);
    const int dy = rem / dw, dx = rem - dy * dw;
    const Tap X = cv_tap(dx, scale_x, sw), Y = cv_tap(dy, scale_y, sh);
    const unsigned char *b = src + (long long)n * sh * sw * 3;
    const unsigned char *r0 = b + (long long)Y.i0 * sw * 3, *r1 = b + (long long)Y.i1 * sw * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int R0 = r0[X.i0 * 3 + c] * X.a0 + r0[X.i1 * 3 + c] * X.a1;
        const int R1 = r1[X.i0 * 3 + c] * X.a0 + r1[X.i1 * 3 + c] * X.a1;
        const int v = (((Y.a0 * (R0 >> 4)) >> 16) + ((Y.a1 * (R1 >> 4)) >> 16) + 2) >> 2;
        dst[idx * 3 + c] = (unsigned char)min(max(v, 0), 255);
    }
}

// curinput (main:550-558): feats[n,y,x,3j+c] = hist_j[n,y,x,2-c]/255 for the 8 history slots and the current
// small frame (slot 8).  slots: 9 device pointers to u8 [B,h,w,3] frames (BGR as cv2 stores them).
struct Slots9 { const unsigned char *p[9]; };
// The 27 floats of a pixel are 108 bytes, so a thread storing its own pixel's values scatters 4-byte stores 108 bytes apart (round 4:
// 144 us for the 170 MB of eight 384x512 stacks, 1.2 TB/s) and divides by 255 twenty-seven times.  A workgroup's 256 pixels are
// CONTIGUOUS in the output (27 648 bytes): the values go through LDS -- [pixel][27], an odd stride: conflict-free -- and leave as
// 16-byte stores of the whole block; value / 255.0f comes from a 256-entry table of the same IEEE quotients.
typedef float f32x4_clip __attribute__((ext_vector_type(4)));
struct AssembleLds { float lut[256]; float v[256 * 27 + 4]; };
__device__ __forceinline__ void assemble_store_block(AssembleLds &L, float *__restrict__ feats, long long first_px, long long total_px)
{
    __syncthreads();
    const long long n_valid = min((long long)256, total_px - first_px);          // pixels of this block that exist
    const int nfl = (int)n_valid * 27;
    float *o = feats + first_px * 27;                                           // 16-byte aligned: first_px is a multiple of 256
    for (int e = threadIdx.x * 4; e < nfl; e += 1024) {
        if (e + 4 <= nfl) *reinterpret_cast<f32x4_clip *>(o + e) = *reinterpret_cast<const f32x4_clip *>(L.v + e);
        else for (int i = 0; e + i < nfl; ++i) o[e + i] = L.v[e + i];
    }
}
__global__ __launch_bounds__(256) void assemble_input_kernel(Slots9 s, int B, int h, int w, float *__restrict__ feats)
{
    __shared__ __attribute__((aligned(16))) AssembleLds L;
    L.lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
    __syncthreads();
    const long long total = (long long)B * h * w, first = (long long)blockIdx.x * 256, idx = first + threadIdx.x;
    if (idx < total) {
        float *o = L.v + threadIdx.x * 27;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const unsigned char *q = s.p[j] + idx * 3;
            o[3 * j + 0] = L.lut[q[2]];        // cv2.cvtColor(.., COLOR_RGB2BGR) swaps channels 0 and 2
            o[3 * j + 1] = L.lut[q[1]];
            o[3 * j + 2] = L.lut[q[0]];
        }
    }
    assemble_store_block(L, feats, first, total);
}

// resizedInput (main:568): swap(frame)/255 as float
__global__ __launch_bounds__(256) void frame_to_float_kernel(const unsigned char *__restrict__ f, long long npix, float *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npix) return;
    const unsigned char *q = f + idx * 3;
    out[idx * 3 + 0] = (float)q[2] / 255.0f;
    out[idx * 3 + 1] = (float)q[1] / 255.0f;
    out[idx * 3 + 2] = (float)q[0] / 255.0f;
}

// totaloutputFrame[i] = swap(warped*255) (main:625) and its np.uint8 view used for history and the writer
// (main:556,630): truncation toward zero; values outside [0,255] (possible where tf_warp extrapolates) saturate
// here, where numpy's float->uint8 cast is undefined.
__global__ __launch_bounds__(256) void quantise_output_kernel(const float *__restrict__ warped, long long npix,
                                                              unsigned char *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npix) return;
    const float *q = warped + idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = q[2 - c] * 255.0f;
        out[idx * 3 + c] = (unsigned char)fminf(fmaxf(truncf(v), 0.f), 255.f);
    }
}

// The native-resolution evaluator (main:758-866; also evaluate_blurNma / evaluate_medianNma of main_flownetS_pyramid.py) keeps its
// history as  totaloutputFrame[i] = cv2.cvtColor(cv2.resize(warped, (512, 384)) * 255, COLOR_RGB2BGR)  (main:861) and reads it back
// through np.uint8 (main:849, 863): cv2.resize on a float32 image is the same half-pixel-centre bilinear as the 8-bit path with float
// coefficients (HResizeLinear: S[x0]*a0 + S[x1]*a1, then VResizeLinear: b0*R0 + b1*R1), then * 255, channels swapped, truncated.
// Pinned like resize_u8_kernel (known answers + an independent float bilinear), not against cv2 itself; the oracle restates the same
// arithmetic.  src f32 [B,sh,sw,3] -> dst u8 [B,dh,dw,3].
__global__ __launch_bounds__(256) void resize_f32_to_u8_kernel(const float *__restrict__ src, int B, int sh, int sw,
                                                               unsigned char *__restrict__ dst, int dh, int dw, double scale_x, double scale_y)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * dh * dw) return;
    const int n = (int)(idx / (dh * dw));
    const int rem = (int)(idx - (long long)n * dh * dw);
    const int dy = rem / dw, dx = rem - dy * dw;
    auto tap = [](int d, double scale, int sn, int &i0, int &i1, float &a0, float &a1) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= sn - 1) { f = 0.f; s = sn - 1; }
        i0 = s; i1 = min(s + 1, sn - 1); a0 = 1.f - f; a1 = f;
    };
    int x0, x1, y0, y1;
    float ax0, ax1, ay0, ay1;
    tap(dx, scale_x, sw, x0, x1, ax0, ax1);
    tap(dy, scale_y, sh, y0, y1, ay0, ay1);
    const float *b = src + (long long)n * sh * sw * 3;
    const float *r0 = b + (long long)y0 * sw * 3, *r1 = b + (long long)y1 * sw * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float R0 = r0[x0 * 3 + c] * ax0 + r0[x1 * 3 + c] * ax1;
        const float R1 = r1[x0 * 3 + c] * ax0 + r1[x1 * 3 + c] * ax1;
        const float v = (ay0 * R0 + ay1 * R1) * 255.0f;
        dst[idx * 3 + (2 - c)] = (unsigned char)fminf(fmaxf(truncf(v), 0.f), 255.f);
    }
}

hipError_t launch_resize_f32_to_u8(const float *src, int B, int sh, int sw, unsigned char *dst, int dh, int dw, hipStream_t stream)
{
    const long long total = (long long)B * dh * dw;
    resize_f32_to_u8_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(src, B, sh, sw, dst, dh, dw, cv_scale(dw, sw), cv_scale(dh, sh));
    return hipGetLastError();
}

hipError_t launch_resize_u8(const unsigned char *src, int B, int sh, int sw, unsigned char *dst, int dh, int dw, hipStream_t stream)
{
    const long long total = (long long)B * dh * dw;
    resize_u8_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(src, B, sh, sw, dst, dh, dw, cv_scale(dw, sw), cv_scale(dh, sh));
    return hipGetLastError();
}

// assemble_input with the current frame's cv2.resize inside (main:550 + 553-558 in one launch): slot 8 (and every history slot whose
// pointer is null: the first frame of a clip, main:548-549) is resize_u8_kernel's pixel computed here from the full-resolution frame
__global__ __launch_bounds__(256) void assemble_input_resized_kernel(Slots9 s, const unsigned char *__restrict__ frame, int B, int h, int w, int sh,
                                                                     int sw, float *__restrict__ feats, double scale_x, double scale_y)
{
    __shared__ __attribute__((aligned(16))) AssembleLds L;
    L.lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
    __syncthreads();
    const long long total = (long long)B * h * w, first = (long long)blockIdx.x * 256, idx = first + threadIdx.x;
    if (idx < total) {
        const int n = (int)(idx / (h * w));
        const int rem = (int)(idx - (long long)n * h * w);
        const int dy = rem / w, dx = rem - dy * w;
        const Tap X = cv_tap(dx, scale_x, sw), Y = cv_tap(dy, scale_y, sh);
        const unsigned char *b = frame + (long long)n * sh * sw * 3;
        const unsigned char *r0 = b + (long long)Y.i0 * sw * 3, *r1 = b + (long long)Y.i1 * sw * 3;
        unsigned char cur[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int R0 = r0[X.i0 * 3 + c] * X.a0 + r0[X.i1 * 3 + c] * X.a1;
            const int R1 = r1[X.i0 * 3 + c] * X.a0 + r1[X.i1 * 3 + c] * X.a1;
            const int v = (((Y.a0 * (R0 >> 4)) >> 16) + ((Y.a1 * (R1 >> 4)) >> 16) + 2) >> 2;
            cur[c] = (unsigned char)min(max(v, 0), 255);
        }
        float *o = L.v + threadIdx.x * 27;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const unsigned char *q = (j < 8 && s.p[j] != nullptr) ? s.p[j] + idx * 3 : cur;
            o[3 * j + 0] = L.lut[q[2]];
            o[3 * j + 1] = L.lut[q[1]];
            o[3 * j + 2] = L.lut[q[0]];
        }
    }
    assemble_store_block(L, feats, first, total);
}

hipError_t launch_assemble_input_resized(const unsigned char *const *slots8, const unsigned char *frame, int B, int h, int w, int sh, int sw,
                                         float *feats, hipStream_t stream)
{
    Slots9 s;
    for (int j = 0; j < 8; ++j) s.p[j] = slots8[j];
    s.p[8] = nullptr;
    const long long total = (long long)B * h * w;
    assemble_input_resized_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(s, frame, B, h, w, sh, sw, feats, cv_scale(w, sw),
                                                                                                cv_scale(h, sh));
    return hipGetLastError();
}

hipError_t launch_assemble_input(const unsigned char *const *slots9, int B, int h, int w, float *feats, hipStream_t stream)
{
    Slots9 s;
    for (int j = 0; j < 9; ++j) s.p[j] = slots9[j];
    const long long total = (long long)B * h * w;
    assemble_input_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(s, B, h, w, feats);
    return hipGetLastError();
}

hipError_t launch_frame_to_float(const unsigned char *f, long long npix, float *out, hipStream_t stream)
{
    frame_to_float_kernel<<<dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream>>>(f, npix, out);
    return hipGetLastError();
}

hipError_t launch_quantise_output(const float *warped, long long npix, unsigned char *out, hipStream_t stream)
{
    quantise_output_kernel<<<dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream>>>(warped, npix, out);
    return hipGetLastError();
}

}  // namespace vstab

// ---------------------------------------------------------------------------------
// Flow post-filters of the reference's other evaluators.
// ---------------------------------------------------------------------------------
namespace vstab {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// tf.nn.conv2d(of_c, constant(1/(k*k), [k,k,1,1]), SAME) per flow channel (main_flownetS_pyramid.py:634-641):
// a k x k box sum with zero padding times 1/(k*k).  Separable: this kernel sums k taps along one axis.
__global__ __launch_bounds__(256) void box_sum_axis_kernel(const float *__restrict__ in, int B, int h, int w, int k, int axis,
                                                           float scale, float *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * h * w) return;
    const int n = (int)(idx / (h * w));
    const int rem = (int)(idx - (long long)n * h * w);
    const int y = rem / w, x = rem - y * w;
    const int r = (k - 1) / 2;                 // odd k: SAME pads (k-1)/2 on both sides
    const f32x2 *b = reinterpret_cast<const f32x2 *>(in) + (long long)n * h * w;
    f32x2 s = {0.f, 0.f};
    if (axis == 0) {
        const int lo = max(x - r, 0), hi = min(x + (k - 1 - r), w - 1);
        for (int i = lo; i <= hi; ++i) s += b[y * w + i];
    } else {
        const int lo = max(y - r, 0), hi = min(y + (k - 1 - r), h - 1);
        for (int i = lo; i <= hi; ++i) s += b[i * w + x];
    }
    s.x *= scale; s.y *= scale;
    reinterpret_cast<f32x2 *>(out)[idx] = s;
}

hipError_t launch_flow_box_blur(const float *flow, int B, int h, int w, int k, float *tmp, float *out, hipStream_t stream)
{
    if (k < 1 || !(k & 1)) return hipErrorInvalidValue;
    const long long total = (long long)B * h * w;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    box_sum_axis_kernel<<<grid, block, 0, stream>>>(flow, B, h, w, k, 0, 1.0f, tmp);
    box_sum_axis_kernel<<<grid, block, 0, stream>>>(tmp, B, h, w, k, 1, 1.0f / ((float)k * (float)k), out);
    return hipGetLastError();
}

// out = a*x + b*y  (0.9*smoothof + 0.1*prevof, main_flownetS_pyramid.py:643; prevof EMA :695)
__global__ __launch_bounds__(256) void axpby_kernel(const float *__restrict__ x, float a, const float *__restrict__ y, float b,
                                                    float *__restrict__ out, long long n)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) out[idx] = a * x[idx] + b * y[idx];
}

hipError_t launch_axpby(const float *x, float a, const float *y, float b, float *out, long long n, hipStream_t stream)
{
    axpby_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(x, a, y, b, out, n);
    return hipGetLastError();
}

// scipy.signal.medfilt(np.squeeze(of), k) of evaluate_medianNma (main_flownetS_pyramid.py:809): an order filter
// over a kh x kw x kc window of the [h,w,2] field -- the window also spans the CHANNEL axis (the reference passes
// the scalar 5, i.e. 5x5x5) -- with zero padding on every axis; the output is element n/2 of the sorted window.
// One thread per output element; the window sits in LDS (tile + halo) and the median is found by rank counting
// (the value v with #{w < v} <= n/2 < #{w <= v}), which needs no sorting storage and is exact.
constexpr int MED_T = 16;                       // output tile edge
__global__ __launch_bounds__(256) void medfilt_kernel(const float *__restrict__ in, int h, int w, int kh, int kw, int kc,
                                                      float *__restrict__ out)
{
    extern __shared__ float tile[];             // [(MED_T+kh-1)][(MED_T+kw-1)][2]
    const int ry = kh >> 1, rx = kw >> 1, rc = kc >> 1;
    const int th = MED_T + kh - 1, tw = MED_T + kw - 1;
    const int y0 = blockIdx.y * MED_T - ry, x0 = blockIdx.x * MED_T - rx;
    const float *b = in + (long long)blockIdx.z * h * w * 2;
    for (int i = threadIdx.x; i < th * tw; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int y = y0 + ty, x = x0 + tx;
        f32x2 v = {0.f, 0.f};
        if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) v = *reinterpret_cast<const f32x2 *>(b + ((long long)y * w + x) * 2);
        tile[2 * i] = v.x; tile[2 * i + 1] = v.y;
    }
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    const int y = blockIdx.y * MED_T + ly, x = blockIdx.x * MED_T + lx;
    if (y >= h || x >= w) return;
    const int n = kh * kw * kc, target = n >> 1;
    for (int c = 0; c < 2; ++c) {
        // window value number i: (dy, dx, dc); a tap outside the two channels is a padded zero
        auto val = [&](int i) -> float {
            const int dc = i % kc, r = i / kc;
            const int dx = r % kw, dy = r / kw;
            const int cc = c + dc - rc;
            return (unsigned)cc < 2u ? tile[((ly + dy) * tw + lx + dx) * 2 + cc] : 0.f;
        };
        float res = 0.f;
        for (int i = 0; i < n; ++i) {
            const float v = val(i);
            int less = 0, leq = 0;
            for (int j = 0; j < n; ++j) { const float u = val(j); less += u < v; leq += u <= v; }
            if (less <= target && target < leq) { res = v; break; }
        }
        out[(((long long)blockIdx.z * h + y) * w + x) * 2 + c] = res;
    }
}

hipError_t launch_flow_medfilt(const float *flow, int B, int h, int w, int kh, int kw, int kc, float *out, hipStream_t stream)
{
    if (kh < 1 || kw < 1 || kc < 1 || !(kh & 1) || !(kw & 1) || !(kc & 1) || kh > 31 || kw > 31 || kc > 5) return hipErrorInvalidValue;
    const size_t lds = (size_t)(MED_T + kh - 1) * (MED_T + kw - 1) * 2 * sizeof(float);
    dim3 grid((unsigned)((w + MED_T - 1) / MED_T), (unsigned)((h + MED_T - 1) / MED_T), (unsigned)B);
    medfilt_kernel<<<grid, dim3(256), lds, stream>>>(flow, h, w, kh, kw, kc, out);
    return hipGetLastError();
}

// out[b,:,:,c] = mean over (h,w) of flow[b,:,:,c]  (main_flownetS_pyramid_highTV_noBBloss.py:629): one workgroup
// per sample, wavefront-shuffle + LDS reduction, then the same workgroup fills the field.
__global__ __launch_bounds__(256) void mean_fill_kernel(const float *__restrict__ flow, int hw, float *__restrict__ out)
{
    __shared__ float red[8];
    const f32x2 *b = reinterpret_cast<const f32x2 *>(flow) + (long long)blockIdx.x * hw;
    float sx = 0.f, sy = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) { const f32x2 v = b[i]; sx += v.x; sy += v.y; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave * 2] = sx; red[wave * 2 + 1] = sy; }
    __syncthreads();
    const float mx = (red[0] + red[2] + red[4] + red[6]) / (float)hw, my = (red[1] + red[3] + red[5] + red[7]) / (float)hw;
    f32x2 m; m.x = mx; m.y = my;
    f32x2 *o = reinterpret_cast<f32x2 *>(out) + (long long)blockIdx.x * hw;
    for (int i = threadIdx.x; i < hw; i += 256) o[i] = m;
}

hipError_t launch_flow_mean_fill(const float *flow, int B, int h, int w, float *out, hipStream_t stream)
{
    mean_fill_kernel<<<dim3((unsigned)B), dim3(256), 0, stream>>>(flow, h * w, out);
    return hipGetLastError();
}

}  // namespace vstab
