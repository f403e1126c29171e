// Kernel-tuning harness (not part of the product): times ONE conv-like launch of the forward
// schedule in isolation with random data, for tile / split-K experiments.
//   conv_bench <layer 0..14> <B> <H> <W> [tile -1|0|1|2|3] [ksplit -1|n] [iters]
// Build: scripts/build_tools.sh ; run on the GPU box.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../include/vstab.h"
#include "../coupe/optical_flow_based_deep_video_stabilization_amd/csrc/vstab_internal.h"
using namespace vstab;
void fill_wino_gemm(vstab::ConvParams &p, int B, int H, int W, int cin, int cout);     // api.cpp: the 16-position GEMM of a Winograd-form stage
#ifdef VSTAB_STAMP
namespace vstab { hipError_t conv_read_stamps(unsigned long long *host, size_t n); }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 5) { printf("usage: conv_bench layer B H W [tile] [ksplit] [iters]\n"); return 2; }
    // layer 100 + i = the Winograd-domain GEMM of encoder stage i (3, 5, 7, 9 = conv3_1, conv4_1, conv5_1, conv6_1)
    const bool wino = atoi(argv[1]) >= 100;
    const int layer = atoi(argv[1]) % 100, B = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]);
    const int tile_o = argc > 5 ? atoi(argv[5]) : -1, ks_o = argc > 6 ? atoi(argv[6]) : -1, iters = argc > 7 ? atoi(argv[7]) : 20;
    int32_t v[64];
    if (vstab_host_layer_plan(B, H, W, 27, layer, v, 64) < 0) { printf("plan failed\n"); return 1; }
    ConvParams p{};
    p.B = v[0]; p.Hi = v[1]; p.Wi = v[2]; p.Cs_in = v[3]; p.KH = v[4]; p.NSEG = v[5]; p.SEG = v[6]; p.SEGP = v[7];
    p.SEG_STRIDE = v[8]; p.s_in = v[9]; p.s_out = v[10]; p.Ho = v[11]; p.Wo = v[12]; p.Cs_out = v[13]; p.c_off = v[14];
    p.N = v[15]; p.Npad = v[16]; p.act = v[17]; p.nphase = v[18]; p.ksplit = v[19]; p.Mmax = v[20];
    int tile = v[21]; const bool vec4 = v[22];
    if (wino) {
        const int cin = v[3], cout = v[15], hi = v[1], wi = v[2];
        fill_wino_gemm(p, B, hi, wi, cin, cout);
        tile = TILE_128x64;
    }
    const int KT = p.KH * p.NSEG * p.SEGP / 32;
    if (tile_o >= 0) { tile = tile_o; const int BN = (tile == 0 || tile == 3) ? 128 : ((tile == 1 || tile == 4) ? 64 : 32); p.Npad = round_up(p.N, BN); }
    if (ks_o >= 1) p.ksplit = ks_o;
    double macs = 0;
    for (int k = 0; k < p.nphase; ++k) {
        ConvPhase &ph = p.ph[k];
        if (wino) { ph.w_off = (long long)k * KT * p.Npad * 32; macs += (double)ph.M * KT * 32 * p.N; continue; }
        ph.Hg = v[26 + 7 * k]; ph.Wg = v[27 + 7 * k]; ph.M = v[28 + 7 * k]; ph.off_y = v[29 + 7 * k]; ph.off_x = v[30 + 7 * k];
        ph.o_y = v[31 + 7 * k]; ph.o_x = v[32 + 7 * k]; ph.w_off = (long long)k * KT * p.Npad * 32;
        macs += (double)ph.M * KT * 32 * p.N;       // padded-K MACs actually issued
    }
    const size_t in_n = (size_t)p.B * p.Hi * p.Wi * p.Cs_in, out_n = (size_t)p.B * p.Ho * p.Wo * p.Cs_out;
    const size_t w_n = (size_t)p.nphase * KT * p.Npad * 32, part_n = (size_t)p.nphase * p.ksplit * p.Mmax * p.Npad;
    std::vector<float> h(std::max(in_n, w_n));
    const char *fill = getenv("VSTAB_BENCH_FILL");             // rand (default) | zero | const | sparse: operand toggling moves the clocks
    const int fmode = !fill ? 0 : (fill[0] == 'z' ? 1 : (fill[0] == 'c' ? 2 : (fill[0] == 's' ? 3 : 0)));
    for (auto &x : h) {
        const float r = (float)rand() / RAND_MAX - 0.5f;
        x = fmode == 0 ? r : (fmode == 1 ? 0.f : (fmode == 2 ? 0.25f : (r > 0.f ? r : 0.f)));
    }
    float *din, *dw, *db, *dout, *dpart;
    CK(hipMalloc(&din, in_n * 4)); CK(hipMalloc(&dw, w_n * 4)); CK(hipMalloc(&db, p.Npad * 4)); CK(hipMalloc(&dout, out_n * 4));
    CK(hipMalloc(&dpart, (part_n + 4) * 4));
    CK(hipMemcpy(din, h.data(), in_n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, h.data(), w_n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(db, 0, p.Npad * 4));
    p.in_bytes = (unsigned)(in_n * 4); p.w_bytes = (unsigned)((size_t)KT * p.Npad * 128);
    p.in = din; p.wpk = dw; p.bias = db; p.out = dout; p.partial = dpart;
    CK(conv_set_attributes());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(launch_conv(p, (ConvTile)tile, vec4, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) CK(launch_conv(p, (ConvTile)tile, vec4, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const int BN = (tile == 0 || tile == 3) ? 128 : ((tile == 1 || tile == 4) ? 64 : 32), BM = (tile == 3 || tile == 4) ? 64 : (tile == 5 ? 256 : 128);
    const long long blocks = (long long)((p.Mmax + BM - 1) / BM) * (p.Npad / BN) * p.nphase * p.ksplit;
#ifdef VSTAB_STAMP
    {   // per-workgroup phases of the LAST launch, in shader cycles (s_memtime), and the clock the kernel ran at
        const size_t nb = (size_t)std::min<long long>(blocks, 8192);
        std::vector<unsigned long long> st(nb * 8);
        CK(vstab::conv_read_stamps(st.data(), st.size()));
        std::vector<double> pro, loop, epi, tot, clk;
        unsigned long long t_first = ~0ull, t_last = 0;
        for (size_t b = 0; b < nb; ++b) {
            const unsigned long long *s = &st[b * 8];
            if (!s[3] || s[3] < s[0]) continue;
            pro.push_back((double)(s[1] - s[0])); loop.push_back((double)(s[2] - s[1])); epi.push_back((double)(s[3] - s[2])); tot.push_back((double)(s[3] - s[0]));
            if (s[5] > s[4]) clk.push_back((double)(s[3] - s[0]) / (double)(s[5] - s[4]) * 100.0);      // MHz: s_memrealtime ticks at 100 MHz
            t_first = std::min(t_first, s[4]); t_last = std::max(t_last, s[5]);
        }
        auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        const int ktiles = (KT + p.ksplit - 1) / p.ksplit;
        const double mfma_per_tile = 4.0 * (BM / 64) * (BN / 64) * 4 * 64;        // cycles of matrix-pipe time per K-tile per wave (64 per v_mfma_f32_32x32x2)
        const double l = med(loop);
        printf("stamps (%zu workgroups, medians, shader cycles): prologue %.0f  loop %.0f = %.1f per K-tile (matrix pipe %.0f -> %.3f)  epilogue %.0f  total %.0f;"
               " clock %.0f MHz; kernel span %.1f us\n", pro.size(), med(pro), l, l / ktiles, mfma_per_tile, mfma_per_tile * ktiles / l, med(epi), med(tot), med(clk),
               (double)(t_last - t_first) / 100.0);
    }
#endif
    printf("layer %2d tile %d ksplit %2d blocks %6lld KT %4d  %8.2f us  %7.1f TF(issued)  frac %.3f\n", layer, tile, p.ksplit, blocks, KT,
           ms * 1e3, 2 * macs / (ms * 1e-3) / 1e12, 2 * macs / (ms * 1e-3) / 1e12 / 157.3);
    return 0;
}
