// Few-row convolutions / transposed convolutions as weight STREAMS with the split-K reduction finished INSIDE the launch: the last
// encoder stages and first decoder steps of a small one-sample forward (model.py:838-852 at main:568-569's one sess.run per frame;
// BASELINE configs[0]: conv5 ... conv6_1, deconv5, deconv4 of one 256x256 sample have 4 ... 64 GEMM rows per phase against 9 ... 38 MB of
// weights).
//
// As launches of conv_mfma_kernel these layers run 64x128 tiles at split-K 16 ... 48 and a combine launch over that many slabs.  Here:
//   * no LDS in the K loop, no barriers: a WAVE owns 32 rows x 32 columns x a K range and loads its MFMA fragments straight from
//     global memory into registers.  The packed weights ([K-tile][Npad][32], 16-byte chunks XOR-swizzled: pack.cpp) are already laid
//     out so that lane (i, h)'s four 16-byte pieces of a K-tile are its B fragments for the 16 k-steps (the map conv_mfma.hip reads
//     out of LDS: k = 8q + 4h + j); the A fragments are the same 16-byte im2col gathers conv_mfma.hip's fetch makes, one row per lane;
//   * D = 4 K-tiles of prefetch per wave in registers: the loads are plain `buffer_load_dwordx4`, so hipcc's counted
//     `s_waitcnt vmcnt(n)` keeps the ring full (the order of requests and MFMAs is pinned with sched_barriers, see below);
//   * a workgroup = 4 waves = 4 consecutive quarters of one K slice; their partial 32x32 blocks are summed through LDS in wave order
//     (deterministic), so the split-K factor over workgroups stays 4x smaller than the number of K ranges;
//   * split-K over workgroups ends inside the launch: every workgroup publishes its slab tile write-through (`sc1` 16-byte stores),
//     drains, takes a ticket (agent-scope atomic add); the workgroup that draws the last ticket of a tile makes one agent-scope
//     acquire, sums the tile's slabs in slab order (`sc1` loads; the same association as splitk_combine_kernel), adds the bias,
//     applies the activation and stores the output rows -- no combine launch, no dependence on placement or dispatch order
//     (cdna_hip_programming.md section 5 "In-launch split-K reduction", section 6 Guideline 16).  With 32x32 tiles and the 4-wave
//     pre-sum a tile's slabs are 4 KB x ks (ks <= 16).  The ticket words live in the caller's workspace (round 5; zeroed by the first
//     workgroup of the forward's first launch, conv_rowwin.hip, or by a memset node when that kernel does not run) and every last arriver puts its word back to zero.
//
// What it buys (profiles/README.md "r04 one-sample path", interleaved A/B on one box): the launch itself is a chain of latencies --
// first operands, a handful of MFMAs, LDS pre-sum, store drain, ticket, acquire, slab loads, output -- that takes 12 ... 17 us where the
// tiled kernel takes 11 ... 16 us and its combine launch 5 more: one 256x256 frame 0.375 -> 0.358 ms with every layer of <= 64 rows
// per phase on this kernel.  It does NOT pay above that: a lane's operand loads are one row each (the MFMA wants rows across lanes),
// so a wave instruction touches 32 cache lines whatever it fetches and the address unit, not HBM, bounds the stream -- at 48 ... 192
// rows per phase (one 384x512 sample) the tiled kernel + combine launch is as fast or faster, and the plan keeps it there
// (SKINNY_MAX_ROWS).
#include <algorithm>

#include <hip/hip_ext.h>

#include "vstab_internal.h"

namespace vstab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) unsigned gu32;

template <int MB, int D>
__global__ __launch_bounds__(256) void conv_skinny_kernel(const ConvParams p, unsigned *counters)
{
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    float *sR = reinterpret_cast<float *>(sk_smem);                   // [4 waves][MB][32 rows][32 cols]; its first word doubles as the ticket broadcast

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int li = lane & 31, lh = lane >> 5, sw = (li >> 1) & 7;
    unsigned bx_, by_, bz_;
    {
        // XCD-aware numbering (the hardware deals workgroups to the 8 XCDs round-robin; xcd_remap_calc gives XCD c a contiguous band of the
        // new linear id), decomposed column block fastest, then ROW TILE, then (phase, K slice): one XCD's band is whole K slices with ALL
        // their row tiles and column blocks, so a slice's weights -- the big operand -- are fetched into one L2 once and the second row
        // tile of a 33 ... 64-row layer reads them there (rounds 1-4 put the row tile slowest: two XCDs streamed the same weights,
        // 61 MB from HBM for conv6_1's 37.7 MB at one 384x512 sample).  The A rows of a slice are shared by its column blocks as before.
        unsigned nl, t0_, t1_;
        xcd_remap_calc(gridDim.x * gridDim.y * gridDim.z, 1, 1, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nl, t0_, t1_);
        by_ = nl % gridDim.y;
        const unsigned t2 = nl / gridDim.y;
        bx_ = t2 % gridDim.x;
        bz_ = t2 / gridDim.x;
    }
    const int z = (int)bz_;
    const int phase = z / p.ksplit, split = z - phase * p.ksplit;
    const ConvPhase ph = p.ph[phase];
    const int m0 = (int)bx_ * (32 * MB);
    if (m0 >= ph.M) return;             // uniform for the workgroup (phases of unequal size): such a tile has no ticket word to complete
    const int n0 = (int)by_ * 32;

    const bool own = ph.KH != 0;
    const int L_KH = own ? ph.KH : p.KH, L_NSEG = own ? ph.NSEG : p.NSEG, L_SEG = own ? ph.SEG : p.SEG;
    const int L_SEGP = own ? ph.SEGP : p.SEGP, L_STRIDE = own ? ph.SEG_STRIDE : p.SEG_STRIDE;
    const int kps = L_SEGP >> 5;
    const int KT = L_KH * L_NSEG * kps;
    const int kts = (KT + p.ksplit - 1) / p.ksplit;
    const int g0 = split * kts, g1 = min(KT, g0 + kts);               // this workgroup's K-tiles
    const int per = (max(g1 - g0, 0) + 3) >> 2;                       // ... in four consecutive quarters, one per wave
    const int kt0 = min(g1, g0 + wave * per), kt1 = min(g1, kt0 + per);
    const int n = kt1 - kt0;

    // ---- this lane's MB rows: {input element offset of the run start, iy0, x-range low, x-range high}
    int4 R[MB];
    const int hw = ph.Hg * ph.Wg;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = m0 + mb * 32 + li;
        R[mb] = make_int4(0, -(1 << 28), 0, 0);
        if (m < ph.M) {
            const int nn = m / hw, rem = m - nn * hw;
            const int j = rem / ph.Wg, i = rem - j * ph.Wg;
            const int iy0 = j * p.s_in + ph.off_y, ix0 = i * p.s_in + ph.off_x;
            R[mb] = make_int4(((nn * p.Hi + iy0) * p.Wi + ix0) * p.Cs_in, iy0, -ix0 * p.Cs_in, (p.Wi - ix0) * p.Cs_in);
        }
    }
    const int row_pitch = p.Wi * p.Cs_in;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wpk + ph.w_off), 0, p.w_bytes, 0x00020000);
    const unsigned OOB = 0xC0000000u;
    // this lane's four 16-byte pieces of a packed 32-float weight row: chunk (2q + h) ^ swizzle
    unsigned wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[q] = (unsigned)((n0 + li) * 32 + (((2 * q + lh) ^ sw) << 2)) * 4u;
    const unsigned wstep = (unsigned)p.Npad * 128u;

    // K-tile cursor (row tap, segment, chunk) of the next tile to request
    int c_ky, c_sg, c_kc, c_left = n;
    {
        const int per_row = L_NSEG * kps;
        c_ky = kt0 / per_row;
        const int r = kt0 - c_ky * per_row;
        c_sg = r / kps;
        c_kc = r - c_sg * kps;
    }
    unsigned c_woff = (unsigned)kt0 * wstep;

    f32x4 fa[D][MB][4], fb[D][4];
    auto request = [&](f32x4 (&a)[MB][4], f32x4 (&b)[4]) {
        const bool live = c_left > 0;                                 // wave uniform; past the end every lane asks out of range: zeros, no traffic
        const int ky = c_ky, qseg = c_kc * 32, qabs0 = c_sg * L_STRIDE + qseg;
        {
            const int kc1 = c_kc + 1;
            const bool wrap_kc = kc1 == kps;
            const int sg1 = c_sg + (wrap_kc ? 1 : 0);
            const bool wrap_sg = sg1 == L_NSEG;
            c_kc = wrap_kc ? 0 : kc1;
            c_sg = wrap_sg ? 0 : sg1;
            c_ky += wrap_sg ? 1 : 0;
            --c_left;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rwt, live ? wq[q] + c_woff : OOB, 0, 0));
        c_woff += wstep;
        const int rowoff = ky * row_pitch + qabs0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const bool yok = live & ((unsigned)(R[mb].y + ky) < (unsigned)p.Hi);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = (2 * q + lh) * 4;
                const int qa = qabs0 + c4;
                const bool ok = yok & (qseg + c4 < L_SEG) & (qa >= R[mb].z) & (qa < R[mb].w);
                const unsigned off = ok ? (unsigned)(R[mb].x + rowoff + c4) * 4u : OOB;
                a[mb][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0));
            }
        }
    };

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    auto mm = [&](const f32x4 (&a)[MB][4], const f32x4 (&b)[4]) {       // same k order as conv_mfma.hip's loop: (K-tile, q, j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][q][jj], b[q][jj], acc[mb], 0, 0, 0);
    };
    // The ring: D tiles requested up front; a steady-state trip multiplies stage s and at once requests tile t + s + D into the
    // registers just consumed.  The sched_barriers keep hipcc from sinking a stage's requests to the end of the trip (it did:
    // the wait at the loop head then became vmcnt(0) and the ring one tile deep); with the order pinned the waits it inserts
    // are counted ones that leave D - 1 tiles in flight.
#pragma unroll
    for (int s = 0; s < D; ++s) request(fa[s], fb[s]);
    int t = 0;
    for (; t + D <= n; t += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            mm(fa[s], fb[s]);
            __builtin_amdgcn_sched_barrier(0);
            request(fa[s], fb[s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
        if (t + s < n) mm(fa[s], fb[s]);                              // wave uniform: the last n % D tiles (already requested)

    // ---- the four waves' partial blocks -> LDS -> summed in wave order
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            sR[((wave * MB + mb) * 32 + row) * 32 + li] = acc[mb][r];
        }
    __syncthreads();
    f32x4 sum[MB];
    // thread -> (row, 16-byte column group) of the tile, MB elements per thread
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int e = tid + 256 * i, row = e >> 3, c4 = (e & 7) * 4;
        f32x4 s = *reinterpret_cast<const f32x4 *>(sR + ((0 * MB + (row >> 5)) * 32 + (row & 31)) * 32 + c4);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4 *>(sR + ((w * MB + (row >> 5)) * 32 + (row & 31)) * 32 + c4);
        sum[i] = s;
    }

    const int ks = p.ksplit;
    if (ks > 1) {
        const long long slab = (long long)p.Mmax * p.Npad;            // floats per (phase, split) slab
        const __amdgpu_buffer_rsrc_t rpz = __builtin_amdgcn_make_buffer_rsrc(
            p.partial + (long long)phase * ks * slab, 0, (unsigned)min((long long)ks * slab * 4, 0xFFFFFFFFLL), 0x00020000);
        // publish this slice's tile write-through, drain, ticket
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int e = tid + 256 * i, row = e >> 3, c4 = (e & 7) * 4;
            if (m0 + row < ph.M) {
                const unsigned off = (unsigned)(((long long)split * slab + (long long)(m0 + row) * p.Npad + n0 + c4) * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, sum[i]), rpz, off, 0, 16);      // aux 16 = sc1
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains its own stores ...
        __syncthreads();                                              // ... before the one lane that signals for all of them (also: sR is free again)
        unsigned *flag = reinterpret_cast<unsigned *>(sR);
        gu32 *cnt = (gu32 *)(counters + ((size_t)phase * gridDim.x + bx_) * gridDim.y + by_);
        if (tid == 0) *flag = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *flag;
        if (ticket != (unsigned)(ks - 1)) return;                     // uniform: not the last slice of this tile
        if (tid == 0) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the word is zero again for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int e = tid + 256 * i, row = e >> 3, c4 = (e & 7) * 4;
            const unsigned off0 = (unsigned)(((long long)(m0 + row) * p.Npad + n0 + c4) * 4);
            const unsigned sstep = (unsigned)(slab * 4);
            const bool rok = m0 + row < ph.M;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 8 <= ks; k += 8) {                             // eight slab loads in flight, added in slab order (= splitk_combine_kernel)
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpz, rok ? off0 + (unsigned)(k + u) * sstep : OOB, 0, 16));
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            if (k < ks) {
                f32x4 v[7];
#pragma unroll
                for (int u = 0; u < 7; ++u) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpz, (rok && k + u < ks) ? off0 + (unsigned)(k + u) * sstep : OOB, 0, 16));
#pragma unroll
                for (int u = 0; u < 7; ++u)
                    if (k + u < ks) s += v[u];
            }
            sum[i] = s;
        }
    }

    // ---- bias, activation, output rows (the workgroup of an unsplit launch, or the last arriver of a tile)
    const float slope = p.act == 1 ? 0.1f : 0.0f;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int e = tid + 256 * i, row = e >> 3, c4 = (e & 7) * 4;
        const int m = m0 + row, col = n0 + c4;
        if (m >= ph.M || col >= p.N) continue;
        const int nn = m / hw, rem = m - nn * hw;
        const int j = rem / ph.Wg, ii = rem - j * ph.Wg;
        float *o = p.out + ((long long)(nn * p.Ho + j * p.s_out + ph.o_y) * p.Wo + ii * p.s_out + ph.o_x) * p.Cs_out + p.c_off + col;
        f32x4 v = sum[i] + *reinterpret_cast<const f32x4 *>(p.bias + col);
        if (p.out_vec4 && col + 4 <= p.N) {
            if (p.act == 3) v += *reinterpret_cast<const f32x4 *>(o);
            else if (p.act) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], slope * v[c]);
            }
            *reinterpret_cast<f32x4 *>(o) = v;
        } else {
            for (int c = 0; c < 4 && col + c < p.N; ++c) {
                float x = v[c];
                if (p.act == 3) x += o[c];
                else if (p.act) x = fmaxf(x, slope * x);
                o[c] = x;
            }
        }
    }
}

// does a launch qualify?  16-byte operand path, whole 32-column blocks of packed weights, few rows
bool conv_skinny_applicable(const ConvParams &p, bool vec4)
{
    if (!vec4 || (p.Npad & 31) || (p.N & 3) || (p.Cs_in & 3) || (p.SEG & 3) || (p.SEG_STRIDE & 3)) return false;
    if (p.Mmax < 1 || p.Mmax > SKINNY_MAX_ROWS || p.nphase < 1 || p.nphase > 16) return false;
    return skinny_tiles(p) <= SKINNY_MAX_TILES;
}

long long skinny_tiles(const ConvParams &p)
{
    return (long long)((p.Mmax + 31) / 32) * (p.Npad / 32) * p.nphase;
}

// split-K factor over WORKGROUPS (each splits its slice four ways again): about ONE workgroup per CU, at least two K-tiles per wave.
// Round 5 A/B (profiles/ab_r05j_skinny_target.txt): 256 workgroups beat 512 (conv6 / conv6_1 of one 384x512 sample 18.4 / 21.6 -> 16.6 /
// 19.6 us, conv6_1 of one 256x256 sample 15.0 -> 13.5 us) and 192 / 128 / 1024 all lose: the launch is a chain of latencies whose
// reduction end grows with the number of slabs, and half as many workgroups publish half as many.
int conv_skinny_split(const ConvParams &p, int cap)
{
    const int KT = p.KH * p.NSEG * (p.SEGP / 32);
    const long long base = skinny_tiles(p);
#ifndef VSTAB_SKINNY_TARGET
#define VSTAB_SKINNY_TARGET 256          // workgroups a launch aims for (A/B builds: scripts/build_variant_lib.sh -DVSTAB_SKINNY_TARGET=...)
#endif
    long long ks = (VSTAB_SKINNY_TARGET + base / 2) / base;
    ks = std::min<long long>(ks, KT / 8);
    ks = std::min<long long>(ks, cap);
    if (ks < 1) ks = 1;
    const int kts = (int)((KT + ks - 1) / ks);
    return (KT + kts - 1) / kts;
}

hipError_t launch_conv_skinny(const ConvParams &p_in, unsigned *counters, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    ConvParams p = p_in;
    if (!conv_skinny_applicable(p, true) || p.ksplit < 1) return hipErrorInvalidValue;
    if (p.in_bytes >= 0x80000000u || p.w_bytes >= 0x80000000u) return hipErrorInvalidValue;
    if (p.ksplit > 1 && (p.partial == nullptr || ((uintptr_t)p.partial & 15) || counters == nullptr)) return hipErrorInvalidValue;
    if ((long long)p.ksplit * p.Mmax * p.Npad * 4 >= 0x100000000LL) return hipErrorInvalidValue;      // one phase's slabs behind one descriptor
    p.no_remap = 0;
    p.out_vec4 = (((uintptr_t)p.out & 15) == 0 && (p.Cs_out & 3) == 0 && (p.c_off & 3) == 0) ? 1 : 0;
    // 32-row tiles always: with 64-row ones (two MFMA row blocks per wave, half the workgroups) every launch measured slower -- a lane's
    // operand loads are one row each, so a wave instruction touches 32 lines whatever it fetches, and twice the loads per K-tile queue
    // behind one address unit (profiles/ab_r04d_*: one sample at 256x256 0.371 -> 0.358 ms, at 384x512 0.607 -> 0.589)
    const dim3 grid((p.Mmax + 31) / 32, p.Npad / 32, p.nphase * p.ksplit), block(256);
    const size_t lds = (size_t)4 * 32 * 32 * 4;
    const bool timed = ev_start != nullptr && ev_stop != nullptr;
#define VSTAB_LAUNCH(KERNEL)                                                                                   \
    do {                                                                                                        \
        if (timed) hipExtLaunchKernelGGL(KERNEL, grid, block, lds, stream, ev_start, ev_stop, 0, p, counters);   \
        else KERNEL<<<grid, block, lds, stream>>>(p, counters);                                                 \
    } while (0)
    VSTAB_LAUNCH((conv_skinny_kernel<1, 4>));
#undef VSTAB_LAUNCH
    return hipGetLastError();
}

}  // namespace vstab
