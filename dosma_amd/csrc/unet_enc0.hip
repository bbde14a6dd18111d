// unet_enc0.hip -- the first encoder block of the parity-mode ("fp16x3") U-Net as ONE kernel, gfx950 only:
//
//     Conv2D(32, 3x3) on the single-channel image + ReLU -> Conv2D(32, 3x3) + ReLU -> BatchNorm -> [skip] -> MaxPooling2D(2x2)
//     /root/reference/dosma/models/oaiunet2d.py:213-243 (level 0 of the contracting path)
//
// Why: at 384 x 384 x 160 slices a 32-channel split feature map is 3 GB.  As three kernels (first convolution, conv_s3_kernel,
// pooling) the block moved 12.75 GB through HBM -- the first feature map written and read back, the block output read back
// for pooling -- at 3.3-3.9 TB/s: it was bound by that traffic, not by the matrix pipes.  Here the first feature map exists only
// as the halo image of the second convolution in LDS and the pooled map is made from the staged output tile: 0.1 GB in
// (the image), 3.75 GB out.
//
// One persistent block of 8 waves per CU = TWO GROUPS of 4 waves (one per SIMD each), each working on its own tile of 8 rows x 32
// columns (a wave: two rows), half a tile period apart: while one group multiplies (conv 2), the other does the vector work of
// its tile on the same SIMDs (enc0_kernel's main loop has the details and what was measured).  (Two 4-wave blocks per CU with
// 4 x 32 tiles measured 1.79 instead of 1.63 ms: 1.59 x halo instead of 1.33 x; with 8 x 32 tiles they need 2 x 82.1 KB of LDS.)
// Per tile:
//   patch   12 x 36 input pixels (prefetched into registers during the previous tile's MFMA loop)           -> LDS
//   conv 1  on the 10 x 34 halo, as MFMA too: K = 16 = nine taps + a constant-one tap that carries the bias, operands
//           split into fp16 hi + lo parts like everywhere else (three MFMAs per 32 pixels); ReLU; split; halo image
//           (zero outside the slice: that is conv 2's SAME padding)
//   conv 2  nine taps x two k-steps x (hi hi + hi lo + lo hi); its 36 KB of weights are LDS-RESIDENT (no ring, no
//           request stream, no counted waits); A = weights, B = pixels, so that a lane of the accumulator tile holds
//           ONE pixel and a register one channel
//   out     bias, ReLU, BatchNorm affine, split; [pixel][hi 64 B | lo 64 B] image through a wave window in LDS (eight 8-byte
//           writes per lane instead of thirty-two 2-byte ones) -- the windows lie IN the group's halo image, dead by then;
//           128-byte pixel-chunk stores of the skip tensor; each wave pools its two staged rows 2 x 2 and stores a row of the
//           next level's input
//
// The same file holds the two other 32-channel layers of the top level: mid0_kernel (Conv2D 64 -> 32) and out0_kernel (last
// convolution + classifier).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "qmri_internal.h"

namespace qmri {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __fp16 h16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kRows = 8;                // rows of a tile (x 32 columns)
constexpr int kGW = 4;                  // waves of a GROUP (one per SIMD): a wave owns two rows of its group's tile
constexpr int kWaves = 2 * kGW;         // two groups per block, half a tile period apart (see enc0_kernel)
constexpr int kThreads = kWaves * 64;
constexpr int kGThreads = kGW * 64;
constexpr int kPitch = 34;              // halo row pitch of the kRows x 32 tile
constexpr int kHalo = (kRows + 2) * kPitch;        // halo pixels
constexpr int kGroups = (kHalo + 31) / 32;         // 32-pixel groups of the first convolution (the last one is partly empty)
constexpr int kHaloBytes = kHalo * 128;
constexpr int kPatchW = 36, kPatchH = kRows + 4;
constexpr int kPatchFloats = kPatchW * (kPatchH + 1);  // (+ a row: the discarded pixels of the last conv 1 group read past the patch)
constexpr int kWBytes = 9 * 4096;       // conv 2 weights: [tap][plane][32 channels][64 B], conv_s3_kernel's slot image
constexpr int kSplitM = 8;              // conv 2's 18 half-steps run as [0, kSplitM) | barrier | [kSplitM, 18): see enc0_kernel
static_assert(kGW * 8192 <= kHaloBytes, "a group's staging windows (two rows per wave) live in its halo image");

// LDS position of the 16-byte piece (plane, q = channels 8 q .. 8 q + 7) of halo pixel hp
__device__ __forceinline__ int halo_off(int hp, int plane, int q) {
    return hp * 128 + (((plane ^ ((hp >> 1) & 1)) * 4 + (q ^ ((hp >> 2) & 3))) * 16);
}
// staging window of a wave: pixel px (0..31), piece p8 = plane * 4 + q at position p8 ^ ((px >> 1) & 7)
__device__ __forceinline__ int stage_off(int px, int p8) { return px * 128 + ((p8 ^ ((px >> 1) & 7)) * 16); }

__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
    const h16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    hi = __builtin_bit_cast(unsigned, h);
    // a - float(hi part), one v_fma_mix_f32 each (a * 1.0 - hi: a single rounding like the subtraction; hipcc emits v_cvt_f32_f16 + v_sub_f32)
    float d0, d1;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(d0) : "v"(a), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d1) : "v"(b), "v"(hi));
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d0, d1));
}
// the same, tracking max |value| for the saturation flag of the split layout (qmri_internal.h: ConvS3Args::sat)
__device__ __forceinline__ void split2m(float a, float b, unsigned &hi, unsigned &lo, float &amax) {
#ifndef QMRI_NO_SAT_TRACK  // (timing experiment: what the saturation tracking costs)
    amax = fmaxf(fmaxf(amax, fabsf(a)), fabsf(b));  // one v_max3_f32 with |.| modifiers
#endif
    split2(a, b, hi, lo);
}

#ifdef QMRI_S3_EXPERIMENTS
__device__ unsigned long long enc0_tstat[8];  // cycles of wave 0: [0] MFMA loop [1] barrier A [2] output staging [3] barrier B [4] conv 1 + stores + pool [5] barrier C [6] tiles
#define ENC0_T(i)                                                      \
    {                                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        tacc[i] += now_ - tmark;                                       \
        tmark = now_;                                                  \
    }
#else
#define ENC0_T(i)
#endif

// LDS traffic only: the skip / pooled stores of a tile stay in flight across the barriers (__syncthreads would drain them)
#define S_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#ifdef QMRI_ENC0_TIMELINE  // (experiment build: block 8 records when its two groups reach the marks of half-periods 20 .. 23, 10 ns ticks;
                           //  read back with qmri_debug_enc0_timeline -- scripts/enc0_timeline.py)
__device__ unsigned long long g_enc0_tl[4 * 2 * 8];
#define ENC0_TS(k)                                                                                                   \
    do {                                                                                                             \
        if (blockIdx.x == 8 && (tid & 255) == 0 && h >= 20 && h < 24) g_enc0_tl[((h - 20) * 2 + grp) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define ENC0_TS(k)
#endif

__global__ __launch_bounds__(kThreads, 1) void enc0_kernel(const Enc0Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *wlds = smem;                         // conv 2 weights (both groups)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / kGW, wg = wave % kGW;        // group, wave of the group (a workgroup's waves 0-3 and 4-7 each cover the four SIMDs)
    const int gtid = tid & (kGThreads - 1);
    unsigned char *halo = wlds + kWBytes + grp * kHaloBytes;                                              // this group's conv 1 output on the halo
    float *patch = reinterpret_cast<float *>(wlds + kWBytes + 2 * kHaloBytes) + grp * kPatchFloats;      // its [12][36] input pixels
    unsigned char *win0 = halo + wg * 8192;             // this wave's two staging windows (rows 2 wg, 2 wg + 1): IN the halo image, dead by then
    const int l31 = lane & 31, kgrp = lane >> 5;
    float amax = 0.f;  // max |v| of everything this lane split (input pixels, both feature maps): Enc0Args::sat

    // ---- once per block: resident weights, per-lane operand constants ----
    for (int i = tid; i < kWBytes / 16; i += kThreads)
        reinterpret_cast<uint4 *>(wlds)[i] = reinterpret_cast<const uint4 *>(A.w2)[i];
    // conv 1 as a 32 (channels) x 16 (taps) A operand: lane = (channel l31, taps 8 kgrp ..), hi and lo parts
    const f16x8 c1h = reinterpret_cast<const f16x8 *>(A.c1_img)[lane * 2];
    const f16x8 c1l = reinterpret_cast<const f16x8 *>(A.c1_img)[lane * 2 + 1];
    // epilogue parameters of this lane's 16 channels: register e <-> channel (e & 3) + 8 (e >> 2) + 4 kgrp
    float pb[16], ps[16], pt[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ch = (e & 3) + 8 * (e >> 2) + 4 * kgrp;
        pb[e] = A.bias2[ch];
        ps[e] = A.scale2[ch];
        pt[e] = A.shift2[ch];
    }
    // conv 2, A operand (weights): byte offset of this lane's piece inside a tap image, k-step 0 (k-step 1: ^ 32; lo: + 2048)
    const int woff = l31 * 64 + ((kgrp ^ ((l31 >> 2) & 3)) * 16);
    // conv 2, B operand (pixels): halo pixel of this lane's output pixel for tap (0, 0): rows 2 wg + r, column l31
    int boff[2][9];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int hp0 = (2 * wg + r + 1) * kPitch + l31 + 1;
#pragma unroll
        for (int t = 0; t < 9; ++t) boff[r][t] = halo_off(hp0 + (t / 3 - 1) * kPitch + (t % 3 - 1), 0, kgrp);
    }

    const int tiles_x = A.W / 32, tiles_y = A.H / kRows;
    const int per_img = tiles_x * tiles_y;
    const int ntiles = A.B * per_img;
    auto tile_origin = [&](int t, int &b, int &y0, int &x0) {
        b = t / per_img;
        const int r = t - b * per_img;
        const int ty = r / tiles_x;
        y0 = ty * kRows;
        x0 = (r - ty * tiles_x) * 32;
    };
    // XCD x (= blockIdx % 8: its own L2) walks a CONTIGUOUS eighth of the tiles, its CUs side by side in it: neighbouring tiles share
    // halo rows / columns through that L2 instead of fetching them from HBM once per XCD
    int t_first = blockIdx.x, t_stride = gridDim.x, t_end = ntiles;
    if ((gridDim.x & 7) == 0) {
        const int per = (ntiles + 7) / 8, xcd = blockIdx.x & 7;
        t_first = xcd * per + (blockIdx.x >> 3);
        t_stride = gridDim.x >> 3;
        t_end = (xcd + 1) * per < ntiles ? (xcd + 1) * per : ntiles;
    }
    if (t_first >= t_end) return;
    const int n_blk = (t_end - t_first + t_stride - 1) / t_stride;  // tiles of this block: the groups take them alternately
    const int gstride = 2 * t_stride;
    // pixel i of the (kRows + 4) x 36 input patch of a tile, zero outside the slice (a thread owns pixels gtid and gtid + kGThreads)
    auto load_patch = [&](int t, int i) -> float {
        if (t >= t_end || i >= kPatchW * kPatchH) return 0.f;
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
        const int r = i / kPatchW, c = i - r * kPatchW;
        const int yy = y0 - 2 + r, xx = x0 - 2 + c;
        if ((unsigned)yy >= (unsigned)A.H || (unsigned)xx >= (unsigned)A.W) return 0.f;
        return A.x[((long long)b * A.H + yy) * A.W + xx];
    };
    static_assert(kPatchW * kPatchH <= 2 * kGThreads, "two patch pixels per thread");
    auto store_patch = [&](float v0, float v1) {
        patch[gtid] = v0;
        if (gtid + kGThreads < kPatchW * kPatchH) patch[gtid + kGThreads] = v1;
    };
    // conv 1 on one group of 32 consecutive halo pixels -> halo image
    auto conv1_group = [&](int g, int y0, int x0) {
        const int hp = g * 32 + l31;
        const int r = hp / kPitch, c = hp - r * kPitch;  // halo row / column (pixels beyond the halo: r = kRows + 2, discarded)
        float tp[8];
        if (kgrp == 0) {  // taps 0..7: (r + dy, c + dx) of the patch, whose origin is (y0 - 2, x0 - 2)
            const float *p = patch + r * kPatchW + c;
            tp[0] = p[0]; tp[1] = p[1]; tp[2] = p[2];
            tp[3] = p[kPatchW]; tp[4] = p[kPatchW + 1]; tp[5] = p[kPatchW + 2];
            tp[6] = p[2 * kPatchW]; tp[7] = p[2 * kPatchW + 1];
        } else {          // tap 8, the constant-one tap of the bias, six zeros
            tp[0] = r < kRows + 2 ? patch[(r + 2) * kPatchW + c + 2] : 0.f;
            tp[1] = 1.f;
#pragma unroll
            for (int i = 2; i < 8; ++i) tp[i] = 0.f;
        }
        unsigned bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split2m(tp[2 * i], tp[2 * i + 1], bh[i], bl[i], amax);
        const f16x8 xh = __builtin_bit_cast(f16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
        const f16x8 xl = __builtin_bit_cast(f16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1l, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1h, xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1h, xh, acc, 0, 0, 0);
        // this lane: halo pixel hp, channels (e & 3) + 8 (e >> 2) + 4 kgrp
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const bool inside = hp < kHalo && (unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W;
        const float sc = inside ? A.c1_winv : 0.f;  // outside the slice: conv 2's zero padding
        if (hp < kHalo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h0, l0, h1, l1;
            split2m(fmaxf(acc[4 * q] * sc, 0.f), fmaxf(acc[4 * q + 1] * sc, 0.f), h0, l0, amax);
            split2m(fmaxf(acc[4 * q + 2] * sc, 0.f), fmaxf(acc[4 * q + 3] * sc, 0.f), h1, l1, amax);
            *reinterpret_cast<uint2 *>(halo + halo_off(hp, 0, q) + 8 * kgrp) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(halo + halo_off(hp, 1, q) + 8 * kgrp) = make_uint2(l0, l1);
        }
    };

    // (the discarded pixels of the last conv 1 group read one row past the patch: their products go nowhere, but their operands
    //  enter the saturation tracking -- that row is zero, not whatever the previous kernel left in LDS)
    if (gtid < kPatchW) patch[kPatchW * kPatchH + gtid] = 0.f;
    // ---- prologue: patch and conv 1 of each group's first tile ----
    int cur = t_first + grp * t_stride;  // the tile whose halo image this group holds
    int t_b = 0, t_y0 = 0, t_x0 = 0;
    if (cur < t_end) {
        tile_origin(cur, t_b, t_y0, t_x0);
        store_patch(load_patch(cur, gtid), load_patch(cur, gtid + kGThreads));
    }
    __syncthreads();
    if (cur < t_end)
        for (int g = wg; g < kGroups; g += kGW) conv1_group(g, t_y0, t_x0);
    __syncthreads();

    // ---- THE TWO GROUPS TAKE TURNS ON THE MATRIX PIPES ----
    // Time runs in half-periods h = 0, 1, ...; in half-period h group h & 1 multiplies its tile (conv 2: 108 MFMAs per wave, a
    // wave alone on its SIMD's matrix pipe) while the OTHER group, on the same four SIMDs, does everything that is not conv 2 for
    // the tile it multiplied in the half-period before: output arithmetic, staging, skip stores, pooling, then conv 1 of its next
    // tile.  Both branches hold two s_barriers -- the block's waves arrive at the same barriers from different code: group-local
    // hand-overs (halo image complete / staging windows read) ride on them, the groups share nothing but the weights.
    // Measured (160 slices of 384 x 384, same box, alternating): 1.64-1.70 ms against 1.71-1.79 for the block of eight waves in
    // ONE phase at a time (MfmaUtil 0.42) -- 4.5 %, not the 40 % the MFMA share promises: the vector work of a tile (~930
    // instructions per wave, LDS round trips between them) takes a lone wave per SIMD 2.4 x its multiply phase, and LDS time --
    // operand reads at 1.0 per MFMA + staging + halo image, ~3.8 k cycles per tile -- is not far below the period.  Also tried:
    // three slots (M | O | C: two vector phases on one SIMD just add, 1.72-1.77 ms); two groups of EIGHT waves (16 per block, 112
    // registers, parameters and conv 1 operands from LDS: 1.33 reads per MFMA, no faster); the multiplying group taking three of the
    // other group's eleven conv 1 groups after its loop (+-0); s_setprio 3 on the multiply phase (+-1 %).  The timeline
    // (QMRI_ENC0_TIMELINE, scripts/enc0_timeline.py, profiles/r04_enc0_timeline.txt) says why: beside a vector partner the multiply
    // phase runs at 55-65 % of the matrix pipe's rate (3.0-3.8 us for the 108 MFMAs of a wave = 1.97 us of pipe time), the vector
    // phase takes 1.7 us when its group is the older one and 3.4 us when it is the younger (VALU issue goes by priority, then age):
    // on one SIMD the two streams add more than they overlap (MI355X_MICROARCH.md, "Two waves per SIMD").  The outputs are
    // bit-identical to the one-phase kernel's (scripts/unet_bits.py).
    f32x16 acc[2];
    float pn0 = 0.f, pn1 = 0.f;
    bool pending = false;  // an output tile waits in acc
    for (int h = 0; h <= n_blk; ++h) {
        ENC0_TS(0);
        if ((h & 1) == grp) {
            // ================= conv 2 of tile `cur`: 2 rows x 32 pixels x 32 channels per wave, K = 9 taps x 32 =================
            if (cur < t_end) {
                const int next = cur + gstride;
                pn0 = load_patch(next, gtid), pn1 = load_patch(next, gtid + kGThreads);  // in flight during the MFMA loop
                // Software pipeline, spelled out for the scheduler (hipcc's own order is read, wait for it, multiply): the six
                // operands of half-step s + 2 are read while the six MFMAs of half-step s run, one read per MFMA gap.
                struct Frag {
                    f16x8 wh, wl, xh[2], xl[2];
                };
                auto load_frag = [&](Frag &f, int t, int kk) {
                    const unsigned char *wt = wlds + t * 4096 + (woff ^ (kk * 32));
                    f.wh = *reinterpret_cast<const f16x8 *>(wt);
                    f.wl = *reinterpret_cast<const f16x8 *>(wt + 2048);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int o = boff[r][t] ^ (kk * 32);
                        f.xh[r] = *reinterpret_cast<const f16x8 *>(halo + o);
                        f.xl[r] = *reinterpret_cast<const f16x8 *>(halo + (o ^ 64));
                    }
                };
                auto mma = [&](const Frag &f) {  // (per accumulator the same order as ever: hi hi, lo hi, hi lo)
#pragma unroll
                    for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh, f.xh[r], acc[r], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wl, f.xh[r], acc[r], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh, f.xl[r], acc[r], 0, 0, 0);
                };
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                Frag f[3];  // half-step s (tap s / 2, k-step s % 2) lives in f[s % 3]; its operands are read two half-steps ahead
                load_frag(f[0], 0, 0);
                load_frag(f[1], 0, 1);
                __builtin_amdgcn_sched_barrier(0);  // the head start stays a head start: the groups below count from here
#pragma unroll
                for (int s = 0; s < 18; ++s) {
                    if (s + 2 < 18) load_frag(f[(s + 2) % 3], (s + 2) / 2, (s + 2) % 2);
                    mma(f[s % 3]);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    if (s == kSplitM - 1) {
                        ENC0_TS(1);
                        S_BARRIER();  // (the other group: staging windows read, its conv 1 may overwrite them)
                        ENC0_TS(2);
                    }
                }
                pending = true;
            } else {
                S_BARRIER();
            }
            ENC0_TS(3);
            S_BARRIER();  // (the other group: its next halo image is complete)
            ENC0_TS(4);
        } else {
            // ================= everything else, for the tile multiplied in the previous half-period =================
            const int next = cur + gstride;
            if (pending) {
                // ---- output: bias, ReLU, BatchNorm, split, staged image ([pixel][hi 64 B | lo 64 B], a 4 KB window per row) ----
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    unsigned char *win = win0 + r * 4096;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int e = 4 * q + i;
                            v[i] = fmaf(fmaxf(fmaf(acc[r][e], A.winv2, pb[e]), 0.f), ps[e], pt[e]);
                        }
                        unsigned h0, l0, h1, l1;
                        split2m(v[0], v[1], h0, l0, amax);
                        split2m(v[2], v[3], h1, l1, amax);
                        *reinterpret_cast<uint2 *>(win + stage_off(l31, q) + 8 * kgrp) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2 *>(win + stage_off(l31, 4 + q) + 8 * kgrp) = make_uint2(l0, l1);
                    }
                }
                store_patch(pn0, pn1);
                // ---- skip tensor: rows t_y0 + 2 wg + r, 32 pixels x 128 B; 8 lanes = one pixel-chunk (pieces permuted by the window
                // swizzle).  The windows are this wave's own: no barrier between their writes and these reads ----
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const unsigned char *win = win0 + r * 4096;
                    const long long row = ((long long)t_b * A.H + t_y0 + 2 * wg + r) * A.W + t_x0;
                    unsigned char *ybase = static_cast<unsigned char *>(A.y) + (row * A.ldy + A.yoff) * 4;
                    auto rd = [&](int t) -> uint4 { return *reinterpret_cast<const uint4 *>(win + (t * 8 + (lane >> 3)) * 128 + (lane & 7) * 16); };
                    auto wr = [&](int t, const uint4 &v) {
                        const int px = t * 8 + (lane >> 3);
                        const int p8 = (lane & 7) ^ ((px >> 1) & 7);
                        *reinterpret_cast<uint4 *>(ybase + (long long)px * A.ldy * 4 + p8 * 16) = v;
                    };
                    const uint4 v0 = rd(0), v1 = rd(1), v2 = rd(2), v3 = rd(3);
                    wr(0, v0);
                    wr(1, v1);
                    wr(2, v2);
                    wr(3, v3);
                }
                {
                    // ---- MaxPooling2D(2 x 2): pooled row wg <- this wave's two staged rows; lane = (pooled pixel, 8-channel group) ----
                    const int pp = lane >> 2, g = lane & 3;
                    float m[8];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const unsigned char *w2 = win0 + (s >> 1) * 4096;
                        const int px = 2 * pp + (s & 1);
                        const uint4 hi = *reinterpret_cast<const uint4 *>(w2 + stage_off(px, g));
                        const uint4 lo = *reinterpret_cast<const uint4 *>(w2 + stage_off(px, 4 + g));
                        const unsigned hh[4] = {hi.x, hi.y, hi.z, hi.w}, ll[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const h16x2 a = __builtin_bit_cast(h16x2, hh[i]), b = __builtin_bit_cast(h16x2, ll[i]);
                            const float v0 = (float)a[0] + (float)b[0], v1 = (float)a[1] + (float)b[1];
                            m[2 * i] = s == 0 ? v0 : fmaxf(m[2 * i], v0);
                            m[2 * i + 1] = s == 0 ? v1 : fmaxf(m[2 * i + 1], v1);
                        }
                    }
                    unsigned hq[4], lq[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) split2(m[2 * i], m[2 * i + 1], hq[i], lq[i]);
                    const long long prow = ((long long)t_b * (A.H / 2) + (t_y0 / 2) + wg) * (A.W / 2) + (t_x0 / 2) + pp;
                    unsigned char *dst = static_cast<unsigned char *>(A.pool_y) + prow * A.pool_ld * 4 + g * 16;
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(hq[0], hq[1], hq[2], hq[3]);
                    *reinterpret_cast<uint4 *>(dst + 64) = make_uint4(lq[0], lq[1], lq[2], lq[3]);
                }
            }
            ENC0_TS(1);
            S_BARRIER();  // this group: every wave has read its staging windows, the next patch is visible
            ENC0_TS(2);
            if (pending) {
                // ---- conv 1 of this group's next tile -> its halo image (over the staging windows) ----
                cur = next;
                if (cur < t_end) {
                    tile_origin(cur, t_b, t_y0, t_x0);
                    for (int g = wg; g < kGroups; g += kGW) conv1_group(g, t_y0, t_x0);
                }
                pending = false;
            }
            ENC0_TS(3);
            S_BARRIER();  // this group: halo image complete
            ENC0_TS(4);
        }
    }
    if (A.sat && amax > 65504.f) *A.sat = 1;
}


// ---------------------------------------------------------------------------------------------------------------------
// out0_kernel -- the LAST convolution of the network with the classifier on its back, same skeleton:
//     Conv2D(32, 3x3) + ReLU -> BatchNorm -> Conv2D(n_classes <= 4, 1x1) [-> logit > 0]     oaiunet2d.py:266-289, 305-308
// The 32-channel input (a split tensor in HBM) arrives by LDS-DMA, the halo of tile t + 1 while tile t is multiplied (two halo
// buffers, ONE barrier per tile); the weights are LDS-resident; the classifier runs on the fp32 accumulators -- a lane holds
// 16 channels of ONE pixel, its partner lane (+32) the other 16 -- so the last feature map is never split, staged or stored.
//
// LDS-DMA of 16 bytes per lane (see unet_s3.hip for the M0 convention) through the BUFFER path: one wave-uniform descriptor
// (base = the tile's halo origin) + a 32-bit byte offset per lane; an offset beyond num_records reads as ZERO, which is the
// padding outside the slice (no zero line, no per-lane 64-bit address).  Measured 3-4 % faster than global_load_lds here.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_buf(unsigned voffset, i32x4 rsrc, unsigned lds_dst_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voffset), "s"(rsrc), "s"(lds_dst_wave_base) : "memory");
}
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(size_t)(lds_void *)p; }

constexpr int kO_Waves = 8;
constexpr int kO_Threads = kO_Waves * 64;
constexpr int kO_Halo = (kO_Waves + 2) * kPitch;       // 340 halo pixels
constexpr int kO_NJ = (kO_Halo + 7) / 8;               // 43 DMA instructions of 8 pixel-chunks
constexpr int kO_HaloBytes = kO_NJ * 1024;
constexpr int kO_PerWave = (kO_NJ + kO_Waves - 1) / kO_Waves;  // 6

// NW waves = NW rows of 32 pixels per tile.  12 (three waves per SIMD; the last tile row of a slice may be partly outside it)
// where the per-wave phases between the MFMA loops need the extra latency hiding; 8 is the two-waves-per-SIMD form.
template <int NW>
__global__ __launch_bounds__(NW * 64) void out0_kernel(const Out0Args A) {
    constexpr int kT_Threads = NW * 64;
    constexpr int kT_Halo = (NW + 2) * kPitch;
    constexpr int kT_NJ = (kT_Halo + 7) / 8;
    constexpr int kT_HaloBytes = kT_NJ * 1024;
    constexpr int kT_PerWave = (kT_NJ + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *wlds = smem;                       // weights, 9 x 4096 B
    unsigned char *halo = wlds + kWBytes;             // two halo buffers
    float *hw = reinterpret_cast<float *>(halo + 2 * kT_HaloBytes);  // classifier: [32 channels][4], then bias [4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kgrp = lane >> 5;

    for (int i = tid; i < kWBytes / 16; i += kT_Threads)
        reinterpret_cast<uint4 *>(wlds)[i] = reinterpret_cast<const uint4 *>(A.w)[i];
    // classifier with the BatchNorm affine and the weight scale folded in (winv = 2^-k > 0):
    //   logit_c = head_b[c] + sum_ch head_w[ch][c] (scale[ch] relu(acc winv + bias[ch]) + shift[ch])
    //           = cst[c] + sum_ch hw[ch][c] max(acc + pbias[ch], 0),   hw = head_w scale winv,  pbias = bias / winv,
    //             cst = head_b + sum_ch head_w shift
    // -> per channel one add, one max, four fma and five parameters (were: two fma, one max, four fma and seven)
    float *prm = hw + 32 * 4 + 4;  // pbias [32]
    for (int i = tid; i < 32 * 4 + 4 + 32; i += kT_Threads) {
        const int NC = A.nc;
        if (i < 128) {
            const int ch = i >> 2, c = i & 3;
            hw[i] = c < NC ? A.head_w[ch * NC + c] * A.scale[ch] * A.winv : 0.f;
        } else if (i < 132) {
            const int c = i - 128;
            float acc0 = c < NC ? A.head_b[c] : 0.f;
            if (c < NC)
                for (int ch = 0; ch < 32; ++ch) acc0 = fmaf(A.head_w[ch * NC + c], A.shift[ch], acc0);
            hw[i] = acc0;
        } else {
            prm[i - 132] = A.bias[i - 132] / A.winv;
        }
    }
    const int woff = l31 * 64 + ((kgrp ^ ((l31 >> 2) & 3)) * 16);
    const int hp0 = (wave + 1) * kPitch + l31 + 1;
    int boff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) boff[t] = halo_off(hp0 + (t / 3 - 1) * kPitch + (t % 3 - 1), 0, kgrp);

    // this lane's share of the halo requests: instruction j = wave + 8 i moves pixels 8 j .. 8 j + 7, lane = (pixel, piece);
    // the swizzle of the LDS image lives on the SOURCE address (the DMA writes lane * 16 linearly)
    // per piece, once per kernel: halo row / column, and the byte offset of the piece's 16 source bytes relative to the tile's
    // first pixel (the tile origin is wave-uniform: a request costs two 64-bit adds and a border test, not a chain of 64-bit mads)
    // per piece, once per kernel: halo row / column, and the byte offset of the piece's 16 source bytes relative to the tile's
    // halo origin = pixel (y0 - 1, x0 - 1) (the buffer descriptor's base, wave-uniform)
    int d_yx[kT_PerWave];
    unsigned d_off[kT_PerWave];
#pragma unroll
    for (int i = 0; i < kT_PerWave; ++i) {
        const int j = wave + NW * i;
        const int hp = j * 8 + (lane >> 3), p8 = lane & 7;
        const int plane = (p8 >> 2) ^ ((hp >> 1) & 1), q = (p8 & 3) ^ ((hp >> 2) & 3);
        const int hy = hp / kPitch, hx = hp - hy * kPitch;
        d_yx[i] = (j < kT_NJ && hp < kT_Halo) ? (hy | (hx << 8)) : -1;
        d_off[i] = (unsigned)(((long long)hy * A.W + hx) * A.ldx * 4 + plane * 64 + q * 16);
    }
    const unsigned halo_lds = lds_off(halo);
    const unsigned char *xbase = static_cast<const unsigned char *>(A.x);

    const int tiles_x = A.W / 32, tiles_y = (A.H + NW - 1) / NW;  // (the last row of tiles may reach beyond the slice)
    const int per_img = tiles_x * tiles_y;
    const int ntiles = A.B * per_img;
    auto tile_origin = [&](int t, int &b, int &y0, int &x0) {
        b = t / per_img;
        const int r = t - b * per_img;
        const int ty = r / tiles_x;
        y0 = ty * NW;
        x0 = (r - ty * tiles_x) * 32;
    };
    // source address of this wave's request i for a tile (zero line outside the slice), and the request itself
    auto request_halo = [&](int t, int buf) {
        int b, y0, x0;
        tile_origin(t, b, y0, x0);  // (once per tile: two integer divisions)
        // descriptor: base = halo pixel (0, 0) of the tile (it may lie before the tensor: those pieces are out of the slice and
        // get the out-of-range offset), stride 0, 1 GB of records, raw 32-bit format
        const unsigned long long base =
            (unsigned long long)xbase + (unsigned long long)(((((long long)b * A.H + y0 - 1) * A.W + x0 - 1) * A.ldx + A.xoff) * 4);
        i32x4 rsrc;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffull));
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xffffull));
        rsrc[2] = 0x40000000;
        rsrc[3] = 0x00020000;
#pragma unroll
        for (int i = 0; i < kT_PerWave; ++i) {
            const int j = wave + NW * i;
            if (j < kT_NJ) {  // (wave-uniform)
                const int yy = y0 - 1 + (d_yx[i] & 0xFF), xx = x0 - 1 + ((d_yx[i] >> 8) & 0xFF);
                const bool ok = d_yx[i] >= 0 && (unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W;
                dma16_buf(ok ? d_off[i] : 0xFFFFFFF0u, rsrc, halo_lds + (unsigned)(buf * kT_HaloBytes + j * 1024));
            }
        }
    };

    // XCD x (= blockIdx % 8: its own L2) walks a CONTIGUOUS eighth of the tiles, its CUs side by side in it: neighbouring tiles share
    // halo rows / columns through that L2 instead of fetching them from HBM once per XCD
    int t_first = blockIdx.x, t_stride = gridDim.x, t_end = ntiles;
    if ((gridDim.x & 7) == 0) {
        const int per = (ntiles + 7) / 8, xcd = blockIdx.x & 7;
        t_first = xcd * per + (blockIdx.x >> 3);
        t_stride = gridDim.x >> 3;
        t_end = (xcd + 1) * per < ntiles ? (xcd + 1) * per : ntiles;
    }
    int tile = t_first;
    if (tile >= t_end) return;
    int t_b, t_y0, t_x0;
    tile_origin(tile, t_b, t_y0, t_x0);
    request_halo(tile, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int buf = 0;

#ifdef QMRI_S3_EXPERIMENTS
    unsigned long long tmark = __builtin_amdgcn_s_memtime();
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    while (true) {
        const int next = tile + t_stride;
        // the next tile's halo: requested in front of the MFMA loop.  (From INSIDE the loop, one request every three half-steps,
        // measured 1.55 instead of 1.40 ms: the request statement is a memory barrier for hipcc and cuts the LDS read pipeline.)
        const bool more = next < t_end;
        if (more) request_halo(next, buf ^ 1);
        const unsigned char *hb = halo + buf * kT_HaloBytes;

        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        struct Frag {
            f16x8 wh, wl, xh, xl;
        };
        auto load_frag = [&](Frag &f, int t, int kk) {
            const unsigned char *wt = wlds + t * 4096 + (woff ^ (kk * 32));
            const int o = boff[t] ^ (kk * 32);
            f.wh = *reinterpret_cast<const f16x8 *>(wt);
            f.xh = *reinterpret_cast<const f16x8 *>(hb + o);
            f.wl = *reinterpret_cast<const f16x8 *>(wt + 2048);
            f.xl = *reinterpret_cast<const f16x8 *>(hb + (o ^ 64));
        };
        auto mma = [&](const Frag &f) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh, f.xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wl, f.xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh, f.xl, acc, 0, 0, 0);
        };
        Frag f[3];
        load_frag(f[0], 0, 0);
        load_frag(f[1], 0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 18; ++h) {
            if (h + 2 < 18) load_frag(f[(h + 2) % 3], (h + 2) / 2, (h + 2) % 2);
            mma(f[h % 3]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }

        ENC0_T(1)
        // ---- bias, ReLU, BatchNorm; classifier on this lane's 16 channels; the partner lane adds the other 16 ----
        float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = 8 * q + 4 * kgrp;
            const float4 b4 = *reinterpret_cast<const float4 *>(prm + c0);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = fmaxf(acc[4 * q + r] + bb[r], 0.f);
                const float4 w4 = *reinterpret_cast<const float4 *>(hw + (c0 + r) * 4);
                z[0] = fmaf(v, w4.x, z[0]);
                z[1] = fmaf(v, w4.y, z[1]);
                z[2] = fmaf(v, w4.z, z[2]);
                z[3] = fmaf(v, w4.w, z[3]);
            }
        }
        // both halves of the wave need the sum over all 32 channels: v_permlane32_swap of z with itself leaves [lower | lower] and
        // [upper | upper] in its two results (one instruction, no LDS crossbar trip like ds_bpermute)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(z[c]), __float_as_uint(z[c]), false, false);
            z[c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        if (kgrp == 0 && t_y0 + wave < A.H) {
            const long long pix = ((long long)t_b * A.H + t_y0 + wave) * A.W + t_x0 + l31;
            const int NC = A.nc;
            float zz[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) zz[c] = z[c] + hw[128 + c];
            if (NC == 4) {
                if (A.logits) *reinterpret_cast<float4 *>(A.logits + pix * 4) = make_float4(zz[0], zz[1], zz[2], zz[3]);
                if (A.mask)
                    *reinterpret_cast<unsigned *>(A.mask + pix * 4) = (zz[0] > 0.f ? 1u : 0u) | (zz[1] > 0.f ? 0x100u : 0u) |
                                                                      (zz[2] > 0.f ? 0x10000u : 0u) | (zz[3] > 0.f ? 0x1000000u : 0u);
            } else {
                for (int c = 0; c < NC; ++c) {
                    if (A.logits) A.logits[pix * NC + c] = zz[c];
                    if (A.mask) A.mask[pix * NC + c] = zz[c] > 0.f ? 1 : 0;
                }
            }
        }
        ENC0_T(2)
        if (next >= t_end) break;
        const bool stored_row = t_y0 + wave < A.H;  // (wave-uniform)
        tile = next;
        tile_origin(tile, t_b, t_y0, t_x0);
        buf ^= 1;
        // the next halo has landed (this wave's part; then everyone's) and every wave is done with the old buffer.  The wait is
        // COUNTED: the logits / mask stores issued after the requests stay in flight (vmcnt(0) would expose their latency per tile)
        // (round 5: a wave whose row lies below the slice -- last tile row of a height that is no multiple of NW -- issued NO stores; the
        //  same count would then leave its last halo pieces unwaited)
        if (A.nc == 4 && A.logits && A.mask && stored_row)
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (A.nc == 4 && (A.logits || A.mask) && stored_row)
            asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        ENC0_T(3)
#ifdef QMRI_S3_EXPERIMENTS
        tacc[6] += 1;
#endif
    }
#ifdef QMRI_S3_EXPERIMENTS
    if (tid == 0)
        for (int i = 0; i < 7; ++i) atomicAdd(&enc0_tstat[i], tacc[i]);  // (shared with enc0_kernel: read and reset between layers)
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---------------------------------------------------------------------------------------------------------------------
// mid0_kernel -- Conv2D(64 -> 32, 3x3) + ReLU on the 384^2-class level (the first convolution after the top concatenation,
// oaiunet2d.py:266-276), same skeleton with TWO input chunks: 73 KB of LDS-resident weights + one halo buffer per chunk
// (2 x 43 KB) fill the CU's LDS, so the buffers rotate by chunk, not by tile: chunk 0 of tile t + 1 is requested in front of
// the MFMA loop of chunk 1 of tile t, chunk 1 of tile t + 1 in front of the loop of its chunk 0 -- after the epilogue of tile t, which
// stages the output tile in the chunk-1 buffer.  Waits are counted (the 4 output stores of a wave stay in flight).
constexpr int kM_Chunks = 2;

__global__ __launch_bounds__(kO_Threads, 2) void mid0_kernel(const Mid0Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *wlds = smem;                                  // [chunk][tap] x 4096 B
    unsigned char *halo = wlds + kM_Chunks * kWBytes;            // buffer c = chunk c of the current (or next) tile
    float *prm = reinterpret_cast<float *>(halo + kM_Chunks * kO_HaloBytes);  // bias [32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kgrp = lane >> 5;
    float amax = 0.f;  // Mid0Args::sat

    for (int i = tid; i < kM_Chunks * kWBytes / 16; i += kO_Threads)
        reinterpret_cast<uint4 *>(wlds)[i] = reinterpret_cast<const uint4 *>(A.w)[i];
    float pb[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) pb[e] = A.bias[(e & 3) + 8 * (e >> 2) + 4 * kgrp];
    const int woff = l31 * 64 + ((kgrp ^ ((l31 >> 2) & 3)) * 16);
    const int hp0 = (wave + 1) * kPitch + l31 + 1;
    int boff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) boff[t] = halo_off(hp0 + (t / 3 - 1) * kPitch + (t % 3 - 1), 0, kgrp);

    int d_yx[kO_PerWave];
    unsigned d_off[kO_PerWave];  // (see out0_kernel)
#pragma unroll
    for (int i = 0; i < kO_PerWave; ++i) {
        const int j = wave + kO_Waves * i;
        const int hp = j * 8 + (lane >> 3), p8 = lane & 7;
        const int plane = (p8 >> 2) ^ ((hp >> 1) & 1), q = (p8 & 3) ^ ((hp >> 2) & 3);
        const int hy = hp / kPitch, hx = hp - hy * kPitch;
        d_yx[i] = (j < kO_NJ && hp < kO_Halo) ? (hy | (hx << 8)) : -1;
        d_off[i] = (unsigned)(((long long)hy * A.W + hx) * A.ldx * 4 + plane * 64 + q * 16);
    }
    const unsigned halo_lds = lds_off(halo);
    const unsigned char *xbase = static_cast<const unsigned char *>(A.x);

    const int tiles_x = A.W / 32, tiles_y = A.H / kO_Waves;
    const int per_img = tiles_x * tiles_y;
    const int ntiles = A.B * per_img;
    auto tile_origin = [&](int t, int &b, int &y0, int &x0) {
        b = t / per_img;
        const int r = t - b * per_img;
        const int ty = r / tiles_x;
        y0 = ty * kO_Waves;
        x0 = (r - ty * tiles_x) * 32;
    };
    // buffer descriptor of (tile, chunk): base = halo pixel (0, 0) of the tile, channel offset of the chunk; per-lane offsets of the
    // pieces of a tile (out of range = outside the slice = zeros)
    auto chunk_rsrc = [&](int b, int y0, int x0, int chunk) -> i32x4 {
        const unsigned long long base = (unsigned long long)xbase +
                                        (unsigned long long)(((((long long)b * A.H + y0 - 1) * A.W + x0 - 1) * A.ldx + A.xoff + chunk * 32) * 4);
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffull));
        r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xffffull));
        r[2] = 0x40000000;
        r[3] = 0x00020000;
        return r;
    };
    auto piece_off = [&](int y0, int x0, int i) -> unsigned {
        const int yy = y0 - 1 + (d_yx[i] & 0xFF), xx = x0 - 1 + ((d_yx[i] >> 8) & 0xFF);
        const bool ok = d_yx[i] >= 0 && (unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W;
        return ok ? d_off[i] : 0xFFFFFFF0u;
    };
    auto issue = [&](unsigned voff, i32x4 rsrc, int i, int chunk) {  // -> buffer `chunk`
        const int j = wave + kO_Waves * i;
        if (j < kO_NJ) dma16_buf(voff, rsrc, halo_lds + (unsigned)(chunk * kO_HaloBytes + j * 1024));
    };

    // XCD x (= blockIdx % 8: its own L2) walks a CONTIGUOUS eighth of the tiles, its CUs side by side in it: neighbouring tiles share
    // halo rows / columns through that L2 instead of fetching them from HBM once per XCD
    int t_first = blockIdx.x, t_stride = gridDim.x, t_end = ntiles;
    if ((gridDim.x & 7) == 0) {
        const int per = (ntiles + 7) / 8, xcd = blockIdx.x & 7;
        t_first = xcd * per + (blockIdx.x >> 3);
        t_stride = gridDim.x >> 3;
        t_end = (xcd + 1) * per < ntiles ? (xcd + 1) * per : ntiles;
    }
    int tile = t_first;
    if (tile >= t_end) return;
    int t_b, t_y0, t_x0;
    tile_origin(tile, t_b, t_y0, t_x0);
#pragma unroll
    for (int i = 0; i < kO_PerWave; ++i) issue(piece_off(t_y0, t_x0, i), chunk_rsrc(t_b, t_y0, t_x0, 0), i, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (first tile: chunk 0 before anything else)

    while (true) {
        const int next = tile + t_stride;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        struct Frag {
            f16x8 wh, wl, xh, xl;
        };
        auto mma = [&](const Frag &f) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh, f.xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wl, f.xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh, f.xl, acc, 0, 0, 0);
        };
        // the MFMA loop of chunk c carries the requests of the buffer that is free meanwhile: chunk 1 of THIS tile during chunk 0
        // (buffer 1 was the staging area of the previous tile's epilogue), chunk 0 of the NEXT tile during chunk 1
        auto chunk_mfma = [&](int c, const unsigned (&voff)[kO_PerWave], i32x4 rsrc, bool req) {
            const unsigned char *hb = halo + c * kO_HaloBytes;
            const unsigned char *wb = wlds + c * kWBytes;
            auto load_frag = [&](Frag &f, int t, int kk) {
                const unsigned char *wt = wb + t * 4096 + (woff ^ (kk * 32));
                const int o = boff[t] ^ (kk * 32);
                f.wh = *reinterpret_cast<const f16x8 *>(wt);
                f.xh = *reinterpret_cast<const f16x8 *>(hb + o);
                f.wl = *reinterpret_cast<const f16x8 *>(wt + 2048);
                f.xl = *reinterpret_cast<const f16x8 *>(hb + (o ^ 64));
            };
            if (req) {  // (in front of the loop: a request inside it is a memory barrier for hipcc and cuts the LDS read pipeline)
#pragma unroll
                for (int i = 0; i < kO_PerWave; ++i) issue(voff[i], rsrc, i, c ^ 1);
            }
            Frag f[3];
            load_frag(f[0], 0, 0);
            load_frag(f[1], 0, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 18; ++h) {
                if (h + 2 < 18) load_frag(f[(h + 2) % 3], (h + 2) / 2, (h + 2) % 2);
                mma(f[h % 3]);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        };
        unsigned v1[kO_PerWave], v0[kO_PerWave];
#pragma unroll
        for (int i = 0; i < kO_PerWave; ++i) v1[i] = piece_off(t_y0, t_x0, i);
        chunk_mfma(0, v1, chunk_rsrc(t_b, t_y0, t_x0, 1), true);
        // chunk 1 of this tile has landed (nothing newer is in this wave's queue); everyone is done with buffer 0
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const bool more = next < t_end;
        int n_b = t_b, n_y0 = t_y0, n_x0 = t_x0;
        if (more) tile_origin(next, n_b, n_y0, n_x0);
#pragma unroll
        for (int i = 0; i < kO_PerWave; ++i) v0[i] = piece_off(n_y0, n_x0, i);
        chunk_mfma(1, v0, chunk_rsrc(n_b, n_y0, n_x0, 0), more);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone is done with buffer 1: it becomes the staging area

        // ---- bias, ReLU, split; [pixel][hi | lo] image in this wave's window of buffer 1; 128-byte pixel-chunk stores ----
        unsigned char *win = halo + kO_HaloBytes + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(fmaf(acc[4 * q + i], A.winv, pb[4 * q + i]), 0.f);
            unsigned h0, l0, h1, l1;
            split2m(v[0], v[1], h0, l0, amax);
            split2m(v[2], v[3], h1, l1, amax);
            *reinterpret_cast<uint2 *>(win + stage_off(l31, q) + 8 * kgrp) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(win + stage_off(l31, 4 + q) + 8 * kgrp) = make_uint2(l0, l1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            const long long row = ((long long)t_b * A.H + t_y0 + wave) * A.W + t_x0;
            unsigned char *ybase = static_cast<unsigned char *>(A.y) + (row * A.ldy + A.yoff) * 4;
            const int px0 = lane >> 3, pos = lane & 7;
            const uint4 v0 = *reinterpret_cast<const uint4 *>(win + (px0) * 128 + pos * 16);
            const uint4 v1 = *reinterpret_cast<const uint4 *>(win + (px0 + 8) * 128 + pos * 16);
            const uint4 v2 = *reinterpret_cast<const uint4 *>(win + (px0 + 16) * 128 + pos * 16);
            const uint4 v3 = *reinterpret_cast<const uint4 *>(win + (px0 + 24) * 128 + pos * 16);
            // (px >> 1) & 7 of px = px0 + 8 t: ((px0 >> 1) + 4 t) & 7
            *reinterpret_cast<uint4 *>(ybase + (long long)(px0) * A.ldy * 4 + ((pos ^ ((px0 >> 1) & 7)) * 16)) = v0;
            *reinterpret_cast<uint4 *>(ybase + (long long)(px0 + 8) * A.ldy * 4 + ((pos ^ (((px0 >> 1) + 4) & 7)) * 16)) = v1;
            *reinterpret_cast<uint4 *>(ybase + (long long)(px0 + 16) * A.ldy * 4 + ((pos ^ ((px0 >> 1) & 7)) * 16)) = v2;
            *reinterpret_cast<uint4 *>(ybase + (long long)(px0 + 24) * A.ldy * 4 + ((pos ^ (((px0 >> 1) + 4) & 7)) * 16)) = v3;
        }
        if (next >= t_end) break;
        tile = next;
        t_b = n_b;
        t_y0 = n_y0;
        t_x0 = n_x0;
        // every wave has read its staging window back (buffer 1 may be requested into again), and chunk 0 of the new tile has
        // landed: in this wave's queue it is followed only by the 4 output stores, which stay in flight
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (A.sat && amax > 65504.f) *A.sat = 1;
}


}  // namespace

// (one patch row more than is used: the partly empty last halo group reads a row beyond it that it then discards)
// 36864 (weights) + 2 x 43520 (one halo image per group) + 2 x 1872 (one input patch per group) = 127648 B: ONE 8-wave block per CU
size_t enc0_lds_bytes() { return (size_t)kWBytes + 2 * (size_t)kHaloBytes + 2 * (size_t)kPatchFloats * 4; }

bool enc0_supported(const Enc0Args &k) { return k.H % kRows == 0 && k.W % 32 == 0 && k.B > 0; }

hipError_t enc0_launch(const Enc0Args &k, int num_cu, hipStream_t stream) {
    if (!enc0_supported(k)) return hipErrorInvalidValue;
    const size_t lds = enc0_lds_bytes();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(enc0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const long long ntiles = (long long)k.B * (k.H / kRows) * (k.W / 32);
    const int grid = ntiles < num_cu ? (int)ntiles : num_cu;
    (void)hipGetLastError();
    hipLaunchKernelGGL(enc0_kernel, dim3((unsigned)grid), dim3(kThreads), lds, stream, k);
    return hipGetLastError();
}

constexpr int kOut0Waves = 12;
static size_t out0_lds_bytes_n(int nw) { return (size_t)kWBytes + 2 * (size_t)(((nw + 2) * kPitch + 7) / 8) * 1024 + (32 * 4 + 4 + 32) * 4; }
size_t out0_lds_bytes() { return out0_lds_bytes_n(kOut0Waves); }

bool out0_supported(const Out0Args &k) { return k.W % 32 == 0 && k.B > 0 && k.H > 0 && k.nc >= 1 && k.nc <= 4; }

// (round 5 tried the same layer with the weights in registers -- out0r_kernel, 8 waves of two rows, 16-row tiles, a third of the LDS
//  bytes per pixel -- in four forms: 1.45-1.52 ms against this kernel's 1.27-1.31, profiles/r05_out0r_ab.txt; removed again)
hipError_t out0_launch(const Out0Args &k, int num_cu, hipStream_t stream) {
    if (!out0_supported(k)) return hipErrorInvalidValue;
    const size_t lds = out0_lds_bytes();
    auto fn = out0_kernel<kOut0Waves>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const long long ntiles = (long long)k.B * ((k.H + kOut0Waves - 1) / kOut0Waves) * (k.W / 32);
    const int grid = ntiles < num_cu ? (int)ntiles : num_cu;
    (void)hipGetLastError();
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kOut0Waves * 64), lds, stream, k);
    return hipGetLastError();
}

size_t mid0_lds_bytes() { return (size_t)kM_Chunks * kWBytes + (size_t)kM_Chunks * kO_HaloBytes + 32 * 4; }

bool mid0_supported(const Mid0Args &k) { return k.H % kO_Waves == 0 && k.W % 32 == 0 && k.B > 0; }

// (round 6 built the same layer with FOUR half-chunk halo buffers -- one multiplied, three in flight instead of one, the epilogue staged
//  plane by plane through 2 KB windows that are the wave's own next request destinations -- on the theory that the layer is bound by
//  bytes in flight: correct (97 GPU tests), 2 490-2 500 us against 2 315 for this kernel, profiles/r06_mid0q_ab.txt.  Four barriers and four
//  pipeline restarts per tile cost more than the deeper prefetch gains; removed again.)
hipError_t mid0_launch(const Mid0Args &k, int num_cu, hipStream_t stream) {
    if (!mid0_supported(k)) return hipErrorInvalidValue;
    const size_t lds = mid0_lds_bytes();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mid0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const long long ntiles = (long long)k.B * (k.H / kO_Waves) * (k.W / 32);
    const int grid = ntiles < num_cu ? (int)ntiles : num_cu;
    (void)hipGetLastError();
    hipLaunchKernelGGL(mid0_kernel, dim3((unsigned)grid), dim3(kO_Threads), lds, stream, k);
    return hipGetLastError();
}

}  // namespace qmri

#ifdef QMRI_ENC0_TIMELINE
extern "C" int qmri_debug_enc0_timeline(unsigned long long *out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(qmri::g_enc0_tl), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

#ifdef QMRI_S3_EXPERIMENTS
extern "C" int qmri_enc0_debug_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(qmri::enc0_tstat), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(qmri::enc0_tstat), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif