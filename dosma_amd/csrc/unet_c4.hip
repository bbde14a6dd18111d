// unet_c4.hip -- the parity-mode ("fp16x3") 3x3 convolution for layers with >= 128 output channels, round 4:
// ONE WAVE PER SIMD, 128 x 128 register tiles.
//
// Same layer as conv_s3_kernel<128> (unet_s3.hip): Conv2D(3x3, SAME) + bias + ReLU (+ the BatchNormalization affine after
// the block's second ReLU, + fused MaxPooling2D) of /root/reference/dosma/models/oaiunet2d.py:213-226, 266-279 on SPLIT
// activations (fp16 hi + lo parts, a product = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate).
//
// Why another kernel.  conv_s3_kernel<128> runs 8 waves (two per SIMD) of 2 x 2 MFMA tiles on a 256-pixel x 128-channel
// block tile and issues 1.10-1.22 PFLOP/s (MfmaUtil 0.68-0.70).  scripts/probes/mfma_lds.hip (profiles/r04_mfma_lds.txt) rebuilds
// that step as a skeleton with its LDS reads, its three LDS-DMA requests and its bookkeeping instructions: 1.34 PF of the 1.64 PF
// the part sustains on MFMAs alone -- and 1.47 PF for FOUR waves (one per SIMD) of 4 x 4 tiles stepping K by 16.  What the bigger
// tile buys is not LDS bandwidth (ds_read_b128 runs at 256 B/clk; the reads are a third of the pipe) but fewer DMA requests,
// reads, barriers and bookkeeping instructions per MFMA: per k-step of 16 a wave issues 48 MFMAs against 16 reads and 2-4
// requests (conv_s3_kernel<128>: 12 MFMAs against 8 reads and 1.5 requests).
//
// Structure
//   * block = 4 waves, __launch_bounds__(256, 1): up to 512 registers per lane -- the 256 accumulators of a wave's
//     128-pixel x 128-channel tile (AccVGPRs) + two sets of operand fragments (128) + addressing.
//   * block tile = 512 output positions x 128 channels: 16 image rows x 32 pixels (halo 18 x 34 = 1.20 x the tile; the 8-row
//     tile of conv_s3_kernel reads 1.33 x), or 512 consecutive positions of the flattened zero-framed image stack for levels
//     narrower than 64 pixels (same construction as unet_s3.hip).  Wave w owns rows 4 w .. 4 w + 3 (positions 128 w ..).
//   * K runs in steps of ONE k-step: (32-channel chunk, 16-channel half, tap).  The halo of a half-chunk is 64 bytes per pixel
//     (16 hi + 16 lo parts), 39 KB per buffer, two buffers -- the 77 KB halo of a whole chunk would not fit twice beside the
//     weight ring.  The source layout stays the network's (128-byte pixel-chunks): a half-chunk request fetches two 32-byte
//     runs per pixel.  LDS image: pixel p at p * 64, piece c = plane * 2 + kgroup at position c ^ ((p >> 2) & 3): every
//     ds_read_b128 lane group is conflict-free at any tap shift (brute-forced: scripts/lds_bank_check.py).
//   * weights: ring of 8 slots of 8 KB ([plane][128 channels][2 x 16 B], piece g of row n at position g ^ ((n >> 3) & 1)),
//     requested FIVE steps ahead; halo pieces of the next half-chunk are requested during taps 0-4.  All by LDS-DMA through
//     buffer_load_dwordx4 ... lds: the address is an SGPR descriptor + a per-lane 32-bit offset + a scalar offset, so a
//     request costs no vector ALU, and a lane outside the image reads beyond num_records = ZERO (the SAME padding; no zero
//     line, no select; scripts/probes/buffer_lds.hip).  Waits are counted (s_waitcnt vmcnt(N): everything issued three or
//     more steps ago has landed), one s_barrier per step of 48 MFMAs per wave.
//   * the operand fragments of step s + 1 are read while step s multiplies (two register sets); the A-operand offsets of
//     all nine tap shifts are per-lane constants (36 registers), the second halo buffer is an immediate offset.
//   * epilogue as in unet_s3.hip (wave-private 4 KB staging window in the finished half-chunk's halo buffer, 16-byte stores
//     of whole 128-byte pixel-chunks, fused 2x2 max-pool on row pairs, saturation tracking).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "qmri_internal.h"

namespace qmri {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __fp16 h16x2;  // what v_cvt_pkrtz_f16_f32 returns
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) f16x8 lds_f16x8;

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kMTile = 512;              // output positions per tile: 16 MFMA row-tiles of 32
constexpr int kRT = 4, kCT = 4;          // 32-pixel row-tiles / 32-channel column tiles per wave
constexpr int kBN = 128;                 // output channels per block
constexpr int kPitch2D = 34;
constexpr int kHalo2D = 18 * kPitch2D;   // 612 halo pixels of the 16 x 32 tile
constexpr int kNJ = 39;                  // DMA instructions (16 pixels x 64 B) per halo buffer: 624 >= 612 (2D), >= 512 + 2 * 50 + 2 (flat)
constexpr int kHBuf = kNJ * 1024;        // bytes per halo buffer
constexpr int kHSlots = 10;              // halo pieces per wave (4 * 10 >= 39; the 40th repeats the wave's first)
constexpr int kRing = 8;                 // weight ring slots
constexpr int kSlot = 8192;              // [2 planes][128 rows][32 B]
constexpr int kAhead = 5;                // weights are requested this many steps ahead
constexpr unsigned kPadOff = 0xFFF00000u;  // a voffset beyond num_records: the lane's 16 bytes arrive as zeros

__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(size_t)(lds_void *)p; }

// LDS-DMA of 16 bytes per lane: LDS destination = M0 + lane * 16, source = descriptor base + voffset (per lane, range-checked
// against num_records: beyond it the lane receives zeros) + soffset (scalar, not range-checked).  Inline asm for the same reason
// as unet_s3.hip's dma16: hipcc neither counts nor drains it; every wait in this file is a hand-counted s_waitcnt vmcnt(N).
__device__ __forceinline__ void dma_buf16(unsigned voff, const i32x4 &rsrc, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ void nt_store16(void *dst, const uint4 &v) {
    u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(dst));
}

// raw buffer descriptor (gfx9 V#): base, stride 0, num_records = kPadOff bytes, DATA_FORMAT = 32 (0x00020000)
__device__ __forceinline__ i32x4 make_rsrc(const void *base) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
    r.z = (int)kPadOff;
    r.w = 0x00020000;
    return r;
}

// decode a flat position of the zero-framed image stack: f = R * P + c, R = b * (H + 1) + y + 1, c = x + 1
__device__ __forceinline__ int flat_to_pix(int f, int P, int H, int W, int B) {
    if (f < P) return -1;
    const int R = f / P, c = f - R * P;
    if (c < 1 || c > W) return -1;
    const int r1 = R - 1;
    const int b = r1 / (H + 1), y = r1 - b * (H + 1);
    if (y >= H || b >= B) return -1;
    return (b * H + y) * W + (c - 1);
}

// An accumulator element for the vector ALU: the tiles live in AccVGPRs (256 of them per lane), VALU instructions cannot
// read those, and left to itself hipcc gives the 16-register tuples an ArchVGPR class as soon as plain code touches their
// elements (343 spilled registers).  The explicit read keeps them where the MFMAs leave them.
__device__ __forceinline__ float acc_get(const float &a) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

struct Frags {
    f16x8 ah[kRT], al[kRT], bh[kCT], bl[kCT];
};

}  // namespace

template <bool FLAT>
__global__ __launch_bounds__(kThreads, 1) void conv_c4_kernel(const ConvS3Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *halo = smem;                                        // [2][kNJ * 16 pixels][64 B swizzled]
    unsigned char *ring = smem + 2 * kHBuf;                            // [kRing][2 planes][128][32 B swizzled]
    int *outpix = reinterpret_cast<int *>(ring + kRing * kSlot);       // [512] output pixel of a tile position, or -1 (FLAT)
    float *prm = reinterpret_cast<float *>(outpix + kMTile);           // bias | scale | shift, [128] each
    unsigned *hofft = reinterpret_cast<unsigned *>(prm + 3 * kBN);     // [kHSlots][256] per-lane halo source offsets (see below)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5;
    const int P = FLAT ? A.P : kPitch2D;
    const int hpix = FLAT ? kMTile + 2 * P + 2 : kHalo2D;
    const int ntiles = A.ntiles;
    const int wsteps = A.steps;  // chunks * 18 steps per work item

    const i32x4 xr = make_rsrc(A.x);
    const i32x4 wr = make_rsrc(A.w_c4);
    const unsigned halo_lds = lds_off(halo), ring_lds = lds_off(ring);

    // ---- work distribution: as conv_s3_kernel (XCD x walks a contiguous eighth of the work items) ----
    int wstride = gridDim.x, w_end = A.nwork;
    int work = blockIdx.x;
    if ((gridDim.x & 7) == 0 && A.nwork >= 64) {
        const int per_xcd = (A.nwork + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        wstride = gridDim.x >> 3;
        work = xcd * per_xcd + (blockIdx.x >> 3);
        w_end = (xcd + 1) * per_xcd < A.nwork ? (xcd + 1) * per_xcd : A.nwork;
    }
    const int nwork = w_end;
    if (work >= nwork) return;

    int t_nb = 0, t_b = 0, t_y0 = 0, t_x0 = 0, t_f0 = 0;
    auto decode_work = [&](int w, int &nb, int &b, int &y0, int &x0, int &f0) {
        nb = w / ntiles;
        const int t = w - nb * ntiles;
        if (FLAT) {
            f0 = A.P + t * kMTile;
            b = y0 = x0 = 0;
        } else {
            const int per_img = A.tiles_y * A.tiles_x;
            b = t / per_img;
            const int r = t - b * per_img;
            const int ty = r / A.tiles_x;
            y0 = ty * 16;
            x0 = (r - ty * A.tiles_x) * 32;
            f0 = 0;
        }
    };
    decode_work(work, t_nb, t_b, t_y0, t_x0, t_f0);

    // ---- halo requests: piece i of this wave is DMA instruction j = wave + 4 i (16 pixels x 64 B) of a halo buffer ----
    // lane -> halo pixel hp = 16 j + (lane >> 2), LDS position pos = lane & 3 holds piece c = pos ^ ((hp >> 2) & 3) = plane * 2 + g:
    // source bytes [plane * 64 + (2 half + g) * 16, + 16) of the pixel's 128-byte chunk record (half and chunk go into soffset)
    // hofft[i][tid]: per-lane source offset of piece i for the tile being REQUESTED (chunk 0, half 0), or kPadOff.  A table in LDS,
    // not ten registers: each value is used once per half-chunk, the register allocator spills such values to scratch, and a
    // scratch reload is a vector-memory load -- hipcc waits for it with vmcnt(0), i.e. for every DMA request in flight.
    auto set_halo_sources = [&](int b, int y0, int x0, int f0) {
#pragma unroll
        for (int i = 0; i < kHSlots; ++i) {
            int j = wave + kWaves * i;
            if (j >= kNJ) j = wave;
            const int hp = j * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((hp >> 2) & 3);
            const unsigned srcb = (unsigned)((c >> 1) * 64 + (c & 1) * 16);
            int pix = -1;
            if (hp < hpix) {
                if (FLAT) {
                    pix = flat_to_pix(f0 - P - 1 + hp, P, A.H, A.W, A.B);
                } else {
                    const int hy = hp / kPitch2D, hx = hp - hy * kPitch2D;
                    const int yy = y0 + hy - 1, xx = x0 + hx - 1;
                    if ((unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W) pix = (b * A.H + yy) * A.W + xx;
                }
            }
            hofft[i * kThreads + tid] = pix >= 0 ? (unsigned)pix * (unsigned)A.ldx * 4u + srcb : kPadOff;
        }
    };
    // the half-chunk being requested: work item, half-chunk index hc = 2 chunk + half, destination buffer = hc & 1
    int req_work = work, req_hc = 0;
    const int nhc = 2 * A.chunks;
    auto halo_soff = [&](int hc) -> unsigned { return (unsigned)(A.xoff * 4 + (hc >> 1) * 128 + (hc & 1) * 32); };
    auto issue_halo = [&](int i, int hc) {
        int j = wave + kWaves * i;
        if (j >= kNJ) j = wave;
        dma_buf16(hofft[i * kThreads + tid], xr, halo_soff(hc), halo_lds + (unsigned)((hc & 1) * kHBuf + j * 1024));
    };

    // ---- weight requests: this wave's two 1 KB pieces (2 wave, 2 wave + 1) of the slot of step (w_nb, w_s) ----
    // The request pointer runs kAhead steps ahead of the computation and therefore crosses into the NEXT work item's weights
    // during the last steps of the current one: w_next = scalar offset of that work item's first slot, refreshed once per
    // work item (past the end of the work: this item's own first slots again -- requested, never read).  No branch per step.
    const unsigned wlane = (unsigned)lane * 16u;
    auto first_slot_of = [&](int nb) -> unsigned { return (unsigned)nb * (unsigned)wsteps * (unsigned)kSlot + (unsigned)wave * 2048u; };
    unsigned w_so = first_slot_of(t_nb);   // scalar offset of the next slot to request (this wave's 2 KB of it)
    unsigned w_next = w_so;
    int w_left = wsteps;                   // slots of the current stretch still to request
    int w_slot = 0;
    auto issue_weights = [&]() {
        const unsigned dst = ring_lds + (unsigned)(w_slot * kSlot) + (unsigned)wave * 2048u;
        dma_buf16(wlane, wr, w_so, dst);
        dma_buf16(wlane, wr, w_so + 1024u, dst + 1024u);
        w_slot = (w_slot + 1) & (kRing - 1);
        --w_left;
        const bool wrap = w_left == 0;
        w_so = wrap ? w_next : w_so + (unsigned)kSlot;
        w_left = wrap ? wsteps : w_left;
    };
    auto refresh_w_next = [&](int cur_work, int cur_nb) {
        const int nw = cur_work + wstride;
        w_next = first_slot_of(nw < nwork ? nw / ntiles : cur_nb);
    };
    refresh_w_next(work, t_nb);

    // ---- per-lane LDS read offsets ----
    // B operand (pixels): row-tile i, tap t, plane 0: halo pixel hp = abase[i] + shift(t) (this lane's pixel of the row-tile),
    // piece khalf at position khalf ^ ((hp >> 2) & 3); plane 1 is ^ 32; the second halo buffer is + kHBuf (an immediate).
    // The four offsets of a step are computed when its operands are read (5 vector instructions each against the step's 48
    // MFMAs): a table of all nine shifts is 36 registers the 256 ArchVGPRs beside the accumulators do not have.
    int abase[kRT];
#pragma unroll
    for (int i = 0; i < kRT; ++i) {
        const int rt = wave * kRT + i;
        abase[i] = FLAT ? rt * 32 + (lane & 31) + P + 1 : (rt + 1) * kPitch2D + (lane & 31) + 1;
    }
    auto a_offsets = [&](unsigned (&ao)[kRT], int t) {
        const int shift = (t / 3 - 1) * P + (t % 3 - 1);
#pragma unroll
        for (int i = 0; i < kRT; ++i) {
            int ab = abase[i];
            asm volatile("" : "+v"(ab));  // (opaque: otherwise all 36 offsets are hoisted out of the loop -- and spilled)
            const int hp = ab + shift;
            ao[i] = halo_lds + (unsigned)(hp * 64 + ((khalf ^ ((hp >> 2) & 3)) * 16));
        }
    };
    // A operand (weights): column tile j, plane p: row n = 32 j + (lane & 31) -> j and p are immediates (1024 j + 4096 p)
    const unsigned boff = ring_lds + (unsigned)((lane & 31) * 32 + ((khalf ^ (((lane & 31) >> 3) & 1)) * 16));

    auto lds16 = [](unsigned off) -> f16x8 { return *reinterpret_cast<const lds_f16x8 *>((size_t)off); };
    auto load_frags = [&](Frags &f, int tap, int buf_imm, unsigned wb) {
        unsigned ao[kRT];
        a_offsets(ao, tap);
#pragma unroll
        for (int i = 0; i < kRT; ++i) {
            f.ah[i] = lds16(ao[i] + (unsigned)buf_imm);
            f.al[i] = lds16((ao[i] ^ 32u) + (unsigned)buf_imm);
        }
#pragma unroll
        for (int j = 0; j < kCT; ++j) {
            f.bh[j] = lds16(wb + (unsigned)(j * 1024));
            f.bl[j] = lds16(wb + (unsigned)(j * 1024 + 4096));
        }
    };

    f32x16 acc[kRT][kCT];
    // `first`: the first k-step of a work item starts the accumulators from the MFMA's constant-zero C operand -- 256 registers
    // are never zeroed by hand (hipcc materialises the zeros in 256 ArchVGPRs first and spills everything else around them)
    auto mma_row = [&](const Frags &f, int i, auto first) {
        constexpr bool kFirst = decltype(first)::value;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < kCT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], kFirst ? zero : acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < kCT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < kCT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: halo (chunk 0, half 0) of the first tile, weights of steps 0 .. kAhead - 1, epilogue parameters ----
    set_halo_sources(t_b, t_y0, t_x0, t_f0);
#pragma unroll
    for (int i = 0; i < kHSlots; ++i) issue_halo(i, 0);
    req_hc = 1;
#pragma unroll
    for (int r = 0; r < kAhead; ++r) issue_weights();
    for (int i = tid; i < 3 * kBN; i += kThreads) {
        const int c = i % kBN, which = i / kBN;
        const int n = t_nb * kBN + c;
        prm[i] = which == 0 ? (A.bias ? A.bias[n] : 0.f) : which == 1 ? (A.scale ? A.scale[n] : 1.f) : (A.shift ? A.shift[n] : 0.f);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    float amax = 0.f;
    int slot = 0;   // ring slot of the step being computed
    int chunk = 0;  // chunk being computed
    Frags f0, f1;
    load_frags(f0, 0, 0, boff);  // operands of the very first step

    // One step = one k-step of 16: (half H, tap T) of the current chunk.  CUR holds its operands; the operands of the next step
    // (half NH, tap NT: buffer NH, next ring slot) are read into NXT while it multiplies.  Requests: the weights of the step
    // kAhead ahead (2 per wave), and in taps 0-4 two halo pieces of the half-chunk after this one.  The counted wait at the end
    // lets the requests of this step and the two before it stay in flight (c(T) = 4, 4, 4, 4, 4, 2, 2, 2, 2 per tap).
#define C4_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory")
#define C4_STEP_(H, T, CUR, NXT, FIRST)                                                                                  \
    {                                                                                                              \
        constexpr int NH_ = (T) == 8 ? 1 - (H) : (H);                                                              \
        constexpr int NT_ = (T) == 8 ? 0 : (T) + 1;                                                                \
        const unsigned wbn_ = boff + (unsigned)(((slot + 1) & (kRing - 1)) * kSlot);                               \
        load_frags(NXT, NT_, NH_ * kHBuf, wbn_);                                                             \
        mma_row(CUR, 0, FIRST);                                                                                        \
        issue_weights();                                                                                           \
        mma_row(CUR, 1, FIRST);                                                                                        \
        if constexpr ((T) <= 4) {                                                                                  \
            issue_halo(2 * (T), req_hc);                                                                           \
            issue_halo(2 * (T) + 1, req_hc);                                                                       \
        }                                                                                                          \
        mma_row(CUR, 2, FIRST);                                                                                        \
        mma_row(CUR, 3, FIRST);                                                                                        \
        /* the step opens with an MFMA; the 16 reads ride two per MFMA behind the first eight */                    \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                     \
        }                                                                                                          \
        constexpr int kN_ = (T) == 0 ? 8 : (T) == 1 ? 10 : (T) <= 4 ? 12 : (T) == 5 ? 10 : (T) == 6 ? 8 : 6;       \
        C4_WAIT(kN_);                                                                                              \
        slot = (slot + 1) & (kRing - 1);                                                                           \
    }
#define C4_STEP(H, T, CUR, NXT) C4_STEP_(H, T, CUR, NXT, std::false_type{})
#define C4_HALF(H)                \
    if (chunk == 0)               \
        C4_STEP_(H, 0, f0, f1, std::true_type{}) \
    else                          \
        C4_STEP(H, 0, f0, f1)     \
    C4_STEP(H, 1, f1, f0)         \
    C4_STEP(H, 2, f0, f1)         \
    C4_STEP(H, 3, f1, f0)         \
    C4_STEP(H, 4, f0, f1)         \
    C4_STEP(H, 5, f1, f0)         \
    C4_STEP(H, 6, f0, f1)         \
    C4_STEP(H, 7, f1, f0)         \
    C4_STEP(H, 8, f0, f1)
    // (nine steps swap the roles of f0 / f1: the second half of a chunk runs with them exchanged)
#define C4_HALF_X(H)              \
    C4_STEP(H, 0, f1, f0)         \
    C4_STEP(H, 1, f0, f1)         \
    C4_STEP(H, 2, f1, f0)         \
    C4_STEP(H, 3, f0, f1)         \
    C4_STEP(H, 4, f1, f0)         \
    C4_STEP(H, 5, f0, f1)         \
    C4_STEP(H, 6, f1, f0)         \
    C4_STEP(H, 7, f0, f1)         \
    C4_STEP(H, 8, f1, f0)

    while (true) {
        // ---- one 32-channel chunk: half 0 (buffer 0), half 1 (buffer 1) ----
        // requests during half 0: (chunk, half 1) -> buffer 1; during half 1: (chunk + 1, half 0) or the next tile's first -> buffer 0
        C4_HALF(0)
        ++req_hc;
        if (req_hc == nhc) {  // the next request opens a new tile: where do its halo pixels come from
            req_hc = 0;
            req_work += wstride;
            int nb_, b_, y0_, x0_, f0_;
            decode_work(req_work < nwork ? req_work : work, nb_, b_, y0_, x0_, f0_);  // (past the end: this tile again, never read)
            set_halo_sources(b_, y0_, x0_, f0_);
        }
        C4_HALF_X(1)
        ++req_hc;
        if (++chunk < A.chunks) continue;

        // ======================= epilogue of this work item =======================
        // staging: buffer 1 (the half-chunk that just finished; buffer 0 already holds the next tile's first), 4 KB per wave
        {
            unsigned char *stage = halo + kHBuf + wave * 4096;
            if (FLAT) {
                for (int i = tid; i < kMTile; i += kThreads) outpix[i] = flat_to_pix(t_f0 + i, P, A.H, A.W, A.B);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            auto out_pixel = [&](int i, int t) -> int {  // output pixel of this lane's store t of row-tile i (lane >> 3 = pixel of 8)
                if (FLAT) return outpix[(wave * kRT + i) * 32 + t * 8 + (lane >> 3)];
                const int yy = t_y0 + wave * kRT + i, xx = t_x0 + t * 8 + (lane >> 3);
                if (yy >= A.H) return -1;
                return (t_b * A.H + yy) * A.W + xx;
            };
            const float winv = A.winv;
            struct Prm4 {
                f32x4 b, s, t;
            };
            auto load_prm = [&](int j, int q) -> Prm4 {
                Prm4 p;
                const float *pp = prm + j * 32 + 8 * q + 4 * khalf;
                p.b = *reinterpret_cast<const f32x4 *>(pp);
                p.s = *reinterpret_cast<const f32x4 *>(pp + kBN);
                p.t = *reinterpret_cast<const f32x4 *>(pp + 2 * kBN);
                return p;
            };
            auto affine = [&](float a, const Prm4 &p, int r) -> float {
                float v = fmaf(a, winv, p.b[r]);
                if (A.relu) v = fmaxf(v, 0.f);
                return fmaf(v, p.s[r], p.t[r]);
            };
            const int n0 = t_nb * kBN;
            const int px_l = lane & 31;
            auto stage_piece = [&](int px, int p8) { return px * 128 + ((p8 ^ ((px >> 1) & 7)) * 16); };
#pragma unroll
            for (int j = 0; j < kCT; ++j) {
                const int cbase = n0 + j * 32;
#pragma unroll
                for (int i = 0; i < kRT; ++i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const Prm4 p = load_prm(j, q);
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = affine(acc_get(acc[i][j][4 * q + r]), p, r);
                        amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
                        amax = fmaxf(fmaxf(amax, fabsf(v[2])), fabsf(v[3]));
                        asm volatile("" : "+v"(amax));  // (pinned here: hipcc otherwise sinks the whole max chain behind the epilogue and keeps all 256 values alive -- in scratch -- until then)
                        const h16x2 h0 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h1 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
                        const h16x2 l0 = __builtin_amdgcn_cvt_pkrtz(v[0] - (float)h0[0], v[1] - (float)h0[1]);
                        const h16x2 l1 = __builtin_amdgcn_cvt_pkrtz(v[2] - (float)h1[0], v[3] - (float)h1[1]);
                        *reinterpret_cast<uint2 *>(stage + stage_piece(px_l, q) + 8 * khalf) =
                            make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                        *reinterpret_cast<uint2 *>(stage + stage_piece(px_l, 4 + q) + 8 * khalf) =
                            make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // four 16-byte stores per lane: 8 lanes = one 128-byte pixel-chunk (pieces permuted by the window swizzle)
                    uint4 v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const uint4 *>(stage + (t * 8 + (lane >> 3)) * 128 + (lane & 7) * 16);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int pix = out_pixel(i, t);
                        const int p8 = (lane & 7) ^ (((t * 8 + (lane >> 3)) >> 1) & 7);
                        if (pix >= 0) {
                            const long long doff = ((long long)pix * A.ldy + A.yoff + cbase) * 4 + p8 * 16;
                            nt_store16(static_cast<unsigned char *>(A.y) + doff, v[t]);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the window is rewritten by the next tile)
                    // one accumulator tile at a time: left alone the scheduler reads many tiles out of the AccVGPRs ahead of their
                    // use (v_accvgpr_read has no memory dependence to hold it back) and spills the main loop's registers
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- fused MaxPooling2D(2x2): row pairs (0, 1) and (2, 3) of this wave -> 16 pooled pixels x 32 channels each ----
                if (!FLAT && A.pool_y) {
#pragma unroll
                    for (int pr = 0; pr < kRT / 2; ++pr) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const Prm4 p = load_prm(j, q);
                            float m[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float vmax = fmaxf(affine(acc_get(acc[2 * pr][j][4 * q + r]), p, r), affine(acc_get(acc[2 * pr + 1][j][4 * q + r]), p, r));
                                const float other = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, vmax), 0xB1, 0xF, 0xF, true));  // quad_perm [1, 0, 3, 2]
                                m[r] = fmaxf(vmax, other);
                            }
                            const h16x2 h0 = __builtin_amdgcn_cvt_pkrtz(m[0], m[1]), h1 = __builtin_amdgcn_cvt_pkrtz(m[2], m[3]);
                            const h16x2 l0 = __builtin_amdgcn_cvt_pkrtz(m[0] - (float)h0[0], m[1] - (float)h0[1]);
                            const h16x2 l1 = __builtin_amdgcn_cvt_pkrtz(m[2] - (float)h1[0], m[3] - (float)h1[1]);
                            if (!(px_l & 1)) {
                                const int pp = (px_l >> 1) + 16 * pr;  // pooled column 0..15 of pair pr -> window pixel 0..31
                                *reinterpret_cast<uint2 *>(stage + stage_piece(pp, q) + 8 * khalf) =
                                    make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                                *reinterpret_cast<uint2 *>(stage + stage_piece(pp, 4 + q) + 8 * khalf) =
                                    make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const int Hp = A.H >> 1, Wp = A.W >> 1;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {  // 32 window pixels x 8 pieces = 256 stores of 16 bytes: four per lane
                        const int id = t * 64 + lane, px = id >> 3, pos = id & 7;
                        const int pr = px >> 4, pc = px & 15;
                        const int yy = (t_y0 >> 1) + wave * (kRT / 2) + pr;
                        if (yy < Hp) {
                            const uint4 v = *reinterpret_cast<const uint4 *>(stage + px * 128 + pos * 16);
                            const int p8 = pos ^ ((px >> 1) & 7);
                            const long long pix = (long long)(t_b * Hp + yy) * Wp + (t_x0 >> 1) + pc;
                            unsigned char *dst = static_cast<unsigned char *>(A.pool_y) + (pix * A.pool_ld + cbase) * 4 + p8 * 16;
                            nt_store16(dst, v);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
        // ---- next work item ----
        work += wstride;
        if (work >= nwork) break;
        const int prev_nb = t_nb;
        decode_work(work, t_nb, t_b, t_y0, t_x0, t_f0);
        refresh_w_next(work, t_nb);
        chunk = 0;
        // every wave is done with its staging window (the next half-chunk's DMA lands there) and with the parameters
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t_nb != prev_nb) {
            for (int i = tid; i < 3 * kBN; i += kThreads) {
                const int c = i % kBN, which = i / kBN;
                const int n = t_nb * kBN + c;
                prm[i] = which == 0 ? (A.bias ? A.bias[n] : 0.f) : which == 1 ? (A.scale ? A.scale[n] : 1.f) : (A.shift ? A.shift[n] : 0.f);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        load_frags(f0, 0, 0, boff + (unsigned)(slot * kSlot));  // operands of the new work item's first step (landed: see C4_STEP)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this block's DMA may land after it has exited
    if (A.sat && amax > 65504.f) *A.sat = 1;
}

static size_t c4_lds_bytes() { return (size_t)2 * kHBuf + (size_t)kRing * kSlot + kMTile * 4 + (size_t)3 * kBN * 4 + (size_t)kHSlots * kThreads * 4; }

// which layers the kernel takes: >= 128 output channels in blocks of 128, 32-channel input chunks, a level it tiles
bool conv_c4_supported(const ConvS3Args &k) {
    if (k.deconv || k.one || k.head_w) return false;
    if (k.Cin % 32 || k.Cout % 128) return false;
    if (k.W % 32 == 0) return !k.pool_y || (!(k.H & 1) && !(k.W & 1));
    return k.W + 2 <= 50 && !k.pool_y;  // flattened zero-framed stack (halo: 512 + 2 (W + 2) + 2 <= 624 pixels)
}

// number of work items (channel blocks x tiles) of a layer on this kernel: the dispatcher's cost model wants it
int conv_c4_work_items(const ConvS3Args &k) {
    const int nb = k.Cout / kBN;
    if (k.W % 32 == 0) return nb * k.B * (k.W / 32) * ((k.H + 15) / 16);
    const int P = k.W + 2;
    const long long span = (long long)k.B * (k.H + 1) * P - P;
    return nb * (int)((span + kMTile - 1) / kMTile);
}

template <bool FLAT>
static hipError_t c4_launch_t(ConvS3Args &k, int num_cu, hipStream_t stream) {
    auto fn = conv_c4_kernel<FLAT>;
    const size_t lds = c4_lds_bytes();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int grid = k.nwork < num_cu ? k.nwork : num_cu;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), lds, stream, k);
    return hipGetLastError();
}

hipError_t conv_c4_launch(const ConvS3Args &k0, int num_cu, hipStream_t stream) {
    ConvS3Args k = k0;
    if (!conv_c4_supported(k) || !k.w_c4) return hipErrorInvalidValue;
    const bool flat = k.W % 32 != 0;
    k.chunks = k.Cin / 32;
    k.steps = k.chunks * 18;
    k.nb = k.Cout / kBN;
    if (flat) {
        k.P = k.W + 2;
        const long long span = (long long)k.B * (k.H + 1) * k.P - k.P;
        k.ntiles = (int)((span + kMTile - 1) / kMTile);
        k.tiles_x = k.tiles_y = 0;
    } else {
        k.P = kPitch2D;
        k.tiles_x = k.W / 32;
        k.tiles_y = (k.H + 15) / 16;
        k.ntiles = k.B * k.tiles_x * k.tiles_y;
    }
    k.nj = kNJ;
    k.nwork = k.nb * k.ntiles;
    // 32-bit source offsets: pixel * ldx * 4 + 128 must stay below the descriptor's num_records
    const unsigned long long span_bytes = (unsigned long long)k.B * k.H * k.W * (unsigned long long)k.ldx * 4ull;
    if (span_bytes >= (unsigned long long)kPadOff) return hipErrorInvalidValue;
    (void)hipGetLastError();
    return flat ? c4_launch_t<true>(k, num_cu, stream) : c4_launch_t<false>(k, num_cu, stream);
}

}  // namespace qmri
