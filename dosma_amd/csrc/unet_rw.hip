// unet_rw.hip -- "register-weights" 3x3 convolution for the high-resolution U-Net levels (gfx950).
//
// Same layer as conv_igemm_kernel (unet_kernels.hip; /root/reference/dosma/models/oaiunet2d.py:213-226, 266-279):
// Conv2D(3x3, SAME) + bias + ReLU (+ the BatchNormalization affine that follows the 2nd ReLU), plain-bf16 mode,
// for the layers with Cout = 32 and Cin in {32, 64} (level 0: 384 x 384 pixels).  Those layers have K = 288 / 576
// and 32 output channels: 70-290 flop per byte of activation traffic, i.e. HBM-bound, and in the general kernel
// they were limited by per-block fixed costs instead (18-36 KB of weights re-staged through LDS for every
// 256-pixel tile -- as many bytes as the activations --, two barriers per 32-channel chunk, LDS reads of the
// weight fragments equal to those of the activations, a 125 us VALU head).  Here:
//   * a block is PERSISTENT over 32 x 8-pixel tiles and keeps ALL weights of the layer in registers as MFMA
//     fragments (9 taps x Cin/16 k-steps x 4 VGPRs = 72 / 144 VGPRs), loaded once per block;
//   * the (34 x 10) x Cin bf16 halo of the NEXT tile is fetched by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers, no ds_write pass) into the second halo buffer while the current tile is computed.  The DMA
//     image is lane-linear, so bank conflicts are removed by an XOR swizzle of the 16-byte chunk index applied
//     to the SOURCE address and to the ds_read address (chunk ^ f(pixel)); with one MFMA row-tile = 32
//     consecutive pixels of one image row the ds_read_b128 lane groups are conflict-free;
//   * MFMAs are issued as D = W x A^T (weights as the row operand): a lane then owns ONE pixel and 16 channels in
//     runs of 4 -> the 1x1 head is 16 FMAs per class straight from the accumulators (+ one lane ^ 32 exchange),
//     and the output tile goes to LDS in 8-byte writes for the 16-byte global stores / the fused 2x2 max-pool;
//   * the K loop has no barrier; a tile costs two barriers.
// Fused producers / consumers as in the general kernel: first layer (Cin = 1) computed into the halo, 2x2
// max-pool of the tile, 1x1 classification head + threshold.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "qmri_internal.h"

namespace qmri {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

__device__ uint4 g_zero16;  // source of the halo pixels outside the image (zero padding)

constexpr int kTW = 32;                       // output tile: rows of 32 pixels (one MFMA row-tile each)
constexpr int kHW = kTW + 2;                  // halo width
constexpr int kWinW = kTW + 4;                // fused first layer: window of the 1-channel image, 36 wide

// NT = 32-channel output tiles (waves along N): Cout = 32 NT.  A block's 4 waves are NT along N x 4 / NT along M,
// each wave owns 2 pixel rows x 32 channels, so the tile is 8 (NT = 1) or 4 (NT = 2) rows of 32 pixels.
template <int CIN, int NT, int RPW = 2>
struct RwCfg {
    static constexpr int COUT = 32 * NT;
    static constexpr int TH = RPW * (4 / NT);            // tile rows: RPW image rows (MFMA row-tiles) per wave
    static constexpr int HH = TH + 2;                    // halo rows
    static constexpr int HPIX = kHW * HH;                // 340 / 204
    static constexpr int WINH = TH + 4;
    static constexpr int OROW = NT == 1 ? 36 : 68;       // bf16 per output-tile row in LDS: 72 / 136 B -> conflict-free ds_write_b64
    static constexpr int ROWB = CIN * 2;                 // bytes per halo pixel
    static constexpr int GROUPS = CIN / 8;               // 16-byte chunks per pixel
    static constexpr int GSHIFT = CIN == 32 ? 2 : 3;     // log2(GROUPS)
    // LDS image: every pixel row is followed by one 16-byte pad slot (80 / 144-byte rows): the 16 lanes of a
    // ds_read_b128 group read the same chunk of 16 pixels of ONE image row, and the odd multiple of 16 bytes makes
    // that conflict-free (brute-forced over all lane groups and offsets; padding once per 256 bytes is 2-way) while
    // the address stays linear in hp (one v_mad per tap; chunk and k-step are immediate offsets).  The DMA image is
    // lane-linear, so the pad slots are DMA lanes too: they read the zero line.
    static constexpr int SPP = GROUPS + 1;               // 16-byte slots per pixel incl. the pad
    static constexpr int LROW = ROWB + 16;               // bytes per pixel row in LDS
    static constexpr int NSLOT = HPIX * SPP;
    static constexpr int NGLDS = (NSLOT + 63) / 64;      // wave-level DMA instructions per tile (1 KiB each)
    static constexpr int PER_WAVE = (NGLDS + 3) / 4;
    static constexpr int HALO_BYTES = NGLDS * 1024;      // lane-linear image incl. the tail of the last instruction
    static constexpr int KSTEPS = CIN / 16;
    static constexpr int NFRAG = 9 * KSTEPS;
    static constexpr int OTILE_BYTES = kTW * TH * OROW * 2;
    static constexpr int MISC_FLOATS = kWinW * WINH + 320 + 132 + 3 * COUT;  // c1 window | c1 w, b | head w, b | bias, scale, shift
};
template <int CIN, int NT, int NBUF, int RPW = 2>
constexpr size_t rw_lds_bytes() {
    using C = RwCfg<CIN, NT, RPW>;
    return (size_t)NBUF * C::HALO_BYTES + C::OTILE_BYTES + C::MISC_FLOATS * 4;
}

typedef __attribute__((address_space(3))) void lds_void;

// Three 16-byte LDS reads the compiler does not see as LDS accesses: while an LDS-DMA is in flight hipcc puts
// s_waitcnt vmcnt(0) in front of every ds_read it cannot prove disjoint from the DMA's destination, which would
// drain the prefetch of the next tile at the start of the epilogue.
template <int STRIDE_BYTES>
__device__ __forceinline__ void lds_read3_f4(const float *p, float4 &a, float4 &b, float4 &c) {
    const unsigned off = (unsigned)(size_t)(lds_void *)p;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:%4\n\tds_read_b128 %2, %3 offset:%5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c)
                 : "v"(off), "n"(STRIDE_BYTES), "n"(2 * STRIDE_BYTES)
                 : "memory");
}
typedef const __attribute__((address_space(1))) void glb_void;

template <int CIN, int NT, bool C1, bool HEAD, int NBUF, int MINW, int RPW = 2>
__global__ __launch_bounds__(256, MINW) void conv_rw_kernel(const ConvKArgs A) {
    using C = RwCfg<CIN, NT, RPW>;
    static_assert(NT == 1 || (!C1 && !HEAD), "the fused first layer and the head belong to 32-channel layers");
    constexpr int kTH = C::TH, kHPix = C::HPIX, kORow = C::OROW, kWinH = C::WINH, COUT = C::COUT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *halo0 = smem;
    __bf16 *otile = reinterpret_cast<__bf16 *>(smem + NBUF * C::HALO_BYTES);
    float *c1img = reinterpret_cast<float *>(smem + NBUF * C::HALO_BYTES + C::OTILE_BYTES);  // [12][36]
    float *c1w = c1img + kWinW * kWinH;  // [9][32] + [32]
    float *hw = c1w + 320;               // head [32][4] (classes padded with zeros) + bias [4]
    float *prm = hw + 132;               // bias | scale | shift, [COUT] each

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % NT, wm = wave / NT;  // this wave's 32-channel tile and row pair

    // ---- all weights of the layer -> registers, once per block: fragment (tap, kk): lane (row = lane & 31,
    // k-group = lane >> 5) holds W[row][(chunk * 9 + tap) * 32 + (kk & 1) * 16 + (lane >> 5) * 8 .. + 8], chunk = kk >> 1
    bf16x8 wfrag[C::NFRAG];
    {
        const __bf16 *wrow = A.w_hi + (long long)(wn * 32 + (lane & 31)) * (9 * CIN) + (lane >> 5) * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < C::KSTEPS; ++kk)
                wfrag[t * C::KSTEPS + kk] =
                    *reinterpret_cast<const bf16x8 *>(wrow + ((kk >> 1) * 9 + t) * 32 + (kk & 1) * 16);
    }
    if (tid < 3 * COUT) {
        const int c = tid % COUT, which = tid / COUT;
        prm[tid] = which == 0 ? (A.bias ? A.bias[c] : 0.f) : which == 1 ? (A.scale ? A.scale[c] : 1.f)
                                                                         : (A.shift ? A.shift[c] : 0.f);
    }
    bf16x8 cw_hi, cw_lo;  // fused first layer: W1 as the MFMA row operand, rows = channels, K slot = tap (9 of 16 used)
    if (C1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 8 * (lane >> 5) + j;
            const float w = tap < 9 ? A.c1_w[tap * 32 + (lane & 31)] : 0.f;
            const __bf16 h = static_cast<__bf16>(w);
            cw_hi[j] = h;
            cw_lo[j] = static_cast<__bf16>(w - static_cast<float>(h));
        }
        for (int i = tid; i < 320; i += 256) c1w[i] = i < 288 ? A.c1_w[i] : A.c1_b[i - 288];
    }
    // 1x1 head on MFMA: logits[c][px] = sum_co HW[c][co] * V[co][px] with V = this lane's own epilogue values as the
    // B operand -- the K slots are permuted the same way on both operands (slot (group, j) of k-step s <-> channel
    // 16 s + 4 group + (j & 3) + 8 (j >> 2), the accumulator layout), so no data moves between lanes.  Head-only
    // layer: the BatchNormalization affine is folded into the head (HW * scale, bias + HW . shift).
    const bool want_tile = A.y || A.pool_y;
    const bool fold = HEAD && !want_tile;
    bf16x8 hf_hi[2], hf_lo[2];
    float hb_eff[4] = {0.f, 0.f, 0.f, 0.f};
    float4 bias_r[4];  // head-only layer: this lane's 16 biases live in registers (no LDS read in its epilogue)
    if (HEAD) {
        __syncthreads();  // prm is complete
        for (int i = tid; i < 132; i += 256) {
            const int c = i & 3, k = i >> 2;
            float v = 0.f;
            if (c < A.head_nc) {
                if (k < 32) {
                    v = A.head_w[k * A.head_nc + c] * (fold ? prm[32 + k] : 1.f);
                } else {
                    v = A.head_b[c];
                    if (fold)
                        for (int q = 0; q < 32; ++q) v += A.head_w[q * A.head_nc + c] * prm[64 + q];
                }
            }
            hw[i] = v;
        }
        __syncthreads();
        const int m = lane & 31, grp = lane >> 5;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int co = 16 * s2 + 4 * grp + (j & 3) + 8 * (j >> 2);
                const float w = m < 4 ? hw[co * 4 + m] : 0.f;
                const __bf16 h = static_cast<__bf16>(w);
                hf_hi[s2][j] = h;
                hf_lo[s2][j] = static_cast<__bf16>(w - static_cast<float>(h));
            }
#pragma unroll
        for (int c = 0; c < 4; ++c) hb_eff[c] = hw[128 + c];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) bias_r[g] = *reinterpret_cast<const float4 *>(prm + wn * 32 + 8 * g + 4 * (lane >> 5));

    // un-shifted halo pixel of this lane in the wave's two row-tiles: tile row 2 * wave + i, column lane & 31
    int hpb[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) hpb[i] = (RPW * wm + i + 1) * kHW + (lane & 31) + 1;

    const int tiles_x = A.W / kTW, tiles_y = A.H / kTH;
    const int tiles_per_img = tiles_x * tiles_y;
    const int ntiles = A.B * tiles_per_img;

    // LDS-DMA of one tile's halo: instruction j (= wave + 4 i) fills bytes [j * 1024, (j + 1) * 1024) of the image;
    // lane l carries the 16-byte slot s = j * 64 + l -> (pixel hp, chunk c) or a pad slot.  What does not depend on the
    // tile is computed once: the source offset relative to the tile's origin, and the halo row / column for the
    // border test.  Pixels outside the image (zero padding), pad slots and the image's tail read the zero line.
    int dma_rel[C::PER_WAVE];
    int dma_yx[C::PER_WAVE];  // hy | hx << 8, or -1: never a pixel
#pragma unroll
    for (int i = 0; i < C::PER_WAVE; ++i) {
        const int s_ = (wave + 4 * i) * 64 + lane;
        const int hp_ = s_ / C::SPP, c_ = s_ - hp_ * C::SPP;
        const int hy_ = hp_ / kHW, hx_ = hp_ - hy_ * kHW;
        const bool px_ = c_ < C::GROUPS && hp_ < kHPix;
        dma_rel[i] = ((hy_ - 1) * A.W + (hx_ - 1)) * (int)A.ldx + A.xoff + c_ * 8;
        dma_yx[i] = px_ ? (hy_ | (hx_ << 8)) : -1;
    }
#define QMRI_RW_ISSUE_HALO(tile_, buf_)                                                                        \
    {                                                                                                          \
        const int b_ = (tile_) / tiles_per_img;                                                                \
        const int tr_ = (tile_) - b_ * tiles_per_img;                                                          \
        const int ty_ = tr_ / tiles_x, tx_ = tr_ - ty_ * tiles_x;                                              \
        const __bf16 *org_ = static_cast<const __bf16 *>(A.x) +                                                \
                             ((long long)b_ * A.H * A.W + (long long)(ty_ * kTH) * A.W + tx_ * kTW) * A.ldx;   \
        _Pragma("unroll") for (int i_ = 0; i_ < C::PER_WAVE; ++i_) {                                           \
            const int j_ = wave + 4 * i_;                                                                      \
            if (j_ < C::NGLDS) {                                                                               \
                const int yy_ = ty_ * kTH + (dma_yx[i_] & 0xFF) - 1, xx_ = tx_ * kTW + ((dma_yx[i_] >> 8) & 0xFF) - 1; \
                const bool ok_ = dma_yx[i_] >= 0 && (unsigned)yy_ < (unsigned)A.H && (unsigned)xx_ < (unsigned)A.W; \
                const void *g_ = ok_ ? static_cast<const void *>(org_ + dma_rel[i_]) : static_cast<const void *>(&g_zero16); \
                __builtin_amdgcn_global_load_lds((glb_void *)g_, (lds_void *)(halo0 + (buf_) * C::HALO_BYTES + j_ * 1024), \
                                                 16, 0, 0);                                                    \
            }                                                                                                  \
        }                                                                                                      \
    }

    // Output of the PREVIOUS tile (deferred stores): the global stores of tile t - 1 are issued at the start of tile
    // t, right before the DMA of tile t + 1, so that both have a whole compute phase to complete before the
    // s_waitcnt vmcnt(0) at the top of tile t + 1 (vmcnt is one in-order counter: a store issued just before the
    // wait would expose its full latency on every tile -- measured: DMA, MFMA and epilogue times simply added up).
    bool have_prev = false;
    int pb_ = 0, py0 = 0, px0 = 0;
    float4 plog[RPW];
    unsigned pmask[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        plog[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        pmask[i] = 0u;
    }

#define QMRI_RW_FLUSH_PREV()                                                                                   \
    if (have_prev) {                                                                           \
        const long long pib_ = (long long)pb_ * A.H * A.W;                                                     \
        if (A.y) {                                                                                             \
            for (int idx = tid; idx < kTW * kTH * (COUT / 8); idx += 256) {                                    \
                const int row = idx / (COUT / 8), c = idx % (COUT / 8);                                        \
                const int yy = py0 + (row >> 5), xx = px0 + (row & 31);                                        \
                const uint2 v0 = *reinterpret_cast<const uint2 *>(otile + row * kORow + c * 8);                \
                const uint2 v1 = *reinterpret_cast<const uint2 *>(otile + row * kORow + c * 8 + 4);            \
                __bf16 *dst = static_cast<__bf16 *>(A.y) + (pib_ + (long long)yy * A.W + xx) * A.ldy + A.yoff + c * 8; \
                *reinterpret_cast<uint4 *>(dst) = make_uint4(v0.x, v0.y, v1.x, v1.y);                          \
            }                                                                                                  \
        }                                                                                                      \
        if (A.pool_y) { /* MaxPooling2D(2x2) of the tile (oaiunet2d.py:234-243): 4 x 16 pooled pixels x 4 chunks */ \
            const int Hp = A.H >> 1, Wp = A.W >> 1;                                                            \
            const int q = tid / (COUT / 8), c = tid % (COUT / 8); /* 16 x TH/2 pooled pixels x COUT/8 = 256 */  \
            const int qy = q >> 4, qx = q & 15;                                                                \
            const __bf16 *p0 = otile + ((2 * qy) * kTW + 2 * qx) * kORow + c * 8;                              \
            bf16x8 o;                                                                                          \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                    \
                const float m = fmaxf(fmaxf(static_cast<float>(p0[k]), static_cast<float>(p0[kORow + k])),     \
                                      fmaxf(static_cast<float>(p0[kTW * kORow + k]),                           \
                                            static_cast<float>(p0[(kTW + 1) * kORow + k])));                   \
                o[k] = static_cast<__bf16>(m);                                                                 \
            }                                                                                                  \
            __bf16 *dst = static_cast<__bf16 *>(A.pool_y) +                                                    \
                          ((long long)(pb_ * Hp + (py0 >> 1) + qy) * Wp + (px0 >> 1) + qx) * A.pool_ld + c * 8; \
            *reinterpret_cast<bf16x8 *>(dst) = o;                                                              \
        }                                                                                                      \
        if (HEAD && lane < 32) { /* rows 0..3 of the head MFMA = classes, column = this lane's pixel */    \
            const int NC = A.head_nc;                                                                          \
            _Pragma("unroll") for (int i = 0; i < RPW; ++i) {                                                  \
                const long long pix = pib_ + (long long)(py0 + RPW * wm + i) * A.W + px0 + lane;               \
                if (NC == 4) {                                                                                 \
                    if (A.logits) *reinterpret_cast<float4 *>(A.logits + pix * 4) = plog[i];                   \
                    if (A.mask) *reinterpret_cast<unsigned *>(A.mask + pix * 4) = pmask[i];                    \
                } else {                                                                                       \
                    const float zz[4] = {plog[i].x, plog[i].y, plog[i].z, plog[i].w};                          \
                    _Pragma("unroll") for (int c = 0; c < 4; ++c) if (c < NC) {                                \
                        if (A.logits) A.logits[pix * NC + c] = zz[c];                                          \
                        if (A.mask) A.mask[pix * NC + c] = (pmask[i] >> (8 * c)) & 1u;                         \
                    }                                                                                          \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    }

    int it = 0;
    if (!C1 && NBUF == 2 && (int)blockIdx.x < ntiles) QMRI_RW_ISSUE_HALO((int)blockIdx.x, 0)

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int y0 = ty * kTH, x0 = tx * kTW;
        const long long img_base = (long long)b * A.H * A.W;
        const int cur = NBUF == 2 ? (it & 1) : 0;
        unsigned char *halo = halo0 + cur * C::HALO_BYTES;

        if constexpr (C1) {
            // ---- fused first layer (oaiunet2d.py:213-219 on the 1-channel image): the halo of THIS convolution's input
            // is relu(conv3x3(image) + bias), 32 channels, computed into LDS; pixels outside the image are the zero
            // padding of this convolution, not conv1 outputs
            const float *img = A.c1_x + img_base;
            float wv[2];  // the 36 x 12 window: 432 values, <= 2 per thread; loads issued before the deferred stores
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = tid + r * 256;
                const int wy = i / kWinW, wx = i - wy * kWinW;
                const int yy = y0 + wy - 2, xx = x0 + wx - 2;
                const bool ok = i < kWinW * kWinH && yy >= 0 && yy < A.H && xx >= 0 && xx < A.W;
                wv[r] = ok ? img[(long long)yy * A.W + xx] : 0.f;
            }
            __syncthreads();  // previous tile: every wave is done with the halo, the window; its output tile is complete
            QMRI_RW_FLUSH_PREV()
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (tid + r * 256 < kWinW * kWinH) c1img[tid + r * 256] = wv[r];
            __syncthreads();
            // conv1 on MFMA: D[co][px] = sum_tap W1[tap][co] * x[px + tap], K = 16 slots (9 taps + zeros); the image
            // is split into bf16 hi + lo parts and so are the weights (hi*hi + hi*lo + lo*hi ~ fp32 products), one
            // MFMA row-tile = 32 consecutive halo pixels, 11 row-tiles per halo shared by the 4 waves
            for (int mt = wave; mt < (kHPix + 31) / 32; mt += 4) {
                const int hp = mt * 32 + (lane & 31);
                const int hpc = hp < kHPix ? hp : kHPix - 1;
                const int hy = hpc / kHW, hx = hpc - hy * kHW;
                const int yy = y0 + hy - 1, xx = x0 + hx - 1;
                const bool inside = hp < kHPix && yy >= 0 && yy < A.H && xx >= 0 && xx < A.W;
                const float *wp = c1img + hy * kWinW + hx;
                const int gsel = lane >> 5;
                bf16x8 xh, xl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // k-group 0 (lanes < 32): taps 0..7;  k-group 1: tap 8, then zero slots
                    const int tap = gsel ? 8 : j;
                    const float v = wp[(tap / 3) * kWinW + tap % 3];
                    const float x = (gsel && j > 0) ? 0.f : v;
                    const __bf16 h = static_cast<__bf16>(x);
                    xh[j] = h;
                    xl[j] = static_cast<__bf16>(x - static_cast<float>(h));
                }
                f32x16 z;
#pragma unroll
                for (int e = 0; e < 16; ++e) z[e] = 0.f;
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cw_hi, xh, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cw_hi, xl, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cw_lo, xh, z, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co0 = 8 * g + 4 * gsel;
                    const float4 bb = *reinterpret_cast<const float4 *>(c1w + 288 + co0);
                    bf16x4 o;
                    o[0] = static_cast<__bf16>(inside ? fmaxf(z[4 * g] + bb.x, 0.f) : 0.f);
                    o[1] = static_cast<__bf16>(inside ? fmaxf(z[4 * g + 1] + bb.y, 0.f) : 0.f);
                    o[2] = static_cast<__bf16>(inside ? fmaxf(z[4 * g + 2] + bb.z, 0.f) : 0.f);
                    o[3] = static_cast<__bf16>(inside ? fmaxf(z[4 * g + 3] + bb.w, 0.f) : 0.f);
                    if (hp < kHPix) *reinterpret_cast<bf16x4 *>(halo + hp * C::LROW + co0 * 2) = o;
                }
            }
            __syncthreads();
        } else {
            if (NBUF == 1) {
                __syncthreads();  // previous tile: every wave is done with the halo; its output tile is complete
                QMRI_RW_FLUSH_PREV()
                QMRI_RW_ISSUE_HALO(tile, 0)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of the current halo have landed
            __syncthreads();                                  // ... and so have everyone else's
            if (NBUF == 2) {
                QMRI_RW_FLUSH_PREV()
                if (tile + (int)gridDim.x < ntiles) QMRI_RW_ISSUE_HALO(tile + (int)gridDim.x, cur ^ 1)
            }
        }

        // ---- 9 taps x Cin/16 k-steps, weights from registers, no barrier.  A fragment of (tap, row-tile): pixel
        // hp = hpb + dy * 34 + dx, chunk kk * 2 + (lane >> 5): the k-step is an immediate offset of 32 bytes
        f32x16 acc[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int code = (int)((A.taps >> (4 * t)) & 0xF);  // (dy+1) | (dx+1) << 2
                const int shift = ((code & 3) - 1) * kHW + ((code >> 2) - 1);
                int ab[RPW];
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    const int hp = hpb[i] + shift;
                    ab[i] = hp * C::LROW + (lane >> 5) * 16;
                }
#pragma unroll
                for (int kk = 0; kk < C::KSTEPS; ++kk) {
#pragma unroll
                    for (int i = 0; i < RPW; ++i) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(halo + ab[i] + kk * 32);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[t * C::KSTEPS + kk], a, acc[i], 0, 0, 0);
                    }
                }
            }
        }

        // ---- epilogue.  D = W x A^T: a lane owns ONE pixel (column lane & 31 of the row-tile) and 16 channels in 4
        // runs of 4 consecutive ones: channel(e) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
        // y = scale * relu(acc + bias) + shift in fp32; the 1x1 head + threshold (oaiunet2d.py:285, 306) from the
        // registers (MFMA); y / pool: 4 x 8-byte LDS writes per row-tile (bf16 x 4 channels).  The global stores are
        // deferred to the next iteration (QMRI_RW_FLUSH_PREV).
        if (want_tile) __syncthreads();  // every wave has flushed the previous output tile
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = (RPW * wm + i) * kTW + (lane & 31);  // pixel of the tile
            bf16x8 vb[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co0 = 8 * g + 4 * (lane >> 5);
                float4 pb = bias_r[g], ps, pt;
                if (!fold) lds_read3_f4<COUT * 4>(prm + wn * 32 + co0, pb, ps, pt);
                float v[4] = {acc[i][4 * g] + pb.x, acc[i][4 * g + 1] + pb.y, acc[i][4 * g + 2] + pb.z,
                              acc[i][4 * g + 3] + pb.w};
                if (A.relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                if (!fold) {
                    v[0] = v[0] * ps.x + pt.x;
                    v[1] = v[1] * ps.y + pt.y;
                    v[2] = v[2] * ps.z + pt.z;
                    v[3] = v[3] * ps.w + pt.w;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) vb[g >> 1][(g & 1) * 4 + q] = static_cast<__bf16>(v[q]);
                if (want_tile) {
                    bf16x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = vb[g >> 1][(g & 1) * 4 + q];
                    *reinterpret_cast<bf16x4 *>(otile + r * kORow + wn * 32 + co0) = o;
                }
            }
            if (HEAD) {
                f32x16 z;
#pragma unroll
                for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf_hi[s2], vb[s2], z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf_lo[s2], vb[s2], z, 0, 0, 0);
                }
                const float z0 = z[0] + hb_eff[0], z1 = z[1] + hb_eff[1], z2 = z[2] + hb_eff[2], z3 = z[3] + hb_eff[3];
                plog[i] = make_float4(z0, z1, z2, z3);
                pmask[i] = (z0 > 0.f ? 1u : 0u) | (z1 > 0.f ? 0x100u : 0u) | (z2 > 0.f ? 0x10000u : 0u) |
                           (z3 > 0.f ? 0x1000000u : 0u);
            }
        }
        have_prev = true;
        pb_ = b;
        py0 = y0;
        px0 = x0;
    }
    if (want_tile) __syncthreads();  // the last output tile is complete
    QMRI_RW_FLUSH_PREV()
#undef QMRI_RW_FLUSH_PREV
#undef QMRI_RW_ISSUE_HALO
}

template <int CIN, int NT, bool C1, bool HEAD, int NBUF, int MINW, int RPW = 2>
hipError_t rw_launch_one(const ConvKArgs &k, int num_cu, hipStream_t stream) {
    auto fn = conv_rw_kernel<CIN, NT, C1, HEAD, NBUF, MINW, RPW>;
    const size_t lds = rw_lds_bytes<CIN, NT, NBUF, RPW>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    const long long ntiles = (long long)k.B * (k.H / RwCfg<CIN, NT, RPW>::TH) * (k.W / kTW);
    long long grid = (long long)num_cu * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, stream, k);
    return hipGetLastError();
}

int env_int(const char *name, int dflt) {
    const char *e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}

}  // namespace

// true if the layer is one this kernel handles (plain-bf16 mode only)
bool conv_rw_supported(const ConvKArgs &k) {
    static const int enabled = env_int("QMRI_CONV_RW", 1);
    if (!enabled) return false;
    if (k.deconv || k.ntaps != 9 || (k.Cout != 32 && k.Cout != 64) || (k.Cin != 32 && k.Cin != 64)) return false;
    if (k.Cout == 64 && (k.c1_x || k.head_w)) return false;
    if (k.H % 8 || k.W % kTW || k.sy != 1 || k.sx != 1 || k.py || k.px || k.Ho != k.H || k.Wo != k.W) return false;
    if (k.c1_x && k.Cin != 32) return false;
    if (!k.c1_x && (k.ldx % 8 || k.xoff % 8)) return false;  // 16-byte DMA pieces
    if (k.head_w && (k.head_nc < 1 || k.head_nc > 4)) return false;
    return true;
}

hipError_t conv_rw_launch(const ConvKArgs &k0, hipStream_t stream) {
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) return e;
        num_cu = prop.multiProcessorCount;
    }
    (void)hipGetLastError();
    static const int nbuf64 = env_int("QMRI_RW_NBUF64", 3);
    const ConvKArgs &k = k0;
    if (k.Cout == 64) {
        if (k.Cin == 64) return rw_launch_one<64, 2, false, false, 1, 2>(k, num_cu, stream);
        return rw_launch_one<32, 2, false, false, 2, 2>(k, num_cu, stream);
    }
    if (k.Cin == 64) {
        if (k.head_w) return nbuf64 == 2 ? rw_launch_one<64, 1, false, true, 2, 1>(k, num_cu, stream)
                                         : rw_launch_one<64, 1, false, true, 1, 2>(k, num_cu, stream);
        if (nbuf64 == 3) return rw_launch_one<64, 1, false, false, 2, 2, 1>(k, num_cu, stream);  // 32 x 4 tiles, double-buffered
        return nbuf64 == 2 ? rw_launch_one<64, 1, false, false, 2, 1>(k, num_cu, stream)
                           : rw_launch_one<64, 1, false, false, 1, 2>(k, num_cu, stream);
    }
    if (k.c1_x) return k.head_w ? rw_launch_one<32, 1, true, true, 1, 2>(k, num_cu, stream)
                                : rw_launch_one<32, 1, true, false, 1, 3>(k, num_cu, stream);
    return k.head_w ? rw_launch_one<32, 1, false, true, 2, 2>(k, num_cu, stream)
                    : rw_launch_one<32, 1, false, false, 2, 2>(k, num_cu, stream);
}

}  // namespace qmri
