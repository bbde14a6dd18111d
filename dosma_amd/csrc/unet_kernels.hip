// unet_kernels.hip -- 2D U-Net inference kernels for gfx950 (MI355X): the layers of
//     /root/reference/dosma/models/oaiunet2d.py:197-289  (IWOAIOAIUnet2D.__load_keras_model__)
// that the reference runs through Keras/TensorFlow (`model.predict`, oaiunet2d.py:305).
//
//   conv_igemm_kernel  K7 + K8 + K10 + K11 of SURVEY.md section 2.3:
//       implicit-GEMM convolution on MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate) over NHWC fp32
//       activations, with a fused epilogue  y = scale[c] * relu(acc + bias[c]) + shift[c]
//       (Conv2D bias + ReLU, and the inference-mode BatchNormalization that FOLLOWS the second ReLU
//       of every block, oaiunet2d.py:228, 281 -- it cannot be folded into the weights because of
//       the ReLU in between and the zero padding after it).
//       A "tap list" generalises it: the 9 taps of a 3x3 SAME convolution, or the 1/2/2/4 taps of one
//       output phase of Conv2DTranspose(3x3, stride 2, SAME) (sub-pixel decomposition,
//       oaiunet2d.py:259-261) whose outputs go to the strided positions (2y+py, 2x+px).
//       Input and output carry a pixel stride and a channel offset, so producers write straight into
//       the halves of the concat buffer: Concatenate (oaiunet2d.py:257-264) costs no kernel.
//       Precision: activations are split on the fly into bf16 hi (+ lo) parts while staging to LDS;
//       SPLIT3 = hi*hi + hi*lo + lo*hi (3 MFMAs, ~fp32 accuracy, meets the 1e-3 logit bar),
//       otherwise plain bf16 (1 MFMA).
//   conv3x3_c1_kernel  first layer (Cin = 1, K = 9: not GEMM shaped) -- fp32 VALU
//   maxpool2_kernel    K9: MaxPooling2D(2x2)
//   head_kernel        K12: Conv2D(n_classes, 1x1) logits (+ mask = logit > 0  <=>  sigmoid > 0.5)
//   whiten kernels     K13: (x - mean) / (std + eps) over the whole volume, fp64 reductions
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "qmri_internal.h"

namespace qmri {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __fp16 h16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kBK = 32;         // channels per K step (one tap x 32 input channels)
constexpr int kLdsRow = 40;     // bf16 elements per LDS row: 32 + 8 pad -> 80 B stride, conflict-free b128 reads
constexpr int kTW = 16;         // output tile width (pixels)

// Tile geometry.  A block owns TH x 16 output pixels x BN output channels and loops over 32-channel
// chunks of the input; per chunk the (TH+2) x 18 input halo is converted to bf16 ONCE and kept in
// LDS, and the taps (9 for a 3x3 convolution) are walked as shifted views of it -- each activation is
// fetched from HBM/L2 ~1.4x instead of 9x and converted once instead of 9 times.
template <int BN, int TH_, bool WALL, int TW = 16, int NW = 4, int WN = 0>
struct TileCfg {
    static constexpr int NT = NW * 64;  // threads per block
    // WALL (small BN): the weights of ALL taps of a chunk are staged at once -> 2 barriers per chunk
    // instead of one per tap (at BN <= 64 a tap is only 2-4 MFMAs per wave, less than a barrier costs)
    static constexpr int TH = TH_;
    static constexpr int BM = TH * TW;
    static constexpr int WAVES_N = WN ? WN : (BN >= 64 ? 2 : 1);  // WN = 1 at BN = 64: 64 x 64 per wave, 1 KB of LDS reads per MFMA instead of 1.5
    static constexpr int WAVES_M = NW / WAVES_N;
    static constexpr int TM = BM / WAVES_M / 32;
    static constexpr int TN = BN / WAVES_N / 32;
    static constexpr int HALO_PIX = (TH + 2) * (TW + 2);
    static constexpr int HALO_PAIRS = (HALO_PIX * 4 + NT - 1) / NT;  // (pixel, 8-channel group) per thread
    static constexpr int B_PAIRS = (BN * 4 + NT - 1) / NT;
    static constexpr int WALL_PAIRS = (9 * BN * 4 + NT - 1) / NT;
};

// Parity mode ("fp16x3"): activations are stored SPLIT -- per pixel and 32-channel chunk 32 fp16 hi parts, then 32 fp16
// lo parts (see ConvS3Args in qmri_internal.h).  8 fp32 values -> their 8 hi (lo) parts as 16 bytes.  hi is rounded
// toward zero (v_cvt_pkrtz: one instruction per pair, never overflows to inf), lo = rtz(v - hi): 21+ significant bits.
__device__ __forceinline__ void split_f16(const float (&v)[8], uint4 &hi, uint4 &lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const h16x2 hh = __builtin_amdgcn_cvt_pkrtz(v[2 * q], v[2 * q + 1]);
        const h16x2 ll = __builtin_amdgcn_cvt_pkrtz(v[2 * q] - (float)hh[0], v[2 * q + 1] - (float)hh[1]);
        h[q] = __builtin_bit_cast(unsigned, hh);
        l[q] = __builtin_bit_cast(unsigned, ll);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

template <int BN, int TH, bool SPLIT3, typename AT, bool DECONV, bool C1 = false, int TW = 16, int NW = 4, int WN = 0>
__global__ __launch_bounds__(NW * 64) void conv_igemm_kernel(const ConvKArgs A) {
    constexpr int kNT = NW * 64;  // NW = 8: a 16 x 16-pixel tile on 8 waves (4 x 2), twice the MFMAs per weight tile and barrier
    // TW: tile width.  16 in general; the deep levels of a 384 x 384 input are 24 and 12 pixels wide, where 16-wide
    // tiles waste 25 % / 44 % of the MFMA rows on padding: TW = 24 (x 8 rows = 6 row-tiles) and TW = 12 (x 16 rows, the
    // whole 12 x 12 image + 4 padding rows = 6 row-tiles) cover them with 0 % / 25 %.  A row-tile is any 32 consecutive
    // pixels of the tile in row-major order (a_row0 is per lane), so nothing else depends on the width.
    constexpr int kTW = TW;
    // DECONV: all four output phases of Conv2DTranspose(3x3, stride 2, SAME) in one pass: the 9 taps are
    // ordered [phase (0,0): 4][phase (0,1): 2][phase (1,0): 2][phase (1,1): 1], each tap accumulates into
    // its phase's accumulator, and the epilogue writes four interleaved output tiles.  The input halo is
    // fetched once instead of four times and the launch count drops 4x.
    constexpr int NPH = DECONV ? 4 : 1;
    constexpr bool ACT_BF16 = sizeof(AT) == 2;  // activations stored as bf16 (plain bf16 mode), or split fp16 hi | lo (SPLIT3;
                                                // AT = float is then only the type of the epilogue's LDS tile)
    static_assert(SPLIT3 != ACT_BF16, "plain mode: bf16 activations; parity mode: split fp16 activations");
    float amax = 0.f;  // parity mode: max |v| of what this thread stores in the split layout (ConvKArgs::sat)
    constexpr bool WALL = BN <= 64 && !SPLIT3;
    using C = TileCfg<BN, TH, WALL, TW, NW, WN>;
    constexpr int NPLANES = SPLIT3 ? 2 : 1;
    constexpr int HALO_BYTES = C::HALO_PIX * kLdsRow * 2;  // one plane of one halo buffer
    constexpr int W_BYTES = BN * kLdsRow * 2;               // one plane of one weight buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *halo_base = smem;                                  // [2][NPLANES][HALO_BYTES]
    unsigned char *w_base = smem + A.halo_bufs * NPLANES * HALO_BYTES;  // [2][NPLANES][W_BYTES]
    // rowpix and the first-layer weights outlive the main loop: they sit behind both the main-loop buffers and the
    // epilogue's output tile (+ head weights), which aliases the start of LDS
    constexpr int EPI_BYTES = C::BM * BN * (int)sizeof(AT) + (BN * 4 + 4) * (int)sizeof(float);
    const int main_bytes = A.halo_bufs * NPLANES * HALO_BYTES + (WALL ? 9 : 2 * NPLANES) * W_BYTES;
    int *rowpix = reinterpret_cast<int *>(smem + (main_bytes > EPI_BYTES ? main_bytes : EPI_BYTES));  // [BM] output pixel or -1
    float *c1w = reinterpret_cast<float *>(rowpix + C::BM);  // fused first layer: [9][32] weights + [32] bias

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / C::WAVES_N;
    const int wn = wave % C::WAVES_N;
    const int n0 = blockIdx.y * BN;

    // block -> (image, tile row, tile column)
    const int tiles_per_img = A.tiles_y * A.tiles_x;
    const int b = blockIdx.x / tiles_per_img;
    const int trem = blockIdx.x - b * tiles_per_img;
    const int ty = trem / A.tiles_x;
    const int tx = trem - ty * A.tiles_x;
    const int y0 = ty * C::TH, x0 = tx * kTW;
    const long long img_base = (long long)b * A.H * A.W;

    for (int r = tid; r < C::BM; r += kNT) {
        const int ly = r / kTW, lx = r % kTW;
        const int yy = y0 + ly, xx = x0 + lx;
        int pix = -1;
        if (yy < A.H && xx < A.W) pix = (b * A.Ho + (yy * A.sy + A.py)) * A.Wo + (xx * A.sx + A.px);
        rowpix[r] = pix;
    }

    // ---- per-thread halo gather coordinates (constant over the K loop) ----
    long long h_src[C::HALO_PAIRS];  // element offset of the pixel's channel 0 (+ 8-channel group), or -1
    int h_dst[C::HALO_PAIRS];        // byte offset inside a halo plane
#pragma unroll
    for (int r = 0; r < C::HALO_PAIRS; ++r) {
        const int idx = tid + r * kNT;
        const int hp = idx >> 2, grp = idx & 3;
        const int hy = hp / (kTW + 2), hx = hp - hy * (kTW + 2);
        const int yy = y0 + hy - 1, xx = x0 + hx - 1;
        const bool ok = hp < C::HALO_PIX && yy >= 0 && yy < A.H && xx >= 0 && xx < A.W;
        h_src[r] = ok ? (img_base + (long long)yy * A.W + xx) * A.ldx + A.xoff + (SPLIT3 ? 0 : grp * 8) : -1;
        h_dst[r] = hp < C::HALO_PIX ? (hp * kLdsRow + grp * 8) * 2 : -1;
    }

    bf16x8 rhb[C::HALO_PAIRS];  // 8 channels = 16 bytes: bf16 values, or the fp16 hi parts (the registers only move bits)
    bf16x8 rhl[C::HALO_PAIRS];  // parity mode: the fp16 lo parts
    bf16x8 rb_hi[WALL ? C::WALL_PAIRS : C::B_PAIRS], rb_lo[C::B_PAIRS];
    const int ntaps = A.ntaps;
    const int K = ntaps * A.Cin;

    // (macros, not lambdas: by-reference captures of the staging arrays made the compiler keep them in
    //  scratch memory; the loads are unconditional from a clamped address + select for the same reason)
#define QMRI_LOAD_HALO(c0_)                                                                        \
    _Pragma("unroll") for (int r = 0; r < C::HALO_PAIRS; ++r) {                                    \
        const bool ok_ = h_src[r] >= 0;                                                            \
        bf16x8 z_;                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) z_[q] = static_cast<__bf16>(0.f);            \
        if constexpr (ACT_BF16) {                                                                  \
            const __bf16 *p_ = static_cast<const __bf16 *>(A.x) + (ok_ ? h_src[r] : 0) + (c0_);    \
            const bf16x8 v_ = *reinterpret_cast<const bf16x8 *>(p_);                               \
            rhb[r] = ok_ ? v_ : z_;                                                                \
        } else { /* split: chunk at ((pixel * ld + off + c0) * 4) bytes: [32 hi][32 lo] fp16 */    \
            const unsigned char *p_ = static_cast<const unsigned char *>(A.x) +                    \
                                      ((ok_ ? h_src[r] : 0) + (c0_)) * 4 + (tid & 3) * 16;         \
            const bf16x8 v_ = *reinterpret_cast<const bf16x8 *>(p_);                               \
            const bf16x8 w_ = *reinterpret_cast<const bf16x8 *>(p_ + 64);                          \
            rhb[r] = ok_ ? v_ : z_;                                                                \
            rhl[r] = ok_ ? w_ : z_;                                                                \
        }                                                                                          \
    }
#define QMRI_STORE_HALO(buf_)                                                                      \
    {                                                                                              \
        unsigned char *base_ = halo_base + (buf_) * NPLANES * HALO_BYTES;                          \
        _Pragma("unroll") for (int r = 0; r < C::HALO_PAIRS; ++r) {                                \
            if (h_dst[r] >= 0) {                                                                   \
                *reinterpret_cast<bf16x8 *>(base_ + h_dst[r]) = rhb[r];                            \
                if (SPLIT3) *reinterpret_cast<bf16x8 *>(base_ + HALO_BYTES + h_dst[r]) = rhl[r];   \
            }                                                                                      \
        }                                                                                          \
    }
    // weights of K step `step` (= chunk * ntaps + tap): [BN][32] slice of W[co][step*32 + c]
#define QMRI_LOAD_W(step_)                                                                         \
    _Pragma("unroll") for (int r = 0; r < C::B_PAIRS; ++r) {                                       \
        const int idx_ = tid + r * kNT;                                                            \
        if (idx_ < BN * 4) {                                                                       \
            const long long off_ =                                                                 \
                (long long)(n0 + (idx_ >> 2)) * K + (long long)(step_) * kBK + (idx_ & 3) * 8;     \
            rb_hi[r] = *reinterpret_cast<const bf16x8 *>(A.w_hi + off_);                           \
            if (SPLIT3) rb_lo[r] = *reinterpret_cast<const bf16x8 *>(A.w_lo + off_);               \
        }                                                                                          \
    }
#define QMRI_STORE_W(buf_)                                                                         \
    {                                                                                              \
        unsigned char *base_ = w_base + (buf_) * NPLANES * W_BYTES;                                \
        _Pragma("unroll") for (int r = 0; r < C::B_PAIRS; ++r) {                                   \
            const int idx_ = tid + r * kNT;                                                        \
            if (idx_ < BN * 4) {                                                                   \
                const int off_ = ((idx_ >> 2) * kLdsRow + (idx_ & 3) * 8) * 2;                     \
                *reinterpret_cast<bf16x8 *>(base_ + off_) = rb_hi[r];                              \
                if (SPLIT3) *reinterpret_cast<bf16x8 *>(base_ + W_BYTES + off_) = rb_lo[r];        \
            }                                                                                      \
        }                                                                                          \
    }

    // WALL: all taps of chunk `ch_`: tap-major [ntaps][BN][32]
#define QMRI_LOAD_WALL(ch_)                                                                        \
    _Pragma("unroll") for (int r = 0; r < C::WALL_PAIRS; ++r) {                                    \
        const int idx_ = tid + r * kNT;                                                            \
        if (idx_ < ntaps * BN * 4) {                                                               \
            const int tap_ = idx_ / (BN * 4), rem_ = idx_ - tap_ * (BN * 4);                       \
            const long long off_ = (long long)(n0 + (rem_ >> 2)) * K +                             \
                                   (long long)((ch_) * ntaps + tap_) * kBK + (rem_ & 3) * 8;       \
            rb_hi[r] = *reinterpret_cast<const bf16x8 *>(A.w_hi + off_);                           \
        }                                                                                          \
    }
#define QMRI_STORE_WALL()                                                                          \
    _Pragma("unroll") for (int r = 0; r < C::WALL_PAIRS; ++r) {                                    \
        const int idx_ = tid + r * kNT;                                                            \
        if (idx_ < ntaps * BN * 4) {                                                               \
            const int off_ = ((idx_ >> 2) * kLdsRow + (idx_ & 3) * 8) * 2;                         \
            *reinterpret_cast<bf16x8 *>(w_base + off_) = rb_hi[r];                                 \
        }                                                                                          \
    }

    f32x16 acc[NPH][C::TM][C::TN];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ph][i][j][e] = 0.f;

    // per-lane halo row of each MFMA row-tile (un-shifted): pixel (ly + 1, lx + 1)
    int a_row0[C::TM];
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
        const int r = (wm * C::TM + i) * 32 + (lane & 31);
        a_row0[i] = ((r / kTW) + 1) * (kTW + 2) + (r % kTW) + 1;
    }

    const int chunks = A.Cin / kBK;
    const int steps = chunks * ntaps;
    constexpr bool C1_CAPABLE = C1 && BN == 32 && !DECONV;  // a separate instantiation carries the fused first layer
    bool did_c1 = false;
    if constexpr (C1_CAPABLE) if (A.c1_x) {
        did_c1 = true;
        // Fused first layer (oaiunet2d.py:213-219 on the 1-channel input): the halo of THIS convolution's
        // input is computed here -- relu(conv3x3(image) + bias), 32 channels -- straight into the staging
        // registers, so the first feature map never goes to HBM.  Pixels outside the image are the zero
        // padding of this convolution, not conv1 outputs.
        for (int i = tid; i < 9 * 32 + 32; i += kNT) c1w[i] = i < 288 ? A.c1_w[i] : A.c1_b[i - 288];
        __syncthreads();
        const int grp = tid & 3;
        const float *img = A.c1_x + img_base;
#pragma unroll
        for (int r = 0; r < C::HALO_PAIRS; ++r) {
            const int hp = (tid + r * kNT) >> 2;
            const int hy = hp / (kTW + 2), hx = hp - hy * (kTW + 2);
            const int yy = y0 + hy - 1, xx = x0 + hx - 1;
            const bool inside = hp < C::HALO_PIX && yy >= 0 && yy < A.H && xx >= 0 && xx < A.W;
            float o[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = c1w[288 + grp * 8 + c];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int y2 = yy + kh - 1, x2 = xx + kw - 1;
                    const bool ok = inside && y2 >= 0 && y2 < A.H && x2 >= 0 && x2 < A.W;
                    const float v = ok ? img[(long long)y2 * A.W + x2] : 0.f;
                    const float4 w0 = *reinterpret_cast<const float4 *>(c1w + (kh * 3 + kw) * 32 + grp * 8);
                    const float4 w1 = *reinterpret_cast<const float4 *>(c1w + (kh * 3 + kw) * 32 + grp * 8 + 4);
                    o[0] = fmaf(v, w0.x, o[0]); o[1] = fmaf(v, w0.y, o[1]);
                    o[2] = fmaf(v, w0.z, o[2]); o[3] = fmaf(v, w0.w, o[3]);
                    o[4] = fmaf(v, w1.x, o[4]); o[5] = fmaf(v, w1.y, o[5]);
                    o[6] = fmaf(v, w1.z, o[6]); o[7] = fmaf(v, w1.w, o[7]);
                }
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = inside ? fmaxf(o[c], 0.f) : 0.f;
            if constexpr (ACT_BF16) {
#pragma unroll
                for (int c = 0; c < 8; ++c) rhb[r][c] = static_cast<__bf16>(o[c]);
            } else {
                uint4 hi_, lo_;
#pragma unroll
                for (int c = 0; c < 8; ++c) amax = fmaxf(amax, o[c]);
                split_f16(o, hi_, lo_);
                rhb[r] = __builtin_bit_cast(bf16x8, hi_);
                rhl[r] = __builtin_bit_cast(bf16x8, lo_);
            }
        }
    }
    if (!did_c1) {
        QMRI_LOAD_HALO(0)
    }
    if constexpr (WALL) {
        QMRI_LOAD_WALL(0)
    } else {
        QMRI_LOAD_W(0)
    }
    int step = 0;
    for (int ch = 0; ch < chunks; ++ch) {
        const int hbuf = ch & (A.halo_bufs - 1);
        if (A.halo_bufs == 1 && ch > 0) __syncthreads();  // single halo buffer: every wave is done with the previous chunk
        QMRI_STORE_HALO(hbuf)
        if constexpr (WALL) {
            __syncthreads();  // every wave is done reading the previous chunk's weights
            QMRI_STORE_WALL()
            __syncthreads();
            if (ch + 1 < chunks) {
                QMRI_LOAD_HALO((ch + 1) * kBK)
                QMRI_LOAD_WALL(ch + 1)
            }
        } else {
            if (ch + 1 < chunks) {
                QMRI_LOAD_HALO((ch + 1) * kBK)
            }  // next chunk's halo in flight under 9 taps of MFMA
        }
        const unsigned char *hb = halo_base + hbuf * NPLANES * HALO_BYTES;
#define QMRI_TAP_BODY(t_, ph_)                                                                          \
        {                                                                                               \
            const unsigned char *wb;                                                                    \
            if constexpr (WALL) {                                                                       \
                wb = w_base + (t_) * W_BYTES;                                                           \
            } else {                                                                                    \
                const int wbuf = step & 1;                                                              \
                QMRI_STORE_W(wbuf)                                                                      \
                __syncthreads();                                                                        \
                if (step + 1 < steps) {                                                                 \
                    QMRI_LOAD_W(step + 1)                                                               \
                }                                                                                       \
                wb = w_base + wbuf * NPLANES * W_BYTES;                                                 \
            }                                                                                           \
            const int code = (int)((A.taps >> (4 * (t_))) & 0xF); /* (dy+1) | (dx+1) << 2 */            \
            const int shift = ((code & 3) - 1) * (kTW + 2) + ((code >> 2) - 1);                         \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                          \
                const int koff = (kk * 16 + (lane >> 5) * 8) * 2;                                       \
                bf16x8 a_hi[C::TM], a_lo[C::TM], b_hi[C::TN], b_lo[C::TN];                              \
                _Pragma("unroll") for (int i = 0; i < C::TM; ++i) {                                     \
                    const int off = (a_row0[i] + shift) * kLdsRow * 2 + koff;                           \
                    a_hi[i] = *reinterpret_cast<const bf16x8 *>(hb + off);                              \
                    if (SPLIT3) a_lo[i] = *reinterpret_cast<const bf16x8 *>(hb + HALO_BYTES + off);     \
                }                                                                                       \
                _Pragma("unroll") for (int j = 0; j < C::TN; ++j) {                                     \
                    const int col = (wn * C::TN + j) * 32 + (lane & 31);                                \
                    b_hi[j] = *reinterpret_cast<const bf16x8 *>(wb + col * kLdsRow * 2 + koff);         \
                    if (SPLIT3)                                                                         \
                        b_lo[j] = *reinterpret_cast<const bf16x8 *>(wb + W_BYTES + col * kLdsRow * 2 + koff); \
                }                                                                                       \
                _Pragma("unroll") for (int i = 0; i < C::TM; ++i)                                       \
                _Pragma("unroll") for (int j = 0; j < C::TN; ++j) {                                     \
                    if constexpr (SPLIT3) { /* fp16 parts: lo*hi + hi*lo + hi*hi */                       \
                        const f16x8 ah_ = __builtin_bit_cast(f16x8, a_hi[i]), al_ = __builtin_bit_cast(f16x8, a_lo[i]); \
                        const f16x8 bh_ = __builtin_bit_cast(f16x8, b_hi[j]), bl_ = __builtin_bit_cast(f16x8, b_lo[j]); \
                        acc[ph_][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_, bh_, acc[ph_][i][j], 0, 0, 0); \
                        acc[ph_][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bl_, acc[ph_][i][j], 0, 0, 0); \
                        acc[ph_][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bh_, acc[ph_][i][j], 0, 0, 0); \
                    } else {                                                                            \
                        acc[ph_][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_hi[j], acc[ph_][i][j], 0, 0, 0); \
                    }                                                                                   \
                }                                                                                       \
            }                                                                                           \
            ++step;                                                                                     \
        }
        if constexpr (DECONV) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                constexpr int kPhaseOfTap[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
                switch (kPhaseOfTap[t]) {  // t is a compile-time constant after unrolling
                    case 0: QMRI_TAP_BODY(t, 0) break;
                    case 1: QMRI_TAP_BODY(t, NPH > 1 ? 1 : 0) break;
                    case 2: QMRI_TAP_BODY(t, NPH > 2 ? 2 : 0) break;
                    default: QMRI_TAP_BODY(t, NPH > 3 ? 3 : 0) break;
                }
            }
        } else {
            for (int t = 0; t < ntaps; ++t) QMRI_TAP_BODY(t, 0)
        }
    }
#undef QMRI_TAP_BODY

    // ---- epilogue: y = scale * relu(acc + bias) + shift.  The tile is transposed through LDS so that the
    // global stores are 16 bytes per lane over each pixel's contiguous channel run (a lane of the MFMA
    // C layout owns one channel of 16 different pixels: storing from there would be 2-4 B per lane).
    // Optional fused consumers of the finished tile, all from LDS:
    //   * MaxPooling2D(2x2) of the tile (oaiunet2d.py:234-243) -> A.pool_y (the next level's input);
    //   * the 1x1 classification head + sigmoid threshold (oaiunet2d.py:285, 306) -> logits / mask.
    AT *otile = reinterpret_cast<AT *>(smem);                       // [BM][BN]
    constexpr int CH = BN * (int)sizeof(AT) / 16;  // 16-byte chunks per pixel row of the tile
    float *hw = reinterpret_cast<float *>(smem + C::BM * BN * sizeof(AT));  // head weights [BN][NC] + bias
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
    // DECONV: phase ph = 2*py + px goes to output pixel (2y + py, 2x + px) = rowpix (phase 0) + py*Wo + px
    const int ph_off = DECONV ? (ph >> 1) * A.Wo + (ph & 1) : 0;
    __syncthreads();  // every wave is done reading the halo / weight buffers (or the previous phase's tile)
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
        const int col = (wn * C::TN + j) * 32 + (lane & 31);
        const int n = n0 + col;
        const float bias = A.bias ? A.bias[n] : 0.f;
        const float scale = A.scale ? A.scale[n] : 1.f;
        const float shift = A.shift ? A.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < C::TM; ++i) {
            const int rbase = (wm * C::TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rbase + (e & 3) + 8 * (e >> 2);
                float v = SPLIT3 ? fmaf(acc[ph][i][j][e], A.winv, bias) : acc[ph][i][j][e] + bias;
                if (A.relu) v = fmaxf(v, 0.f);
                v = v * scale + shift;
                otile[row * BN + col] = static_cast<AT>(v);
            }
        }
    }
    if (A.head_w) {
        for (int i = tid; i < BN * A.head_nc + A.head_nc; i += kNT)
            hw[i] = i < BN * A.head_nc ? A.head_w[i] : A.head_b[i - BN * A.head_nc];
    }
    __syncthreads();
    if (A.y) {
        for (int idx = tid; idx < C::BM * CH; idx += kNT) {
            const int row = idx / CH, c = idx - row * CH;
            const int pix = rowpix[row];
            if (pix >= 0) {
                unsigned char *dst = static_cast<unsigned char *>(A.y) +
                                     ((long long)(pix + ph_off) * A.ldy + A.yoff + n0) * (long long)sizeof(AT) +
                                     c * 16;
                if constexpr (SPLIT3) {
                    // 16-byte piece c of the pixel's split run: chunk c / 8, pieces 0-3 = hi parts, 4-7 = lo parts of
                    // channels (c & 3) * 8 .. + 8 of that chunk
                    const float *src = reinterpret_cast<const float *>(otile) + (size_t)row * BN + (c >> 3) * 32 + (c & 3) * 8;
                    float v8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        v8[q] = src[q];
#ifndef QMRI_NO_SAT_TRACK  // (timing experiment: what the saturation tracking costs)
                        amax = fmaxf(amax, fabsf(v8[q]));
#endif
                    }
                    uint4 hi_, lo_;
                    split_f16(v8, hi_, lo_);
                    *reinterpret_cast<uint4 *>(dst) = (c & 4) ? lo_ : hi_;
                } else {
                    const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(otile) +
                                                                     (size_t)row * BN * sizeof(AT) + c * 16);
                    *reinterpret_cast<uint4 *>(dst) = v;
                }
            }
        }
    }
    }  // phases
    if (A.pool_y) {
        constexpr int VPC = 16 / (int)sizeof(AT);  // values per 16-byte chunk
        const int Hp = A.H >> 1, Wp = A.W >> 1;
        for (int idx = tid; idx < (C::BM / 4) * CH; idx += kNT) {
            const int q = idx / CH, c = idx - q * CH;
            const int qy = q / (kTW / 2), qx = q - qy * (kTW / 2);
            const int yy = (y0 >> 1) + qy, xx = (x0 >> 1) + qx;
            if (yy < Hp && xx < Wp) {
                const int r00 = (2 * qy) * kTW + 2 * qx;
                if constexpr (SPLIT3) {
                    const float *p0 = reinterpret_cast<const float *>(otile) + (size_t)r00 * BN + (c >> 3) * 32 + (c & 3) * 8;
                    float m8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        m8[k] = fmaxf(fmaxf(p0[k], p0[BN + k]), fmaxf(p0[kTW * BN + k], p0[(kTW + 1) * BN + k]));
                    uint4 hi_, lo_;
                    split_f16(m8, hi_, lo_);
                    unsigned char *dst = static_cast<unsigned char *>(A.pool_y) +
                                         (((long long)(b * Hp + yy) * Wp + xx) * A.pool_ld + n0) * 4 + c * 16;
                    *reinterpret_cast<uint4 *>(dst) = (c & 4) ? lo_ : hi_;
                } else {
                const AT *p0 = otile + (size_t)r00 * BN + c * VPC;
                AT o[VPC];
#pragma unroll
                for (int k = 0; k < VPC; ++k) {
                    const float m = fmaxf(fmaxf(static_cast<float>(p0[k]), static_cast<float>(p0[BN + k])),
                                          fmaxf(static_cast<float>(p0[kTW * BN + k]),
                                                static_cast<float>(p0[(kTW + 1) * BN + k])));
                    o[k] = static_cast<AT>(m);
                }
                AT *dst = static_cast<AT *>(A.pool_y) +
                          ((long long)(b * Hp + yy) * Wp + xx) * A.pool_ld + n0 + c * VPC;
                *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(o);
                }
            }
        }
    }
    if (A.head_w) {
        // BN/4 lanes per pixel, 4 channels each (a pixel's channels are one contiguous LDS run ->
        // conflict-free), partial dot products combined with wave shuffles.
        constexpr int LPP = BN / 4;
        const int NC = A.head_nc;
        const int sub = tid % LPP;
        float wr[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) wr[q][c] = c < NC ? hw[(sub * 4 + q) * NC + c] : 0.f;
        for (int row = tid / LPP; row < C::BM; row += kNT / LPP) {
            float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = static_cast<float>(otile[row * BN + sub * 4 + q]);
#pragma unroll
                for (int c = 0; c < 4; ++c) z[c] = fmaf(v, wr[q][c], z[c]);
            }
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1)
#pragma unroll
                for (int c = 0; c < 4; ++c) z[c] += __shfl_xor(z[c], o, 64);
            const int pix = rowpix[row];
            if (pix >= 0 && sub < NC) {
                const float zz = (sub == 0 ? z[0] : sub == 1 ? z[1] : sub == 2 ? z[2] : z[3]) + hw[BN * NC + sub];
                if (A.logits) A.logits[(long long)pix * NC + sub] = zz;
                if (A.mask) A.mask[(long long)pix * NC + sub] = zz > 0.f ? 1 : 0;
            }
        }
    }
    if constexpr (SPLIT3)
        if (A.sat && amax > 65504.f) *A.sat = 1;
}

#undef QMRI_LOAD_HALO
#undef QMRI_STORE_HALO
#undef QMRI_LOAD_W
#undef QMRI_STORE_W
#undef QMRI_LOAD_WALL
#undef QMRI_STORE_WALL

template <int BN, int TH, bool S3, int TW = 16, int NW = 4>
static size_t conv_lds_bytes(int halo_bufs = 2) {
    constexpr bool WALL = BN <= 64 && !S3;
    using C = TileCfg<BN, TH, WALL, TW, NW>;
    const int planes = S3 ? 2 : 1;
    const size_t halo = halo_bufs * (size_t)planes * C::HALO_PIX * kLdsRow * 2;
    const size_t w = (WALL ? 9 : 2 * (size_t)planes) * BN * kLdsRow * 2;
    const size_t epilogue = (size_t)C::BM * BN * (S3 ? 4 : 2) + (BN * 4 + 4) * sizeof(float);  // output tile + head weights
    return (halo + w > epilogue ? halo + w : epilogue) + C::BM * sizeof(int) + (BN == 32 ? (9 * 32 + 32) * sizeof(float) : 0);
}

// rows of the output tile in plain-bf16 mode when the image height is a multiple of 16
// (tunable per channel tile through QMRI_CONV_TH128 / _TH64 / _TH32 = 8 | 16 for experiments)
static int conv_tile_rows(int bn) {
    static int cfg[3] = {-1, -1, -1};
    const int i = bn == 128 ? 0 : (bn == 64 ? 1 : 2);
    if (cfg[i] < 0) {
        const char *names[3] = {"QMRI_CONV_TH128", "QMRI_CONV_TH64", "QMRI_CONV_TH32"};
        const int defaults[3] = {8, 8, 16};
        const char *e = std::getenv(names[i]);
        cfg[i] = e ? std::atoi(e) : defaults[i];
        if (cfg[i] != 16) cfg[i] = 8;
    }
    return cfg[i];
}

hipError_t conv_igemm_launch(const ConvKArgs &k0, int split3, hipStream_t stream) {
    // precision 0 (plain bf16): bf16 activations in HBM;  precision 1 (split-bf16 x3): fp32 activations
    ConvKArgs k = k0;
    if (!split3 && conv_rw_supported(k)) return conv_rw_launch(k, stream);
    // the fused transposed convolution keeps 4 accumulator sets: cap the channel tile at 64
    // (plain-bf16 transposed convolution: 64- and 128-channel tiles measured 10-40 % slower than 32)
    // (64-channel tiles for the deep levels, where 128-channel tiles leave ~1 block per CU: measured 25-45 % slower)
    const int bn = k.deconv ? ((split3 && k.Cout % 64 == 0) ? 64 : 32)
                            : (k.Cout % 128 == 0 ? 128 : (k.Cout % 64 == 0 ? 64 : 32));
    if (k.deconv && (k.ntaps != 9 || k.sy != 2 || k.sx != 2 || k.pool_y || k.head_w)) return hipErrorInvalidValue;
    if (k.Cout % bn != 0 || k.Cin % kBK != 0) return hipErrorInvalidValue;
    if (k.c1_x && (k.Cin != 32 || k.deconv)) return hipErrorInvalidValue;
    if (k.head_w && (k.Cout != bn || k.head_nc < 1 || k.head_nc > 4)) return hipErrorInvalidValue;
    if (k.pool_y && (k.sy != 1 || k.sx != 1 || (k.H & 1) || (k.W & 1))) return hipErrorInvalidValue;
    // tile height: 16 rows where the image height allows it and the accumulators fit, else 8
    int th = 8;
    if (!k.deconv) {
        if (split3) th = bn == 128 ? 8 : 16;
        else th = (k.H % 16 == 0) ? conv_tile_rows(bn) : 8;
    }
    static const int narrow_ok = [] { const char *e = std::getenv("QMRI_CONV_NARROW"); return e ? std::atoi(e) : 1; }();
    int tw = kTW;
    if (narrow_ok && !k.deconv && !split3 && bn == 128 && th == 8) {
        if (k.W == 24 && k.H % 8 == 0) tw = 24;
        else if (k.W == 12 && k.H <= 16) tw = 12, th = 16;
    }
    static const int w8 = [] { const char *e = std::getenv("QMRI_CONV_W8"); return e ? std::atoi(e) : 1; }();
    const bool eight = w8 && !k.deconv && bn == 128 && th == 8 && tw == kTW && k.H % 16 == 0 && k.W % 16 == 0;
    if (eight) th = 16;
    // parity mode, a single 32-channel K chunk (the 384 x 384 level and the first conv of the 192 x 192 level): there is
    // no next chunk to prefetch, so one halo buffer is enough -> half the LDS and two or more blocks per CU
    k.halo_bufs = 2;
    static const int s3_t32 = [] { const char *e = std::getenv("QMRI_S3_T32"); return e ? std::atoi(e) : 1; }();
    int s3_small = 0;
    static const int s3_t64 = [] { const char *e = std::getenv("QMRI_S3_T64"); return e ? std::atoi(e) : 1; }();
    static const int s3_t128 = [] { const char *e = std::getenv("QMRI_S3_T128"); return e ? std::atoi(e) : 2; }();
    static const int s3_tdc = [] { const char *e = std::getenv("QMRI_S3_TDC"); return e ? std::atoi(e) : 0; }();
    if (split3 && k.deconv && s3_tdc > 0) k.halo_bufs = 1;
    static const int b16_one = [] { const char *e = std::getenv("QMRI_B16_ONEBUF"); return e ? std::atoi(e) : 0; }();
    if (!split3 && (b16_one == 1 || (b16_one == 2 && !k.deconv) || (b16_one == 3 && k.deconv))) k.halo_bufs = 1;
    bool eight_s3 = eight;
    if (split3 && !k.deconv && k.Cin == kBK && bn <= 64 && s3_t32 > 0) {
        k.halo_bufs = 1;
        s3_small = s3_t32 == 1 ? (bn == 32 ? 2 : 1) : s3_t32;
        if (s3_small == 2) th = 8;
    } else if (split3 && !k.deconv && bn <= 64 && s3_t64 > 0) {
        k.halo_bufs = 1;
        s3_small = s3_t64;
        if (s3_small == 2) th = 8;
    } else if (split3 && !k.deconv && bn == 128 && s3_t128 > 0) {
        k.halo_bufs = 1;
        if (s3_t128 == 2) th = 8, eight_s3 = false;
    }
    // (the same 4-wave 64 x 64 arrangement in the parity mode: 161 / 280 / 252 -> 171 / 298 / 269 us, not used there)
    static const int wide64_ok = [] { const char *e = std::getenv("QMRI_CONV_WIDE64"); return e ? std::atoi(e) : 1; }();
    const bool wide64 = wide64_ok && !split3 && !k.deconv && bn == 64 && !k.c1_x && k.H % 16 == 0 && k.W % 16 == 0;
    if (wide64) {
        th = 16;
        k.halo_bufs = 1;
    }
    k.tiles_y = (k.H + th - 1) / th;
    k.tiles_x = (k.W + tw - 1) / tw;
    dim3 grid((unsigned)((long long)k.B * k.tiles_y * k.tiles_x), (unsigned)(k.Cout / bn));
    (void)hipGetLastError();
#define QMRI_CONV_CASE_TW(BN_, TH_, TW_, NW_)                                                       \
    do {                                                                                            \
        auto fn = conv_igemm_kernel<BN_, TH_, false, __bf16, false, false, TW_, NW_>;               \
        const size_t lds = conv_lds_bytes<BN_, TH_, false, TW_, NW_>(k.halo_bufs);                             \
        if (lds > 64 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                          \
        }                                                                                           \
        hipLaunchKernelGGL(fn, grid, dim3(64 * NW_), lds, stream, k);                               \
    } while (0)
#define QMRI_CONV_CASE(BN_, TH_, S3_, AT_, DC_, ...)                                                     \
    do {                                                                                            \
        auto fn = conv_igemm_kernel<BN_, TH_, S3_, AT_, DC_, ##__VA_ARGS__>;                                  \
        const size_t lds = conv_lds_bytes<BN_, TH_, S3_>(k.halo_bufs);                                       \
        if (lds > 64 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                          \
        }                                                                                           \
        hipLaunchKernelGGL(fn, grid, dim3(256), lds, stream, k);                                    \
    } while (0)
    // parity mode (split3): two bf16 planes of fp32 activations and weights -> ~100-145 KB of LDS, one block per CU
    // whatever the tile; 8 waves on that block instead of 4 (2 per SIMD) is +45 % on the mid levels
#define QMRI_CONV_CASE_W8(BN_, TH_, S3_, AT_, DC_, C1_)                                             \
    do {                                                                                            \
        auto fn = conv_igemm_kernel<BN_, TH_, S3_, AT_, DC_, C1_, 16, 8>;                           \
        const size_t lds = conv_lds_bytes<BN_, TH_, S3_, 16, 8>(k.halo_bufs);                                \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),                      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        if (e != hipSuccess) return e;                                                              \
        hipLaunchKernelGGL(fn, grid, dim3(512), lds, stream, k);                                    \
    } while (0)
    if (k.deconv) {
        if (!split3) QMRI_CONV_CASE(32, 8, false, __bf16, true);
        else if (bn == 64 && w8) QMRI_CONV_CASE_W8(64, 8, true, float, true, false);
        else if (bn == 64) QMRI_CONV_CASE(64, 8, true, float, true);
        else QMRI_CONV_CASE(32, 8, true, float, true);
    } else if (split3) {
        if (bn == 128 && eight_s3) QMRI_CONV_CASE_W8(128, 16, true, float, false, false);
        else if (bn == 128 && w8) QMRI_CONV_CASE_W8(128, 8, true, float, false, false);
        else if (bn == 128) QMRI_CONV_CASE(128, 8, true, float, false);
        else if (bn == 64 && s3_small == 2) QMRI_CONV_CASE(64, 8, true, float, false);
        else if (bn == 64 && s3_small == 3) QMRI_CONV_CASE(64, 16, true, float, false);
        else if (bn == 64 && w8) QMRI_CONV_CASE_W8(64, 16, true, float, false, false);
        else if (bn == 64) QMRI_CONV_CASE(64, 16, true, float, false);
        else if (k.c1_x && s3_small == 2) QMRI_CONV_CASE(32, 8, true, float, false, true);
        else if (k.c1_x && s3_small == 3) QMRI_CONV_CASE(32, 16, true, float, false, true);
        else if (s3_small == 2) QMRI_CONV_CASE(32, 8, true, float, false);
        else if (s3_small == 3) QMRI_CONV_CASE(32, 16, true, float, false);
        else if (k.c1_x && w8) QMRI_CONV_CASE_W8(32, 16, true, float, false, true);
        else if (k.c1_x) QMRI_CONV_CASE(32, 16, true, float, false, true);
        else if (w8) QMRI_CONV_CASE_W8(32, 16, true, float, false, false);
        else QMRI_CONV_CASE(32, 16, true, float, false);
    } else if (bn == 128) {
        if (eight) {
            auto fn = conv_igemm_kernel<128, 16, false, __bf16, false, false, 16, 8>;
            const size_t lds = conv_lds_bytes<128, 16, false, 16, 8>(k.halo_bufs);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(fn, grid, dim3(512), lds, stream, k);
        } else if (tw == 24) QMRI_CONV_CASE_TW(128, 8, 24, 4);  // (6 waves = 3 x 2 on these 192-pixel tiles: 24 x 24 levels
        else if (tw == 12) QMRI_CONV_CASE_TW(128, 16, 12, 4);   //  50 % slower -- 6 waves do not spread over 4 SIMDs)
        else if (th == 16) QMRI_CONV_CASE(128, 16, false, __bf16, false); else QMRI_CONV_CASE(128, 8, false, __bf16, false);
    } else if (bn == 64) {
        if (wide64) {
            // 16 x 16 pixels x 64 channels on 4 waves of 64 x 64 (TM = TN = 2), one halo buffer -> 72 KB, 2 blocks per CU
            auto fn = conv_igemm_kernel<64, 16, false, __bf16, false, false, 16, 4, 1>;
            const size_t lds = conv_lds_bytes<64, 16, false>(k.halo_bufs);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(fn, grid, dim3(256), lds, stream, k);
        } else if (th == 16) QMRI_CONV_CASE(64, 16, false, __bf16, false); else QMRI_CONV_CASE(64, 8, false, __bf16, false);
    } else {
        if (k.c1_x) {
            if (th == 16) QMRI_CONV_CASE(32, 16, false, __bf16, false, true);
            else QMRI_CONV_CASE(32, 8, false, __bf16, false, true);
        } else {
            if (th == 16) QMRI_CONV_CASE(32, 16, false, __bf16, false); else QMRI_CONV_CASE(32, 8, false, __bf16, false);
        }
    }
#undef QMRI_CONV_CASE
#undef QMRI_CONV_CASE_W8
#undef QMRI_CONV_CASE_TW
    return hipGetLastError();
}

// ---- activation load/store helpers (AT = float or __bf16) ------------------------------------------
template <typename AT>
__device__ __forceinline__ void load4(const AT *p, float (&v)[4]) {
    if constexpr (sizeof(AT) == 4) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        const bf16x4 q = *reinterpret_cast<const bf16x4 *>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = static_cast<float>(q[i]);
    }
}
template <typename AT>
__device__ __forceinline__ void store4(AT *p, const float (&v)[4]) {
    if constexpr (sizeof(AT) == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 q;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = static_cast<__bf16>(v[i]);
        *reinterpret_cast<bf16x4 *>(p) = q;
    }
}

// ---- first layer: Conv2D(C, 3x3, SAME) on ONE input channel + bias + ReLU (fp32 VALU) ---------------
// x [B][H][W] fp32; w [9][Cout] fp32 (tap-major); y NHWC (AT) with pixel stride ldy / channel offset yoff.
// One thread = one pixel x 8 output channels.
template <typename AT>
__global__ __launch_bounds__(256) void conv3x3_c1_kernel(const float *__restrict__ x, int B, int H, int W,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ bias, int Cout,
                                                         AT *__restrict__ y, long long ldy, int yoff) {
    const int groups = Cout / 8;
    const long long total = (long long)B * H * W * groups;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups);
        const long long pix = idx / groups;
        const int xw = (int)(pix % W);
        const long long t = pix / W;
        const int yh = (int)(t % H);
        const float *img = x + (t / H) * (long long)H * W;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = bias[g * 8 + c];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int yy = yh + kh - 1, xx = xw + kw - 1;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float v = img[(long long)yy * W + xx];
                    const float *wt = w + (kh * 3 + kw) * Cout + g * 8;
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = fmaf(v, wt[c], acc[c]);
                }
            }
        const float o0[4] = {fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f)};
        const float o1[4] = {fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f)};
        AT *dst = y + pix * ldy + yoff + g * 8;
        store4(dst, o0);
        store4(dst + 4, o1);
    }
}

hipError_t conv3x3_c1_launch(const float *x, int B, int H, int W, const float *w, const float *bias,
                             int Cout, void *y, long long ldy, int yoff, int act_bf16, hipStream_t stream) {
    const long long total = (long long)B * H * W * (Cout / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    (void)hipGetLastError();
    if (act_bf16)
        hipLaunchKernelGGL(conv3x3_c1_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, x, B, H, W,
                           w, bias, Cout, static_cast<__bf16 *>(y), ldy, yoff);
    else
        hipLaunchKernelGGL(conv3x3_c1_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, x, B, H, W,
                           w, bias, Cout, static_cast<float *>(y), ldy, yoff);
    return hipGetLastError();
}

// ---- MaxPooling2D(2x2): NHWC, input with pixel stride / channel offset, output compact ---------------
template <typename AT>
__global__ __launch_bounds__(256) void maxpool2_kernel(const AT *__restrict__ x, long long ldx, int xoff,
                                                       int B, int H, int W, int C, AT *__restrict__ y) {
    const int Ho = H / 2, Wo = W / 2, cg = C / 4;
    const long long total = (long long)B * Ho * Wo * cg;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cg) * 4;
        const long long p = idx / cg;
        const int xo = (int)(p % Wo);
        const long long t = p / Wo;
        const int yo = (int)(t % Ho);
        const long long b = t / Ho;
        const AT *src = x + ((b * H + 2 * yo) * W + 2 * xo) * ldx + xoff + c;
        float v00[4], v01[4], v10[4], v11[4], o[4];
        load4(src, v00);
        load4(src + ldx, v01);
        load4(src + (long long)W * ldx, v10);
        load4(src + (long long)W * ldx + ldx, v11);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fmaxf(fmaxf(v00[i], v01[i]), fmaxf(v10[i], v11[i]));
        store4(y + p * C + c, o);
    }
}

// MaxPooling2D(k x k) on the bf16 layout, any k (the reference's 3 x 3 branch for odd heights, oaiunet2d.py:236-241)
__global__ __launch_bounds__(256) void maxpoolk_bf16_kernel(const __bf16 *__restrict__ x, long long ldx, int xoff, int B, int H,
                                                            int W, int C, int K, __bf16 *__restrict__ y) {
    const int Ho = H / K, Wo = W / K, cg = C / 4;
    const long long total = (long long)B * Ho * Wo * cg;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cg) * 4;
        const long long p = idx / cg;
        const int xo = (int)(p % Wo);
        const long long t = p / Wo;
        const int yo = (int)(t % Ho);
        const long long b = t / Ho;
        float o[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int q = 0; q < K * K; ++q) {
            const __bf16 *src = x + ((b * H + K * yo + q / K) * W + K * xo + q % K) * ldx + xoff + c;
            float v[4];
            load4(src, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i], v[i]);
        }
        store4(y + p * C + c, o);
    }
}
hipError_t maxpoolk_launch(const void *x, long long ldx, int xoff, int B, int H, int W, int C, int K, void *y, hipStream_t stream) {
    const long long total = (long long)B * (H / K) * (W / K) * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    if (blocks < 1) blocks = 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(maxpoolk_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const __bf16 *>(x), ldx, xoff,
                       B, H, W, C, K, static_cast<__bf16 *>(y));
    return hipGetLastError();
}

hipError_t maxpool2_launch(const void *x, long long ldx, int xoff, int B, int H, int W, int C, void *y,
                           int act_bf16, hipStream_t stream) {
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    (void)hipGetLastError();
    if (act_bf16)
        hipLaunchKernelGGL(maxpool2_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           static_cast<const __bf16 *>(x), ldx, xoff, B, H, W, C, static_cast<__bf16 *>(y));
    else
        hipLaunchKernelGGL(maxpool2_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           static_cast<const float *>(x), ldx, xoff, B, H, W, C, static_cast<float *>(y));
    return hipGetLastError();
}

// ---- head: Conv2D(n_classes <= 4, 1x1) -> logits fp32 [pix][NC] and mask u8 [pix][NC] = logit > 0 -----
// Cin/4 lanes cooperate on one pixel (each reads 4 consecutive channels -> a pixel's channels are one
// contiguous, coalesced segment), partial dot products are reduced with wave shuffles.
template <typename AT, int LPP /*lanes per pixel = Cin/4: 8 for Cin = 32*/>
__global__ __launch_bounds__(256) void head_kernel(const AT *__restrict__ x, long long npix, int Cin,
                                                   const float *__restrict__ w /*[Cin][NC]*/,
                                                   const float *__restrict__ bias, int NC,
                                                   float *__restrict__ logits,
                                                   unsigned char *__restrict__ mask) {
    const int sub = threadIdx.x % LPP;
    float wr[4][4];  // this lane's 4 channels x up to 4 classes
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) wr[q][c] = c < NC ? w[(sub * 4 + q) * NC + c] : 0.f;
    const long long ppb = 256 / LPP;  // pixels per block iteration
    for (long long p = (long long)blockIdx.x * ppb + threadIdx.x / LPP; p < npix; p += (long long)gridDim.x * ppb) {
        float v[4];
        load4(x + p * Cin + sub * 4, v);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(v[q], wr[q][c], acc[c]);
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
        if (sub < NC) {
            const float z = (sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3]) + bias[sub];
            if (logits) logits[p * NC + sub] = z;
            if (mask) mask[p * NC + sub] = z > 0.f ? 1 : 0;  // sigmoid(z) > 0.5  <=>  z > 0
        }
    }
}

hipError_t head_launch(const void *x, long long npix, int Cin, const float *w, const float *bias, int NC,
                       float *logits, unsigned char *mask, int act_bf16, hipStream_t stream) {
    if (NC > 4 || (Cin != 32 && Cin != 64 && Cin != 128 && Cin != 256)) return hipErrorInvalidValue;
    long long blocks = (npix * (Cin / 4) + 255) / 256;
    if (blocks > 65535 * 8) blocks = 65535 * 8;
    (void)hipGetLastError();
#define QMRI_HEAD(LPP_)                                                                                \
    do {                                                                                               \
        if (act_bf16)                                                                                  \
            hipLaunchKernelGGL((head_kernel<__bf16, LPP_>), dim3((unsigned)blocks), dim3(256), 0, stream, \
                               static_cast<const __bf16 *>(x), npix, Cin, w, bias, NC, logits, mask);   \
        else                                                                                           \
            hipLaunchKernelGGL((head_kernel<float, LPP_>), dim3((unsigned)blocks), dim3(256), 0, stream,  \
                               static_cast<const float *>(x), npix, Cin, w, bias, NC, logits, mask);    \
    } while (0)
    if (Cin == 32) QMRI_HEAD(8);
    else if (Cin == 64) QMRI_HEAD(16);
    else if (Cin == 128) QMRI_HEAD(32);
    else QMRI_HEAD(64);
#undef QMRI_HEAD
    return hipGetLastError();
}

// ---- dtype casts (operator-level host entry in bf16 mode) ------------------------------------------
__global__ void cast_f32_bf16_kernel(const float *__restrict__ x, long long n, __bf16 *__restrict__ y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        y[i] = static_cast<__bf16>(x[i]);
}
__global__ void cast_bf16_f32_kernel(const __bf16 *__restrict__ x, long long n, float *__restrict__ y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        y[i] = static_cast<float>(x[i]);
}
hipError_t cast_launch(const void *x, long long n, void *y, int to_bf16, hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    (void)hipGetLastError();
    if (to_bf16)
        hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                           static_cast<const float *>(x), n, static_cast<__bf16 *>(y));
    else
        hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                           static_cast<const __bf16 *>(x), n, static_cast<float *>(y));
    return hipGetLastError();
}

// ---- whiten_volume (seg_model.py:114-127): two-pass mean / std in fp64, then (x - mean)/(std + eps) ---
// The reductions are DETERMINISTIC: every block writes its partial sum to its own slot, one block adds the slots in a fixed order.
// (Rounds 1-4 let the blocks atomicAdd their partial sums: the order of 2048 double additions -- hence the last bits of mean and
//  standard deviation, hence the rounding of a handful of whitened voxels -- changed from run to run; two of 160 slices of a 512 x 512
//  volume then came out with logits 6e-5 apart in one run of ten.  Found by round 5's bitwise test across pass sizes.)
constexpr int kWhitenBlocks = 2048;  // slots of the partial-sum table behind the three statistics (qmri_internal.h: whiten_stats_doubles)
__global__ __launch_bounds__(256) void whiten_partial_kernel(const float *__restrict__ x, long long n, const double *__restrict__ stats,
                                                             int square, double *__restrict__ part) {
    const double center = square ? stats[2] : 0.0;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const double d = (double)x[i] - center;
        acc += square ? d * d : d;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double w4[4];
    if ((threadIdx.x & 63) == 0) w4[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (w4[0] + w4[1]) + (w4[2] + w4[3]);
}
// stats[slot] = sum of part[0 .. nblocks) in a fixed order (thread t adds slots t, t + 256, ...; then a fixed tree); slot 0 also
// leaves the mean in stats[2]
__global__ __launch_bounds__(256) void whiten_final_kernel(const double *__restrict__ part, int nblocks, double *__restrict__ stats,
                                                           int slot, long long n) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += part[i];
    __shared__ double tree[256];
    tree[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) tree[threadIdx.x] += tree[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[slot] = tree[0];
        if (slot == 0) stats[2] = tree[0] / (double)n;
    }
}

__global__ __launch_bounds__(256) void whiten_apply_kernel(const float *__restrict__ x, long long n,
                                                           const double *__restrict__ stats, double eps,
                                                           float *__restrict__ y) {
    // stats[0] = sum(x), stats[1] = sum((x-mean)^2)
    const double mean = stats[0] / (double)n;
    const double sd = sqrt(stats[1] / (double)n);
    const double inv = 1.0 / (sd + eps);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        y[i] = (float)(((double)x[i] - mean) * inv);
}

// ---- activation exponent of the parity mode (unet_engine.hip: Unet::sat): max |x| of the input, and y = x * 2^-S ----
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, long long n, unsigned int *__restrict__ out) {
    // on the BIT PATTERNS of |x|: non-negative floats order like their patterns, +inf sorts above every finite value and a NaN
    // above +inf (fmaxf would drop it) -- the host sees a non-finite input as a pattern >= 0x7f800000
    unsigned int m = 0u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned int b = __float_as_uint(fabsf(x[i]));
        m = b > m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int other = (unsigned int)__shfl_down((int)m, o, 64);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
__global__ __launch_bounds__(256) void scale_copy_kernel(const float *__restrict__ x, long long n, float f, float *__restrict__ y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = x[i] * f;
}
hipError_t absmax_launch(const float *x, long long n, unsigned int *out /*device, zeroed here*/, hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, out);
    return hipGetLastError();
}
hipError_t scale_copy_launch(const float *x, long long n, float f, float *y, hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    (void)hipGetLastError();
    hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, f, y);
    return hipGetLastError();
}

int whiten_stats_doubles() { return 4 + kWhitenBlocks; }

hipError_t whiten_launch(const float *x, long long n, double eps, double *stats /*[whiten_stats_doubles()] device*/, float *y,
                         hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    (void)hipGetLastError();
    static_assert(kWhitenBlocks == 2048, "the grid cap above");
    double *part = stats + 4;  // [kWhitenBlocks] partial sums (the caller's buffer holds whiten_stats_doubles() doubles)
    hipLaunchKernelGGL(whiten_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, stats, 0, part);
    hipLaunchKernelGGL(whiten_final_kernel, dim3(1), dim3(256), 0, stream, part, (int)blocks, stats, 0, n);
    hipLaunchKernelGGL(whiten_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, stats, 1, part);
    hipLaunchKernelGGL(whiten_final_kernel, dim3(1), dim3(256), 0, stream, part, (int)blocks, stats, 1, n);
    hipLaunchKernelGGL(whiten_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, stats, eps, y);
    return hipGetLastError();
}

// ---- volume layout kernels (SegModel.generate_mask's transposes, oaiunet2d.py:295-303, 309-316) --------------
// x (P, S) fp32 -> y (S, P): the reference reorders the sagittal volume (H, W, slices) to (slices, H, W, 1)
__global__ __launch_bounds__(256) void transpose_ps_kernel(const float *__restrict__ x, long long P, int S,
                                                           float *__restrict__ y) {
    __shared__ float tile[32][33];
    const long long p0 = (long long)blockIdx.x * 32;
    const int s0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const long long p = p0 + r;
        const int s = s0 + tx;
        tile[r][tx] = (p < P && s < S) ? x[p * S + s] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int s = s0 + r;
        const long long p = p0 + tx;
        if (s < S && p < P) y[(long long)s * P + p] = tile[tx][r];
    }
}
// mask (S, P, 4) u8 -> planes (C, P, S): one (H, W, slices) uint8 volume per class
__global__ __launch_bounds__(256) void mask_planes_kernel(const unsigned int *__restrict__ m, long long P, int S, int C,
                                                          unsigned char *__restrict__ out) {
    __shared__ unsigned int tile[32][33];
    const long long p0 = (long long)blockIdx.x * 32;
    const int s0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int s = s0 + r;
        const long long p = p0 + tx;
        tile[r][tx] = (s < S && p < P) ? m[(long long)s * P + p] : 0u;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long long p = p0 + r;
        const int s = s0 + tx;
        if (p < P && s < S) {
            const unsigned int v = tile[tx][r];
            for (int c = 0; c < C; ++c) out[((long long)c * P + p) * S + s] = (unsigned char)((v >> (8 * c)) & 0xFFu);
        }
    }
}

hipError_t transpose_ps_launch(const float *x, long long P, int S, float *y, hipStream_t stream) {
    (void)hipGetLastError();
    dim3 grid((unsigned)((P + 31) / 32), (unsigned)((S + 31) / 32));
    hipLaunchKernelGGL(transpose_ps_kernel, grid, dim3(256), 0, stream, x, P, S, y);
    return hipGetLastError();
}
hipError_t mask_planes_launch(const unsigned char *mask_sp4, long long P, int S, int C, unsigned char *out,
                              hipStream_t stream) {
    (void)hipGetLastError();
    dim3 grid((unsigned)((P + 31) / 32), (unsigned)((S + 31) / 32));
    hipLaunchKernelGGL(mask_planes_kernel, grid, dim3(256), 0, stream, reinterpret_cast<const unsigned int *>(mask_sp4), P, S,
                       C, out);
    return hipGetLastError();
}

}  // namespace qmri
