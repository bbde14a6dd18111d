// unet_engine.hip -- host-side engine of the 2D U-Net inference path + its C ABI (include/qmri.h).
//
// Replaces `model.predict(v, batch_size)` of the reference
//     /root/reference/dosma/models/oaiunet2d.py:305   (inside generate_mask, :291-320)
// for the graph built at oaiunet2d.py:197-289: it packs Keras-layout weights once (bf16 hi/lo,
// K-major per output channel; BatchNormalization folded to a per-channel scale/shift applied in the
// producing convolution's epilogue), owns the activation buffers (fp32 NHWC; the skip and the
// transposed-convolution output of a level share one "concat" buffer so Concatenate is free), and
// issues the layer kernels of unet_kernels.hip on the caller's stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "qmri_internal.h"

namespace {

thread_local char u_err[512] = "";
int ufail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(u_err, sizeof(u_err), fmt, ap);
    va_end(ap);
    qmri::set_last_error(u_err);
    return code;
}

#define U_TRY(expr)                                                                             \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return ufail(e_ == hipErrorOutOfMemory ? QMRI_ERR_NOMEM : QMRI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                         hipGetErrorString(e_), __FILE__, __LINE__);                            \
    } while (0)

unsigned short f32_to_bf16_rne(float f) {
    unsigned int u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    const unsigned int lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (unsigned short)(u >> 16);
}
float bf16_to_f32(unsigned short h) {
    unsigned int u = (unsigned int)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) {
        if (p) (void)hipFree(p);
        p = nullptr;
        return hipMalloc(&p, bytes ? bytes : 16);
    }
    template <typename T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

// parity mode ("fp16x3"): weights are stored as fp16 hi + lo parts of 2^wshift * w.  The power-of-two pre-scale keeps the
// lo parts of typical weights (|w| ~ 1e-2: lo ~ 5e-6) out of the fp16 subnormal range -- scripts/unet_precision_sim.py:
// 1.7e-4 -> 1.4e-5 on the logits -- and is undone for free in the epilogue's first fma (acc * 2^-wshift + bias).
int weight_shift(const std::vector<float> &wk) {
    float m = 0.f;
    for (float v : wk) m = std::fmax(m, std::fabs(v));
    if (!(m > 0.f) || !std::isfinite(m)) return 0;
    int e = 0;
    std::frexp(m, &e);           // m = f * 2^e, f in [0.5, 1)
    int sh = 14 - e;             // 2^sh * m in [2^13, 2^14): far below the fp16 maximum 65504
    if (sh < -14) sh = -14;
    if (sh > 24) sh = 24;
    return sh;
}
void split_f16_host(float v, unsigned short &hi, unsigned short &lo) {
    const _Float16 h = static_cast<_Float16>(v);  // round to nearest even
    const _Float16 l = static_cast<_Float16>(v - static_cast<float>(h));
    std::memcpy(&hi, &h, 2);
    std::memcpy(&lo, &l, 2);
}

struct ConvLayer {
    int Cin = 0, Cout = 0, ntaps = 0;
    int dy[9] = {0}, dx[9] = {0};
    int relu = 0;
    int deconv = 0;
    DevBuf w_hi, w_lo, bias, scale, shift;
    DevBuf h_hi, h_lo;   // parity mode, general kernel: fp16 hi / lo parts, same K-major layout as w_hi
    DevBuf w_s3;         // parity mode, conv_s3_kernel: per channel block and K step the [plane][BN][64 B] LDS image
    DevBuf w_c4;         // parity mode, conv_c4_kernel (Cout % 128 == 0): per channel block and k-step (chunk, 16-channel half, tap)
                         // the 8 KB LDS image [plane][128 rows][2 x 16 B], piece g of row r at position g ^ ((r >> 3) & 1)
    DevBuf w_s3b;        // plain-bf16 mode on conv_s3_kernel (Cin % 64 == 0): the same image geometry with 64-channel chunks,
                         // plane p = channels 32 p .. 32 p + 31 of the chunk, bf16 values
    float winv = 1.f;    // 2^-wshift
    bool has_affine = false;
    std::vector<float> bias_h, shift_h;  // host copies of the ADDITIVE epilogue parameters (rescaled by Unet::set_act_shift)

    // fp16 hi / lo images of the host weights W[co][K] (K = (chunk * ntaps + tap) * 32 + c)
    // for_s3: the level is one conv_s3_kernel tiles (W % 32 == 0 or W <= 48): its image + conv_c4_kernel's / deconv_d4_kernel's;
    // otherwise the general kernel's parts -- plus, with `ragged` (W % 32 != 0, wider than the flattened tiling), the c4 / d4 image:
    // those kernels take such a level on image tiles with a ragged last column, the general kernel stays the fallback (QMRI_C4 = 0)
    hipError_t upload_parity(const std::vector<float> &wk, bool for_s3, bool ragged = false) {
        const int sh = weight_shift(wk);
        const float sc = std::ldexp(1.f, sh);
        winv = std::ldexp(1.f, -sh);
        const size_t n = wk.size();
        std::vector<unsigned short> hi(n), lo(n);
        for (size_t i = 0; i < n; ++i) split_f16_host(wk[i] * sc, hi[i], lo[i]);
        hipError_t e = hipSuccess;
        if (!for_s3) {
            e = h_hi.alloc(n * 2);
            if (e == hipSuccess) e = h_lo.alloc(n * 2);
            if (e == hipSuccess) e = hipMemcpy(h_hi.p, hi.data(), n * 2, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(h_lo.p, lo.data(), n * 2, hipMemcpyHostToDevice);
            if (e != hipSuccess || !ragged || ntaps != 9 || Cin % 32) return e;
        }
        // conv_s3_kernel: [nb][step][plane][BN rows][4 positions x 8 halfs], position pos of row n holds the K piece
        // q = pos ^ ((n >> 2) & 3) (the bank-conflict swizzle of the LDS image; the DMA copies the image linearly)
        const int BN = qmri::conv_s3_block_channels(Cout, deconv);
        const int steps = ntaps * (Cin / 32);
        const size_t K = (size_t)steps * 32;
        std::vector<unsigned short> img(for_s3 ? (size_t)Cout * K * 2 : 0);
        for (int nb = 0; for_s3 && nb < Cout / BN; ++nb)
            for (int st = 0; st < steps; ++st)
                for (int plane = 0; plane < 2; ++plane)
                    for (int r = 0; r < BN; ++r)
                        for (int pos = 0; pos < 4; ++pos) {
                            const int q = pos ^ ((r >> 2) & 3);
                            const size_t src = (size_t)(nb * BN + r) * K + (size_t)st * 32 + q * 8;
                            const size_t dst = ((((size_t)nb * steps + st) * 2 + plane) * BN + r) * 32 + pos * 8;
                            const unsigned short *from = (plane ? lo.data() : hi.data()) + src;
                            for (int k = 0; k < 8; ++k) img[dst + k] = from[k];
                        }
        if (for_s3) {
            e = w_s3.alloc(img.size() * 2);
            if (e == hipSuccess) e = hipMemcpy(w_s3.p, img.data(), img.size() * 2, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess || ntaps != 9) return e;
        if (deconv) {
            // deconv_d4_kernel: [nb = Cout / 32][k-step = chunk * 2 + half] slots of nine taps in SHIFT-GROUP order -- A: shift (0, 0),
            // B: (0, -1), C: (-1, 0), D: (-1, -1) -- each [plane][32 rows][2 positions x 8 halfs], piece g of row r at position
            // g ^ ((r >> 3) & 1).  pack_deconv_fused's tap order is by phase: t = 0..3 phase (0,0) with shifts (0,0) (0,-1) (-1,0)
            // (-1,-1); 4, 5 phase (0,1) with (0,0) (-1,0); 6, 7 phase (1,0) with (0,0) (0,-1); 8 phase (1,1).
            static const int kGroupOrder[9] = {0, 4, 6, 8, 1, 7, 2, 5, 3};  // slot s holds tap kGroupOrder[s]: phases 0 1 2 3 | 0 2 | 0 1 | 0
            if (Cout % 32) return e;
            const int chunks_d = Cin / 32;
            std::vector<unsigned short> imd((size_t)Cout * K * 2);
            for (int nb = 0; nb < Cout / 32; ++nb)
                for (int ch = 0; ch < chunks_d; ++ch)
                    for (int half = 0; half < 2; ++half)
                        for (int sl = 0; sl < 9; ++sl) {
                            const int tap = kGroupOrder[sl];
                            const size_t slot = ((((size_t)nb * chunks_d + ch) * 2 + half) * 9 + sl) * (size_t)(32 * 32);
                            for (int plane = 0; plane < 2; ++plane)
                                for (int r = 0; r < 32; ++r)
                                    for (int g = 0; g < 2; ++g) {
                                        const int pos = g ^ ((r >> 3) & 1);
                                        const size_t src = (size_t)(nb * 32 + r) * K + ((size_t)ch * 9 + tap) * 32 + half * 16 + g * 8;
                                        const size_t dst = slot + (size_t)plane * (32 * 16) + (size_t)r * 16 + pos * 8;
                                        const unsigned short *from = (plane ? lo.data() : hi.data()) + src;
                                        for (int k = 0; k < 8; ++k) imd[dst + k] = from[k];
                                    }
                        }
            e = w_c4.alloc(imd.size() * 2);
            if (e == hipSuccess) e = hipMemcpy(w_c4.p, imd.data(), imd.size() * 2, hipMemcpyHostToDevice);
            return e;
        }
        if (Cout % 64) return e;
        // conv_c4_kernel: [nb][chunk][half][tap] slots of [plane][B4 rows][2 positions x 8 halfs], B4 = 128 or 64 channels per block
        const int chunks = Cin / 32;
        const int B4 = qmri::conv_c4_block_channels(Cout);
        std::vector<unsigned short> im4((size_t)Cout * K * 2);
        for (int nb = 0; nb < Cout / B4; ++nb)
            for (int ch = 0; ch < chunks; ++ch)
                for (int half = 0; half < 2; ++half)
                    for (int tap = 0; tap < 9; ++tap) {
                        const size_t slot = ((((size_t)nb * chunks + ch) * 2 + half) * 9 + tap) * (size_t)(B4 * 32);
                        for (int plane = 0; plane < 2; ++plane)
                            for (int r = 0; r < B4; ++r)
                                for (int g = 0; g < 2; ++g) {
                                    const int pos = g ^ ((r >> 3) & 1);
                                    const size_t src = (size_t)(nb * B4 + r) * K + ((size_t)ch * 9 + tap) * 32 + half * 16 + g * 8;
                                    const size_t dst = slot + (size_t)plane * (B4 * 16) + (size_t)r * 16 + pos * 8;
                                    const unsigned short *from = (plane ? lo.data() : hi.data()) + src;
                                    for (int k = 0; k < 8; ++k) im4[dst + k] = from[k];
                                }
                    }
        e = w_c4.alloc(im4.size() * 2);
        if (e == hipSuccess) e = hipMemcpy(w_c4.p, im4.data(), im4.size() * 2, hipMemcpyHostToDevice);
        return e;
    }

    // plain-bf16 image for conv_s3_kernel<.., ONE>: [nb][chunk64 * ntaps + tap][plane][BN rows][4 positions x 8 bf16]
    hipError_t upload_s3_bf16(const std::vector<float> &wk) {
        if (Cin % 64 || Cout % 32) return hipSuccess;
        const int BN = qmri::conv_s3_block_channels(Cout, deconv);
        const int steps = ntaps * (Cin / 64);
        const size_t K = (size_t)ntaps * Cin;
        std::vector<unsigned short> img((size_t)Cout * K);
        for (int nb = 0; nb < Cout / BN; ++nb)
            for (int st = 0; st < steps; ++st) {
                const int c64 = st / ntaps, tap = st % ntaps;
                for (int plane = 0; plane < 2; ++plane)
                    for (int r = 0; r < BN; ++r)
                        for (int pos = 0; pos < 4; ++pos) {
                            const int q = pos ^ ((r >> 2) & 3);
                            const size_t src = (size_t)(nb * BN + r) * K + ((size_t)(c64 * 2 + plane) * ntaps + tap) * 32 + q * 8;
                            const size_t dst = ((((size_t)nb * steps + st) * 2 + plane) * BN + r) * 32 + pos * 8;
                            for (int k = 0; k < 8; ++k) img[dst + k] = f32_to_bf16_rne(wk[src + k]);
                        }
            }
        hipError_t e = w_s3b.alloc(img.size() * 2);
        if (e == hipSuccess) e = hipMemcpy(w_s3b.p, img.data(), img.size() * 2, hipMemcpyHostToDevice);
        return e;
    }

    // pack host weights W[co][t*Cin + ci] (fp32) into bf16 hi/lo and upload
    hipError_t upload(const std::vector<float> &wk, const float *b, const std::vector<float> *sc,
                      const std::vector<float> *sh) {
        const size_t n = wk.size();
        std::vector<unsigned short> hi(n), lo(n);
        for (size_t i = 0; i < n; ++i) {
            hi[i] = f32_to_bf16_rne(wk[i]);
            lo[i] = f32_to_bf16_rne(wk[i] - bf16_to_f32(hi[i]));
        }
        hipError_t e = w_hi.alloc(n * 2);
        if (e == hipSuccess) e = w_lo.alloc(n * 2);
        if (e == hipSuccess) e = hipMemcpy(w_hi.p, hi.data(), n * 2, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(w_lo.p, lo.data(), n * 2, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = bias.alloc((size_t)Cout * 4);
        if (e == hipSuccess) e = hipMemcpy(bias.p, b, (size_t)Cout * 4, hipMemcpyHostToDevice);
        bias_h.assign(b, b + Cout);
        has_affine = sc != nullptr;
        if (has_affine) shift_h = *sh;
        if (has_affine) {
            if (e == hipSuccess) e = scale.alloc((size_t)Cout * 4);
            if (e == hipSuccess) e = shift.alloc((size_t)Cout * 4);
            if (e == hipSuccess) e = hipMemcpy(scale.p, sc->data(), (size_t)Cout * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(shift.p, sh->data(), (size_t)Cout * 4, hipMemcpyHostToDevice);
        }
        return e;
    }
};

// Conv2D 3x3 SAME: Keras kernel (kh, kw, Cin, Cout) -> W[co][(chunk*9 + tap)*32 + c] with ci = chunk*32 + c,
// tap = kh*3 + kw, dy = kh-1, dx = kw-1 (chunk-major K so that all 9 taps reuse one LDS-resident halo)
void pack_conv3x3(const float *k, int Cin, int Cout, ConvLayer &L, std::vector<float> &wk) {
    L.Cin = Cin;
    L.Cout = Cout;
    L.ntaps = 9;
    wk.assign((size_t)Cout * 9 * Cin, 0.f);
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int t = kh * 3 + kw;
            L.dy[t] = kh - 1;
            L.dx[t] = kw - 1;
            for (int ci = 0; ci < Cin; ++ci)
                for (int co = 0; co < Cout; ++co)
                    wk[(size_t)co * 9 * Cin + ((size_t)(ci / 32) * 9 + t) * 32 + ci % 32] =
                        k[(((size_t)kh * 3 + kw) * Cin + ci) * Cout + co];
        }
}

// Conv2DTranspose 3x3 stride 2 SAME, output phase (py, px): out[2m+py] = sum_{kh: 2o+kh = 2m+py} in[o] w[kh]
//   py = 0: kh = 0 (o = m), kh = 2 (o = m-1);   py = 1: kh = 1 (o = m).   Keras kernel (kh, kw, Cout, Cin).
void pack_deconv_phase(const float *k, int Cin, int Cout, int py, int px, ConvLayer &L, std::vector<float> &wk) {
    L.Cin = Cin;
    L.Cout = Cout;
    int khs[2], kws[2], nkh, nkw;
    if (py == 0) { khs[0] = 0; khs[1] = 2; nkh = 2; } else { khs[0] = 1; nkh = 1; }
    if (px == 0) { kws[0] = 0; kws[1] = 2; nkw = 2; } else { kws[0] = 1; nkw = 1; }
    L.ntaps = nkh * nkw;
    wk.assign((size_t)Cout * L.ntaps * Cin, 0.f);
    int t = 0;
    for (int a = 0; a < nkh; ++a)
        for (int b = 0; b < nkw; ++b, ++t) {
            const int kh = khs[a], kw = kws[b];
            L.dy[t] = kh == 2 ? -1 : 0;
            L.dx[t] = kw == 2 ? -1 : 0;
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                    wk[(size_t)co * L.ntaps * Cin + ((size_t)(ci / 32) * L.ntaps + t) * 32 + ci % 32] =
                        k[(((size_t)kh * 3 + kw) * Cout + co) * Cin + ci];
        }
}

// Conv2DTranspose 3x3 stride 2 SAME, all four phases fused: 9 taps in the order
//   phase (py,px) = (0,0): (kh,kw) = (0,0) (0,2) (2,0) (2,2);  (0,1): (0,1) (2,1);  (1,0): (1,0) (1,2);  (1,1): (1,1)
// with dy = (kh == 2 ? -1 : 0), dx = (kw == 2 ? -1 : 0).  Keras kernel (kh, kw, Cout, Cin).
void pack_deconv_fused(const float *k, int Cin, int Cout, ConvLayer &L, std::vector<float> &wk) {
    static const int KH[9] = {0, 0, 2, 2, 0, 2, 1, 1, 1};
    static const int KW[9] = {0, 2, 0, 2, 1, 1, 0, 2, 1};
    L.Cin = Cin;
    L.Cout = Cout;
    L.ntaps = 9;
    L.deconv = 1;
    wk.assign((size_t)Cout * 9 * Cin, 0.f);
    for (int t = 0; t < 9; ++t) {
        const int kh = KH[t], kw = KW[t];
        L.dy[t] = kh == 2 ? -1 : 0;
        L.dx[t] = kw == 2 ? -1 : 0;
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci)
                wk[(size_t)co * 9 * Cin + ((size_t)(ci / 32) * 9 + t) * 32 + ci % 32] =
                    k[(((size_t)kh * 3 + kw) * Cout + co) * Cin + ci];
    }
}

// Conv2DTranspose 3x3 strides 3 SAME (the reference's branch for odd sizes, oaiunet2d.py:250-261): out[3 o + k] = in[o] w[k]
// -- the kernel positions do not overlap, so output phase (ky, kx) is ONE tap at the input pixel itself: a 1x1 convolution
// written to the output pixels (3 y + ky, 3 x + kx).  Keras kernel (kh, kw, Cout, Cin).
void pack_deconv3_phase(const float *k, int Cin, int Cout, int ky, int kx, ConvLayer &L, std::vector<float> &wk) {
    L.Cin = Cin;
    L.Cout = Cout;
    L.ntaps = 1;
    L.dy[0] = L.dx[0] = 0;
    wk.assign((size_t)Cout * Cin, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            wk[(size_t)co * Cin + (size_t)(ci / 32) * 32 + ci % 32] = k[(((size_t)ky * 3 + kx) * Cout + co) * Cin + ci];
}

qmri::ConvKArgs conv_args(const ConvLayer &L, const void *x, long long ldx, int xoff, int B, int H, int W,
                          void *y, long long ldy, int yoff, int Ho, int Wo, int sy, int sx, int py, int px) {
    qmri::ConvKArgs k;
    std::memset(&k, 0, sizeof(k));
    k.x = x;
    k.ldx = ldx;
    k.xoff = xoff;
    k.B = B;
    k.H = H;
    k.W = W;
    k.Cin = L.Cin;
    k.Cout = L.Cout;
    k.ntaps = L.ntaps;
    k.deconv = L.deconv;
    k.taps = 0;
    for (int t = 0; t < L.ntaps; ++t)
        k.taps |= (unsigned long long)((L.dy[t] + 1) | ((L.dx[t] + 1) << 2)) << (4 * t);
    k.w_hi = L.w_hi.as<__bf16>();
    k.w_lo = L.w_lo.as<__bf16>();
    k.winv = 1.f;
    k.bias = L.bias.as<float>();
    k.scale = L.has_affine ? L.scale.as<float>() : nullptr;
    k.shift = L.has_affine ? L.shift.as<float>() : nullptr;
    k.relu = L.relu;
    k.y = y;
    k.ldy = ldy;
    k.yoff = yoff;
    k.Ho = Ho;
    k.Wo = Wo;
    k.sy = sy;
    k.sx = sx;
    k.py = py;
    k.px = px;
    return k;
}

bool s3_width_ok(int W) { return W % 32 == 0 || W + 2 <= 50; }  // what conv_s3_kernel tiles (unet_s3.hip: conv_s3_supported)
bool c4_ragged(int W) { return W % 32 != 0 && W + 2 > 50; }    // what only conv_c4_kernel / deconv_d4_kernel tile: image tiles with a ragged last column

struct Unet {
    int depth = 0, ncls = 0, H = 0, W = 0, maxB = 0, device = 0, split3 = 1, num_cu = 256;
    std::string trace;  // kernel family of every layer of the last forward batch (tests assert the dispatch)
    // QMRI_UNET_CHECKSUMS=1 (debugging aid, round 6's race hunt): after every layer of every pass an exact checksum (sum of the
    // output buffer's 32-bit words in 64 bits) is queued; qmri_unet2d_trace then appends "#<pass>.<layer>=<hex>;" entries of the
    // last forward, so two runs can be compared layer by layer (scripts/unet_layer_bisect.py)
    DevBuf csum_dev;
    std::vector<std::unique_ptr<DevBuf>> keep_bufs;  // QMRI_UNET_KEEP=<layer>: a copy of that layer's output buffer per pass (qmri_debug_unet_keep)
    std::vector<long long> keep_bytes;
    std::vector<std::string> csum_names;
    std::string csum_log;
    int csum_pass = 0;
    std::vector<int> Hl, Wl;   // image size at level l
    std::vector<int> fac;      // pooling / unpooling factor between level l and l + 1: 2, or 3 where the height is odd
    std::vector<std::unique_ptr<ConvLayer>> updec3;  // [level * 9 + phase]: stride-3 transposed convolution, one tap per phase
    std::vector<int> nf;
    DevBuf c1_w, c1_b;  // first layer fp32: [9][nf0], [nf0]
    DevBuf c1_img;      // the same as an MFMA A operand (unet_enc0.hip): [64 lanes][hi 8 | lo 8] fp16 of 2^k * (9 taps, bias, 0 ..)
    float c1_winv = 1.f;
    std::vector<std::unique_ptr<ConvLayer>> down1, down2, up1, up2;  // index by level
    std::vector<std::unique_ptr<ConvLayer>> updec;                   // [level]: fused 4-phase transposed conv
    std::vector<std::unique_ptr<ConvLayer>> updec_ph;                // [level * 4 + phase]: the same as four strided-output convolutions (optional)
    unsigned split_levels = 0;                                       // bit l: level l runs the four per-phase launches
    DevBuf head_w, head_b;
    // activations (fp32 NHWC), index by level
    DevBuf in;
    std::vector<std::unique_ptr<DevBuf>> tmp, cat, pool, upout;
    DevBuf bottom, stats, vol, logits, mask;
    DevBuf vol_in, mask_all, mask_planes;  // whole-volume staging of qmri_unet2d_segment_volume
    long long vol_cap = 0, seg_cap = 0;
    // ---- fp16 range of the split layout (parity mode) ----
    // A feature-map value beyond 65504 cannot be stored as fp16 hi + lo parts (v_cvt_pkrtz clamps): IWOAIOAIUnet2D feeds raw
    // 12 / 16-bit intensities (oaiunet2d.py:322-323).  Every kernel that writes the layout sets `sat` when it meets such a
    // value, and the forward is then REPEATED with the whole network scaled by 2^-act_shift: input * 2^-S, every bias and
    // BatchNorm shift * 2^-S, classifier weights * 2^S.  ReLU, max-pooling, convolution and the BatchNorm scale commute with
    // a positive factor, so in exact arithmetic every activation is 2^-S times what it was and the logits are unchanged, while the
    // stored values move back into range.  In the fp16 hi + lo layout the scaled network is NOT bit-identical for small
    // activations (their lo parts reach the fp16 subnormal range after the scaling and lose bits): what is tested, and claimed,
    // is the 1e-3 logit bound of the parity mode (tests/test_unet_gpu.py::test_raw_intensity_input_runs_in_range).
    DevBuf sat;                       // int flag (device)
    int act_shift = 0;                // S the device-side parameters currently carry
    int act_bump = 0;                 // what saturating forwards have added to the exponent chosen from the input so far
    std::vector<float> c1_k, c1_b_h, head_w_h;  // host copies: first-layer kernel [9][C] + bias [C], classifier weights [C][ncls]
    int *sat_ptr() const { return sat.as<int>(); }
    int set_act_shift(int S);
};

// (re)upload every additive parameter for activation exponent S (see Unet::sat)
int Unet::set_act_shift(int S) {
    if (S == act_shift) return QMRI_OK;
    const float down = std::ldexp(1.f, -S), up = std::ldexp(1.f, S);
    std::vector<float> t;
    auto put = [&](DevBuf &dst, const std::vector<float> &src, float f) -> hipError_t {
        if (!dst.p || src.empty()) return hipSuccess;
        t.resize(src.size());
        for (size_t i = 0; i < src.size(); ++i) t[i] = src[i] * f;
        return hipMemcpy(dst.p, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    };
    auto layer = [&](ConvLayer *L) -> hipError_t {
        if (!L) return hipSuccess;
        hipError_t e = put(L->bias, L->bias_h, down);
        if (e == hipSuccess && L->has_affine) e = put(L->shift, L->shift_h, down);
        return e;
    };
    for (auto *vec : {&down1, &down2, &up1, &up2, &updec, &updec_ph, &updec3})
        for (auto &L : *vec) U_TRY(layer(L.get()));
    U_TRY(put(c1_b, c1_b_h, down));
    U_TRY(put(head_w, head_w_h, up));
    if (c1_img.p) {  // the fused first block's MFMA operand carries the bias on its constant-one tap
        const int C = nf[0];
        std::vector<float> all(c1_k);
        for (float b : c1_b_h) all.push_back(b * down);
        const int sh = weight_shift(all);
        c1_winv = std::ldexp(1.f, -sh);
        std::vector<unsigned short> img(64 * 16, 0);
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 8; ++i) {
                const int ch = lane & 31, k = (lane >> 5) * 8 + i;
                const float v = k < 9 ? c1_k[k * C + ch] : k == 9 ? c1_b_h[ch] * down : 0.f;
                split_f16_host(std::ldexp(v, sh), img[lane * 16 + i], img[lane * 16 + 8 + i]);
            }
        U_TRY(hipMemcpy(c1_img.p, img.data(), img.size() * 2, hipMemcpyHostToDevice));
    }
    act_shift = S;
    return QMRI_OK;
}

}  // namespace

extern "C" {

int qmri_unet2d_destroy(void *handle) {
    delete static_cast<Unet *>(handle);
    return QMRI_OK;
}

int qmri_unet2d_create(const qmri_unet2d_desc *d, void **handle) {
    if (!d || !handle) return ufail(QMRI_ERR_ARG, "desc / handle is NULL");
    *handle = nullptr;
    if (d->depth < 2 || d->depth > 7) return ufail(QMRI_ERR_ARG, "depth must be in [2, 7]");
    if (d->base_features < 32 || d->base_features % 32)
        return ufail(QMRI_ERR_UNSUPPORTED, "base_features must be a multiple of 32 (MFMA K tile)");
    if (d->n_classes < 1 || d->n_classes > 4) return ufail(QMRI_ERR_UNSUPPORTED, "n_classes must be 1..4");
    if (d->H <= 0 || d->W <= 0) return ufail(QMRI_ERR_ARG, "H and W must be positive");
    std::vector<int> Hl(1, d->H), Wl(1, d->W), fac;
    for (int l = 0; l + 1 < d->depth; ++l) {
        // oaiunet2d.py:234-243: MaxPooling2D((2, 2)) where the height is even, (3, 3) where it is odd -- on BOTH axes; the
        // Concatenate with the (strides = pool size) Conv2DTranspose on the way up (:250-264) then needs both axes to divide
        const int f = Hl[l] % 2 == 0 ? 2 : 3;
        if (Hl[l] % f || Wl[l] % f)
            return ufail(QMRI_ERR_ARG,
                         "the reference's graph does not build for %d x %d slices: level %d is %d x %d and is pooled by %d "
                         "(height even -> 2, odd -> 3; oaiunet2d.py:234-264), which must divide both", d->H, d->W, l, Hl[l], Wl[l], f);
        fac.push_back(f);
        Hl.push_back(Hl[l] / f);
        Wl.push_back(Wl[l] / f);
    }
    if (d->max_batch < 1) return ufail(QMRI_ERR_ARG, "max_batch must be >= 1");
    const int expect = d->depth * 8 + (d->depth - 1) * 10 + 2;
    if (d->n_tensors != expect || !d->tensors)
        return ufail(QMRI_ERR_ARG, "expected %d weight tensors in Keras layer order, got %d", expect, d->n_tensors);
    for (int i = 0; i < expect; ++i)
        if (!d->tensors[i]) return ufail(QMRI_ERR_ARG, "weight tensor %d is NULL", i);
    if (d->base_features << (d->depth - 1) > 4096) return ufail(QMRI_ERR_UNSUPPORTED, "too many features");
    U_TRY(hipSetDevice(d->device));

    std::unique_ptr<Unet> U(new Unet);
    U->depth = d->depth;
    U->ncls = d->n_classes;
    U->H = d->H;
    U->W = d->W;
    U->Hl = Hl;
    U->Wl = Wl;
    U->fac = fac;
    U->maxB = d->max_batch;
    U->device = d->device;
    U->split3 = d->precision != 0;
    {
        hipDeviceProp_t prop;
        U_TRY(hipGetDeviceProperties(&prop, d->device));
        U->num_cu = prop.multiProcessorCount;
    }
    for (int l = 0; l < d->depth; ++l) U->nf.push_back(d->base_features << l);
    if (U->nf[0] > 256) return ufail(QMRI_ERR_UNSUPPORTED, "base_features > 256");
    const double eps = d->bn_eps > 0 ? d->bn_eps : 1e-3;

    const float *const *T = d->tensors;
    int ti = 0;
    auto fold_bn = [&](int C, std::vector<float> &sc, std::vector<float> &sh) {
        const float *gamma = T[ti], *beta = T[ti + 1], *mean = T[ti + 2], *var = T[ti + 3];
        ti += 4;
        sc.resize(C);
        sh.resize(C);
        for (int c = 0; c < C; ++c) {
            const double s = (double)gamma[c] / std::sqrt((double)var[c] + eps);
            sc[c] = (float)s;
            sh[c] = (float)((double)beta[c] - (double)mean[c] * s);
        }
    };
    U->down1.resize(d->depth);
    U->down2.resize(d->depth);
    U->up1.resize(d->depth);
    U->up2.resize(d->depth);
    U->updec.resize((size_t)d->depth);
    U->updec_ph.resize((size_t)d->depth * 4);
    U->updec3.resize((size_t)d->depth * 9);
    {
        // QMRI_DECONV_SPLIT: bit mask of levels whose transposed convolution runs as four per-phase convolutions
        const char *e = std::getenv("QMRI_DECONV_SPLIT");
        U->split_levels = e ? (unsigned)std::strtoul(e, nullptr, 0) : 0u;
    }
    std::vector<float> wk, sc, sh;
    for (int l = 0; l < d->depth; ++l) {
        const int C = U->nf[l];
        const int Cin = l == 0 ? 1 : U->nf[l - 1];
        const float *k1 = T[ti], *b1 = T[ti + 1], *k2 = T[ti + 2], *b2 = T[ti + 3];
        ti += 4;
        if (l == 0) {
            // first layer: Keras (3,3,1,C) == [9][C]
            U_TRY(U->c1_w.alloc((size_t)9 * C * 4));
            U_TRY(hipMemcpy(U->c1_w.p, k1, (size_t)9 * C * 4, hipMemcpyHostToDevice));
            U_TRY(U->c1_b.alloc((size_t)C * 4));
            U_TRY(hipMemcpy(U->c1_b.p, b1, (size_t)C * 4, hipMemcpyHostToDevice));
            U->c1_k.assign(k1, k1 + 9 * C);
            U->c1_b_h.assign(b1, b1 + C);
            if (C == 32) {  // K = 16 operand of the fused first block: taps 0..8, the bias on a constant-one tap, zeros
                std::vector<float> all(k1, k1 + 9 * C);
                all.insert(all.end(), b1, b1 + C);
                const int sh = weight_shift(all);
                U->c1_winv = std::ldexp(1.f, -sh);
                std::vector<unsigned short> img(64 * 16, 0);
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int ch = lane & 31, k = (lane >> 5) * 8 + i;
                        const float v = k < 9 ? k1[k * C + ch] : k == 9 ? b1[ch] : 0.f;
                        split_f16_host(std::ldexp(v, sh), img[lane * 16 + i], img[lane * 16 + 8 + i]);
                    }
                U_TRY(U->c1_img.alloc(img.size() * 2));
                U_TRY(hipMemcpy(U->c1_img.p, img.data(), img.size() * 2, hipMemcpyHostToDevice));
            }
        } else {
            U->down1[l].reset(new ConvLayer);
            U->down1[l]->relu = 1;
            pack_conv3x3(k1, Cin, C, *U->down1[l], wk);
            U_TRY(U->down1[l]->upload(wk, b1, nullptr, nullptr));
            U_TRY(U->down1[l]->upload_parity(wk, s3_width_ok(U->Wl[l]), c4_ragged(U->Wl[l])));
            if (s3_width_ok(U->Wl[l])) U_TRY(U->down1[l]->upload_s3_bf16(wk));
        }
        fold_bn(C, sc, sh);
        U->down2[l].reset(new ConvLayer);
        U->down2[l]->relu = 1;
        pack_conv3x3(k2, C, C, *U->down2[l], wk);
        U_TRY(U->down2[l]->upload(wk, b2, &sc, &sh));
        U_TRY(U->down2[l]->upload_parity(wk, s3_width_ok(U->Wl[l]), c4_ragged(U->Wl[l])));
        if (s3_width_ok(U->Wl[l])) U_TRY(U->down2[l]->upload_s3_bf16(wk));
    }
    for (int l = d->depth - 2; l >= 0; --l) {
        const int C = U->nf[l], Cup = U->nf[l + 1];
        const float *kd = T[ti], *bd = T[ti + 1], *k1 = T[ti + 2], *b1 = T[ti + 3], *k2 = T[ti + 4], *b2 = T[ti + 5];
        ti += 6;
        {
            auto &L = U->updec[(size_t)l];
            L.reset(new ConvLayer);
            L->relu = 0;
            pack_deconv_fused(kd, Cup, C, *L, wk);
            U_TRY(L->upload(wk, bd, nullptr, nullptr));
            U_TRY(L->upload_parity(wk, U->fac[l] == 2 && s3_width_ok(U->Wl[l + 1]), U->fac[l] == 2 && c4_ragged(U->Wl[l + 1])));  // tiles of the INPUT grid (level l + 1)
            if (U->fac[l] == 2 && s3_width_ok(U->Wl[l + 1])) U_TRY(L->upload_s3_bf16(wk));
            if (U->fac[l] == 3)
                for (int ph = 0; ph < 9; ++ph) {
                    auto &P = U->updec3[(size_t)l * 9 + ph];
                    P.reset(new ConvLayer);
                    P->relu = 0;
                    pack_deconv3_phase(kd, Cup, C, ph / 3, ph % 3, *P, wk);
                    U_TRY(P->upload(wk, bd, nullptr, nullptr));
                    U_TRY(P->upload_parity(wk, false));
                }
            if (U->split_levels >> l & 1u)
                for (int ph = 0; ph < 4; ++ph) {
                    auto &P = U->updec_ph[(size_t)l * 4 + ph];
                    P.reset(new ConvLayer);
                    P->relu = 0;
                    pack_deconv_phase(kd, Cup, C, ph >> 1, ph & 1, *P, wk);
                    U_TRY(P->upload(wk, bd, nullptr, nullptr));
                    U_TRY(P->upload_parity(wk, false));
                }
        }
        U->up1[l].reset(new ConvLayer);
        U->up1[l]->relu = 1;
        pack_conv3x3(k1, 2 * C, C, *U->up1[l], wk);
        U_TRY(U->up1[l]->upload(wk, b1, nullptr, nullptr));
        U_TRY(U->up1[l]->upload_parity(wk, s3_width_ok(U->Wl[l]), c4_ragged(U->Wl[l])));
        if (s3_width_ok(U->Wl[l])) U_TRY(U->up1[l]->upload_s3_bf16(wk));
        fold_bn(C, sc, sh);
        U->up2[l].reset(new ConvLayer);
        U->up2[l]->relu = 1;
        pack_conv3x3(k2, C, C, *U->up2[l], wk);
        U_TRY(U->up2[l]->upload(wk, b2, &sc, &sh));
        U_TRY(U->up2[l]->upload_parity(wk, s3_width_ok(U->Wl[l]), c4_ragged(U->Wl[l])));
        if (s3_width_ok(U->Wl[l])) U_TRY(U->up2[l]->upload_s3_bf16(wk));
    }
    // head: Keras (1,1,C0,NC) == [C0][NC]
    U_TRY(U->head_w.alloc((size_t)U->nf[0] * U->ncls * 4));
    U_TRY(hipMemcpy(U->head_w.p, T[ti], (size_t)U->nf[0] * U->ncls * 4, hipMemcpyHostToDevice));
    U_TRY(U->head_b.alloc((size_t)U->ncls * 4));
    U_TRY(hipMemcpy(U->head_b.p, T[ti + 1], (size_t)U->ncls * 4, hipMemcpyHostToDevice));
    U->head_w_h.assign(T[ti], T[ti] + (size_t)U->nf[0] * U->ncls);
    U_TRY(U->sat.alloc(2 * sizeof(int)));  // [0] saturation flag, [1] max |input| (bit pattern)
    U_TRY(hipMemset(U->sat.p, 0, 2 * sizeof(int)));

    // activation buffers for max_batch slices
    const long long B = U->maxB;
    U_TRY(U->in.alloc((size_t)B * U->H * U->W * 4));
    U->tmp.resize(d->depth);
    U->cat.resize(d->depth);
    U->pool.resize(d->depth);
    U->upout.resize(d->depth);
    for (int l = 0; l < d->depth; ++l) {
        const long long pix = B * U->Hl[l] * U->Wl[l];
        U->tmp[l].reset(new DevBuf);
        U_TRY(U->tmp[l]->alloc((size_t)pix * U->nf[l] * 4));
        if (l < d->depth - 1) {
            U->cat[l].reset(new DevBuf);
            U_TRY(U->cat[l]->alloc((size_t)pix * 2 * U->nf[l] * 4));
            U->upout[l].reset(new DevBuf);
            U_TRY(U->upout[l]->alloc((size_t)pix * U->nf[l] * 4));
        }
        if (l > 0) {
            U->pool[l].reset(new DevBuf);
            U_TRY(U->pool[l]->alloc((size_t)pix * U->nf[l - 1] * 4));
        }
    }
    {
        const int l = d->depth - 1;
        U_TRY(U->bottom.alloc((size_t)B * U->Hl[l] * U->Wl[l] * U->nf[l] * 4));
    }
    U_TRY(U->stats.alloc((size_t)qmri::whiten_stats_doubles() * sizeof(double)));
    *handle = U.release();
    return QMRI_OK;
}

int qmri_unet2d_trace(void *handle, char *buf, int32_t size) {
    if (!handle || !buf || size <= 0) return ufail(QMRI_ERR_ARG, "handle / buf is NULL");
    const std::string t = static_cast<Unet *>(handle)->trace + static_cast<Unet *>(handle)->csum_log;
    std::snprintf(buf, (size_t)size, "%s", t.c_str());
    return (int)t.size();
}

// debugging aid (QMRI_UNET_CHECKSUMS=1 QMRI_UNET_KEEP=<layer>[,<layer>...]): the kept copy of a layer's output buffer of the last forward; `pass` = 64 * (index of
// the layer in the list) + pass
// -> host; returns the bytes the buffer holds (copies min(that, nbytes)), or a negative error.  Not part of include/qmri.h.
long long qmri_debug_unet_keep(void *handle, int32_t pass, void *host, long long nbytes) {
    if (!handle) return QMRI_ERR_ARG;
    Unet *U = static_cast<Unet *>(handle);
    if (pass < 0 || (size_t)pass >= U->keep_bufs.size() || !U->keep_bufs[(size_t)pass]) return QMRI_ERR_ARG;
    if (hipSetDevice(U->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return QMRI_ERR_HIP;
    const long long have = U->keep_bytes[(size_t)pass];
    if (host && nbytes > 0 &&
        hipMemcpy(host, U->keep_bufs[(size_t)pass]->p, (size_t)(have < nbytes ? have : nbytes), hipMemcpyDeviceToHost) != hipSuccess)
        return QMRI_ERR_HIP;
    return have;
}

int qmri_unet2d_set_precision(void *handle, int32_t precision) {
    if (!handle) return ufail(QMRI_ERR_ARG, "handle is NULL");
    static_cast<Unet *>(handle)->split3 = precision != 0;
    return QMRI_OK;
}

// ---- debugging aid: exact per-layer checksums (QMRI_UNET_CHECKSUMS=1) ----
static bool csum_wanted() {
    static const bool on = std::getenv("QMRI_UNET_CHECKSUMS") && std::atoi(std::getenv("QMRI_UNET_CHECKSUMS")) != 0;
    return on;
}
constexpr int kCsumSlots = 4096;
__global__ __launch_bounds__(256) void csum_kernel(const unsigned int *__restrict__ x, long long nwords, unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long long)gridDim.x * blockDim.x) acc += x[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);  // (integer addition: exact in any order)
}
static int csum(Unet *U, const char *name, const void *p, long long bytes, hipStream_t st) {
    if (!csum_wanted() || !p || bytes <= 0) return QMRI_OK;
    if (!U->csum_dev.p) {
        U_TRY(U->csum_dev.alloc((size_t)kCsumSlots * 8));
        U_TRY(hipMemsetAsync(U->csum_dev.p, 0, (size_t)kCsumSlots * 8, st));
    }
    if ((int)U->csum_names.size() >= kCsumSlots) return QMRI_OK;
    const long long nwords = bytes / 4;
    long long blocks = (nwords + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    (void)hipGetLastError();
    hipLaunchKernelGGL(csum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const unsigned int *>(p), nwords,
                       U->csum_dev.as<unsigned long long>() + U->csum_names.size());
    U_TRY(hipGetLastError());
    U->csum_names.push_back("#" + std::to_string(U->csum_pass) + "." + name);
    // QMRI_UNET_KEEP=<layer>[,<layer>...]: the bytes themselves, per (layer, pass) -- slot = 64 * index in the list + pass
    static const std::vector<std::string> keep_names = [] {
        std::vector<std::string> v;
        if (const char *e = std::getenv("QMRI_UNET_KEEP")) {
            std::string t(e);
            size_t a = 0;
            while (a <= t.size()) {
                const size_t b = t.find(',', a);
                v.push_back(t.substr(a, b == std::string::npos ? std::string::npos : b - a));
                if (b == std::string::npos) break;
                a = b + 1;
            }
        }
        return v;
    }();
    for (size_t ki = 0; ki < keep_names.size(); ++ki) {
        if (keep_names[ki] != name || U->csum_pass >= 64) continue;
        const size_t ps = ki * 64 + (size_t)U->csum_pass;
        if (U->keep_bufs.size() <= ps) {
            U->keep_bufs.resize(ps + 1);
            U->keep_bytes.resize(ps + 1, 0);
        }
        if (!U->keep_bufs[ps] || U->keep_bytes[ps] < bytes) {
            U->keep_bufs[ps].reset(new DevBuf);
            U_TRY(U->keep_bufs[ps]->alloc((size_t)bytes));
        }
        U->keep_bytes[ps] = bytes;
        U_TRY(hipMemcpyAsync(U->keep_bufs[ps]->p, p, (size_t)bytes, hipMemcpyDeviceToDevice, st));
    }
    return QMRI_OK;
}
// (start of a forward: forget the previous one's; end: read the sums back)
static int csum_begin(Unet *U, hipStream_t st) {
    if (!csum_wanted()) return QMRI_OK;
    U->csum_names.clear();
    U->csum_log.clear();
    U->csum_pass = 0;
    if (U->csum_dev.p) U_TRY(hipMemsetAsync(U->csum_dev.p, 0, (size_t)kCsumSlots * 8, st));
    return QMRI_OK;
}
static int csum_end(Unet *U, hipStream_t st) {
    if (!csum_wanted() || U->csum_names.empty()) return QMRI_OK;
    std::vector<unsigned long long> h(U->csum_names.size());
    U_TRY(hipMemcpyAsync(h.data(), U->csum_dev.p, h.size() * 8, hipMemcpyDeviceToHost, st));
    U_TRY(hipStreamSynchronize(st));
    char b[40];
    for (size_t i = 0; i < h.size(); ++i) {
        std::snprintf(b, sizeof(b), "=%016llx;", h[i]);
        U->csum_log += U->csum_names[i] + b;
    }
    return QMRI_OK;
}

// ---- parity mode ("fp16x3"): activations in the split layout (qmri_internal.h: ConvS3Args), 3x3 convolutions on
// conv_s3_kernel where it tiles the level (W % 32 == 0 or W <= 48), otherwise -- and for the transposed convolutions --
// on the general kernel reading / writing the same layout.
static int conv3x3_parity(Unet *U, const char *name, const ConvLayer &L, const void *x, long long ldx, int xoff, int Bt, int H,
                          int W, void *y, long long ldy, int yoff, void *pool_y, int pool_ld, bool head, float *logits,
                          unsigned char *mask, hipStream_t st) {
    char buf[96];
    // (a level conv_s3_kernel does not tile -- W % 32 != 0, wider than the flattened tiling -- has w_c4 only: conv_c4_kernel runs it on image
    //  tiles with a ragged last column, or, with QMRI_C4 = 0 / a layer that kernel does not take, the general kernel below)
    if (L.w_s3.p || L.w_c4.p) do {
        qmri::ConvS3Args k;
        std::memset(&k, 0, sizeof(k));
        k.x = x; k.ldx = ldx; k.xoff = xoff;
        k.B = Bt; k.H = H; k.W = W;
        k.Cin = L.Cin; k.Cout = L.Cout;
        k.w = L.w_s3.p;
        k.w_c4 = L.w_c4.p;
        k.winv = L.winv;
        k.bias = L.bias.as<float>();
        k.scale = L.has_affine ? L.scale.as<float>() : nullptr;
        k.shift = L.has_affine ? L.shift.as<float>() : nullptr;
        k.relu = L.relu;
        k.y = y; k.ldy = ldy; k.yoff = yoff;
        k.sat = U->sat_ptr();
        const int bn = qmri::conv_s3_block_channels(L.Cout, 0);
        const bool flat = qmri::conv_tiles_flat(W);
        const bool fuse_pool = pool_y && !flat && bn >= 64 && !(H & 1) && !(W & 1);
        if (fuse_pool) { k.pool_y = pool_y; k.pool_ld = pool_ld; }
        const bool fuse_head = head && W % 32 == 0 && L.Cout == 32;
        if (fuse_head) {
            k.y = nullptr;  // the last feature map is only consumed by the head: never written to HBM
            k.head_w = U->head_w.as<float>(); k.head_b = U->head_b.as<float>(); k.head_nc = U->ncls;
            k.logits = logits; k.mask = mask;
        }
        const bool c4 = qmri::conv_s3_takes_c4(k, U->num_cu);  // (the launcher's own choice: one wave per SIMD, 128 x 128 register tiles)
        if (!L.w_s3.p && !c4) break;                           // (ragged level, not on conv_c4_kernel: the general kernel)
        U_TRY(qmri::conv_s3_launch(k, U->num_cu, st));
        if (c4)
            snprintf(buf, sizeof(buf), "%s:s3/%s/c4x%d%s;", name, flat ? "flat" : "2d", qmri::conv_c4_block_channels(L.Cout), fuse_pool ? "+pool" : "");
        else
            snprintf(buf, sizeof(buf), "%s:s3/%s/bn%d%s%s;", name, flat ? "flat" : "2d", bn, fuse_pool ? "+pool" : "", fuse_head ? "+head" : "");
        U->trace += buf;
        if (pool_y && !fuse_pool) {
            U_TRY(qmri::maxpool2_split_launch(y, ldy, yoff, Bt, H, W, L.Cout, pool_y, st));
            U->trace += "pool:split;";
        }
        if (head && !fuse_head) {
            U_TRY(qmri::head_split_launch(y, (long long)Bt * H * W, L.Cout, U->head_w.as<float>(), U->head_b.as<float>(), U->ncls,
                                          logits, mask, st));
            U->trace += "head:split;";
        }
        {
            int rc_ = csum(U, name, k.y ? y : (const void *)logits, k.y ? (long long)Bt * H * W * ldy * 4 : (long long)Bt * H * W * U->ncls * 4, st);
            if (rc_ == QMRI_OK && pool_y) rc_ = csum(U, (std::string(name) + ".pool").c_str(), pool_y, (long long)Bt * (H / 2) * (W / 2) * pool_ld * 4, st);
            if (rc_ != QMRI_OK) return rc_;
        }
        return QMRI_OK;
    } while (false);
    auto k = conv_args(L, x, ldx, xoff, Bt, H, W, y, ldy, yoff, H, W, 1, 1, 0, 0);
    k.w_hi = L.h_hi.as<__bf16>();
    k.w_lo = L.h_lo.as<__bf16>();
    k.winv = L.winv;
    k.sat = U->sat_ptr();
    if (pool_y && !((H | W) & 1)) { k.pool_y = pool_y; k.pool_ld = pool_ld; }
    U_TRY(qmri::conv_igemm_launch(k, 1, st));
    snprintf(buf, sizeof(buf), "%s:igemm%s;", name, k.pool_y ? "+pool" : "");
    U->trace += buf;
    if (head) {
        U_TRY(qmri::head_split_launch(y, (long long)Bt * H * W, L.Cout, U->head_w.as<float>(), U->head_b.as<float>(), U->ncls, logits,
                                      mask, st));
        U->trace += "head:split;";
    }
    {
        int rc_ = csum(U, name, y, (long long)Bt * H * W * ldy * 4, st);
        if (rc_ == QMRI_OK && pool_y) rc_ = csum(U, (std::string(name) + ".pool").c_str(), pool_y, (long long)Bt * (H / 2) * (W / 2) * pool_ld * 4, st);
        if (rc_ != QMRI_OK) return rc_;
    }
    return QMRI_OK;
}

// Plain-bf16 layer on conv_s3_kernel<.., ONE> where the layer has 64-channel chunks and the level tiles (QMRI_S3_B16=0: the
// round-1 kernels everywhere).  `k` is the layer as conv_igemm_launch would take it (bf16 NHWC in / out, optional fused pool).
static bool s3_b16_wanted() {
    static const bool on = !(std::getenv("QMRI_S3_B16") && std::atoi(std::getenv("QMRI_S3_B16")) == 0);
    return on;
}
static int conv_bf16(Unet *U, const char *name, const ConvLayer &L, const qmri::ConvKArgs &k, hipStream_t st) {
    char buf[96];
    const int Wt = k.W;  // the grid the kernel tiles: the input grid (for the transposed convolution too)
    // Measured per layer (160 slices of 384 x 384, scripts/unet_ab.sh QMRI_S3_B16 0 1): every transposed convolution gains
    // (651 / 519 / 488 / 565 / 758 -> 306 / 290 / 329 / 376 / 609 us) and so do the convolutions with >= 128 input channels
    // (+5 ... +25 %); with 64 input channels a tile is 9 steps of 2 MFMAs per wave-tile and the round-1 kernels (register
    // weights for <= 64 output channels, implicit GEMM otherwise) stay ahead (516 vs 694, 1183 vs 1449, 262 vs 279 us).
    const bool ok = s3_b16_wanted() && L.w_s3b.p && (L.deconv || L.Cin >= 128) && s3_width_ok(Wt) && !k.head_w && !k.c1_x && (k.ldx % 2 == 0) && (k.xoff % 2 == 0) &&
                    (!k.deconv || (k.sy == 2 && k.sx == 2 && k.py == 0 && k.px == 0)) && (k.deconv || (k.sy == 1 && k.sx == 1));
    if (!ok) {
        U_TRY(qmri::conv_igemm_launch(k, 0, st));
        snprintf(buf, sizeof(buf), "%s:igemm16;", name);
        U->trace += buf;
        return QMRI_OK;
    }
    qmri::ConvS3Args a;
    std::memset(&a, 0, sizeof(a));
    a.one = 1;
    a.x = k.x; a.ldx = k.ldx / 2; a.xoff = k.xoff / 2;
    a.B = k.B; a.H = k.H; a.W = k.W;
    a.Cin = L.Cin / 2; a.Cout = L.Cout;
    a.deconv = L.deconv;
    a.w = L.w_s3b.p; a.winv = 1.f;
    a.bias = k.bias; a.scale = k.scale; a.shift = k.shift; a.relu = k.relu;
    a.y = k.y; a.ldy = k.ldy; a.yoff = k.yoff;
    const int bn = qmri::conv_s3_block_channels(L.Cout, L.deconv);
    const bool flat = Wt % 32 != 0;
    const bool fuse_pool = k.pool_y && !flat && bn >= 64 && !(k.H & 1) && !L.deconv;
    if (fuse_pool) { a.pool_y = k.pool_y; a.pool_ld = k.pool_ld; }
    U_TRY(qmri::conv_s3_launch(a, U->num_cu, st));
    snprintf(buf, sizeof(buf), "%s:s3b/%s/bn%d%s;", name, flat ? "flat" : "2d", bn, fuse_pool ? "+pool" : "");
    U->trace += buf;
    if (k.pool_y && !fuse_pool) {
        U_TRY(qmri::maxpool2_launch(k.y, k.ldy, k.yoff, k.B, k.H, k.W, L.Cout, k.pool_y, 1, st));
        U->trace += "pool:b16;";
    }
    return QMRI_OK;
}

static int forward_batch_parity(Unet *U, int Bt, float *logits, unsigned char *mask, hipStream_t st) {
    const int D = U->depth;
    char nm[64];
    for (int l = 0; l < D; ++l) {
        const int H = U->Hl[l], W = U->Wl[l], C = U->nf[l];
        void *t1 = U->tmp[l]->p;
        // the whole first block in one kernel (QMRI_ENC0=0: the three-kernel route, kept for A/B runs and odd sizes)
        static const bool want_enc0 = !(std::getenv("QMRI_ENC0") && std::atoi(std::getenv("QMRI_ENC0")) == 0);
        if (l == 0 && want_enc0 && D > 1 && C == 32 && U->fac[0] == 2 && U->c1_img.p && U->down2[0]->w_s3.p && H % 8 == 0 && W % 32 == 0) {
            const ConvLayer &L2 = *U->down2[0];
            qmri::Enc0Args k;
            std::memset(&k, 0, sizeof(k));
            k.x = U->in.as<float>(); k.B = Bt; k.H = H; k.W = W;
            k.c1_img = U->c1_img.p; k.c1_winv = U->c1_winv;
            k.w2 = L2.w_s3.p; k.winv2 = L2.winv;
            k.bias2 = L2.bias.as<float>(); k.scale2 = L2.scale.as<float>(); k.shift2 = L2.shift.as<float>();
            k.y = U->cat[0]->p; k.ldy = 2 * C; k.yoff = C;
            k.pool_y = U->pool[1]->p; k.pool_ld = C;
            k.sat = U->sat_ptr();
            U_TRY(qmri::enc0_launch(k, U->num_cu, st));
            U->trace += "down0:enc0;";
            if (csum(U, "down0", k.y, (long long)Bt * H * W * 2 * C * 4, st) != QMRI_OK || csum(U, "down0.pool", k.pool_y, (long long)Bt * (H / 2) * (W / 2) * C * 4, st) != QMRI_OK) return QMRI_ERR_HIP;
            continue;
        }
        if (l == 0) {
            U_TRY(qmri::c1_split_launch(U->in.as<float>(), Bt, H, W, U->c1_w.as<float>(), U->c1_b.as<float>(), C, t1, C, 0, U->sat_ptr(), st));
            U->trace += "down0.conv1:c1/split;";
        } else {
            snprintf(nm, sizeof(nm), "down%d.conv1", l);
            const int rc = conv3x3_parity(U, nm, *U->down1[l], U->pool[l]->p, U->nf[l - 1], 0, Bt, H, W, t1, C, 0, nullptr, 0, false,
                                          nullptr, nullptr, st);
            if (rc != QMRI_OK) return rc;
        }
        snprintf(nm, sizeof(nm), "down%d.conv2", l);
        int rc;
        if (l < D - 1 && U->fac[l] == 3) {  // odd height: MaxPooling2D((3, 3)) as its own kernel
            rc = conv3x3_parity(U, nm, *U->down2[l], t1, C, 0, Bt, H, W, U->cat[l]->p, 2 * C, C, nullptr, 0, false, nullptr, nullptr, st);
            if (rc == QMRI_OK) {
                U_TRY(qmri::maxpoolk_split_launch(U->cat[l]->p, 2 * C, C, Bt, H, W, C, 3, U->pool[l + 1]->p, st));
                U->trace += "pool3:split;";
            }
        } else if (l < D - 1)  // block output (post-BN) = the skip = 2nd half of the level's concat buffer; pooled copy -> next level
            rc = conv3x3_parity(U, nm, *U->down2[l], t1, C, 0, Bt, H, W, U->cat[l]->p, 2 * C, C, U->pool[l + 1]->p, C, false, nullptr,
                                nullptr, st);
        else
            rc = conv3x3_parity(U, nm, *U->down2[l], t1, C, 0, Bt, H, W, U->bottom.p, C, 0, nullptr, 0, false, nullptr, nullptr, st);
        if (rc != QMRI_OK) return rc;
    }
    const void *src = U->bottom.p;
    for (int l = D - 2; l >= 0; --l) {
        const int H = U->Hl[l], W = U->Wl[l], C = U->nf[l], Cup = U->nf[l + 1];
        void *cat = U->cat[l]->p;
        if (U->fac[l] == 3) {
            for (int ph = 0; ph < 9; ++ph) {
                const ConvLayer &L = *U->updec3[(size_t)l * 9 + ph];
                auto k = conv_args(L, src, Cup, 0, Bt, U->Hl[l + 1], U->Wl[l + 1], cat, 2 * C, 0, H, W, 3, 3, ph / 3, ph % 3);
                k.w_hi = L.h_hi.as<__bf16>();
                k.w_lo = L.h_lo.as<__bf16>();
                k.winv = L.winv;
                k.sat = U->sat_ptr();
                U_TRY(qmri::conv_igemm_launch(k, 1, st));
            }
            snprintf(nm, sizeof(nm), "up%d.deconv:igemm/stride3;", l);
            U->trace += nm;
        } else {
            const ConvLayer &L = *U->updec[(size_t)l];
            qmri::ConvS3Args k;
            std::memset(&k, 0, sizeof(k));
            k.x = src; k.ldx = Cup; k.B = Bt; k.H = U->Hl[l + 1]; k.W = U->Wl[l + 1];
            k.Cin = L.Cin; k.Cout = L.Cout; k.deconv = 1;
            k.w = L.w_s3.p; k.winv = L.winv;
            k.w_c4 = L.w_c4.p;  // (a transposed convolution's w_c4 is deconv_d4_kernel's image)
            k.bias = L.bias.as<float>();
            k.y = cat; k.ldy = 2 * C; k.yoff = 0;
            k.sat = U->sat_ptr();
            const bool d4 = (L.w_s3.p || L.w_c4.p) && qmri::conv_s3_takes_d4(k);
            if (L.w_s3.p || d4) {  // (w_c4 only: an input grid only deconv_d4_kernel tiles -- image tiles with a ragged last column)
                U_TRY(qmri::conv_s3_launch(k, U->num_cu, st));
                snprintf(nm, sizeof(nm), "up%d.deconv:s3/%s/%s32;", l, qmri::conv_tiles_flat(U->Wl[l + 1]) ? "flat" : "2d", d4 ? "d4x" : "bn");
            } else {
                auto k = conv_args(L, src, Cup, 0, Bt, H / 2, W / 2, cat, 2 * C, 0, H, W, 2, 2, 0, 0);
                k.w_hi = L.h_hi.as<__bf16>();
                k.w_lo = L.h_lo.as<__bf16>();
                k.winv = L.winv;
                k.sat = U->sat_ptr();
                U_TRY(qmri::conv_igemm_launch(k, 1, st));
                snprintf(nm, sizeof(nm), "up%d.deconv:igemm;", l);
            }
            U->trace += nm;
            snprintf(nm, sizeof(nm), "up%d.deconv", l);
            if (csum(U, nm, cat, (long long)Bt * H * W * 2 * C * 4, st) != QMRI_OK) return QMRI_ERR_HIP;
        }
        void *t1 = U->tmp[l]->p;
        snprintf(nm, sizeof(nm), "up%d.conv1", l);
        int rc = QMRI_OK;
        // 64 -> 32 channels at the top level with LDS-resident weights (QMRI_MID0=0: conv_s3_kernel)
        static const bool want_mid0 = !(std::getenv("QMRI_MID0") && std::atoi(std::getenv("QMRI_MID0")) == 0);
        if (l == 0 && want_mid0 && C == 32 && U->up1[0]->w_s3.p && !U->up1[0]->has_affine && U->up1[0]->relu && H % 8 == 0 && W % 32 == 0) {
            const ConvLayer &L1 = *U->up1[0];
            qmri::Mid0Args k;
            std::memset(&k, 0, sizeof(k));
            k.x = cat; k.ldx = 2 * C; k.xoff = 0; k.B = Bt; k.H = H; k.W = W;
            k.w = L1.w_s3.p; k.winv = L1.winv; k.bias = L1.bias.as<float>();
            k.y = t1; k.ldy = C; k.yoff = 0;
            k.sat = U->sat_ptr();
            U_TRY(qmri::mid0_launch(k, U->num_cu, st));
            U->trace += "up0.conv1:mid0;";
            if (csum(U, "up0.conv1", t1, (long long)Bt * H * W * C * 4, st) != QMRI_OK) return QMRI_ERR_HIP;
        } else {
            rc = conv3x3_parity(U, nm, *U->up1[l], cat, 2 * C, 0, Bt, H, W, t1, C, 0, nullptr, 0, false, nullptr, nullptr, st);
        }
        if (rc != QMRI_OK) return rc;
        void *out = U->upout[l]->p;
        snprintf(nm, sizeof(nm), "up%d.conv2", l);
        // the last convolution + classifier as one kernel with LDS-resident weights (QMRI_OUT0=0: conv_s3_kernel + fused head)
        static const bool want_out0 = !(std::getenv("QMRI_OUT0") && std::atoi(std::getenv("QMRI_OUT0")) == 0);
        if (l == 0 && want_out0 && C == 32 && U->up2[0]->w_s3.p && W % 32 == 0) {
            const ConvLayer &L2 = *U->up2[0];
            qmri::Out0Args k;
            std::memset(&k, 0, sizeof(k));
            k.x = t1; k.ldx = C; k.xoff = 0; k.B = Bt; k.H = H; k.W = W;
            k.w = L2.w_s3.p; k.winv = L2.winv;
            k.bias = L2.bias.as<float>(); k.scale = L2.scale.as<float>(); k.shift = L2.shift.as<float>();
            k.head_w = U->head_w.as<float>(); k.head_b = U->head_b.as<float>(); k.nc = U->ncls;
            k.logits = logits; k.mask = mask;
            U_TRY(qmri::out0_launch(k, U->num_cu, st));
            U->trace += "up0.conv2:out0+head;";
            if (csum(U, "up0.conv2", logits, (long long)Bt * H * W * U->ncls * 4, st) != QMRI_OK) return QMRI_ERR_HIP;
            src = out;
            continue;
        }
        rc = conv3x3_parity(U, nm, *U->up2[l], t1, C, 0, Bt, H, W, out, C, 0, nullptr, 0, l == 0, logits, mask, st);
        if (rc != QMRI_OK) return rc;
        src = out;
    }
    return QMRI_OK;
}

// one batch of `Bt` slices already in U->in (device) -> logits / mask device pointers for that batch.
// Activation buffers are allocated for 4 bytes per channel (split fp16 hi | lo) and reinterpreted as bf16 in the plain mode.
static int forward_batch(Unet *U, int Bt, float *logits, unsigned char *mask, hipStream_t st) {
    U->trace.clear();
    if (U->split3) return forward_batch_parity(U, Bt, logits, mask, st);
    const int D = U->depth;
    const int s3 = 0;
    const int ab = 1;  // activations stored as bf16
    char nm[64];
    // ---- contracting path ----
    for (int l = 0; l < D; ++l) {
        const int H = U->Hl[l], W = U->Wl[l], C = U->nf[l];
        void *t1 = U->tmp[l]->p;
        // first layer computed inside conv2's halo stage (its feature map never goes to HBM): +1.5 %
        // end to end with a dedicated kernel instantiation; QMRI_FUSE_C1=0 turns it off
        static const bool want_fuse_c1 = !(std::getenv("QMRI_FUSE_C1") && std::atoi(std::getenv("QMRI_FUSE_C1")) == 0);
        const bool fuse_c1 = want_fuse_c1 && l == 0 && C == 32 && D > 1;
        if (l == 0) {
            if (!fuse_c1)
                U_TRY(qmri::conv3x3_c1_launch(U->in.as<float>(), Bt, H, W, U->c1_w.as<float>(),
                                              U->c1_b.as<float>(), C, t1, C, 0, ab, st));
        } else {
            auto k = conv_args(*U->down1[l], U->pool[l]->p, U->nf[l - 1], 0, Bt, H, W, t1, C, 0, H, W, 1, 1, 0, 0);
            snprintf(nm, sizeof(nm), "down%d.conv1", l);
            const int rc = conv_bf16(U, nm, *U->down1[l], k, st);
            if (rc != QMRI_OK) return rc;
        }
        if (l < D - 1) {
            // block output (post-BN) goes to the 2nd half of this level's concat buffer = the skip
            void *cat = U->cat[l]->p;
            auto k = conv_args(*U->down2[l], t1, C, 0, Bt, H, W, cat, 2 * C, C, H, W, 1, 1, 0, 0);
            if (U->fac[l] == 2) {
                k.pool_y = U->pool[l + 1]->p;  // MaxPooling2D((2, 2)) fused into the producing epilogue
                k.pool_ld = C;
            }
            if (fuse_c1) {
                k.c1_x = U->in.as<float>();
                k.c1_w = U->c1_w.as<float>();
                k.c1_b = U->c1_b.as<float>();
            }
            snprintf(nm, sizeof(nm), "down%d.conv2", l);
            const int rc = conv_bf16(U, nm, *U->down2[l], k, st);
            if (rc != QMRI_OK) return rc;
            if (U->fac[l] == 3)  // odd height: MaxPooling2D((3, 3)) (oaiunet2d.py:236-241) as its own kernel
                U_TRY(qmri::maxpoolk_launch(cat, 2 * C, C, Bt, H, W, C, 3, U->pool[l + 1]->p, st));
        } else {
            auto k = conv_args(*U->down2[l], t1, C, 0, Bt, H, W, U->bottom.p, C, 0, H, W, 1, 1, 0, 0);
            snprintf(nm, sizeof(nm), "down%d.conv2", l);
            const int rc = conv_bf16(U, nm, *U->down2[l], k, st);
            if (rc != QMRI_OK) return rc;
        }
    }
    // ---- expanding path ----
    const void *src = U->bottom.p;
    for (int l = D - 2; l >= 0; --l) {
        const int H = U->Hl[l], W = U->Wl[l], C = U->nf[l], Cup = U->nf[l + 1];
        void *cat = U->cat[l]->p;
        if (U->fac[l] == 3) {
            for (int ph = 0; ph < 9; ++ph) {
                auto k = conv_args(*U->updec3[(size_t)l * 9 + ph], src, Cup, 0, Bt, U->Hl[l + 1], U->Wl[l + 1], cat, 2 * C, 0, H, W,
                                   3, 3, ph / 3, ph % 3);
                U_TRY(qmri::conv_igemm_launch(k, s3, st));
            }
        } else if (U->split_levels >> l & 1u) {
            for (int ph = 0; ph < 4; ++ph) {
                auto k = conv_args(*U->updec_ph[(size_t)l * 4 + ph], src, Cup, 0, Bt, H / 2, W / 2, cat, 2 * C, 0, H, W, 2, 2,
                                   ph >> 1, ph & 1);
                U_TRY(qmri::conv_igemm_launch(k, s3, st));
            }
        } else {
            auto k = conv_args(*U->updec[(size_t)l], src, Cup, 0, Bt, H / 2, W / 2, cat, 2 * C, 0, H, W, 2, 2, 0, 0);
            snprintf(nm, sizeof(nm), "up%d.deconv", l);
            const int rc = conv_bf16(U, nm, *U->updec[(size_t)l], k, st);
            if (rc != QMRI_OK) return rc;
        }
        void *t1 = U->tmp[l]->p;
        auto k1 = conv_args(*U->up1[l], cat, 2 * C, 0, Bt, H, W, t1, C, 0, H, W, 1, 1, 0, 0);
        snprintf(nm, sizeof(nm), "up%d.conv1", l);
        {
            const int rc = conv_bf16(U, nm, *U->up1[l], k1, st);
            if (rc != QMRI_OK) return rc;
        }
        void *out = U->upout[l]->p;
        auto k2 = conv_args(*U->up2[l], t1, C, 0, Bt, H, W, out, C, 0, H, W, 1, 1, 0, 0);
        const bool fuse_head = l == 0 && C == 32;  // the whole channel run of a pixel is in one tile
        if (fuse_head) {
            k2.y = nullptr;  // the last feature map is only consumed by the head: never written to HBM
            k2.head_w = U->head_w.as<float>();
            k2.head_b = U->head_b.as<float>();
            k2.head_nc = U->ncls;
            k2.logits = logits;
            k2.mask = mask;
        }
        snprintf(nm, sizeof(nm), "up%d.conv2", l);
        {
            const int rc = conv_bf16(U, nm, *U->up2[l], k2, st);
            if (rc != QMRI_OK) return rc;
        }
        src = out;
        if (l == 0 && !fuse_head)
            U_TRY(qmri::head_launch(src, (long long)Bt * U->H * U->W, U->nf[0], U->head_w.as<float>(),
                                    U->head_b.as<float>(), U->ncls, logits, mask, ab, st));
    }
    return QMRI_OK;
}

}  // extern "C"

// ---- the fp16 range of the split layout: choose / raise the activation exponent, repeat on saturation (Unet::sat) ----
// `body()` queues the whole volume through the network on `st` with the parameters as they are.  Plain bf16 activations
// have the fp32 exponent range: one pass, no flag.
template <class F>
static int run_in_range(Unet *U, const float *xd, long long n, bool whitened, hipStream_t st, F &&body) {
    if (!U->split3) {
        const int rc0 = U->set_act_shift(0);
        return rc0 != QMRI_OK ? rc0 : body();
    }
    int S = 0;
    bool nonfinite_in = false;
    if (!whitened) {  // raw intensities (IWOAIOAIUnet2D, oaiunet2d.py:322-323): bring max |x| below 128 to start with
        unsigned int bits = 0;
        U_TRY(qmri::absmax_launch(xd, n, reinterpret_cast<unsigned int *>(U->sat_ptr() + 1), st));
        U_TRY(hipMemcpyAsync(&bits, U->sat_ptr() + 1, sizeof(bits), hipMemcpyDeviceToHost, st));
        U_TRY(hipStreamSynchronize(st));
        float m;
        std::memcpy(&m, &bits, 4);
        // an input holding Inf / NaN has no exponent that brings it into range: it goes through ONCE, unscaled, and the
        // affected slices come out as the reference's do (NaN logits, empty masks) -- no retries, no "exceeds fp16" error
        nonfinite_in = bits >= 0x7f800000u;
        if (!nonfinite_in && m >= 128.f) S = std::ilogb(m) - 6;
    }
    const int bump_at_entry = U->act_bump;
    S += U->act_bump;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (S > 60) break;
        int rc = U->set_act_shift(S);
        if (rc != QMRI_OK) return rc;
        U_TRY(hipMemsetAsync(U->sat.p, 0, sizeof(int), st));
        rc = body();
        if (rc != QMRI_OK) return rc;
        int flag = 0;
        U_TRY(hipMemcpyAsync(&flag, U->sat.p, sizeof(int), hipMemcpyDeviceToHost, st));
        U_TRY(hipStreamSynchronize(st));
        if (!flag || nonfinite_in) {
            U->trace += "act_shift:" + std::to_string(S) + ";";  // (tests: which exponent the forward ran at)
            return QMRI_OK;
        }
        S += 6;             // a feature map left the fp16 range: the same forward, 64 times smaller (exactly)
        U->act_bump += 6;   // (remembered: the next volume of this model starts there)
    }
    U->act_bump = bump_at_entry;  // a volume that cannot be brought into range does not change how the next one starts
    return ufail(QMRI_ERR_UNSUPPORTED,
                 "feature-map values exceed the fp16 hi + lo range of the parity mode even after scaling the network by 2^-%d; "
                 "use precision \"bf16\" for this model / input", S);
}

// one batch of the (already whitened) device volume -> U->in, scaled by 2^-act_shift
static hipError_t load_batch(Unet *U, const float *src, long long count, hipStream_t st) {
    if (U->act_shift == 0) return hipMemcpyAsync(U->in.p, src, (size_t)count * 4, hipMemcpyDeviceToDevice, st);
    return qmri::scale_copy_launch(src, count, std::ldexp(1.f, -U->act_shift), U->in.as<float>(), st);
}

extern "C" {

int qmri_unet2d_forward(void *handle, const float *x, int32_t S, int32_t x_on_device, int32_t whiten,
                        double whiten_eps, float *logits, uint8_t *mask, int32_t out_on_device,
                        void *stream) {
    if (!handle || !x) return ufail(QMRI_ERR_ARG, "handle / x is NULL");
    if (S < 0) return ufail(QMRI_ERR_ARG, "S < 0");
    if (S == 0) return QMRI_OK;
    Unet *U = static_cast<Unet *>(handle);
    hipStream_t st = static_cast<hipStream_t>(stream);
    U_TRY(hipSetDevice(U->device));
    const long long slice = (long long)U->H * U->W;
    const long long n = (long long)S * slice;

    // the volume on the device (whitening needs whole-volume statistics, seg_model.py:127)
    const float *xd = x;
    if (!x_on_device || whiten) {
        if (U->vol_cap < n) {
            U_TRY(U->vol.alloc((size_t)n * 4));
            U->vol_cap = n;
        }
        if (!x_on_device) {
            U_TRY(hipMemcpyAsync(U->vol.p, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
            xd = U->vol.as<float>();
        }
        if (whiten) {
            U_TRY(qmri::whiten_launch(xd, n, whiten_eps, U->stats.as<double>(), U->vol.as<float>(), st));
            xd = U->vol.as<float>();
        }
    }
    float *lg_dev = nullptr;
    unsigned char *mk_dev = nullptr;
    if (!out_on_device) {
        if (logits) {
            if (!U->logits.p) U_TRY(U->logits.alloc((size_t)U->maxB * slice * U->ncls * 4));
            lg_dev = U->logits.as<float>();
        }
        if (mask) {
            if (!U->mask.p) U_TRY(U->mask.alloc((size_t)U->maxB * slice * U->ncls));
            mk_dev = U->mask.as<unsigned char>();
        }
    }
    auto body = [&]() -> int {
        for (int s0 = 0; s0 < S; s0 += U->maxB) {
            const int Bt = (S - s0) < U->maxB ? (S - s0) : U->maxB;
            U->csum_pass = s0 / U->maxB;
            U_TRY(load_batch(U, xd + (long long)s0 * slice, (long long)Bt * slice, st));
            float *lg = out_on_device ? (logits ? logits + (long long)s0 * slice * U->ncls : nullptr) : lg_dev;
            unsigned char *mk = out_on_device ? (mask ? mask + (long long)s0 * slice * U->ncls : nullptr) : mk_dev;
            const int rc = forward_batch(U, Bt, lg, mk, st);
            if (rc != QMRI_OK) return rc;
            if (!out_on_device) {
                if (logits)
                    U_TRY(hipMemcpyAsync(logits + (long long)s0 * slice * U->ncls, lg_dev,
                                         (size_t)Bt * slice * U->ncls * 4, hipMemcpyDeviceToHost, st));
                if (mask)
                    U_TRY(hipMemcpyAsync(mask + (long long)s0 * slice * U->ncls, mk_dev,
                                         (size_t)Bt * slice * U->ncls, hipMemcpyDeviceToHost, st));
                U_TRY(hipStreamSynchronize(st));  // staging buffers are reused by the next batch
            }
        }
        return QMRI_OK;
    };
    if (csum_begin(U, st) != QMRI_OK) return QMRI_ERR_HIP;
    const int rc = run_in_range(U, xd, n, whiten != 0, st, body);
    if (rc != QMRI_OK) return rc;
    if (csum_end(U, st) != QMRI_OK) return QMRI_ERR_HIP;
    if (!out_on_device || !x_on_device) U_TRY(hipStreamSynchronize(st));
    return QMRI_OK;
}

// Whole-volume segmentation in the reference's own layouts (see include/qmri.h): both transposes on the GPU, one
// upload and one download.
int qmri_unet2d_segment_volume(void *handle, const float *vol_hws, int32_t S, int32_t whiten, double whiten_eps,
                               uint8_t *mask_chws, void *stream) {
    if (!handle || !vol_hws || !mask_chws) return ufail(QMRI_ERR_ARG, "handle / volume / mask is NULL");
    if (S < 0) return ufail(QMRI_ERR_ARG, "S < 0");
    if (S == 0) return QMRI_OK;
    Unet *U = static_cast<Unet *>(handle);
    if (U->ncls > 4) return ufail(QMRI_ERR_UNSUPPORTED, "at most 4 classes");
    hipStream_t st = static_cast<hipStream_t>(stream);
    U_TRY(hipSetDevice(U->device));
    const long long P = (long long)U->H * U->W;
    const long long n = (long long)S * P;
    if (U->seg_cap < n) {
        U_TRY(U->vol_in.alloc((size_t)n * 4));
        U_TRY(U->mask_all.alloc((size_t)n * 4));  // 4 bytes per pixel (classes padded to 4)
        U_TRY(U->mask_planes.alloc((size_t)n * U->ncls));
        U->seg_cap = n;
    }
    if (U->vol_cap < n) {
        U_TRY(U->vol.alloc((size_t)n * 4));
        U->vol_cap = n;
    }
    U_TRY(hipMemcpyAsync(U->vol_in.p, vol_hws, (size_t)n * 4, hipMemcpyHostToDevice, st));
    U_TRY(qmri::transpose_ps_launch(U->vol_in.as<float>(), P, S, U->vol.as<float>(), st));
    if (whiten) U_TRY(qmri::whiten_launch(U->vol.as<float>(), n, whiten_eps, U->stats.as<double>(), U->vol.as<float>(), st));
    if (U->ncls < 4) U_TRY(hipMemsetAsync(U->mask_all.p, 0, (size_t)n * 4, st));
    unsigned char *mk_all = U->mask_all.as<unsigned char>();
    auto body = [&]() -> int {
        for (int s0 = 0; s0 < S; s0 += U->maxB) {
            const int Bt = (S - s0) < U->maxB ? (S - s0) : U->maxB;
            U_TRY(load_batch(U, U->vol.as<float>() + (long long)s0 * P, (long long)Bt * P, st));
            unsigned char *mk = U->ncls == 4 ? mk_all + (long long)s0 * P * 4 : nullptr;
            if (U->ncls == 4) {
                const int rc = forward_batch(U, Bt, nullptr, mk, st);
                if (rc != QMRI_OK) return rc;
            } else {  // fewer classes: the network writes ncls bytes per pixel; widen to the 4-byte records on the way
                if (!U->mask.p) U_TRY(U->mask.alloc((size_t)U->maxB * P * U->ncls));
                const int rc = forward_batch(U, Bt, nullptr, U->mask.as<unsigned char>(), st);
                if (rc != QMRI_OK) return rc;
                U_TRY(hipMemcpy2DAsync(mk_all + (long long)s0 * P * 4, 4, U->mask.p, (size_t)U->ncls, (size_t)U->ncls,
                                       (size_t)Bt * P, hipMemcpyDeviceToDevice, st));
            }
        }
        return QMRI_OK;
    };
    {
        const int rc = run_in_range(U, U->vol.as<float>(), n, whiten != 0, st, body);
        if (rc != QMRI_OK) return rc;
    }
    U_TRY(qmri::mask_planes_launch(mk_all, P, S, U->ncls, U->mask_planes.as<unsigned char>(), st));
    {
        // Everything above is queued; the GPU needs tens of milliseconds.  The caller's output array is usually fresh
        // (numpy.empty): touch its pages NOW, so that the copy below does not fault them in one by one (~65 ms per GB
        // here -- 6 ms for the four masks of a 384 x 384 x 160 volume, a sixth of the network's time).
        volatile uint8_t *out = mask_chws;
        const size_t bytes = (size_t)n * U->ncls;
        for (size_t off = 0; off < bytes; off += 4096) out[off] = 0;
        out[bytes - 1] = 0;
    }
    U_TRY(hipMemcpyAsync(mask_chws, U->mask_planes.p, (size_t)n * U->ncls, hipMemcpyDeviceToHost, st));
    U_TRY(hipStreamSynchronize(st));
    return QMRI_OK;
}

// Single-layer entry for unit tests and for users who want the operators alone: host NHWC fp32 in/out.
//   transposed = 0: Conv2D(Cout, 3x3, SAME), kernel (3,3,Cin,Cout)      (oaiunet2d.py:213-226)
//   transposed = 1: Conv2DTranspose(Cout, 3x3, strides 2, SAME), kernel (3,3,Cout,Cin), output 2H x 2W
//   epilogue: y = scale * relu?(acc + bias) + shift  (scale/shift nullable)
int qmri_conv2d_nhwc_host(const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float *kernel,
                          const float *bias, const float *scale, const float *shift, int32_t relu,
                          int32_t Cout, int32_t transposed, int32_t precision, float *y, int32_t device) {
    if (!x || !kernel || !bias || !y) return ufail(QMRI_ERR_ARG, "NULL argument");
    if (Cin % 32 || Cout % 32) return ufail(QMRI_ERR_UNSUPPORTED, "Cin and Cout must be multiples of 32");
    U_TRY(hipSetDevice(device));
    const int Ho = transposed ? 2 * H : H, Wo = transposed ? 2 * W : W;
    const int ab = precision == 0 || precision == 3;  // plain bf16 mode (3: forced onto conv_s3_kernel<.., ONE>): bf16 activations on the device; parity mode: the split layout
    const long long nx = (long long)B * H * W * Cin, ny = (long long)B * Ho * Wo * Cout;
    DevBuf dx, dy, dxb, dyb;
    U_TRY(dx.alloc((size_t)nx * 4));
    U_TRY(dy.alloc((size_t)ny * 4));
    U_TRY(hipMemcpy(dx.p, x, (size_t)nx * 4, hipMemcpyHostToDevice));
    if (ab) {
        U_TRY(dxb.alloc((size_t)nx * 2));
        U_TRY(dyb.alloc((size_t)ny * 2));
        U_TRY(qmri::cast_launch(dx.p, nx, dxb.p, 1, nullptr));
    } else {
        U_TRY(dxb.alloc((size_t)nx * 4));
        U_TRY(dyb.alloc((size_t)ny * 4));
        U_TRY(qmri::split_cast_launch(dx.p, (long long)B * H * W, Cin, dxb.p, 1, nullptr));
    }
    std::vector<float> wk, sc, sh;
    if (scale && shift) {
        sc.assign(scale, scale + Cout);
        sh.assign(shift, shift + Cout);
    }
    {
        ConvLayer L;
        L.relu = relu;
        if (transposed)
            pack_deconv_fused(kernel, Cin, Cout, L, wk);
        else
            pack_conv3x3(kernel, Cin, Cout, L, wk);
        U_TRY(L.upload(wk, bias, sc.empty() ? nullptr : &sc, sc.empty() ? nullptr : &sh));
        hipDeviceProp_t prop;
        U_TRY(hipGetDeviceProperties(&prop, device));
        // precision 1: the kernel the engine would pick for this layer; 2: force the general kernel (tests compare both)
        // 4: conv_c4_kernel or an error; 5: conv_s3_kernel whatever the layer; 6: conv_c4_kernel, whole work items only (tests compare them)
        const bool rag = !ab && (precision == 1 || precision == 4 || precision == 6) && c4_ragged(W);  // image tiles with a ragged last column: conv_c4_kernel / deconv_d4_kernel only
        const bool s3 = !ab && precision != 2 && (s3_width_ok(W) || rag);
        if (!ab) U_TRY(L.upload_parity(wk, s3 && !rag, rag));
        auto run_general = [&]() -> hipError_t {
            auto k = conv_args(L, dxb.p, Cin, 0, B, H, W, dyb.p, Cout, 0, Ho, Wo, transposed ? 2 : 1, transposed ? 2 : 1, 0, 0);
            if (!ab) {
                k.w_hi = L.h_hi.as<__bf16>();
                k.w_lo = L.h_lo.as<__bf16>();
                k.winv = L.winv;
            }
            return qmri::conv_igemm_launch(k, !ab, nullptr);
        };
        if (precision == 3) {
            if (Cin % 64 || !s3_width_ok(W)) return ufail(QMRI_ERR_UNSUPPORTED, "bf16 on conv_s3_kernel needs Cin %% 64 == 0 and a width it tiles");
            U_TRY(L.upload_s3_bf16(wk));
            qmri::ConvS3Args k;
            std::memset(&k, 0, sizeof(k));
            k.one = 1;
            k.x = dxb.p; k.ldx = Cin / 2; k.B = B; k.H = H; k.W = W; k.Cin = Cin / 2; k.Cout = Cout;
            k.deconv = transposed ? 1 : 0;
            k.w = L.w_s3b.p; k.winv = 1.f;
            k.bias = L.bias.as<float>();
            k.scale = L.has_affine ? L.scale.as<float>() : nullptr;
            k.shift = L.has_affine ? L.shift.as<float>() : nullptr;
            k.relu = relu;
            k.y = dyb.p; k.ldy = Cout;
            U_TRY(qmri::conv_s3_launch(k, prop.multiProcessorCount, nullptr));
        } else if (s3) {
            qmri::ConvS3Args k;
            std::memset(&k, 0, sizeof(k));
            k.x = dxb.p; k.ldx = Cin; k.B = B; k.H = H; k.W = W; k.Cin = Cin; k.Cout = Cout;
            k.deconv = transposed ? 1 : 0;
            k.w = L.w_s3.p; k.winv = L.winv;
            k.w_c4 = L.w_c4.p;
            k.c4_mode = (precision == 4 || precision == 6) ? 1 : precision == 5 ? -1 : 0;
            k.c4_split = precision == 6 ? -1 : 0;  // 6: conv_c4_kernel without the channel-split last round
            k.bias = L.bias.as<float>();
            k.scale = L.has_affine ? L.scale.as<float>() : nullptr;
            k.shift = L.has_affine ? L.shift.as<float>() : nullptr;
            k.relu = relu;
            k.y = dyb.p; k.ldy = Cout;
            if (rag && !(transposed ? qmri::conv_s3_takes_d4(k) : qmri::conv_s3_takes_c4(k, prop.multiProcessorCount))) {
                if (precision != 1) return ufail(QMRI_ERR_UNSUPPORTED, "conv_c4_kernel / deconv_d4_kernel do not take this layer");
                U_TRY(run_general());  // (what the engine does with such a layer)
            } else {
                U_TRY(qmri::conv_s3_launch(k, prop.multiProcessorCount, nullptr));
            }
        } else {
            U_TRY(run_general());
        }
        U_TRY(hipDeviceSynchronize());
    }
    if (ab)
        U_TRY(qmri::cast_launch(dyb.p, ny, dy.p, 0, nullptr));
    else
        U_TRY(qmri::split_cast_launch(dyb.p, (long long)B * Ho * Wo, Cout, dy.p, 0, nullptr));
    U_TRY(hipDeviceSynchronize());
    U_TRY(hipMemcpy(y, dy.p, (size_t)B * Ho * Wo * Cout * 4, hipMemcpyDeviceToHost));
    return QMRI_OK;
}

}  // extern "C"
