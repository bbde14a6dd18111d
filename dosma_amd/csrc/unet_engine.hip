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
            return ufail(QMRI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                   \
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

struct ConvLayer {
    int Cin = 0, Cout = 0, ntaps = 0;
    int dy[9] = {0}, dx[9] = {0};
    int relu = 0;
    int deconv = 0;
    DevBuf w_hi, w_lo, bias, scale, shift;
    bool has_affine = false;

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
        has_affine = sc != nullptr;
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

struct Unet {
    int depth = 0, ncls = 0, H = 0, W = 0, maxB = 0, device = 0, split3 = 1;
    std::vector<int> nf;
    DevBuf c1_w, c1_b;  // first layer fp32: [9][nf0], [nf0]
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
};

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
    const int div = 1 << (d->depth - 1);
    if (d->H <= 0 || d->W <= 0 || d->H % div || d->W % div)
        return ufail(QMRI_ERR_UNSUPPORTED,
                     "H and W must be divisible by %d: odd sizes take the reference's 3x3 pooling branch "
                     "(oaiunet2d.py:236-241), which is not implemented", div);
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
    U->maxB = d->max_batch;
    U->device = d->device;
    U->split3 = d->precision != 0;
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
        } else {
            U->down1[l].reset(new ConvLayer);
            U->down1[l]->relu = 1;
            pack_conv3x3(k1, Cin, C, *U->down1[l], wk);
            U_TRY(U->down1[l]->upload(wk, b1, nullptr, nullptr));
        }
        fold_bn(C, sc, sh);
        U->down2[l].reset(new ConvLayer);
        U->down2[l]->relu = 1;
        pack_conv3x3(k2, C, C, *U->down2[l], wk);
        U_TRY(U->down2[l]->upload(wk, b2, &sc, &sh));
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
            if (U->split_levels >> l & 1u)
                for (int ph = 0; ph < 4; ++ph) {
                    auto &P = U->updec_ph[(size_t)l * 4 + ph];
                    P.reset(new ConvLayer);
                    P->relu = 0;
                    pack_deconv_phase(kd, Cup, C, ph >> 1, ph & 1, *P, wk);
                    U_TRY(P->upload(wk, bd, nullptr, nullptr));
                }
        }
        U->up1[l].reset(new ConvLayer);
        U->up1[l]->relu = 1;
        pack_conv3x3(k1, 2 * C, C, *U->up1[l], wk);
        U_TRY(U->up1[l]->upload(wk, b1, nullptr, nullptr));
        fold_bn(C, sc, sh);
        U->up2[l].reset(new ConvLayer);
        U->up2[l]->relu = 1;
        pack_conv3x3(k2, C, C, *U->up2[l], wk);
        U_TRY(U->up2[l]->upload(wk, b2, &sc, &sh));
    }
    // head: Keras (1,1,C0,NC) == [C0][NC]
    U_TRY(U->head_w.alloc((size_t)U->nf[0] * U->ncls * 4));
    U_TRY(hipMemcpy(U->head_w.p, T[ti], (size_t)U->nf[0] * U->ncls * 4, hipMemcpyHostToDevice));
    U_TRY(U->head_b.alloc((size_t)U->ncls * 4));
    U_TRY(hipMemcpy(U->head_b.p, T[ti + 1], (size_t)U->ncls * 4, hipMemcpyHostToDevice));

    // activation buffers for max_batch slices
    const long long B = U->maxB;
    U_TRY(U->in.alloc((size_t)B * U->H * U->W * 4));
    U->tmp.resize(d->depth);
    U->cat.resize(d->depth);
    U->pool.resize(d->depth);
    U->upout.resize(d->depth);
    for (int l = 0; l < d->depth; ++l) {
        const long long pix = B * (U->H >> l) * (U->W >> l);
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
        U_TRY(U->bottom.alloc((size_t)B * (U->H >> l) * (U->W >> l) * U->nf[l] * 4));
    }
    U_TRY(U->stats.alloc(4 * sizeof(double)));
    *handle = U.release();
    return QMRI_OK;
}

int qmri_unet2d_set_precision(void *handle, int32_t precision) {
    if (!handle) return ufail(QMRI_ERR_ARG, "handle is NULL");
    static_cast<Unet *>(handle)->split3 = precision != 0;
    return QMRI_OK;
}

// one batch of `Bt` slices already in U->in (device) -> logits / mask device pointers for that batch.
// Activation buffers are allocated for fp32 and reinterpreted as bf16 in the plain-bf16 mode.
static int forward_batch(Unet *U, int Bt, float *logits, unsigned char *mask, hipStream_t st) {
    const int D = U->depth;
    const int s3 = U->split3;
    const int ab = s3 ? 0 : 1;  // activations stored as bf16?
    const size_t es = ab ? 2 : 4;
    auto at = [&](const DevBuf &b, long long elem_off) -> void * {
        return static_cast<unsigned char *>(b.p) + (size_t)elem_off * es;
    };
    (void)at;
    // ---- contracting path ----
    for (int l = 0; l < D; ++l) {
        const int H = U->H >> l, W = U->W >> l, C = U->nf[l];
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
            U_TRY(qmri::conv_igemm_launch(k, s3, st));
        }
        if (l < D - 1) {
            // block output (post-BN) goes to the 2nd half of this level's concat buffer = the skip
            void *cat = U->cat[l]->p;
            auto k = conv_args(*U->down2[l], t1, C, 0, Bt, H, W, cat, 2 * C, C, H, W, 1, 1, 0, 0);
            k.pool_y = U->pool[l + 1]->p;  // MaxPooling2D fused into the producing epilogue
            k.pool_ld = C;
            if (fuse_c1) {
                k.c1_x = U->in.as<float>();
                k.c1_w = U->c1_w.as<float>();
                k.c1_b = U->c1_b.as<float>();
            }
            U_TRY(qmri::conv_igemm_launch(k, s3, st));
        } else {
            auto k = conv_args(*U->down2[l], t1, C, 0, Bt, H, W, U->bottom.p, C, 0, H, W, 1, 1, 0, 0);
            U_TRY(qmri::conv_igemm_launch(k, s3, st));
        }
    }
    // ---- expanding path ----
    const void *src = U->bottom.p;
    for (int l = D - 2; l >= 0; --l) {
        const int H = U->H >> l, W = U->W >> l, C = U->nf[l], Cup = U->nf[l + 1];
        void *cat = U->cat[l]->p;
        if (U->split_levels >> l & 1u) {
            for (int ph = 0; ph < 4; ++ph) {
                auto k = conv_args(*U->updec_ph[(size_t)l * 4 + ph], src, Cup, 0, Bt, H / 2, W / 2, cat, 2 * C, 0, H, W, 2, 2,
                                   ph >> 1, ph & 1);
                U_TRY(qmri::conv_igemm_launch(k, s3, st));
            }
        } else {
            auto k = conv_args(*U->updec[(size_t)l], src, Cup, 0, Bt, H / 2, W / 2, cat, 2 * C, 0, H, W, 2, 2, 0, 0);
            U_TRY(qmri::conv_igemm_launch(k, s3, st));
        }
        void *t1 = U->tmp[l]->p;
        auto k1 = conv_args(*U->up1[l], cat, 2 * C, 0, Bt, H, W, t1, C, 0, H, W, 1, 1, 0, 0);
        U_TRY(qmri::conv_igemm_launch(k1, s3, st));
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
        U_TRY(qmri::conv_igemm_launch(k2, s3, st));
        src = out;
        if (l == 0 && !fuse_head)
            U_TRY(qmri::head_launch(src, (long long)Bt * U->H * U->W, U->nf[0], U->head_w.as<float>(),
                                    U->head_b.as<float>(), U->ncls, logits, mask, ab, st));
    }
    return QMRI_OK;
}

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
    for (int s0 = 0; s0 < S; s0 += U->maxB) {
        const int Bt = (S - s0) < U->maxB ? (S - s0) : U->maxB;
        U_TRY(hipMemcpyAsync(U->in.p, xd + (long long)s0 * slice, (size_t)Bt * slice * 4,
                             hipMemcpyDeviceToDevice, st));
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
    for (int s0 = 0; s0 < S; s0 += U->maxB) {
        const int Bt = (S - s0) < U->maxB ? (S - s0) : U->maxB;
        U_TRY(hipMemcpyAsync(U->in.p, U->vol.as<float>() + (long long)s0 * P, (size_t)Bt * P * 4, hipMemcpyDeviceToDevice, st));
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
    U_TRY(qmri::mask_planes_launch(mk_all, P, S, U->ncls, U->mask_planes.as<unsigned char>(), st));
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
    const int ab = precision == 0;  // plain bf16 mode: bf16 activations on the device
    const long long nx = (long long)B * H * W * Cin, ny = (long long)B * Ho * Wo * Cout;
    DevBuf dx, dy, dxb, dyb;
    U_TRY(dx.alloc((size_t)nx * 4));
    U_TRY(dy.alloc((size_t)ny * 4));
    U_TRY(hipMemcpy(dx.p, x, (size_t)nx * 4, hipMemcpyHostToDevice));
    if (ab) {
        U_TRY(dxb.alloc((size_t)nx * 2));
        U_TRY(dyb.alloc((size_t)ny * 2));
        U_TRY(qmri::cast_launch(dx.p, nx, dxb.p, 1, nullptr));
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
        auto k = conv_args(L, ab ? dxb.p : dx.p, Cin, 0, B, H, W, ab ? dyb.p : dy.p, Cout, 0, Ho, Wo,
                           transposed ? 2 : 1, transposed ? 2 : 1, 0, 0);
        U_TRY(qmri::conv_igemm_launch(k, precision != 0, nullptr));
        U_TRY(hipDeviceSynchronize());
    }
    if (ab) {
        U_TRY(qmri::cast_launch(dyb.p, ny, dy.p, 0, nullptr));
        U_TRY(hipDeviceSynchronize());
    }
    U_TRY(hipMemcpy(y, dy.p, (size_t)B * Ho * Wo * Cout * 4, hipMemcpyDeviceToHost));
    return QMRI_OK;
}

}  // extern "C"
