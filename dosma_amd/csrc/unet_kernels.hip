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

#include "qmri_internal.h"

namespace qmri {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kBM = 128;        // output pixels per block
constexpr int kBK = 32;         // channels per K step (one tap x 32 input channels)
constexpr int kLdsRow = 40;     // bf16 elements per LDS row: 32 + 8 pad -> 80 B stride, conflict-free b128 reads

template <int BN>
struct TileCfg {
    static constexpr int WAVES_N = BN >= 64 ? 2 : 1;
    static constexpr int WAVES_M = 4 / WAVES_N;
    static constexpr int TM = kBM / WAVES_M / 32;
    static constexpr int TN = BN / WAVES_N / 32;
};

__device__ __forceinline__ void split_bf16(const float4 &a, const float4 &b, bf16x8 &hi, bf16x8 &lo,
                                           bool want_lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = static_cast<__bf16>(v[i]);
        hi[i] = h;
        if (want_lo) lo[i] = static_cast<__bf16>(v[i] - static_cast<float>(h));
    }
}

template <int BN, bool SPLIT3>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvKArgs A) {
    using C = TileCfg<BN>;
    constexpr int NPLANES = SPLIT3 ? 2 : 1;
    constexpr int A_BYTES = kBM * kLdsRow * 2;
    constexpr int B_BYTES = BN * kLdsRow * 2;
    constexpr int BUF_BYTES = NPLANES * (A_BYTES + B_BYTES);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *rowpix = reinterpret_cast<int *>(smem + 2 * BUF_BYTES);  // [kBM] output pixel index (-1 = none)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / C::WAVES_N;
    const int wn = wave % C::WAVES_N;
    const long long M = (long long)A.B * A.H * A.W;
    const long long m0 = (long long)blockIdx.x * kBM;
    const int n0 = blockIdx.y * BN;
    const int K = A.ntaps * A.Cin;

    // ---- per-thread gather coordinates for the A (activation) tile: 2 (row, 8-channel group) pairs ----
    int a_row[2], a_grp[2], a_y[2], a_x[2];
    long long a_base[2];
    bool a_ok[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int idx = tid + r * 256;
        a_row[r] = idx >> 2;
        a_grp[r] = idx & 3;
        const long long m = m0 + a_row[r];
        a_ok[r] = m < M;
        const long long mm = a_ok[r] ? m : 0;
        const int xw = (int)(mm % A.W);
        const long long t = mm / A.W;
        const int yh = (int)(t % A.H);
        a_y[r] = yh;
        a_x[r] = xw;
        a_base[r] = mm;  // pixel index of the un-shifted position
    }
    if (tid < kBM) {
        const long long m = m0 + tid;
        int pix = -1;
        if (m < M) {
            const int xw = (int)(m % A.W);
            const long long t = m / A.W;
            const int yh = (int)(t % A.H);
            const int b = (int)(t / A.H);
            pix = (b * A.Ho + (yh * A.sy + A.py)) * A.Wo + (xw * A.sx + A.px);
        }
        rowpix[tid] = pix;
    }

    constexpr int B_PAIRS = BN * 4 / 256 > 0 ? BN * 4 / 256 : 1;  // (row, group) pairs per thread for B
    float4 ra[2][2];
    bf16x8 rb_hi[B_PAIRS], rb_lo[B_PAIRS];

    auto load_tile = [&](int step) {
        const int k0 = step * kBK;
        const int tap = k0 / A.Cin;
        const int c0 = k0 - tap * A.Cin;
        const int dy = A.tap_dy[tap], dx = A.tap_dx[tap];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int yy = a_y[r] + dy, xx = a_x[r] + dx;
            const bool ok = a_ok[r] && yy >= 0 && yy < A.H && xx >= 0 && xx < A.W;
            if (ok) {
                const float *p = A.x + (a_base[r] + (long long)dy * A.W + dx) * A.ldx + A.xoff + c0 +
                                 a_grp[r] * 8;
                ra[r][0] = *reinterpret_cast<const float4 *>(p);
                ra[r][1] = *reinterpret_cast<const float4 *>(p + 4);
            } else {
                ra[r][0] = make_float4(0.f, 0.f, 0.f, 0.f);
                ra[r][1] = ra[r][0];
            }
        }
#pragma unroll
        for (int r = 0; r < B_PAIRS; ++r) {
            const int idx = tid + r * 256;
            if (idx < BN * 4) {
                const int row = idx >> 2, grp = idx & 3;
                const long long off = (long long)(n0 + row) * K + k0 + grp * 8;
                rb_hi[r] = *reinterpret_cast<const bf16x8 *>(A.w_hi + off);
                if (SPLIT3) rb_lo[r] = *reinterpret_cast<const bf16x8 *>(A.w_lo + off);
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char *base = smem + buf * BUF_BYTES;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            bf16x8 hi, lo;
            split_bf16(ra[r][0], ra[r][1], hi, lo, SPLIT3);
            const int off = (a_row[r] * kLdsRow + a_grp[r] * 8) * 2;
            *reinterpret_cast<bf16x8 *>(base + off) = hi;
            if (SPLIT3) *reinterpret_cast<bf16x8 *>(base + A_BYTES + off) = lo;
        }
        unsigned char *bb = base + NPLANES * A_BYTES;
#pragma unroll
        for (int r = 0; r < B_PAIRS; ++r) {
            const int idx = tid + r * 256;
            if (idx < BN * 4) {
                const int off = ((idx >> 2) * kLdsRow + (idx & 3) * 8) * 2;
                *reinterpret_cast<bf16x8 *>(bb + off) = rb_hi[r];
                if (SPLIT3) *reinterpret_cast<bf16x8 *>(bb + B_BYTES + off) = rb_lo[r];
            }
        }
    };

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int steps = K / kBK;
    load_tile(0);
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        store_tile(buf);
        __syncthreads();
        if (s + 1 < steps) load_tile(s + 1);  // global loads in flight under the MFMAs below
        const unsigned char *base = smem + buf * BUF_BYTES;
        const unsigned char *bb = base + NPLANES * A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int koff = (kk * 16 + (lane >> 5) * 8) * 2;
            bf16x8 a_hi[C::TM], a_lo[C::TM], b_hi[C::TN], b_lo[C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i) {
                const int row = (wm * C::TM + i) * 32 + (lane & 31);
                a_hi[i] = *reinterpret_cast<const bf16x8 *>(base + row * kLdsRow * 2 + koff);
                if (SPLIT3)
                    a_lo[i] = *reinterpret_cast<const bf16x8 *>(base + A_BYTES + row * kLdsRow * 2 + koff);
            }
#pragma unroll
            for (int j = 0; j < C::TN; ++j) {
                const int col = (wn * C::TN + j) * 32 + (lane & 31);
                b_hi[j] = *reinterpret_cast<const bf16x8 *>(bb + col * kLdsRow * 2 + koff);
                if (SPLIT3)
                    b_lo[j] = *reinterpret_cast<const bf16x8 *>(bb + B_BYTES + col * kLdsRow * 2 + koff);
            }
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j) {
                    if (SPLIT3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[i], b_hi[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_lo[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
                }
        }
    }

    // ---- epilogue: y = scale * relu(acc + bias) + shift, written NHWC (32 consecutive channels per
    // half-wave -> 128-byte segments) ----
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
        const int n = n0 + (wn * C::TN + j) * 32 + (lane & 31);
        const float bias = A.bias ? A.bias[n] : 0.f;
        const float scale = A.scale ? A.scale[n] : 1.f;
        const float shift = A.shift ? A.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < C::TM; ++i) {
            const int rbase = (wm * C::TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rbase + (e & 3) + 8 * (e >> 2);
                const int pix = rowpix[row];
                if (pix >= 0) {
                    float v = acc[i][j][e] + bias;
                    if (A.relu) v = fmaxf(v, 0.f);
                    v = v * scale + shift;
                    A.y[(long long)pix * A.ldy + A.yoff + n] = v;
                }
            }
        }
    }
}

hipError_t conv_igemm_launch(const ConvKArgs &k, int split3, hipStream_t stream) {
    const long long M = (long long)k.B * k.H * k.W;
    const int bn = k.Cout % 128 == 0 ? 128 : (k.Cout % 64 == 0 ? 64 : 32);
    if (k.Cout % bn != 0 || k.Cin % kBK != 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)((M + kBM - 1) / kBM), (unsigned)(k.Cout / bn));
    const int planes = split3 ? 2 : 1;
    const size_t lds = 2 * (size_t)planes * (kBM + bn) * kLdsRow * 2 + kBM * sizeof(int);
    (void)hipGetLastError();
#define QMRI_CONV_CASE(BN_, S3_)                                                                    \
    do {                                                                                            \
        auto fn = conv_igemm_kernel<BN_, S3_>;                                                      \
        if (lds > 64 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                          \
        }                                                                                           \
        hipLaunchKernelGGL(fn, grid, dim3(256), lds, stream, k);                                    \
    } while (0)
    if (bn == 128) {
        if (split3) QMRI_CONV_CASE(128, true); else QMRI_CONV_CASE(128, false);
    } else if (bn == 64) {
        if (split3) QMRI_CONV_CASE(64, true); else QMRI_CONV_CASE(64, false);
    } else {
        if (split3) QMRI_CONV_CASE(32, true); else QMRI_CONV_CASE(32, false);
    }
#undef QMRI_CONV_CASE
    return hipGetLastError();
}

// ---- first layer: Conv2D(C, 3x3, SAME) on ONE input channel + bias + ReLU (fp32 VALU) ---------------
// x [B][H][W] fp32; w [9][Cout] fp32 (tap-major); y NHWC with pixel stride ldy / channel offset yoff.
// One thread = one pixel x 8 output channels.
__global__ __launch_bounds__(256) void conv3x3_c1_kernel(const float *__restrict__ x, int B, int H, int W,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ bias, int Cout,
                                                         float *__restrict__ y, long long ldy, int yoff) {
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
        float4 o0 = make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
        float4 o1 = make_float4(fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f));
        float *dst = y + pix * ldy + yoff + g * 8;
        *reinterpret_cast<float4 *>(dst) = o0;
        *reinterpret_cast<float4 *>(dst + 4) = o1;
    }
}

hipError_t conv3x3_c1_launch(const float *x, int B, int H, int W, const float *w, const float *bias,
                             int Cout, float *y, long long ldy, int yoff, hipStream_t stream) {
    const long long total = (long long)B * H * W * (Cout / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    (void)hipGetLastError();
    hipLaunchKernelGGL(conv3x3_c1_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, B, H, W, w, bias,
                       Cout, y, ldy, yoff);
    return hipGetLastError();
}

// ---- MaxPooling2D(2x2): NHWC fp32, input with pixel stride / channel offset, output compact ---------
__global__ __launch_bounds__(256) void maxpool2_kernel(const float *__restrict__ x, long long ldx, int xoff,
                                                       int B, int H, int W, int C, float *__restrict__ y) {
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
        const float *src = x + ((b * H + 2 * yo) * W + 2 * xo) * ldx + xoff + c;
        const float4 v00 = *reinterpret_cast<const float4 *>(src);
        const float4 v01 = *reinterpret_cast<const float4 *>(src + ldx);
        const float4 v10 = *reinterpret_cast<const float4 *>(src + (long long)W * ldx);
        const float4 v11 = *reinterpret_cast<const float4 *>(src + (long long)W * ldx + ldx);
        float4 o;
        o.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
        o.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
        o.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
        o.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
        *reinterpret_cast<float4 *>(y + p * C + c) = o;
    }
}

hipError_t maxpool2_launch(const float *x, long long ldx, int xoff, int B, int H, int W, int C, float *y,
                           hipStream_t stream) {
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    (void)hipGetLastError();
    hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx, xoff, B, H, W,
                       C, y);
    return hipGetLastError();
}

// ---- head: Conv2D(n_classes <= 4, 1x1) -> logits fp32 [pix][NC] and mask u8 [pix][NC] = logit > 0 -----
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ x, long long npix, int Cin,
                                                   const float *__restrict__ w /*[Cin][NC]*/,
                                                   const float *__restrict__ bias, int NC,
                                                   float *__restrict__ logits,
                                                   unsigned char *__restrict__ mask) {
    __shared__ float sw[256 * 4 + 4];
    for (int i = threadIdx.x; i < Cin * NC; i += blockDim.x) sw[i] = w[i];
    if (threadIdx.x < NC) sw[Cin * NC + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
         p += (long long)gridDim.x * blockDim.x) {
        float acc[4];
        for (int c = 0; c < 4; ++c) acc[c] = c < NC ? sw[Cin * NC + c] : 0.f;
        const float *src = x + p * Cin;
        for (int k = 0; k < Cin; k += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + k);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                for (int c = 0; c < NC; ++c) acc[c] = fmaf(vv[q], sw[(k + q) * NC + c], acc[c]);
        }
        for (int c = 0; c < NC; ++c) {
            if (logits) logits[p * NC + c] = acc[c];
            if (mask) mask[p * NC + c] = acc[c] > 0.f ? 1 : 0;  // sigmoid(z) > 0.5  <=>  z > 0
        }
    }
}

hipError_t head_launch(const float *x, long long npix, int Cin, const float *w, const float *bias, int NC,
                       float *logits, unsigned char *mask, hipStream_t stream) {
    if (NC > 4 || Cin > 256 || Cin % 4) return hipErrorInvalidValue;
    long long blocks = (npix + 255) / 256;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    (void)hipGetLastError();
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, npix, Cin, w, bias, NC,
                       logits, mask);
    return hipGetLastError();
}

// ---- whiten_volume (seg_model.py:114-127): two-pass mean / std in fp64, then (x - mean)/(std + eps) ---
__global__ __launch_bounds__(256) void sum_kernel(const float *__restrict__ x, long long n, double center,
                                                  int square, double *__restrict__ out) {
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const double d = (double)x[i] - center;
        acc += square ? d * d : d;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
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

__global__ void mean_from_sum_kernel(double *stats, long long n) { stats[2] = stats[0] / (double)n; }

__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ x, long long n,
                                                    double *__restrict__ stats) {
    const double mean = stats[2];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const double d = (double)x[i] - mean;
        acc += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(stats + 1, part[0] + part[1] + part[2] + part[3]);
}

hipError_t whiten_launch(const float *x, long long n, double eps, double *stats /*[3] device*/, float *y,
                         hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(stats, 0, 3 * sizeof(double), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, 0.0, 0, stats);
    hipLaunchKernelGGL(mean_from_sum_kernel, dim3(1), dim3(1), 0, stream, stats, n);
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, stats);
    hipLaunchKernelGGL(whiten_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, stats, eps, y);
    return hipGetLastError();
}

}  // namespace qmri
