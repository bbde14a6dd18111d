// dess.hip -- analytic T2 from the two qDESS echoes, and echo combination (RSS / RMS), for gfx950.
//
// the elementwise step that precedes segmentation / fitting in
//     /root/reference/dosma/scan_sequences/mri/qdess.py:105-252 (QDess.generate_t2_map)
//     /root/reference/dosma/scan_sequences/mri/qdess.py:254-295 (calc_rss / _combine_echoes)
// The reference evaluates it with ~10 whole-volume numpy passes (ratio, nan_to_num, log, divide,
// nan_to_num, bounds, nan_to_num, around, two suppression masks); here it is ONE streaming pass
// (+ one max-reduction pass when fat / fluid suppression needs a whole-volume maximum).
// HBM-bound: algorithmic bytes per voxel = 2 * sizeof(echo) in + sizeof(out) out (16 B for f32 -> f64).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include <cstdlib>

#include "fp64_fast.h"
#include "qmri_internal.h"

namespace qmri {

__device__ __forceinline__ double nan_to_num_d(double v, double nanv) {
    if (isnan(v)) return nanv;
    if (isinf(v)) return v > 0 ? DBL_MAX : -DBL_MAX;
    return v;
}

// ---- whole-volume maxima of echo1 and of (echo1 - beta * echo2), in the arithmetic numpy would use:
// float32 for float32 echoes (python scalars are weak), float64 otherwise ------------------------------
template <typename S>
struct NfType { using type = double; };
template <>
struct NfType<float> { using type = float; };

template <typename S>
__global__ __launch_bounds__(256) void dess_max_kernel(const S *__restrict__ e1, const S *__restrict__ e2,
                                                       long long n, double beta, double *__restrict__ part) {
    using NF = typename NfType<S>::type;
    double m1 = -INFINITY, m2 = -INFINITY;
    const NF b = static_cast<NF>(beta);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const NF a = static_cast<NF>(e1[i]);
        const NF nf = a - b * static_cast<NF>(e2[i]);
        m1 = fmax(m1, (double)a);
        m2 = fmax(m2, (double)nf);
    }
    for (int o = 32; o > 0; o >>= 1) {
        m1 = fmax(m1, __shfl_down(m1, o, 64));
        m2 = fmax(m2, __shfl_down(m2, o, 64));
    }
    __shared__ double s1[4], s2[4];
    if ((threadIdx.x & 63) == 0) {
        s1[threadIdx.x >> 6] = m1;
        s2[threadIdx.x >> 6] = m2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmax(fmax(s1[0], s1[1]), fmax(s1[2], s1[3]));
        part[2 * blockIdx.x + 1] = fmax(fmax(s2[0], s2[1]), fmax(s2[2], s2[3]));
    }
}

__global__ void dess_max_final_kernel(const double *__restrict__ part, int nblocks, double *__restrict__ out) {
    double m1 = -INFINITY, m2 = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        m1 = fmax(m1, part[2 * i]);
        m2 = fmax(m2, part[2 * i + 1]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        m1 = fmax(m1, __shfl_down(m1, o, 64));
        m2 = fmax(m2, __shfl_down(m2, o, 64));
    }
    if (threadIdx.x == 0) {
        out[0] = m1;
        out[1] = m2;
    }
}

// one voxel of qdess.py:218-245
template <typename S>
__device__ __forceinline__ double dess_voxel(const DessKArgs &A, S s1, S s2, typename NfType<S>::type fat_thr,
                                             typename NfType<S>::type fluid_thr,
                                             typename NfType<S>::type beta) {
    using NF = typename NfType<S>::type;
    const double e1 = static_cast<double>(s1), e2 = static_cast<double>(s2);
    double ratio = nan_to_num_d(div_fast(e2, e1), 0.0);        // :218-219
    double t2 = div_fast(A.c0, log_fast(div_fast(fabs(ratio), A.k)) + A.c1);  // :222
    t2 = nan_to_num_d(t2, 0.0);                                // :224
    if (A.use_bounds && (t2 < A.lo || t2 > A.hi)) t2 = NAN;    // :227-229
    if (A.use_nan_to_num) t2 = nan_to_num_d(t2, A.nan_value);  // :230-235
    if (A.decimals != QMRI_NO_ROUND) {                         // :237-238
        if (A.decimals == 0) t2 = rint(t2);
        else if (A.decimals > 0) t2 = div_p10(rint(t2 * A.p10), A.p10, A.ip10);
        else t2 = rint(div_p10(t2, A.p10, A.ip10)) * A.p10;
    }
    if (A.suppress_fat) t2 = t2 * (static_cast<NF>(s1) > fat_thr ? 1.0 : 0.0);  // :240-241
    if (A.suppress_fluid) {                                                       // :243-245
        const NF nf = static_cast<NF>(s1) - beta * static_cast<NF>(s2);
        t2 = t2 * (nf > fluid_thr ? 1.0 : 0.0);
    }
    return t2;
}

// 4 consecutive voxels per lane: 16-byte (f32) echo loads, 2 x 16-byte (f64) stores -> full-width HBM
// transactions; a scalar tail handles N % 4 and unaligned bases.
template <typename S>
__global__ __launch_bounds__(256) void dess_t2_kernel(const DessKArgs A) {
    using NF = typename NfType<S>::type;
    const S *e1p = static_cast<const S *>(A.echo1);
    const S *e2p = static_cast<const S *>(A.echo2);
    NF fat_thr = 0, fluid_thr = 0;
    if (A.suppress_fat) fat_thr = static_cast<NF>(0.15) * static_cast<NF>(A.maxima[0]);
    if (A.suppress_fluid) fluid_thr = static_cast<NF>(0.1) * static_cast<NF>(A.maxima[1]);
    const NF beta = static_cast<NF>(A.beta);
    struct alignas(sizeof(S) * 4) V4 {
        S v[4];
    };
    const bool vec = A.vec_ok;
    const long long n4 = vec ? A.N / 4 : 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // software pipeline: the next iteration's two 16-byte loads are in flight while this one's log / divisions run
    long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    V4 a = {}, b = {};
    if (q < n4) {
        a = reinterpret_cast<const V4 *>(e1p)[q];
        b = reinterpret_cast<const V4 *>(e2p)[q];
    }
    while (q < n4) {
        const long long qn = q + stride;
        V4 an = a, bn = b;
        if (qn < n4) {
            an = reinterpret_cast<const V4 *>(e1p)[qn];
            bn = reinterpret_cast<const V4 *>(e2p)[qn];
        }
        double o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = dess_voxel<S>(A, a.v[j], b.v[j], fat_thr, fluid_thr, beta);
        if (A.out_f64) {
            double *dst = static_cast<double *>(A.t2) + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(o[j], dst + j);
        } else {
            float *dst = static_cast<float *>(A.t2) + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_nontemporal_store((float)o[j], dst + j);
        }
        a = an;
        b = bn;
        q = qn;
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.N; i += stride) {
        const double t2 = dess_voxel<S>(A, e1p[i], e2p[i], fat_thr, fluid_thr, beta);
        if (A.out_f64) static_cast<double *>(A.t2)[i] = t2;
        else static_cast<float *>(A.t2)[i] = static_cast<float>(t2);
    }
}

template <typename S>
__global__ __launch_bounds__(256) void rss_kernel(const S *__restrict__ e1, const S *__restrict__ e2,
                                                  long long n, int rms, double *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const double a = static_cast<double>(e1[i]), b = static_cast<double>(e2[i]);
        const double s = a * a + b * b;  // qdess.py:283-286 (float64)
        out[i] = sqrt(rms ? s / 2 : s);
    }
}

static int grid_for(long long n, int num_cu) {
    long long blocks = (n + 255) / 256;
    const long long cap = (long long)num_cu * 8;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : (int)blocks;
}

#define QMRI_BY_DTYPE(dt, CALL)              \
    switch (dt) {                            \
        case QMRI_F32: { using S = float; CALL; } break;           \
        case QMRI_F64: { using S = double; CALL; } break;          \
        case QMRI_I16: { using S = short; CALL; } break;           \
        default: { using S = unsigned short; CALL; } break;        \
    }

hipError_t dess_t2_launch(const DessKArgs &k, int dtype, int num_cu, double *scratch /*[2*1024+2]*/,
                          hipStream_t stream) {
    (void)hipGetLastError();
    DessKArgs a = k;
    if (k.suppress_fat || k.suppress_fluid) {
        int nb = grid_for(k.N, num_cu);
        if (nb > 1024) nb = 1024;
        QMRI_BY_DTYPE(dtype, hipLaunchKernelGGL(dess_max_kernel<S>, dim3(nb), dim3(256), 0, stream,
                                                static_cast<const S *>(k.echo1), static_cast<const S *>(k.echo2),
                                                k.N, k.beta, scratch + 2));
        hipLaunchKernelGGL(dess_max_final_kernel, dim3(1), dim3(64), 0, stream, scratch + 2, nb, scratch);
        a.maxima = scratch;
    }
    int nb = grid_for(k.N, num_cu);
    static const int bpc = [] { const char *e = getenv("QMRI_DESS_BPC"); return e ? atoi(e) : 8; }();
    if (nb > num_cu * bpc) nb = num_cu * bpc;  // measured: 4 / 6 blocks per CU 0.149 ms, 8 / 12 0.141 ms per 23.6 M voxels
    QMRI_BY_DTYPE(dtype, hipLaunchKernelGGL(dess_t2_kernel<S>, dim3(nb), dim3(256), 0, stream, a));
    return hipGetLastError();
}

hipError_t rss_launch(const void *e1, const void *e2, int dtype, long long n, int rms, double *out, int num_cu,
                      hipStream_t stream) {
    (void)hipGetLastError();
    const int nb = grid_for(n, num_cu);
    QMRI_BY_DTYPE(dtype, hipLaunchKernelGGL(rss_kernel<S>, dim3(nb), dim3(256), 0, stream,
                                            static_cast<const S *>(e1), static_cast<const S *>(e2), n, rms, out));
    return hipGetLastError();
}

}  // namespace qmri
