// linfit.hip -- per-voxel degree-1 least squares (slope, intercept, r2) for gfx950.
//
// Replaces the joint numpy.polyfit solve of the reference's polyfit()
//     /root/reference/dosma/core/fitting.py:974-984  (np.polyfit(x, y, 1) on the (E, N) matrix)
//     /root/reference/dosma/core/fitting.py:926-944  (_compute_r2_matrix)
//     /root/reference/dosma/core/fitting.py:1076-1103 (_polyfit: per-sequence skip rule)
// and, with log_transform = 1, the log-linearisation of MonoExponentialFit(tc0="polyfit")
//     /root/reference/dosma/core/fitting.py:710-715  (ints -> f32; v + 1e-10*(v==0); log)
//
// HBM-bound streaming kernel: lane i of a wave reads voxel i of each echo row (coalesced 256 B per
// wave-instruction, 4 voxels per lane in flight), closed-form 2x2 normal equations in fp64
// (x is centred, so the system is diagonal), one pass for r2.  Algorithmic bytes per voxel:
// E * sizeof(y) in, 24 B out (f64) / 12 B (f32).
#include <hip/hip_runtime.h>

#include <cmath>

#include "qmri_internal.h"

namespace qmri {

template <typename S>
__global__ __launch_bounds__(256) void linfit_kernel(const LinfitKArgs A) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < A.N; v += stride) {
        const S *col = static_cast<const S *>(A.y) + v;
        double sy = 0.0;
        bool allzero = true, oob = false;
        double lg[QMRI_MAX_ECHOES];
#pragma unroll 4
        for (int e = 0; e < A.E; ++e) {
            double s = static_cast<double>(col[(long long)e * A.ld]);
            allzero = allzero && s == 0.0;
            if (A.use_y_bounds) oob = oob || s < A.y_lo || s > A.y_hi;
            if (A.log_transform) {
                if (s == 0.0) s = 1e-10;
                s = log(s);
            }
            lg[e] = s;
            sy += s;
        }
        double slope = NAN, icpt = NAN, r2 = 0.0;
        if (!(A.skip_rules && (allzero || oob))) {
            const double ym = sy / (double)A.E;
            double sxy = 0.0, syy = 0.0;
            for (int e = 0; e < A.E; ++e) {
                const double dy = lg[e] - ym;
                sxy += (A.x[e] - A.xmean) * dy;
                syy += dy * dy;
            }
            slope = sxy / A.sxx;
            icpt = ym - slope * A.xmean;
            double ssr = 0.0;
            for (int e = 0; e < A.E; ++e) {
                const double r = (slope * A.x[e] + icpt) - lg[e];
                ssr += r * r;
            }
            r2 = 1.0 - ssr / (syy + A.r2_eps);
        }
        if (A.out_f64) {
            double2 o;
            o.x = slope;
            o.y = icpt;
            static_cast<double2 *>(A.popt)[v] = o;
            static_cast<double *>(A.r2)[v] = r2;
        } else {
            float2 o;
            o.x = (float)slope;
            o.y = (float)icpt;
            static_cast<float2 *>(A.popt)[v] = o;
            static_cast<float *>(A.r2)[v] = (float)r2;
        }
    }
}

// ---- general polynomial least squares with ONE design matrix for all voxels ------------------------------------------
// numpy.polyfit(x, Y, deg, rcond, full, w, cov) on the (E, N) matrix (reference polyfit(), fitting.py:873-1013) is a
// LINEAR map of every column: c = S @ y with S = (pinv of the weighted, column-scaled Vandermonde matrix) -- computed once
// on the host from x, w, rcond exactly as numpy builds it (dosma_amd/fitting.py::_polyfit_operator) -- so the per-voxel
// work is a (P x E) mat-vec, the fitted values D @ c for r2 (fitting.py:926-944, unweighted) and the weighted residual sum
// of squares numpy returns with full=True / uses for cov=True.  HBM-bound streaming kernel: E * sizeof(y) in,
// 8 (P + 1 [+ 1]) B out per voxel.  ops = [S (P x E) | D (E x P) | w (E)] doubles in device memory.
template <typename S>
__global__ __launch_bounds__(256) void polyls_kernel(const PolylsKArgs A) {
    __shared__ double ops[QMRI_MAX_ECHOES * QMRI_POLY_MAX_PARAMS * 2 + QMRI_MAX_ECHOES];
    const int P = A.P, E = A.E;
    for (int i = threadIdx.x; i < 2 * P * E + E; i += blockDim.x) ops[i] = A.ops[i];
    __syncthreads();
    const double *Sm = ops, *Dm = ops + P * E, *w = ops + 2 * P * E;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < A.N; v += stride) {
        const S *col = static_cast<const S *>(A.y) + v;
        // (loops are unrolled to the compile-time maxima with guards: a runtime trip count would put ys / c in scratch)
        double ys[QMRI_MAX_ECHOES];
        double sy = 0.0;
        bool allzero = true, oob = false;
#pragma unroll
        for (int e = 0; e < QMRI_MAX_ECHOES; ++e) {
            double s = 0.0;
            if (e < E) {
                s = static_cast<double>(col[(long long)e * A.ld]);
                allzero = allzero && s == 0.0;
                if (A.use_y_bounds) oob = oob || s < A.y_lo || s > A.y_hi;
                sy += s;
            }
            ys[e] = s;
        }
        const bool skip = A.skip_rules && (allzero || oob);
        double c[QMRI_POLY_MAX_PARAMS];
#pragma unroll
        for (int j = 0; j < QMRI_POLY_MAX_PARAMS; ++j) {
            double a = 0.0;
            if (j < P) {
#pragma unroll
                for (int e = 0; e < QMRI_MAX_ECHOES; ++e)
                    if (e < E) a = fma(Sm[j * E + e], ys[e], a);
            }
            c[j] = skip ? NAN : a;
        }
        double r2 = 0.0, wres = NAN;
        if (!skip) {
            const double ym = sy / (double)E;
            double ssr = 0.0, syy = 0.0, swr = 0.0;
#pragma unroll
            for (int e = 0; e < QMRI_MAX_ECHOES; ++e) {
                if (e < E) {
                    double yh = 0.0;
#pragma unroll
                    for (int j = 0; j < QMRI_POLY_MAX_PARAMS; ++j)
                        if (j < P) yh = fma(Dm[e * P + j], c[j], yh);
                    const double r = yh - ys[e], dy = ys[e] - ym, wr = w[e] * r;
                    ssr = fma(r, r, ssr);
                    syy = fma(dy, dy, syy);
                    swr = fma(wr, wr, swr);
                }
            }
            r2 = 1.0 - ssr / (syy + A.r2_eps);
            wres = swr;
        }
#pragma unroll
        for (int j = 0; j < QMRI_POLY_MAX_PARAMS; ++j)
            if (j < P) A.popt[v * P + j] = c[j];
        A.r2[v] = r2;
        if (A.resid) A.resid[v] = wres;
    }
}

// Any E and P (beyond the register-resident limits above): the same linear map with RUN-TIME loops.  S, D and w stay in
// global memory (every lane reads the same element: one broadcast line from L1 / L2), the samples of a voxel are re-read
// once per parameter and once for r2 (they are L2-resident: the block's columns of E rows), the coefficients go to their
// output row first and are read back for the fitted values.  O(P E) cached loads per voxel instead of E HBM loads: the
// route for rare sizes, not the hot one.
template <typename S>
__global__ __launch_bounds__(256) void polyls_big_kernel(const PolylsKArgs A) {
    const int P = A.P, E = A.E;
    const double *Sm = A.ops, *Dm = A.ops + (size_t)P * E, *w = A.ops + (size_t)2 * P * E;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < A.N; v += stride) {
        const S *col = static_cast<const S *>(A.y) + v;
        double sy = 0.0;
        bool allzero = true, oob = false;
        for (int e = 0; e < E; ++e) {
            const double s = static_cast<double>(col[(long long)e * A.ld]);
            allzero = allzero && s == 0.0;
            if (A.use_y_bounds) oob = oob || s < A.y_lo || s > A.y_hi;
            sy += s;
        }
        const bool skip = A.skip_rules && (allzero || oob);
        double *c = A.popt + v * P;
        for (int j = 0; j < P; ++j) {
            double a = 0.0;
            for (int e = 0; e < E; ++e) a = fma(Sm[(size_t)j * E + e], static_cast<double>(col[(long long)e * A.ld]), a);
            c[j] = skip ? NAN : a;
        }
        double r2 = 0.0, wres = NAN;
        if (!skip) {
            const double ym = sy / (double)E;
            double ssr = 0.0, syy = 0.0, swr = 0.0;
            for (int e = 0; e < E; ++e) {
                double yh = 0.0;
                for (int j = 0; j < P; ++j) yh = fma(Dm[(size_t)e * P + j], c[j], yh);
                const double ye = static_cast<double>(col[(long long)e * A.ld]);
                const double r = yh - ye, dy = ye - ym, wr = w[e] * r;
                ssr = fma(r, r, ssr);
                syy = fma(dy, dy, syy);
                swr = fma(wr, wr, swr);
            }
            r2 = 1.0 - ssr / (syy + A.r2_eps);
            wres = swr;
        }
        A.r2[v] = r2;
        if (A.resid) A.resid[v] = wres;
    }
}

hipError_t polyls_launch(const PolylsKArgs &k, int num_cu, hipStream_t stream) {
    long long blocks = (k.N + 255) / 256;
    const long long cap = (long long)num_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    (void)hipGetLastError();
    if (k.E > QMRI_MAX_ECHOES || k.P > QMRI_POLY_MAX_PARAMS) {
        switch (k.y_dtype) {
            case QMRI_F32: hipLaunchKernelGGL(polyls_big_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, k); break;
            case QMRI_F64: hipLaunchKernelGGL(polyls_big_kernel<double>, dim3((int)blocks), dim3(256), 0, stream, k); break;
            case QMRI_I16: hipLaunchKernelGGL(polyls_big_kernel<short>, dim3((int)blocks), dim3(256), 0, stream, k); break;
            default: hipLaunchKernelGGL(polyls_big_kernel<unsigned short>, dim3((int)blocks), dim3(256), 0, stream, k); break;
        }
        return hipGetLastError();
    }
    switch (k.y_dtype) {
        case QMRI_F32: hipLaunchKernelGGL(polyls_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, k); break;
        case QMRI_F64: hipLaunchKernelGGL(polyls_kernel<double>, dim3((int)blocks), dim3(256), 0, stream, k); break;
        case QMRI_I16: hipLaunchKernelGGL(polyls_kernel<short>, dim3((int)blocks), dim3(256), 0, stream, k); break;
        default: hipLaunchKernelGGL(polyls_kernel<unsigned short>, dim3((int)blocks), dim3(256), 0, stream, k); break;
    }
    return hipGetLastError();
}

hipError_t linfit_launch(const LinfitKArgs &k, int num_cu, hipStream_t stream) {
    long long blocks = (k.N + 255) / 256;
    const long long cap = (long long)num_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    switch (k.y_dtype) {
        case QMRI_F32: hipLaunchKernelGGL(linfit_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, k); break;
        case QMRI_F64: hipLaunchKernelGGL(linfit_kernel<double>, dim3((int)blocks), dim3(256), 0, stream, k); break;
        case QMRI_I16: hipLaunchKernelGGL(linfit_kernel<short>, dim3((int)blocks), dim3(256), 0, stream, k); break;
        default: hipLaunchKernelGGL(linfit_kernel<unsigned short>, dim3((int)blocks), dim3(256), 0, stream, k); break;
    }
    return hipGetLastError();
}

}  // namespace qmri
