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
