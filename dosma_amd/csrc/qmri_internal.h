// qmri_internal.h -- shared between the HIP kernels and the C-ABI layer (not installed).
#ifndef QMRI_INTERNAL_H
#define QMRI_INTERNAL_H

#include <hip/hip_runtime.h>

#include "qmri.h"

#define QMRI_NO_ROUND (-1000000)

namespace qmri {

// Kernel-argument block of the fit kernel (passed by value: wave-uniform -> SGPR / scalar loads).
struct FitKArgs {
    const void *y;
    long long ld;
    long long N;
    int E;
    int y_dtype;
    int vec_ok;  // rows are 4-element aligned: 16-byte-per-lane staging loads are legal
    int init;
    int use_y_bounds;
    int refill_idle;  // idle lanes at which a wave pulls new voxels
    double y_lo, y_hi;
    const unsigned char *mask;
    const double *a0v;
    const double *b0v;
    double a0, b0;
    double ftol, xtol, gtol, factor, r2_eps;
    int maxfev;
    int out_f64;
    qmri_post post;
    double p10;  // 10 ** |post.decimals|
    void *popt;
    void *r2;
    void *tc;
    signed char *info;
    short *nfev;
    unsigned int *tile_counter;  // zeroed before every launch
    int *nonfinite;
    double xmean, sxx;           // of x: closed-form degree-1 least squares (log-linear init)
    double x[QMRI_MAX_ECHOES];
};

// Kernel-argument block of the degree-1 least-squares kernel (linfit.hip).
struct LinfitKArgs {
    const void *y;
    long long ld;
    long long N;
    int E;
    int y_dtype;
    int log_transform;
    int skip_rules;
    int use_y_bounds;
    int out_f64;
    double y_lo, y_hi;
    double r2_eps;
    void *popt;
    void *r2;
    double xmean, sxx;
    double x[QMRI_MAX_ECHOES];
};
hipError_t linfit_launch(const LinfitKArgs &k, int num_cu, hipStream_t stream);

int monoexp_tile_voxels();
const char *monoexp_variant_name(int E, int y_dtype);
int monoexp_blocks_per_cu(const FitKArgs &k);
int monoexp_waves_per_block(const FitKArgs &k);
hipError_t monoexp_launch(const FitKArgs &k, int grid, hipStream_t stream);

}  // namespace qmri
#endif
