/*
 * qmri.h -- C ABI of the MI355X-native qMRI hot path (libqmri_hip.so, built from dosma_amd/csrc).
 *
 * The reference (ad12/DOSMA) is pure Python and has no FFI: the seam this library plugs into is
 * the Python-level method  _Fitter._fit(x, y, **kw) -> (popt, r2)
 *     /root/reference/dosma/core/fitting.py:148-155 (abstract), :422-435 (CurveFitter override)
 * which forwards to the module function  curve_fit(func, x, y, ...)
 *     /root/reference/dosma/core/fitting.py:755-870
 * whose body is a per-voxel Python loop over scipy.optimize.curve_fit (:855-868 -> :1026-1073).
 * Everything below replaces exactly that loop (+ the elementwise plumbing around it that
 * MonoExponentialFit.fit adds, :701-737), for func == monoexponential (:1016-1018).
 *
 * Conventions
 *   - plain C types only; every buffer is caller-owned; no torch / numpy types cross this line;
 *   - every entry returns 0 on success or a negative qmri_status; qmri_last_error() gives the
 *     message for the calling thread;
 *   - "device" pointers are HIP device pointers on args->device; "host" pointers are ordinary
 *     process memory;  x (the E sample positions) is ALWAYS a host pointer (E <= QMRI_MAX_ECHOES);
 *   - no global mutable state besides one lazily created context per device (a 64-byte tile
 *     counter); calls on different streams / devices may run concurrently from different threads.
 */
#ifndef QMRI_H
#define QMRI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMRI_VERSION 100 /* 0.1.0 */
#define QMRI_MAX_ECHOES 32
/* qmri_lmfit_*: samples per voxel of the general lmdif kernel.  Its per-lane columns ((n + 3) E doubles) live in LDS:
 * (n + 3) * E * 512 B <= 160 KB, i.e. E <= 64 for the 2-parameter mono-exponential, E <= 45 for the bi-exponential.  The
 * mono-exponential entry (qmri_monoexp_fit_*) keeps its samples in registers: E <= QMRI_MAX_ECHOES; beyond that the
 * Python mirror routes CurveFitter / MonoExponentialFit through qmri_lmfit_* (the reference has no limit, fitting.py:755-870). */
#define QMRI_LM_MAX_ECHOES 64

typedef enum qmri_status {
    QMRI_OK = 0,
    QMRI_ERR_ARG = -1,         /* bad argument (shape, dtype, NULL)        -> ValueError          */
    QMRI_ERR_UNSUPPORTED = -2, /* valid request this build does not cover -> NotImplementedError */
    QMRI_ERR_HIP = -3,         /* HIP runtime error / no device            -> RuntimeError        */
    QMRI_ERR_NONFINITE = -4,   /* y holds NaN/Inf: the reference raises ValueError for the whole
                                  call (scipy check_finite, via fitting.py:1030)                   */
    QMRI_ERR_NOMEM = -5        /* device memory exhausted (hipErrorOutOfMemory): the caller may
                                  retry with a smaller batch                   -> MemoryError        */
} qmri_status;

/* element type of y (what a MedicalVolume of DICOM / NIfTI data holds) */
typedef enum qmri_dtype {
    QMRI_F32 = 0,
    QMRI_F64 = 1,
    QMRI_I16 = 2,
    QMRI_U16 = 3
} qmri_dtype;

/* initial guess p0 = (a0, b0) of  y = a * exp(b * x) */
typedef enum qmri_init {
    QMRI_INIT_SCALAR = 0,    /* (a0, b0) for every voxel: MonoExponentialFit(tc0=float) -> (1, -1/tc0)
                                fitting.py:720; curve_fit(p0=tuple) fitting.py:822-825            */
    QMRI_INIT_PER_VOXEL = 1, /* a0v[N] and/or b0v[N] (NULL -> scalar): curve_fit(p0=arrays),
                                fitting.py:849-851, 1039-1053                                      */
    QMRI_INIT_LOGLIN = 2     /* log-linear least squares computed on-chip:
                                MonoExponentialFit(tc0="polyfit"), fitting.py:701-718             */
} qmri_init;

/*
 * Post-processing of the (N,2) parameter array, _Fitter._process_params (fitting.py:109-146), in the
 * reference's order:  ufunc (b -> 1/|b|, fitting.py:725)  ->  bounds (value < lb or > ub -> NaN)
 * ->  r2 < threshold -> whole row NaN  ->  nan_to_num (NaN -> value, +-inf -> +-DBL_MAX), then the
 * scatter fill for voxels outside the mask (fitting.py:205-215) and np.around of the tc map (:736-737).
 */
#define QMRI_NO_ROUND (-1000000)
typedef struct qmri_post {
    int32_t enable;         /* 0: popt/r2 are curve_fit()'s raw outputs                              */
    int32_t inv_abs_b;      /* 1: param 1 <- 1/|b|                                                   */
    int32_t use_bounds;     /* 1: apply lb/ub below                                                  */
    int32_t use_r2_thr;     /* 1: rows with r2 < r2_threshold -> NaN                                 */
    int32_t use_nan_to_num; /* 1: np.nan_to_num(x, nan=nan_value); also the fill outside the mask    */
    int32_t decimals;       /* tc = np.around(param 1, decimals) written to args->tc (numpy semantics:
                               negative = tens, hundreds ...); QMRI_NO_ROUND: tc is not rounded      */
    double lb[2], ub[2];    /* per-parameter bounds (use -inf/+inf for "none")                       */
    double r2_threshold;
    double nan_value;
} qmri_post;

typedef struct qmri_monoexp_args {
    /* ---- inputs ---- */
    const void *y;       /* [E][ld] echo-major samples (the (E, N) C-contiguous array of fitting.py:194-196) */
    int32_t y_dtype;     /* qmri_dtype                                                                */
    int32_t E;           /* samples per voxel, 2 <= E <= QMRI_MAX_ECHOES                              */
    int64_t N;           /* voxels                                                                    */
    int64_t ld;          /* elements between consecutive echoes (>= N)                                */
    const double *x;     /* HOST [E] sample positions (echo / spin-lock times)                        */
    const uint8_t *mask; /* nullable [N]: fit only voxels with mask != 0 (fitting.py:107, 199-200)    */
    int32_t init;        /* qmri_init                                                                 */
    int32_t use_y_bounds;/* 1: voxels with a sample < y_lo or > y_hi are skipped like all-zero ones
                            (curve_fit(y_bounds=...), fitting.py:1064-1067)                           */
    double y_lo, y_hi;
    double a0, b0;       /* scalar initial guess                                                      */
    const double *a0v;   /* nullable [N] (indexed by voxel, not by position in the mask)              */
    const double *b0v;   /* nullable [N]                                                              */
    /* ---- solver constants (fitting.py:761-763 + scipy.optimize.leastsq defaults) ---- */
    double ftol;         /* 1e-5  (DOSMA)                                                             */
    double xtol;         /* 1.49012e-8 (scipy)                                                        */
    double gtol;         /* 0.0   (scipy)                                                             */
    double factor;       /* 100.0 (scipy)                                                             */
    double r2_eps;       /* 1e-8  (DOSMA, fitting.py:752)                                             */
    int32_t maxfev;      /* 100   (DOSMA)                                                             */
    int32_t reserved1;
    qmri_post post;
    /* ---- outputs ---- */
    void *popt;          /* [N][2] (a, b)  -- or (a, tc) after post.inv_abs_b; nullable when tc is given
                            (MonoExponentialFit needs only tc and r2: a third less output traffic)    */
    void *r2;            /* [N]                                                                       */
    void *tc;            /* nullable [N]: time-constant map = processed param 1, rounded per post.decimals */
    int32_t out_dtype;   /* QMRI_F32 or QMRI_F64 (the reference returns float64)                      */
    int32_t reserved2;
    int8_t *info;        /* nullable [N]: MINPACK info (1..4 success, 5..8 failure), 0 = skipped
                            (all-zero voxel, fitting.py:1065-1067), -1 = outside mask                 */
    int16_t *nfev;       /* nullable [N]: function evaluations as lmdif counts them                   */
    /* ---- placement ---- */
    int32_t device;      /* HIP device ordinal                                                        */
    int32_t reserved3;
    void *stream;        /* hipStream_t (NULL = default stream); the call is asynchronous on it       */
    /* ---- host entry only ---- */
    const void *const *y_rows; /* nullable HOST [E]: echo e is the contiguous array y_rows[e][0..N) and y / ld are
                            ignored -- the E MedicalVolumes are fitted where they lie, without the 4 E N-byte
                            np.concatenate of fitting.py:194-196                                       */
} qmri_monoexp_args;

/* Fill the solver constants and post block with the reference's defaults. */
void qmri_monoexp_defaults(qmri_monoexp_args *args);

/*
 * Per-voxel mono-exponential Levenberg-Marquardt fit on the GPU; all data pointers are DEVICE
 * pointers (except x).  Asynchronous on args->stream.  Replaces the loop at fitting.py:855-868.
 * If `nonfinite_flag` (device int32, nullable) is given it is set to 1 when a fitted voxel holds a
 * non-finite sample; the caller maps that to the reference's ValueError.
 */
int qmri_monoexp_fit_device(const qmri_monoexp_args *args, int32_t *nonfinite_flag);

/*
 * Same, but y / mask / a0v / b0v / popt / r2 / tc / info / nfev are HOST pointers: the library
 * stages them through device memory in slabs (H2D, fit, D2H overlapped on two streams) and returns
 * when the outputs are complete.  Returns QMRI_ERR_NONFINITE like the reference's ValueError.
 * This is what the Python drop-in (dosma_amd.fitting) calls for CPU MedicalVolumes.
 */
int qmri_monoexp_fit_host(const qmri_monoexp_args *args);

/*
 * Degree-1 least squares per voxel: popt[N][2] = (slope, intercept) in numpy.polyfit order, r2[N].
 * Replaces polyfit(x, y, 1) of the reference (fitting.py:873-1013; joint solve :974-984, r2 :926-944).
 *   log_transform 1: fit log(v + 1e-10*(v==0)) -- the log-linearisation of
 *                    MonoExponentialFit(tc0="polyfit") (fitting.py:710-715);
 *   skip_rules    1: all-zero / out-of-y_bounds voxels -> (NaN, NaN), r2 = 0 (fitting.py:1095-1097,
 *                    the per-sequence branch taken when num_workers is not None).
 */
typedef struct qmri_linfit_args {
    const void *y;       /* [E][ld] echo-major */
    int32_t y_dtype;     /* qmri_dtype */
    int32_t E;           /* 2 <= E <= QMRI_MAX_ECHOES */
    int64_t N;
    int64_t ld;
    const double *x;     /* HOST [E] */
    int32_t log_transform;
    int32_t skip_rules;
    int32_t use_y_bounds;
    int32_t out_dtype;   /* QMRI_F32 | QMRI_F64 */
    double y_lo, y_hi;
    double r2_eps;       /* 1e-8 */
    void *popt;          /* [N][2] */
    void *r2;            /* [N] */
    int32_t device;
    int32_t reserved;
    void *stream;
} qmri_linfit_args;

int qmri_linfit_device(const qmri_linfit_args *args); /* device pointers, asynchronous on args->stream */
int qmri_linfit_host(const qmri_linfit_args *args);   /* host pointers, synchronous */

/*
 * ---- General polynomial least squares: numpy.polyfit(x, Y, deg, rcond, full, w, cov) per voxel -------------------
 * Replaces the joint solve / per-sequence loop of the reference's polyfit() for any degree and for the numpy options the
 * degree-1 kernel above does not take (/root/reference/dosma/core/fitting.py:873-1013; _polyfit :1076-1103).  The fit of
 * every column is the SAME linear map of its samples, so the caller builds it once from x, w and rcond the way numpy does
 * (weighted Vandermonde matrix, column scaling, SVD with the rcond cut-off) and hands over
 *     solve  [P][E]   coefficients (highest power first) = solve @ y
 *     design [E][P]   fitted values = design @ coefficients (the unweighted Vandermonde matrix; r2 as fitting.py:926-944)
 *     w      [E]      weights of the residual sum of squares numpy reports with full=True / scales cov with (NULL = ones)
 * Outputs: popt [N][P], r2 [N], resid [N] (nullable): sum_e (w_e (fit_e - y_e))^2.
 */
#define QMRI_POLY_MAX_PARAMS 8
typedef struct qmri_polyls_args {
    const void *y;        /* [E][ld] echo-major */
    int32_t y_dtype;      /* qmri_dtype */
    int32_t E;            /* P <= E <= QMRI_MAX_ECHOES */
    int64_t N;
    int64_t ld;
    int32_t P;            /* deg + 1 <= QMRI_POLY_MAX_PARAMS */
    int32_t skip_rules;   /* 1: all-zero / out-of-y_bounds columns -> (NaN..), r2 = 0 (the per-sequence branch, :1095-1097) */
    const double *solve;  /* HOST [P][E] */
    const double *design; /* HOST [E][P] */
    const double *w;      /* HOST [E] or NULL */
    int32_t use_y_bounds;
    int32_t device;
    double y_lo, y_hi;
    double r2_eps;        /* 1e-8 */
    double *popt;         /* [N][P] float64 */
    double *r2;           /* [N] */
    double *resid;        /* nullable [N] */
    void *stream;
} qmri_polyls_args;
int qmri_polyls_device(const qmri_polyls_args *args); /* y / popt / r2 / resid device pointers, asynchronous */
int qmri_polyls_host(const qmri_polyls_args *args);   /* host pointers, synchronous */

/*
 * ---- General Levenberg-Marquardt fit (models other than the mono-exponential hot path) -------------------
 * Replaces curve_fit(func, x, y, p0, ftol=1e-5, maxfev=100) of the reference
 *   (/root/reference/dosma/core/fitting.py:755-870; per-voxel wrapper :1026-1073) for
 *     QMRI_MODEL_BIEXP    func = biexponential(x, a1, b1, a2, b2) = a1 e^{b1 x} + a2 e^{b2 x}   (:1021-1023)
 *     QMRI_MODEL_MONOEXP  func = monoexponential (:1016-1018) with TRUE forward differences (cross-check of
 *                         qmri_monoexp_fit_*, which emulates them; ~3x the exponentials)
 * i.e. scipy leastsq -> MINPACK lmdif (fdjac2 forward differences, n + 1 model evaluations per iteration).
 * Same skip / failure / r2 rules as qmri_monoexp_fit_*; popt is [N][n] float64, n = 2 or 4.
 */
typedef enum qmri_model { QMRI_MODEL_MONOEXP = 0, QMRI_MODEL_BIEXP = 1 } qmri_model;
#define QMRI_LM_MAX_PARAMS 4
typedef struct qmri_lmfit_args {
    int32_t model;       /* qmri_model */
    int32_t y_dtype;     /* qmri_dtype */
    const void *y;       /* [E][ld] echo-major */
    int32_t E;           /* n <= E <= QMRI_LM_MAX_ECHOES and (n + 3) * E <= 320 (LDS) */
    int32_t maxfev;      /* 100 */
    int64_t N;
    int64_t ld;
    const double *x;     /* HOST [E] */
    double p0[QMRI_LM_MAX_PARAMS];          /* scalar initial guess (scipy: ones) */
    const double *p0v[QMRI_LM_MAX_PARAMS];  /* nullable per-voxel initial guess [N] per parameter */
    double ftol, xtol, gtol, factor, epsfcn, r2_eps;
    int32_t use_y_bounds;
    int32_t device;
    double y_lo, y_hi;
    double *popt;        /* [N][n] */
    double *r2;          /* [N] */
    int8_t *info;        /* nullable [N]: MINPACK info (0 = skipped) */
    int16_t *nfev;       /* nullable [N] */
    void *stream;
} qmri_lmfit_args;
void qmri_lmfit_defaults(qmri_lmfit_args *args); /* zero + the reference's constants; model = BIEXP */
/* nonfinite_flag: device int32, set to 1 if any sample is not finite (scipy check_finite -> ValueError) */
int qmri_lmfit_device(const qmri_lmfit_args *args, int32_t *nonfinite_flag);
int qmri_lmfit_host(const qmri_lmfit_args *args); /* host pointers, synchronous; QMRI_ERR_NONFINITE */

/*
 * ---- 2D U-Net segmentation (IWOAIOAIUnet2D / IWOAIOAIUnet2DNormalized) --------------------------------
 * Replaces `model.predict(v, batch_size)` inside SegModel.generate_mask
 *   (/root/reference/dosma/models/oaiunet2d.py:305; graph :197-289; whitening seg_model.py:114-127).
 * Weights are handed over ONCE as host fp32 arrays in Keras layouts and Keras layer-creation order
 * (what `load_weights` of the .h5 iterates over):
 *   for level l = 0 .. depth-1 :  conv1.kernel (3,3,Cin,C) conv1.bias (C)  conv2.kernel (3,3,C,C) conv2.bias
 *                                 bn.gamma bn.beta bn.moving_mean bn.moving_variance            (8 tensors)
 *   for level l = depth-2 .. 0 :  deconv.kernel (3,3,C,Cup) deconv.bias  conv1.kernel (3,3,2C,C) conv1.bias
 *                                 conv2.kernel conv2.bias  bn.gamma bn.beta bn.mean bn.var      (10 tensors)
 *   head.kernel (1,1,C0,n_classes) head.bias                                                    (2 tensors)
 */
typedef struct qmri_unet2d_desc {
    int32_t depth;          /* 6 */
    int32_t base_features;  /* 32: level l has base_features << l channels */
    int32_t n_classes;      /* 4 (fc, tc, pc, men) */
    int32_t H, W;           /* slice size; multiples of 2^(depth-1) */
    int32_t max_batch;      /* slices per pass through the network (the reference's batch_size) */
    int32_t precision;      /* 0: "bf16" -- bf16 operands, 1 MFMA per product (logits within ~0.25);
                             * 1: "fp16x3" -- the parity mode: fp16 hi + lo operand parts, hi*hi + hi*lo + lo*hi on MFMA,
                             *    fp32 accumulate, activations kept split in HBM (logits within 1e-3 of an fp64 run) */
    int32_t device;
    const float *const *tensors; /* HOST pointers, order above */
    int32_t n_tensors;
    int32_t reserved;
    double bn_eps;          /* BatchNormalization epsilon: 1e-3 (oaiunet2d.py:228) */
} qmri_unet2d_desc;

int qmri_unet2d_create(const qmri_unet2d_desc *desc, void **handle);
int qmri_unet2d_set_precision(void *handle, int32_t precision);
/* Which kernel family ran every layer of the last forward batch ("down1.conv2:s3/2d/bn64+pool;..."): returns the length
 * of the full string, writes at most size - 1 characters + NUL.  For tests that assert the dispatch. */
int qmri_unet2d_trace(void *handle, char *buf, int32_t size);
/*
 * x [S][H][W] fp32 (host or device), optional whole-volume whitening (x - mean)/(std + eps) first.
 * Outputs (nullable): logits [S][H][W][n_classes] fp32 (pre-sigmoid), mask u8 = sigmoid > 0.5.
 * Parity mode: feature maps are stored as fp16 hi + lo parts.  Input without whitening (raw intensities, the reference's
 * IWOAIOAIUnet2D: oaiunet2d.py:322-323) is brought below 128 by a power of two first and the network's additive parameters
 * are scaled with it (exact: the logits are unchanged); if a feature map still leaves the fp16 range the kernels flag it and
 * the forward is REPEATED at a higher exponent -- never stored clamped.  The call therefore returns with the stream
 * synchronised (also with device pointers); after four attempts it fails with QMRI_ERR_UNSUPPORTED and says so.  The
 * exponent the last forward ran at is the "act_shift:N" entry of qmri_unet2d_trace.
 */
int qmri_unet2d_forward(void *handle, const float *x, int32_t S, int32_t x_on_device, int32_t whiten,
                        double whiten_eps, float *logits, uint8_t *mask, int32_t out_on_device,
                        void *stream);
/*
 * Whole-volume segmentation in the reference's own array layouts, host pointers, synchronous:
 *   vol_hws    [H][W][S] fp32 -- MedicalVolume.volume in the sagittal orientation; the reference transposes it to
 *              (S, H, W, 1) for model.predict (oaiunet2d.py:295-303)
 *   mask_chws  [n_classes][H][W][S] uint8 -- one volume per class; the reference transposes each class of
 *              (model.predict(...) > 0.5) back to (H, W, S) (:306-316)
 * Both transposes, the optional whitening and the threshold run on the GPU: one upload, one download.
 */
int qmri_unet2d_segment_volume(void *handle, const float *vol_hws, int32_t S, int32_t whiten, double whiten_eps,
                               uint8_t *mask_chws, void *stream);
int qmri_unet2d_destroy(void *handle);
/*
 * One layer on host NHWC fp32 arrays (operator-level entry; also what the unit tests drive):
 *   transposed 0: Conv2D(Cout, 3x3, padding=same), kernel (3,3,Cin,Cout)       oaiunet2d.py:213-226
 *   transposed 1: Conv2DTranspose(Cout, 3x3, strides=2, padding=same), kernel (3,3,Cout,Cin) :259-261
 *   y = scale * relu?(conv + bias) + shift   (scale/shift nullable: the folded BatchNormalization)
 *   precision 0 / 1 as in qmri_unet2d_desc; 2 = the parity mode forced onto the general kernel (conv_igemm) for shapes
 *   that would otherwise run on conv_s3_kernel (tests compare the two)
 */
int qmri_conv2d_nhwc_host(const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float *kernel,
                          const float *bias, const float *scale, const float *shift, int32_t relu,
                          int32_t Cout, int32_t transposed, int32_t precision, float *y, int32_t device);

/*
 * ---- analytic T2 of a qDESS scan and echo combination -----------------------------
 * Replaces the whole-volume numpy passes of QDess.generate_t2_map
 *   (/root/reference/dosma/scan_sequences/mri/qdess.py:105-252; arithmetic :204-245) and calc_rss (:254-295).
 *   t2 = c0 / (log(|echo2/echo1| / k) + c1) with the reference's nan_to_num / bounds / rounding / suppression
 *   steps; c0 = -2000 (TR - TE), k and c1 are the scalar sequence constants of qdess.py:204-212, computed
 *   by the caller (dosma_amd/scan_sequences/qdess.py does it with the reference's expressions).
 */
typedef struct qmri_dess_args {
    const void *echo1;    /* [N] */
    const void *echo2;    /* [N] */
    int32_t dtype;        /* qmri_dtype of the echoes */
    int32_t out_dtype;    /* QMRI_F64 (reference) or QMRI_F32 */
    int64_t N;
    double c0, k, c1;
    int32_t use_bounds;   /* nan_bounds: values outside [lo, hi] -> NaN */
    int32_t use_nan_to_num;
    double lo, hi, nan_value;
    int32_t decimals;     /* <= -1000000: no rounding */
    int32_t suppress_fat;   /* t2 *= echo1 > 0.15 max(echo1) */
    int32_t suppress_fluid; /* t2 *= (echo1 - beta echo2) > 0.1 max(echo1 - beta echo2) */
    int32_t device;
    double beta;
    void *t2;             /* [N] out */
    void *stream;
} qmri_dess_args;
int qmri_dess_t2_device(const qmri_dess_args *args); /* device pointers, asynchronous */
int qmri_dess_t2_host(const qmri_dess_args *args);   /* host pointers, synchronous */
/* out[i] = sqrt(e1^2 + e2^2) (mode 0, RSS) or sqrt((e1^2 + e2^2)/2) (mode 1, RMS), float64 like the reference */
int qmri_rss_host(const void *echo1, const void *echo2, int32_t dtype, int64_t N, int32_t mode, double *out,
                  int32_t device);

/* ------------------------------------------------------------------------------------------------------------------
 * Region statistics of a quantitative map: SURVEY.md 8(f) row N1, the masked reductions of
 * QuantitativeValue.to_metrics (/root/reference/dosma/core/quant_vals.py:145-229): for every region
 * (label_keys[r], and a last region "total" = every voxel with a positive label; without labels: every voxel)
 * the count, mean (nanmean), population standard deviation (nanstd) and median (nanmedian: the middle element,
 * or the mean of the two middle elements) of the voxels that are finite and inside `bounds`.
 * The median is exact (radix selection on the order-preserving integer image of the double), mean / std are fp64
 * sums (two passes like numpy's nanstd; the summation order differs from numpy's pairwise sums: ~1e-15 relative). */
typedef struct qmri_region_stats_args {
    const void *values;        /* [N] map */
    int32_t v_dtype;           /* QMRI_F32 | QMRI_F64 */
    const void *labels;        /* [N] label map (element type l_kind), or NULL: one region "total" over all voxels */
    int64_t N;
    int32_t nkeys;             /* number of labelled regions (<= QMRI_MAX_REGIONS - 1), 0 if labels == NULL */
    int32_t l_kind;            /* label element type: 0 int32, 1 uint8, 2 int16 */
    const int32_t *label_keys; /* [nkeys] (host memory in both entry points) */
    int32_t use_bounds;        /* bounds = (lo, hi) */
    double lo, hi;
    int32_t closed;            /* interval ends that belong to it: 0 neither, 1 left, 2 right, 3 both */
    double *out;               /* [nkeys + 1][4] (host memory): count, mean, std, median; NaN statistics for count 0 */
    int32_t device;
} qmri_region_stats_args;
#define QMRI_MAX_REGIONS 16
int qmri_region_stats_host(const qmri_region_stats_args *args); /* values / labels in host memory, synchronous */
/* values / labels already on the device (e.g. the tc map a fit has just written); label_keys and out stay host
 * pointers; the call enqueues on `hip_stream` and returns after the 4 x (nkeys + 1) results have arrived */
int qmri_region_stats_device(const qmri_region_stats_args *args, void *hip_stream);

/* Mean kernel time in ms of the last qmri_monoexp_fit_device call on this thread that was issued
 * with timing enabled (qmri_set_timing(1)); measured with hipEvents on the launch stream. */
void qmri_set_timing(int enable);
float qmri_last_kernel_ms(void);

/* Page-locked host memory for result arrays (dosma_amd/_hostpool.py recycles it): resident pages -- no first-touch
 * zeroing when a freshly allocated result is written (65 ms per GB on the reference host) -- and a true DMA target.
 * The reference has no counterpart (numpy allocates its results, dosma/core/fitting.py:205-215). */
int qmri_host_alloc(uint64_t bytes, void **out);
int qmri_host_free(void *p);

int qmri_version(void);
int qmri_device_count(void);
/* free / total bytes of device memory (hipMemGetInfo): callers size their per-model activation buffers from it */
int qmri_device_mem_info(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes);
/* Self-test of the two elementary functions the fit kernels carry themselves (dosma_amd/csrc/fp64_fast.h): for n host values x,
 * exp_sk(x) next to the device library's exp(x) -- bit-identical by construction -- and log_sk(x) (the logarithm of the
 * log-linear starting point, fitting.py:701-718) next to the device library's log(x).  Host arrays of n doubles each. */
int qmri_selftest_fp64(int32_t device, const double *x, int64_t n, double *exp_sk_out, double *exp_lib_out,
                       double *log_sk_out, double *log_lib_out);
const char *qmri_last_error(void);
/* name of the fit kernel variant that `args` would dispatch to (for profiles / tests) */
const char *qmri_monoexp_kernel_name(const qmri_monoexp_args *args);

#ifdef __cplusplus
}
#endif
#endif /* QMRI_H */
