// qmri_internal.h -- shared between the HIP kernels and the C-ABI layer (not installed).
#ifndef QMRI_INTERNAL_H
#define QMRI_INTERNAL_H

#include <hip/hip_runtime.h>

#include "qmri.h"


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
    // masked launches: tiles with at least one selected voxel, compacted by monoexp_mask_prepass (the other tiles
    // are filled there); tile_list == nullptr -> every tile of the volume
    const unsigned int *tile_list;
    const unsigned int *tile_list_count;
    int *nonfinite;
    double xmean, sxx;           // of x: closed-form degree-1 least squares (log-linear init)
    // equally spaced sample times x_i = x_0 + i * x_step (x_0 >= 0, x_step > 0): exp(b x_i) = exp(b x_0) * exp(b x_step)^i,
    // two exponentials + E multiplications per model evaluation instead of E exponentials
    int uniform_x;
    int x0_pow;                  // uniform_x and x_0 = k * x_step, 0 <= k <= 4: exp(b x_0) = exp(b x_step)^k, one exponential; else -1
    int lmpar_closed_form;       // evaluate lmpar's Newton iteration in closed form (monoexp_lm.hip lmpar2)
    double x_step;
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
// Kernel-argument block of the general polynomial least-squares kernel (linfit.hip: polyls_kernel).
struct PolylsKArgs {
    const void *y;
    long long ld;
    long long N;
    int E, P;            // samples, parameters (deg + 1 <= QMRI_POLY_MAX_PARAMS)
    int y_dtype;
    int skip_rules;
    int use_y_bounds;
    double y_lo, y_hi;
    double r2_eps;
    const double *ops;   // DEVICE: [S (P x E) | D (E x P) | w (E)]
    double *popt;        // [N][P]
    double *r2;          // [N]
    double *resid;       // nullable [N]
};
hipError_t polyls_launch(const PolylsKArgs &k, int num_cu, hipStream_t stream);
// Kernel-argument block of the general lmdif kernel (lm_generic.hip).
struct LmKArgs {
    const void *y;
    long long ld;
    long long N;
    int E;
    int y_dtype;
    int maxfev;
    int use_y_bounds;
    double y_lo, y_hi;
    double p0[QMRI_LM_MAX_PARAMS];
    const double *p0v[QMRI_LM_MAX_PARAMS];
    double ftol, xtol, gtol, factor, epsfcn, r2_eps;
    double *popt;
    double *r2;
    signed char *info;
    short *nfev;
    int *nonfinite;
    double x[QMRI_LM_MAX_ECHOES];
};
int lm_generic_nparams(int model);
// counter: one zeroed 8-byte device word per launch (E <= 12: the pulling kernel), or nullptr (general-E kernel)
hipError_t lm_generic_launch(const LmKArgs &k, int model, int num_cu, unsigned long long *counter, hipStream_t stream);
// masked launches: classify tiles, fill the empty ones, list the others (list: [tiles] u32, count: 1 u32, zeroed)
hipError_t monoexp_mask_prepass(const FitKArgs &k, unsigned int *list, unsigned int *count, int num_cu, hipStream_t stream);
hipError_t linfit_launch(const LinfitKArgs &k, int num_cu, hipStream_t stream);

// First encoder block of the parity-mode U-Net in one kernel (unet_enc0.hip): image -> skip tensor + pooled map.
struct Enc0Args {
    const float *x;        // [B][H][W] input slices (whitened where the model asks for it)
    int B, H, W;           // H % 8 == 0, W % 32 == 0
    const void *c1_img;    // first convolution as an MFMA A operand: [64 lanes][hi 8 | lo 8] fp16 of 2^k * (taps 0..8, bias, 0...)
    float c1_winv;         // 2^-k
    const void *w2;        // second convolution: conv_s3_kernel's weight image for 32 -> 32 channels (9 x 4096 B)
    float winv2;
    const float *bias2, *scale2, *shift2;  // [32]: bias, BatchNorm scale / shift
    void *y;               // skip tensor (split layout), pixel stride ldy channels, channel offset yoff
    long long ldy;
    int yoff;
    void *pool_y;          // pooled map (split layout), pixel stride pool_ld channels
    int pool_ld;
    int *sat;              // nullable: set to 1 when a value stored in the split layout exceeds the fp16 range (see ConvS3Args)
};
size_t enc0_lds_bytes();
bool enc0_supported(const Enc0Args &k);
hipError_t enc0_launch(const Enc0Args &k, int num_cu, hipStream_t stream);

// Last convolution of the parity-mode U-Net + the 1x1 classifier in one kernel (unet_enc0.hip: out0_kernel).
struct Out0Args {
    const void *x;         // 32-channel input (split layout), pixel stride ldx channels, channel offset xoff
    long long ldx;
    int xoff;
    int B, H, W;           // H % 8 == 0, W % 32 == 0
    const void *w;         // conv_s3_kernel's weight image for 32 -> 32 channels (9 x 4096 B)
    float winv;
    const float *bias, *scale, *shift;  // [32]
    const float *head_w;   // [32][nc]
    const float *head_b;   // [nc]
    int nc;                // 1..4 classes
    float *logits;         // nullable [B*H*W][nc]
    unsigned char *mask;   // nullable [B*H*W][nc]: logit > 0
};
size_t out0_lds_bytes();
bool out0_supported(const Out0Args &k);
hipError_t out0_launch(const Out0Args &k, int num_cu, hipStream_t stream);

// Conv2D(64 -> 32, 3x3) + ReLU with LDS-resident weights (unet_enc0.hip: mid0_kernel).
struct Mid0Args {
    const void *x;         // 64-channel input (split layout), pixel stride ldx channels, channel offset xoff
    long long ldx;
    int xoff;
    int B, H, W;           // H % 8 == 0, W % 32 == 0
    const void *w;         // conv_s3_kernel's weight image for 64 -> 32 channels (2 x 9 x 4096 B)
    float winv;
    const float *bias;     // [32]
    void *y;               // 32-channel output (split layout), pixel stride ldy channels, channel offset yoff
    long long ldy;
    int yoff;
    int *sat;              // nullable: saturation flag (see ConvS3Args)
};
size_t mid0_lds_bytes();
bool mid0_supported(const Mid0Args &k);
hipError_t mid0_launch(const Mid0Args &k, int num_cu, hipStream_t stream);

// Kernel-argument block of the implicit-GEMM convolution (unet_kernels.hip).
struct ConvKArgs {
    const void *x;       // NHWC input (fp32, or bf16 in plain-bf16 mode), pixel stride ldx (elements), channel offset xoff
    long long ldx;
    int xoff;
    int B, H, W;         // input grid: GEMM rows = B*H*W
    int Cin, Cout;
    int ntaps;
    int deconv;               // 1: fused 4-phase Conv2DTranspose (ntaps = 9 in phase order, sy = sx = 2)
    unsigned long long taps;  // 4 bits per tap: (dy+1) | (dx+1) << 2, dy, dx in {-1, 0, 1}
    int tiles_y, tiles_x;     // filled by conv_igemm_launch
    int halo_bufs;            // filled by conv_igemm_launch: 2 (double-buffered halo) or 1 (single K chunk)
    const __bf16 *w_hi;  // [Cout][chunk][tap][32] bf16: K index = (chunk*ntaps + tap)*32 + c (K-major per cout)
    const __bf16 *w_lo;  // parity mode: w_hi / w_lo hold the fp16 hi / lo parts of 2^wshift * weights (same 2-byte slots)
    float winv;          // parity mode: 2^-wshift (the accumulators hold 2^wshift * convolution)
    const float *bias;   // [Cout] or nullptr
    const float *scale;  // [Cout] or nullptr   y = scale * relu(acc + bias) + shift
    const float *shift;
    int relu;
    void *y;             // NHWC output (same element type as x), pixel stride ldy, channel offset yoff
    long long ldy;
    int yoff;
    int Ho, Wo;          // output image size
    int sy, sx, py, px;  // output pixel = (y*sy + py, x*sx + px)
    // fused producer: the first layer Conv2D(32, 3x3)+ReLU on the 1-channel image [B][H][W] (nullable)
    const float *c1_x;
    const float *c1_w;   // [9][32] tap-major
    const float *c1_b;   // [32]
    // fused consumers of the finished tile (all nullable)
    void *pool_y;        // MaxPooling2D(2x2) of the output, compact NHWC with pool_ld channels per pixel
    int pool_ld;
    int head_nc;         // 1x1 head: classes
    const float *head_w; // [Cout][head_nc]
    const float *head_b; // [head_nc]
    float *logits;       // [pixels][head_nc]
    unsigned char *mask; // [pixels][head_nc]  (logit > 0)
    int *sat;            // parity mode, nullable: saturation flag (see ConvS3Args)
};
hipError_t conv_igemm_launch(const ConvKArgs &k, int split3, hipStream_t stream);

// Parity-mode ("fp16x3") 3x3 convolution on SPLIT activations (unet_s3.hip).  Activation layout (x, y, pool_y): per pixel
// `ld` channel slots of 4 bytes; the 32-channel chunk starting at channel c (a multiple of 32) is the 128 bytes at
// ((pixel * ld + off + c) * 4): 32 fp16 hi parts, then 32 fp16 lo parts (value = hi + lo).
struct ConvS3Args {
    const void *x;
    long long ldx;
    int xoff;
    int B, H, W;
    int Cin, Cout;
    int deconv;          // 1: Conv2DTranspose(3x3, strides 2, SAME): (B, H, W) is the INPUT grid, the output grid is 2H x 2W,
                         //    9 taps packed in phase order (pack_deconv_fused)
    const void *w;       // packed by pack_s3_weights: [Cout / BN][chunk * 9 + tap][plane hi | lo][BN][4 x 16 B swizzled], fp16, x 2^wshift
    float winv;          // 2^-wshift: the accumulators hold 2^wshift * convolution
    const float *bias;   // [Cout] or nullptr
    const float *scale;  // [Cout] or nullptr   y = scale * relu(acc * winv + bias) + shift
    const float *shift;
    int relu;
    void *y;             // nullable (head-only layer)
    long long ldy;
    int yoff;
    void *pool_y;        // fused MaxPooling2D(2x2), compact split layout with pool_ld channel slots per pixel (nullable)
    int pool_ld;
    int head_nc;         // fused 1x1 head (Cout = 32 only)
    const float *head_w; // [32][head_nc]
    const float *head_b;
    float *logits;
    unsigned char *mask;
    // filled by conv_s3_launch
    int chunks, steps, nb, ntiles, nwork, tiles_x, tiles_y, P, nj;
    int dbg;             // QMRI_S3_DBG timing experiments (0 in production)
    int one;             // 1: plain-bf16 mode on this kernel -- x / y / pool_y are bf16 NHWC tensors, w is the bf16 image of
                         // pack for 64-channel chunks (plane p = channels 32 p .. 32 p + 31 of the chunk); Cin, ldx, xoff are
                         // given in 4-BYTE units (channels / 2), Cout / ldy / yoff / pool_ld in channels
    // Saturation flag (nullable).  The split layout stores a value as fp16 hi + lo parts; v_cvt_pkrtz clamps at 65504, so a
    // feature-map value beyond the fp16 range would be stored wrong WITHOUT any error.  Every kernel that writes the layout
    // tracks max |v| of what it stores and sets *sat = 1 when it exceeds 65504; the engine then repeats the forward with the
    // whole network scaled down by a power of two (exact: unet_engine.hip, `act_shift`) instead of returning clamped results.
    int *sat;
    // conv_c4_kernel (unet_c4.hip): the same layer on one wave per SIMD with 128 x 128 register tiles, for Cout % 128 == 0.
    // w_c4 = the weights packed per (channel block, chunk, 16-channel half, tap) as the 8 KB LDS image of a ring slot
    // ([plane][128 rows][2 x 16 B], piece g of row n at position g ^ ((n >> 3) & 1)); nullable.  c4_mode: 0 = the launcher's
    // cost model picks the kernel per layer, 1 = conv_c4_kernel (error if it does not take the layer), -1 = conv_s3_kernel.
    const void *w_c4;
    int c4_mode;
    int c4_split;        // filled by conv_c4_launch: split the items of the last, partial round by channels (QMRI_C4_SPLIT=0 turns it off)
    int tile_group;      // filled by the launchers: channel blocks per tile-major GROUP (the blocks of a tile side by side on one XCD, the
                         // tile's halo fetched once per group).  conv_c4_launch: 0 = channel-major, 1 = groups of TWO blocks (the only
                         // group size conv_c4_kernel knows); conv_d4_launch: 0 / 1 = channel-major, G = 2..4 blocks per group
};
bool conv_s3_supported(const ConvS3Args &k);
// which tiling conv_c4_kernel / deconv_d4_kernel give a level of width W: the flattened zero-framed stack up to W = 48 (unless 32 wide),
// image tiles of 32 columns otherwise -- with a ragged last column tile when W % 32 != 0 (conv_s3_kernel has no such tile: there the
// general kernel is the fallback)
inline bool conv_tiles_flat(int W) { return W % 32 != 0 && W + 2 <= 50; }
bool conv_c4_supported(const ConvS3Args &k);
// deconv_d4_kernel (unet_d4.hip): the transposed convolution on one wave per SIMD (4 row-tiles x 4 phases x 32 channels).  w_c4 then
// holds ITS weight image: per (32-channel block, k-step = chunk * 2 + half) nine taps of [plane][32 rows][2 x 16 B] in shift-group
// order (ConvLayer::upload_parity).  c4_mode as for conv_c4_kernel.
bool conv_d4_supported(const ConvS3Args &k);
bool conv_s3_takes_d4(const ConvS3Args &k);  // will conv_s3_launch run this transposed convolution on deconv_d4_kernel
hipError_t conv_d4_launch(const ConvS3Args &k, int num_cu, hipStream_t stream);
int conv_c4_block_channels(int Cout);  // 128, or 64 for Cout = 64 (mod 128): the packing of w_c4 depends on it
bool conv_s3_takes_c4(const ConvS3Args &k, int num_cu);  // will conv_s3_launch run this layer on conv_c4_kernel
hipError_t conv_c4_launch(const ConvS3Args &k, int num_cu, hipStream_t stream);
int conv_s3_block_channels(int Cout, int deconv);  // channel-block size the kernel uses for a layer: the weight packing depends on it
hipError_t conv_s3_launch(const ConvS3Args &k, int num_cu, hipStream_t stream);
// streaming kernels on the split layout (unet_s3.hip)
hipError_t c1_split_launch(const float *x, int B, int H, int W, const float *w, const float *bias, int Cout, void *y,
                           long long ldy, int yoff, int *sat, hipStream_t stream);
hipError_t maxpool2_split_launch(const void *x, long long ldx, int xoff, int B, int H, int W, int C, void *y, hipStream_t stream);
hipError_t maxpoolk_split_launch(const void *x, long long ldx, int xoff, int B, int H, int W, int C, int K, void *y,
                                 hipStream_t stream);
hipError_t maxpoolk_launch(const void *x, long long ldx, int xoff, int B, int H, int W, int C, int K, void *y, hipStream_t stream);  // bf16 layout
hipError_t head_split_launch(const void *x, long long npix, int Cin, const float *w, const float *bias, int NC, float *logits,
                             unsigned char *mask, hipStream_t stream);
hipError_t split_cast_launch(const void *x, long long npix, int C, void *y, int to_split, hipStream_t stream);
// register-weights kernel for the Cout = 32 layers in plain-bf16 mode (unet_rw.hip)
bool conv_rw_supported(const ConvKArgs &k);
hipError_t conv_rw_launch(const ConvKArgs &k, hipStream_t stream);
hipError_t conv3x3_c1_launch(const float *x, int B, int H, int W, const float *w, const float *bias,
                             int Cout, void *y, long long ldy, int yoff, int act_bf16, hipStream_t stream);
hipError_t maxpool2_launch(const void *x, long long ldx, int xoff, int B, int H, int W, int C, void *y,
                           int act_bf16, hipStream_t stream);
hipError_t head_launch(const void *x, long long npix, int Cin, const float *w, const float *bias, int NC,
                       float *logits, unsigned char *mask, int act_bf16, hipStream_t stream);
hipError_t cast_launch(const void *x, long long n, void *y, int to_bf16, hipStream_t stream);
hipError_t transpose_ps_launch(const float *x, long long P, int S, float *y, hipStream_t stream);
hipError_t mask_planes_launch(const unsigned char *mask_sp4, long long P, int S, int C, unsigned char *out,
                              hipStream_t stream);
hipError_t absmax_launch(const float *x, long long n, unsigned int *out, hipStream_t stream);
hipError_t scale_copy_launch(const float *x, long long n, float f, float *y, hipStream_t stream);
int whiten_stats_doubles();  // size of whiten_launch's `stats` buffer: [sum, sum of squares, mean, -] + the per-block partial sums
hipError_t whiten_launch(const float *x, long long n, double eps, double *stats, float *y,
                         hipStream_t stream);

// Kernel-argument block of the analytic DESS T2 kernel (dess.hip).
struct DessKArgs {
    const void *echo1, *echo2;
    long long N;
    double c0, k, c1;          // t2 = c0 / (log(|e2/e1| / k) + c1)
    int use_bounds;
    int use_nan_to_num;
    double lo, hi, nan_value;
    int decimals;              // QMRI_NO_ROUND = none
    int suppress_fat, suppress_fluid, out_f64;
    int vec_ok;                // bases are 32-byte aligned: 4 voxels per lane
    double p10, beta;
    double ip10;               // RN(1 / p10): x / p10 = Markstein's correctly rounded q + fma(-p10, q, x) * ip10 (dess.hip)
    const double *maxima;      // device [2]: max(echo1), max(echo1 - beta*echo2)
    void *t2;
};
hipError_t dess_t2_launch(const DessKArgs &k, int dtype, int num_cu, double *scratch, hipStream_t stream);
hipError_t rss_launch(const void *e1, const void *e2, int dtype, long long n, int rms, double *out, int num_cu,
                      hipStream_t stream);

size_t region_stats_state_bytes();
hipError_t region_stats_launch(const void *values, int f64, const void *labels, int l_kind, long long N, int nkeys,
                               const int *keys,
                               int use_bounds, double lo, double hi, int closed, void *state, double *out_dev,
                               int num_cu, hipStream_t stream);

void set_last_error(const char *msg);  // thread-local message behind qmri_last_error()

int monoexp_tile_voxels();
const char *monoexp_variant_name(int E, int y_dtype);
int monoexp_blocks_per_cu(const FitKArgs &k);
int monoexp_waves_per_block(const FitKArgs &k);
hipError_t monoexp_launch(const FitKArgs &k, int grid, hipStream_t stream);

}  // namespace qmri
#endif
