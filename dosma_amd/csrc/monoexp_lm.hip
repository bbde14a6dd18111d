// monoexp_lm.hip -- per-voxel mono-exponential Levenberg-Marquardt fit for gfx950 (MI355X).
//
// Replaces the per-voxel Python loop of the reference
//     /root/reference/dosma/core/fitting.py:855-868   (curve_fit: for i in range(N): fitter(y_T[i]))
//     /root/reference/dosma/core/fitting.py:1026-1073 (_curve_fit: skip rule, scipy.optimize.curve_fit,
//                                                     r2, RuntimeError -> NaN)
// whose arithmetic is MINPACK lmdif as configured by scipy.optimize.leastsq (mode 1, factor 100,
// xtol 1.49012e-8, gtol 0) with DOSMA's ftol = 1e-5 and maxfev = 100.
//
// Design (CDNA4, no MFMA -- this is not a dense contraction):
//   * one voxel per lane, the whole LM state (x, R, Q^T f, diag, delta, par, ...) in fp64 VGPRs;
//     fp64 because the parity target is an EARLY-STOPPED solver: the stop tests compare
//     1 - (|f+|/|f|)^2 against 1e-5, which fp32 cannot resolve;
//   * MINPACK's control flow is kept exactly (lmpar trust region, ratio tests, info codes, nfev
//     accounting with +n per Jacobian); the forward-difference Jacobian is reproduced from the E
//     exponentials of the accepted trial point (no extra exps), so one LM iteration costs E exps --
//     or one / two exps + E multiplications when the sample times are equally spaced (FitKArgs::uniform_x);
//   * lmpar's Newton iteration on the LM parameter is evaluated in closed form for n = 2 (lmpar2): same
//     iterates as qrsolv's Givens sweeps to rounding, a quarter of the instructions;
//   * waves are independent persistent workers: a wave claims a tile of SUB consecutive voxels
//     with one atomic, stages it echo-major into its private LDS slice with coalesced
//     16-byte-per-lane loads, and its lanes *pull* voxels from that tile whenever they finish one
//     (wave ballot + mbcnt rank) -- so masked-out / all-zero voxels and early-converging voxels do
//     not leave lanes idle behind the slowest voxel of a fixed 64-voxel group;
//   * newly pulled voxels enter the same instruction stream as running ones (state INIT shares the
//     model evaluation and the QR with state ITER), so there is no separate divergent init path;
//   * the epilogue of MonoExponentialFit (1/|b|, bounds, r2 threshold, nan_to_num, rounding) is fused
//     into the lane's final store, which is batched with the refill (lanes park in ST_DONE meanwhile).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cmath>
#include <type_traits>

#include "fp64_fast.h"
#include "qmri_internal.h"

namespace qmri {

#ifndef QMRI_KSUB
#define QMRI_KSUB 128
#endif
constexpr int kSub = QMRI_KSUB;  // voxels per tile (per wave).  128 (2 per lane, 8-byte staging loads for f32) since round 3: a
                                 // wave's LDS slice is 12.4 KB at 8 samples, so that 12 waves fit the CU's 160 KB (256: 19.7 KB)
constexpr int kVpl = kSub / 64;  // voxels per lane of a tile
static_assert(kSub == 256 || kSub == 128, "tile = 4 or 2 voxels per lane");
#ifndef QMRI_REFILL
#define QMRI_REFILL 16
#endif
#ifndef QMRI_TILE_OUT
#define QMRI_TILE_OUT 0  // 1: variants up to 8 samples retire their results tile by tile (full-line stores) instead of through the result ring -- measured: writes 1.0 x, but 21.4 instead of 17.4 ms (see flush_tile)
#endif
#ifndef QMRI_XS_MIN
#define QMRI_XS_MIN 0   // the LDS table of the sample times is used for QMRI_XS_MIN < EMAX <= 12 (8: rounds-1/2 behaviour for <= 8 samples)
#endif
#ifndef QMRI_SMALL_E_BLOCKS
#define QMRI_SMALL_E_BLOCKS 3  // blocks of 4 waves per CU the EMAX <= 8 variants are register-bounded for: THREE waves per SIMD
                               // (168 VGPRs + 48 spilled at 8 samples, 4 at 4 samples) since round 3 -- 20.7 -> 18.2 ms.  Round 2
                               // measured the opposite (168 + 102 spilled: 1.17e9 instead of 1.89e9 voxel-fits/s): the cold kernel
                               // arguments were still in the scalar registers then, and the LDS slices allowed two blocks only
#endif
#ifndef QMRI_Y_F64_MAX
#define QMRI_Y_F64_MAX 8
#endif
#ifndef QMRI_MIN_WAVES
#define QMRI_MIN_WAVES 1
#endif
constexpr int kRefillIdle = QMRI_REFILL;  // refill the wave when at least this many lanes are idle

// Separately rounded product / difference (no FMA contraction): numpy evaluates the model as
// a * exp(b * x) - y with one rounding per operation, and lmdif's forward-difference Jacobian is
// sensitive to exactly that rounding (it can cancel to 0), so the model evaluation keeps it.
__device__ __forceinline__ double mul_rn(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double sub_rn(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}

// QMRI_COLD_PATH() (fp64_fast.h) marks a rarely taken block: an asm with side effects cannot be speculated, so the
// block stays behind a real branch.  (Left alone, the compiler if-converts the guarded slow paths below -- IEEE
// divisions and square roots of 13-25 instructions each -- and executes them on EVERY call, selecting afterwards.)

// max / min of two values that are never signalling NaNs: the bare instruction (fmax / fmin first canonicalise every operand
// that is not the result of an arithmetic instruction -- a v_max_f64 x, x each -- to quiet signalling NaNs; quiet NaNs are
// handled by the instruction itself exactly as fmax / fmin do: the other operand is returned)
__device__ __forceinline__ double max_q(double a, double b) {
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double min_q(double a, double b) {
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// The guards of the fast reciprocal / square-root forms below: ONE v_cmp_class instead of two range compares (the forms
// are safe for every normal argument: the refinement terms stay normal; 0, denormals, inf, NaN and negative arguments take
// the IEEE path as before).
__device__ __forceinline__ bool is_pos_normal(double x) { return __builtin_amdgcn_class(x, 0x100); }

__device__ __forceinline__ void sqrt_rsqrt(double x, double &s, double &rs);
__device__ __forceinline__ double norm2(double a, double b) {
    const double s = a * a + b * b;
    if (__builtin_expect(is_pos_normal(s), 1)) {
        double n, rn;
        sqrt_rsqrt(s, n, rn);
        return n;
    }
    QMRI_COLD_PATH();
    const double m = fmax(fabs(a), fabs(b));
    if (!(m > 0.0) || isinf(m)) return (isnan(a) || isnan(b)) ? (a + b) : m;
    const double ra = a / m, rb = b / m;
    return m * sqrt(ra * ra + rb * rb);
}

// ---- fp64 reciprocal / square root without the IEEE division expansion ------------------------------
// An fp64 `a / b` compiles to ~11 VALU instructions (v_div_scale x2, v_rcp, 5 fma, v_div_fmas,
// v_div_fixup) and sqrt() to ~12; one LM iteration of the straightforward restatement spends ~50 of
// them.  The solver state is bounded away from the exponent range limits, so the scale / fixup steps
// are only needed on a guarded slow path.  Results are within 1-2 ulp of IEEE (parity is 1e-4, not bitwise).
__device__ __forceinline__ double frcp(double b) {
    const double r0 = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r0, 1.0);
    double r = fma(r0, e, r0);
    e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    // 0, inf, NaN and the (sub)normal extremes, where the refinement would produce 0 * inf: the hardware result
    // as it is (exact for 0 / inf / NaN).  An IEEE division here gets if-converted by the compiler -- 13 more
    // instructions on EVERY call -- for a range the solver only visits on voxels that fail anyway.
    return __builtin_amdgcn_class(b, 0x108) ? r : r0;  // +-normal
}

// s = sqrt(x), rs = 1/sqrt(x) by Goldschmidt iteration from v_rsq_f64 (9 instructions for both)
__device__ __forceinline__ void sqrt_rsqrt(double x, double &s, double &rs) {
    if (__builtin_expect(is_pos_normal(x), 1)) {
        const double r = __builtin_amdgcn_rsq(x);
        double g = x * r;
        double h = 0.5 * r;
        double e = fma(-h, g, 0.5);
        g = fma(g, e, g);
        h = fma(h, e, h);
        e = fma(-h, g, 0.5);
        g = fma(g, e, g);
        h = fma(h, e, h);
        s = g;
        rs = h + h;
    } else {
        QMRI_COLD_PATH();
        s = sqrt(x);
        rs = 1.0 / s;
    }
}

// ||(a, b)|| and its reciprocal; overflow-safe like MINPACK's enorm on the slow path
__device__ __forceinline__ void norm2r(double a, double b, double &n, double &rn) {
    const double q = a * a + b * b;
    if (__builtin_expect(is_pos_normal(q), 1)) {
        sqrt_rsqrt(q, n, rn);
    } else {
        QMRI_COLD_PATH();
        n = norm2(a, b);
        rn = 1.0 / n;
    }
}

// Givens rotation of MINPACK qrsolv that annihilates sdk against rkk: (c, s) = (rkk, sdk) / h,
// h = hypot(rkk, sdk).  MINPACK's two-branch tangent/cotangent form differs only by a common sign of
// (c, s), which cancels in the solution; returns h (the new diagonal) and 1/h.
__device__ __forceinline__ void givens(double rkk, double sdk, double &c, double &s, double &h,
                                       double &rh) {
    norm2r(rkk, sdk, h, rh);
    c = rkk * rh;
    s = sdk * rh;
}

// qrsolv for n = 2.  R = [r11 r12; 0 r22] (pivoted columns l0, l1), dg = sqrt(par)*diag by ORIGINAL
// parameter index, qtb = Q^T f.  Returns x by original index and the triangular factor S
// (reciprocal diagonal rsd0, rsd1 and off-diagonal s10) that lmpar's Newton correction needs.
__device__ __forceinline__ void qrsolv2(double r11, double r12, double r22, double ir11, double ir22,
                                        int l0, double dg0, double dg1, double qtb0, double qtb1,
                                        double &x0, double &x1, double &rsd0, double &rsd1,
                                        double &s10) {
    const double dl0 = l0 ? dg1 : dg0;
    const double dl1 = l0 ? dg0 : dg1;
    double rr11 = r22, wa0 = qtb0, wa1 = qtb1;
    double sd0 = r11, sd1;
    rsd0 = ir11;
    rsd1 = ir22;
    s10 = r12;
    if (dl0 != 0.0) {
        double c, s, h, rh;
        givens(r11, dl0, c, s, h, rh);
        sd0 = h;
        rsd0 = rh;
        const double qtbpj = -s * wa0;
        wa0 = c * wa0;
        const double sdi = -s * s10;
        s10 = c * s10;
        if (sdi != 0.0) {
            givens(rr11, sdi, c, s, h, rh);
            rr11 = h;
            rsd1 = rh;
            wa1 = c * wa1 + s * qtbpj;
        }
    }
    if (dl1 != 0.0) {
        double c, s, h, rh;
        givens(rr11, dl1, c, s, h, rh);
        rr11 = h;
        rsd1 = rh;
        wa1 = c * wa1;
    }
    sd1 = rr11;
    if (sd0 == 0.0) {
        wa0 = 0.0;
        wa1 = 0.0;
    } else if (sd1 == 0.0) {
        wa1 = 0.0;
        wa0 = wa0 * rsd0;
    } else {
        wa1 = wa1 * rsd1;
        wa0 = (wa0 - s10 * wa1) * rsd0;
    }
    x0 = l0 ? wa1 : wa0;
    x1 = l0 ? wa0 : wa1;
}

#ifdef QMRI_STATS
// debug build only (scripts/fit_stats.py): lane-utilisation counters per phase
// [0] loop rounds (x64 = lane slots) [1] busy lanes [2] lanes entering lmpar [3] lane lmpar iterations
// [4] wave lmpar iterations (max over lanes, summed) [5] lanes in Jacobian+QR [6] lanes finishing
// [7] rounds with a refill
__device__ unsigned long long g_fit_stats[16];  // [8..14]: s_memtime cycles in refill+epilogue, lmpar set-up, lmpar loop, model eval, accept logic, FD Jacobian, QR
#define QMRI_TIC() (st_t1 = __builtin_readcyclecounter())
#define QMRI_TOC(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_t1; st_t1 = t_; } while (0)
#define QMRI_STAT_ADD(i, v) st_acc[i] += (unsigned long long)(v)
#else
#define QMRI_STAT_ADD(i, v)
#define QMRI_TIC()
#define QMRI_TOC(i)
#endif

// MINPACK lmpar for n = 2: step p (by original index) with ||diag*p|| ~ delta, and the LM parameter.
// ir11, ir22 = 1/r11, 1/r22 and idg0, idg1 = 1/diag are maintained by the caller (they change only at
// a QR).  ((fp/delta)/temp)/temp is evaluated as fp / (delta * temp^2): no square root of temp^2.
__device__ __forceinline__ void lmpar2(double r11, double r12, double r22, double ir11, double ir22,
                                       int l0, double dg0, double dg1, double idg0, double idg1,
                                       double qtb0, double qtb1, double delta, double &par, double &x0,
                                       double &x1, bool closed_form
#ifdef QMRI_STATS
                                       , int &iters_out, unsigned long long *st_acc, unsigned long long &st_t1
#endif
                                       ) {
#ifdef QMRI_STATS
    iters_out = 0;
#endif
    const double dwarf = DBL_MIN;
    const double dl0 = l0 ? dg1 : dg0;  // diag(ipvt(0))
    const double dl1 = l0 ? dg0 : dg1;  // diag(ipvt(1))
    const double idl0 = l0 ? idg1 : idg0;
    const double idl1 = l0 ? idg0 : idg1;
    // Gauss-Newton direction
    double w0 = qtb0, w1 = qtb1;
    int nsing = 2;
    if (r11 == 0.0) {
        nsing = 0;
        w0 = 0.0;
        w1 = 0.0;
    } else if (r22 == 0.0) {
        nsing = 1;
        w1 = 0.0;
        w0 = w0 * ir11;
    } else {
        w1 = w1 * ir22;
        w0 = (w0 - r12 * w1) * ir11;
    }
    x0 = l0 ? w1 : w0;
    x1 = l0 ? w0 : w1;
    double wa20 = dg0 * x0, wa21 = dg1 * x1;
    double dxnorm, idx;
    norm2r(wa20, wa21, dxnorm, idx);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        par = 0.0;
        return;
    }
    double parl = 0.0;
    if (nsing >= 2) {
        double t0 = dl0 * ((l0 ? wa21 : wa20) * idx);
        double t1 = dl1 * ((l0 ? wa20 : wa21) * idx);
        t0 = t0 * ir11;
        t1 = (t1 - r12 * t0) * ir22;
        parl = fp * frcp(delta * (t0 * t0 + t1 * t1));
    }
    const double g0 = (r11 * qtb0) * idl0;
    const double g1 = (r12 * qtb0 + r22 * qtb1) * idl1;
    const double gnorm = norm2(g0, g1);
    double paru = gnorm * frcp(delta);
    if (paru == 0.0) paru = dwarf / fmin(delta, 0.1);
    par = max_q(par, parl);
    par = min_q(par, paru);
    if (par == 0.0) par = gnorm * idx;
    QMRI_TOC(9);
    // Closed-form evaluation of the same Newton iteration.  In the scaled variable z = D P^T x the damped normal
    // equations are (B + par I) z = g with B = D^-1 R^T R D^-1 = [s0^2, s0 t; s0 t, t^2 + v^2], s0 = r11/d0, t = r12/d1,
    // v = r22/d1 (all <= 1 in magnitude: diag >= column norm) and g = D^-1 R^T qtb = (s0 qtb0, t qtb0 + v qtb1).  With
    // the cancelling terms removed analytically,
    //     det = s0^2 v^2 + par (s0^2 + t^2 + v^2 + par)                (a sum of non-negative terms)
    //     z0  = s0 ((v^2 + par) qtb0 - t v qtb1) / det,   z1 = (s0^2 v qtb1 + par g1) / det
    // -- the one remaining difference is the one back-substitution has as well -- and lmpar's correction
    // ((fp/delta)/temp)/temp, temp^2 = z^T (B + par I)^-1 z / |z|^2, through the Cholesky form
    //     a1 z^T (B + par I)^-1 z = z0^2 + (a1 z1 - s0 t z0)^2 / det,   a1 = s0^2 + par    (again no cancellation).
    // ~45 instructions per iteration against ~115 for qrsolv's three Givens rotations (each a norm + reciprocal)
    // and the two triangular solves; the iterates agree with MINPACK's to rounding (1e-13 relative at worst).
    // Lanes whose magnitudes could leave the range where the squares are safe take the qrsolv loop below.
    const double amax = fmax(fabs(qtb0), fabs(qtb1));
    if (closed_form && nsing >= 2 && amax < 1e100 && amax > 1e-100 && delta < 1e100 && delta > 1e-100) {
        const double s0 = r11 * idl0, t = r12 * idl1, v = r22 * idl1;
        const double s02 = s0 * s0, v2 = v * v;
        const double S = s02 + fma(t, t, v2), d0 = s02 * v2;
        const double c1 = (t * v) * qtb1, c2 = (s02 * v) * qtb1, b12 = s0 * t;
        const double gg1 = fma(t, qtb0, v * qtb1);
        double z0 = 0.0, z1 = 0.0;
        for (int iter = 1;; ++iter) {
#ifdef QMRI_STATS
            iters_out = iter;
#endif
            if (par == 0.0) par = fmax(dwarf, 0.001 * paru);
            const double det = fma(par, S + par, d0);
            const double rdet = frcp(det);
            z0 = (s0 * fma(v2 + par, qtb0, -c1)) * rdet;
            z1 = fma(par, gg1, c2) * rdet;
            norm2r(z0, z1, dxnorm, idx);
            const double fp_old = fp;
            fp = dxnorm - delta;
            if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= fp_old && fp_old < 0.0) || iter == 10) break;
            const double a1 = s02 + par;
            const double n1 = fma(a1, z1, -(b12 * z0));
            const double qq = fma(n1 * rdet, n1, z0 * z0);
            const double parc = (fp * a1) * (dxnorm * dxnorm) * frcp(delta * qq);
            if (fp > 0.0) parl = max_q(parl, par);
            if (fp < 0.0) paru = min_q(paru, par);
            par = max_q(parl, par + parc);
        }
        const double xl0 = z0 * idl0, xl1 = z1 * idl1;
        x0 = l0 ? xl1 : xl0;
        x1 = l0 ? xl0 : xl1;
        return;
    }
    for (int iter = 1;; ++iter) {
#ifdef QMRI_STATS
        iters_out = iter;
#endif
        if (par == 0.0) par = fmax(dwarf, 0.001 * paru);
        const double sp = sqrt(par);
        double rsd0, rsd1, s10;
        qrsolv2(r11, r12, r22, ir11, ir22, l0, sp * dg0, sp * dg1, qtb0, qtb1, x0, x1, rsd0, rsd1, s10);
        wa20 = dg0 * x0;
        wa21 = dg1 * x1;
        norm2r(wa20, wa21, dxnorm, idx);
        const double fp_old = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= fp_old && fp_old < 0.0) || iter == 10)
            break;
        double t0 = dl0 * ((l0 ? wa21 : wa20) * idx);
        double t1 = dl1 * ((l0 ? wa20 : wa21) * idx);
        t0 = t0 * rsd0;
        t1 = (t1 - s10 * t0) * rsd1;
        const double parc = fp * frcp(delta * (t0 * t0 + t1 * t1));
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

__device__ __forceinline__ double nan_to_num(double v, double nanv) {
    if (isnan(v)) return nanv;
    if (isinf(v)) return v > 0 ? DBL_MAX : -DBL_MAX;
    return v;
}

// numpy.around(v, d): multiply by 10**d, rint, divide (d >= 0); divide / rint / multiply (d < 0)
__device__ __forceinline__ double around(double v, int decimals, double p10) {
    if (decimals == 0) return rint(v);
    if (decimals > 0) return rint(v * p10) / p10;
    return rint(v / p10) * p10;
}

// final store of one voxel: raw (a, b, r2) -> reference post-processing -> popt / r2 / tc / info / nfev
__device__ __forceinline__ void finish_voxel(const FitKArgs &A, long long v, double pa, double pb,
                                             double r2, int info, int nfev, bool outside_mask) {
    double tc = pb;
    const qmri_post &P = A.post;
    if (outside_mask) {
        // scatter fill (fitting.py:205-215): nan_to_num value if given, else NaN -- also for r2
        const double fill = (P.enable && P.use_nan_to_num) ? P.nan_value : NAN;
        pa = pb = tc = r2 = fill;
    } else if (P.enable) {
        if (P.inv_abs_b) pb = 1.0 / fabs(pb);
        if (P.use_bounds) {
            if (pa < P.lb[0] || pa > P.ub[0]) pa = NAN;
            if (pb < P.lb[1] || pb > P.ub[1]) pb = NAN;
        }
        if (P.use_r2_thr && r2 < P.r2_threshold) pa = pb = NAN;
        if (P.use_nan_to_num) {
            pa = nan_to_num(pa, P.nan_value);
            pb = nan_to_num(pb, P.nan_value);
        }
        tc = pb;
    }
    if (A.out_f64) {
        double2 o;
        o.x = pa;
        o.y = pb;
        if (A.popt) static_cast<double2 *>(A.popt)[v] = o;
        static_cast<double *>(A.r2)[v] = r2;
    } else {
        float2 o;
        o.x = static_cast<float>(pa);
        o.y = static_cast<float>(pb);
        if (A.popt) static_cast<float2 *>(A.popt)[v] = o;
        static_cast<float *>(A.r2)[v] = static_cast<float>(r2);
    }
    if (A.tc) {
        if (P.enable && P.decimals != QMRI_NO_ROUND) tc = around(tc, P.decimals, A.p10);
        if (A.out_f64)
            static_cast<double *>(A.tc)[v] = tc;
        else
            static_cast<float *>(A.tc)[v] = static_cast<float>(tc);
    }
    if (A.info) A.info[v] = static_cast<signed char>(info);
    if (A.nfev) A.nfev[v] = static_cast<short>(nfev);
}

// ---- tile staging: global (any dtype, echo-major) -> wave-private LDS slice [E][kSub] of LT --------
template <typename S, typename LT>
__device__ __forceinline__ void stage_rows(const S *__restrict__ g, long long ld, int E, int count,
                                           LT *__restrict__ tile, int lane, bool vec_ok) {
    for (int e = 0; e < E; ++e) {
        const S *row = g + (long long)e * ld;
        LT *dst = tile + e * kSub;
        if (vec_ok && count == kSub) {
            // kVpl consecutive elements per lane: one 8-byte (f32) / 4-byte (i16) / 16-byte (f64) load at 2 per lane
            // non-temporal: every sample is read exactly once -- the streaming rows should not push the partially written
            // result lines of the tiles in flight out of the L2 (scattered per-voxel stores merge there or not at all)
            typedef S VecS __attribute__((ext_vector_type(kVpl)));
            const VecS t = __builtin_nontemporal_load(reinterpret_cast<const VecS *>(row + lane * kVpl));
            struct alignas(sizeof(LT) * kVpl) L4 {
                LT v[kVpl];
            };
            L4 o;
#pragma unroll
            for (int k = 0; k < kVpl; ++k) o.v[k] = static_cast<LT>(t[k]);
            *reinterpret_cast<L4 *>(dst + lane * kVpl) = o;
        } else {
#pragma unroll
            for (int k = 0; k < kSub / 64; ++k) {
                const int j = k * 64 + lane;
                dst[j] = j < count ? static_cast<LT>(row[j]) : LT(0);
            }
        }
    }
}

enum : int { ST_IDLE = 0, ST_INIT = 1, ST_ITER = 2, ST_DONE = 3 };  // DONE: converged, outputs not written yet

// Result ring of a wave: terminated voxels are parked here (voxel index + MINPACK info / nfev, a, b, |f|, SStot) and
// post-processed + stored 64 at a time with every lane active (see the refill block of the kernel).
constexpr int kRing = 128;        // entries: up to 63 waiting + 64 appended in one refill
constexpr int kRingBytes = kRing * 5 * 8;

// LDS bytes one wave owns: samples [E][kSub] of LT + SStot, a0, b0 (double) + the result ring + 1-byte queue entries
// (variants up to 8 samples only: from 9 samples on the ring would cost the fourth wave of a block -- 2 x 4 waves no longer
//  fit the CU's 160 KB -- and the in-lane epilogue is the better deal)
__host__ __device__ constexpr bool use_result_ring(int E) { return E <= 8 && !QMRI_TILE_OUT; }
// Tile-ordered retirement (variants up to 8 samples): a wave keeps the side arrays of TWO tiles -- SStot / a0 / b0 / |f| per
// voxel, whose a0 / b0 slots take the solver's (a, b) once the voxel has been pulled, + info, nfev and a status byte -- so that
// the stragglers of a tile can finish while the next one is being pulled; a tile whose last voxel has terminated is
// post-processed and stored as a whole: every output row of it in full, coalesced lines (see flush_tile in the kernel).
__host__ __device__ constexpr bool use_tile_out(int E) { return E <= 8 && QMRI_TILE_OUT; }
template <typename LT>
__host__ __device__ constexpr size_t lds_bytes_per_wave(int E) {
    return (size_t)E * kSub * sizeof(LT) +
           (use_tile_out(E) ? (size_t)2 * kSub * (4 * sizeof(double) + 4) + kSub
                            : (size_t)kSub * (3 * sizeof(double) + 1) + (use_result_ring(E) ? kRingBytes : 0));
}

// Two blocks (8 waves) per CU = two waves per SIMD is what the VALU-bound solver needs; the register allocation of the
// EMAX <= 8 variants sits within a few registers of the 256 that allows (252 for the headline variant, 256 + 2..46
// "AGPR" overflow for others = one wave per SIMD, -40 %), so the bound is stated: a handful of spills (<= 10 VGPRs) beats
// losing the second wave.  EMAX = 16: 14.0 ms instead of 16.8 for the exact-size variant (148 spills), but 52.9 instead
// of 18.8 for the partial one (340 spills); EMAX = 32 does not fit either way.
// LISTED: the launch walks a compact tile list made by the mask pre-pass (instantiated for the variants up to 8 samples; the
// others walk the list with the general code).  A listed launch is a region of interest -- typically a few tiles per wave --
// and partitions a SHORT list statically; keeping that logic out of the dense instantiation keeps the headline kernel's
// register allocation as it was (any extra state in its claim path cost 20 spilled VGPRs and 6 % at the 168-register limit).
template <int EMAX, bool FULL, typename LT, bool LISTED = false>
__global__ __launch_bounds__(256, EMAX <= 8 ? QMRI_SMALL_E_BLOCKS : ((EMAX <= 16 && FULL) ? 2 : QMRI_MIN_WAVES)) void monoexp_lm_kernel(
    const FitKArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int E = FULL ? EMAX : A.E;
    const double rE = 1.0 / (double)E;
    // wave-private LDS slice: samples [E][kSub] | SStot [kSub] | a0 [kSub] | b0 [kSub] | result ring [5][kRing] | queue [kSub]
    unsigned char *slice = smem + (size_t)wave * lds_bytes_per_wave<LT>(E);
    LT *tile = reinterpret_cast<LT *>(slice);
    double *t_sst = reinterpret_cast<double *>(slice + (size_t)E * kSub * sizeof(LT));
    double *t_a0 = t_sst + kSub;
    double *t_b0 = t_a0 + kSub;
    unsigned long long *r_vox = reinterpret_cast<unsigned long long *>(t_b0 + kSub);  // voxel | info << 40 | nfev << 48
    double *r_a = reinterpret_cast<double *>(r_vox + kRing);
    double *r_b = r_a + kRing;
    double *r_fn = r_b + kRing;
    double *r_sst = r_fn + kRing;
    constexpr bool kUseRing = use_result_ring(EMAX);
    constexpr bool kTileOut = use_tile_out(EMAX);
    // tile-ordered retirement: side arrays of tile buffer b at side0 + b * 4 * kSub doubles: SStot | a0 -> a | b0 -> b | |f|
    double *const side0 = t_sst;
    unsigned char *const o_info = reinterpret_cast<unsigned char *>(side0 + 8 * kSub);   // [2][kSub] MINPACK info
    unsigned short *const o_nfev = reinterpret_cast<unsigned short *>(o_info + 2 * kSub);  // [2][kSub]
    unsigned char *const o_stat = reinterpret_cast<unsigned char *>(o_nfev + 2 * kSub);   // [2][kSub] 0 pending, 1 solved, 2 skipped (NaN, 0), 3 outside the mask
    unsigned char *t_queue = kTileOut ? o_stat + 2 * kSub
                                      : (kUseRing ? reinterpret_cast<unsigned char *>(r_sst + kRing) : reinterpret_cast<unsigned char *>(t_b0 + kSub));
    int rhead = 0, rcount = 0;  // wave-uniform: first waiting entry, number of waiting entries
    int cur = 0;                        // tile buffer the queue / the pulls refer to
    int outst0 = 0, outst1 = 0;         // voxels of the tile in buffer b that need the solver and have not terminated
    bool act0 = false, act1 = false;    // buffer b holds a tile that has not been stored yet
    long long base0 = 0, base1 = 0;     // first voxel of that tile
    int cnt0 = 0, cnt1 = 0;             // its voxel count
    int tslot = 0;                      // per lane: (buffer << 7) | index of its voxel in the tile
    const double epsmch = DBL_EPSILON;
    // the sample times as an LDS table: the LM step reads x[i] with broadcast ds_read (LDS port) instead of holding 2 E
    // scalar registers across the whole loop -- they were the largest block the scalar allocator spilled and restored
    // (v_readlane = VALU slots) next to every use
    // Measured: at two waves per SIMD it paid from 9 samples on only (12 samples 6.96 -> 6.68 ms, 8 samples 6.07 -> 6.26 ms: the
    // table's read latency sits in the dependent chain of a short evaluation); at three waves per SIMD the third wave hides that
    // latency and the headline (8 samples) goes 18.15 -> 17.35 ms (same-box A/B).  16 samples: the loaded values push the vector
    // registers over 256 -> up to 12 samples.
    constexpr bool kXsLds = EMAX > QMRI_XS_MIN && EMAX <= 12;
    __shared__ double xs_tab[kXsLds ? EMAX : 1];
    if (kXsLds) {
        if (threadIdx.x < EMAX) xs_tab[threadIdx.x] = (FULL || (int)threadIdx.x < A.E) ? A.x[threadIdx.x] : 0.0;
        __syncthreads();
    }
#define QMRI_XS(i) (kXsLds ? xs_tab[(i)] : A.x[(i)])

    // ---- per-lane LM state (fp64 registers) ----
    int state = ST_IDLE;
    long long vox = 0;
    // the voxel's samples: as doubles where the registers allow (EMAX <= 8: 8 conversions less per round), else as loaded
    typedef typename std::conditional<(EMAX <= QMRI_Y_F64_MAX), double, LT>::type YT;
    YT yv[EMAX];
    double pa = 0, pb = 0;                    // current point x = (a, b)
    double fnorm = 0, par = 0, delta = 0, xnorm = 0, gnorm = 0;
    double dg0 = 1, dg1 = 1;                  // diag (by parameter)
    double r11 = 0, r12 = 0, r22 = 0;         // R of the pivoted QR
    double ir11 = 0, ir22 = 0;                // 1/r11, 1/r22
    double idg0 = 1, idg1 = 1;                // 1/diag
    double rfn = 0;                           // 1/fnorm
    double qtf0 = 0, qtf1 = 0;                // first two components of Q^T fvec
    double sstot = 0;
    int l0 = 0;                               // ipvt(0): 0 = columns in order, 1 = swapped
    int nfev = 0;
    int done_info = 0;                        // MINPACK info of a lane parked in ST_DONE
    bool first = true;                        // MINPACK iter == 1

    // ---- wave-uniform queue of the fit-able voxels of the current tile ----
    long long tile_base = 0;
    int qpos = 0, qend = 0;
    bool more = true;
    const long long ntiles = A.tile_list ? (long long)*A.tile_list_count : (A.N + kSub - 1) / kSub;
    const int nwaves = (int)gridDim.x * (int)(blockDim.x >> 6);
    unsigned int tnext = 0, tend = 0;  // tiles this wave has claimed
    unsigned int chunk;
    {
        const long long want = ntiles / ((long long)nwaves * 4);
        chunk = want > 16 ? 16u : (want < 1 ? 1u : (unsigned int)want);
    }
    // LISTED: a SHORT list (a thin ROI: a few tiles per wave) is partitioned statically, wave w takes tiles [w, w + 1) *
    // ceil(ntiles / nwaves): with one atomic per claim plus one per wave to find the list exhausted, 3 072 waves sharing 5 600
    // tiles spent 100 of the kernel's 178 us on the counter word.  Long lists keep the guided self-scheduling below.
    bool static_part = false;
    if constexpr (LISTED) {
        if (ntiles <= (long long)nwaves * 4) {
            static_part = true;
            const long long per = (ntiles + nwaves - 1) / nwaves;
            const long long w0 = (long long)((int)blockIdx.x * (int)(blockDim.x >> 6) + wave) * per;
            tnext = (unsigned int)(w0 < ntiles ? w0 : ntiles);
            tend = (unsigned int)(w0 + per < ntiles ? w0 + per : ntiles);
        }
    }

    // post-processing + stores of the first n (<= 64) waiting ring entries, one per lane
    auto flush_ring = [&](const FitKArgs &K, int n) {
        if (lane < n) {
            const int slot = (rhead + lane) & (kRing - 1);
            const unsigned long long w = r_vox[slot];
            const long long v = (long long)(w & ((1ull << 40) - 1));
            const int info = (int)(signed char)((w >> 40) & 0xFF);
            const int nf = (int)(w >> 48);
            // fitting.py:1032-1035 (success) / :1069-1072 (RuntimeError -> NaN, 0)
            double oa = NAN, ob = NAN, r2 = 0.0;
            if (info >= 1 && info <= 4) {
                const double fn = r_fn[slot];
                oa = r_a[slot];
                ob = r_b[slot];
                r2 = 1.0 - (fn * fn) / (r_sst[slot] + K.r2_eps);
            }
#ifdef QMRI_FIT_NOSTORE  // timing experiment (results wrong): what do the solver's result stores + epilogue cost?
            if (v == -1)
#endif
            finish_voxel(K, v, oa, ob, r2, info, nf, false);
        }
        rhead = (rhead + n) & (kRing - 1);
        rcount -= n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // tile-ordered retirement: post-processing + stores of the whole tile in buffer b, lane l takes voxels l, l + 64: every
    // output row of the tile is written in full, consecutive lines (the per-voxel stores of the ring were 1.5 x the bytes)
    auto flush_tile = [&](const FitKArgs &K, int b) {
        const long long base = b ? base1 : base0;
        const int cnt = b ? cnt1 : cnt0;
        const double *S = side0 + b * 4 * kSub;
#pragma unroll
        for (int k = 0; k < kVpl; ++k) {
            const int j = k * 64 + lane;
            if (j < cnt) {
                const int st = o_stat[b * kSub + j];
                const int info = st == 1 ? (int)(signed char)o_info[b * kSub + j] : (st == 3 ? -1 : 0);
                const int nf = st == 1 ? (int)o_nfev[b * kSub + j] : 0;
                // fitting.py:1032-1035 (success) / :1069-1072 (RuntimeError -> NaN, 0) / :1064-1067 (skip rule) / :205-215 (fill)
                double oa = NAN, ob = NAN, r2 = 0.0;
                if (st == 1 && info >= 1 && info <= 4) {
                    const double fn = S[3 * kSub + j];
                    oa = S[kSub + j];
                    ob = S[2 * kSub + j];
                    r2 = 1.0 - (fn * fn) / (S[j] + K.r2_eps);
                }
                finish_voxel(K, base + j, oa, ob, r2, info, nf, st == 3);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

#ifdef QMRI_STATS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t1 = 0;
#endif
    for (;;) {
        QMRI_STAT_ADD(0, 1);
        QMRI_TIC();
        // ======================= refill: idle lanes pull voxels =======================
        {
            // The arguments only the refill / epilogue needs (pointers, mask / init / post-processing options) are read from
            // the kernel-argument segment HERE, through a pointer the compiler cannot see through: left to itself it loads
            // all ~110 scalar registers' worth of FitKArgs once, keeps them live across the LM step below and spills the
            // excess into VGPR lanes (192 spills, ~170 v_readlane in the hot loop).  An s_load per refill costs nothing.
            typedef const __attribute__((address_space(4))) FitKArgs *KArgP;
            KArgP kp_ = (KArgP)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp_));
            const FitKArgs &C = *(const FitKArgs *)kp_;
            // A lane that has terminated parks in ST_DONE: its outputs are written here, together with those of the other
            // lanes that finished since the last refill (finishing on the spot ran the ~150-instruction epilogue in
            // almost every round for the ~3 lanes of 64 that terminate per round).
            unsigned long long idle = __ballot(state == ST_IDLE || state == ST_DONE);
            const int nidle = __popcll(idle);
            if (nidle >= A.refill_idle || nidle == 64) {
                // Terminated lanes do NOT run the reference's post-processing here: a refill finds 8-16 of them, and the
                // ~150-instruction epilogue (r2, 1/|b|, bounds, nan_to_num, rounding, three stores) at 15-25 % lane
                // occupancy was 8 % of the kernel (measured with the stores + epilogue compiled out: 21.4 -> 19.7 ms).  They
                // append (voxel, info, nfev, a, b, |f|, SStot) to the wave's result ring -- five LDS writes -- and the
                // epilogue runs on 64 ring entries at a time with every lane active (flush_ring).
                if (kTileOut) {
                    // terminated lanes leave (a, b, |f|, info, nfev) in their voxel's slots of its tile buffer -- three LDS
                    // writes + three bytes -- and a tile whose last voxel has terminated is retired as a whole
                    const bool done = state == ST_DONE;
                    const unsigned long long dmask = __ballot(done);
                    if (dmask) {
                        const int b = tslot >> 7, j = tslot & (kSub - 1);
                        if (done) {
                            double *S = side0 + b * 4 * kSub;
                            S[kSub + j] = pa;
                            S[2 * kSub + j] = pb;
                            S[3 * kSub + j] = fnorm;
                            o_info[b * kSub + j] = (unsigned char)done_info;
                            o_nfev[b * kSub + j] = (unsigned short)nfev;
                            o_stat[b * kSub + j] = 1;
                            state = ST_IDLE;
                            nfev = 0;
                        }
                        const int n1 = __popcll(__ballot(done && b == 1));
                        outst1 -= n1;
                        outst0 -= __popcll(dmask) - n1;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (act0 && outst0 == 0) {
                        flush_tile(C, 0);
                        act0 = false;
                    }
                    if (act1 && outst1 == 0) {
                        flush_tile(C, 1);
                        act1 = false;
                    }
                } else if (!kUseRing) {
                    if (state == ST_DONE) {
                        double oa = NAN, ob = NAN, r2 = 0.0;
                        if (done_info >= 1 && done_info <= 4) {
                            oa = pa;
                            ob = pb;
                            r2 = 1.0 - (fnorm * fnorm) / (sstot + C.r2_eps);
                        }
                        finish_voxel(C, vox, oa, ob, r2, done_info, nfev, false);
                        state = ST_IDLE;
                        nfev = 0;
                    }
                } else {
                    const unsigned long long dmask = __ballot(state == ST_DONE);
                    if (state == ST_DONE) {
                        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(dmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dmask, 0u));
                        const int slot = (rhead + rcount + rank) & (kRing - 1);
                        r_vox[slot] = (unsigned long long)vox | ((unsigned long long)(done_info & 0xFF) << 40) |
                                      ((unsigned long long)(nfev & 0xFFFF) << 48);
                        r_a[slot] = pa;
                        r_b[slot] = pb;
                        r_fn[slot] = fnorm;
                        r_sst[slot] = sstot;
                        state = ST_IDLE;
                        nfev = 0;
                    }
                    rcount += __popcll(dmask);
                    if (rcount >= 64) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        flush_ring(C, 64);
                    }
                }
                // ---- queue empty: claim tiles until one has fit-able voxels (or the volume is done) ----
                while (qpos >= qend && more) {
                    int nb = 0;  // (tile-ordered retirement) the buffer the next tile goes to: one that holds no unstored tile
                    if (kTileOut) {
                        const bool other_free = cur ? !act0 : !act1, cur_free = cur ? !act1 : !act0;
                        if (!other_free && !cur_free) break;  // stragglers of both tiles still run: pull again when one has retired
                        nb = other_free ? (cur ^ 1) : cur;
                    }
                    // guided self-scheduling: one atomic claims `chunk` consecutive tiles (16 early, 1 at the end).
                    // One atomic per tile made the single counter the bottleneck of sparse volumes: a 2 % ROI mask
                    // over 17.7 M voxels is 69 k claims for 0.05 ms of fitting -> 0.83 ms (BASELINE configs[2]).
                    if (tnext >= tend) {
                        if (LISTED && static_part) {  // statically partitioned list: this wave's share is done
                            more = false;
                            break;
                        }
                        unsigned int t0 = 0;
                        if (lane == 0) t0 = atomicAdd(C.tile_counter, chunk);
                        t0 = __builtin_amdgcn_readfirstlane(t0);
                        tnext = t0;
                        tend = t0 + chunk;
                        const long long left = ntiles - (long long)tend;
                        const long long want = left / ((long long)nwaves * 4);
                        chunk = want > 16 ? 16u : (want < 1 ? 1u : (unsigned int)want);
                    }
                    const unsigned int ti = tnext++;
                    if ((long long)ti >= ntiles) {
                        more = false;
                        break;
                    }
                    const unsigned int t = C.tile_list ? C.tile_list[ti] : ti;
                    const long long start = (long long)t * kSub;
                    tile_base = start;
                    const long long rem = C.N - start;
                    const int count = rem < kSub ? (int)rem : kSub;
                    if (kTileOut) {
                        cur = nb;
                        t_sst = side0 + nb * 4 * kSub;
                        t_a0 = t_sst + kSub;
                        t_b0 = t_a0 + kSub;
                        if (nb) { base1 = start; cnt1 = count; } else { base0 = start; cnt0 = count; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    bool any_selected = true;  // a tile without a single voxel in the mask needs no samples at all
                    if (C.mask) {
                        bool s = false;
                        for (int k = 0; k < kSub / 64; ++k) {
                            const int j = k * 64 + lane;
                            if (j < count) s = s || C.mask[start + j] != 0;
                        }
                        any_selected = __ballot(s) != 0;
                    }
                    if (any_selected)
                    switch (C.y_dtype) {
                        case QMRI_F32:
                            stage_rows(static_cast<const float *>(C.y) + start, C.ld, E, count, tile,
                                       lane, C.vec_ok);
                            break;
                        case QMRI_F64:
                            stage_rows(static_cast<const double *>(C.y) + start, C.ld, E, count, tile,
                                       lane, C.vec_ok);
                            break;
                        case QMRI_I16:
                            stage_rows(static_cast<const short *>(C.y) + start, C.ld, E, count, tile,
                                       lane, C.vec_ok);
                            break;
                        default:
                            stage_rows(static_cast<const unsigned short *>(C.y) + start, C.ld, E, count,
                                       tile, lane, C.vec_ok);
                            break;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // ---- tile preparation with ALL lanes (4 voxels per lane, coalesced side traffic):
                    // mask / skip / non-finite rules, SStot, initial guess, and the compacted queue of
                    // voxels that really need the solver.  Skipped voxels are finished right here.
                    int qn = 0;
#pragma unroll 1
                    for (int k = 0; k < kSub / 64; ++k) {
                        const int j = k * 64 + lane;
                        bool need_fit = false;
                        if (j < count) {
                            const long long v = start + j;
                            bool selected = true;
                            if (C.mask) selected = C.mask[v] != 0;
                            if (!selected) {
                                if (kTileOut) o_stat[cur * kSub + j] = 3;
                                else finish_voxel(C, v, 0, 0, 0, -1, 0, true);
                            } else {
                                bool allzero = true, finite = true, oob = false;
                                double mean = 0.0;
                                double sv[EMAX];
#pragma unroll
                                for (int i = 0; i < EMAX; ++i)
                                    if (FULL || i < E) {
                                        const double q = static_cast<double>(tile[i * kSub + j]);
                                        sv[i] = q;
                                        allzero = allzero && (q == 0.0);
                                        finite = finite && isfinite(q);
                                        if (C.use_y_bounds) oob = oob || q < C.y_lo || q > C.y_hi;
                                        mean += q;
                                    }
                                if (!finite) {
                                    // reference: ValueError for the whole call (scipy check_finite)
                                    *C.nonfinite = 1;
                                    if (kTileOut) o_stat[cur * kSub + j] = 2;
                                    else finish_voxel(C, v, NAN, NAN, 0.0, 0, 0, false);
                                } else if (allzero || oob) {
                                    // skip rule, fitting.py:1064-1067
                                    if (kTileOut) o_stat[cur * kSub + j] = 2;
                                    else finish_voxel(C, v, NAN, NAN, 0.0, 0, 0, false);
                                } else {
                                    need_fit = true;
                                    if (kTileOut) o_stat[cur * kSub + j] = 0;
                                    mean = mean * rE;
                                    double st = 0.0;
#pragma unroll
                                    for (int i = 0; i < EMAX; ++i)
                                        if (FULL || i < E) {
                                            const double d = sv[i] - mean;
                                            st += d * d;
                                        }
                                    t_sst[j] = st;
                                    if (C.init == QMRI_INIT_PER_VOXEL) {
                                        t_a0[j] = C.a0v ? C.a0v[v] : C.a0;
                                        t_b0[j] = C.b0v ? C.b0v[v] : C.b0;
                                    } else if (C.init == QMRI_INIT_LOGLIN) {
                                        // fitting.py:701-718: v + 1e-10*(v==0); log; degree-1 LS in x;
                                        // r2 on the log data; r2 < 0 or NaN -> params 0 -> p0 = (1, 0)
                                        double sl = 0.0;
#pragma unroll
                                        for (int i = 0; i < EMAX; ++i)
                                            if (FULL || i < E) {
                                                double q = sv[i];
                                                if (q == 0.0) q = 1e-10;
                                                // (log_sk of fp64_fast.h: ~35 instructions with the constants as scalar operands made in place; the
                                                // device library's log is 78 and pins 14 scalar registers of polynomial constants across the whole
                                                // main loop, for every recipe)
                                                sv[i] = log_sk(q);
                                                sl += sv[i];
                                            }
                                        const double lmean = sl * rE;
                                        double sxy = 0.0, syy = 0.0;
#pragma unroll
                                        for (int i = 0; i < EMAX; ++i)
                                            if (FULL || i < E) {
                                                const double dy = sv[i] - lmean;
                                                sxy += ((kXsLds ? xs_tab[i] : C.x[i]) - C.xmean) * dy;
                                                syy += dy * dy;
                                            }
                                        const double slope = div_fast(sxy, C.sxx);
                                        const double icpt = lmean - slope * C.xmean;
                                        double ssr = 0.0;
#pragma unroll
                                        for (int i = 0; i < EMAX; ++i)
                                            if (FULL || i < E) {
                                                const double r = (slope * (kXsLds ? xs_tab[i] : C.x[i]) + icpt) - sv[i];
                                                ssr += r * r;
                                            }
                                        const double r2l = 1.0 - div_fast(ssr, syy + 1e-8);
                                        const bool good = r2l >= 0.0 && !isnan(slope) && !isnan(icpt);
                                        t_a0[j] = good ? exp_sk(icpt) : 1.0;
                                        t_b0[j] = good ? slope : 0.0;
                                    }
                                }
                            }
                        }
                        const unsigned long long fit = __ballot(need_fit);
                        if (need_fit) {
                            const int r = __builtin_amdgcn_mbcnt_hi(
                                (unsigned)(fit >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fit, 0u));
                            t_queue[qn + r] = (unsigned char)j;
                        }
                        qn += __popcll(fit);
                    }
                    qpos = 0;
                    qend = qn;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (kTileOut) {
                        if (qn == 0) {
                            flush_tile(C, cur);  // nothing for the solver in this tile: it retires at once
                        } else if (cur) {
                            act1 = true;
                            outst1 = qn;
                        } else {
                            act0 = true;
                            outst0 = qn;
                        }
                    }
                }
                // ---- pull: idle lane number r takes queue entry qpos + r ----
                if (qpos < qend) {
                    const int rank = __builtin_amdgcn_mbcnt_hi(
                        (unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                    const int q = qpos + rank;
                    if (state == ST_IDLE && q < qend) {
                        const int j = t_queue[q];
#pragma unroll
                        for (int i = 0; i < EMAX; ++i)
                            if (FULL || i < E) yv[i] = tile[i * kSub + j];
                            else yv[i] = 0;
                        sstot = t_sst[j];
                        vox = tile_base + j;
                        tslot = (cur << 7) | j;
                        pa = C.a0;
                        pb = C.b0;
                        if (C.init != QMRI_INIT_SCALAR) {
                            pa = t_a0[j];
                            pb = t_b0[j];
                        }
                        state = ST_INIT;
                    }
                    qpos = qpos + nidle;
                    if (qpos > qend) qpos = qend;
                }
            }
        }
        QMRI_TOC(8);
        if (!__ballot(state != ST_IDLE)) break;
        QMRI_STAT_ADD(1, __popcll(__ballot(state == ST_INIT || state == ST_ITER)));
        QMRI_STAT_ADD(2, __popcll(__ballot(state == ST_ITER)));

        // ======================= one LM step for every busy lane =======================
#ifdef QMRI_STATS
        int lm_it_lane = 0;
        bool did_qr = false, did_finish = false;
#endif
        if (state == ST_INIT || state == ST_ITER) {
            double ta, tb;           // trial point
            double p0 = 0, p1 = 0;   // step (by parameter)
            double pnorm = 0;
            if (state == ST_ITER) {
#ifdef QMRI_STATS
                int lm_it = 0;
                lmpar2(r11, r12, r22, ir11, ir22, l0, dg0, dg1, idg0, idg1, qtf0, qtf1, delta, par,
                       p0, p1, A.lmpar_closed_form != 0, lm_it, st_acc, st_t1);
                QMRI_TOC(10);
                lm_it_lane = lm_it;
#else
                lmpar2(r11, r12, r22, ir11, ir22, l0, dg0, dg1, idg0, idg1, qtf0, qtf1, delta, par,
                       p0, p1, A.lmpar_closed_form != 0);
#endif
                p0 = -p0;
                p1 = -p1;
                ta = pa + p0;
                tb = pb + p1;
                pnorm = norm2(dg0 * p0, dg1 * p1);
                if (first) delta = min_q(delta, pnorm);
            } else {
                ta = pa;
                tb = pb;
            }
            QMRI_TOC(10);
            // ---- evaluate the model at the trial point: E exps shared by fvec and the Jacobian ----
            double ev[EMAX], fv[EMAX];
            double ss = 0.0;
            if (A.uniform_x) {
                // equally spaced x: e_i = e_0 q^i from two exponentials (a few ulp of accumulated rounding, far inside
                // the 1e-4 parity bar; x_0 >= 0 < x_step keeps 0 * inf out of the products)
                const double q1 = exp_sk(mul_rn(tb, A.x_step));
                const double q2 = q1 * q1, q4 = q2 * q2;
                if (A.x0_pow >= 0) {  // x_0 = k x_step (TE = dTE, 2 dTE, ...): one exponential for the whole voxel
                    double e0 = 1.0;
                    for (int k = 0; k < A.x0_pow; ++k) e0 *= q1;
                    ev[0] = e0;
                } else {
                    ev[0] = exp_sk(mul_rn(tb, QMRI_XS(0)));
                }
#pragma unroll
                for (int i = 1; i < EMAX; ++i)
                    ev[i] = i >= 4 ? ev[i - 4] * q4 : (i >= 2 ? ev[i - 2] * q2 : ev[0] * q1);
            } else {
#pragma unroll
                for (int i = 0; i < EMAX; ++i)
                    if (FULL || i < E) ev[i] = exp_sk(mul_rn(tb, QMRI_XS(i)));
                    else ev[i] = 0.0;
            }
#pragma unroll
            for (int i = 0; i < EMAX; ++i)
                {
                    // partial variants (E < EMAX): the padded residuals and Jacobian rows are exact zeros, produced here and
                    // in the forward differences below; every consumer loop runs over all EMAX entries unguarded (sums of
                    // zeros), which keeps the partial kernels within a few registers of the exact-size ones
                    fv[i] = (FULL || i < E) ? sub_rn(mul_rn(ta, ev[i]), static_cast<double>(yv[i])) : 0.0;
                    ss += fv[i] * fv[i];
                }
            double fnorm1, rfn1 = 0.0;  // rfn1 = 1 / fnorm1 on the fast path, else 0 (-> frcp when needed)
            if (__builtin_expect(is_pos_normal(ss), 1)) {
                sqrt_rsqrt(ss, fnorm1, rfn1);
            } else {
                QMRI_COLD_PATH();
                double m = 0.0;
                bool anynan = false;
#pragma unroll
                for (int i = 0; i < EMAX; ++i)
                    {
                        m = fmax(m, fabs(fv[i]));
                        anynan = anynan || isnan(fv[i]);
                    }
                if (anynan) {
                    fnorm1 = NAN;
                } else if (!(m > 0.0) || isinf(m)) {
                    fnorm1 = m;
                } else {
                    double s2 = 0.0;
#pragma unroll
                    for (int i = 0; i < EMAX; ++i)
                        {
                            const double r = fv[i] / m;
                            s2 += r * r;
                        }
                    fnorm1 = m * sqrt(s2);
                }
            }
            ++nfev;
            QMRI_TOC(11);

            bool accepted;
            int info = 0;
            if (state == ST_ITER) {
                double actred = -1.0;
                if (0.1 * fnorm1 < fnorm) {
                    const double t = fnorm1 * rfn;
                    actred = 1.0 - t * t;
                }
                // R * P^T p;  prered = (||R P^T p|| / fnorm)^2 + 2 (sqrt(par) pnorm / fnorm)^2
                const double pl0 = l0 ? p1 : p0, pl1 = l0 ? p0 : p1;
                const double w0 = r11 * pl0 + r12 * pl1, w1 = r22 * pl1;
                const double t1sq = (w0 * w0 + w1 * w1) * rfn * rfn;
                const double pr = pnorm * rfn;
                const double t2sq = par * pr * pr;
                const double prered = t1sq + t2sq / 0.5;
                const double dirder = -(t1sq + t2sq);
                double ratio = 0.0;
                if (prered != 0.0) ratio = actred * frcp(prered);
                if (ratio <= 0.25) {
                    double temp = 0.5;
                    if (actred < 0.0) temp = 0.5 * dirder * frcp(dirder + 0.5 * actred);
                    if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                    delta = temp * min_q(delta, pnorm * 10.0);
                    par = par * frcp(temp);
                } else if (par == 0.0 || ratio >= 0.75) {
                    delta = pnorm / 0.5;
                    par = 0.5 * par;
                }
                accepted = ratio >= 1e-4;
                if (accepted) {
                    pa = ta;
                    pb = tb;
                    xnorm = norm2(dg0 * pa, dg1 * pb);
                    fnorm = fnorm1;
                    rfn = rfn1 != 0.0 ? rfn1 : frcp(fnorm);
                    first = false;
                }
                const bool small = fabs(actred) <= A.ftol && prered <= A.ftol && 0.5 * ratio <= 1.0;
                if (small) info = 1;
                if (delta <= A.xtol * xnorm) info = small ? 3 : 2;
                if (info == 0) {
                    if (nfev >= A.maxfev) info = 5;
                    if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
                    if (delta <= epsmch * xnorm) info = 7;
                    if (gnorm <= epsmch) info = 8;
                }
            } else {
                accepted = true;
                fnorm = fnorm1;
                rfn = rfn1 != 0.0 ? rfn1 : frcp(fnorm);
                par = 0.0;
                first = true;
            }

            QMRI_TOC(12);
            if (info == 0 && accepted) {
#ifdef QMRI_STATS
                did_qr = true;
#endif
                // ---- Jacobian at the (new) current point + Householder QR with column pivoting ----
                // lmdif's forward differences (fdjac2: h_j = sqrt(eps)*|x_j|, J_j = (f(x+h_j e_j)-f)/h_j),
                // charged n = 2 evaluations, but evaluated WITHOUT new exponentials: the a-column reuses
                // e_i exactly; the b-column uses e_i * exp(d_i), d_i = (b+h)x_i - b x_i ~ 1e-8 |b x_i|,
                // exp(d) = 1 + d + d^2/2 + d^3/6 (error < 1e-30).  This keeps the truncation bias and
                // the cancellation-to-zero behaviour of the reference's Jacobian (an analytic J does
                // not: it keeps iterating where lmdif sees a zero gradient and stops with info = 4).
                nfev += 2;
                double c2[EMAX];
                double n1 = 0.0, n2 = 0.0;
                {
                    const double eps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON)
                    double ha = eps * fabs(pa), hb = eps * fabs(pb);
                    if (ha == 0.0) ha = eps;
                    if (hb == 0.0) hb = eps;
                    const double a1 = pa + ha, b1 = pb + hb;
                    const double rha = frcp(ha), rhb = frcp(hb);
#pragma unroll
                    for (int i = 0; i < EMAX; ++i)
                        if (FULL || i < E) {
                            const double yi = static_cast<double>(yv[i]);
                            const double e = ev[i];
                            const double xi = QMRI_XS(i);
                            const double d = sub_rn(mul_rn(b1, xi), mul_rn(pb, xi));
                            const double e1 = e + e * (d + d * d * (0.5 + d * (1.0 / 6.0)));
                            // the columns are stored in the order the pivoting picks on tissue data (|d/db| = a x e >> |d/da| = e):
                            // ev <- column b, c2 <- column a, exchanged below only when column a is the longer one
                            const double jb = sub_rn(sub_rn(mul_rn(pa, e1), yi), fv[i]) * rhb;
                            const double ja = sub_rn(sub_rn(mul_rn(a1, e), yi), fv[i]) * rha;
                            ev[i] = jb;
                            c2[i] = ja;
                            n1 += ja * ja;
                            n2 += jb * jb;
                        } else {
                            c2[i] = 0.0;
                            ev[i] = 0.0;
                        }
                }
                // (E-vector norms: sums of squares of finite values can only overflow to inf, never NaN)
                QMRI_TOC(13);
                double acn0, acn1, iacn0, iacn1;
                sqrt_rsqrt(n1, acn0, iacn0);
                sqrt_rsqrt(n2, acn1, iacn1);
                l0 = acn1 > acn0 ? 1 : 0;
                // P = pivot column, Q = the other one (in place: ev <- P, c2 <- Q)
                if (__builtin_expect(!l0, 0)) {
#pragma unroll
                    for (int i = 0; i < EMAX; ++i)
                        {
                            const double t = ev[i];
                            ev[i] = c2[i];
                            c2[i] = t;
                        }
                }
                double ajn = l0 ? acn1 : acn0;
                double iajn = l0 ? iacn1 : iacn0;
                if (ajn != 0.0) {
                    if (ev[0] < 0.0) {
                        ajn = -ajn;
                        iajn = -iajn;
                    }
                    double s1 = 0.0, s2 = 0.0;
#pragma unroll
                    for (int i = 0; i < EMAX; ++i)
                        {
                            ev[i] = ev[i] * iajn;
                            if (i == 0) ev[0] += 1.0;
                            s1 += ev[i] * c2[i];
                            s2 += ev[i] * fv[i];
                        }
                    const double iv0 = frcp(ev[0]);  // ev[0] in [1, 2]
                    const double t1 = s1 * iv0;
                    const double t2 = -s2 * iv0;
#pragma unroll
                    for (int i = 0; i < EMAX; ++i)
                        {
                            c2[i] -= t1 * ev[i];
                            fv[i] += t2 * ev[i];
                        }
                }
                r11 = -ajn;
                ir11 = -iajn;
                r12 = c2[0];
                qtf0 = fv[0];
                double m2 = 0.0;
#pragma unroll
                for (int i = 1; i < EMAX; ++i)
                    m2 += c2[i] * c2[i];
                double ajn2, iajn2;
                sqrt_rsqrt(m2, ajn2, iajn2);
                if (ajn2 != 0.0) {
                    if (c2[1] < 0.0) {
                        ajn2 = -ajn2;
                        iajn2 = -iajn2;
                    }
                    double s = 0.0;
#pragma unroll
                    for (int i = 1; i < EMAX; ++i)
                        {
                            c2[i] = c2[i] * iajn2;
                            if (i == 1) c2[1] += 1.0;
                            s += c2[i] * fv[i];
                        }
                    // only the 2nd component of Q^T fvec is needed after the 2nd reflection:
                    // fv[1] += (-s / v[1]) * v[1]
                    fv[1] -= s;
                }
                r22 = -ajn2;
                ir22 = -iajn2;
                qtf1 = fv[1];
                if (state == ST_INIT) {
                    dg0 = acn0 == 0.0 ? 1.0 : acn0;
                    dg1 = acn1 == 0.0 ? 1.0 : acn1;
                    idg0 = acn0 == 0.0 ? 1.0 : iacn0;
                    idg1 = acn1 == 0.0 ? 1.0 : iacn1;
                    xnorm = norm2(dg0 * pa, dg1 * pb);
                    delta = A.factor * xnorm;
                    if (delta == 0.0) delta = A.factor;
                }
                // scaled gradient norm
                gnorm = 0.0;
                if (fnorm != 0.0) {
                    const double al0 = l0 ? acn1 : acn0, al1 = l0 ? acn0 : acn1;
                    const double ial0 = l0 ? iacn1 : iacn0, ial1 = l0 ? iacn0 : iacn1;
                    const double q0 = qtf0 * rfn, q1 = qtf1 * rfn;
                    if (al0 != 0.0) gnorm = fabs((r11 * q0) * ial0);
                    if (al1 != 0.0) gnorm = fmax(gnorm, fabs((r12 * q0 + r22 * q1) * ial1));
                }
#ifdef QMRI_TRACE
                printf("qr x=(%.17g,%.17g) acn=(%.17g,%.17g) l0=%d R=(%.17g,%.17g,%.17g) qtf=(%.17g,%.17g) gnorm=%.6g\n",
                       pa, pb, acn0, acn1, l0, r11, r12, r22, qtf0, qtf1, gnorm);
#endif
                if (gnorm <= A.gtol) info = 4;
                if (acn0 > dg0) {
                    dg0 = acn0;
                    idg0 = iacn0;
                }
                if (acn1 > dg1) {
                    dg1 = acn1;
                    idg1 = iacn1;
                }
                state = ST_ITER;
            }

            QMRI_TOC(14);
            if (info != 0) {
                done_info = info;
                state = ST_DONE;
#ifdef QMRI_STATS
                did_finish = true;
#endif
            }
        }
#ifdef QMRI_STATS
        for (int k = 1; k <= 10; ++k) {
            const unsigned long long m = __ballot(lm_it_lane >= k);
            st_acc[3] += __popcll(m);
            st_acc[4] += m ? 1 : 0;
        }
        st_acc[5] += __popcll(__ballot(did_qr));
        st_acc[6] += __popcll(__ballot(did_finish));
#endif
    }
    if (kTileOut) {  // (the loop ends with every lane idle: every tile has retired in the last refill visit; belt and braces)
        if (act0 && outst0 == 0) flush_tile(A, 0);
        if (act1 && outst1 == 0) flush_tile(A, 1);
    }
    if (rcount > 0) {  // (the loop ends with every lane idle: what is left in the ring is < 64 entries)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        flush_ring(A, rcount);
    }
#ifdef QMRI_STATS
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_fit_stats[i], st_acc[i]);
#endif
}

// The scatter fill of fitting.py:205-215 for one whole tile outside the mask: the same values finish_voxel writes with
// outside_mask = true, as 16-byte stores (a pre-pass group is 256 consecutive voxels, so every output row of it is one
// aligned contiguous run).
__device__ __forceinline__ void fill_tile(const FitKArgs &A, long long start, int lane) {
    const qmri_post &P = A.post;
    const double fill = (P.enable && P.use_nan_to_num) ? P.nan_value : NAN;
    double tc = fill;
    if (A.tc && P.enable && P.decimals != QMRI_NO_ROUND) tc = around(tc, P.decimals, A.p10);
    const long long v0 = start + 4 * lane;  // this lane's 4 voxels
    if (A.out_f64) {
        const double2 f2 = {fill, fill};
        if (A.popt) {
            double2 *p = static_cast<double2 *>(A.popt) + v0;
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = f2;
        }
        double2 *r = reinterpret_cast<double2 *>(static_cast<double *>(A.r2) + v0);
        r[0] = f2;
        r[1] = f2;
        if (A.tc) {
            const double2 t2 = {tc, tc};
            double2 *q = reinterpret_cast<double2 *>(static_cast<double *>(A.tc) + v0);
            q[0] = t2;
            q[1] = t2;
        }
    } else {
        const float ff = static_cast<float>(fill);
        const float4 f4 = {ff, ff, ff, ff};
        if (A.popt) {
            float4 *p = reinterpret_cast<float4 *>(static_cast<float2 *>(A.popt) + v0);
            p[0] = f4;
            p[1] = f4;
        }
        *reinterpret_cast<float4 *>(static_cast<float *>(A.r2) + v0) = f4;
        if (A.tc) {
            const float tf = static_cast<float>(tc);
            *reinterpret_cast<float4 *>(static_cast<float *>(A.tc) + v0) = float4{tf, tf, tf, tf};
        }
    }
    if (A.info) *reinterpret_cast<unsigned int *>(A.info + v0) = 0xFFFFFFFFu;  // info = -1 four times
    if (A.nfev) *reinterpret_cast<unsigned long long *>(A.nfev + v0) = 0ull;
}

// ---- masked volumes: tile classification pre-pass ---------------------------------------------------
// One wave per 256-voxel group: no voxel selected -> the scatter fill of fitting.py:205-215 for the whole group
// (streaming writes); otherwise the tile index goes to a compact list that the fit kernel walks.  A cartilage ROI is
// ~2 % of the voxels in one or two slabs: without the list the fit waves spend their time claiming empty tiles, and
// the few waves whose claims fall inside the slab do all the fitting.
__global__ __launch_bounds__(256) void monoexp_mask_prepass_kernel(const FitKArgs A, unsigned int *list,
                                                                   unsigned int *count) {
    const int lane = threadIdx.x & 63;
    // The pre-pass works on GROUPS of kPre = 256 voxels (one 4-byte mask word per lane, 16-byte fills) whatever the fit kernel's
    // tile size; a group with selected voxels puts its kPre / kSub tiles -- those that hold a selected voxel -- on the list.
    constexpr int kPre = 256, kPerGroup = kPre / kSub, kLanesPerTile = 64 / kPerGroup;
    const long long ngroups = (A.N + kPre - 1) / kPre;
    // wide stores need 16-byte aligned output rows (a group starts at a multiple of 256 elements of each of them)
    const bool wide = ((reinterpret_cast<uintptr_t>(A.popt) | reinterpret_cast<uintptr_t>(A.r2) |
                        reinterpret_cast<uintptr_t>(A.tc) | reinterpret_cast<uintptr_t>(A.info) |
                        reinterpret_cast<uintptr_t>(A.nfev)) & 15) == 0;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    // one 4-byte mask load per lane covers a whole group; the next group's word is in flight while this one is classified
    // and filled (the loop is otherwise one dependent load latency per group)
    const bool mask_words = (reinterpret_cast<uintptr_t>(A.mask) & 3) == 0;
    auto group_word = [&](long long g) -> unsigned int {
        const long long start = g * kPre;
        if (g >= ngroups) return 0u;
        if (mask_words && A.N - start >= kPre) return reinterpret_cast<const unsigned int *>(A.mask + start)[lane];
        unsigned int w = 0;
        for (int k = 0; k < 4; ++k) {
            const long long j = start + 4 * lane + k;
            if (j < A.N) w |= A.mask[j] != 0 ? (1u << (8 * k)) : 0u;
        }
        return w;
    };
    // A wave walks a CONTIGUOUS range of groups and publishes the tiles it found with ONE atomic per 64 of them: the list
    // counter is a single device-scope word (~88 read-modify-writes per microsecond), and one atomic per non-empty tile made
    // the pre-pass of a thin ROI atomic-bound (5 600 tiles: 137 us where the mask + fill traffic needs 76).
    const long long per_wave = (ngroups + nwaves - 1) / nwaves;
    const long long g_lo = wave0 * per_wave, g_hi = g_lo + per_wave < ngroups ? g_lo + per_wave : ngroups;
    unsigned long long pend = 0;        // bit i: tile pend_tile0 + i holds a selected voxel
    long long pend_tile0 = g_lo * kPerGroup;
    auto publish = [&]() {
        if (pend) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(count, (unsigned int)__popcll(pend));
            base = __builtin_amdgcn_readfirstlane(base);
            if ((pend >> lane) & 1ull) list[base + __popcll(pend & ((1ull << lane) - 1ull))] = (unsigned int)(pend_tile0 + lane);
            pend = 0;
        }
    };
    unsigned int word = group_word(g_lo);
    for (long long g = g_lo; g < g_hi; ++g) {
        const unsigned int next = group_word(g + 1 < g_hi ? g + 1 : ngroups);
        const long long start = g * kPre;
        const long long rem = A.N - start;
        const int cnt = rem < kPre ? (int)rem : kPre;
        const unsigned long long sel = __ballot(word != 0);  // lane l holds voxels 4 l .. 4 l + 3 of the group
        if (sel) {
            if ((g * kPerGroup - pend_tile0) + kPerGroup > 64) {
                publish();
                pend_tile0 = g * kPerGroup;
            }
#pragma unroll
            for (int h = 0; h < kPerGroup; ++h) {
                const unsigned long long part = kPerGroup == 1 ? ~0ull : (((1ull << kLanesPerTile) - 1ull) << (h * kLanesPerTile));
                const long long tstart = start + (long long)h * kSub;
                if (tstart >= A.N) break;
                if (sel & part) {
                    pend |= 1ull << (int)(g * kPerGroup + h - pend_tile0);
                } else {  // this tile of the group holds no selected voxel: its fill (the edge of the region)
                    const long long trem = A.N - tstart;
                    const int tcnt = trem < kSub ? (int)trem : kSub;
                    for (int k = 0; k < kVpl; ++k) {
                        const int j = k * 64 + lane;
                        if (j < tcnt) finish_voxel(A, tstart + j, 0, 0, 0, -1, 0, true);
                    }
                }
            }
        } else if (cnt == kPre && wide) {
            fill_tile(A, start, lane);
        } else {
            for (int k = 0; k < kPre / 64; ++k) {
                const int j = k * 64 + lane;
                if (j < cnt) finish_voxel(A, start + j, 0, 0, 0, -1, 0, true);
            }
        }
        word = next;
    }
    publish();
}

hipError_t monoexp_mask_prepass(const FitKArgs &k, unsigned int *list, unsigned int *count, int num_cu,
                                hipStream_t stream) {
    const long long ngroups = (k.N + 255) / 256;
    long long blocks = (ngroups + 3) / 4;
    if (blocks > (long long)num_cu * 8) blocks = (long long)num_cu * 8;
    (void)hipGetLastError();
    hipLaunchKernelGGL(monoexp_mask_prepass_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, k, list, count);
    return hipGetLastError();
}

// ---- host-side dispatch ---------------------------------------------------------------------------
// waves per block: each wave owns E * kSub * sizeof(LT) bytes of LDS; keep two blocks per CU when the
// tile is small enough (<= 80 KB per block), otherwise one block of up to 160 KB.
template <typename LT>
static int waves_per_block(int E) {
    const size_t tile = lds_bytes_per_wave<LT>(E);
    int w = (int)((160 * 1024 / QMRI_SMALL_E_BLOCKS) / tile);  // (room for QMRI_SMALL_E_BLOCKS blocks per CU)
    if (w < 1) w = (int)((160 * 1024) / tile);
    if (w > 4) w = 4;
    return w;  // 0 -> does not fit (cannot happen for E <= QMRI_MAX_ECHOES)
}

template <int EMAX, bool FULL, typename LT>
static hipError_t launch_one(const FitKArgs &k, int grid, hipStream_t stream) {
    const int wpb = waves_per_block<LT>(k.E);
    if (wpb < 1) return hipErrorInvalidValue;
    const size_t lds = (size_t)wpb * lds_bytes_per_wave<LT>(k.E);
    auto fn = monoexp_lm_kernel<EMAX, FULL, LT>;
    if constexpr (EMAX <= 8)
        if (k.tile_list) fn = monoexp_lm_kernel<EMAX, FULL, LT, true>;
    (void)hipGetLastError();  // do not inherit a stale error from an unrelated earlier call
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * wpb), lds, stream, k);
    return hipGetLastError();
}

template <int EMAX, bool FULL, typename LT>
static int occupancy_one(int E) {
    int nb = 0;
    const int wpb = waves_per_block<LT>(E);
    const size_t lds = (size_t)wpb * lds_bytes_per_wave<LT>(E);
    auto fn = monoexp_lm_kernel<EMAX, FULL, LT>;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * wpb, lds) != hipSuccess) nb = 1;
    (void)hipGetLastError();
    return nb < 1 ? 1 : nb;
}

template <int EMAX, bool FULL, typename LT>
static int wpb_one(int E) {
    return waves_per_block<LT>(E);
}

#define QMRI_DISPATCH(EM, FN, ...)                                                     \
    (k.y_dtype == QMRI_F64                                                             \
         ? (k.E == EM ? FN<EM, true, double>(__VA_ARGS__) : FN<EM, false, double>(__VA_ARGS__)) \
         : (k.E == EM ? FN<EM, true, float>(__VA_ARGS__) : FN<EM, false, float>(__VA_ARGS__)))

#define QMRI_DISPATCH_FULL(EM, FN, ...) \
    (k.y_dtype == QMRI_F64 ? FN<EM, true, double>(__VA_ARGS__) : FN<EM, true, float>(__VA_ARGS__))

int monoexp_tile_voxels() { return kSub; }

const char *monoexp_variant_name(int E, int y_dtype) {
    const bool d = y_dtype == QMRI_F64;
    if (E <= 4) return d ? (E == 4 ? "monoexp_lm<4,full,f64>" : "monoexp_lm<4,part,f64>")
                         : (E == 4 ? "monoexp_lm<4,full,f32>" : "monoexp_lm<4,part,f32>");
    if (E == 5) return d ? "monoexp_lm<5,full,f64>" : "monoexp_lm<5,full,f32>";
    if (E == 6) return d ? "monoexp_lm<6,full,f64>" : "monoexp_lm<6,full,f32>";
    if (E == 7) return d ? "monoexp_lm<7,full,f64>" : "monoexp_lm<7,full,f32>";
    if (E <= 8) return d ? "monoexp_lm<8,full,f64>" : "monoexp_lm<8,full,f32>";
    if (E == 9) return d ? "monoexp_lm<9,full,f64>" : "monoexp_lm<9,full,f32>";
    if (E == 10) return d ? "monoexp_lm<10,full,f64>" : "monoexp_lm<10,full,f32>";
    if (E == 11) return d ? "monoexp_lm<11,full,f64>" : "monoexp_lm<11,full,f32>";
    if (E == 12) return d ? "monoexp_lm<12,full,f64>" : "monoexp_lm<12,full,f32>";
    if (E == 13) return d ? "monoexp_lm<13,full,f64>" : "monoexp_lm<13,full,f32>";
    if (E == 14) return d ? "monoexp_lm<14,full,f64>" : "monoexp_lm<14,full,f32>";
    if (E == 15) return d ? "monoexp_lm<15,full,f64>" : "monoexp_lm<15,full,f32>";
    if (E == 16) return d ? "monoexp_lm<16,full,f64>" : "monoexp_lm<16,full,f32>";
    return d ? (E == 32 ? "monoexp_lm<32,full,f64>" : "monoexp_lm<32,part,f64>")
             : (E == 32 ? "monoexp_lm<32,full,f32>" : "monoexp_lm<32,part,f32>");
}

int monoexp_waves_per_block(const FitKArgs &k) {
    if (k.E <= 4) return QMRI_DISPATCH(4, wpb_one, k.E);
    // 5..8 samples: an exact-size instantiation each (the partial EMAX = 8 variant needs 256 + 46 registers = one wave per SIMD)
    if (k.E == 5) return QMRI_DISPATCH_FULL(5, wpb_one, k.E);
    if (k.E == 6) return QMRI_DISPATCH_FULL(6, wpb_one, k.E);
    if (k.E == 7) return QMRI_DISPATCH_FULL(7, wpb_one, k.E);
    if (k.E <= 8) return QMRI_DISPATCH_FULL(8, wpb_one, k.E);
    if (k.E == 9) return QMRI_DISPATCH_FULL(9, wpb_one, k.E);
    if (k.E == 10) return QMRI_DISPATCH_FULL(10, wpb_one, k.E);
    if (k.E == 11) return QMRI_DISPATCH_FULL(11, wpb_one, k.E);
    if (k.E == 12) return QMRI_DISPATCH_FULL(12, wpb_one, k.E);
    if (k.E == 13) return QMRI_DISPATCH_FULL(13, wpb_one, k.E);
    if (k.E == 14) return QMRI_DISPATCH_FULL(14, wpb_one, k.E);
    if (k.E == 15) return QMRI_DISPATCH_FULL(15, wpb_one, k.E);
    if (k.E == 16) return QMRI_DISPATCH_FULL(16, wpb_one, k.E);
    return QMRI_DISPATCH(32, wpb_one, k.E);
}

int monoexp_blocks_per_cu(const FitKArgs &k) {
    if (k.E <= 4) return QMRI_DISPATCH(4, occupancy_one, k.E);
    // 5..8 samples: an exact-size instantiation each (the partial EMAX = 8 variant needs 256 + 46 registers = one wave per SIMD)
    if (k.E == 5) return QMRI_DISPATCH_FULL(5, occupancy_one, k.E);
    if (k.E == 6) return QMRI_DISPATCH_FULL(6, occupancy_one, k.E);
    if (k.E == 7) return QMRI_DISPATCH_FULL(7, occupancy_one, k.E);
    if (k.E <= 8) return QMRI_DISPATCH_FULL(8, occupancy_one, k.E);
    if (k.E == 9) return QMRI_DISPATCH_FULL(9, occupancy_one, k.E);
    if (k.E == 10) return QMRI_DISPATCH_FULL(10, occupancy_one, k.E);
    if (k.E == 11) return QMRI_DISPATCH_FULL(11, occupancy_one, k.E);
    if (k.E == 12) return QMRI_DISPATCH_FULL(12, occupancy_one, k.E);
    if (k.E == 13) return QMRI_DISPATCH_FULL(13, occupancy_one, k.E);
    if (k.E == 14) return QMRI_DISPATCH_FULL(14, occupancy_one, k.E);
    if (k.E == 15) return QMRI_DISPATCH_FULL(15, occupancy_one, k.E);
    if (k.E == 16) return QMRI_DISPATCH_FULL(16, occupancy_one, k.E);
    return QMRI_DISPATCH(32, occupancy_one, k.E);
}

hipError_t monoexp_launch(const FitKArgs &k, int grid, hipStream_t stream) {
    if (k.E <= 4) return QMRI_DISPATCH(4, launch_one, k, grid, stream);
    // 5..8 samples: an exact-size instantiation each (the partial EMAX = 8 variant needs 256 + 46 registers = one wave per SIMD)
    if (k.E == 5) return QMRI_DISPATCH_FULL(5, launch_one, k, grid, stream);
    if (k.E == 6) return QMRI_DISPATCH_FULL(6, launch_one, k, grid, stream);
    if (k.E == 7) return QMRI_DISPATCH_FULL(7, launch_one, k, grid, stream);
    if (k.E <= 8) return QMRI_DISPATCH_FULL(8, launch_one, k, grid, stream);
    if (k.E == 9) return QMRI_DISPATCH_FULL(9, launch_one, k, grid, stream);
    if (k.E == 10) return QMRI_DISPATCH_FULL(10, launch_one, k, grid, stream);
    if (k.E == 11) return QMRI_DISPATCH_FULL(11, launch_one, k, grid, stream);
    if (k.E == 12) return QMRI_DISPATCH_FULL(12, launch_one, k, grid, stream);
    if (k.E == 13) return QMRI_DISPATCH_FULL(13, launch_one, k, grid, stream);
    if (k.E == 14) return QMRI_DISPATCH_FULL(14, launch_one, k, grid, stream);
    if (k.E == 15) return QMRI_DISPATCH_FULL(15, launch_one, k, grid, stream);
    if (k.E == 16) return QMRI_DISPATCH_FULL(16, launch_one, k, grid, stream);
    return QMRI_DISPATCH(32, launch_one, k, grid, stream);
}

}  // namespace qmri

#ifdef QMRI_STATS
extern "C" int qmri_debug_fit_stats(unsigned long long *out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(qmri::g_fit_stats), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(qmri::g_fit_stats), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
