// lm_generic.hip -- general (m, n) MINPACK lmdif on the GPU, one voxel per lane.  gfx950 only.
//
// What it replaces: curve_fit(func, x, y, p0, ftol=1e-5, maxfev=100) of the reference for models other than
// the mono-exponential hot path -- /root/reference/dosma/core/fitting.py:755-870 (loop :855-868 over
// _curve_fit :1026-1073) with func = biexponential (:1021-1023), the only other model the reference ships
// (SURVEY.md 8(f) row N4).  It also runs the mono-exponential model with TRUE forward differences (n + 1
// model evaluations per iteration, exactly lmdif's fdjac2), which the tests use to cross-check the fast
// kernel's emulated differences (monoexp_lm.hip) on the GPU itself.
//
// Algorithm: scipy.optimize.leastsq -> MINPACK-1 lmdif / fdjac2 / qrfac / lmpar / qrsolv / enorm (More, Garbow,
// Hillstrom, ANL-80-74), mode = 1, restated for n <= 4 in registers; the m x n Jacobian, the residuals and the
// samples of a voxel live in a lane-private LDS column ([row][lane] layout: a wave-instruction touches 64
// consecutive doubles, conflict-free).  Lanes never synchronise; divergence is the SIMT mask.
// The model is evaluated with numpy's rounding (one rounding per operation, no FMA contraction).
#include "qmri_internal.h"
#include "fp64_fast.h"

#pragma clang fp contract(off)

namespace qmri {
namespace {

constexpr double kEpsmch = 2.220446049250313e-16;
constexpr double kDwarf = 2.2250738585072014e-308;

// MINPACK enorm over a strided vector (stride S doubles)
template <int S>
__device__ double enorm_s(const double *x, int n) {
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    for (int i = 0; i < n; ++i) {
        const double xabs = fabs(x[(size_t)i * S]);
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) {
                const double r = x3max / xabs;
                s3 = 1.0 + s3 * r * r;
                x3max = xabs;
            } else if (xabs != 0.0) {
                const double r = xabs / x3max;
                s3 += r * r;
            }
        } else {
            if (xabs > x1max) {
                const double r = x1max / xabs;
                s1 = 1.0 + s1 * r * r;
                x1max = xabs;
            } else {
                const double r = xabs / x1max;
                s1 += r * r;
            }
        }
    }
    if (s1 != 0.0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0) {
        if (s2 >= x3max) return sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

template <int NP>
__device__ double enorm_r(const double (&x)[NP]) {
    return enorm_s<1>(x, NP);
}

// register-array access with a run-time index (n <= 4).  Through a vector value: a chain of selects over a[j] is folded by
// LLVM into ONE load with a selected address, which pins every such array (and all its other accesses) in scratch memory;
// extractelement / insertelement with a run-time index stay in registers.
template <int NP, typename T>
__device__ __forceinline__ T get(const T (&a)[NP], int idx) {
    typedef T vec_t __attribute__((ext_vector_type(NP)));
    vec_t t;
#pragma unroll
    for (int j = 0; j < NP; ++j) t[j] = a[j];
    return t[idx];
}
template <int NP, typename T>
__device__ __forceinline__ void put(T (&a)[NP], int idx, T val) {
    typedef T vec_t __attribute__((ext_vector_type(NP)));
    vec_t t;
#pragma unroll
    for (int j = 0; j < NP; ++j) t[j] = a[j];
    t[idx] = val;
#pragma unroll
    for (int j = 0; j < NP; ++j) a[j] = t[j];
}

// residuals f(x; p) - y into a strided column  (scipy _wrap_func: func(xdata, *params) - ydata)
template <int MODEL, int NP>
__device__ __forceinline__ void residuals(const LmKArgs &a, const double (&p)[NP], const double *ys, double *out) {
    for (int i = 0; i < a.E; ++i) {
        const double x = a.x[i];
        double f;
        if (MODEL == QMRI_MODEL_MONOEXP) {
            f = p[0] * exp(p[1] * x);  // fitting.py:1016-1018
        } else {
            f = p[0] * exp(p[1] * x) + p[NP > 2 ? 2 : 0] * exp(p[NP > 3 ? 3 : 0] * x);  // fitting.py:1021-1023
        }
        out[(size_t)i * 64] = f - ys[(size_t)i * 64];
    }
}

// qrsolv on the register copy of R (r[j][i], i <= j = upper triangle; the rest is scratch)
template <int NP>
__device__ __forceinline__ void qrsolv(double (&r)[NP][NP], const int (&ipvt)[NP], const double (&diag)[NP], const double (&qtb)[NP],
                       double (&x)[NP], double (&sdiag)[NP]) {
    double wa[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
#pragma unroll
        for (int i = j; i < NP; ++i) r[j][i] = r[i][j];
        x[j] = r[j][j];
        wa[j] = qtb[j];
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const double dl = get<NP>(diag, ipvt[j]);
        if (dl != 0.0) {
#pragma unroll
            for (int k = j; k < NP; ++k) sdiag[k] = 0.0;
            sdiag[j] = dl;
            double qtbpj = 0.0;
#pragma unroll
            for (int k = j; k < NP; ++k) {
                if (sdiag[k] == 0.0) continue;
                double c, s;
                const double rkk = r[k][k];
                if (fabs(rkk) < fabs(sdiag[k])) {
                    const double cotan = rkk / sdiag[k];
                    s = 0.5 / sqrt(0.25 + 0.25 * cotan * cotan);
                    c = s * cotan;
                } else {
                    const double tn = sdiag[k] / rkk;
                    c = 0.5 / sqrt(0.25 + 0.25 * tn * tn);
                    s = c * tn;
                }
                r[k][k] = c * rkk + s * sdiag[k];
                const double temp = c * wa[k] + s * qtbpj;
                qtbpj = -s * wa[k] + c * qtbpj;
                wa[k] = temp;
#pragma unroll
                for (int i = k + 1; i < NP; ++i) {
                    const double t = c * r[k][i] + s * sdiag[i];
                    sdiag[i] = -s * r[k][i] + c * sdiag[i];
                    r[k][i] = t;
                }
            }
        }
        sdiag[j] = r[j][j];
        r[j][j] = x[j];
    }
    int nsing = NP;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (sdiag[j] == 0.0 && nsing == NP) nsing = j;
        if (nsing < NP) wa[j] = 0.0;
    }
#pragma unroll
    for (int j = NP - 1; j >= 0; --j) {
        if (j < nsing) {
            double sum = 0.0;
#pragma unroll
            for (int i = j + 1; i < NP; ++i)
                if (i < nsing) sum += r[j][i] * wa[i];
            wa[j] = (wa[j] - sum) / sdiag[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) put<NP>(x, ipvt[j], wa[j]);
}

template <int NP>
__device__ __forceinline__ void lmpar(double (&r)[NP][NP], const int (&ipvt)[NP], const double (&diag)[NP], const double (&qtb)[NP],
                      double delta, double &par, double (&x)[NP], double (&sdiag)[NP]) {
    double wa1[NP], wa2[NP];
    int nsing = NP;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        wa1[j] = qtb[j];
        if (r[j][j] == 0.0 && nsing == NP) nsing = j;
        if (nsing < NP) wa1[j] = 0.0;
    }
#pragma unroll
    for (int j = NP - 1; j >= 0; --j) {
        if (j < nsing) {
            wa1[j] /= r[j][j];
            const double temp = wa1[j];
#pragma unroll
            for (int i = 0; i < j; ++i) wa1[i] -= r[j][i] * temp;
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) put<NP>(x, ipvt[j], wa1[j]);

    int iter = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm_r<NP>(wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        par = 0.0;
        return;
    }
    double parl = 0.0;
    if (nsing >= NP) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int l = ipvt[j];
            wa1[j] = get<NP>(diag, l) * (get<NP>(wa2, l) / dxnorm);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            double sum = 0.0;
#pragma unroll
            for (int i = 0; i < j; ++i) sum += r[j][i] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j][j];
        }
        const double temp = enorm_r<NP>(wa1);
        parl = ((fp / delta) / temp) / temp;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i <= j; ++i) sum += r[j][i] * qtb[i];
        wa1[j] = sum / get<NP>(diag, ipvt[j]);
    }
    const double gnorm = enorm_r<NP>(wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = kDwarf / (delta < 0.1 ? delta : 0.1);
    if (par < parl) par = parl;
    if (par > paru) par = paru;
    if (par == 0.0) par = gnorm / dxnorm;

    for (;;) {
        ++iter;
        if (par == 0.0) par = kDwarf > 0.001 * paru ? kDwarf : 0.001 * paru;
        double temp = sqrt(par);
#pragma unroll
        for (int j = 0; j < NP; ++j) wa1[j] = temp * diag[j];
        qrsolv<NP>(r, ipvt, wa1, qtb, x, sdiag);
#pragma unroll
        for (int j = 0; j < NP; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm_r<NP>(wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int l = ipvt[j];
            wa1[j] = get<NP>(diag, l) * (get<NP>(wa2, l) / dxnorm);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            wa1[j] /= sdiag[j];
            const double t = wa1[j];
#pragma unroll
            for (int i = j + 1; i < NP; ++i) wa1[i] -= r[j][i] * t;
        }
        temp = enorm_r<NP>(wa1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0 && par > parl) parl = par;
        if (fp < 0.0 && par < paru) paru = par;
        par = parl > par + parc ? parl : par + parc;
    }
}

template <typename T>
__device__ __forceinline__ double load_sample(const void *y, size_t idx) {
    return (double)static_cast<const T *>(y)[idx];
}
__device__ __forceinline__ double load_any(const void *y, int dtype, size_t idx) {
    switch (dtype) {
        case QMRI_F32: return load_sample<float>(y, idx);
        case QMRI_F64: return load_sample<double>(y, idx);
        case QMRI_I16: return load_sample<short>(y, idx);
        default: return load_sample<unsigned short>(y, idx);
    }
}

template <int MODEL, int NP>
__global__ __launch_bounds__(64) void lm_generic_kernel(const LmKArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int m = a.E;
    double *FJ = lds + lane;                     // FJ[(j*m + i)*64]
    double *FV = FJ + (size_t)NP * m * 64;       // residuals at x
    double *W4 = FV + (size_t)m * 64;            // residuals at the trial point / work column
    double *YS = W4 + (size_t)m * 64;            // samples
#define FJA(j, i) FJ[((size_t)(j) * m + (i)) * 64]

    for (long long v = (long long)blockIdx.x * 64 + lane; v < a.N; v += (long long)gridDim.x * 64) {
        bool allzero = true, finite = true, inb = true;
        for (int i = 0; i < m; ++i) {
            const double s = load_any(a.y, a.y_dtype, (size_t)i * a.ld + v);
            YS[(size_t)i * 64] = s;
            allzero &= s == 0.0;
            finite &= (s - s) == 0.0;
            if (a.use_y_bounds) inb &= !(s < a.y_lo) && !(s > a.y_hi);
        }
        double x[NP];
        int info = 0, nfev = 0;
        double r2 = 0.0;
        if (!finite) atomicOr(a.nonfinite, 1);
        if (!allzero && finite && inb) {  // fitting.py:1064-1067 skip rule otherwise
#pragma unroll
            for (int j = 0; j < NP; ++j) x[j] = a.p0v[j] ? a.p0v[j][v] : a.p0[j];
            double diag[NP], qtf[NP], wa1[NP], wa2[NP], wa3[NP], sdiag[NP];
            double r[NP][NP];
            int ipvt[NP];
            int iter = 1;
            double par = 0.0, delta = 0.0, xnorm = 0.0, gnorm = 0.0;
            residuals<MODEL, NP>(a, x, YS, FV);
            nfev = 1;
            double fnorm = enorm_s<64>(FV, m);
            const double eps = sqrt(a.epsfcn > kEpsmch ? a.epsfcn : kEpsmch);
            for (;;) {  // outer loop
                // fdjac2
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double temp = x[j];
                    double h = eps * fabs(temp);
                    if (h == 0.0) h = eps;
                    x[j] = temp + h;
                    residuals<MODEL, NP>(a, x, YS, W4);
                    x[j] = temp;
                    for (int i = 0; i < m; ++i) FJA(j, i) = (W4[(size_t)i * 64] - FV[(size_t)i * 64]) / h;
                }
                nfev += NP;
                // qrfac (pivoting), wa1 = rdiag, wa2 = acnorm, wa3 = work
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    wa2[j] = enorm_s<64>(&FJA(j, 0), m);
                    wa1[j] = wa2[j];
                    wa3[j] = wa1[j];
                    ipvt[j] = j;
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    int kmax = j;
#pragma unroll
                    for (int k = j; k < NP; ++k)
                        if (wa1[k] > get<NP>(wa1, kmax)) kmax = k;
                    if (kmax != j) {
                        for (int i = 0; i < m; ++i) {
                            const double t = FJA(j, i);
                            FJA(j, i) = FJA(kmax, i);
                            FJA(kmax, i) = t;
                        }
                        put<NP>(wa1, kmax, wa1[j]);
                        put<NP>(wa3, kmax, wa3[j]);
                        const int t = ipvt[j];
                        ipvt[j] = get<NP>(ipvt, kmax);
                        put<NP>(ipvt, kmax, t);
                    }
                    double ajnorm = enorm_s<64>(&FJA(j, j), m - j);
                    if (ajnorm != 0.0) {
                        if (FJA(j, j) < 0.0) ajnorm = -ajnorm;
                        for (int i = j; i < m; ++i) FJA(j, i) /= ajnorm;
                        FJA(j, j) += 1.0;
#pragma unroll
                        for (int k = j + 1; k < NP; ++k) {
                            double sum = 0.0;
                            for (int i = j; i < m; ++i) sum += FJA(j, i) * FJA(k, i);
                            const double temp = sum / FJA(j, j);
                            for (int i = j; i < m; ++i) FJA(k, i) -= temp * FJA(j, i);
                            if (wa1[k] != 0.0) {
                                double t = FJA(k, j) / wa1[k];
                                double d = 1.0 - t * t;
                                if (d < 0.0) d = 0.0;
                                wa1[k] *= sqrt(d);
                                t = wa1[k] / wa3[k];
                                if (0.05 * t * t <= kEpsmch) {
                                    wa1[k] = enorm_s<64>(&FJA(k, j + 1), m - j - 1);
                                    wa3[k] = wa1[k];
                                }
                            }
                        }
                    }
                    wa1[j] = -ajnorm;
                }
                if (iter == 1) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        diag[j] = wa2[j];
                        if (wa2[j] == 0.0) diag[j] = 1.0;
                    }
#pragma unroll
                    for (int j = 0; j < NP; ++j) wa3[j] = diag[j] * x[j];
                    xnorm = enorm_r<NP>(wa3);
                    delta = a.factor * xnorm;
                    if (delta == 0.0) delta = a.factor;
                }
                // qtf = first n components of Q^T fvec; R's diagonal into fjac
                for (int i = 0; i < m; ++i) W4[(size_t)i * 64] = FV[(size_t)i * 64];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    if (FJA(j, j) != 0.0) {
                        double sum = 0.0;
                        for (int i = j; i < m; ++i) sum += FJA(j, i) * W4[(size_t)i * 64];
                        const double temp = -sum / FJA(j, j);
                        for (int i = j; i < m; ++i) W4[(size_t)i * 64] += FJA(j, i) * temp;
                    }
                    FJA(j, j) = wa1[j];
                    qtf[j] = W4[(size_t)j * 64];
                }
                // R (upper triangle) into registers: r[j][i] = fjac[i, j], i <= j
#pragma unroll
                for (int j = 0; j < NP; ++j)
#pragma unroll
                    for (int i = 0; i < NP; ++i) r[j][i] = i <= j ? FJA(j, i) : 0.0;
                gnorm = 0.0;
                if (fnorm != 0.0) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const double acn = get<NP>(wa2, ipvt[j]);
                        if (acn == 0.0) continue;
                        double sum = 0.0;
#pragma unroll
                        for (int i = 0; i <= j; ++i) sum += r[j][i] * (qtf[i] / fnorm);
                        const double g = fabs(sum / acn);
                        if (g > gnorm) gnorm = g;
                    }
                }
                if (gnorm <= a.gtol) info = 4;
                if (info != 0) break;
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if (wa2[j] > diag[j]) diag[j] = wa2[j];

                double ratio;
                do {  // inner loop
                    lmpar<NP>(r, ipvt, diag, qtf, delta, par, wa1, sdiag);
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        wa1[j] = -wa1[j];
                        wa2[j] = x[j] + wa1[j];
                        wa3[j] = diag[j] * wa1[j];
                    }
                    const double pnorm = enorm_r<NP>(wa3);
                    if (iter == 1 && pnorm < delta) delta = pnorm;
                    residuals<MODEL, NP>(a, wa2, YS, W4);
                    ++nfev;
                    const double fnorm1 = enorm_s<64>(W4, m);
                    double actred = -1.0;
                    if (0.1 * fnorm1 < fnorm) {
                        const double t = fnorm1 / fnorm;
                        actred = 1.0 - t * t;
                    }
#pragma unroll
                    for (int j = 0; j < NP; ++j) wa3[j] = 0.0;
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const double temp = get<NP>(wa1, ipvt[j]);
#pragma unroll
                        for (int i = 0; i <= j; ++i) wa3[i] += r[j][i] * temp;
                    }
                    const double temp1 = enorm_r<NP>(wa3) / fnorm;
                    const double temp2 = (sqrt(par) * pnorm) / fnorm;
                    const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
                    const double dirder = -(temp1 * temp1 + temp2 * temp2);
                    ratio = 0.0;
                    if (prered != 0.0) ratio = actred / prered;
                    if (ratio <= 0.25) {
                        double temp = 0.5;
                        if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                        delta = temp * (delta < pnorm / 0.1 ? delta : pnorm / 0.1);
                        par /= temp;
                    } else if (par == 0.0 || ratio >= 0.75) {
                        delta = pnorm / 0.5;
                        par *= 0.5;
                    }
                    if (ratio >= 1e-4) {
#pragma unroll
                        for (int j = 0; j < NP; ++j) {
                            x[j] = wa2[j];
                            wa2[j] = diag[j] * x[j];
                        }
                        for (int i = 0; i < m; ++i) FV[(size_t)i * 64] = W4[(size_t)i * 64];
                        xnorm = enorm_r<NP>(wa2);
                        fnorm = fnorm1;
                        ++iter;
                    }
                    const bool small = fabs(actred) <= a.ftol && prered <= a.ftol && 0.5 * ratio <= 1.0;
                    if (small) info = 1;
                    if (delta <= a.xtol * xnorm) info = 2;
                    if (small && info == 2) info = 3;
                    if (info != 0) break;
                    if (nfev >= a.maxfev) info = 5;
                    if (fabs(actred) <= kEpsmch && prered <= kEpsmch && 0.5 * ratio <= 1.0) info = 6;
                    if (delta <= kEpsmch * xnorm) info = 7;
                    if (gnorm <= kEpsmch) info = 8;
                    if (info != 0) break;
                } while (ratio < 1e-4);
                if (info != 0) break;
            }
            if (info >= 1 && info <= 4) {  // fitting.py:1032-1035
                residuals<MODEL, NP>(a, x, YS, W4);
                double ss_res = 0.0, mean = 0.0, ss_tot = 0.0;
                for (int i = 0; i < m; ++i) ss_res += W4[(size_t)i * 64] * W4[(size_t)i * 64];
                for (int i = 0; i < m; ++i) mean += YS[(size_t)i * 64];
                mean /= (double)m;
                for (int i = 0; i < m; ++i) {
                    const double d = YS[(size_t)i * 64] - mean;
                    ss_tot += d * d;
                }
                r2 = 1.0 - ss_res / (ss_tot + a.r2_eps);
            }
        }
        const bool ok = info >= 1 && info <= 4;
        const double qnan = __builtin_nan("");
#pragma unroll
        for (int j = 0; j < NP; ++j) a.popt[(size_t)v * NP + j] = ok ? x[j] : qnan;
        a.r2[v] = ok ? r2 : 0.0;
        if (a.info) a.info[v] = (signed char)info;
        if (a.nfev) a.nfev[v] = (short)nfev;
    }
#undef FJA
}


// ---------------------------------------------------------------------------------------------------------
// lm_pull_kernel: the same lmdif, restructured for throughput (E <= 12; lm_generic_kernel stays the general-E route).
//   * ONE loop per wave, one LM round per trip; a lane whose voxel has finished pulls the next one from a device counter
//     as soon as kRefill lanes are idle (a voxel of the 4-parameter problem takes anything from 5 to 100 evaluations:
//     without the pull a wave runs at the pace of its slowest voxel).
//   * The echo count is a template parameter: residuals, exponentials and trial exponentials live in REGISTERS and every
//     loop over the echoes is unrolled (12 independent exp chains in flight instead of one); only the m x n Jacobian
//     and the samples are lane-private LDS columns.
//   * fdjac2 with the exponentials it already has: perturbing an amplitude re-uses exp(b x) of the current point
//     bit for bit (the model is a * exp(b x) [+ c * exp(d x)], one rounding per operation), so a round evaluates
//     (rates + 1) x m exponentials instead of (n + 1) x (terms) x m.  Every value equals what residuals() returns.
constexpr int kRefill = 16;  // measured 8 / 16 / 32: 505 / 542 / 535 Mvox/s (bi-exponential, 12 echoes)

struct EnormAcc {  // MINPACK enorm, one element at a time (same operations in the same order as enorm_s)
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0, agiant;
    __device__ explicit EnormAcc(int n) : agiant(1.304e19 / (double)n) {}
    __device__ __forceinline__ void add(double v) {
        const double rdwarf = 3.834e-20;
        const double xabs = fabs(v);
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) {
                const double r = x3max / xabs;
                s3 = 1.0 + s3 * r * r;
                x3max = xabs;
            } else if (xabs != 0.0) {
                const double r = xabs / x3max;
                s3 += r * r;
            }
        } else {
            if (xabs > x1max) {
                const double r = x1max / xabs;
                s1 = 1.0 + s1 * r * r;
                x1max = xabs;
            } else {
                const double r = xabs / x1max;
                s1 += r * r;
            }
        }
    }
    __device__ __forceinline__ double result() const {
        if (s1 != 0.0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
        if (s2 != 0.0) {
            if (s2 >= x3max) return sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
            return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
        }
        return x3max * sqrt(s3);
    }
};


// column[i0 .. M) /= d with ONE reciprocal: q = RN(a r), q + r (a - d q) (Markstein) -- correctly rounded whenever
// r = RN(1 / d), else within an ulp; outside a safe magnitude range (or with `big` numerators) the IEEE division
template <int M, bool FULL>
__device__ __forceinline__ void col_div(double *col /* stride 64 */, int i0, int m, double d, bool safe_num) {
    const double ad = fabs(d);
    if (safe_num && ad > 1e-140 && ad < 1e140) {
        const double r = rcp_nr(d);
#pragma unroll
        for (int i = 0; i < M; ++i)
            if (i >= i0 && (FULL || i < m)) {
                const double av = col[i * 64];
                const double q = av * r;
                col[i * 64] = fma(fma(-d, q, av), r, q);
            }
        return;
    }
    QMRI_COLD_PATH();
    for (int i = i0; i < m; ++i) col[i * 64] /= d;
}

// enorm of column[i0 .. M): when every |x| is in MINPACK's mid range the three-accumulator form reduces to sqrt(sum x^2)
// with the same operations; anything else (zeros included) takes the general routine
template <int M, bool FULL>
__device__ __forceinline__ double col_enorm(const double *col /* stride 64 */, int i0, int m) {
    const double rdwarf = 3.834e-20, agiant = 1.304e19 / (double)(m - i0);
    double s2 = 0.0;
    bool mid = true;
#pragma unroll
    for (int i = 0; i < M; ++i)
        if (i >= i0 && (FULL || i < m)) {
            const double xabs = fabs(col[i * 64]);
            mid &= xabs > rdwarf && xabs < agiant;
            s2 += xabs * xabs;
        }
    if (mid) return sqrt(s2);
    QMRI_COLD_PATH();
    return enorm_s<64>(col + (size_t)i0 * 64, m - i0);
}

template <int MODEL, int NP, int M, bool FULL>
__global__ __launch_bounds__(64) void lm_pull_kernel(const LmKArgs a, unsigned long long *counter, int refill) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int m = FULL ? M : a.E;
    double *FJ = lds + lane;                  // FJ[(j * M + i) * 64]
    double *YS = FJ + (size_t)NP * M * 64;    // samples
#define FJA(j, i) FJ[((j) * M + (i)) * 64]
#define YSA(i) YS[(i) * 64]
#define LIVE(i) (FULL || (i) < m)
    constexpr bool BI = MODEL == QMRI_MODEL_BIEXP;
    constexpr int P2 = NP > 2 ? 2 : 0, P3 = NP > 3 ? 3 : 0;

    // ---- per-voxel state ----
    double fv[M], e1[M], e2[M], t1[M], t2[M];
    double x[NP], diag[NP], qtf[NP], wa1[NP], wa2[NP], wa3[NP], sdiag[NP];
    double r[NP][NP];
    int ipvt[NP];
    int iter = 1, nfev = 0, info = 0;
    double par = 0.0, delta = 0.0, xnorm = 0.0, gnorm = 0.0, fnorm = 0.0;
    bool need_jac = true, active = false;
    long long v = 0;
    bool exhausted = false;  // wave-uniform: the counter has passed N
    const double eps = sqrt(a.epsfcn > kEpsmch ? a.epsfcn : kEpsmch);
    const double qnan = __builtin_nan("");

    // Arguments only the pull / store code needs (pointers, sample layout, initial guess) are re-read from the kernel-argument
    // segment where they are used, through a pointer the compiler cannot see through -- held in scalar registers across the
    // LM round they were 100-200 spills into VGPR lanes (the same measure as in monoexp_lm.hip: 192 -> 52 spills, -4 %).
    auto cold_args = [&]() -> const LmKArgs & {
        typedef const __attribute__((address_space(4))) LmKArgs *KArgP;
        KArgP p = (KArgP)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return *(const LmKArgs *)p;
    };
    auto store = [&](bool ok, double r2) {
        const LmKArgs &c = cold_args();
#pragma unroll
        for (int j = 0; j < NP; ++j) c.popt[(size_t)v * NP + j] = ok ? x[j] : qnan;
        c.r2[v] = ok ? r2 : 0.0;
        if (c.info) c.info[v] = (signed char)info;
        if (c.nfev) c.nfev[v] = (short)nfev;
    };

    for (;;) {
        // ---------------- pull ----------------
        const unsigned long long idle = __ballot(!active);
        const int nidle = __popcll(idle);
        if (exhausted && nidle == 64) break;
        if (!exhausted && (nidle == 64 || nidle >= refill)) {
            const LmKArgs &c = cold_args();
            const int leader = __ffsll((long long)idle) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(counter, (unsigned long long)nidle);
            base = __shfl(base, leader);
            if (base + (unsigned long long)nidle >= (unsigned long long)c.N) exhausted = true;
            if (!active) {
                v = (long long)base + __popcll(idle & ((1ull << lane) - 1ull));
                if (v < c.N) {
                    bool allzero = true, finite = true, inb = true;
#pragma unroll
                    for (int i = 0; i < M; ++i)
                        if (LIVE(i)) {
                            const double s = load_any(c.y, c.y_dtype, (size_t)i * c.ld + v);
                            YSA(i) = s;
                            allzero &= s == 0.0;
                            finite &= (s - s) == 0.0;
                            if (c.use_y_bounds) inb &= !(s < c.y_lo) && !(s > c.y_hi);
                        }
                    info = 0;
                    nfev = 0;
                    if (!finite) atomicOr(c.nonfinite, 1);
                    if (!allzero && finite && inb) {  // fitting.py:1064-1067 skip rule otherwise
#pragma unroll
                        for (int j = 0; j < NP; ++j) x[j] = c.p0v[j] ? c.p0v[j][v] : c.p0[j];
                        EnormAcc en(m);
#pragma unroll
                        for (int i = 0; i < M; ++i)
                            if (LIVE(i)) {
                                const double xi = a.x[i];
                                e1[i] = exp(x[1] * xi);
                                double f = x[0] * e1[i];
                                if (BI) {
                                    e2[i] = exp(x[P3] * xi);
                                    f = f + x[P2] * e2[i];
                                }
                                fv[i] = f - YSA(i);
                                en.add(fv[i]);
                            }
                        fnorm = en.result();
                        nfev = 1;
                        iter = 1;
                        par = 0.0;
                        need_jac = true;
                        active = true;
                    } else {
                        store(false, 0.0);
                    }
                }
            }
        }
        if (!active) continue;

        // ---------------- one LM round ----------------
        if (need_jac) {
            // fdjac2: column j = (f(x + h e_j) - f(x)) / h; an amplitude column needs no new exponential
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const double temp = x[j];
                double h = eps * fabs(temp);
                if (h == 0.0) h = eps;
                const double xp = temp + h;
                bool small_num = true;
#pragma unroll
                for (int i = 0; i < M; ++i)
                    if (LIVE(i)) {
                        const double xi = a.x[i];
                        double f;
                        if (j == 0) {
                            f = xp * e1[i];
                            if (BI) f = f + x[P2] * e2[i];
                        } else if (j == 1) {
                            f = x[0] * exp(xp * xi);
                            if (BI) f = f + x[P2] * e2[i];
                        } else if (j == 2) {
                            f = x[0] * e1[i] + xp * e2[i];
                        } else {
                            f = x[0] * e1[i] + x[P2] * exp(xp * xi);
                        }
                        const double num = (f - YSA(i)) - fv[i];
                        small_num &= fabs(num) < 1e140;  // (false for inf / NaN too)
                        FJA(j, i) = num;
                    }
                col_div<M, FULL>(&FJA(j, 0), 0, m, h, small_num);
            }
            nfev += NP;
            // qrfac (pivoting), wa1 = rdiag, wa2 = acnorm, wa3 = work
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                wa2[j] = col_enorm<M, FULL>(&FJA(j, 0), 0, m);
                wa1[j] = wa2[j];
                wa3[j] = wa1[j];
                ipvt[j] = j;
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                int kmax = j;
#pragma unroll
                for (int k = j; k < NP; ++k)
                    if (wa1[k] > get<NP>(wa1, kmax)) kmax = k;
                if (kmax != j) {
#pragma unroll
                    for (int i = 0; i < M; ++i)
                        if (LIVE(i)) {
                            const double t = FJA(j, i);
                            FJA(j, i) = FJ[((size_t)kmax * M + i) * 64];
                            FJ[((size_t)kmax * M + i) * 64] = t;
                        }
                    put<NP>(wa1, kmax, wa1[j]);
                    put<NP>(wa3, kmax, wa3[j]);
                    const int t = ipvt[j];
                    ipvt[j] = get<NP>(ipvt, kmax);
                    put<NP>(ipvt, kmax, t);
                }
                double ajnorm = col_enorm<M, FULL>(&FJA(j, 0), j, m);
                if (ajnorm != 0.0) {
                    if (FJA(j, j) < 0.0) ajnorm = -ajnorm;
                    col_div<M, FULL>(&FJA(j, 0), j, m, ajnorm, true);
                    FJA(j, j) += 1.0;
#pragma unroll
                    for (int k = j + 1; k < NP; ++k) {
                        double sum = 0.0;
#pragma unroll
                        for (int i = j; i < M; ++i)
                            if (LIVE(i)) sum += FJA(j, i) * FJA(k, i);
                        const double temp = sum / FJA(j, j);
#pragma unroll
                        for (int i = j; i < M; ++i)
                            if (LIVE(i)) FJA(k, i) -= temp * FJA(j, i);
                        if (wa1[k] != 0.0) {
                            double t = FJA(k, j) / wa1[k];
                            double d = 1.0 - t * t;
                            if (d < 0.0) d = 0.0;
                            wa1[k] *= sqrt(d);
                            t = wa1[k] / wa3[k];
                            if (0.05 * t * t <= kEpsmch) {
                                wa1[k] = col_enorm<M, FULL>(&FJA(k, 0), j + 1, m);
                                wa3[k] = wa1[k];
                            }
                        }
                    }
                }
                wa1[j] = -ajnorm;
            }
            if (iter == 1) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    diag[j] = wa2[j];
                    if (wa2[j] == 0.0) diag[j] = 1.0;
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) wa3[j] = diag[j] * x[j];
                xnorm = enorm_r<NP>(wa3);
                delta = a.factor * xnorm;
                if (delta == 0.0) delta = a.factor;
            }
            // qtf = first n components of Q^T fvec (work copy in registers); R's diagonal into fjac
            double w[M];
#pragma unroll
            for (int i = 0; i < M; ++i) w[i] = fv[i];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (FJA(j, j) != 0.0) {
                    double sum = 0.0;
#pragma unroll
                    for (int i = j; i < M; ++i)
                        if (LIVE(i)) sum += FJA(j, i) * w[i];
                    const double temp = -sum / FJA(j, j);
#pragma unroll
                    for (int i = j; i < M; ++i)
                        if (LIVE(i)) w[i] += FJA(j, i) * temp;
                }
                FJA(j, j) = wa1[j];
                qtf[j] = w[j];
            }
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int i = 0; i < NP; ++i) r[j][i] = i <= j ? FJA(j, i) : 0.0;
            gnorm = 0.0;
            if (fnorm != 0.0) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double acn = get<NP>(wa2, ipvt[j]);
                    if (acn == 0.0) continue;
                    double sum = 0.0;
#pragma unroll
                    for (int i = 0; i <= j; ++i) sum += r[j][i] * (qtf[i] / fnorm);
                    const double g = fabs(sum / acn);
                    if (g > gnorm) gnorm = g;
                }
            }
            if (gnorm <= a.gtol) info = 4;
            if (info == 0) {
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if (wa2[j] > diag[j]) diag[j] = wa2[j];
            }
        }
        if (info == 0) {
            // one pass of lmdif's inner loop
            lmpar<NP>(r, ipvt, diag, qtf, delta, par, wa1, sdiag);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = enorm_r<NP>(wa3);
            if (iter == 1 && pnorm < delta) delta = pnorm;
            double s2 = 0.0;
            bool mid = true;
            const double agiant = 1.304e19 / (double)m;
#pragma unroll
            for (int i = 0; i < M; ++i)
                if (LIVE(i)) {
                    const double xi = a.x[i];
                    t1[i] = exp(wa2[1] * xi);
                    double f = wa2[0] * t1[i];
                    if (BI) {
                        t2[i] = exp(wa2[P3] * xi);
                        f = f + wa2[P2] * t2[i];
                    }
                    const double xabs = fabs(f - YSA(i));
                    mid &= xabs > 3.834e-20 && xabs < agiant;
                    s2 += xabs * xabs;
                }
            ++nfev;
            double fnorm1 = sqrt(s2);  // = enorm when every residual is in its mid range
            if (!mid) {
                QMRI_COLD_PATH();
                EnormAcc en(m);
                for (int i = 0; i < M; ++i)
                    if (LIVE(i)) {
                        double f = wa2[0] * t1[i];
                        if (BI) f = f + wa2[P2] * t2[i];
                        en.add(f - YSA(i));
                    }
                fnorm1 = en.result();
            }
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) {
                const double t = fnorm1 / fnorm;
                actred = 1.0 - t * t;
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) wa3[j] = 0.0;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const double temp = get<NP>(wa1, ipvt[j]);
#pragma unroll
                for (int i = 0; i <= j; ++i) wa3[i] += r[j][i] * temp;
            }
            const double temp1 = enorm_r<NP>(wa3) / fnorm;
            const double temp2 = (sqrt(par) * pnorm) / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            double ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * (delta < pnorm / 0.1 ? delta : pnorm / 0.1);
                par /= temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par *= 0.5;
            }
            need_jac = ratio >= 1e-4;
            if (need_jac) {  // accept: the trial point's exponentials and residuals become the current ones
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                }
#pragma unroll
                for (int i = 0; i < M; ++i)
                    if (LIVE(i)) {
                        e1[i] = t1[i];
                        double f = x[0] * t1[i];
                        if (BI) {
                            e2[i] = t2[i];
                            f = f + x[P2] * t2[i];
                        }
                        fv[i] = f - YSA(i);
                    }
                xnorm = enorm_r<NP>(wa2);
                fnorm = fnorm1;
                ++iter;
            }
            const bool small = fabs(actred) <= a.ftol && prered <= a.ftol && 0.5 * ratio <= 1.0;
            if (small) info = 1;
            if (delta <= a.xtol * xnorm) info = 2;
            if (small && info == 2) info = 3;
            if (info == 0) {
                if (nfev >= a.maxfev) info = 5;
                if (fabs(actred) <= kEpsmch && prered <= kEpsmch && 0.5 * ratio <= 1.0) info = 6;
                if (delta <= kEpsmch * xnorm) info = 7;
                if (gnorm <= kEpsmch) info = 8;
            }
        }
        if (info != 0) {
            const bool ok = info >= 1 && info <= 4;
            double r2 = 0.0;
            if (ok) {  // fitting.py:1032-1035 (fv IS the residual vector at the accepted x)
                double ss_res = 0.0, mean = 0.0, ss_tot = 0.0;
#pragma unroll
                for (int i = 0; i < M; ++i)
                    if (LIVE(i)) ss_res += fv[i] * fv[i];
#pragma unroll
                for (int i = 0; i < M; ++i)
                    if (LIVE(i)) mean += YSA(i);
                mean /= (double)m;
#pragma unroll
                for (int i = 0; i < M; ++i)
                    if (LIVE(i)) {
                        const double d = YSA(i) - mean;
                        ss_tot += d * d;
                    }
                r2 = 1.0 - ss_res / (ss_tot + cold_args().r2_eps);
            }
            store(ok, r2);
            active = false;
        }
    }
#undef FJA
#undef YSA
#undef LIVE
}

template <int MODEL, int NP, int M, bool FULL>
static hipError_t pull_launch_t(const LmKArgs &k, unsigned long long *counter, int num_cu, hipStream_t stream) {
    auto fn = lm_pull_kernel<MODEL, NP, M, FULL>;
    const size_t lds = (size_t)(NP + 1) * M * 64 * sizeof(double);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    long long blocks = (k.N + 63) / 64;
    const long long cap = (long long)num_cu * (long long)(160 * 1024 / lds < 8 ? 160 * 1024 / lds : 8);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(64), lds, stream, k, counter, kRefill);
    return hipGetLastError();
}
template <int MODEL, int NP>
static hipError_t pull_launch_m(const LmKArgs &k, unsigned long long *counter, int num_cu, hipStream_t stream) {
    if (k.E == 8) return pull_launch_t<MODEL, NP, 8, true>(k, counter, num_cu, stream);
    if (k.E < 8) return pull_launch_t<MODEL, NP, 8, false>(k, counter, num_cu, stream);
    if (k.E == 12) return pull_launch_t<MODEL, NP, 12, true>(k, counter, num_cu, stream);
    return pull_launch_t<MODEL, NP, 12, false>(k, counter, num_cu, stream);
}

}  // namespace

int lm_generic_nparams(int model) { return model == QMRI_MODEL_MONOEXP ? 2 : model == QMRI_MODEL_BIEXP ? 4 : 0; }

hipError_t lm_generic_launch(const LmKArgs &k, int model, int num_cu, unsigned long long *counter, hipStream_t stream) {
    const int np = lm_generic_nparams(model);
    if (counter && k.E <= 12) {  // the pull kernel (counter: one zeroed 8-byte word per launch)
        (void)hipGetLastError();
        return model == QMRI_MODEL_MONOEXP ? pull_launch_m<QMRI_MODEL_MONOEXP, 2>(k, counter, num_cu, stream)
                                           : pull_launch_m<QMRI_MODEL_BIEXP, 4>(k, counter, num_cu, stream);
    }
    const size_t lds = (size_t)(np + 3) * k.E * 64 * sizeof(double);
    long long blocks = (k.N + 63) / 64;
    const long long cap = (long long)num_cu * 8;
    if (blocks > cap) blocks = cap;
    (void)hipGetLastError();
    hipError_t e = hipSuccess;
    if (model == QMRI_MODEL_MONOEXP) {
        auto fn = lm_generic_kernel<QMRI_MODEL_MONOEXP, 2>;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(64), lds, stream, k);
    } else {
        auto fn = lm_generic_kernel<QMRI_MODEL_BIEXP, 4>;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(64), lds, stream, k);
    }
    return hipGetLastError();
}

}  // namespace qmri
