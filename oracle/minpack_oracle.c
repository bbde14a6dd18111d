/*
 * oracle/minpack_oracle.c -- CPU restatement of the arithmetic on DOSMA's per-voxel fit path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker*: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  Nothing under dosma_amd/ links, loads or calls it.
 *
 * What it restates
 * ----------------
 * The reference computes one nonlinear least-squares fit per voxel
 *     dosma/core/fitting.py:1026-1073  _curve_fit()
 *       -> :1030  scipy.optimize.curve_fit(func, x, y, p0=p0, ftol=1e-5, maxfev=100)
 *       -> :1032-1035  r2 = 1 - ss_res / (ss_tot + eps)
 *       -> :1065-1067  skip rule (all samples == 0  -> popt = NaN, r2 = 0)
 *       -> :1069-1072  RuntimeError (MINPACK info not in 1..4) -> popt = NaN, r2 = 0
 * The arithmetic itself is NOT in /root/reference: it is the third-party dependency
 * scipy (unpinned: requirements.txt:12, setup.py:108; 1.15.3 installed in this image), whose
 * curve_fit(method="lm") calls leastsq() -> MINPACK `lmdif` with the settings
 *     ftol = 1e-5 (DOSMA), xtol = 1.49012e-8, gtol = 0, maxfev = 100 (DOSMA),
 *     epsfcn = DBL_EPSILON, factor = 100, mode = 1 (diag = None)
 * (scipy/optimize/_minpack_py.py: leastsq defaults; curve_fit raises RuntimeError unless
 * ier in {1,2,3,4}).  MINPACK's source is not shipped with scipy's wheel, so the routines
 * below are written from the published algorithm (More, Garbow, Hillstrom, "User Guide for
 * MINPACK-1", ANL-80-74, 1980: lmdif, fdjac2, qrfac, lmpar, qrsolv, enorm), for the general
 * (m, n) case.  The restatement is PINNED: tests/test_oracle.py checks it against scipy itself
 * (present both in the build container and on the GPU box) and against golden vectors produced
 * by running the real reference code (oracle/make_golden.py -> tests/golden/).
 *
 * jac_mode: 0 = forward-difference Jacobian (what lmdif/scipy does; the parity anchor)
 *           1 = analytic Jacobian with lmdif's nfev accounting
 *           2 = forward differences emulated without extra exponentials (what the HIP kernel does;
 *               lets tests separate "emulated-vs-true FD" differences from HIP bugs)
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef ORACLE_TRACE
#include <stdio.h>
#endif

#define MAXN 8   /* parameters */
#define MAXM 64  /* samples per voxel */

typedef void (*model_fn)(int m, int n, const double *p, const double *xs, const double *ys,
                         double *fvec);
typedef void (*jac_fn)(int m, int n, const double *p, const double *xs, double *fjac, int ld);

/* ---- MINPACK enorm: scaled Euclidean norm with three accumulators ------------------------- */
static double enorm(int n, const double *x) {
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    for (int i = 0; i < n; ++i) {
        const double xabs = fabs(x[i]);
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

/* ---- qrfac: Householder QR with column pivoting; a is column-major a[j*ld + i] ------------- */
static void qrfac(int m, int n, double *a, int ld, int *ipvt, double *rdiag, double *acnorm,
                  double *wa) {
    const double epsmch = DBL_EPSILON;
    for (int j = 0; j < n; ++j) {
        acnorm[j] = enorm(m, a + (size_t)j * ld);
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        ipvt[j] = j;
    }
    const int minmn = m < n ? m : n;
    for (int j = 0; j < minmn; ++j) {
        int kmax = j;
        for (int k = j; k < n; ++k)
            if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < m; ++i) {
                const double t = a[(size_t)j * ld + i];
                a[(size_t)j * ld + i] = a[(size_t)kmax * ld + i];
                a[(size_t)kmax * ld + i] = t;
            }
            rdiag[kmax] = rdiag[j];
            wa[kmax] = wa[j];
            const int t = ipvt[j];
            ipvt[j] = ipvt[kmax];
            ipvt[kmax] = t;
        }
        double ajnorm = enorm(m - j, a + (size_t)j * ld + j);
        if (ajnorm != 0.0) {
            if (a[(size_t)j * ld + j] < 0.0) ajnorm = -ajnorm;
            for (int i = j; i < m; ++i) a[(size_t)j * ld + i] /= ajnorm;
            a[(size_t)j * ld + j] += 1.0;
            for (int k = j + 1; k < n; ++k) {
                double sum = 0.0;
                for (int i = j; i < m; ++i) sum += a[(size_t)j * ld + i] * a[(size_t)k * ld + i];
                const double temp = sum / a[(size_t)j * ld + j];
                for (int i = j; i < m; ++i) a[(size_t)k * ld + i] -= temp * a[(size_t)j * ld + i];
                if (rdiag[k] != 0.0) {
                    double t = a[(size_t)k * ld + j] / rdiag[k];
                    double d = 1.0 - t * t;
                    if (d < 0.0) d = 0.0;
                    rdiag[k] *= sqrt(d);
                    t = rdiag[k] / wa[k];
                    if (0.05 * t * t <= epsmch) {
                        rdiag[k] = enorm(m - j - 1, a + (size_t)k * ld + j + 1);
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

/* ---- qrsolv: solve [R; D] x ~ [Q^T b; 0] by Givens rotations ------------------------------- */
static void qrsolv(int n, double *r, int ld, const int *ipvt, const double *diag,
                   const double *qtb, double *x, double *sdiag, double *wa) {
    for (int j = 0; j < n; ++j) {
        for (int i = j; i < n; ++i) r[(size_t)j * ld + i] = r[(size_t)i * ld + j];
        x[j] = r[(size_t)j * ld + j];
        wa[j] = qtb[j];
    }
    for (int j = 0; j < n; ++j) {
        const int l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int k = j; k < n; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int k = j; k < n; ++k) {
                if (sdiag[k] == 0.0) continue;
                double c, s;
                const double rkk = r[(size_t)k * ld + k];
                if (fabs(rkk) < fabs(sdiag[k])) {
                    const double cotan = rkk / sdiag[k];
                    s = 0.5 / sqrt(0.25 + 0.25 * cotan * cotan);
                    c = s * cotan;
                } else {
                    const double tn = sdiag[k] / rkk;
                    c = 0.5 / sqrt(0.25 + 0.25 * tn * tn);
                    s = c * tn;
                }
                r[(size_t)k * ld + k] = c * rkk + s * sdiag[k];
                const double temp = c * wa[k] + s * qtbpj;
                qtbpj = -s * wa[k] + c * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; ++i) {
                    const double t = c * r[(size_t)k * ld + i] + s * sdiag[i];
                    sdiag[i] = -s * r[(size_t)k * ld + i] + c * sdiag[i];
                    r[(size_t)k * ld + i] = t;
                }
            }
        }
        sdiag[j] = r[(size_t)j * ld + j];
        r[(size_t)j * ld + j] = x[j];
    }
    int nsing = n;
    for (int j = 0; j < n; ++j) {
        if (sdiag[j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        const int j = nsing - k;
        double sum = 0.0;
        for (int i = j + 1; i < nsing; ++i) sum += r[(size_t)j * ld + i] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa[j];
}

/* ---- lmpar: Levenberg-Marquardt parameter for the trust region ||D x|| <= delta ------------ */
static void lmpar(int n, double *r, int ld, const int *ipvt, const double *diag, const double *qtb,
                  double delta, double *par, double *x, double *sdiag, double *wa1, double *wa2) {
    const double dwarf = DBL_MIN;
    int nsing = n;
    for (int j = 0; j < n; ++j) {
        wa1[j] = qtb[j];
        if (r[(size_t)j * ld + j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        const int j = nsing - k;
        wa1[j] /= r[(size_t)j * ld + j];
        const double temp = wa1[j];
        for (int i = 0; i < j; ++i) wa1[i] -= r[(size_t)j * ld + i] * temp;
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa1[j];

    int iter = 0;
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm(n, wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        *par = 0.0;
        return;
    }
    double parl = 0.0;
    if (nsing >= n) {
        for (int j = 0; j < n; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; ++j) {
            double sum = 0.0;
            for (int i = 0; i < j; ++i) sum += r[(size_t)j * ld + i] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[(size_t)j * ld + j];
        }
        const double temp = enorm(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; ++j) {
        double sum = 0.0;
        for (int i = 0; i <= j; ++i) sum += r[(size_t)j * ld + i] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = enorm(n, wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = dwarf / (delta < 0.1 ? delta : 0.1);
    if (*par < parl) *par = parl;
    if (*par > paru) *par = paru;
    if (*par == 0.0) *par = gnorm / dxnorm;

    for (;;) {
        ++iter;
        if (*par == 0.0) *par = dwarf > 0.001 * paru ? dwarf : 0.001 * paru;
        double temp = sqrt(*par);
        for (int j = 0; j < n; ++j) wa1[j] = temp * diag[j];
        qrsolv(n, r, ld, ipvt, wa1, qtb, x, sdiag, wa2);
        for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm(n, wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10)
            break;
        for (int j = 0; j < n; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; ++j) {
            wa1[j] /= sdiag[j];
            const double t = wa1[j];
            for (int i = j + 1; i < n; ++i) wa1[i] -= r[(size_t)j * ld + i] * t;
        }
        temp = enorm(n, wa1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0 && *par > parl) parl = *par;
        if (fp < 0.0 && *par < paru) paru = *par;
        *par = parl > *par + parc ? parl : *par + parc;
    }
}

/* ---- lmdif driver (mode = 1) -------------------------------------------------------------- */
static int lm_solve(model_fn fcn, jac_fn jac, int jac_mode, int m, int n, double *x,
                    const double *xs, const double *ys, double ftol, double xtol, double gtol,
                    int maxfev, double epsfcn, double factor, int *nfev_out) {
    const double epsmch = DBL_EPSILON;
    double fvec[MAXM], wa4[MAXM], fjac[MAXM * MAXN];
    double diag[MAXN], qtf[MAXN], wa1[MAXN], wa2[MAXN], wa3[MAXN];
    int ipvt[MAXN];
    const int ld = m;
    int info = 0, nfev = 0, iter = 1;
    double par = 0.0, delta = 0.0, xnorm = 0.0, gnorm = 0.0;

    if (n <= 0 || m < n || ftol < 0.0 || xtol < 0.0 || gtol < 0.0 || maxfev <= 0 || factor <= 0.0) {
        *nfev_out = 0;
        return 0;
    }
    fcn(m, n, x, xs, ys, fvec);
    nfev = 1;
    double fnorm = enorm(m, fvec);

    for (;;) { /* outer loop */
        if (jac_mode == 0) {
            const double eps = sqrt(epsfcn > epsmch ? epsfcn : epsmch);
            for (int j = 0; j < n; ++j) {
                const double temp = x[j];
                double h = eps * fabs(temp);
                if (h == 0.0) h = eps;
                x[j] = temp + h;
                fcn(m, n, x, xs, ys, wa4);
                x[j] = temp;
                for (int i = 0; i < m; ++i) fjac[(size_t)j * ld + i] = (wa4[i] - fvec[i]) / h;
            }
        } else if (jac_mode == 2) {
            /* forward differences emulated WITHOUT extra exps (monoexponential only; what the HIP
             * kernel does): column a is lmdif's difference quotient exactly (same e_i); column b uses
             * e_i * exp(delta_i) with exp(delta) = 1 + d + d^2/2 + d^3/6 for the ~1e-8 argument shift. */
            const double eps = sqrt(epsfcn > epsmch ? epsfcn : epsmch);
            const double a = x[0], b = x[1];
            double ha = eps * fabs(a), hb = eps * fabs(b);
            if (ha == 0.0) ha = eps;
            if (hb == 0.0) hb = eps;
            const double a1 = a + ha, b1 = b + hb;
            for (int i = 0; i < m; ++i) {
                const double bx = b * xs[i];
                const double e = exp(bx);
                fjac[i] = ((a1 * e - ys[i]) - fvec[i]) / ha;
                const double d = b1 * xs[i] - bx;
                const double e1 = e + e * (d + d * d * (0.5 + d * (1.0 / 6.0)));
                fjac[ld + i] = ((a * e1 - ys[i]) - fvec[i]) / hb;
            }
        } else {
            jac(m, n, x, xs, fjac, ld);
        }
        nfev += n; /* lmdif charges n evaluations per Jacobian; kept for the analytic mode too */

        qrfac(m, n, fjac, ld, ipvt, wa1, wa2, wa3);
        if (iter == 1) {
            for (int j = 0; j < n; ++j) {
                diag[j] = wa2[j];
                if (wa2[j] == 0.0) diag[j] = 1.0;
            }
            for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        for (int i = 0; i < m; ++i) wa4[i] = fvec[i];
        for (int j = 0; j < n; ++j) {
            if (fjac[(size_t)j * ld + j] != 0.0) {
                double sum = 0.0;
                for (int i = j; i < m; ++i) sum += fjac[(size_t)j * ld + i] * wa4[i];
                const double temp = -sum / fjac[(size_t)j * ld + j];
                for (int i = j; i < m; ++i) wa4[i] += fjac[(size_t)j * ld + i] * temp;
            }
            fjac[(size_t)j * ld + j] = wa1[j];
            qtf[j] = wa4[j];
        }
        gnorm = 0.0;
        if (fnorm != 0.0) {
            for (int j = 0; j < n; ++j) {
                const int l = ipvt[j];
                if (wa2[l] == 0.0) continue;
                double sum = 0.0;
                for (int i = 0; i <= j; ++i) sum += fjac[(size_t)j * ld + i] * (qtf[i] / fnorm);
                const double g = fabs(sum / wa2[l]);
                if (g > gnorm) gnorm = g;
            }
        }
#ifdef ORACLE_TRACE
        printf("qr x=(%.17g,%.17g) acn=(%.17g,%.17g) l0=%d R=(%.17g,%.17g,%.17g) qtf=(%.17g,%.17g) gnorm=%.6g\n",
               x[0], x[1], wa2[0], wa2[1], ipvt[0], fjac[0], fjac[ld], fjac[ld + 1], qtf[0], qtf[1], gnorm);
#endif
        if (gnorm <= gtol) info = 4;
        if (info != 0) break;
        for (int j = 0; j < n; ++j)
            if (wa2[j] > diag[j]) diag[j] = wa2[j];

        double ratio;
        do { /* inner loop */
            lmpar(n, fjac, ld, ipvt, diag, qtf, delta, &par, wa1, wa2, wa3, wa4);
            for (int j = 0; j < n; ++j) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = enorm(n, wa3);
            if (iter == 1 && pnorm < delta) delta = pnorm;
            fcn(m, n, wa2, xs, ys, wa4);
            ++nfev;
            const double fnorm1 = enorm(m, wa4);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) {
                const double t = fnorm1 / fnorm;
                actred = 1.0 - t * t;
            }
            for (int j = 0; j < n; ++j) {
                wa3[j] = 0.0;
                const double temp = wa1[ipvt[j]];
                for (int i = 0; i <= j; ++i) wa3[i] += fjac[(size_t)j * ld + i] * temp;
            }
            const double temp1 = enorm(n, wa3) / fnorm;
            const double temp2 = (sqrt(par) * pnorm) / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
#ifdef ORACLE_TRACE
            printf("it nfev=%d trial=(%.17g,%.17g) p=(%.6g,%.6g) par=%.6g delta=%.6g fnorm=%.17g fnorm1=%.17g actred=%.6g prered=%.6g ratio=%.6g\n",
                   nfev, wa2[0], wa2[1], wa1[0], wa1[1], par, delta, fnorm, fnorm1, actred, prered, ratio);
#endif
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
                for (int j = 0; j < n; ++j) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                }
                for (int i = 0; i < m; ++i) fvec[i] = wa4[i];
                xnorm = enorm(n, wa2);
                fnorm = fnorm1;
                ++iter;
            }
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0 && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= epsmch * xnorm) info = 7;
            if (gnorm <= epsmch) info = 8;
            if (info != 0) break;
        } while (ratio < 1e-4);
        if (info != 0) break;
    }
    *nfev_out = nfev;
    return info;
}

/* ---- models: residual = f(x; p) - y  (scipy _wrap_func: func(xdata, *params) - ydata) ------- */
/* dosma/core/fitting.py:1016-1018  monoexponential(x, a, b) = a * exp(b * x) */
static void monoexp_res(int m, int n, const double *p, const double *xs, const double *ys,
                        double *fvec) {
    (void)n;
    for (int i = 0; i < m; ++i) fvec[i] = p[0] * exp(p[1] * xs[i]) - ys[i];
}
static void monoexp_jac(int m, int n, const double *p, const double *xs, double *fjac, int ld) {
    (void)n;
    for (int i = 0; i < m; ++i) {
        const double e = exp(p[1] * xs[i]);
        fjac[i] = e;
        fjac[ld + i] = p[0] * xs[i] * e;
    }
}
/* dosma/core/fitting.py:1021-1023  biexponential(x, a1, b1, a2, b2) */
static void biexp_res(int m, int n, const double *p, const double *xs, const double *ys,
                      double *fvec) {
    (void)n;
    for (int i = 0; i < m; ++i)
        fvec[i] = p[0] * exp(p[1] * xs[i]) + p[2] * exp(p[3] * xs[i]) - ys[i];
}
static void biexp_jac(int m, int n, const double *p, const double *xs, double *fjac, int ld) {
    (void)n;
    for (int i = 0; i < m; ++i) {
        const double e1 = exp(p[1] * xs[i]), e2 = exp(p[3] * xs[i]);
        fjac[i] = e1;
        fjac[ld + i] = p[0] * xs[i] * e1;
        fjac[2 * ld + i] = e2;
        fjac[3 * ld + i] = p[2] * xs[i] * e2;
    }
}

/*
 * oracle_curve_fit: the per-voxel loop of dosma/core/fitting.py:855-868 over _curve_fit (:1026-1073).
 *   model     0 = monoexponential (n = 2), 1 = biexponential (n = 4)
 *   y         [E][N] echo-major, float64 (scipy converts with asarray_chkfinite(ydata, float))
 *   p0s       [n] scalar initial guess; p0v[j] (nullable, length N) overrides parameter j per voxel
 *   popt      [N][n]; r2 [N]; info/nfev nullable [N] (info 0 = skipped by the all-zero rule)
 * Returns 0, or -1 on bad arguments, or -2 if a sample is not finite (the reference raises
 * ValueError for the whole call: scipy check_finite).
 */
int oracle_curve_fit(int model, const double *x, int E, const double *y, int64_t N,
                     const double *p0s, const double *const *p0v, double ftol, double xtol,
                     double gtol, int maxfev, double epsfcn, double factor, double r2_eps,
                     int jac_mode, double *popt, double *r2, int32_t *info_out,
                     int32_t *nfev_out) {
    const int n = model == 0 ? 2 : 4;
    model_fn fcn = model == 0 ? monoexp_res : biexp_res;
    jac_fn jac = model == 0 ? monoexp_jac : biexp_jac;
    if (E <= 0 || E > MAXM || (model != 0 && model != 1)) return -1;
    int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1024) reduction(| : bad)
#endif
    for (int64_t v = 0; v < N; ++v) {
        double ys[MAXM], p[MAXN], fv[MAXM];
        int allzero = 1, finite = 1;
        for (int e = 0; e < E; ++e) {
            ys[e] = y[(size_t)e * N + v];
            if (ys[e] != 0.0) allzero = 0;
            if (!isfinite(ys[e])) finite = 0;
        }
        int info = 0, nfev = 0;
        if (allzero) {
            for (int j = 0; j < n; ++j) popt[(size_t)v * n + j] = NAN;
            r2[v] = 0.0;
        } else if (!finite) {
            bad |= 1;
            for (int j = 0; j < n; ++j) popt[(size_t)v * n + j] = NAN;
            r2[v] = 0.0;
        } else {
            for (int j = 0; j < n; ++j) p[j] = (p0v && p0v[j]) ? p0v[j][v] : p0s[j];
            info = lm_solve(fcn, jac, jac_mode, E, n, p, x, ys, ftol, xtol, gtol, maxfev, epsfcn,
                            factor, &nfev);
            if (info >= 1 && info <= 4) {
                /* fitting.py:1032-1035 */
                fcn(E, n, p, x, ys, fv); /* fv = model - y ; residuals = -fv */
                double ss_res = 0.0, mean = 0.0, ss_tot = 0.0;
                for (int e = 0; e < E; ++e) ss_res += fv[e] * fv[e];
                for (int e = 0; e < E; ++e) mean += ys[e];
                mean /= (double)E;
                for (int e = 0; e < E; ++e) ss_tot += (ys[e] - mean) * (ys[e] - mean);
                for (int j = 0; j < n; ++j) popt[(size_t)v * n + j] = p[j];
                r2[v] = 1.0 - ss_res / (ss_tot + r2_eps);
            } else {
                for (int j = 0; j < n; ++j) popt[(size_t)v * n + j] = NAN;
                r2[v] = 0.0;
            }
        }
        if (info_out) info_out[v] = info;
        if (nfev_out) nfev_out[v] = nfev;
    }
    return bad ? -2 : 0;
}

int oracle_max_samples(void) { return MAXM; }
