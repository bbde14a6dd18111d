"""CPU oracle for DOSMA's per-voxel curve-fit path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; it is the checker, never the product.  ``dosma_amd`` never imports it.

Three layers, each citing the reference lines it follows (paths relative to /root/reference):

1. :func:`curve_fit_scipy` -- the reference's own call pattern, verbatim in behaviour:
   one ``scipy.optimize.curve_fit`` per voxel (dosma/core/fitting.py:1026-1073, loop :855-859).
   Slow (a few k fits/s) -- used to pin layer 2 and as the reported CPU baseline.
2. :func:`curve_fit_c` -- ``oracle/minpack_oracle.c``: MINPACK lmdif restated in plain C from the
   published algorithm, ~1e6 fits/s, so parity tests can run on 1e5..1e6 voxels in seconds.
3. numpy restatements of the plumbing around the solver: log-linear initial guess
   (:701-718 + :926-944 + :976), post-processing (:109-146), mask gather/scatter (:199-215),
   rounding (:736-737) -> :func:`monoexp_fit_arrays`.

Pinning: tests/test_oracle.py checks (2) against (1) on seeded data and both against the golden
vectors in tests/golden/ which were produced by the *real* reference (oracle/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_SO = os.path.join(_BUILD, "libminpack_oracle.so")

# DOSMA / scipy constants (dosma/core/fitting.py:761-763; scipy leastsq defaults)
FTOL = 1e-5
XTOL = 1.49012e-8
GTOL = 0.0
MAXFEV = 100
EPSFCN = float(np.finfo(np.float64).eps)
FACTOR = 100.0
R2_EPS = 1e-8


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, no dependencies). Returns the path of the .so."""
    src = os.path.join(_HERE, "minpack_oracle.c")
    if (not force) and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(src):
        return _SO
    os.makedirs(_BUILD, exist_ok=True)
    cmd = ["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
           "-o", _SO, src, "-lm"]
    subprocess.check_call(cmd)
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        lib.oracle_curve_fit.restype = ctypes.c_int
        lib.oracle_curve_fit.argtypes = [
            ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int64, dp, ctypes.POINTER(dp),
            ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double,
            ctypes.c_double, ctypes.c_double, ctypes.c_int, dp, dp,
            ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
        ]
        _lib = lib
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def curve_fit_c(x, y, p0=(1.0, 1.0), model="monoexponential", jac_mode=0, ftol=FTOL, xtol=XTOL,
                gtol=GTOL, maxfev=MAXFEV, epsfcn=EPSFCN, factor=FACTOR, eps=R2_EPS,
                threads=1, full_output=False):
    """C restatement of ``curve_fit`` (dosma/core/fitting.py:755-870) for the two built-in models.

    ``y`` is (E, N) echo-major; ``p0`` is a sequence with one entry per parameter, each a scalar
    or a length-N array (the reference's p0_scalars / p0_seq split, fitting.py:1106-1161).
    Raises ValueError on non-finite samples like scipy's ``check_finite``.
    """
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.asarray(y)
    if y.ndim == 1:
        y = y.reshape(-1, 1)
    y = np.ascontiguousarray(y, dtype=np.float64)
    E, N = y.shape
    imodel = {"monoexponential": 0, "biexponential": 1}[model]
    n = 2 if imodel == 0 else 4
    p0 = [1.0 if v is None else v for v in p0]
    assert len(p0) == n
    p0s = np.zeros(n)
    keep = []
    ptrs = (ctypes.POINTER(ctypes.c_double) * n)()
    for j, v in enumerate(p0):
        if isinstance(v, np.ndarray) and v.ndim >= 1:
            arr = np.ascontiguousarray(v, dtype=np.float64).reshape(-1)
            assert arr.shape[0] == N
            keep.append(arr)
            ptrs[j] = _dp(arr)
        else:
            p0s[j] = float(v)
            ptrs[j] = None
    popt = np.empty((N, n))
    r2 = np.empty(N)
    info = np.empty(N, dtype=np.int32)
    nfev = np.empty(N, dtype=np.int32)
    if threads:
        os.environ["OMP_NUM_THREADS"] = str(int(threads))
        try:
            omp = ctypes.CDLL("libgomp.so.1")
            omp.omp_set_num_threads(int(threads))
        except OSError:
            pass
    rc = lib.oracle_curve_fit(
        imodel, _dp(x), E, _dp(y), N, _dp(p0s), ptrs, ftol, xtol, gtol, int(maxfev), epsfcn,
        factor, eps, int(jac_mode), _dp(popt), _dp(r2),
        info.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        nfev.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if rc == -2:
        raise ValueError("array must not contain infs or NaNs")
    if rc != 0:
        raise RuntimeError(f"oracle_curve_fit failed rc={rc}")
    if full_output:
        return popt, r2, info, nfev
    return popt, r2


def monoexponential(x, a, b):
    """dosma/core/fitting.py:1016-1018."""
    return a * np.exp(b * x)


def biexponential(x, a1, b1, a2, b2):
    """dosma/core/fitting.py:1021-1023."""
    return a1 * np.exp(b1 * x) + a2 * np.exp(b2 * x)


def _one_voxel_scipy(args):
    """dosma/core/fitting.py:1026-1073 (_curve_fit) for one voxel, on scipy itself."""
    import warnings

    from scipy import optimize as sop

    func, x, y, p0, ftol, maxfev, eps, nparams, full = args
    if (y == 0).all():
        out = (np.nan,) * nparams, 0.0
        return out + ((0, 0) if full else ())
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if full:
                popt, _, infodict, _, ier = sop.curve_fit(
                    func, x, y, p0=p0, ftol=ftol, maxfev=maxfev, full_output=True)
                nfev = infodict["nfev"]
            else:
                popt, _ = sop.curve_fit(func, x, y, p0=p0, ftol=ftol, maxfev=maxfev)
        residuals = y - func(x, *popt)
        ss_res = np.sum(residuals ** 2)
        ss_tot = np.sum((y - np.mean(y)) ** 2)
        r2 = 1 - (ss_res / (ss_tot + eps))
        out = tuple(popt), float(r2)
        return out + ((int(ier), int(nfev)) if full else ())
    except RuntimeError as err:
        out = (np.nan,) * nparams, 0.0
        if full:
            # scipy raises for ier not in 1..4; ier itself is not exposed -> report 5 (maxfev) when
            # the message says so, else 9
            ier = 5 if "maxfev" in str(err) else 9
            return out + (ier, -1)
        return out


def curve_fit_scipy(x, y, p0=None, func=monoexponential, ftol=FTOL, maxfev=MAXFEV, eps=R2_EPS,
                    num_workers=0, chunksize=1000, full_output=False):
    """The reference's per-voxel loop (dosma/core/fitting.py:855-868) on scipy.optimize.curve_fit.

    ``p0``: None, or a sequence with one scalar / length-N array per parameter.
    ``num_workers`` > 0 uses multiprocessing.Pool like the reference (:866-867).
    """
    import inspect

    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y)
    if y.ndim == 1:
        y = y.reshape(-1, 1)
    N = y.shape[-1]
    nparams = len(inspect.signature(func).parameters) - 1
    y_T = y.T

    def p0_at(i):
        if p0 is None:
            return None
        return tuple(
            1.0 if v is None else (float(v[i]) if isinstance(v, np.ndarray) and v.ndim else float(v))
            for v in p0)

    jobs = [(func, x, y_T[i], p0_at(i), ftol, maxfev, eps, nparams, full_output) for i in range(N)]
    if num_workers:
        import multiprocessing as mp

        with mp.Pool(num_workers) as pool:
            data = pool.map(_one_voxel_scipy, jobs, chunksize=chunksize)
    else:
        data = [_one_voxel_scipy(j) for j in jobs]
    popt = np.stack([np.asarray(d[0], dtype=np.float64) for d in data], axis=0)
    r2 = np.asarray([d[1] for d in data], dtype=np.float64)
    if full_output:
        return (popt, r2, np.asarray([d[2] for d in data], dtype=np.int32),
                np.asarray([d[3] for d in data], dtype=np.int32))
    return popt, r2


# ----------------------------------------------------------------------------- plumbing (numpy)
def loglin_init(x, y):
    """Log-linear initial guess of MonoExponentialFit (dosma/core/fitting.py:701-718).

    ``y`` (E, N) any real dtype.  Follows: ints -> float32 (:710-712); ``v + 1e-10 * (v == 0)``
    (:713, float64 result); ``log`` (:715); ``np.polyfit(x, logv, 1)`` as one joint solve (:976);
    r2 matrix (:926-944); PolyFitter post-processing with r2_threshold = 0, nan_to_num = 0.0
    (:703-709 -> :140-144); ``a0 = exp(intercept)``, ``b0 = slope`` (:717).
    Returns (a0, b0) float64 arrays of length N.
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y)
    if np.issubdtype(y.dtype, np.integer):
        y = y.astype(np.float32)
    v = y + 1e-10 * (y == 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        logv = np.log(v)
        popts = np.polyfit(x, logv, 1)  # (2, N): slope, intercept
        xs = np.stack([x ** 1, x ** 0], axis=-1)
        yhat = xs @ popts
        ss_res = np.sum((yhat - logv) ** 2, axis=0)
        ss_tot = np.sum((logv - np.mean(logv, axis=0, keepdims=True)) ** 2, axis=0)
        r2 = 1 - (ss_res / (ss_tot + R2_EPS))
    params = popts.T.copy()  # (N, 2)
    with np.errstate(invalid="ignore"):
        params[r2 < 0] = np.nan
    params = np.nan_to_num(params, nan=0.0)
    return np.exp(params[:, 1]), params[:, 0]


def process_params(popt, r2, out_ufuncs=None, out_bounds=None, r2_threshold=None, nan_to_num=None):
    """``_Fitter._process_params`` (dosma/core/fitting.py:109-146) on a (N, P) float64 array."""
    x = np.array(popt, dtype=np.float64, copy=True)
    nparams = x.shape[-1]
    with np.errstate(all="ignore"):
        if callable(out_ufuncs):
            x = out_ufuncs(x)
        elif out_ufuncs is not None:
            for i in range(min(nparams, len(out_ufuncs))):
                if out_ufuncs[i] is not None:
                    x[..., i] = out_ufuncs[i](x[..., i])
        if out_bounds is not None:
            ob = np.asarray(out_bounds, dtype=np.float64)
            if ob.ndim == 2:
                if ob.shape[0] < nparams:
                    extra = np.tile([[-np.inf, np.inf]], (nparams - ob.shape[0], 1))
                    ob = np.concatenate([ob, extra], axis=0)
                ob = ob.T
            lb, ub = ob[0], ob[1]
            x[(x < lb) | (x > ub)] = np.nan
        if r2_threshold is not None:
            x[r2 < r2_threshold] = np.nan
        if nan_to_num is not None:
            x = np.nan_to_num(x, nan=nan_to_num, copy=False)
    return x


def monoexp_fit_arrays(x, y, mask=None, bounds=(0, 100.0), tc0=30.0, r2_threshold=0.9,
                       decimal_precision=1, solver="c", jac_mode=0, threads=1):
    """``MonoExponentialFit.fit`` (dosma/core/fitting.py:678-739) on raw arrays.

    ``y`` (E, N) echo-major flattened volumes, ``mask`` optional (N,) (``> 0`` selects).
    Returns (tc (N,), r2 (N,), popt_full (N, 2)) float64 -- what the reference's MedicalVolumes hold.
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y)
    E, N = y.shape
    sel = None if mask is None else (np.asarray(mask).reshape(-1) > 0)
    if isinstance(tc0, str):
        assert tc0 == "polyfit"
        # the reference fits the whole (masked) volume; scatter with fill 0.0 then exp (:716-717)
        ysel = y if sel is None else y[:, sel]
        a_sel, b_sel = loglin_init(x, ysel)
        if sel is None:
            a0, b0 = a_sel, b_sel
        else:
            # params outside the mask are nan_to_num = 0.0 -> a0 = exp(0) = 1, b0 = 0
            pa = np.ones(N)
            pb = np.zeros(N)
            pa[sel] = a_sel
            pb[sel] = b_sel
            a0, b0 = pa, pb
        p0 = [a0, b0]
    else:
        p0 = [1.0, -1 / tc0]
    ysel = y if sel is None else y[:, sel]
    p0sel = [v[sel] if (sel is not None and isinstance(v, np.ndarray)) else v for v in p0]
    if solver == "c":
        popt, r2 = curve_fit_c(x, ysel, p0sel, jac_mode=jac_mode, threads=threads)
    else:
        popt, r2 = curve_fit_scipy(x, ysel, p0sel)
    popt = process_params(
        popt, r2, out_ufuncs=(None, lambda v: 1 / np.abs(v)),
        out_bounds=((-np.inf, np.inf), bounds), r2_threshold=r2_threshold, nan_to_num=0.0)
    if sel is not None:
        popt_full = np.zeros((N, 2))
        r2_full = np.zeros(N)
        popt_full[sel] = popt
        r2_full[sel] = r2
        popt, r2 = popt_full, r2_full
    tc = popt[:, 1]
    if decimal_precision is not None:
        tc = np.around(tc, decimal_precision)
    return tc, r2, popt


# ----------------------------------------------------------------------------- qDESS
def dess_t2_numpy(echo_1, echo_2, tr, te, tg, alpha_deg, gl_area, t1, diffusivity=1.25e-9,
                  suppress_fat=False, suppress_fluid=False, beta=1.2, nan_bounds=(0, 100),
                  nan_to_num=0.0, decimals=1):
    """numpy restatement of QDess.generate_t2_map's arithmetic (dosma/scan_sequences/mri/qdess.py:190-245)."""
    import math

    xp = np
    TR, TE, Tg, T1 = tr * 1e-3, te * 1e-3, tg * 1e-6, t1 * 1e-3
    alpha = math.radians(alpha_deg)
    Gl = gl_area / (Tg * 1e6) * 100
    gamma = 4258 * 2 * math.pi
    dkL = gamma * Gl * Tg
    k = (xp.power((xp.sin(alpha / 2)), 2)
         * (1 + xp.exp(-TR / T1 - TR * xp.power(dkL, 2) * diffusivity))
         / (1 - xp.cos(alpha) * xp.exp(-TR / T1 - TR * xp.power(dkL, 2) * diffusivity)))
    c1 = (TR - Tg / 3) * (xp.power(dkL, 2)) * diffusivity
    with np.errstate(all="ignore"):
        mask = xp.ones(echo_1.shape)
        ratio = xp.nan_to_num(mask * echo_2 / echo_1)
        t2map = -2000 * (TR - TE) / (xp.log(abs(ratio) / k) + c1)
        t2map = xp.nan_to_num(t2map)
        if nan_bounds is not None:
            lower, upper = nan_bounds
            t2map[(t2map < lower) | (t2map > upper)] = xp.nan
        if nan_to_num is not None:
            t2map = xp.nan_to_num(t2map) if isinstance(nan_to_num, bool) else xp.nan_to_num(t2map, nan=nan_to_num)
        if decimals is not None:
            t2map = xp.around(t2map, decimals)
        if suppress_fat:
            t2map = t2map * (echo_1 > 0.15 * xp.max(echo_1))
        if suppress_fluid:
            vol_null_fluid = echo_1 - beta * echo_2
            t2map = t2map * (vol_null_fluid > 0.1 * xp.max(vol_null_fluid))
    return t2map
