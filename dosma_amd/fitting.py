"""Drop-in fitting interface of DOSMA on the MI355X kernels.

Mirrors the public surface of the reference's ``dosma/core/fitting.py`` for the hot path named in
BASELINE.json -- same names, arguments and error behaviour:

* :func:`curve_fit`               (reference :755-870)   <- the seam: the per-voxel loop becomes ONE HIP launch
* :class:`CurveFitter`            (reference :238-458)
* :class:`MonoExponentialFit`     (reference :607-749)   <- init, fit and post-processing fused in the kernel
* :func:`polyfit`, :class:`PolyFitter` (reference :461-604, 873-1013): any degree, ``rcond`` / ``w`` / ``full`` / ``cov``
* :func:`monoexponential`, :func:`biexponential` (reference :1016-1023)

What runs where: everything per-voxel runs in ``libqmri_hip.so`` (include/qmri.h).  Python only does
what the reference's Python does *around* the loop: argument checking, reorientation, flattening to
the (E, N) echo-major array, p0 formatting, wrapping results in MedicalVolumes.  There is no CPU
solver for the models the kernels implement: a missing library or GPU raises ``dosma_amd._lib.QmriError``.
A request the kernels do NOT implement -- a generic Python ``func``, scipy ``**kwargs`` that select another
solver (``bounds=``, ``sigma=``, ``jac=``, ``method=``), more samples per voxel than the kernels keep -- is
served the way the reference serves every request: one ``scipy.optimize.curve_fit`` per voxel
(``dosma_amd/_scipy_loop.py``; SURVEY 8(b)'s dispatch rule, reference :827-870, :1026-1073).

Like the reference (:403-406, 745-746, 809-810) inputs must live on the CPU; the library stages
them to the GPU itself.
"""
import inspect
import warnings
from copy import deepcopy
from numbers import Number
from typing import Callable, Mapping, Sequence

import numpy as np

from dosma_amd import _lib
from dosma_amd import defaults
from dosma_amd.defaults import preferences
from dosma_amd.med_volume import MedicalVolume

__all__ = [
    "CurveFitter",
    "PolyFitter",
    "MonoExponentialFit",
    "curve_fit",
    "polyfit",
    "monoexponential",
    "biexponential",
]

__EPSILON__ = 1e-8


def monoexponential(x, a, b):
    """Function: :math:`f(x) = a * e^{b*x}` (reference :1016-1018)."""
    return a * np.exp(b * x)


def biexponential(x, a1, b1, a2, b2):
    """Function: :math:`f(x) = a1*e^{b1*x} + a2*e^{b2*x}` (reference :1021-1023)."""
    return a1 * np.exp(b1 * x) + a2 * np.exp(b2 * x)


def _inv_abs(v):
    """``1 / |v|`` -- the time-constant transform of MonoExponentialFit (reference :725)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return 1 / np.abs(v)


_inv_abs.__qmri_ufunc__ = "inv_abs"


def _model_of(func) -> str:
    """Which built-in kernel model ``func`` is.  Raises NotImplementedError for anything else (the callers then take the
    reference's per-voxel scipy route: :func:`_kernel_route`).

    ``func`` is recognised by identity, by a ``__qmri_model__`` attribute, or -- for a user's own
    2-parameter callable such as ``lambda x, a, b: a * np.exp(b * x)`` -- by evaluating it on a few
    probe points against ``a * exp(b * x)``.
    """
    if func is monoexponential or getattr(func, "__qmri_model__", None) == "monoexponential":
        return "monoexponential"
    if func is biexponential or getattr(func, "__qmri_model__", None) == "biexponential":
        return "biexponential"
    try:
        params = list(inspect.signature(func).parameters)
    except (TypeError, ValueError):
        params = []
    nparams = len(params) - 2 if "self" in params else len(params) - 1
    if nparams == 2:
        rng = np.random.default_rng(12345)
        xs = rng.uniform(0.1, 3.0, 7)
        try:
            ok = all(
                np.allclose(np.asarray(func(xs, a, b), dtype=float), a * np.exp(b * xs),
                            rtol=1e-12, atol=0)
                for a, b in rng.uniform(-2, 2, (4, 2)))
        except Exception:
            ok = False
        if ok:
            return "monoexponential"
    name = getattr(func, "__name__", type(func).__name__)
    raise NotImplementedError(
        f"func={name!r} is neither the mono-exponential (a*exp(b*x)) nor the bi-exponential model the kernels implement")


def _func_param_names(func):
    args = list(inspect.signature(func).parameters)
    return args[2:] if "self" in args else args[1:]


def _format_p0(p0, param_args, N):
    """Normalise ``p0`` like the reference's module-level ``_format_p0`` (:1106-1161).

    Returns a list with one entry per parameter: a float, or a float64 array of length N.
    """
    nparams = len(param_args)
    if p0 is None:
        return [1.0] * nparams  # scipy: p0 = ones(n)
    if isinstance(p0, Number):
        p0 = (p0,) * nparams
    elif isinstance(p0, np.ndarray) and p0.ndim > 1:
        p0 = tuple(p0[..., i] for i in range(p0.shape[-1]))

    if isinstance(p0, Mapping):
        extra = set(p0) - set(param_args)
        if extra:
            raise ValueError(f"`p0` has unknown keys: {extra}. "
                             f"Function signature has parameters {param_args}.")
        merged = {p: 1.0 for p in param_args}
        merged.update(p0)
        p0 = [merged[k] for k in param_args]
    elif isinstance(p0, (np.ndarray, Sequence)):
        if len(p0) != nparams:
            raise ValueError(f"`p0` has length {len(p0)} but function has {nparams} parameters")
        p0 = list(p0)
    else:
        raise ValueError(f"p0={p0} not supported")

    out = []
    for name, v in zip(param_args, p0):
        if v is None:
            out.append(1.0)
        elif isinstance(v, np.ndarray) and v.ndim >= 1:
            if len(v) != N:
                raise ValueError(f"Got {len(v)} values for param '{name}'. Expected {N}")
            out.append(np.ascontiguousarray(v, dtype=np.float64).reshape(-1))
        else:
            out.append(float(v))
    return out


_SUPPORTED_DTYPES = (np.float32, np.float64, np.int16, np.uint16)


def _as_kernel_samples(y):
    """Samples in a dtype the kernel reads natively; other real dtypes are widened losslessly."""
    if y.dtype in [np.dtype(t) for t in _SUPPORTED_DTYPES]:
        return np.ascontiguousarray(y)
    if y.dtype == np.bool_ or (np.issubdtype(y.dtype, np.integer) and y.dtype.itemsize <= 1):
        return np.ascontiguousarray(y, dtype=np.int16)
    if np.issubdtype(y.dtype, np.floating) and y.dtype.itemsize < 4:
        return np.ascontiguousarray(y, dtype=np.float32)
    if np.issubdtype(y.dtype, np.integer) or np.issubdtype(y.dtype, np.floating):
        # int32/int64/longdouble: scipy converts the samples to float64 anyway (asarray_chkfinite)
        return np.ascontiguousarray(y, dtype=np.float64)
    raise TypeError(f"Cannot fit data of dtype {y.dtype}")


def _as_kernel_rows(rows):
    """Per-echo flattened volumes in ONE dtype the kernel reads natively (the reference's np.concatenate promotes
    mixed dtypes to a common one, fitting.py:195)."""
    dt = np.result_type(*[r.dtype for r in rows])
    probe = _as_kernel_samples(np.empty(0, dtype=dt)).dtype
    return [np.ascontiguousarray(r, dtype=probe) for r in rows]


def curve_fit(
    func,
    x,
    y,
    y_bounds=None,
    p0=None,
    maxfev=100,
    ftol=1e-5,
    eps=1e-8,
    show_pbar=False,
    num_workers=0,
    chunksize: int = None,
    **kwargs,
):
    """Non-linear least squares fit of ``func`` to every column of ``y`` -- one GPU launch.

    Same contract as the reference's ``curve_fit`` (:755-870): ``x`` (E,), ``y`` (E,) or (E, N)
    echo-major, ``p0`` None / number / sequence / dict / (N, P) array with scalar or length-N entries;
    returns ``popts`` (N, P) float64 and ``r_squared`` (N,) float64; a voxel that is all zero, out of
    ``y_bounds``, or whose fit does not converge (MINPACK info not in 1..4) is ``(nan, nan), 0``.
    ``show_pbar`` / ``num_workers`` / ``chunksize`` are accepted for compatibility and ignored (there
    is no per-voxel Python loop to parallelise).  Of the extra scipy ``**kwargs`` the MINPACK options
    ``xtol`` / ``gtol`` / ``factor`` (and ``method="lm"``) go to the kernels; a generic ``func``, ``bounds=``,
    ``sigma=``, ``jac=``, another ``method`` ... take the reference's own route: one ``scipy.optimize.curve_fit``
    per voxel on the CPU (``_scipy_loop.py``; there ``num_workers`` / ``chunksize`` mean what they mean in the reference).
    """
    if isinstance(x, MedicalVolume) or isinstance(y, MedicalVolume):
        raise TypeError("`x` and `y` must be array-like (use CurveFitter for MedicalVolumes)")
    x = np.asarray(x)
    y = np.asarray(y)
    if y.ndim == 1:
        y = y.reshape(y.shape + (1,))
    if y.ndim != 2:
        raise ValueError("`y` must have shape (M,) or (M, N)")
    N = y.shape[-1]
    param_args = _func_param_names(func)
    given_p0 = p0 is not None
    p0 = _format_p0(p0, param_args, N)

    if y_bounds is not None and ((y < y_bounds[0]).any() or (y > y_bounds[1]).any()):
        warnings.warn("Out of bounds values found. Failure in fit will result in np.nan")

    model, solver = _kernel_route(func, kwargs, x.reshape(-1).shape[0], "curve_fit")
    if model is None:
        return _scipy_loop(func, x, y, p0, given_p0=given_p0, y_bounds=y_bounds, maxfev=maxfev, ftol=ftol, eps=eps,
                           num_workers=num_workers, chunksize=chunksize, scipy_kwargs=kwargs, why=solver)

    # general lmdif kernel (lm_generic.hip): the bi-exponential, and the mono-exponential beyond the 32 samples per voxel its own
    # kernel keeps in registers (the reference has no limit: fitting.py:755-870)
    if model != "monoexponential" or x.reshape(-1).shape[0] > _lib.MAX_ECHOES:
        if x.reshape(-1).shape[0] < len(param_args):
            raise TypeError("The number of func parameters must not exceed the number of data points")  # scipy
        out = _lib.lmfit_host(model, x.astype(np.float64).reshape(-1), _as_kernel_samples(y), p0,
                              ftol=ftol, maxfev=maxfev, r2_eps=eps, y_bounds=y_bounds, **solver)
        return out["popt"], out["r2"]
    per_voxel = any(isinstance(v, np.ndarray) for v in p0)
    out = _lib.monoexp_fit_host(
        x.astype(np.float64).reshape(-1), _as_kernel_samples(y),
        init=_lib.INIT_PER_VOXEL if per_voxel else _lib.INIT_SCALAR,
        p0=tuple(1.0 if isinstance(v, np.ndarray) else v for v in p0),
        a0v=p0[0] if isinstance(p0[0], np.ndarray) else None,
        b0v=p0[1] if isinstance(p0[1], np.ndarray) else None,
        ftol=ftol, maxfev=maxfev, r2_eps=eps, y_bounds=y_bounds, **solver,
    )
    return out["popt"], out["r2"]


def _polyfit_operator(x, deg, rcond, w):
    """The linear map ``numpy.polyfit(x, Y, deg, rcond, w=w)`` applies to every column of Y, built the way numpy builds
    it (numpy/lib/_polynomial_impl.py: weighted Vandermonde matrix, columns scaled to unit norm, lstsq with the rcond
    cut-off on the singular values).  Returns dict(solve (P, E), design (E, P), rank, singular_values, rcond, vbase
    (P, P) = inv(lhs^T lhs) of the scaled system un-scaled -- what ``cov=True`` multiplies with the residual variance)."""
    order = int(deg) + 1
    x = np.asarray(x, dtype=np.float64) + 0.0
    if deg < 0:
        raise ValueError("expected deg >= 0")
    if x.ndim != 1:
        raise TypeError("expected 1D vector for x")
    if x.size == 0:
        raise TypeError("expected non-empty vector for x")
    if rcond is None:
        rcond = len(x) * np.finfo(x.dtype).eps
    design = np.vander(x, order)
    lhs = design.copy()
    if w is not None:
        w = np.asarray(w, dtype=np.float64) + 0.0
        if w.ndim != 1:
            raise TypeError("expected a 1-d array for weights")
        if w.shape[0] != x.shape[0]:
            raise TypeError("expected w and y to have the same length")
        lhs *= w[:, np.newaxis]
    scale = np.sqrt((lhs * lhs).sum(axis=0))
    lhs /= scale
    u, sv, vt = np.linalg.svd(lhs, full_matrices=False)
    keep = sv > rcond * sv[0]
    rank = int(keep.sum())
    inv_sv = np.where(keep, 1.0 / np.where(keep, sv, 1.0), 0.0)
    pinv = (vt.T * inv_sv) @ u.T                      # (P, E): minimum-norm solution of the scaled system
    solve = pinv / scale[:, np.newaxis]
    if w is not None:
        solve = solve * w[np.newaxis, :]
    vbase = None
    if rank == order:
        vbase = np.linalg.inv(lhs.T @ lhs) / np.outer(scale, scale)
    return dict(solve=solve, design=design, rank=rank, singular_values=sv, rcond=rcond, vbase=vbase, w=w)


def polyfit(x, y, deg: int, rcond=None, full=False, w=None, cov=False, eps=1e-8, y_bounds=None,
            show_pbar=False, num_workers=None, chunksize: int = None):
    """Least-squares polynomial fit of every column of ``y`` (reference :873-1013) on the GPU.

    Returns ``popts`` (N, deg+1) in ``numpy.polyfit`` order (highest power first) and r2 (N,); with ``full=True`` also
    ``residuals, rank, singular_values, rcond`` and with ``cov=True`` the covariance matrices ``V`` (deg+1, deg+1, N),
    as ``numpy.polyfit`` returns them.  ``num_workers`` not None selects the reference's per-sequence rules (all-zero /
    out-of-``y_bounds`` columns -> NaN, r2 0; :1095-1097); like the reference it cannot be combined with full / cov.
    Degree 1 without numpy options runs on the closed-form kernel (linfit.hip); everything else is one (deg+1, E) linear
    map per voxel on the general kernel, built on the host exactly as numpy builds it (:func:`_polyfit_operator`).
    Any number of samples and any degree: up to 32 samples and degree 7 the map is applied from registers, beyond that by
    the kernel's streaming variant (slower per voxel, same results).
    """
    scatter_data = num_workers is not None
    if (cov or full) and scatter_data:
        raise ValueError("`cov` or `full` cannot be used with multiprocessing")
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    y = np.asarray(y)
    if y.ndim == 1:
        y = y.reshape(y.shape + (1,))
    if y_bounds is not None and ((y < y_bounds[0]).any() or (y > y_bounds[1]).any()):
        warnings.warn("Out of bounds values found. Failure in fit will result in np.nan")
    if x.shape[0] != y.shape[0]:
        raise TypeError("expected x and y to have same length")
    if deg == 1 and not (full or cov) and w is None and rcond is None and x.shape[0] <= _lib.MAX_ECHOES:
        out = _lib.linfit_host(x, _as_kernel_samples(y), r2_eps=eps, y_bounds=y_bounds, per_sequence_rules=scatter_data)
        return out["popt"], out["r2"]

    op = _polyfit_operator(x, deg, rcond, w)
    order = int(deg) + 1
    if op["rank"] != order and not full:
        warnings.warn("Polyfit may be poorly conditioned", np.exceptions.RankWarning if hasattr(np, "exceptions") else RuntimeWarning,
                      stacklevel=2)
    out = _lib.polyls_host(_as_kernel_samples(y), op["solve"], op["design"], w=op["w"], per_sequence_rules=scatter_data,
                           y_bounds=y_bounds if scatter_data else None, r2_eps=eps, want_resid=full or bool(cov))
    popts, r_squared = out["popt"], out["r2"]
    if full:
        # numpy (lstsq) reports the residual sums only for a full-rank, over-determined system
        resid = out["resid"] if (op["rank"] == order and x.shape[0] > order) else np.array([], dtype=np.float64)
        return popts, r_squared, resid, op["rank"], op["singular_values"], op["rcond"]
    if cov:
        if op["vbase"] is None:
            raise np.linalg.LinAlgError("Singular matrix")
        if cov == "unscaled":
            fac = 1  # numpy: Vbase[:, :, newaxis] * 1 -> shape (P, P, 1)
        else:
            if x.shape[0] <= order:
                raise ValueError("the number of data points must exceed order to scale the covariance matrix")
            fac = out["resid"] / (x.shape[0] - order)
        return popts, r_squared, op["vbase"][:, :, np.newaxis] * fac
    return popts, r_squared


# ---------------------------------------------------------------------------------------------------
# Post-processing options shared by the fitters.  The reference spreads this over four private helpers of
# `_Fitter` (fitting.py:60-146); here the options are validated once into a small table and applied by one
# function.  Message strings and the public attribute names are the reference's (they are API), the code is not.
# ---------------------------------------------------------------------------------------------------
def _checked_ufuncs(spec, nparams):
    """``out_ufuncs``: one callable for the whole (N, P) array, or a sequence with None / a callable per parameter."""
    per_param = not callable(spec)
    if per_param:
        try:
            entries = list(spec)
        except TypeError:
            entries = [spec]
        if any(e is not None and not callable(e) for e in entries):
            raise TypeError(f"`out_ufuncs` must be callable or sequence of callables. Got {spec}")
        if isinstance(spec, Sequence) and len(entries) > nparams:
            warnings.warn(f"len(out_ufuncs)={len(entries)}, but only {nparams} parameters. "
                          f"Extra ufuncs will be ignored.")
    return spec


def _checked_bounds(spec):
    """``out_bounds``: (lb, ub) for every parameter, or one (lb, ub) row per (leading) parameter."""
    table = np.asarray(spec)
    well_formed = table.ndim in (1, 2) and table.shape[-1] == 2
    if not well_formed:
        raise ValueError("Invalid `out_bounds` - shape must be ([num_params,] 2)")
    lower, upper = table[..., 0], table[..., 1]
    if (lower > upper).any():
        raise ValueError("Invalid `out_bounds` - lower bound must be <= upper bound")
    return table


def _resolved_r2_threshold(spec):
    """``r2_threshold``: None, a number, or the string "preferences" (-> preferences.fitting_r2_threshold)."""
    if not isinstance(spec, str):
        return spec
    if spec == "preferences":
        return preferences.fitting_r2_threshold
    raise ValueError(f"Invalid value r2_threshold='{spec}'. "
                     f"Expected `None`, a number between [0, 1], or 'preferences'.")


def _bounds_table(out_bounds, nparams):
    """(lb[P], ub[P]) from ``out_bounds``: a single pair applies to every parameter; missing rows are open."""
    lb = np.full(nparams, -np.inf)
    ub = np.full(nparams, np.inf)
    table = np.asarray(out_bounds, dtype=np.float64)
    if table.ndim == 1:
        lb[:], ub[:] = table[0], table[1]
    else:
        rows = min(nparams, table.shape[0])
        lb[:rows], ub[:rows] = table[:rows, 0], table[:rows, 1]
    return lb, ub


def _apply_post(popt, r_squared, out_ufuncs, out_bounds, r2_threshold, nan_to_num):
    """Reference order (fitting.py:109-146): ufuncs, bounds -> NaN, r2 threshold -> NaN rows, nan_to_num."""
    nparams = popt.shape[-1]
    with np.errstate(all="ignore"):
        if callable(out_ufuncs):
            popt = out_ufuncs(popt)
        elif out_ufuncs is not None:
            for j, fn in enumerate(list(out_ufuncs)[:nparams]):
                if fn is not None:
                    popt[..., j] = fn(popt[..., j])
        if out_bounds is not None:
            lb, ub = _bounds_table(out_bounds, nparams)
            outside = np.logical_or(popt < lb, popt > ub)
            popt[outside] = np.nan
        if r2_threshold is not None:
            popt[r_squared < r2_threshold] = np.nan
        if nan_to_num is not None:
            popt = np.nan_to_num(popt, nan=nan_to_num, copy=False)
    return popt


# scipy.optimize.leastsq options the kernels implement (everything else scipy accepts selects another solver)
_KERNEL_SOLVER_OPTIONS = ("xtol", "gtol", "factor")


def _solver_options(kwargs, who):
    """Split the reference's ``**kwargs`` (forwarded to scipy.optimize.curve_fit, fitting.py:827-830) into the
    MINPACK options the kernels take.  ``method="lm"`` is the default solver and accepted; anything else
    (``bounds=``, ``sigma=``, ``jac=``, ``method="trf"`` ...) is a different algorithm -> NotImplementedError (caught by
    :func:`_kernel_route`: such requests run the reference's per-voxel scipy loop)."""
    opts = {}
    rest = {}
    for k, v in kwargs.items():
        if k in _KERNEL_SOLVER_OPTIONS:
            opts[k] = float(v)
        elif k == "method" and v in (None, "lm"):
            continue
        else:
            rest[k] = v
    if rest:
        raise NotImplementedError(
            f"{who}(**{sorted(rest)}) selects a scipy solver other than the default unbounded Levenberg-Marquardt ('lm') "
            "configuration the kernels implement")
    return opts


def _kernel_route(func, kwargs, n_samples, who):
    """``(model, solver options)`` when libqmri_hip.so implements the request, else ``(None, reason)``: the caller then
    runs the reference's per-voxel scipy loop ("otherwise the scipy fallback (reference behaviour)")."""
    try:
        model = _model_of(func)
        solver = _solver_options(kwargs, who)
    except NotImplementedError as e:
        return None, str(e)
    limit = _lib.LM_MAX_SAMPLES.get(model)
    if limit is not None and n_samples > limit:
        return None, f"{n_samples} samples per voxel (the {model} kernels keep up to {limit})"
    return model, solver


def _scipy_loop(func, x, y, p0, *, given_p0, y_bounds, maxfev, ftol, eps, num_workers, chunksize, scipy_kwargs, why):
    """One scipy.optimize.curve_fit per column of ``y`` (E, N) under the reference's rules (dosma_amd/_scipy_loop.py)."""
    from . import _scipy_loop as loop

    warnings.warn(f"dosma_amd: {why} -> the reference's per-voxel scipy.optimize.curve_fit loop on the CPU "
                  f"({y.shape[1]} voxels)", RuntimeWarning, stacklevel=3)
    return loop.loop_fit(func, x, y, p0 if given_p0 else None, nparams=len(_func_param_names(func)), y_bounds=y_bounds,
                         maxfev=maxfev, ftol=ftol, eps=eps, num_workers=num_workers, chunksize=chunksize,
                         scipy_kwargs=scipy_kwargs)


class _Fitter:
    """Plumbing shared by the fitters: the post-processing options, the mask, and the wrapping of
    (N, P) / (N,) results into MedicalVolumes (reference :51-235)."""

    nan_to_num = None
    out_ufuncs = None
    out_bounds = None
    r2_threshold = None
    y_bounds = None

    def _set_post_options(self, nparams, *, y_bounds, out_ufuncs, out_bounds, r2_threshold, nan_to_num):
        self.y_bounds = y_bounds
        self.out_ufuncs = None if out_ufuncs is None else _checked_ufuncs(out_ufuncs, nparams)
        self.out_bounds = None if out_bounds is None else _checked_bounds(out_bounds)
        self.r2_threshold = _resolved_r2_threshold(r2_threshold)
        self.nan_to_num = nan_to_num

    def _process_mask(self, mask, y: MedicalVolume):
        """``mask`` -> boolean MedicalVolume in ``y``'s orientation (reference :95-107)."""
        if isinstance(mask, np.ndarray):
            mask = y._partial_clone(volume=mask, headers=None)
        if not isinstance(mask, MedicalVolume):
            raise TypeError("`mask` must be a MedicalVolume or ndarray")
        aligned = mask.reformat_as(y)
        if not aligned.is_same_dimensions(y, defaults.AFFINE_DECIMAL_PRECISION):
            raise RuntimeError("`mask` and `y` dimension mismatch")
        return aligned > 0

    def _process_params(self, x, r_squared):
        """Host post-processing for the general case (arbitrary Python ``out_ufuncs``).  The
        MonoExponentialFit recipe never comes here: it is fused into the kernel."""
        return _apply_post(x, r_squared, self.out_ufuncs, self.out_bounds, self.r2_threshold, self.nan_to_num)

    def _bounds_per_param(self, nparams):
        return _bounds_table(self.out_bounds, nparams)

    # ---- shared front half of fit(): checks + flatten to (E, N) ----
    def _prepare(self, x, y, mask, rows=False):
        if (not isinstance(y, (list, tuple))) or (not all(isinstance(_y, MedicalVolume) for _y in y)):
            raise TypeError("`y` must be sequence of MedicalVolumes.")
        x = np.asarray(x)
        if x.shape[-1] != len(y):
            raise ValueError(
                "Dimension mismatch: x.shape[-1]={:d}, but len(y)={:d}".format(x.shape[-1], len(y)))
        orientation = y[0].orientation
        y = [_y.reformat(orientation) for _y in y]
        mask_flat = None
        if mask is not None:
            mask = self._process_mask(mask, y[0])
            mask_flat = np.ascontiguousarray(mask.volume.reshape(-1))
        if rows:  # the flattened volumes themselves (views when contiguous): no (E, N) stacking copy
            return x, y, [_y.volume.reshape(-1) for _y in y], mask_flat
        svs = np.concatenate([_y.volume.reshape((1, -1)) for _y in y], axis=0)
        return x, y, svs, mask_flat

    def _wrap(self, y0: MedicalVolume, popt, r_squared, copy_headers):
        original_shape = y0.shape
        popt = popt.reshape(original_shape + popt.shape[-1:])
        r_squared = r_squared.reshape(original_shape)
        if copy_headers:
            headers = y0.headers()
            if headers is not None:
                headers = deepcopy(headers)
                headers = np.expand_dims(headers, axis=-1)
            popt_headers, r2_headers = headers, True
        else:
            popt_headers, r2_headers = None, None
        return (y0._partial_clone(volume=popt, headers=popt_headers),
                y0._partial_clone(volume=r_squared, headers=r2_headers))


class CurveFitter(_Fitter):
    """Non-linear least squares fit of ``func`` per voxel of co-registered MedicalVolumes.

    Same constructor and ``fit`` contract as the reference's ``CurveFitter`` (:238-458).  The mono-exponential and the
    bi-exponential model run on the GPU (see :func:`curve_fit`; ``num_workers`` / ``chunksize`` / ``verbose`` are then
    ignored); ``**kwargs`` are forwarded like the reference does (``maxfev``, ``ftol``, ``eps``, and the MINPACK options
    ``xtol`` / ``gtol`` / ``factor``).  Any other ``func``, or kwargs that select another scipy solver, run the reference's
    per-voxel scipy loop on the CPU (``_scipy_loop.py``).
    """

    def __init__(
        self,
        func: Callable,
        p0: Sequence[float] = None,
        y_bounds=None,
        out_ufuncs=None,
        out_bounds=None,
        r2_threshold="preferences",
        nan_to_num: float = None,
        num_workers: int = 0,
        chunksize: int = None,
        verbose: bool = False,
        **kwargs,
    ):
        self._func = func
        self._func_name = getattr(func, "__name__", type(func).__name__)
        self._func_nparams = len(_func_param_names(func))
        self._set_post_options(self._func_nparams, y_bounds=y_bounds, out_ufuncs=out_ufuncs, out_bounds=out_bounds,
                               r2_threshold=r2_threshold, nan_to_num=nan_to_num)
        self.p0 = self._format_p0(p0)
        self.num_workers, self.chunksize, self.verbose = num_workers, chunksize, verbose
        self.kwargs = kwargs

    def _format_p0(self, p0, ref: MedicalVolume = None, flatten: bool = False, depth: int = 0):
        """Per-parameter structure of scalars / flattened full-volume arrays (reference :344-380).

        Unlike the reference the arrays are NOT mask-selected here: the kernel indexes per-voxel
        guesses by voxel, and skips voxels outside the mask itself.
        """
        if p0 is None or isinstance(p0, Number):
            return p0
        if isinstance(p0, MedicalVolume) and depth > 0:
            if ref is not None:
                p0 = p0.reformat_as(ref)
                assert p0.is_same_dimensions(ref, err=True)
            return p0.A.flatten() if flatten else p0
        if isinstance(p0, np.ndarray) and depth > 0:
            if ref is not None and p0.shape != ref.shape:
                raise ValueError(f"Got p0.shape={p0.shape}, but y.shape={ref.shape}")
            return p0.flatten() if flatten else p0
        if isinstance(p0, Mapping):
            return {k: self._format_p0(v, ref, flatten, depth + 1) for k, v in p0.items()}
        if isinstance(p0, Sequence):
            return tuple(self._format_p0(v, ref, flatten, depth + 1) for v in p0)
        if isinstance(p0, (np.ndarray, MedicalVolume)):
            # a single array: the last axis is the parameter axis
            return tuple(self._format_p0(p0[..., i], ref, flatten, depth + 1)
                         for i in range(p0.shape[-1]))
        raise ValueError(f"p0={p0} not supported")

    def _fusable_post(self):
        """Post-processing block for the kernel if ``out_ufuncs`` is expressible there, else None."""
        uf = self.out_ufuncs
        inv = False
        if uf is not None:
            if isinstance(uf, Callable):
                return None
            uf = list(uf)[: self._func_nparams]
            if len(uf) > 0 and uf[0] is not None:
                return None
            if len(uf) > 1 and uf[1] is not None:
                if getattr(uf[1], "__qmri_ufunc__", None) != "inv_abs":
                    return None
                inv = True
        bounds = None
        if self.out_bounds is not None:
            lb, ub = self._bounds_per_param(2)
            bounds = ((lb[0], ub[0]), (lb[1], ub[1]))
        return dict(inv_abs_b=inv, bounds=bounds, r2_threshold=self.r2_threshold,
                    nan_to_num=self.nan_to_num)

    def fit(self, x, y: Sequence[MedicalVolume], mask=None, p0=np._NoValue, copy_headers: bool = True,
            _decimals=None, _tc_only=False):
        """Fit every voxel; returns ``(popt, r2)`` MedicalVolumes (``popt`` has a trailing parameter
        axis).  Voxels outside ``mask`` hold NaN (or ``nan_to_num``) like the reference (:205-215)."""
        # the reference forwards **kwargs to its module-level curve_fit (:422-435): maxfev / ftol / eps are that
        # function's own arguments, the rest go to scipy
        fit_kw = dict(self.kwargs)
        named = {k: fit_kw.pop(k) for k in ("maxfev", "ftol", "eps") if k in fit_kw}
        model, solver = _kernel_route(self._func, fit_kw, np.asarray(x).reshape(-1).shape[0] if not isinstance(x, MedicalVolume) else 0,
                                      "CurveFitter")
        if model is None:
            return self._fit_scipy_loop(x, y, mask, p0, copy_headers, named, fit_kw, solver, _decimals, _tc_only)
        if "maxfev" in named:
            solver["maxfev"] = int(named["maxfev"])
        if "ftol" in named:
            solver["ftol"] = float(named["ftol"])
        if "eps" in named:
            solver["r2_eps"] = float(named["eps"])
        if isinstance(x, MedicalVolume):
            raise RuntimeError("`x` must be on the CPU")
        x, y, svs, mask_flat = self._prepare(x, y, mask, rows=model == "monoexponential")
        N = svs[0].shape[0] if isinstance(svs, list) else svs.shape[1]
        if p0 is np._NoValue:
            p0 = self.p0
        p0 = self._format_p0(p0, ref=y[0], flatten=True)
        p0 = _format_p0(p0, _func_param_names(self._func), N)
        per_voxel = any(isinstance(v, np.ndarray) for v in p0)
        init = _lib.INIT_PER_VOXEL if per_voxel else _lib.INIT_SCALAR
        if getattr(self, "_loglin_init", False):
            init = _lib.INIT_LOGLIN

        if self.y_bounds is not None and any((r < self.y_bounds[0]).any() or (r > self.y_bounds[1]).any()
                                             for r in (svs if isinstance(svs, list) else [svs])):
            warnings.warn("Out of bounds values found. Failure in fit will result in np.nan")

        if model != "monoexponential":
            return self._fit_general(model, x, y, svs, mask_flat, p0, copy_headers, solver)
        if len(svs) > _lib.MAX_ECHOES:
            # more samples per voxel than the mono-exponential kernel keeps in registers: the general lmdif kernel (true
            # forward differences, lane-private LDS columns; up to 64 samples), everything around it as the reference does it
            return self._fit_many_samples(x, y, svs, mask, mask_flat, p0, copy_headers, solver, _decimals, _tc_only)

        post = self._fusable_post()
        if post is not None and _decimals is not None:
            post["decimals"] = _decimals
        tc_only = _tc_only and post is not None  # MonoExponentialFit: only (tc, r2) leave the GPU
        out = _lib.monoexp_fit_host(
            x.astype(np.float64).reshape(-1), _as_kernel_rows(svs), mask=mask_flat, init=init,
            p0=tuple(1.0 if isinstance(v, np.ndarray) else v for v in p0),
            a0v=p0[0] if isinstance(p0[0], np.ndarray) else None,
            b0v=p0[1] if isinstance(p0[1], np.ndarray) else None,
            post=post, want_tc=tc_only or (_decimals is not None and post is not None),
            y_bounds=self.y_bounds, want_popt=not tc_only, **solver,
        )
        if tc_only:
            shape = y[0].shape
            headers = deepcopy(y[0].headers()) if (copy_headers and y[0].headers() is not None) else None
            return (y[0]._partial_clone(volume=out["tc"].reshape(shape), headers=headers),
                    y[0]._partial_clone(volume=out["r2"].reshape(shape), headers=True if headers is not None else None))
        popt, r2 = out["popt"], out["r2"]
        if post is None:
            # arbitrary Python ufuncs: the reference's own post-processing, on the fitted rows only
            if mask_flat is None:
                popt = self._process_params(popt, r2)
            else:
                popt[mask_flat] = self._process_params(popt[mask_flat], r2[mask_flat])
                if self.nan_to_num is not None:
                    popt[~mask_flat] = self.nan_to_num
                    r2[~mask_flat] = self.nan_to_num
        popt_mv, r2_mv = self._wrap(y[0], popt, r2, copy_headers)
        if "tc" in out:
            self._last_tc = out["tc"]
        return popt_mv, r2_mv

    def _loglin_p0(self, x, y, mask):
        """tc0 = "polyfit" exactly as the reference builds it (:701-718): degree-1 PolyFitter on log(y + eps * (y == 0)),
        p0 = (exp(intercept), slope) per voxel (flattened, all voxels)."""
        vols = [sv.astype(np.float32) if np.issubdtype(sv.dtype, np.integer) else sv for sv in y]
        vols = [np.log(sv + 1e-10 * (sv == 0)) for sv in vols]
        params, _ = PolyFitter(1, r2_threshold=0, num_workers=None, nan_to_num=0.0).fit(x, vols, mask=mask, copy_headers=False)
        pv = params.volume.reshape(-1, 2)
        return [np.ascontiguousarray(np.exp(pv[:, 1]), dtype=np.float64), np.ascontiguousarray(pv[:, 0], dtype=np.float64)]

    def _tc_and_r2(self, y0, popt, r2, copy_headers, decimals):
        """What MonoExponentialFit.fit returns (:739-744) from the (popt, r2) pair of a route without the fused epilogue."""
        tc = popt.volume[..., 1]
        if decimals is not None:
            tc = np.around(tc, decimals)
        headers = deepcopy(y0.headers()) if (copy_headers and y0.headers() is not None) else None
        return (y0._partial_clone(volume=np.ascontiguousarray(tc), headers=headers),
                y0._partial_clone(volume=r2.volume, headers=True if headers is not None else None))

    def _fit_scipy_loop(self, x, y, mask, p0, copy_headers, named, scipy_kwargs, why, decimals=None, tc_only=False):
        """A request the kernels do not implement, as the reference runs it (:157-235, :422-435): gather the masked columns,
        one scipy.optimize.curve_fit per voxel, post-process, scatter back.  MonoExponentialFit's private requests (the
        log-linear initial guess, the rounded tc map as the only result) are honoured here as on the kernel routes."""
        if isinstance(x, MedicalVolume):
            raise RuntimeError("`x` must be on the CPU")
        mask_in = mask
        x, y, svs, mask_flat = self._prepare(x, y, mask)
        N = svs.shape[1]
        if p0 is np._NoValue:
            p0 = self.p0
        given_p0 = p0 is not None
        p0 = self._format_p0(p0, ref=y[0], flatten=True)
        p0 = _format_p0(p0, _func_param_names(self._func), N)
        if getattr(self, "_loglin_init", False):
            p0, given_p0 = self._loglin_p0(x, y, mask_in), True
        sel = None if mask_flat is None else np.flatnonzero(mask_flat)
        cols = svs if sel is None else svs[:, sel]
        p0 = [v[sel] if (sel is not None and isinstance(v, np.ndarray)) else v for v in p0]
        if self.y_bounds is not None and ((cols < self.y_bounds[0]).any() or (cols > self.y_bounds[1]).any()):
            warnings.warn("Out of bounds values found. Failure in fit will result in np.nan")
        popt_s, r2_s = _scipy_loop(self._func, x, cols, p0, given_p0=given_p0, y_bounds=self.y_bounds,
                                   maxfev=int(named.get("maxfev", 100)), ftol=float(named.get("ftol", 1e-5)),
                                   eps=float(named.get("eps", 1e-8)), num_workers=self.num_workers,
                                   chunksize=self.chunksize, scipy_kwargs=scipy_kwargs, why=why)
        popt_s = self._process_params(popt_s, r2_s)
        if sel is None:
            popt, r2 = popt_s, r2_s
        else:
            fill = np.nan if self.nan_to_num is None else self.nan_to_num
            popt = np.full((N, popt_s.shape[-1]), fill, dtype=np.float64)
            r2 = np.full(N, fill, dtype=np.float64)
            popt[sel] = popt_s
            r2[sel] = r2_s
        popt, r2 = self._wrap(y[0], popt, r2, copy_headers)
        return self._tc_and_r2(y[0], popt, r2, copy_headers, decimals) if tc_only else (popt, r2)

    def _fit_many_samples(self, x, y, rows, mask, mask_flat, p0, copy_headers, solver, decimals, tc_only):
        if getattr(self, "_loglin_init", False):
            p0 = self._loglin_p0(x, y, mask)
        svs = np.stack([np.asarray(r) for r in rows], axis=0)
        popt, r2 = self._fit_general("monoexponential", x, y, svs, mask_flat, p0, copy_headers, solver)
        return self._tc_and_r2(y[0], popt, r2, copy_headers, decimals) if tc_only else (popt, r2)

    def _fit_general(self, model, x, y, svs, mask_flat, p0, copy_headers, solver):
        """Models on the general lmdif kernel (bi-exponential): gather the masked columns like the
        reference (:199-200), fit, post-process on the host (:109-146), scatter back (:205-215)."""
        N = svs.shape[1]
        sel = None if mask_flat is None else np.flatnonzero(mask_flat)
        cols = svs if sel is None else np.ascontiguousarray(svs[:, sel])
        p0 = [v[sel] if (sel is not None and isinstance(v, np.ndarray)) else v for v in p0]
        out = _lib.lmfit_host(model, x.astype(np.float64).reshape(-1), _as_kernel_samples(cols), p0,
                              y_bounds=self.y_bounds, **solver)
        popt_s = self._process_params(out["popt"], out["r2"])
        r2_s = out["r2"]
        if sel is None:
            popt, r2 = popt_s, r2_s
        else:
            fill = np.nan if self.nan_to_num is None else self.nan_to_num
            popt = np.full((N, popt_s.shape[-1]), fill, dtype=np.float64)
            r2 = np.full(N, fill, dtype=np.float64)
            popt[sel] = popt_s
            r2[sel] = r2_s
        return self._wrap(y[0], popt, r2, copy_headers)

    def __str__(self) -> str:
        attrs = ["p0", "y_bounds", "out_bounds", "r2_threshold", "nan_to_num", "num_workers",
                 "chunksize", "verbose"]
        vals = [f"func={self._func_name}"]
        vals += [f"{k}={getattr(self, k)}" for k in attrs]
        vals += [f"{k}={v}" for k, v in self.kwargs.items()]
        return f"{self.__class__.__name__}(\n\t" + "\n\t".join(v + "," for v in vals) + "\n)"


class PolyFitter(_Fitter):
    """Linear least squares polynomial fit per voxel (reference :461-604), any degree, on the GPU (degrees above 7 or more
    than 32 samples take the general kernel's streaming variant)."""

    def __init__(self, deg: int, rcond: float = None, y_bounds=None, out_ufuncs=None,
                 out_bounds=None, r2_threshold="preferences", nan_to_num: float = None,
                 num_workers: int = None, chunksize: int = None, verbose: bool = False):
        self.deg, self.rcond = deg, rcond
        self._set_post_options(deg + 1, y_bounds=y_bounds, out_ufuncs=out_ufuncs, out_bounds=out_bounds,
                               r2_threshold=r2_threshold, nan_to_num=nan_to_num)
        self.num_workers, self.chunksize, self.verbose = num_workers, chunksize, verbose

    def fit(self, x, y: Sequence[MedicalVolume], mask=None, copy_headers: bool = True):
        x, y, svs, mask_flat = self._prepare(x, y, mask)
        sel = svs if mask_flat is None else np.ascontiguousarray(svs[:, mask_flat])
        popt, r2 = polyfit(x, sel, deg=self.deg, rcond=self.rcond, y_bounds=self.y_bounds,
                           num_workers=self.num_workers, chunksize=self.chunksize)
        popt = self._process_params(popt, r2)
        if mask_flat is not None:
            fill = np.nan if self.nan_to_num is None else self.nan_to_num
            popt_full = np.full((svs.shape[1], popt.shape[-1]), fill, dtype=np.float64)
            r2_full = np.full(svs.shape[1], fill, dtype=np.float64)
            popt_full[mask_flat] = popt
            r2_full[mask_flat] = r2
            popt, r2 = popt_full, r2_full
        return self._wrap(y[0], popt, r2, copy_headers)

    def __str__(self) -> str:
        attrs = ["deg", "rcond", "y_bounds", "out_bounds", "r2_threshold", "nan_to_num",
                 "num_workers", "chunksize", "verbose"]
        vals = [f"{k}={getattr(self, k)}" for k in attrs]
        return f"{self.__class__.__name__}(\n\t" + "\n\t".join(v + "," for v in vals) + "\n)"


class MonoExponentialFit:
    """Per-voxel time constant of ``y = a * exp(-x / tc)`` (T2, T1rho, T2* maps).

    Same constructor, ``fit`` contract and defaults as the reference's ``MonoExponentialFit``
    (:607-749).  The whole recipe -- optional log-linear initial guess (``tc0="polyfit"``, :701-718),
    the LM fit, ``tc = 1/|b|``, bounds, r2 threshold, ``nan_to_num(0)`` and rounding (:722-737) -- is ONE
    kernel launch.
    """

    def __init__(self, x: Sequence[float] = None, y: Sequence[MedicalVolume] = None,
                 mask: MedicalVolume = None, bounds=(0, 100.0), tc0=30.0,
                 r2_threshold="preferences", decimal_precision: int = 1, num_workers: int = 0,
                 chunksize: int = 1000, verbose: bool = False):
        # argument checks first (the reference interleaves them with the assignments, :646-676)
        if not (isinstance(tc0, Number) or tc0 == "polyfit"):
            raise ValueError("`tc0` must either be a float or the string 'polyfit'.")
        if len(bounds) != 2:
            raise ValueError("`bounds` should provide lower/upper bound in format (lb, ub)")
        for name, given in (("y", y), ("mask", mask)):
            if given is not None:
                warnings.warn(
                    f"Setting `{name}` in the constructor can result in significant memory overhead. "
                    f"Specify `{name}` in `{type(self).__name__}.fit({name}=...)` instead.")
        if y is not None:
            self._check_y(x, y)
        self.x, self.y, self.mask = x, y, mask
        self.bounds, self.tc0 = bounds, tc0
        self.r2_threshold, self.decimal_precision = r2_threshold, decimal_precision
        self.num_workers, self.chunksize, self.verbose = num_workers, chunksize, verbose
        self._eps = 1e-10  # epsilon for polyfit (reference :676) -- fixed in the kernel

    def fit(self, x=None, y: Sequence[MedicalVolume] = None, mask=None):
        """Returns ``(tc, r2)`` MedicalVolumes (float64), like the reference (:678-739)."""
        x = self.x if x is None else x
        y = self.y if y is None else y
        mask = self.mask if mask is None else mask
        self._check_y(x, y)
        orientation = y[0].orientation
        y = [sv.reformat(orientation) for sv in y]
        if isinstance(mask, np.ndarray):
            mask = MedicalVolume(mask, affine=y[0].affine)
        if mask is not None and not isinstance(mask, MedicalVolume):
            raise TypeError("`mask` must be a MedicalVolume")
        mask = mask.reformat(orientation) if mask is not None else None

        fitter = CurveFitter(
            monoexponential,
            y_bounds=None,
            out_ufuncs=(None, _inv_abs),
            out_bounds=((-np.inf, np.inf), self.bounds),
            r2_threshold=self.r2_threshold,
            num_workers=self.num_workers,
            chunksize=self.chunksize,
            verbose=self.verbose,
            nan_to_num=0.0,
        )
        if isinstance(self.tc0, str):
            fitter._loglin_init = True
            p0 = None
        else:
            p0 = {"a": 1.0, "b": -1 / self.tc0}
        decimals = self.decimal_precision
        # the kernel's fused epilogue (1/|b| -> bounds -> r2 threshold -> nan_to_num -> around) leaves exactly the two
        # maps the reference returns (:721-739); the (a, tc) pairs never cross PCIe
        tc_map, r_squared = fitter.fit(x, y, mask=mask, p0=p0, _decimals=decimals, _tc_only=True)
        return tc_map, r_squared

    def _check_y(self, x, y):
        if (not isinstance(y, Sequence)) or (not all(isinstance(sv, MedicalVolume) for sv in y)):
            raise TypeError("`y` must be list of MedicalVolumes.")
        if len(x) != len(y):
            raise ValueError("`len(x)`={:d}, but `len(y)`={:d}".format(len(x), len(y)))
