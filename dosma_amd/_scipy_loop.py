"""The requests the kernels do not implement, served the way the reference serves EVERY request: one
``scipy.optimize.curve_fit`` per voxel.

SURVEY 8(b)'s dispatch rule: the GPU path iff ``func`` is a model the kernels implement and no scipy ``**kwargs``
select another solver; "otherwise the scipy fallback (reference behaviour)".  This module is that "otherwise" --
a generic Python ``func``, ``bounds=`` (scipy then solves with ``trf`` and counts ``max_nfev``), ``sigma=``,
``jac=``, ``method=``, more samples per voxel than the kernels keep -- with the reference's per-voxel rules
(/root/reference/dosma/core/fitting.py:1026-1073): an all-zero or out-of-``y_bounds`` voxel is ``(nan, ...), 0``
without a solve, a ``RuntimeError`` of the solver is ``(nan, ...), 0``, and ``r2 = 1 - SSres / (SStot + eps)``.

It is NOT a CPU path for the mono- / bi-exponential models: those run in libqmri_hip.so or fail loudly
(tests/test_abi.py::test_no_cpu_fallback).  It is the only file of the package that imports scipy, and only when
such a request arrives (tests/test_abi.py::test_product_never_imports_the_oracle).
"""
import multiprocessing as mp
from functools import partial

import numpy as np

__all__ = ["loop_fit"]


def _one_voxel(item, *, func, x, nparams, y_bounds, ftol, eps, scipy_kwargs):
    from scipy import optimize as sop

    y, p0 = item
    nothing = (np.full(nparams, np.nan), 0.0)
    if y_bounds is not None and ((y < y_bounds[0]).any() or (y > y_bounds[1]).any()):
        return nothing
    if (y == 0).all():
        return nothing
    try:
        popt, _ = sop.curve_fit(func, x, y, p0=p0, ftol=ftol, **scipy_kwargs)
    except RuntimeError:
        return nothing
    res = y - func(x, *popt)
    ss_tot = np.sum((y - np.mean(y)) ** 2)
    return np.asarray(popt, dtype=np.float64), float(1 - np.sum(res ** 2) / (ss_tot + eps))


def loop_fit(func, x, y, p0_columns, *, nparams, y_bounds=None, maxfev=100, ftol=1e-5, eps=1e-8, num_workers=0,
             chunksize=None, scipy_kwargs=None):
    """``y`` (E, N); ``p0_columns``: None (scipy's default: ones) or one entry per parameter, a float or a length-N
    array.  Returns ``popt`` (N, nparams) float64, ``r2`` (N,) float64."""
    kw = dict(scipy_kwargs or {})
    # the reference's maxfev spelling (fitting.py:827-830): `maxfev` for MINPACK, `max_nfev` once `bounds=` selects trf
    if "bounds" not in kw:
        kw["maxfev"] = maxfev
    elif "max_nfev" not in kw:
        kw["max_nfev"] = maxfev
    x = np.asarray(x)
    y = np.asarray(y)
    N = y.shape[1]
    cols = y.T

    def p0_of(i):
        if p0_columns is None:
            return None
        return tuple(float(v[i]) if isinstance(v, np.ndarray) else v for v in p0_columns)

    one = partial(_one_voxel, func=func, x=x, nparams=nparams, y_bounds=y_bounds, ftol=ftol, eps=eps, scipy_kwargs=kw)
    items = ((cols[i], p0_of(i)) for i in range(N))
    workers = min(int(num_workers or 0), N)
    if workers > 0:
        with mp.Pool(workers) as pool:
            rows = pool.map(one, list(items), chunksize=chunksize)
    else:
        rows = [one(it) for it in items]
    popt = np.full((N, nparams), np.nan)
    r2 = np.zeros(N)
    for i, (p, r) in enumerate(rows):
        popt[i] = p
        r2[i] = r
    return popt, r2
