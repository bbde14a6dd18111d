"""ctypes binding of the C ABI in include/qmri.h (libqmri_hip.so).

This is the only place Python touches the native library.  There is NO CPU fallback: if the
library is missing or no HIP device is present, calls raise.
"""
import ctypes
import os
import sys
import threading

import numpy as np

from dosma_amd import _hostpool

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqmri_hip.so")

QMRI_OK = 0
QMRI_ERR_ARG = -1
QMRI_ERR_UNSUPPORTED = -2
QMRI_ERR_HIP = -3
QMRI_ERR_NONFINITE = -4
QMRI_ERR_NOMEM = -5

QMRI_F32, QMRI_F64, QMRI_I16, QMRI_U16 = 0, 1, 2, 3
INIT_SCALAR, INIT_PER_VOXEL, INIT_LOGLIN = 0, 1, 2
MAX_ECHOES = 32
# samples per voxel the general lmdif kernel keeps (include/qmri.h: E <= QMRI_LM_MAX_ECHOES and (n + 3) * E <= 320)
LM_MAX_SAMPLES = {"monoexponential": 64, "biexponential": 45}

_NP2Q = {np.dtype(np.float32): QMRI_F32, np.dtype(np.float64): QMRI_F64,
         np.dtype(np.int16): QMRI_I16, np.dtype(np.uint16): QMRI_U16}


class QmriPost(ctypes.Structure):
    _fields_ = [
        ("enable", ctypes.c_int32), ("inv_abs_b", ctypes.c_int32), ("use_bounds", ctypes.c_int32),
        ("use_r2_thr", ctypes.c_int32), ("use_nan_to_num", ctypes.c_int32),
        ("decimals", ctypes.c_int32),
        ("lb", ctypes.c_double * 2), ("ub", ctypes.c_double * 2),
        ("r2_threshold", ctypes.c_double), ("nan_value", ctypes.c_double),
    ]


class QmriMonoexpArgs(ctypes.Structure):
    _fields_ = [
        ("y", ctypes.c_void_p), ("y_dtype", ctypes.c_int32), ("E", ctypes.c_int32),
        ("N", ctypes.c_int64), ("ld", ctypes.c_int64),
        ("x", ctypes.POINTER(ctypes.c_double)), ("mask", ctypes.c_void_p),
        ("init", ctypes.c_int32), ("use_y_bounds", ctypes.c_int32),
        ("y_lo", ctypes.c_double), ("y_hi", ctypes.c_double),
        ("a0", ctypes.c_double), ("b0", ctypes.c_double),
        ("a0v", ctypes.c_void_p), ("b0v", ctypes.c_void_p),
        ("ftol", ctypes.c_double), ("xtol", ctypes.c_double), ("gtol", ctypes.c_double),
        ("factor", ctypes.c_double), ("r2_eps", ctypes.c_double),
        ("maxfev", ctypes.c_int32), ("reserved1", ctypes.c_int32),
        ("post", QmriPost),
        ("popt", ctypes.c_void_p), ("r2", ctypes.c_void_p), ("tc", ctypes.c_void_p),
        ("out_dtype", ctypes.c_int32), ("reserved2", ctypes.c_int32),
        ("info", ctypes.c_void_p), ("nfev", ctypes.c_void_p),
        ("device", ctypes.c_int32), ("reserved3", ctypes.c_int32),
        ("stream", ctypes.c_void_p),
        ("y_rows", ctypes.POINTER(ctypes.c_void_p)),
    ]


class QmriLinfitArgs(ctypes.Structure):
    _fields_ = [
        ("y", ctypes.c_void_p), ("y_dtype", ctypes.c_int32), ("E", ctypes.c_int32),
        ("N", ctypes.c_int64), ("ld", ctypes.c_int64),
        ("x", ctypes.POINTER(ctypes.c_double)),
        ("log_transform", ctypes.c_int32), ("skip_rules", ctypes.c_int32),
        ("use_y_bounds", ctypes.c_int32), ("out_dtype", ctypes.c_int32),
        ("y_lo", ctypes.c_double), ("y_hi", ctypes.c_double), ("r2_eps", ctypes.c_double),
        ("popt", ctypes.c_void_p), ("r2", ctypes.c_void_p),
        ("device", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("stream", ctypes.c_void_p),
    ]


class QmriPolylsArgs(ctypes.Structure):
    _fields_ = [
        ("y", ctypes.c_void_p), ("y_dtype", ctypes.c_int32), ("E", ctypes.c_int32),
        ("N", ctypes.c_int64), ("ld", ctypes.c_int64),
        ("P", ctypes.c_int32), ("skip_rules", ctypes.c_int32),
        ("solve", ctypes.POINTER(ctypes.c_double)), ("design", ctypes.POINTER(ctypes.c_double)),
        ("w", ctypes.POINTER(ctypes.c_double)),
        ("use_y_bounds", ctypes.c_int32), ("device", ctypes.c_int32),
        ("y_lo", ctypes.c_double), ("y_hi", ctypes.c_double), ("r2_eps", ctypes.c_double),
        ("popt", ctypes.c_void_p), ("r2", ctypes.c_void_p), ("resid", ctypes.c_void_p),
        ("stream", ctypes.c_void_p),
    ]


POLY_MAX_PARAMS = 8
MAX_ECHOES = 32


class QmriUnet2dDesc(ctypes.Structure):
    _fields_ = [
        ("depth", ctypes.c_int32), ("base_features", ctypes.c_int32), ("n_classes", ctypes.c_int32),
        ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("max_batch", ctypes.c_int32),
        ("precision", ctypes.c_int32), ("device", ctypes.c_int32),
        ("tensors", ctypes.POINTER(ctypes.c_void_p)), ("n_tensors", ctypes.c_int32),
        ("reserved", ctypes.c_int32), ("bn_eps", ctypes.c_double),
    ]


class QmriDessArgs(ctypes.Structure):
    _fields_ = [
        ("echo1", ctypes.c_void_p), ("echo2", ctypes.c_void_p),
        ("dtype", ctypes.c_int32), ("out_dtype", ctypes.c_int32), ("N", ctypes.c_int64),
        ("c0", ctypes.c_double), ("k", ctypes.c_double), ("c1", ctypes.c_double),
        ("use_bounds", ctypes.c_int32), ("use_nan_to_num", ctypes.c_int32),
        ("lo", ctypes.c_double), ("hi", ctypes.c_double), ("nan_value", ctypes.c_double),
        ("decimals", ctypes.c_int32), ("suppress_fat", ctypes.c_int32),
        ("suppress_fluid", ctypes.c_int32), ("device", ctypes.c_int32),
        ("beta", ctypes.c_double), ("t2", ctypes.c_void_p), ("stream", ctypes.c_void_p),
    ]


class QmriRegionStatsArgs(ctypes.Structure):
    _fields_ = [
        ("values", ctypes.c_void_p), ("v_dtype", ctypes.c_int32), ("labels", ctypes.c_void_p), ("N", ctypes.c_int64),
        ("nkeys", ctypes.c_int32), ("l_kind", ctypes.c_int32), ("label_keys", ctypes.c_void_p),
        ("use_bounds", ctypes.c_int32),
        ("lo", ctypes.c_double), ("hi", ctypes.c_double), ("closed", ctypes.c_int32), ("out", ctypes.c_void_p),
        ("device", ctypes.c_int32),
    ]


MAX_REGIONS = 16  # QMRI_MAX_REGIONS


class QmriLmfitArgs(ctypes.Structure):
    _fields_ = [
        ("model", ctypes.c_int32), ("y_dtype", ctypes.c_int32), ("y", ctypes.c_void_p),
        ("E", ctypes.c_int32), ("maxfev", ctypes.c_int32), ("N", ctypes.c_int64), ("ld", ctypes.c_int64),
        ("x", ctypes.POINTER(ctypes.c_double)),
        ("p0", ctypes.c_double * 4), ("p0v", ctypes.c_void_p * 4),
        ("ftol", ctypes.c_double), ("xtol", ctypes.c_double), ("gtol", ctypes.c_double),
        ("factor", ctypes.c_double), ("epsfcn", ctypes.c_double), ("r2_eps", ctypes.c_double),
        ("use_y_bounds", ctypes.c_int32), ("device", ctypes.c_int32),
        ("y_lo", ctypes.c_double), ("y_hi", ctypes.c_double),
        ("popt", ctypes.c_void_p), ("r2", ctypes.c_void_p), ("info", ctypes.c_void_p),
        ("nfev", ctypes.c_void_p), ("stream", ctypes.c_void_p),
    ]


MODELS = {"monoexponential": 0, "biexponential": 1}
MODEL_NPARAMS = {"monoexponential": 2, "biexponential": 4}
# "fp16x3" is the parity mode (logits within 1e-3 of an fp64 run); "fp16x3-general" forces it onto the general
# convolution kernel in the operator-level entry (tests); "bf16" is the single-MFMA throughput mode
PRECISION = {"bf16": 0, "fp16x3": 1, "fp16x3-general": 2, "bf16-s3": 3, "fp16x3-c4": 4, "fp16x3-s3": 5, "fp16x3-c4-nosplit": 6}  # "bf16-s3": plain bf16 forced onto conv_s3_kernel (operator entry, tests)

EXPORTS = (
    "qmri_version", "qmri_device_count", "qmri_last_error", "qmri_monoexp_defaults",
    "qmri_monoexp_fit_device", "qmri_monoexp_fit_host", "qmri_set_timing", "qmri_last_kernel_ms",
    "qmri_monoexp_kernel_name", "qmri_linfit_device", "qmri_linfit_host", "qmri_polyls_device", "qmri_polyls_host",
    "qmri_unet2d_create", "qmri_unet2d_set_precision", "qmri_unet2d_trace", "qmri_unet2d_forward", "qmri_unet2d_destroy",
    "qmri_unet2d_segment_volume",
    "qmri_conv2d_nhwc_host", "qmri_dess_t2_device", "qmri_dess_t2_host", "qmri_rss_host",
    "qmri_lmfit_defaults", "qmri_lmfit_device", "qmri_lmfit_host", "qmri_region_stats_host",
    "qmri_region_stats_device", "qmri_host_alloc", "qmri_host_free", "qmri_device_mem_info", "qmri_selftest_fp64",
)

_lib = None
_lock = threading.Lock()


class QmriError(RuntimeError):
    pass


class QmriOutOfMemory(QmriError, MemoryError):
    """QMRI_ERR_NOMEM: the device allocation failed; a smaller batch may fit."""


def library_path() -> str:
    return _SO


# DOSMA_AMD_LIB=<path>: load another build of the SAME library (e.g. one compiled with -DQMRI_S3_EXPERIMENTS for the timing
# experiments of scripts/s3_experiments.sh).  Still the HIP library -- there is no other implementation to point this at.
if os.environ.get("DOSMA_AMD_LIB"):
    _SO = os.path.abspath(os.environ["DOSMA_AMD_LIB"])


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``) and load it
    by file name; libqmri_hip.so asks for the SONAME.  If ours pulled in /opt/rocm's copy first, a
    later ``import torch`` would bring a SECOND runtime into the process and see no GPU.  Loading
    torch's copy first (cheap: no ``import torch``) makes both resolve to the same runtime, so
    device pointers and streams can be shared (bench.py, multi-GPU driver).
    """
    import importlib.util
    import sys

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            if "torch" in sys.modules:
                raise


def load():
    """Load libqmri_hip.so (raises if it has not been built: ``python -m dosma_amd.build``)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_SO) and os.environ.get("DOSMA_AMD_AUTOBUILD", "1") != "0":
            # a fresh checkout on a box with the ROCm toolchain: build the extension in-tree (about 30 s, once).  This is
            # still the HIP path -- there is nothing else to fall back to -- and it fails below if hipcc is missing.
            try:
                from . import build as _build
                _build.build()
            except Exception as err:  # noqa: BLE001 - reported through the QmriError below
                sys.stderr.write(f"dosma_amd: building {_SO} failed: {err}\n")
        if not os.path.exists(_SO):
            raise QmriError(
                f"{_SO} not found: the HIP extension is not built (run `python -m dosma_amd.build`). "
                "dosma_amd has no CPU fallback.")
        _share_hip_runtime_with_torch()
        lib = ctypes.CDLL(_SO)
        lib.qmri_version.restype = ctypes.c_int
        lib.qmri_host_alloc.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
        lib.qmri_host_alloc.restype = ctypes.c_int
        lib.qmri_host_free.argtypes = [ctypes.c_void_p]
        lib.qmri_host_free.restype = ctypes.c_int
        lib.qmri_device_count.restype = ctypes.c_int
        lib.qmri_device_mem_info.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        lib.qmri_device_mem_info.restype = ctypes.c_int
        dp = ctypes.POINTER(ctypes.c_double)
        lib.qmri_selftest_fp64.argtypes = [ctypes.c_int32, dp, ctypes.c_int64, dp, dp, dp, dp]
        lib.qmri_selftest_fp64.restype = ctypes.c_int
        lib.qmri_last_error.restype = ctypes.c_char_p
        lib.qmri_monoexp_kernel_name.restype = ctypes.c_char_p
        lib.qmri_monoexp_kernel_name.argtypes = [ctypes.POINTER(QmriMonoexpArgs)]
        lib.qmri_monoexp_defaults.argtypes = [ctypes.POINTER(QmriMonoexpArgs)]
        lib.qmri_monoexp_defaults.restype = None
        lib.qmri_monoexp_fit_device.argtypes = [ctypes.POINTER(QmriMonoexpArgs), ctypes.c_void_p]
        lib.qmri_monoexp_fit_device.restype = ctypes.c_int
        lib.qmri_monoexp_fit_host.argtypes = [ctypes.POINTER(QmriMonoexpArgs)]
        lib.qmri_monoexp_fit_host.restype = ctypes.c_int
        lib.qmri_set_timing.argtypes = [ctypes.c_int]
        lib.qmri_set_timing.restype = None
        lib.qmri_last_kernel_ms.restype = ctypes.c_float
        for name in ("qmri_linfit_device", "qmri_linfit_host"):
            fn = getattr(lib, name)
            fn.argtypes = [ctypes.POINTER(QmriLinfitArgs)]
            fn.restype = ctypes.c_int
        for name in ("qmri_polyls_device", "qmri_polyls_host"):
            fn = getattr(lib, name)
            fn.argtypes = [ctypes.POINTER(QmriPolylsArgs)]
            fn.restype = ctypes.c_int
        lib.qmri_unet2d_create.argtypes = [ctypes.POINTER(QmriUnet2dDesc), ctypes.POINTER(ctypes.c_void_p)]
        lib.qmri_unet2d_create.restype = ctypes.c_int
        lib.qmri_unet2d_set_precision.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        lib.qmri_unet2d_set_precision.restype = ctypes.c_int
        lib.qmri_unet2d_trace.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32]
        lib.qmri_unet2d_trace.restype = ctypes.c_int
        lib.qmri_unet2d_forward.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
            ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        lib.qmri_unet2d_forward.restype = ctypes.c_int
        lib.qmri_unet2d_segment_volume.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        lib.qmri_unet2d_segment_volume.restype = ctypes.c_int
        lib.qmri_unet2d_destroy.argtypes = [ctypes.c_void_p]
        lib.qmri_unet2d_destroy.restype = ctypes.c_int
        lib.qmri_conv2d_nhwc_host.argtypes = [
            ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]
        lib.qmri_conv2d_nhwc_host.restype = ctypes.c_int
        for name in ("qmri_dess_t2_device", "qmri_dess_t2_host"):
            fn = getattr(lib, name)
            fn.argtypes = [ctypes.POINTER(QmriDessArgs)]
            fn.restype = ctypes.c_int
        lib.qmri_rss_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                      ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]
        lib.qmri_rss_host.restype = ctypes.c_int
        lib.qmri_lmfit_defaults.argtypes = [ctypes.POINTER(QmriLmfitArgs)]
        lib.qmri_lmfit_defaults.restype = None
        lib.qmri_lmfit_device.argtypes = [ctypes.POINTER(QmriLmfitArgs), ctypes.c_void_p]
        lib.qmri_lmfit_device.restype = ctypes.c_int
        lib.qmri_lmfit_host.argtypes = [ctypes.POINTER(QmriLmfitArgs)]
        lib.qmri_lmfit_host.restype = ctypes.c_int
        _lib = lib
        return lib


def check(rc: int):
    if rc == QMRI_OK:
        return
    msg = load().qmri_last_error().decode("utf-8", "replace")
    if rc in (QMRI_ERR_ARG, QMRI_ERR_NONFINITE):
        raise ValueError(msg)
    if rc == QMRI_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == QMRI_ERR_NOMEM:
        raise QmriOutOfMemory(msg)
    raise QmriError(msg)


_default_device = int(os.environ.get("DOSMA_AMD_DEVICE", "0"))


def set_default_device(index: int):
    """HIP device used by every host-level call of this process that does not name one (one process per GPU:
    ``dosma_amd.dist.init`` sets it to the local rank).  Also settable with ``DOSMA_AMD_DEVICE``."""
    global _default_device
    _default_device = int(index)


def default_device() -> int:
    return _default_device


def _dev(device):
    return _default_device if device is None else int(device)


def require_device() -> int:
    n = load().qmri_device_count()
    if n <= 0:
        raise QmriError("no HIP device visible: dosma_amd computes on the GPU only (no CPU fallback)")
    return n


def device_mem_info(device=None):
    """(free, total) bytes of device memory."""
    f, t = ctypes.c_uint64(), ctypes.c_uint64()
    check(load().qmri_device_mem_info(_dev(device), ctypes.byref(f), ctypes.byref(t)))
    return int(f.value), int(t.value)


def selftest_fp64(x, device=None):
    """exp_sk / exp and log_sk / log of the device for the float64 values x -> dict of four arrays (see include/qmri.h)."""
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    out = {k: np.empty_like(x) for k in ("exp_sk", "exp_lib", "log_sk", "log_lib")}
    ptr = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))  # noqa: E731
    check(load().qmri_selftest_fp64(_dev(device), ptr(x), x.size, ptr(out["exp_sk"]), ptr(out["exp_lib"]),
                                    ptr(out["log_sk"]), ptr(out["log_lib"])))
    return out


def default_args() -> QmriMonoexpArgs:
    a = QmriMonoexpArgs()
    load().qmri_monoexp_defaults(ctypes.byref(a))
    return a


def qdtype(dt) -> int:
    dt = np.dtype(dt)
    if dt not in _NP2Q:
        raise ValueError(f"unsupported sample dtype {dt}")
    return _NP2Q[dt]


def _ptr(arr):
    return None if arr is None else ctypes.c_void_p(arr.ctypes.data)


def set_post(a: QmriMonoexpArgs, inv_abs_b=False, bounds=None, r2_threshold=None, nan_to_num=None,
             decimals=None):
    """Fill the fused post-processing block (``_Fitter._process_params`` semantics)."""
    p = a.post
    p.enable = 1
    p.inv_abs_b = 1 if inv_abs_b else 0
    p.use_bounds = 0
    if bounds is not None:
        p.use_bounds = 1
        for j in range(2):
            p.lb[j] = float(bounds[j][0])
            p.ub[j] = float(bounds[j][1])
    p.use_r2_thr = 0 if r2_threshold is None else 1
    p.r2_threshold = 0.0 if r2_threshold is None else float(r2_threshold)
    p.use_nan_to_num = 0 if nan_to_num is None else 1
    p.nan_value = 0.0 if nan_to_num is None else float(nan_to_num)
    p.decimals = -1000000 if decimals is None else int(decimals)


def monoexp_fit_host(x, y, *, mask=None, init=INIT_SCALAR, p0=(1.0, 1.0), a0v=None, b0v=None,
                     post=None, want_tc=False, want_info=False, out_dtype=np.float64, device=None,
                     ftol=None, maxfev=None, r2_eps=None, y_bounds=None, out=None, want_popt=True,
                     xtol=None, gtol=None, factor=None):
    """Run the HIP fit on host (numpy) buffers.  ``y``: (E, N) C-contiguous, echo-major.

    Returns dict(popt (N,2), r2 (N,), [tc (N,)], [info (N,) int8, nfev (N,) int16]).
    """
    lib = load()
    require_device()
    x = np.ascontiguousarray(x, dtype=np.float64)
    a = default_args()
    if isinstance(y, (list, tuple)):
        # one contiguous 1-D array per echo (the flattened MedicalVolumes): fitted in place, no (E, N) stacking copy
        rows = [np.ascontiguousarray(r).reshape(-1) for r in y]
        E, N = len(rows), rows[0].shape[0]
        if any(r.shape[0] != N or r.dtype != rows[0].dtype for r in rows):
            raise ValueError("echo rows must have equal length and dtype")
        row_ptrs = (ctypes.c_void_p * E)(*[r.ctypes.data for r in rows])
        a.y_rows = ctypes.cast(row_ptrs, ctypes.POINTER(ctypes.c_void_p))
        a.y_dtype = qdtype(rows[0].dtype)
        keep = [x, rows, row_ptrs]
    else:
        y = np.asarray(y)
        if y.ndim != 2:
            raise ValueError("y must be (E, N)")
        if not y.flags.c_contiguous:
            y = np.ascontiguousarray(y)
        E, N = y.shape
        a.y = _ptr(y)
        a.y_dtype = qdtype(y.dtype)
        keep = [x, y]
    if x.shape != (E,):
        raise ValueError(f"x has shape {x.shape}, expected ({E},)")
    a.E, a.N, a.ld = E, N, N
    a.x = x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
        if mask.shape[0] != N:
            raise ValueError("mask length mismatch")
        a.mask = _ptr(mask)
        keep.append(mask)
    a.init = init
    a.a0, a.b0 = float(p0[0]), float(p0[1])
    for name, arr in (("a0v", a0v), ("b0v", b0v)):
        if arr is not None:
            arr = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
            if arr.shape[0] != N:
                raise ValueError(f"Got {arr.shape[0]} values for {name}. Expected {N}")
            setattr(a, name, _ptr(arr))
            keep.append(arr)
    if ftol is not None:
        a.ftol = float(ftol)
    if maxfev is not None:
        a.maxfev = int(maxfev)
    if r2_eps is not None:
        a.r2_eps = float(r2_eps)
    for opt, val in (("xtol", xtol), ("gtol", gtol), ("factor", factor)):  # MINPACK options (scipy leastsq names)
        if val is not None:
            setattr(a, opt, float(val))
    if y_bounds is not None:
        a.use_y_bounds = 1
        a.y_lo, a.y_hi = float(y_bounds[0]), float(y_bounds[1])
    if post is not None:
        set_post(a, **post)
    od = np.dtype(out_dtype)
    a.out_dtype = QMRI_F64 if od == np.float64 else QMRI_F32
    if not want_popt and not want_tc:
        raise ValueError("nothing to return: want_popt=False needs want_tc=True")
    if out is None:  # page-locked, recycled blocks for large results (no first-touch zeroing, true DMA): _hostpool.py
        out = {"r2": _hostpool.empty(N, od)}
        if want_popt:
            out["popt"] = _hostpool.empty((N, 2), od)
        if want_tc:
            out["tc"] = _hostpool.empty(N, od)
    else:  # reuse of a previous call's result arrays (no first-touch cost)
        need = {"r2": (N,)}
        if want_popt:
            need["popt"] = (N, 2)
        if want_tc:
            need["tc"] = (N,)
        for k, shp in need.items():
            if k not in out or out[k].shape != shp or out[k].dtype != od or not out[k].flags.c_contiguous:
                raise ValueError(f"`out[{k!r}]` does not match the request")
    a.r2 = _ptr(out["r2"])
    if want_popt:
        a.popt = _ptr(out["popt"])
    if want_tc:
        a.tc = _ptr(out["tc"])
    if want_info:
        out["info"] = _hostpool.empty(N, np.int8)
        out["nfev"] = _hostpool.empty(N, np.int16)
        a.info, a.nfev = _ptr(out["info"]), _ptr(out["nfev"])
    a.device = _dev(device)
    check(lib.qmri_monoexp_fit_host(ctypes.byref(a)))
    del keep
    return out


def linfit_host(x, y, *, log_transform=False, per_sequence_rules=False, y_bounds=None, r2_eps=1e-8,
                out_dtype=np.float64, device=None):
    """Degree-1 least squares per column of ``y`` (E, N) on the GPU -> dict(popt (N,2), r2 (N,))."""
    lib = load()
    require_device()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y)
    E, N = y.shape
    if x.shape != (E,):
        raise ValueError(f"x has shape {x.shape}, expected ({E},)")
    a = QmriLinfitArgs()
    a.y, a.y_dtype, a.E, a.N, a.ld = _ptr(y), qdtype(y.dtype), E, N, N
    a.x = x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    a.log_transform = 1 if log_transform else 0
    a.skip_rules = 1 if per_sequence_rules else 0
    a.y_lo, a.y_hi = -np.inf, np.inf
    if y_bounds is not None:
        a.use_y_bounds = 1
        a.y_lo, a.y_hi = float(y_bounds[0]), float(y_bounds[1])
    a.r2_eps = float(r2_eps)
    od = np.dtype(out_dtype)
    a.out_dtype = QMRI_F64 if od == np.float64 else QMRI_F32
    out = {"popt": _hostpool.empty((N, 2), od), "r2": _hostpool.empty(N, od)}
    a.popt, a.r2 = _ptr(out["popt"]), _ptr(out["r2"])
    a.device = _dev(device)
    check(lib.qmri_linfit_host(ctypes.byref(a)))
    return out


def polyls_host(y, solve, design, *, w=None, per_sequence_rules=False, y_bounds=None, r2_eps=1e-8, want_resid=False,
                device=None):
    """General polynomial least squares per column of ``y`` (E, N) on the GPU: ``popt = solve @ y`` (P, E), fitted values
    ``design @ popt`` for r2, weighted residual sum of squares.  Returns dict(popt (N, P), r2 (N,), [resid (N,)])."""
    lib = load()
    require_device()
    y = np.ascontiguousarray(y)
    E, N = y.shape
    solve = np.ascontiguousarray(solve, dtype=np.float64)
    design = np.ascontiguousarray(design, dtype=np.float64)
    P = solve.shape[0]
    if solve.shape != (P, E) or design.shape != (E, P):
        raise ValueError(f"solve must be (P, {E}) and design ({E}, P)")
    # (deg + 1 <= POLY_MAX_PARAMS and E <= MAX_ECHOES run on the register-resident kernel, anything larger on its
    #  streaming variant: numpy.polyfit has no such limits and neither has this entry)
    a = QmriPolylsArgs()
    a.y, a.y_dtype, a.E, a.N, a.ld, a.P = _ptr(y), qdtype(y.dtype), E, N, N, P
    dp = ctypes.POINTER(ctypes.c_double)
    a.solve, a.design = solve.ctypes.data_as(dp), design.ctypes.data_as(dp)
    keep = [solve, design]
    if w is not None:
        w = np.ascontiguousarray(w, dtype=np.float64).reshape(-1)
        if w.shape[0] != E:
            raise TypeError("expected x and w to have same length")
        a.w = w.ctypes.data_as(dp)
        keep.append(w)
    a.skip_rules = 1 if per_sequence_rules else 0
    a.y_lo, a.y_hi = -np.inf, np.inf
    if y_bounds is not None:
        a.use_y_bounds = 1
        a.y_lo, a.y_hi = float(y_bounds[0]), float(y_bounds[1])
    a.r2_eps = float(r2_eps)
    out = {"popt": _hostpool.empty((N, P), np.float64), "r2": _hostpool.empty(N, np.float64)}
    a.popt, a.r2 = _ptr(out["popt"]), _ptr(out["r2"])
    if want_resid:
        out["resid"] = _hostpool.empty(N, np.float64)
        a.resid = _ptr(out["resid"])
    a.device = _dev(device)
    check(lib.qmri_polyls_host(ctypes.byref(a)))
    del keep
    return out


def lmfit_host(model, x, y, p0, *, ftol=None, maxfev=None, r2_eps=None, y_bounds=None, want_info=False,
               device=None, xtol=None, gtol=None, factor=None):
    """General lmdif (true forward differences) on the GPU.  ``model``: "biexponential" | "monoexponential";
    ``y`` (E, N) echo-major; ``p0``: one entry per parameter, a float or a float64 array of length N.
    Returns dict(popt (N, n), r2 (N,), [info, nfev])."""
    lib = load()
    require_device()
    n = MODEL_NPARAMS[model]
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y)
    E, N = y.shape
    if x.shape != (E,):
        raise ValueError(f"x has shape {x.shape}, expected ({E},)")
    if len(p0) != n:
        raise ValueError(f"`p0` has length {len(p0)} but the model has {n} parameters")
    a = QmriLmfitArgs()
    lib.qmri_lmfit_defaults(ctypes.byref(a))
    a.model = MODELS[model]
    a.y, a.y_dtype, a.E, a.N, a.ld = _ptr(y), qdtype(y.dtype), E, N, N
    a.x = x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    keep = []
    for j, v in enumerate(p0):
        if isinstance(v, np.ndarray):
            v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1)
            if v.shape[0] != N:
                raise ValueError(f"per-voxel p0[{j}] has {v.shape[0]} entries, expected {N}")
            keep.append(v)
            a.p0v[j] = v.ctypes.data
        else:
            a.p0[j] = float(v)
    if ftol is not None:
        a.ftol = float(ftol)
    if maxfev is not None:
        a.maxfev = int(maxfev)
    if r2_eps is not None:
        a.r2_eps = float(r2_eps)
    for opt, val in (("xtol", xtol), ("gtol", gtol), ("factor", factor)):
        if val is not None:
            setattr(a, opt, float(val))
    if y_bounds is not None:
        a.use_y_bounds = 1
        a.y_lo, a.y_hi = float(y_bounds[0]), float(y_bounds[1])
    out = {"popt": _hostpool.empty((N, n), np.float64), "r2": _hostpool.empty(N, np.float64)}
    a.popt, a.r2 = _ptr(out["popt"]), _ptr(out["r2"])
    if want_info:
        out["info"] = _hostpool.empty(N, np.int8)
        out["nfev"] = _hostpool.empty(N, np.int16)
        a.info, a.nfev = _ptr(out["info"]), _ptr(out["nfev"])
    a.device = _dev(device)
    check(lib.qmri_lmfit_host(ctypes.byref(a)))
    del keep
    return out


def conv2d_nhwc_host(x, kernel, bias, *, scale=None, shift=None, relu=True, transposed=False,
                     precision="fp16x3", device=None):
    """One conv layer on the GPU from host arrays: x (B,H,W,Cin) f32; kernel in the Keras layout --
    (3,3,Cin,Cout) for Conv2D, (3,3,Cout,Cin) for Conv2DTranspose(strides=2).  Returns NHWC f32."""
    lib = load()
    require_device()
    x = np.ascontiguousarray(x, dtype=np.float32)
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    B, H, W, Cin = x.shape
    Cout = kernel.shape[2] if transposed else kernel.shape[3]
    if kernel.shape != ((3, 3, Cout, Cin) if transposed else (3, 3, Cin, Cout)):
        raise ValueError(f"kernel shape {kernel.shape} does not match the input channels {Cin}")
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
    y = np.empty((B, 2 * H, 2 * W, Cout) if transposed else (B, H, W, Cout), dtype=np.float32)
    check(lib.qmri_conv2d_nhwc_host(_ptr(x), B, H, W, Cin, _ptr(kernel), _ptr(bias), _ptr(sc), _ptr(sh),
                                    1 if relu else 0, Cout, 1 if transposed else 0, PRECISION[precision],
                                    _ptr(y), _dev(device)))
    return y


class Unet2dEngine:
    """Owns a native U-Net instance (weights packed on the GPU + activation buffers)."""

    def __init__(self, tensors, H, W, *, depth=6, base_features=32, n_classes=4, max_batch=16,
                 precision="fp16x3", device=None, bn_eps=1e-3):
        lib = load()
        require_device()
        self._lib = lib
        self._handle = ctypes.c_void_p()
        keep = [np.ascontiguousarray(t, dtype=np.float32) for t in tensors]
        arr = (ctypes.c_void_p * len(keep))(*[t.ctypes.data for t in keep])
        d = QmriUnet2dDesc()
        d.depth, d.base_features, d.n_classes = depth, base_features, n_classes
        d.H, d.W, d.max_batch = int(H), int(W), int(max_batch)
        d.precision, d.device = PRECISION[precision], _dev(device)
        d.tensors = ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p))
        d.n_tensors = len(keep)
        d.bn_eps = float(bn_eps)
        check(lib.qmri_unet2d_create(ctypes.byref(d), ctypes.byref(self._handle)))
        self.H, self.W, self.n_classes, self.max_batch = int(H), int(W), int(n_classes), int(max_batch)

    def set_precision(self, precision):
        if precision in ("fp16x3-general", "bf16-s3"):
            raise ValueError("the engine picks its kernels itself: 'fp16x3' or 'bf16'")
        check(self._lib.qmri_unet2d_set_precision(self._handle, PRECISION[precision]))

    def trace(self):
        """Kernel family of every layer of the last forward batch: ["down0.conv1:c1/split", "down0.conv2:s3/2d/bn32", ...]."""
        buf = ctypes.create_string_buffer(8192)
        self._lib.qmri_unet2d_trace(self._handle, buf, len(buf))
        return [t for t in buf.value.decode().split(";") if t]

    def forward_host(self, x, *, whiten=False, eps=0.0, want_logits=True, want_mask=True):
        """x (S, H, W) float32 host array -> (logits (S,H,W,C) f32 | None, mask (S,H,W,C) u8 | None)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        S = x.shape[0]
        if x.shape[1:] != (self.H, self.W):
            raise ValueError(f"slices are {x.shape[1:]}, model was built for {(self.H, self.W)}")
        logits = _hostpool.empty((S, self.H, self.W, self.n_classes), np.float32) if want_logits else None
        mask = _hostpool.empty((S, self.H, self.W, self.n_classes), np.uint8) if want_mask else None
        check(self._lib.qmri_unet2d_forward(self._handle, _ptr(x), S, 0, 1 if whiten else 0, float(eps),
                                            _ptr(logits), _ptr(mask), 0, None))
        return logits, mask

    def segment_volume(self, vol_hws, *, whiten=False, eps=0.0):
        """(H, W, S) float32 volume as MedicalVolume holds it (sagittal) -> (C, H, W, S) uint8 masks, one
        contiguous (H, W, S) volume per class; the (S, H, W) <-> (H, W, S) transposes of the reference's
        generate_mask run on the GPU (one upload, one download)."""
        v = np.ascontiguousarray(vol_hws, dtype=np.float32)
        if v.ndim != 3 or v.shape[:2] != (self.H, self.W):
            raise ValueError(f"volume is {v.shape}, model was built for slices of {(self.H, self.W)}")
        S = v.shape[2]
        out = _hostpool.empty((self.n_classes, self.H, self.W, S), np.uint8)
        check(self._lib.qmri_unet2d_segment_volume(self._handle, _ptr(v), S, 1 if whiten else 0, float(eps),
                                                   _ptr(out), None))
        return out

    def forward_device(self, x_ptr, S, logits_ptr, mask_ptr, *, whiten=False, eps=0.0, stream=None):
        check(self._lib.qmri_unet2d_forward(self._handle, x_ptr, int(S), 1, 1 if whiten else 0, float(eps),
                                            logits_ptr, mask_ptr, 1, stream))

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.qmri_unet2d_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dess_t2_host(echo1, echo2, c0, k, c1, *, bounds=None, nan_to_num=None, decimals=None,
                 suppress_fat=False, suppress_fluid=False, beta=1.2, out_dtype=np.float64, device=None):
    """Analytic DESS T2 map on the GPU from two host echo arrays of equal shape -> array of that shape."""
    lib = load()
    require_device()
    e1 = np.ascontiguousarray(echo1)
    e2 = np.ascontiguousarray(echo2)
    if e1.shape != e2.shape or e1.dtype != e2.dtype:
        raise ValueError("echo volumes must have the same shape and dtype")
    a = QmriDessArgs()
    a.echo1, a.echo2, a.dtype, a.N = _ptr(e1), _ptr(e2), qdtype(e1.dtype), e1.size
    a.c0, a.k, a.c1 = float(c0), float(k), float(c1)
    if bounds is not None:
        a.use_bounds, a.lo, a.hi = 1, float(bounds[0]), float(bounds[1])
    if nan_to_num is not None:
        a.use_nan_to_num = 1
        a.nan_value = 0.0 if isinstance(nan_to_num, bool) else float(nan_to_num)
    a.decimals = -1000000 if decimals is None else int(decimals)
    a.suppress_fat, a.suppress_fluid, a.beta = int(bool(suppress_fat)), int(bool(suppress_fluid)), float(beta)
    od = np.dtype(out_dtype)
    a.out_dtype = QMRI_F64 if od == np.float64 else QMRI_F32
    out = _hostpool.empty(e1.shape, od)
    a.t2 = _ptr(out)
    a.device = _dev(device)
    check(lib.qmri_dess_t2_host(ctypes.byref(a)))
    return out


_CLOSED = {"neither": 0, "left": 1, "right": 2, "both": 3}


def region_stats_host(values, labels=None, keys=(), bounds=None, closed="right", device=None):
    """Count / mean / std / median of ``values`` per region on the GPU (quant_vals.py:145-229): one row per entry of
    ``keys`` (voxels whose label equals the key) plus a last row "total" (label > 0; every voxel when ``labels`` is
    None), over the finite voxels inside ``bounds``.  Returns a float64 array (len(keys) + 1, 4)."""
    lib = load()
    require_device()
    v = np.ascontiguousarray(values)
    if v.dtype not in (np.float32, np.float64):
        v = v.astype(np.float64)
    v = v.reshape(-1)
    a = QmriRegionStatsArgs()
    a.values, a.v_dtype, a.N = _ptr(v), qdtype(v.dtype), v.size
    keep = [v]
    keys = [int(k) for k in keys]
    if labels is not None:
        lab = np.ascontiguousarray(labels)
        kind = {np.dtype(np.int32): 0, np.dtype(np.uint8): 1, np.dtype(np.bool_): 1, np.dtype(np.int16): 2}.get(lab.dtype)
        if kind is None:  # other integer widths / integer-valued floats: one conversion pass on the host
            lab, kind = lab.astype(np.int32), 0
        lab = lab.reshape(-1)
        a.l_kind = kind
        if lab.size != v.size:
            raise ValueError("label map and values differ in size")
        if len(keys) > MAX_REGIONS - 1:
            raise ValueError(f"at most {MAX_REGIONS - 1} labelled regions per call")
        ks = np.asarray(keys, dtype=np.int32)
        a.labels, a.nkeys, a.label_keys = _ptr(lab), len(keys), _ptr(ks)
        keep += [lab, ks]
    else:
        keys = []
    if bounds is not None:
        if closed not in _CLOSED:
            raise ValueError(f"`closed={closed}` is not supported")
        a.use_bounds, a.lo, a.hi, a.closed = 1, float(bounds[0]), float(bounds[1]), _CLOSED[closed]
    out = np.empty((len(keys) + 1, 4), dtype=np.float64)
    a.out, a.device = _ptr(out), _dev(device)
    check(lib.qmri_region_stats_host(ctypes.byref(a)))
    return out


def region_stats_device(values_ptr, v_dtype, n, labels_ptr=None, l_dtype=None, keys=(), bounds=None, closed="right",
                        device=None, stream=None):
    """The same statistics for a map (and label map) that already live on the GPU: raw device pointers, e.g.
    ``tensor.data_ptr()``; ``v_dtype`` float32 / float64, ``l_dtype`` int32 / uint8 / int16."""
    lib = load()
    require_device()
    a = QmriRegionStatsArgs()
    a.values, a.v_dtype, a.N = ctypes.c_void_p(int(values_ptr)), qdtype(v_dtype), int(n)
    keys = [int(k) for k in keys] if labels_ptr is not None else []
    keep = []
    if labels_ptr is not None:
        kind = {np.dtype(np.int32): 0, np.dtype(np.uint8): 1, np.dtype(np.int16): 2}.get(np.dtype(l_dtype))
        if kind is None:
            raise ValueError(f"unsupported label dtype {l_dtype}")
        if len(keys) > MAX_REGIONS - 1:
            raise ValueError(f"at most {MAX_REGIONS - 1} labelled regions per call")
        ks = np.asarray(keys, dtype=np.int32)
        a.labels, a.l_kind, a.nkeys, a.label_keys = ctypes.c_void_p(int(labels_ptr)), kind, len(keys), _ptr(ks)
        keep.append(ks)
    if bounds is not None:
        if closed not in _CLOSED:
            raise ValueError(f"`closed={closed}` is not supported")
        a.use_bounds, a.lo, a.hi, a.closed = 1, float(bounds[0]), float(bounds[1]), _CLOSED[closed]
    out = np.empty((len(keys) + 1, 4), dtype=np.float64)
    a.out, a.device = _ptr(out), _dev(device)
    check(lib.qmri_region_stats_device(ctypes.byref(a), ctypes.c_void_p(int(stream)) if stream else None))
    return out


def rss_host(echo1, echo2, method="rss", device=None):
    lib = load()
    require_device()
    e1 = np.ascontiguousarray(echo1)
    e2 = np.ascontiguousarray(echo2)
    if e1.shape != e2.shape or e1.dtype != e2.dtype:
        raise ValueError("echo volumes must have the same shape and dtype")
    if method not in ("rss", "rms"):
        raise ValueError(f"`method={method}` is not supported")
    out = _hostpool.empty(e1.shape, np.float64)
    check(lib.qmri_rss_host(_ptr(e1), _ptr(e2), qdtype(e1.dtype), e1.size, 0 if method == "rss" else 1,
                            _ptr(out), _dev(device)))
    return out
