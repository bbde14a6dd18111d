"""``MedicalVolume``: the boundary type of the fit / segmentation path (ndarray + RAS+ affine + headers).

Own, minimal implementation of what the hot path touches in the reference's
``dosma/core/med_volume.py`` -- constructor :152-158, ``reformat`` :177-275, ``reformat_as`` :277-288,
``is_same_dimensions`` :339-385, ``_partial_clone`` :1118-1130, ``__getitem__`` :1222-1249,
``__array_ufunc__`` :1328-1356 -- so that ``CurveFitter.fit`` / ``MonoExponentialFit.fit`` /
``generate_mask`` take and return the same kind of object.  Not reproduced (out of scope, SURVEY.md
section 2): DICOM/NIfTI I/O, SimpleITK / nibabel / torch interop, CuPy devices, memmaps.

Like the reference (fitting.py:403-406, 745-746) the fit path only accepts CPU volumes: data are
host ndarrays; the library moves them to the GPU internally.
"""
from copy import deepcopy
from numbers import Number

import numpy as np
from numpy.lib.mixins import NDArrayOperatorsMixin

from dosma_amd import orientation as stdo

__all__ = ["MedicalVolume"]

SCANNER_ORIGIN_DECIMAL_PRECISION = 4  # dosma/defaults.py:35

_HANDLED_FUNCTIONS = {}


def _implements(*np_functions):
    def deco(fn):
        for f in np_functions:
            _HANDLED_FUNCTIONS[f] = fn
        return fn

    return deco


class MedicalVolume(NDArrayOperatorsMixin):
    """nD image (first three axes spatial) with a 4x4 RAS+ affine and optional per-slice headers."""

    def __init__(self, volume, affine, headers=None):
        self._volume = volume if isinstance(volume, np.memmap) else np.asarray(volume)
        self._affine = np.array(affine, dtype=np.float64)
        if self._affine.shape != (4, 4):
            raise ValueError("`affine` must be a 4x4 matrix")
        self._headers = self._format_headers(headers) if headers is not None else None

    # ------------------------------------------------------------------ basic properties
    @property
    def volume(self):
        return self._volume

    @volume.setter
    def volume(self, value):
        value = np.asarray(value)
        if value.ndim != self._volume.ndim:
            raise ValueError("New volume must be same as current volume")
        if value.shape != self._volume.shape:
            self._headers = None
        self._volume = value

    A = volume

    @property
    def affine(self):
        return self._affine

    @property
    def shape(self):
        return self._volume.shape

    @property
    def ndim(self):
        return self._volume.ndim

    @property
    def dtype(self):
        return self._volume.dtype

    @property
    def device(self):
        return "cpu"

    @property
    def orientation(self):
        return stdo.orientation_from_affine(self._affine)

    @property
    def pixel_spacing(self):
        return tuple(np.sqrt(np.sum(self._affine[:3, :3] ** 2, axis=0)))

    @property
    def scanner_origin(self):
        return tuple(self._affine[:3, 3])

    def headers(self, flatten=False):
        if flatten and self._headers is not None:
            return self._headers.flatten()
        return self._headers

    def cpu(self):
        return self

    def to(self, device):
        if str(device) not in ("cpu", "-1"):
            raise RuntimeError(
                "dosma_amd.MedicalVolume lives on the host; the library stages data to the GPU itself")
        return self

    def astype(self, dtype, **kwargs):
        self._volume = self._volume.astype(dtype, **kwargs)
        return self

    def clone(self, headers=True):
        return MedicalVolume(self._volume.copy(), self._affine.copy(),
                             headers=deepcopy(self._headers) if headers else self._headers)

    def _partial_clone(self, **kwargs):
        """New volume taking ``volume`` / ``affine`` / ``headers`` from kwargs, the rest from self.

        ``headers=True`` deep-copies the headers (reference :1118-1130).
        """
        if kwargs.get("volume", None) is False:
            kwargs["volume"] = self._volume
        for key in ("volume", "affine"):
            if key not in kwargs or kwargs[key] is True:
                kwargs[key] = getattr(self, "_" + key).copy()
        if "headers" not in kwargs:
            kwargs["headers"] = self._headers
        elif isinstance(kwargs["headers"], bool) and kwargs["headers"]:
            kwargs["headers"] = deepcopy(self._headers)
        return self.__class__(**kwargs)

    def _format_headers(self, headers):
        headers = np.asarray(headers)
        if headers.ndim > self._volume.ndim:
            raise ValueError(f"`headers` has too many dimensions. Got headers.ndim={headers.ndim}, "
                             f"but volume.ndim={self._volume.ndim}")
        for dim in range(-headers.ndim, 0):
            if headers.shape[dim] not in (1, self._volume.shape[dim]):
                raise ValueError(f"`headers` must follow standard broadcasting shape. Got "
                                 f"headers.shape={headers.shape}, but volume.shape={self._volume.shape}")
        return headers.reshape((1,) * (self._volume.ndim - headers.ndim) + headers.shape)

    # ------------------------------------------------------------------ orientation
    def save_volume(self, file_path: str, data_format=None):
        """Write the volume (reference med_volume.py:160-175); only NIfTI is built."""
        from dosma_amd.io import ImageDataFormat, get_writer

        get_writer(ImageDataFormat.nifti if data_format is None else data_format).save(self, file_path)

    def reformat(self, new_orientation, inplace=False):
        """Transpose + flip the first three axes so that ``self.orientation == new_orientation``.

        The affine follows: transposed axes swap columns; a flipped axis negates its column and moves
        the origin to the other end of that axis (rounded like the reference, :246-257).
        """
        new_orientation = tuple(new_orientation)
        cur = self.orientation
        if new_orientation == cur:
            return self if inplace else self._partial_clone(volume=self._volume)

        perm = stdo.get_transpose_inds(cur, new_orientation)
        full_perm = perm + tuple(range(3, self._volume.ndim))
        volume = np.transpose(self._volume, full_perm)
        headers = None if self._headers is None else np.transpose(self._headers, full_perm)
        affine = self._affine.copy()
        affine[:, :3] = self._affine[:, list(perm)]

        flips = stdo.get_flip_inds(tuple(cur[i] for i in perm), new_orientation)
        if flips:
            volume = np.flip(volume, axis=tuple(flips))
            if headers is not None:
                headers = np.flip(headers, axis=tuple(flips))
            origin = affine[:3, 3].copy()
            for ax in flips:
                origin = origin + affine[:3, ax] * (volume.shape[ax] - 1)
                affine[:3, ax] = -affine[:3, ax]
            affine[:3, 3] = origin
        affine[:3, 3] = np.round(affine[:3, 3], SCANNER_ORIGIN_DECIMAL_PRECISION)
        affine[affine == 0] = 0  # no negative zeros

        if inplace:
            self._volume, self._affine, self._headers = volume, affine, headers
            out = self
        else:
            out = self._partial_clone(volume=volume, affine=affine, headers=headers)
        assert out.orientation == new_orientation
        return out

    def reformat_as(self, other, inplace=False):
        return self.reformat(other.orientation, inplace=inplace)

    def _allclose_spacing(self, mv, precision=None):
        if precision is not None:
            tol = 10 ** (-precision)
            return bool(np.allclose(mv.affine[:3, :3], self.affine[:3, :3], atol=tol)
                        and np.allclose(mv.scanner_origin, self.scanner_origin, rtol=tol))
        return bool((mv.affine == self.affine).all())

    def is_same_dimensions(self, mv, precision=None, err=False):
        if not isinstance(mv, MedicalVolume):
            raise TypeError("`mv` must be a MedicalVolume.")
        close = self._allclose_spacing(mv, precision)
        same_o = mv.orientation == self.orientation
        same_s = mv.volume.shape == self.volume.shape
        out = close and same_o and same_s
        if err and not out:
            tol = f" (tol: 1e-{precision})" if precision else ""
            if not close:
                raise ValueError(f"Affine matrices not equal{tol}:\n{self._affine}\n{mv._affine}")
            if not same_o:
                raise ValueError(f"Orientations not equal: {self.orientation}, {mv.orientation}")
            raise ValueError(f"Shapes not equal: {self._volume.shape}, {mv._volume.shape}")
        return out

    def is_identical(self, mv):
        if not isinstance(mv, MedicalVolume):
            raise TypeError("`mv` must be a MedicalVolume.")
        return self.is_same_dimensions(mv) and bool((mv.volume == self.volume).all())

    # ------------------------------------------------------------------ indexing
    def _canonical_index(self, index):
        """Expand Ellipsis / pad to ndim; integer or fancy indices on spatial axes are rejected."""
        if not isinstance(index, tuple):
            index = (index,)
        n_real = sum(1 for s in index if s is not None and s is not Ellipsis)
        if sum(1 for s in index if s is Ellipsis) > 1:
            raise IndexError("an index can only have a single ellipsis")
        out = []
        for s in index:
            if s is Ellipsis:
                out.extend([slice(None)] * (self.ndim - n_real))
            else:
                out.append(s)
        if n_real > self.ndim:
            raise IndexError("too many indices for MedicalVolume")
        axis = 0
        for s in out:
            if s is None:
                if axis < 3:
                    raise IndexError("Cannot insert an axis among the spatial axes")
                continue
            if axis < 3 and not isinstance(s, slice):
                raise IndexError("Spatial axes can only be indexed with slices")
            axis += 1
        out.extend([slice(None)] * (self.ndim - axis))
        return tuple(out)

    def __getitem__(self, index):
        if isinstance(index, MedicalVolume):
            index = index.reformat_as(self).A
        if isinstance(index, np.ndarray) and index.dtype == bool and index.shape == self.shape:
            raise IndexError("Boolean-mask indexing changes the shape; index `.volume` instead")
        index = self._canonical_index(index)
        volume = self._volume[index]
        if any(d == 0 for d in volume.shape):
            raise IndexError("Empty slice requested")
        headers = self._headers
        if headers is not None:
            hidx = []
            for ax, s in enumerate(i for i in index if i is not None):
                if headers.shape[ax] == 1:
                    hidx.append(0 if isinstance(s, (int, np.integer)) else slice(None))
                else:
                    hidx.append(s)
            headers = headers[tuple(hidx)]
        t = np.eye(4)
        for ax in range(3):
            start, _, step = index[ax].indices(self.shape[ax])
            t[ax, ax] = step
            t[ax, 3] = start
        return self._partial_clone(volume=volume, affine=self._affine @ t, headers=headers)

    def __setitem__(self, index, value):
        if isinstance(value, MedicalVolume):
            value = value._volume
        if isinstance(index, MedicalVolume):
            index = index.reformat_as(self).A
        self._volume[index] = value

    # ------------------------------------------------------------------ numpy protocol
    def __array__(self, dtype=None, copy=None):
        arr = np.asarray(self._volume)
        return arr if dtype is None else arr.astype(dtype, copy=False)

    def _unwrap(self, obj):
        if isinstance(obj, MedicalVolume):
            self.is_same_dimensions(obj, err=True)
            return obj._volume
        if isinstance(obj, (Number, np.ndarray, np.generic)):
            return obj
        return NotImplemented

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method not in ("__call__", "reduce"):
            return NotImplemented
        args = [self._unwrap(i) for i in inputs]
        if any(a is NotImplemented for a in args):
            return NotImplemented
        if method == "__call__":
            volume = ufunc(*args, **kwargs)
            if volume.shape != self._volume.shape:
                raise ValueError(f"{type(self).__name__} does not support operations that change "
                                 "shape. Use operations on `self.volume` to modify array objects.")
            return self._partial_clone(volume=volume)
        return self._reduce(ufunc.reduce, *args, **kwargs)

    def _reduce(self, func, *args, **kwargs):
        axis = kwargs.get("axis", None)
        if axis is not None:
            seq = axis if isinstance(axis, (tuple, list)) else (axis,)
            seq = tuple(a if a >= 0 else self.ndim + a for a in seq)
            if any(a < 3 for a in seq):
                raise ValueError("Cannot reduce MedicalVolume along spatial dimensions")
            kwargs["axis"] = seq if isinstance(axis, (tuple, list)) else seq[0]
        out = func(*args, **kwargs)
        if np.isscalar(out) or out.ndim == 0:
            return out
        return self._partial_clone(volume=out, headers=None)

    def __array_function__(self, func, types, args, kwargs):
        if func not in _HANDLED_FUNCTIONS:
            return NotImplemented
        return _HANDLED_FUNCTIONS[func](*args, **kwargs)

    def __repr__(self):
        return (f"{type(self).__name__}(\n  shape={self.shape},\n  ornt={self.orientation}),\n"
                f"  spacing={self.pixel_spacing},\n  origin={self.scanner_origin},\n  device=cpu\n)")

    def sum(self, axis=None, **kw):
        return self._reduce(np.sum, self._volume, axis=axis, **kw)

    def mean(self, axis=None, **kw):
        return self._reduce(np.mean, self._volume, axis=axis, **kw)


# numpy functions the path (and its callers' tests) apply to MedicalVolumes:
# np.around (fitting.py:736-737), np.nan_to_num, np.clip, np.all / np.any, np.stack of volumes.
@_implements(np.around, np.round)
def _around(a, decimals=0, out=None):
    return a._partial_clone(volume=np.around(a.volume, decimals=decimals))


@_implements(np.nan_to_num)
def _nan_to_num(x, copy=True, nan=0.0, posinf=None, neginf=None):
    vol = np.nan_to_num(x.volume, copy=copy, nan=nan, posinf=posinf, neginf=neginf)
    if not copy:
        x._volume = vol
        return x
    return x._partial_clone(volume=vol)


@_implements(np.clip)
def _clip(a, a_min, a_max, **kw):
    return a._partial_clone(volume=np.clip(a.volume, a_min, a_max, **kw))


@_implements(np.all)
def _all(a, axis=None, **kw):
    return a._reduce(np.all, a.volume, axis=axis, **kw)


@_implements(np.any)
def _any(a, axis=None, **kw):
    return a._reduce(np.any, a.volume, axis=axis, **kw)


@_implements(np.amax, np.max)
def _amax(a, axis=None, **kw):
    return a._reduce(np.amax, a.volume, axis=axis, **kw)


@_implements(np.amin, np.min)
def _amin(a, axis=None, **kw):
    return a._reduce(np.amin, a.volume, axis=axis, **kw)


@_implements(np.stack)
def _stack(xs, axis=-1):
    xs = list(xs)
    if not xs or not all(isinstance(x, MedicalVolume) for x in xs):
        raise TypeError("np.stack: all inputs must be MedicalVolumes")
    if isinstance(axis, int) and (0 <= axis < 3 or (axis < 0 and axis + xs[0].ndim + 1 < 3)):
        raise ValueError("Cannot stack across spatial dimension")
    for x in xs[1:]:
        xs[0].is_same_dimensions(x, err=True)
    return xs[0]._partial_clone(volume=np.stack([x.volume for x in xs], axis=axis), headers=None)
