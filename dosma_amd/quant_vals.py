"""Quantitative-value containers: the result type of the fit path.

Mirror of the reference's ``dosma/core/quant_vals.py`` -- ``QuantitativeValue`` (:29-303: the fitted map
+ ``additional_volumes["r2"]``, ``to_metrics`` :145-229, ``get_qv`` :245-263) and ``T1Rho`` / ``T2`` /
``T2Star`` (:306-336).  What the scan classes do with the outputs of ``MonoExponentialFit.fit``:
``qv.T1Rho(t1rho_map); qv.add_additional_volume("r2", r2)`` (cube_quant.py:180-183).
``save_data`` / ``load_data`` (:78-126) go through ``dosma_amd.io`` (NIfTI-1).
"""
from abc import ABC
from enum import Enum
import os
import warnings
from typing import Callable, Dict, Tuple, Union

import numpy as np

from dosma_amd.med_volume import MedicalVolume

__all__ = ["QuantitativeValueType", "QuantitativeValue", "T1Rho", "T2", "T2Star"]


def _inside(values, bounds, closed):
    """Boolean map of ``values`` inside the interval ``bounds`` (None: everywhere)."""
    if not bounds:
        return np.ones(np.shape(values), dtype=bool)
    if len(bounds) != 2 or not bounds[0] <= bounds[1]:
        raise AssertionError(f"bounds must be (lower, upper) with lower <= upper, got {bounds}")
    ends = {"right": (False, True), "left": (True, False), "both": (True, True), "neither": (False, False)}
    if closed not in ends:
        raise AssertionError(closed)
    lo_in, hi_in = ends[closed]
    lo_ok = values >= bounds[0] if lo_in else values > bounds[0]
    hi_ok = values <= bounds[1] if hi_in else values < bounds[1]
    return lo_ok & hi_ok


class QuantitativeValueType(Enum):
    T1_RHO = 1
    T2 = 2
    T2_STAR = 3


class QuantitativeValue(ABC):
    ID = 0
    NAME = ""

    def __init__(self, volumetric_map: MedicalVolume = None):
        assert self.ID > 0, "Attribute `ID` not initialized for {}".format(type(self))
        assert self.NAME != "", "Attribute `NAME` not initialized for {}".format(type(self))
        if volumetric_map is not None and not isinstance(volumetric_map, MedicalVolume):
            raise TypeError("`volumetric_map` must be of type MedicalVolume")
        self.volumetric_map = volumetric_map
        self.additional_volumes = {}

    def add_additional_volume(self, name: str, volume: MedicalVolume):
        if not isinstance(volume, MedicalVolume):
            raise TypeError("`volumes` must be of type MedicalVolume")
        self.additional_volumes[name] = volume

    def save_data(self, dir_path, data_format=None):
        """Maps go to ``dir_path/NAME/NAME.nii.gz`` (+ ``NAME-<extra>.nii.gz``), always NIfTI (reference :78-108)."""
        from dosma_amd.io import ImageDataFormat

        if data_format is not None and data_format != ImageDataFormat.nifti:
            import warnings

            warnings.warn("Due to bit depth issues, only nifti format is supported for quantitative values. "
                          "Writing as nifti file...")
        if self.volumetric_map is not None:
            filepath = os.path.join(dir_path, self.NAME, "{}.nii.gz".format(self.NAME))
            self.volumetric_map.save_volume(filepath, data_format=ImageDataFormat.nifti)
        for name, vol in self.additional_volumes.items():
            filepath = os.path.join(dir_path, self.NAME, "{}-{}.nii.gz".format(self.NAME, name))
            vol.save_volume(filepath, data_format=ImageDataFormat.nifti)

    def load_data(self, dir_path):
        """Reload ``dir_path/NAME/NAME.nii.gz``; additional volumes are not reloaded (reference :110-124)."""
        from dosma_amd.io import generic_load

        file_path = os.path.join(dir_path, self.NAME, "{}.nii.gz".format(self.NAME))
        self.volumetric_map = generic_load(file_path, expected_num_volumes=1)

    def to_metrics(self, mask: MedicalVolume = None, labels: Dict[int, str] = None,
                   bounds: Tuple[float, float] = None, closed: str = "right",
                   fns: Dict[str, Callable] = None):
        """One row per region -- ``Category, Mean, Std, Median, # Voxels`` (+ one column per entry of ``fns``) -- as a
        ``pandas.DataFrame`` (reference :145-229).  Regions: every label of ``mask`` (or of ``labels``) followed by
        "total" (all labelled voxels); without a mask a single "total" row over the whole map.  Only finite voxels
        inside ``bounds`` count; ``closed`` says which ends of the interval belong to it."""
        import pandas as pd

        values = self.volumetric_map.volume
        _inside(values[:0], bounds, closed)  # argument checks of the reference (AssertionError), before any work
        label_map = None
        if mask is not None:
            label_map = mask.reformat(self.volumetric_map.orientation).volume
            if labels is None:
                labels = {int(v): f"label_{int(v)}" for v in np.unique(label_map) if v > 0}
        extra = dict(fns or {})
        names = (list(labels.values()) if mask is not None else []) + ["total"]
        stats = self._region_stats_gpu(values, label_map, labels, bounds, closed) if not extra else None
        if stats is not None:
            rows = [{"Category": name, "Mean": st[1], "Std": st[2], "Median": st[3], "# Voxels": int(st[0])}
                    for name, st in zip(names, stats)]
            return pd.DataFrame(rows, columns=["Category", "Mean", "Std", "Median", "# Voxels"])
        # host evaluation, the reference's own (numpy) route: needed for user callables (`fns` take the region's voxels
        # as a numpy array) and for label maps the GPU kernel does not take (see _region_stats_gpu)
        usable = np.isfinite(values) & _inside(values, bounds, closed)
        if mask is None:
            regions = [("total", usable)]
        else:
            label_map = np.where(usable, label_map, 0)
            regions = [(name, label_map == key) for key, name in labels.items()]
            regions.append(("total", label_map > 0))
        rows = []
        with np.errstate(all="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)  # empty regions -> NaN, quietly
            for name, where in regions:
                v = values[where]
                row = {"Category": name, "Mean": np.nanmean(v), "Std": np.nanstd(v), "Median": np.nanmedian(v),
                       "# Voxels": int(v.size)}
                row.update({key: fn(v) for key, fn in extra.items()})
                rows.append(row)
        return pd.DataFrame(rows, columns=["Category", "Mean", "Std", "Median", "# Voxels", *extra])

    @staticmethod
    def _region_stats_gpu(values, label_map, labels, bounds, closed):
        """Rows (count, mean, std, median) per label + "total" from the HIP kernels (csrc/region_stats.hip), or None
        when this call has to take the host route: no GPU in the process, label keys <= 0, or a label map that is
        not integer valued (the kernel compares int32 labels)."""
        from . import _lib

        try:
            if _lib.load().qmri_device_count() <= 0:
                return None
        except _lib.QmriError:
            return None
        if label_map is None:
            return _lib.region_stats_host(values, None, (), bounds, closed)
        keys = [int(k) for k in labels]
        if any(k <= 0 for k in keys) or any(int(k) != k for k in labels):
            return None
        lab = np.asarray(label_map)
        if lab.dtype.kind == "f":
            if not np.array_equal(lab, np.rint(lab)):
                return None
        elif lab.dtype.kind not in "iub":
            return None
        rows, total = [], None
        step = _lib.MAX_REGIONS - 1
        for i in range(0, max(len(keys), 1), step):
            part = _lib.region_stats_host(values, lab, keys[i:i + step], bounds, closed)
            rows.extend(part[:-1])
            total = part[-1]
        return rows + [total]

    @staticmethod
    def get_qv(qv_id: Union[int, str]):
        for qv in [T1Rho(), T2(), T2Star()]:
            if qv.NAME.lower() == qv_id or qv.NAME == qv_id or qv.ID == qv_id:
                return qv
        raise ValueError("Quantitative Value with name or id {} not found".format(qv_id))

    @property
    def qv_type(self) -> QuantitativeValueType:
        raise NotImplementedError(f"Quantitative value type not implemented for {type(self)}")


class T1Rho(QuantitativeValue):
    ID = 1
    NAME = "t1_rho"

    @property
    def qv_type(self):
        return QuantitativeValueType.T1_RHO


class T2(QuantitativeValue):
    ID = 2
    NAME = "t2"

    @property
    def qv_type(self):
        return QuantitativeValueType.T2


class T2Star(QuantitativeValue):
    ID = 3
    NAME = "t2_star"

    @property
    def qv_type(self):
        return QuantitativeValueType.T2_STAR
