"""Quantitative-value containers: the result type of the fit path.

Mirror of the reference's ``dosma/core/quant_vals.py`` -- ``QuantitativeValue`` (:29-303: the fitted map
+ ``additional_volumes["r2"]``, ``to_metrics`` :145-229, ``get_qv`` :245-263) and ``T1Rho`` / ``T2`` /
``T2Star`` (:306-336).  What the scan classes do with the outputs of ``MonoExponentialFit.fit``:
``qv.T1Rho(t1rho_map); qv.add_additional_volume("r2", r2)`` (cube_quant.py:180-183).
``save_data`` / ``load_data`` (:78-126) go through ``dosma_amd.io`` (NIfTI-1).
"""
from abc import ABC
from collections import defaultdict
from enum import Enum
import os
from typing import Callable, Dict, Tuple, Union

import numpy as np

from dosma_amd.med_volume import MedicalVolume

__all__ = ["QuantitativeValueType", "QuantitativeValue", "T1Rho", "T2", "T2Star"]


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
        """Mean / Std / Median / # Voxels per label (``pandas.DataFrame``), reference :145-229.

        Valid voxels are finite and inside ``bounds`` (interval closed on ``closed``)."""
        import pandas as pd

        volume = self.volumetric_map.volume
        valid_mask = np.isfinite(volume)
        if bounds:
            assert len(bounds) == 2, len(bounds)
            lb, ub = bounds[0], bounds[1]
            assert lb <= ub, f"lower:{lb}, upper: {ub}"
            assert closed in ("right", "left", "both", "neither"), closed
            lb_mask = volume >= lb if closed in ("left", "both") else volume > lb
            ub_mask = volume <= ub if closed in ("right", "both") else volume < ub
            valid_mask &= lb_mask & ub_mask
        if mask is not None:
            mask = mask.reformat(self.volumetric_map.orientation).volume
            if labels is None:
                labels = {int(i): f"label_{int(i)}" for i in np.unique(mask) if i > 0}
            labels = dict(labels)
            labels.update({-1: "total"})
            mask = mask.copy()
            mask[~valid_mask] = 0
        else:
            labels = {-2: "total"}
        fns = fns or {}
        metrics = defaultdict(list)
        with np.errstate(all="ignore"):
            import warnings

            for label, name in labels.items():
                if label == -2:
                    vals = volume[valid_mask]
                elif label == -1:
                    vals = volume[mask > 0]
                else:
                    vals = volume[mask == label]
                metrics["Category"].append(name)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore", category=RuntimeWarning)
                    metrics["Mean"].append(np.nanmean(vals))
                    metrics["Std"].append(np.nanstd(vals))
                    metrics["Median"].append(np.nanmedian(vals))
                metrics["# Voxels"].append(int(np.prod(vals.shape)))
                for fname, fn in fns.items():
                    metrics[fname].append(fn(vals))
        return pd.DataFrame(metrics)

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
