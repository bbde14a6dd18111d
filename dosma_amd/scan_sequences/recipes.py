"""Relaxometry scan recipes: thin callers of :class:`MonoExponentialFit`.

The constants are the reference's: CubeQuant T1rho ``bounds=(0, 500), tc0="polyfit", decimal_precision=3``
(dosma/scan_sequences/mri/cube_quant.py:20-26, 170-176), Cones T2* ``(0, inf), "polyfit", 3``
(cones.py:21-27, 163-169), Mapss T1rho from echoes 0-3 ``(0, 500)`` and T2 from echoes [0, 4, 5, 6]
``(0, 100)``, sorted by time (mapss.py:26-34, 170-204, 217-223).  DICOM loading, registration between
echoes (elastix) and tissue bookkeeping are out of scope: volumes and times are explicit arguments.
"""
from typing import Sequence

import numpy as np

from dosma_amd.fitting import MonoExponentialFit
from dosma_amd.med_volume import MedicalVolume
from dosma_amd.quant_vals import T1Rho, T2, T2Star

__all__ = ["CubeQuant", "Cones", "Mapss"]


def _fit(qv_cls, times, volumes, mask, bounds, r2_threshold="preferences"):
    times = np.asarray(times, dtype=np.float64)
    order = np.argsort(times, kind="stable")
    xs = [float(times[i]) for i in order]
    ys = [volumes[i] for i in order]
    tc, r2 = MonoExponentialFit(bounds=bounds, tc0="polyfit", decimal_precision=3,
                                r2_threshold=r2_threshold).fit(xs, ys, mask)
    qv = qv_cls(tc)
    qv.add_additional_volume("r2", r2)
    return qv


class _Relaxometry:
    def __init__(self, volumes: Sequence[MedicalVolume], times: Sequence[float]):
        if len(volumes) != len(times):
            raise ValueError("one acquisition time per volume is required")
        self.volumes = list(volumes)
        self.times = [float(t) for t in times]


class CubeQuant(_Relaxometry):
    """T1rho from spin-lock times (cube_quant.py:139-185)."""

    NAME = "cubequant"
    __T1_RHO_LOWER_BOUND__, __T1_RHO_UPPER_BOUND__ = 0, 500

    def generate_t1_rho_map(self, tissue=None, mask: MedicalVolume = None, num_workers: int = 0):
        return _fit(T1Rho, self.times, self.volumes, mask,
                    (self.__T1_RHO_LOWER_BOUND__, self.__T1_RHO_UPPER_BOUND__))


class Cones(_Relaxometry):
    """T2* from echo times (cones.py:130-178)."""

    NAME = "cones"
    __T2_STAR_LOWER_BOUND__, __T2_STAR_UPPER_BOUND__ = 0, np.inf

    def generate_t2_star_map(self, tissue=None, mask: MedicalVolume = None, num_workers: int = 0):
        return _fit(T2Star, self.times, self.volumes, mask,
                    (self.__T2_STAR_LOWER_BOUND__, self.__T2_STAR_UPPER_BOUND__))


class Mapss(_Relaxometry):
    """MAPSS: 7 echoes; T1rho from echoes 0-3, T2 from echoes [0, 4, 5, 6] (mapss.py:154-248)."""

    NAME = "mapss"

    def generate_t1_rho_map(self, tissue=None, mask: MedicalVolume = None, num_workers: int = 0):
        idx = [0, 1, 2, 3]
        return _fit(T1Rho, [self.times[i] for i in idx], [self.volumes[i] for i in idx], mask, (0, 500))

    def generate_t2_map(self, tissue=None, mask: MedicalVolume = None, num_workers: int = 0):
        idx = [0, 4, 5, 6]
        return _fit(T2, [self.times[i] for i in idx], [self.volumes[i] for i in idx], mask, (0, 100))
