"""qDESS: analytic T2 map, echo combination and segmentation entry.

Mirror of the arithmetic of the reference's ``dosma/scan_sequences/mri/qdess.py``:
``QDess.generate_t2_map`` (:105-252), ``calc_rss`` / ``_combine_echoes`` (:254-295), ``segment`` (:64-103).
Not mirrored (out of scope): DICOM loading, private-tag lookup, tissue objects, saving --
so the sequence parameters that the reference reads from DICOM headers (``RepetitionTime``, ``EchoTime``,
``FlipAngle``, private tags 0x001910b6 / 0x001910b7) are explicit arguments here.
"""
import math
import warnings
from copy import deepcopy
from typing import Sequence, Tuple

import numpy as np

from dosma_amd import _lib
from dosma_amd.fitting import _as_kernel_samples
from dosma_amd.med_volume import MedicalVolume
from dosma_amd.quant_vals import T2

__all__ = ["QDess"]


class QDess:
    NAME = "qdess"

    def __init__(self, volumes: Sequence[MedicalVolume]):
        if not isinstance(volumes, (list, tuple)) or len(volumes) != 2 or not all(
                isinstance(v, MedicalVolume) for v in volumes):
            raise ValueError("QDess needs the two echo volumes [echo1, echo2]")
        volumes[0].is_same_dimensions(volumes[1], err=True)
        self.volumes = list(volumes)

    def segment(self, model, tissue=None, use_rss: bool = False):
        """``model.generate_mask`` on echo 1 or on the RSS of the echoes (reference :64-103)."""
        volume = self.calc_rss() if use_rss else self.volumes[0]
        return model.generate_mask(volume)

    def generate_t2_map(self, tissue=None, suppress_fat: bool = False, suppress_fluid: bool = False,
                        beta: float = 1.2, gl_area: float = None, tg: float = None, tr: float = None,
                        te: float = None, alpha: float = None, diffusivity: float = 1.25e-9,
                        t1: float = None, nan_bounds: Tuple[float, float] = (0, 100),
                        nan_to_num: float = 0.0, decimals: int = 1):
        """Analytic T2 from the echo ratio (reference :105-252); returns a :class:`T2` quantitative value.

        ``tr``, ``te`` in ms, ``tg`` in microseconds, ``alpha`` in degrees, ``t1`` in ms (or taken from
        ``tissue.T1_EXPECTED``).  One fused GPU pass instead of the reference's ~10 numpy passes.
        """
        if gl_area is None or tg is None:
            raise ValueError(
                "Dicom headers do not contain tags for `gl_area` and `tg`. Please input manually")
        if tr is None or te is None or alpha is None:
            raise ValueError("`tr`, `te` and `alpha` are required (no DICOM header is read here)")
        if t1 is None:
            if tissue is None or not hasattr(tissue, "T1_EXPECTED"):
                raise ValueError("`t1` is required when no tissue with `T1_EXPECTED` is given")
            t1 = tissue.T1_EXPECTED
        for name, v in (("alpha", alpha), ("diffusivity", diffusivity), ("t1", t1)):
            if np.ndim(v) != 0:
                raise NotImplementedError(f"array-valued `{name}` is not implemented on the GPU")
        echo_1, echo_2 = self.volumes[0].volume, self.volumes[1].volume

        # All timing in seconds -- the reference's own scalar expressions (:190-212), in numpy float64
        TR = float(tr) * 1e-3
        TE = float(te) * 1e-3
        Tg = float(tg) * 1e-6
        T1 = float(t1) * 1e-3
        alpha = math.radians(float(alpha))
        if np.allclose(math.sin(alpha / 2), 0):
            warnings.warn("sin(flip angle) is close to 0 - t2 map may fail.")
        Gl = float(gl_area) / (Tg * 1e6) * 100
        gamma = 4258 * 2 * math.pi
        dkL = gamma * Gl * Tg
        xp = np
        k = (
            xp.power((xp.sin(alpha / 2)), 2)
            * (1 + xp.exp(-TR / T1 - TR * xp.power(dkL, 2) * diffusivity))
            / (1 - xp.cos(alpha) * xp.exp(-TR / T1 - TR * xp.power(dkL, 2) * diffusivity))
        )
        c1 = (TR - Tg / 3) * (xp.power(dkL, 2)) * diffusivity
        c0 = -2000 * (TR - TE)

        e1 = _as_kernel_samples(np.asarray(echo_1))
        e2 = _as_kernel_samples(np.asarray(echo_2))
        if e1.dtype != e2.dtype:
            e1, e2 = e1.astype(np.float64), e2.astype(np.float64)
        t2map = _lib.dess_t2_host(e1, e2, c0, float(k), float(c1), bounds=nan_bounds, nan_to_num=nan_to_num,
                                  decimals=decimals, suppress_fat=suppress_fat,
                                  suppress_fluid=suppress_fluid, beta=beta)
        wrapped = T2(self.volumes[0]._partial_clone(volume=t2map, headers=True))
        if tissue is not None and hasattr(tissue, "add_quantitative_value"):
            tissue.add_quantitative_value(wrapped)
        return wrapped

    def calc_rss(self):
        """Root-sum-of-squares of the two echoes (reference :254-260)."""
        return self._combine_echoes("rss")

    def _combine_echoes(self, method="rss"):
        e1 = _as_kernel_samples(np.asarray(self.volumes[0].volume))
        e2 = _as_kernel_samples(np.asarray(self.volumes[1].volume))
        if e1.dtype != e2.dtype:
            e1, e2 = e1.astype(np.float64), e2.astype(np.float64)
        mv = deepcopy(self.volumes[0])
        mv.volume = _lib.rss_host(e1, e2, method)
        return mv
