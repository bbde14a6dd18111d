"""dosma_amd -- MI355X-native hot path of DOSMA (per-voxel mono-exponential fitting, 2D-UNet seg).

Drop-in names of the reference's public API for this path (``dosma/__init__.py:12-31``):
``MedicalVolume``, ``CurveFitter``, ``PolyFitter``, ``MonoExponentialFit``, ``curve_fit``, ``polyfit``,
``monoexponential``, ``biexponential``.  Everything per-voxel runs in ``libqmri_hip.so`` (see
``include/qmri.h``); there is no CPU fallback.
"""
from dosma_amd.defaults import preferences  # noqa: F401
from dosma_amd.fitting import (  # noqa: F401
    CurveFitter,
    MonoExponentialFit,
    PolyFitter,
    biexponential,
    curve_fit,
    monoexponential,
    polyfit,
)
from dosma_amd.med_volume import MedicalVolume  # noqa: F401
from dosma_amd._lib import default_device, set_default_device  # noqa: F401
from dosma_amd.io import ImageDataFormat, NiftiReader, NiftiWriter, read, write  # noqa: F401,E402
from dosma_amd.orientation import to_affine  # noqa: F401,E402
from dosma_amd import models, quant_vals, scan_sequences  # noqa: F401,E402

__version__ = "0.1.0"
