"""The two preferences the hot path reads, plus the affine tolerance.

Reference: ``dosma/defaults.py`` (YAML-backed singleton, :41-300) with the shipped template
``dosma/resources/templates/.preferences.yml:3-4, 11-12`` -- ``fitting/r2.threshold: 0.9`` and
``segmentation/batch.size: 16``; ``AFFINE_DECIMAL_PRECISION = 4`` (:34).  The YAML/CLI machinery is
out of scope; the values are plain attributes here and can be reassigned.
"""

AFFINE_DECIMAL_PRECISION = 4
SCANNER_ORIGIN_DECIMAL_PRECISION = 4


class _Preferences:
    fitting_r2_threshold = 0.9
    segmentation_batch_size = 16

    def get(self, key):
        return {"fitting/r2.threshold": self.fitting_r2_threshold,
                "segmentation/batch.size": self.segmentation_batch_size}[key]


preferences = _Preferences()
