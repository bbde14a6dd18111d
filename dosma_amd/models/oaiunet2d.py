"""2D U-Net knee segmentation models on the MI355X kernels.

Mirror of the reference's ``dosma/models/oaiunet2d.py``: ``OAIUnet2D`` (:24-175), ``IWOAIOAIUnet2D``
(:178-323), ``IWOAIOAIUnet2DNormalized`` (:326-345) -- same aliases, weights-file checks, class order,
preprocessing and ``generate_mask`` plumbing (deepcopy, reformat to SAGITTAL, (slice, x, y, 1) batches,
``> sigmoid_threshold`` -> uint8, reformat back).  The network itself (:197-289) runs in
``libqmri_hip.so`` (conv / transposed conv / BN affine on MFMA).
"""
import os
from copy import deepcopy

import numpy as np

from dosma_amd.med_volume import MedicalVolume
from dosma_amd.models.seg_model import HipSegModel, whiten_volume  # noqa: F401
from dosma_amd.orientation import SAGITTAL

__all__ = ["OAIUnet2D", "IWOAIOAIUnet2D", "IWOAIOAIUnet2DNormalized"]


class OAIUnet2D(HipSegModel):
    """Single-class model of Chaudhari et al. (reference :24-175): whiten(eps=1e-8), one output."""

    ALIASES = ["oai-unet2d", "oai_unet2d"]
    sigmoid_threshold = 0.5
    _WHITEN = True
    _WHITEN_EPS = 1e-8

    def _n_classes(self):
        return 1

    def _check_threshold(self):
        if self.sigmoid_threshold != 0.5:
            # the kernel thresholds the logit at 0 (sigmoid > 0.5); other thresholds shift the logit cut
            return float(np.log(self.sigmoid_threshold / (1 - self.sigmoid_threshold)))
        return 0.0

    def _segment(self, volume: MedicalVolume):
        """-> (sagittal MedicalVolume of the input, list of per-class (H, W, S) uint8 arrays)."""
        if not isinstance(volume, MedicalVolume) or volume.ndim != 3:
            raise ValueError("`volume` must be a 3D MedicalVolume")
        # the reference deep-copies and reformats in place (:292-296); a reformat to SAGITTAL gives the same volume
        # (a view when only axes move) without copying the voxels twice
        vol_sag = volume.reformat(SAGITTAL)
        vol = vol_sag.volume
        eng = self.seg_model
        if vol.shape[:2] != (eng.H, eng.W):
            raise ValueError(f"model was built for slices of {(eng.H, eng.W)}, volume has {vol.shape[:2]}")
        cut = self._check_threshold()
        if cut == 0.0:
            planes = eng.segment_volume(vol, whiten=self._WHITEN, eps=self._WHITEN_EPS)  # (C, H, W, S)
            return vol_sag, [planes[i] for i in range(planes.shape[0])]
        logits, _ = self._predict(vol, self._WHITEN, self._WHITEN_EPS, want_logits=True)
        mask = (logits > cut).astype(np.uint8)  # (S, H, W, C)
        return vol_sag, [np.ascontiguousarray(np.transpose(mask[..., i], (1, 2, 0))) for i in range(mask.shape[-1])]

    def _wrap_mask(self, vol_sag: MedicalVolume, plane: np.ndarray, orientation):
        """uint8 (H, W, S) mask -> MedicalVolume with the sagittal volume's affine / headers (the reference's
        ``deepcopy(vol_copy); .volume = mask`` :312-316 without copying the float voxels), back in ``orientation``."""
        m = vol_sag._partial_clone(volume=plane, headers=True)
        return m.reformat(orientation, inplace=True) if m.orientation != tuple(orientation) else m

    def generate_mask(self, volume: MedicalVolume):
        vol_sag, planes = self._segment(volume)
        return self._wrap_mask(vol_sag, planes[0], volume.orientation)

    def __preprocess_volume__(self, volume: np.ndarray):
        return whiten_volume(volume, eps=1e-8)


class IWOAIOAIUnet2D(OAIUnet2D):
    """Team 6, 2019 IWOAI challenge (reference :178-323): no preprocessing, classes fc / tc / pc / men."""

    ALIASES = ["iwoai-2019-t6"]
    _WEIGHTS_FILE = "iwoai-2019-unet2d_fc-tc-pc-men_weights.h5"
    CATEGORIES = ("fc", "tc", "pc", "men")
    _WHITEN = False
    _WHITEN_EPS = 0.0

    def __init__(self, input_shape, weights_path, force_weights=False):
        if not force_weights and not isinstance(weights_path, dict):
            base = os.path.basename(str(weights_path))
            if os.path.splitext(base)[0] != os.path.splitext(self._WEIGHTS_FILE)[0]:
                raise ValueError(f"Weights {weights_path} not supported for {type(self)}")
        super().__init__(input_shape, weights_path)

    def _n_classes(self):
        return 4

    def generate_mask(self, volume: MedicalVolume):
        vol_sag, planes = self._segment(volume)
        return {category: self._wrap_mask(vol_sag, planes[i], volume.orientation)
                for i, category in enumerate(self.CATEGORIES)}

    def __preprocess_volume__(self, volume: np.ndarray):
        return volume


class IWOAIOAIUnet2DNormalized(IWOAIOAIUnet2D):
    """Same network with zero-mean / unit-std input (reference :326-345)."""

    ALIASES = ("iwoai-2019-t6-normalized",)
    _WEIGHTS_FILE = "iwoai-2019-unet2d-normalized_fc-tc-pc-men_weights.h5"
    _WHITEN = True
    _WHITEN_EPS = 0.0

    def __preprocess_volume__(self, volume: np.ndarray):
        return whiten_volume(volume)
