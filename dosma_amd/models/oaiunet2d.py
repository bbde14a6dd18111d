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

    # ---- the two halves of the reference's generate_mask around ``model.predict`` (oaiunet2d.py:140-170, 291-320) ----
    # Both are pinned to golden g9, produced by the reference's own generate_mask with a stand-in ``predict``
    # (tests/test_models_host.py::test_generate_mask_pre_post_vs_reference); the fused GPU route below is pinned to the
    # same fixture through pass-through network weights (tests/test_unet_gpu.py::test_generate_mask_vs_reference_golden).
    def _check_volume(self, volume):
        if not isinstance(volume, MedicalVolume) or volume.ndim != 3:
            raise ValueError("`volume` must be a 3D MedicalVolume")

    def _to_sagittal(self, volume: MedicalVolume) -> MedicalVolume:
        """The reference deep-copies and reformats in place (:141-144, :292-295); a reformat to SAGITTAL gives the same
        volume (a view when only axes move) without copying the voxels twice."""
        self._check_volume(volume)
        vol_sag = volume.reformat(SAGITTAL)
        eng = getattr(self, "seg_model", None)
        if eng is not None and hasattr(eng, "H") and vol_sag.volume.shape[:2] != (eng.H, eng.W):
            raise ValueError(f"model was built for slices of {(eng.H, eng.W)}, volume has {vol_sag.volume.shape[:2]}")
        return vol_sag

    def _to_network_input(self, volume: MedicalVolume):
        """-> (sagittal MedicalVolume, the ``(slice, x, y, 1)`` array ``model.predict`` receives in the reference):
        reformat to SAGITTAL, ``__preprocess_volume__``, transpose (2, 0, 1), trailing channel axis (:146-151, :297-302)."""
        vol_sag = self._to_sagittal(volume)
        v = self.__preprocess_volume__(vol_sag.volume)
        return vol_sag, np.expand_dims(np.transpose(v, (2, 0, 1)), axis=-1)

    def _from_network_output(self, probs: np.ndarray, vol_sag: MedicalVolume, orientation):
        """``predict``'s ``(slice, x, y, classes)`` probabilities -> the masks ``generate_mask`` returns: ``>
        sigmoid_threshold`` as uint8, back to ``(x, y, slice)``, one clone of the sagittal volume per class, each
        reformatted to the caller's orientation (:156-170, :306-320)."""
        mask = (np.asarray(probs) > self.sigmoid_threshold).astype(np.uint8)
        planes = [np.transpose(mask[..., i], (1, 2, 0)) for i in range(mask.shape[-1])]
        return self._wrap_planes(vol_sag, planes, orientation)

    def _wrap_planes(self, vol_sag: MedicalVolume, planes, orientation):
        """Per-class ``(x, y, slice)`` uint8 planes -> one MedicalVolume (single-class template, :140-170)."""
        return self._wrap_mask(vol_sag, planes[0], orientation)

    def _segment(self, volume: MedicalVolume):
        """The fused GPU route of the two functions above + the network: -> (sagittal MedicalVolume, per-class (H, W, S)
        uint8 planes).  Transposes, whitening, network, threshold and class planes run in ONE library call
        (``qmri_unet2d_segment_volume``); a ``sigmoid_threshold`` other than 0.5 moves the cut on the logit."""
        vol_sag = self._to_sagittal(volume)
        planes = self.seg_model.segment_volume(vol_sag.volume, whiten=self._WHITEN, eps=self._WHITEN_EPS)  # (C, H, W, S)
        return vol_sag, [planes[i] for i in range(planes.shape[0])]

    def _predict_probabilities(self, v: np.ndarray) -> np.ndarray:
        """``model.predict`` of the reference on an already preprocessed ``(slice, x, y, 1)`` array -> float32 sigmoid
        outputs ``(slice, x, y, classes)`` (the network on the GPU; the sigmoid of its logits on the host)."""
        x = np.ascontiguousarray(v[..., 0], dtype=np.float32)
        logits, _ = self.seg_model.forward_host(x, whiten=False, eps=0.0, want_logits=True)
        with np.errstate(over="ignore"):
            return (1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).astype(np.float32)

    def _wrap_mask(self, vol_sag: MedicalVolume, plane: np.ndarray, orientation):
        """uint8 (H, W, S) mask -> MedicalVolume with the sagittal volume's affine / headers (the reference's
        ``deepcopy(vol_copy); .volume = mask`` :312-316 without copying the float voxels), back in ``orientation``."""
        m = vol_sag._partial_clone(volume=plane, headers=True)
        return m.reformat(orientation, inplace=True) if m.orientation != tuple(orientation) else m

    def generate_mask(self, volume: MedicalVolume):
        if self.sigmoid_threshold == 0.5:  # sigmoid(z) > 0.5  <=>  z > 0: the fused route thresholds the logits on the GPU
            vol_sag, planes = self._segment(volume)
            return self._wrap_planes(vol_sag, planes, volume.orientation)
        # any other threshold: the reference's own sequence, step by step (:140-170)
        vol_sag, v = self._to_network_input(volume)
        return self._from_network_output(self._predict_probabilities(v), vol_sag, volume.orientation)

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

    def _wrap_planes(self, vol_sag: MedicalVolume, planes, orientation):
        """One MedicalVolume per class, keyed in the template's class order (:309-320)."""
        return {category: self._wrap_mask(vol_sag, planes[i], orientation)
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
