"""Segmentation-model base class on the HIP U-Net engine.

Mirror of the reference's ``dosma/models/seg_model.py`` (``SegModel`` :14-79, ``KerasSegModel`` :82-106,
``whiten_volume`` :114-127) for the inference path: same constructor ``(input_shape, weights_path,
force_weights=False)``, same ``generate_mask(volume)`` / ``__call__`` contract, ``batch_size`` from
``preferences.segmentation_batch_size``.  ``build_model`` creates a native engine
(``libqmri_hip.so``: ``qmri_unet2d_*``) instead of a Keras model; there is no TensorFlow/CPU path.
"""
from abc import ABC, abstractmethod

import numpy as np

from dosma_amd.defaults import preferences
from dosma_amd.med_volume import MedicalVolume

__all__ = ["SegModel", "HipSegModel", "whiten_volume"]

__VOLUME_DIMENSIONS__ = 3


class SegModel(ABC):
    """
    Args:
        input_shape (Tuple[int]): ``(height, width, channels)`` of one slice.
        weights_path (str | dict): Keras ``.h5`` (read with h5py or the built-in reader) or ``.npz`` weights file, or a
            dict of arrays in Keras layouts (``dosma_amd.models.weights``).
        force_weights (bool, optional): load weights without checking the file name.
    """

    ALIASES = [""]

    def __init__(self, input_shape, weights_path, force_weights=False):
        self.batch_size = preferences.segmentation_batch_size
        self.seg_model = self.build_model(input_shape, weights_path)

    @abstractmethod
    def build_model(self, input_shape, weights_path):
        pass

    @abstractmethod
    def generate_mask(self, volume: MedicalVolume):
        """Segment the volume; returns uint8 {0,1} MedicalVolume(s) of ``volume.shape``."""
        pass

    def __call__(self, *args, **kwargs):
        return self.generate_mask(*args, **kwargs)

    def __preprocess_volume__(self, volume: np.ndarray):
        return volume

    def __postprocess_volume__(self, volume: np.ndarray):
        return volume


class HipSegModel(SegModel):
    """Counterpart of the reference's ``KerasSegModel``: builds the native engine and loads weights."""

    #: "fp16x3" (fp16 hi + lo operand parts, 3 MFMAs per product: logits within 1e-3 of an fp64 run) or "bf16" (single MFMA)
    precision = "fp16x3"
    #: slices per pass through the network on the GPU.  The reference's ``batch_size`` (Keras ``predict(batch_size=)``,
    #: ``preferences.segmentation_batch_size`` = 16) only chunks the work -- results do not depend on it -- and is kept
    #: as an attribute for compatibility; the engine uses the larger of the two.  The persistent convolution kernels want
    #: the whole volume in one pass (the 12 x 12 level of 16 slices is 10 tiles for 256 CUs): 160 slices of 384 x 384 need
    #: ~25 GB of activation buffers in the parity mode, ~12 GB in bf16, of the 288 GB
    gpu_batch = 160
    device = None  # None: dosma_amd.set_default_device / DOSMA_AMD_DEVICE (0)

    def build_model(self, input_shape, weights_path=None):
        from dosma_amd import _lib
        from dosma_amd.models import weights as W

        if type(input_shape) is not tuple or len(input_shape) != 3 or input_shape[2] != 1:
            raise ValueError("input_size must be a tuple of size (height, width, 1)")
        if weights_path is None:
            raise ValueError("weights are required")
        if isinstance(weights_path, dict):
            w = weights_path
        elif str(weights_path).endswith(".npz"):
            w = W.load_npz(weights_path)
        else:
            w = W.load_keras_h5(weights_path)
        n_classes = self._n_classes()
        W.validate(w, n_classes=n_classes)
        # activation buffers: ~1.1 KB per pixel and slice in the parity mode (25 GB for 160 slices of 384 x 384), half of
        # that in bf16.  One model takes at most 64 GB and at most half of what is FREE on the device right now (other
        # models / processes share it); if the allocation still fails (a race with another process) the batch is halved
        # down to the reference's ``batch_size``.
        per_slice = input_shape[0] * input_shape[1] * (1100 if self.precision != "bf16" else 560)
        budget = 64e9
        try:
            free, _total = _lib.device_mem_info(self.device)
            budget = min(budget, 0.5 * free)
        except _lib.QmriError:
            pass
        floor = max(int(self.batch_size), 1)
        max_batch = max(floor, min(int(self.gpu_batch), max(1, int(budget // per_slice))))
        tensors = W.to_abi_order(w)
        while True:
            try:
                return _lib.Unet2dEngine(tensors, input_shape[0], input_shape[1], n_classes=n_classes,
                                         max_batch=max_batch, precision=self.precision, device=self.device)
            except _lib.QmriOutOfMemory:  # the library's status code (QMRI_ERR_NOMEM), not the wording of the message
                if max_batch <= floor:
                    raise
                max_batch = max(floor, max_batch // 2)

    def _n_classes(self):
        return 4

    def __del__(self):
        eng = getattr(self, "seg_model", None)
        if eng is not None:
            eng.close()


def whiten_volume(x: np.ndarray, eps: float = 0.0):
    """``(x - mean) / (std + eps)`` over all pixels (reference seg_model.py:114-127).

    Host (numpy) version for callers that want the array; ``generate_mask`` whitens on the GPU."""
    x = np.asarray(x)
    if len(x.shape) != __VOLUME_DIMENSIONS__:
        raise ValueError(f"Input has {x.ndim} dimensions. Expected {__VOLUME_DIMENSIONS__}")
    return (x - np.mean(x)) / (np.std(x) + eps)
