"""SKM-TEA / 2021 Stanford qDESS knee 2D U-Net on the MI355X kernels.

Mirror of the reference's ``dosma/models/stanford_qdess.py:27-205``: the same 6-level graph as
``OAIUnet2D`` (:64-156) with a 4-class sigmoid head, classes in the order **pc, fc, tc, men** (:191), input
whitened with ``eps=1e-8`` (:203-205).  A 4D ``(..., 2)`` dual-echo volume is first reduced to the
root-sum-of-squares of the two echoes (:177-178) -- on the GPU (``qmri_rss_host``)."""
import numpy as np

from dosma_amd import _lib
from dosma_amd.fitting import _as_kernel_samples
from dosma_amd.med_volume import MedicalVolume
from dosma_amd.models.oaiunet2d import IWOAIOAIUnet2D

__all__ = ["StanfordQDessUNet2D"]


class StanfordQDessUNet2D(IWOAIOAIUnet2D):
    ALIASES = ("stanford-qdess-2021-unet2d", "skm-tea-unet2d")
    CATEGORIES = ("pc", "fc", "tc", "men")
    _WHITEN = True
    _WHITEN_EPS = 1e-8
    sigmoid_threshold = 0.5

    def __init__(self, input_shape, weights_path, force_weights=False):
        # the reference takes any weights file for this template (:64, no name check)
        super().__init__(input_shape, weights_path, force_weights=True)

    def generate_mask(self, volume: MedicalVolume):
        """3D volume = the RSS of the two qDESS echoes; 4D ``(..., 2)`` = echo 1 and echo 2 (reference :158-201)."""
        if not isinstance(volume, MedicalVolume) or volume.ndim not in (3, 4):
            raise ValueError("`volume` must either be 3D or 4D")
        if volume.ndim == 4:
            arr = volume.volume
            if arr.shape[-1] != 2:
                raise ValueError("4D volumes must have shape (..., 2): echo 1 and echo 2")
            # any real dtype, like the reference's np.sqrt(np.sum(v ** 2)) (:177-178): the echoes are widened to a
            # dtype the kernel reads, exactly as QDess._combine_echoes does
            e1 = _as_kernel_samples(np.ascontiguousarray(arr[..., 0]))
            e2 = _as_kernel_samples(np.ascontiguousarray(arr[..., 1]))
            if e1.dtype != e2.dtype:
                e1, e2 = e1.astype(np.float64), e2.astype(np.float64)
            rss = _lib.rss_host(e1, e2, method="rss")
            headers = volume.headers()
            volume = volume._partial_clone(volume=rss, headers=None if headers is None else headers[..., 0])
        return super().generate_mask(volume)

    def __preprocess_volume__(self, volume: np.ndarray):
        from dosma_amd.models.seg_model import whiten_volume

        return whiten_volume(volume, eps=1e-8)
