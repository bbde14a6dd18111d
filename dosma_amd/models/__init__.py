"""Segmentation models (reference ``dosma/models/__init__.py``)."""
from dosma_amd.models.oaiunet2d import IWOAIOAIUnet2D, IWOAIOAIUnet2DNormalized, OAIUnet2D  # noqa: F401
from dosma_amd.models.stanford_qdess import StanfordQDessUNet2D  # noqa: F401
from dosma_amd.models.seg_model import SegModel, whiten_volume  # noqa: F401
from dosma_amd.models.util import SUPPORTED_MODELS, get_model, model_from_config  # noqa: F401
