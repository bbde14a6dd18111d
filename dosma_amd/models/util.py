"""Model registry (reference ``dosma/models/util.py``: ``get_model`` :24-35, ``model_from_config`` :38-94)."""
import os
from functools import partial
from typing import Sequence

import yaml

from dosma_amd.models.oaiunet2d import IWOAIOAIUnet2D, IWOAIOAIUnet2DNormalized, OAIUnet2D
from dosma_amd.models.seg_model import SegModel
from dosma_amd.models.stanford_qdess import StanfordQDessUNet2D

__all__ = ["get_model", "model_from_config", "SUPPORTED_MODELS"]

__SUPPORTED_MODELS__ = [OAIUnet2D, IWOAIOAIUnet2D, IWOAIOAIUnet2DNormalized, StanfordQDessUNet2D]
SUPPORTED_MODELS = [x.ALIASES[0] for x in __SUPPORTED_MODELS__]


def get_model(model_str, input_shape, weights_path, **kwargs):
    for m in __SUPPORTED_MODELS__:
        if model_str in m.ALIASES or model_str == m.__name__:
            return m(input_shape, weights_path, **kwargs)
    raise LookupError("%s model type not supported" % model_str)


def model_from_config(cfg_file_or_dict, weights_dir=None, **kwargs) -> SegModel:
    """Build a model from a config {DOSMA_MODEL, CATEGORIES, WEIGHTS_FILE}; its ``generate_mask`` returns
    a dict keyed by the config's categories (reference :38-94)."""

    def _gen_mask(func, *_args, **_kwargs):
        out = func(*_args, **_kwargs)
        if isinstance(out, dict):
            out = list(out.values())
        elif not isinstance(out, Sequence):
            out = [out]
        if not len(categories) == len(out):
            raise ValueError("Got {} outputs, but {} categories".format(len(out), len(categories)))
        return {cat: o for cat, o in zip(categories, out)}

    if isinstance(cfg_file_or_dict, str):
        with open(cfg_file_or_dict, "r") as f:
            cfg = yaml.safe_load(f)
    else:
        cfg = cfg_file_or_dict
    base_model = cfg["DOSMA_MODEL"]
    categories = cfg["CATEGORIES"]
    weights = cfg["WEIGHTS_FILE"]
    if not isinstance(weights, dict) and not os.path.isfile(weights):
        assert weights_dir, "`weights_dir` must be specified"
        weights = os.path.join(weights_dir, cfg["WEIGHTS_FILE"])
    try:
        model: SegModel = get_model(base_model, weights_path=weights, force_weights=True, **kwargs)
    except LookupError as e:
        raise LookupError("BASE_MODEL '{}' not supported \n{}".format(base_model, e))
    model.generate_mask = partial(_gen_mask, model.generate_mask)
    return model
