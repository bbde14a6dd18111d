"""Model registry (reference ``dosma/models/util.py``: ``get_model`` :24-35, ``model_from_config`` :38-94)."""
import os
from typing import Sequence

import yaml

from dosma_amd.models.oaiunet2d import IWOAIOAIUnet2D, IWOAIOAIUnet2DNormalized, OAIUnet2D
from dosma_amd.models.seg_model import SegModel
from dosma_amd.models.stanford_qdess import StanfordQDessUNet2D

__all__ = ["get_model", "model_from_config", "SUPPORTED_MODELS"]

# the reference registers the first three; the SKM-TEA template is reachable there by class only
__SUPPORTED_MODELS__ = [OAIUnet2D, IWOAIOAIUnet2D, IWOAIOAIUnet2DNormalized, StanfordQDessUNet2D]
SUPPORTED_MODELS = [cls.ALIASES[0] for cls in __SUPPORTED_MODELS__]


def get_model(model_str, input_shape, weights_path, **kwargs):
    """Model instance by alias or class name; ``LookupError`` for anything else."""
    by_name = {}
    for cls in __SUPPORTED_MODELS__:
        for key in (*cls.ALIASES, cls.__name__):
            by_name[key] = cls
    if model_str not in by_name:
        raise LookupError("%s model type not supported" % model_str)
    return by_name[model_str](input_shape, weights_path, **kwargs)


def _load_config(cfg_file_or_dict):
    if not isinstance(cfg_file_or_dict, str):
        return cfg_file_or_dict
    with open(cfg_file_or_dict, "r") as f:
        return yaml.safe_load(f)


def _as_volume_list(out):
    if isinstance(out, dict):
        return list(out.values())
    return list(out) if isinstance(out, Sequence) else [out]


def model_from_config(cfg_file_or_dict, weights_dir=None, **kwargs) -> SegModel:
    """Model described by a config ``{DOSMA_MODEL, CATEGORIES, WEIGHTS_FILE}`` (yaml path or dict).  Its
    ``generate_mask`` returns a dict keyed by the config's categories, whatever the base model returns; the weights
    file name is not checked against the base model's (``force_weights=True``), as in the reference."""
    cfg = _load_config(cfg_file_or_dict)
    names = list(cfg["CATEGORIES"])
    weights = cfg["WEIGHTS_FILE"]
    if not isinstance(weights, dict) and not os.path.isfile(weights):
        assert weights_dir, "`weights_dir` must be specified"
        weights = os.path.join(weights_dir, cfg["WEIGHTS_FILE"])
    base = cfg["DOSMA_MODEL"]
    try:
        model = get_model(base, weights_path=weights, force_weights=True, **kwargs)
    except LookupError as e:
        raise LookupError("BASE_MODEL '{}' not supported \n{}".format(base, e))
    predict = model.generate_mask

    def generate_named_masks(*args, **kw):
        vols = _as_volume_list(predict(*args, **kw))
        if len(vols) != len(names):
            raise ValueError("Got {} outputs, but {} categories".format(len(vols), len(names)))
        return dict(zip(names, vols))

    model.generate_mask = generate_named_masks
    return model
