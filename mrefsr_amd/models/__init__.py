from copy import deepcopy

from ..utils.registry import MODEL_REGISTRY
from . import multi_ref_restoration_model  # noqa: F401

__all__ = ['build_model', 'MODEL_REGISTRY']


def build_model(opt):
    """contract of basicsr/models/__init__.py:19-30: opt['model_type'] selects the class"""
    opt = deepcopy(opt)
    return MODEL_REGISTRY.get(opt['model_type'])(opt)
