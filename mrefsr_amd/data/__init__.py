"""Mirror of the reference's dataset registry for the two datasets of the path
(basicsr/data/__init__.py:25-37 build_dataset; basicsr/data/multi_ref_dataset.py) and the
single-reference twin (basicsr/data/single_ref_dataset.py)."""
from copy import deepcopy

from ..utils.registry import Registry

DATASET_REGISTRY = Registry('dataset')

from . import multi_ref_dataset  # noqa: E402,F401
from . import single_ref_dataset  # noqa: E402,F401

__all__ = ['build_dataset', 'DATASET_REGISTRY']


def build_dataset(dataset_opt):
    """dataset_opt['type'] selects the class; the whole dict is its ``opt`` (reference contract)."""
    dataset_opt = deepcopy(dataset_opt)
    return DATASET_REGISTRY.get(dataset_opt['type'])(dataset_opt)
