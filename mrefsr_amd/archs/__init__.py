"""Mirror of the reference's basicsr.archs for the modules on the hot path.  Classes register
themselves in ARCH_REGISTRY under the reference's names; ``build_network`` has the contract of
basicsr/archs/__init__.py:19-25 (``type`` selects the class, the other keys are its kwargs)."""
from copy import deepcopy

from ..utils.registry import ARCH_REGISTRY
from . import (contras_multi_extractor_arch, corres_generation_arch, ref_mrapa_restoration_arch,  # noqa: F401
               ref_restoration_arch, vgg_arch)

__all__ = ['build_network', 'ARCH_REGISTRY']


def build_network(opt):
    opt = deepcopy(opt)
    network_type = opt.pop('type')
    return ARCH_REGISTRY.get(network_type)(**opt)
