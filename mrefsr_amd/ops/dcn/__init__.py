# same export list as basicsr/ops/dcn/__init__.py:1-7
from .deform_conv import (DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack, deform_conv,
                          modulated_deform_conv)

__all__ = [
    'DeformConv', 'DeformConvPack', 'ModulatedDeformConv', 'ModulatedDeformConvPack', 'deform_conv',
    'modulated_deform_conv'
]

from . import deform_conv_ext  # noqa: E402,F401  (name-compatible stand-in for the reference's pybind module)
