"""Drop the MI355X path into an existing ``basicsr`` checkout (the reference) without editing its
train.py / test.py: replace the registry entries of the hot-path classes and the ``basicsr.ops``
native modules by their mrefsr_amd counterparts.  See INTEGRATION.md.

    import basicsr            # the reference package (already importable in that environment)
    import mrefsr_amd.compat as compat
    compat.install_into_basicsr()
    # from here on  build_network({'type': 'MRAPARestorationNet', ...})  builds the HIP-backed arch
"""
import sys
import types

_ARCHS = ('MRAPARestorationNet', 'RestorationNet', 'CorrespondenceGenerationArch', 'ContrasMultiExtractorSep',
          'ContrasExtractorSep', 'VGGFeatureExtractor')
_MODELS = ('MultiRefRestorationModel', 'RefRestorationModel')


def _replace(registry, name, obj):
    """registries refuse duplicate names (basicsr/utils/registry.py:42); overwrite the slot"""
    registry._obj_map[name] = obj


def install_into_basicsr(models=True, ops=True, mmcv=True):
    from basicsr.utils.registry import ARCH_REGISTRY as REF_ARCH, MODEL_REGISTRY as REF_MODEL

    from .archs import ARCH_REGISTRY
    from .models import MODEL_REGISTRY
    for name in _ARCHS:
        _replace(REF_ARCH, name, ARCH_REGISTRY.get(name))
    if models:
        for name in _MODELS:
            _replace(REF_MODEL, name, MODEL_REGISTRY.get(name))
    if ops:
        from .ops import dcn, fused_act, upfirdn2d
        from .ops.dcn import deform_conv_ext
        for modname, mod in (('basicsr.ops.dcn', dcn), ('basicsr.ops.fused_act', fused_act),
                             ('basicsr.ops.upfirdn2d', upfirdn2d), ('basicsr.ops.dcn.deform_conv_ext', deform_conv_ext)):
            sys.modules[modname] = mod
    if mmcv and 'mmcv.ops' not in sys.modules:
        # the reference's arch files import two names from mmcv.ops (ref_mrapa_restoration_arch.py:5, ref_restoration_arch.py:5)
        # and subclass one of them (DynAgg(ModulatedDeformConv2d), :11-29): mmcv's attribute is `deform_groups`
        shim = types.ModuleType('mmcv.ops')
        shim.ModulatedDeformConv2d = ModulatedDeformConv2d
        shim.modulated_deform_conv2d = modulated_deform_conv2d
        pkg = sys.modules.setdefault('mmcv', types.ModuleType('mmcv'))
        pkg.ops = shim
        sys.modules['mmcv.ops'] = shim
    return True


def modulated_deform_conv2d(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deform_groups=1):
    """mmcv.ops.modulated_deform_conv2d as the reference calls it (ref_mrapa_restoration_arch.py:74-76) on the HIP DCNv2"""
    from .ops.dcn import modulated_deform_conv
    return modulated_deform_conv(input, offset, mask, weight, bias, stride, padding, dilation, groups, deform_groups)


def _make_mmcv_module():
    import math

    import torch
    from torch import nn

    class ModulatedDeformConv2d(nn.Module):
        """constructor, parameters and attributes of mmcv.ops.ModulatedDeformConv2d that the reference's DynAgg relies on
        (super().__init__(in, out, k, stride, padding, dilation, groups, deform_groups); self.deform_groups, .kernel_size
        (pair), .stride, .padding, .dilation, .groups, .weight, .bias); init as the vendored twin deform_conv.py:322-329"""

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deform_groups=1, bias=True):
            super().__init__()
            self.in_channels, self.out_channels = in_channels, out_channels
            self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
            self.stride, self.padding, self.dilation = stride, padding, dilation
            self.groups, self.deform_groups = groups, deform_groups
            self.transposed, self.output_padding = False, (0,)
            self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
            self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
            stdv = 1.0 / math.sqrt(in_channels * self.kernel_size[0] * self.kernel_size[1])
            with torch.no_grad():
                self.weight.uniform_(-stdv, stdv)

        def forward(self, x, offset, mask):
            return modulated_deform_conv2d(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                           self.groups, self.deform_groups)

    return ModulatedDeformConv2d


ModulatedDeformConv2d = _make_mmcv_module()
