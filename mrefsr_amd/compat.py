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
        for modname, mod in (('basicsr.ops.dcn', dcn), ('basicsr.ops.fused_act', fused_act),
                             ('basicsr.ops.upfirdn2d', upfirdn2d)):
            sys.modules[modname] = mod
    if mmcv and 'mmcv.ops' not in sys.modules:
        # the reference's arch file imports two names from mmcv.ops (ref_mrapa_restoration_arch.py:5)
        from .archs.ref_mrapa_restoration_arch import DynAgg  # noqa: F401
        from .ops.dcn import ModulatedDeformConv, modulated_deform_conv
        shim = types.ModuleType('mmcv.ops')
        shim.ModulatedDeformConv2d = ModulatedDeformConv
        shim.modulated_deform_conv2d = modulated_deform_conv
        pkg = sys.modules.setdefault('mmcv', types.ModuleType('mmcv'))
        pkg.ops = shim
        sys.modules['mmcv.ops'] = shim
    return True
