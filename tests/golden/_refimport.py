"""Import the reference's hot-path Python (read-only, /root/reference) in the BUILD container.

Only tests/golden/gen_golden.py uses this, and only where /root/reference exists; nothing that
runs on the GPU box imports it.  The reference package cannot be imported whole
(basicsr/__init__.py needs a generated version.py and star-imports cv2 / torchvision / mmcv
users), so:
  * a bare ``basicsr`` package object is registered whose __path__ points at the reference tree
    (its __init__ is skipped);
  * cv2, lmdb are MagicMock modules (never called on this path);
  * ``torchvision.models.vgg.vgg16/vgg19`` are replaced by the public VGG layer stacks with
    caller-supplied weights (torchvision itself is absent from the image);
  * ``mmcv.ops.ModulatedDeformConv2d / modulated_deform_conv2d`` (third party, un-vendored,
    unpinned) are bound to oracle/dcn_torch.py, the restatement of the vendored CUDA spec.
No reference source is copied; bytecode writing is disabled.
"""
import importlib
import os
import sys
import types
from unittest import mock

sys.dont_write_bytecode = True
REF_ROOT = os.environ.get('MREFSR_REFERENCE', '/root/reference')
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'basicsr', 'archs'))


_VGG_CFG = {
    'vgg16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
    'vgg19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M'],
}


def _make_vgg(name):
    import torch
    from torch import nn

    def ctor(pretrained=False, **kw):
        layers, cin = [], 3
        # only the first three stages are ever sliced by the path (conv3_1 / relu3_1)
        for v in _VGG_CFG[name][:10]:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        net = nn.Module()
        net.features = nn.Sequential(*layers)
        # deterministic, scale-preserving init (no download possible): kaiming-normal, seeded
        g = torch.Generator().manual_seed(1234 if name == 'vgg16' else 4321)
        for m in net.features:
            if isinstance(m, nn.Conv2d):
                fan_in = m.weight.shape[1] * 9
                m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.01
        return net

    return ctor


def install():
    """Register stubs + the bare basicsr package.  Idempotent."""
    if 'basicsr' in sys.modules and getattr(sys.modules['basicsr'], '_mrefsr_ref_stub', False):
        return
    assert available(), f'{REF_ROOT} not present'
    import torch  # noqa: F401
    from torch import nn

    for name in ('cv2', 'lmdb'):
        sys.modules[name] = mock.MagicMock(name=name)
    # the three cv2 calls and the one mmcv call the dataset path makes (multi_ref_dataset.py:158-180,
    # img_util.py:26, transforms.py:121-124), restated with numpy / PIL so the reference's own
    # dataset classes can run here: imread -> BGR uint8; cvtColor(BGR2RGB) -> channel reversal;
    # flip(img, code, img) in place; mmcv.impad pads bottom / right
    import numpy as _np
    from PIL import Image as _Image
    cv2 = sys.modules['cv2']
    cv2.COLOR_BGR2RGB = 4
    cv2.imread = lambda p, *a: _np.ascontiguousarray(_np.array(_Image.open(p).convert('RGB'))[:, :, ::-1])
    cv2.cvtColor = lambda img, code: _np.ascontiguousarray(img[:, :, ::-1])

    def _flip(src, code, dst=None):
        out = src[:, ::-1].copy() if code == 1 else src[::-1].copy()
        if dst is not None:
            dst[...] = out
            return dst
        return out
    cv2.flip = _flip

    tv = types.ModuleType('torchvision')
    tv.__version__ = '0.0.0'
    tv.__path__ = []
    tvm = types.ModuleType('torchvision.models')
    tvm.__path__ = []
    tvv = types.ModuleType('torchvision.models.vgg')
    tvv.vgg16 = _make_vgg('vgg16')
    tvv.vgg19 = _make_vgg('vgg19')
    tvm.vgg = tvv
    tv.models = tvm
    tv.transforms = mock.MagicMock(name='torchvision.transforms')
    sys.modules['torchvision.transforms'] = tv.transforms
    tvu = mock.MagicMock(name='torchvision.utils')
    tvo = mock.MagicMock(name='torchvision.ops')
    sys.modules.update({'torchvision': tv, 'torchvision.models': tvm, 'torchvision.models.vgg': tvv,
                        'torchvision.utils': tvu, 'torchvision.ops': tvo})

    from oracle import dcn_torch

    class ModulatedDeformConv2d(nn.Module):
        """Attribute surface of mmcv.ops.ModulatedDeformConv2d used by DynAgg
        (ref_mrapa_restoration_arch.py:11-43); init = vendored twin ops/dcn/deform_conv.py:322-329."""

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                     groups=1, deform_groups=1, bias=True):
            super().__init__()
            import math
            import torch
            self.in_channels, self.out_channels = in_channels, out_channels
            self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
            self.stride, self.padding, self.dilation = stride, padding, dilation
            self.groups, self.deform_groups = groups, deform_groups
            self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
            self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
            stdv = 1.0 / math.sqrt(in_channels * self.kernel_size[0] * self.kernel_size[1])
            self.weight.data.uniform_(-stdv, stdv)

    mm = types.ModuleType('mmcv')
    mm.__path__ = []
    mmo = types.ModuleType('mmcv.ops')
    mmo.ModulatedDeformConv2d = ModulatedDeformConv2d
    mmo.modulated_deform_conv2d = dcn_torch.modulated_deform_conv2d
    mm.ops = mmo

    def _impad(img, shape=None, pad_val=0, **kw):
        import numpy as np
        h, w = img.shape[:2]
        out = np.full((max(shape[0], h), max(shape[1], w)) + img.shape[2:], pad_val, dtype=img.dtype)
        out[:h, :w] = img
        return out
    mm.impad = _impad
    sys.modules.update({'mmcv': mm, 'mmcv.ops': mmo})

    pkg = types.ModuleType('basicsr')
    pkg.__path__ = [os.path.join(REF_ROOT, 'basicsr')]
    pkg._mrefsr_ref_stub = True
    sys.modules['basicsr'] = pkg
    for sub in ('archs', 'ops', 'utils', 'metrics', 'models', 'data'):
        m = types.ModuleType(f'basicsr.{sub}')
        m.__path__ = [os.path.join(REF_ROOT, 'basicsr', sub)]
        sys.modules[f'basicsr.{sub}'] = m
        setattr(pkg, sub, m)
    # basicsr.utils exports used by the arch files
    reg = importlib.import_module('basicsr.utils.registry')
    sys.modules['basicsr.utils'].registry = reg
    import logging
    sys.modules['basicsr.utils'].get_root_logger = lambda *a, **k: logging.getLogger('basicsr')
    # basicsr.ops.dcn is imported by arch_util.py:13; its CUDA ext is absent -> names only
    dcn = types.ModuleType('basicsr.ops.dcn')
    dcn.ModulatedDeformConvPack = object
    dcn.modulated_deform_conv = None
    sys.modules['basicsr.ops.dcn'] = dcn
    # re-export what the skipped package __init__ files would have (utils/__init__.py:1-7,
    # metrics/__init__.py, archs/__init__.py:19-25) by importing the reference's own submodules
    u = sys.modules['basicsr.utils']
    for sub, names in (('color_util', ('bgr2ycbcr', 'rgb2ycbcr', 'rgb2ycbcr_pt')),
                       ('img_util', ('tensor2img', 'imwrite', 'img2tensor', 'crop_border', 'imfrombytes')),
                       ('file_client', ('FileClient',)),
                       ('misc', ('ProgressBar', 'scandir', 'set_random_seed')),
                       ('logger', ('get_root_logger', 'MessageLogger', 'AvgTimer'))):
        mod = importlib.import_module(f'basicsr.utils.{sub}')
        for n in names:
            setattr(u, n, getattr(mod, n))
    met = importlib.import_module('basicsr.metrics.psnr_ssim')
    sys.modules['basicsr.metrics'].calculate_psnr = met.calculate_psnr
    sys.modules['basicsr.metrics'].calculate_ssim = met.calculate_ssim

    def build_network(opt):  # archs/__init__.py:19-25 (the package __init__ itself cannot be imported)
        from copy import deepcopy
        opt = deepcopy(opt)
        return reg.ARCH_REGISTRY.get(opt.pop('type'))(**opt)

    sys.modules['basicsr.archs'].build_network = build_network
    for m in ('ref_mrapa_restoration_arch', 'ref_restoration_arch', 'corres_generation_arch', 'contras_multi_extractor_arch',
              'contras_extractor_arch', 'vgg_arch'):
        importlib.import_module(f'basicsr.archs.{m}')


def ref_module(name):
    install()
    return importlib.import_module(name)
