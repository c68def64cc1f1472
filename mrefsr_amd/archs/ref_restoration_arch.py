"""Single-reference predecessor of the MRAPA network (C2-Matching): mirror of
basicsr/archs/ref_restoration_arch.py:101-259.  Same DynAgg / ContentExtractor as the
multi-reference arch; the fusion heads are ``lrelu(conv3x3(cat[x, swapped]))`` instead of MRAPAFusion
(:160-162, :228-229).  Every kernel is shared with ref_mrapa_restoration_arch.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.registry import ARCH_REGISTRY
from .arch_util import ResidualBlockNoBN, conv_act, make_layer, srntt_init_weights
from .ref_mrapa_restoration_arch import ContentExtractor, DynAgg


@ARCH_REGISTRY.register()
class RestorationNet(nn.Module):

    def __init__(self, ngf=64, n_blocks=16, groups=8):
        super().__init__()
        self.content_extractor = ContentExtractor(in_nc=3, out_nc=3, nf=ngf, n_blocks=n_blocks)
        self.dyn_agg_restore = SingleRefDynamicAggregationRestoration(ngf, n_blocks, groups)
        srntt_init_weights(self, init_type='normal', init_gain=0.02)
        self.re_init_dcn_offset()

    def re_init_dcn_offset(self):
        for name in ('small_dyn_agg', 'medium_dyn_agg', 'large_dyn_agg'):
            getattr(self.dyn_agg_restore, name).init_offset()

    def forward(self, x, pre_offset, img_ref_feat):
        """x (B,3,h,w); pre_offset / img_ref_feat: the dicts of CorrespondenceGenerationArch.forward."""
        base = F.interpolate(x, None, 4, 'bilinear', False)
        content_feat = self.content_extractor(x)
        return self.dyn_agg_restore(content_feat, pre_offset, img_ref_feat) + base


class SingleRefDynamicAggregationRestoration(nn.Module):
    """ref_restoration_arch.py:140-259 (class DynamicAggregationRestoration there; state-dict keys
    are attribute names, so the different class name is invisible to checkpoints)."""

    def __init__(self, ngf=64, n_blocks=16, groups=8):
        super().__init__()
        for scale, c in (('small', 256), ('medium', 128), ('large', 64)):
            setattr(self, f'{scale}_offset_conv1', nn.Conv2d(ngf + c, c, 3, 1, 1, bias=True))
            setattr(self, f'{scale}_offset_conv2', nn.Conv2d(c, c, 3, 1, 1, bias=True))
            setattr(self, f'{scale}_dyn_agg', DynAgg(c, c, 3, stride=1, padding=1, dilation=1, deform_groups=groups,
                                                     extra_offset_mask=True))
            setattr(self, f'head_{scale}', nn.Sequential(nn.Conv2d(ngf + c, ngf, kernel_size=3, stride=1, padding=1),
                                                          nn.LeakyReLU(0.1, True)))
            setattr(self, f'body_{scale}', make_layer(ResidualBlockNoBN, n_blocks, num_feat=ngf))
            if scale != 'large':
                setattr(self, f'tail_{scale}', nn.Sequential(nn.Conv2d(ngf, ngf * 4, kernel_size=3, stride=1, padding=1),
                                                              nn.PixelShuffle(2), nn.LeakyReLU(0.1, True)))
        self.tail_large = nn.Sequential(nn.Conv2d(ngf, ngf // 2, kernel_size=3, stride=1, padding=1),
                                        nn.LeakyReLU(0.1, True),
                                        nn.Conv2d(ngf // 2, 3, kernel_size=3, stride=1, padding=1))
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def forward(self, x, pre_offset, img_ref_feat):
        for scale, key in (('small', 'relu3_1'), ('medium', 'relu2_1'), ('large', 'relu1_1')):
            ref = img_ref_feat[key]
            off = conv_act(getattr(self, f'{scale}_offset_conv1'), torch.cat([x, ref], 1), 0.1)
            off = conv_act(getattr(self, f'{scale}_offset_conv2'), off, 0.1)
            swapped = getattr(self, f'{scale}_dyn_agg')([ref, off], pre_offset[key], act_slope=0.1)
            h = conv_act(getattr(self, f'head_{scale}')[0], torch.cat([x, swapped], 1), 0.1)
            h = getattr(self, f'body_{scale}')(h) + x
            if scale == 'large':
                x = conv_act(self.tail_large[2], conv_act(self.tail_large[0], h, 0.1))
            else:
                tail = getattr(self, f'tail_{scale}')
                x = tail[1](conv_act(tail[0], h, 0.1))
        return x
