"""Single-reference predecessor of the MRAPA network (C2-Matching): mirror of
basicsr/archs/ref_restoration_arch.py:101-259.  Same DynAgg / ContentExtractor as the
multi-reference arch; the fusion heads are ``lrelu(conv3x3(cat[x, swapped]))`` instead of MRAPAFusion
(:160-162, :228-229).  Every kernel is shared with ref_mrapa_restoration_arch.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.registry import ARCH_REGISTRY
from . import nhwc
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
        if nhwc.BF16 and nhwc.active(x):
            x = x.bfloat16().float()
        base = F.interpolate(x, None, 4, 'bilinear', False)
        if (nhwc.active(x) or nhwc.train_active(x)) and self.dyn_agg_restore.nhwc_ok():
            # channels-last engine (archs/nhwc.py; under autograd archs/nhwc_train.py), as in MRAPARestorationNet
            ce = self.content_extractor
            feat = nhwc.res_chain(ce.body, nhwc.conv(ce.conv_first, nhwc.image_to_nhwc4(x), slope=0.1))
            refs = {key: nhwc.to_nhwc(v if v.dtype == feat.dtype else v.to(feat.dtype)) for key, v in img_ref_feat.items()}
            out = self.dyn_agg_restore.forward_nhwc(feat, pre_offset, refs)
            return nhwc.rnd_((nhwc.as_nchw(out).float() + nhwc.rnd_(base)).contiguous())
        img_ref_feat = {key: v.contiguous() for key, v in img_ref_feat.items()}
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

    def nhwc_ok(self):
        ngf = self.small_offset_conv1.in_channels - 256      # first source of the two-source convolutions
        return ngf % 16 == 0 and all(getattr(self, f'{s}_dyn_agg').nhwc_ok() for s in ('small', 'medium', 'large'))

    def forward_nhwc(self, x, pre_offset, ref_feat):
        """channels-last inference form: x [B,h,w,ngf], ref_feat values [B,H,W,C] -> [B,4h,4w,3];
        torch.cat([x, ref]) of :217 / :228 = two-source convolutions"""
        for scale, key in (('small', 'relu3_1'), ('medium', 'relu2_1'), ('large', 'relu1_1')):
            ref = ref_feat[key]
            off = nhwc.conv(getattr(self, f'{scale}_offset_conv1'), x, x2=ref, slope=0.1)
            off = nhwc.conv(getattr(self, f'{scale}_offset_conv2'), off, slope=0.1)
            swapped = getattr(self, f'{scale}_dyn_agg').forward_nhwc(ref, off, pre_offset[key], act_slope=0.1)
            h = nhwc.conv(getattr(self, f'head_{scale}')[0], x, x2=swapped, slope=0.1)
            h = nhwc.res_chain(getattr(self, f'body_{scale}'), h)
            h = h + x if h.requires_grad else nhwc.rnd_(h.add_(x))
            if scale == 'large':
                return nhwc.conv(self.tail_large[2], nhwc.conv(self.tail_large[0], h, slope=0.1))
            x = nhwc.conv(getattr(self, f'tail_{scale}')[0], h, slope=0.1, epilogue=2)

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
