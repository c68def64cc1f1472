"""Mirror of basicsr/archs/ref_mrapa_restoration_arch.py (DynAgg :11-76, ContentExtractor :79-98,
MRAPARestorationNet :101-137, DynamicAggregationRestoration :140-259, MRAPAFusion :262-348) with
identical constructor arguments, forward signatures and state-dict keys (23,711,633 parameters).

What is different underneath:
  * DynAgg: chunk / cat / repeat / re-order / add / sigmoid / mean-abs (:56-73) are one HIP pass
    (mrefsr_dynagg_prep_f32); the `.mean() > 100` host sync per call becomes a device-side
    accumulator read only when asked (DynAgg.offset_guard()).  The DCN itself is the fused
    gather+MFMA kernel of csrc/dcn.hip (the reference needs mmcv here), with the LeakyReLU that
    always follows it fused into the epilogue.
  * DynamicAggregationRestoration: the python loop over the K references (:216/:231/:246) runs
    as ONE batch of K*B images per layer; the x-half of offset_conv1 (shared by all K
    references) is computed once instead of K times.
  * MRAPAFusion: the three permute().contiguous() copies + two bmm + softmax (:321-335) are one
    HIP kernel reading NCHW in place (mrefsr_mrattn_fwd_f32 / _bwd_f32).
  * Inference (no autograd): the whole network runs channels-last on the fp32-equivalent convolutions
    of csrc/conv_nhwc.hip / conv_wino.hip (fp16 two-term split, direct or Winograd F(2x2, 3x3) form;
    archs/nhwc.py): torch.cat, bias, LeakyReLU / PReLU, residual adds, MaxPool and PixelShuffle are
    convolution epilogues / prologues, the attention core and the DCN read and write [N,H,W,C] directly.
  * Training (autograd on net_g): the same channels-last storage, one custom autograd node per fused
    launch (archs/nhwc_train.py): forward and input-gradient convolutions on the same kernels, weight
    gradients in csrc/wgrad.hip, the DCN backward in csrc/dcn_bwd.hip; generic torch forms only for what the engine refuses.
"""
import logging
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import hip
from ..ops.dcn import modulated_deform_conv
from ..utils.registry import ARCH_REGISTRY
from . import nhwc, nhwc_train
from .arch_util import ResidualBlockNoBN, conv_act, default_init_weights, make_layer, srntt_init_weights

TAIL_FUSED = os.environ.get('MREFSR_TAIL_FUSED', '1') != '0'   # (0: the ATen tail, for A/B)


class _DynAggPrep(Function):
    """(conv_offset_mask output, pre_offset) -> (offset, mask)   ref :56-69"""

    @staticmethod
    def forward(ctx, om, pre_offset, dg, abs_sum, om_bias=None):
        om = om.contiguous()
        offset, mask = hip.dynagg_prep(om, pre_offset.contiguous(), dg, abs_sum, om_bias)
        ctx.dg = dg
        ctx.save_for_backward(mask)
        return offset, mask

    @staticmethod
    @once_differentiable
    def backward(ctx, g_offset, g_mask):
        mask, = ctx.saved_tensors
        return hip.dynagg_prep_bwd(g_offset.contiguous(), g_mask.contiguous(), mask, ctx.dg), None, None, None, None


class _MultiRefAttention(Function):
    """softmax_t(<q, emb_t>) . ass_t per pixel   ref :321-335"""

    @staticmethod
    def forward(ctx, q, emb, ass, t, t_major):
        q, emb, ass = q.contiguous(), emb.contiguous(), ass.contiguous()
        need = q.requires_grad or emb.requires_grad or ass.requires_grad
        out, prob = hip.mrattn_fwd(q, emb, ass, t, want_prob=need, t_major=t_major)
        ctx.t, ctx.t_major = t, t_major
        if need:
            ctx.save_for_backward(q, emb, ass, prob)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        q, emb, ass, prob = ctx.saved_tensors
        g_q, g_emb, g_ass = hip.mrattn_bwd(q, emb, ass, prob, g_out.contiguous(), ctx.t, ctx.t_major)
        return g_q, g_emb, g_ass, None, None


class DynAgg(nn.Module):
    """Modulated deformable conv whose offsets are initialised with the pre-computed
    correspondence offsets (ref :11-76).  Attribute surface of mmcv's ModulatedDeformConv2d
    (weight (Co,Ci/g,k,k), bias, kernel_size tuple, deform_groups) so checkpoints and callers
    carry over; parameter init = uniform(+-1/sqrt(Ci*k*k)), bias 0 (vendored twin
    ops/dcn/deform_conv.py:322-329)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1,
                 deform_groups=1, extra_offset_mask=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deform_groups = groups, deform_groups
        self.transposed, self.output_padding = False, (0,)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        stdv = 1. / (in_channels * self.kernel_size[0] * self.kernel_size[1]) ** 0.5
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
        self.extra_offset_mask = extra_offset_mask
        channels_ = self.deform_groups * 3 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(self.in_channels, channels_, kernel_size=self.kernel_size, stride=self.stride,
                                          padding=self.padding, bias=True)
        self.init_offset()
        # device-side |learned offset| accumulator (sum, element count): replaces the per-call `offset_mean > 100` host sync
        # of ref :70-73.  A plain tensor attribute that follows the module across devices (_apply), NOT a registered buffer:
        # DistributedDataParallel broadcasts buffers from rank 0 on every forward (a collective the reference does not
        # have, and it would overwrite the other ranks' statistics).
        self._offset_abs_sum = torch.zeros(1, dtype=torch.float64)
        self._offset_count = 0

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        moved = fn(self._offset_abs_sum)
        self._offset_abs_sum = moved.double() if moved.dtype != torch.float64 else moved   # .half() / .float() must not touch it
        return self

    def init_offset(self):
        with torch.no_grad():
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def offset_guard(self, reset=True):
        """mean |learned offset| since the last reset; logs the reference's warning if > 100.
        (One host sync, paid only by the caller who asks.)"""
        if self._offset_count == 0:
            return 0.0
        mean = float(self._offset_abs_sum.item()) / self._offset_count
        if mean > 100:
            logging.getLogger('basicsr').warning('Offset mean is {}, larger than 100.'.format(mean))
        if reset:
            self._offset_abs_sum.zero_()
            self._offset_count = 0
        return mean

    def forward(self, x, pre_offset, act_slope=1.0):
        """x = [input, features] (extra_offset_mask) or a tensor; pre_offset [b,9,h,w,2] ([x,y]).
        ``act_slope`` != 1 fuses the following LeakyReLU (extension; default = reference)."""
        feat = x[1] if self.extra_offset_mask else x
        if self.extra_offset_mask:
            x = x[0]
        if self.kernel_size != (3, 3):
            raise NotImplementedError('DynAgg: the pre-offset injection assumes a 3x3 kernel (9 taps), as the reference')
        com = self.conv_offset_mask
        if torch.is_grad_enabled() and (feat.requires_grad or com.weight.requires_grad):
            offset, mask = _DynAggPrep.apply(com(feat), pre_offset, self.deform_groups, self._offset_abs_sum)
        else:  # inference: convolution without bias, the bias is added inside the glue kernel (one pass less)
            out = F.conv2d(feat, com.weight, None, com.stride, com.padding)
            offset, mask = _DynAggPrep.apply(out, pre_offset, self.deform_groups, self._offset_abs_sum, com.bias)
        self._offset_count += offset.numel()
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                     self.groups, self.deform_groups, act_slope)


    def forward_nhwc(self, x, feat, pre_offset, act_slope=1.0):
        """channels-last inference form: x (sampled features) [B,H,W,C], feat [B,H,W,C] -> [B,H,W,Co]"""
        # conv_offset_mask with the glue of :56-73 as its epilogue: planar offset / mask straight out of the convolution
        com = self.conv_offset_mask
        if nhwc_train.recording(com.weight, self.weight, feat, x):   # training: the same two launches as autograd nodes
            offset, mask = nhwc_train.conv_dynagg(feat, com.weight, com.bias, pre_offset.contiguous(), self.deform_groups, self._offset_abs_sum)
            self._offset_count += offset.numel()
            return nhwc_train.dcn(x, offset, mask, self.weight, self.bias, self.deform_groups, act_slope)
        terms = 6 if (nhwc.TERMS == 16 and hip.is_range_free()) else nhwc.TERMS
        offset, mask = hip.conv_dynagg(feat, hip.packed_weight(com.weight, None, terms), com.bias.detach(), pre_offset.contiguous(),
                                       self.deform_groups, self._offset_abs_sum)
        self._offset_count += offset.numel()
        slot = hip.amax_slot(x.device) if (x.dtype == torch.float32 and not nhwc.BF16 and nhwc.WINO_INSCALE) else None
        y = hip.dcn_fwd(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups,
                        self.deform_groups, act_slope, channels_last=True, bf16_arith=nhwc.BF16, out_amax=slot)
        if slot is not None:   # (max |out| from the kernel's epilogue: the input scale of MRAPAFusion's 3x3 convolutions)
            setattr(y, nhwc.AMAX_ATTR, slot)
        return y

    def nhwc_ok(self):
        return (self.kernel_size == (3, 3) and self.stride == 1 and self.padding == 1 and self.dilation == 1 and self.groups == 1
                and self.extra_offset_mask and hip.dcn_mfma_eligible(self.in_channels, self.out_channels, self.deform_groups))


class ContentExtractor(nn.Module):

    def __init__(self, in_nc=3, out_nc=3, nf=64, n_blocks=16):
        super().__init__()
        self.conv_first = nn.Conv2d(in_nc, nf, 3, 1, 1)
        self.body = make_layer(ResidualBlockNoBN, n_blocks, num_feat=nf)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        default_init_weights([self.conv_first], 0.1)

    def forward(self, x):
        return self.body(conv_act(self.conv_first, x, 0.1))


def _stack_refs(pre_offset_list, img_ref_feat_list):
    """reference API (K dicts of [B,...]) -> dicts of k-major stacked [K*B,...] tensors"""
    k = len(pre_offset_list)
    keys = ('relu3_1', 'relu2_1', 'relu1_1')
    if k == 1:
        return pre_offset_list[0], img_ref_feat_list[0], 1
    pre = {key: torch.cat([p[key] for p in pre_offset_list], 0) for key in keys}
    feat = {key: torch.cat([f[key] for f in img_ref_feat_list], 0) for key in keys}
    return pre, feat, k


@ARCH_REGISTRY.register()
class MRAPARestorationNet(nn.Module):

    def __init__(self, ngf=64, n_blocks=16, groups=8):
        super().__init__()
        self.content_extractor = ContentExtractor(in_nc=3, out_nc=3, nf=ngf, n_blocks=n_blocks)
        self.dyn_agg_restore = DynamicAggregationRestoration(ngf, n_blocks, groups)
        srntt_init_weights(self, init_type='normal', init_gain=0.02)
        self.re_init_dcn_offset()

    def re_init_dcn_offset(self):
        for name in ('small_dyn_agg', 'medium_dyn_agg', 'large_dyn_agg'):
            getattr(self.dyn_agg_restore, name).init_offset()

    def forward(self, x, pre_offset_list, img_ref_feat_list, k=None):
        """x (B,3,h,w); pre_offset_list / img_ref_feat_list: K dicts as produced by
        CorrespondenceGenerationArch.forward -> (B,3,4h,4w)   (reference signature).
        With ``k`` given, the two arguments are single dicts of k-major stacked [K*B,...] tensors
        (the batched path; goes through forward() so DistributedDataParallel hooks still run)."""
        if k is not None:
            return self.forward_stacked(x, pre_offset_list, img_ref_feat_list, k)
        pre, feat, k = _stack_refs(pre_offset_list, img_ref_feat_list)
        return self.forward_stacked(x, pre, feat, k)

    def forward_stacked(self, x, pre_offset, img_ref_feat, k):
        """same with the K references already stacked k-major on the batch axis ([K*B,...])."""
        if nhwc.BF16 and nhwc.active(x):
            x = x.bfloat16().float()
        if (nhwc.active(x) or (nhwc.train_active(x) and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0)) and self.dyn_agg_restore.nhwc_ok(x):
            ce = self.content_extractor
            feat = nhwc.res_chain(ce.body, nhwc.conv(ce.conv_first, nhwc.image_to_nhwc4(x), slope=0.1))
            refs = {key: nhwc.to_nhwc(v if v.dtype == feat.dtype else v.to(feat.dtype)) for key, v in img_ref_feat.items()}
            out = self.dyn_agg_restore.forward_nhwc(feat, pre_offset, refs, k)
            if TAIL_FUSED and not nhwc.BF16 and out.dtype == torch.float32 and x.dtype == torch.float32 and not (out.requires_grad or x.requires_grad):
                return hip.tail_bilinear_add(out, x, 4)   # F.interpolate + add + NCHW copy of :132-137 in one pass (torch's interpolation bits)
            base = F.interpolate(x, None, 4, 'bilinear', False)
            return nhwc.rnd_((nhwc.as_nchw(out).float() + nhwc.rnd_(base)).contiguous())
        base = F.interpolate(x, None, 4, 'bilinear', False)
        # autograd / MIOpen path: NCHW storage (the frozen VGG taps arrive as channels-last views)
        img_ref_feat = {key: v.contiguous() for key, v in img_ref_feat.items()}
        content_feat = self.content_extractor(x)
        return self.dyn_agg_restore.forward_stacked(content_feat, pre_offset, img_ref_feat, k) + base


class DynamicAggregationRestoration(nn.Module):

    def __init__(self, ngf=64, n_blocks=16, groups=8):
        super().__init__()
        self.ngf = ngf
        # relu3_1 scale
        self.small_offset_conv1 = nn.Conv2d(ngf + 256, 256, 3, 1, 1, bias=True)
        self.small_offset_conv2 = nn.Conv2d(256, 256, 3, 1, 1, bias=True)
        self.small_dyn_agg = DynAgg(256, 256, 3, stride=1, padding=1, dilation=1, deform_groups=groups,
                                    extra_offset_mask=True)
        self.head_small = MRAPAFusion(nf=ngf, ref_nf=256)
        self.body_small = make_layer(ResidualBlockNoBN, n_blocks, num_feat=ngf)
        self.tail_small = nn.Sequential(nn.Conv2d(ngf, ngf * 4, kernel_size=3, stride=1, padding=1), nn.PixelShuffle(2),
                                        nn.LeakyReLU(0.1, True))
        # relu2_1 scale
        self.medium_offset_conv1 = nn.Conv2d(ngf + 128, 128, 3, 1, 1, bias=True)
        self.medium_offset_conv2 = nn.Conv2d(128, 128, 3, 1, 1, bias=True)
        self.medium_dyn_agg = DynAgg(128, 128, 3, stride=1, padding=1, dilation=1, deform_groups=groups,
                                     extra_offset_mask=True)
        self.head_medium = MRAPAFusion(nf=ngf, ref_nf=128)
        self.body_medium = make_layer(ResidualBlockNoBN, n_blocks, num_feat=ngf)
        self.tail_medium = nn.Sequential(nn.Conv2d(ngf, ngf * 4, kernel_size=3, stride=1, padding=1), nn.PixelShuffle(2),
                                         nn.LeakyReLU(0.1, True))
        # relu1_1 scale
        self.large_offset_conv1 = nn.Conv2d(ngf + 64, 64, 3, 1, 1, bias=True)
        self.large_offset_conv2 = nn.Conv2d(64, 64, 3, 1, 1, bias=True)
        self.large_dyn_agg = DynAgg(64, 64, 3, stride=1, padding=1, dilation=1, deform_groups=groups,
                                    extra_offset_mask=True)
        self.head_large = MRAPAFusion(nf=ngf, ref_nf=64)
        self.body_large = make_layer(ResidualBlockNoBN, n_blocks, num_feat=ngf)
        self.tail_large = nn.Sequential(nn.Conv2d(ngf, ngf // 2, kernel_size=3, stride=1, padding=1),
                                        nn.LeakyReLU(0.1, True),
                                        nn.Conv2d(ngf // 2, 3, kernel_size=3, stride=1, padding=1))
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def _swap(self, x, ref_feat, pre_offset, k, conv1, conv2, dyn_agg):
        """K references at once: lrelu(conv1(cat[x, ref])) -> lrelu(conv2) -> lrelu(DynAgg) (ref
        :217-225).  conv1 over cat[x, ref] == conv(x; W[:, :ngf]) + conv(ref; W[:, ngf:]) + b; the
        x half does not depend on the reference and is computed once."""
        b = x.shape[0]
        wx, wr = conv1.weight[:, :self.ngf], conv1.weight[:, self.ngf:]
        ox = F.conv2d(x, wx, None, 1, 1)
        if torch.is_grad_enabled() and (x.requires_grad or conv1.weight.requires_grad):
            orf = F.conv2d(ref_feat, wr, conv1.bias, 1, 1)
            off = F.leaky_relu((orf.view(k, b, *orf.shape[1:]) + ox.unsqueeze(0)).view_as(orf), 0.1, inplace=True)
        else:  # one fused pass: lrelu(conv(ref) + bias + conv(x) broadcast over the K references)
            off = hip.bias_act_res_(F.conv2d(ref_feat, wr, None, 1, 1), conv1.bias, 0.1, pre=ox)
        off = conv_act(conv2, off, 0.1)
        return dyn_agg([ref_feat, off], pre_offset, act_slope=0.1)

    # ---- channels-last inference path (archs/nhwc.py)
    def nhwc_ok(self, x):
        return (self.ngf % 16 == 0
                and all(getattr(self, f'{s}_dyn_agg').nhwc_ok() for s in ('small', 'medium', 'large')))

    def _swap_nhwc(self, x, ref, pre_offset, conv1, conv2, dyn_agg):
        """_swap on [N,H,W,C] tensors: the x half of conv1 once ([B,...]), the reference half over all
        K*B images with the x half added in its epilogue (batch-broadcast) before the LeakyReLU"""
        ngf = self.ngf
        ox = nhwc.conv(conv1, x, cin_slice=(0, ngf), bias=False, amax=False)   # (an epilogue addend of the next launch, not a convolution input)
        off = nhwc.conv(conv1, ref, cin_slice=(ngf, conv1.in_channels), pre=ox, slope=0.1)
        off = nhwc.conv(conv2, off, slope=0.1)
        return dyn_agg.forward_nhwc(ref, off, pre_offset, act_slope=0.1)

    def forward_nhwc(self, x, pre_offset, ref_feat, k):
        """x [B,h,w,ngf]; ref_feat / pre_offset: k-major stacked, ref_feat as [K*B,H,W,C] -> [B,4h,4w,3]"""
        for scale, key in (('small', 'relu3_1'), ('medium', 'relu2_1'), ('large', 'relu1_1')):
            swapped = self._swap_nhwc(x, ref_feat[key], pre_offset[key], getattr(self, f'{scale}_offset_conv1'),
                                      getattr(self, f'{scale}_offset_conv2'), getattr(self, f'{scale}_dyn_agg'))
            h = getattr(self, f'head_{scale}').forward_nhwc(x, swapped, k)
            h = nhwc.res_chain(getattr(self, f'body_{scale}'), h)
            h = h + x if h.requires_grad else nhwc.rnd_(nhwc.add_(h, x))
            if scale == 'large':
                return nhwc.conv(self.tail_large[2], nhwc.conv(self.tail_large[0], h, slope=0.1, amax=False), amax=False)   # (32 channels -> direct kernel -> image)
            # Conv -> PixelShuffle(2) -> LeakyReLU: activation and shuffle commute, both are the conv epilogue
            x = nhwc.conv(getattr(self, f'tail_{scale}')[0], h, slope=0.1, epilogue=2)

    @staticmethod
    def _tail_up(tail, x):
        """Conv -> PixelShuffle(2) -> LeakyReLU(0.1); the activation commutes with the shuffle, so
        it is fused into the convolution epilogue"""
        return tail[1](conv_act(tail[0], x, 0.1))

    def forward(self, x, pre_offset_list, img_ref_feat_list):
        pre, feat, k = _stack_refs(pre_offset_list, img_ref_feat_list)
        return self.forward_stacked(x, pre, feat, k)

    def forward_stacked(self, x, pre_offset, img_ref_feat, k):
        swapped = self._swap(x, img_ref_feat['relu3_1'], pre_offset['relu3_1'], k, self.small_offset_conv1,
                             self.small_offset_conv2, self.small_dyn_agg)
        h = self.head_small.forward_stacked(x, swapped, k)
        x = self._tail_up(self.tail_small, self.body_small(h) + x)

        swapped = self._swap(x, img_ref_feat['relu2_1'], pre_offset['relu2_1'], k, self.medium_offset_conv1,
                             self.medium_offset_conv2, self.medium_dyn_agg)
        h = self.head_medium.forward_stacked(x, swapped, k)
        x = self._tail_up(self.tail_medium, self.body_medium(h) + x)

        swapped = self._swap(x, img_ref_feat['relu1_1'], pre_offset['relu1_1'], k, self.large_offset_conv1,
                             self.large_offset_conv2, self.large_dyn_agg)
        h = self.head_large.forward_stacked(x, swapped, k)
        h = self.body_large(h) + x
        return conv_act(self.tail_large[2], conv_act(self.tail_large[0], h, 0.1))


class MRAPAFusion(nn.Module):
    """Multi-reference attention + spatial attention + fusion (ref :262-348)."""

    def __init__(self, nf=64, ref_nf=256):
        super().__init__()
        self.patch_size = 3
        channels = ref_nf
        self.conv_emb1 = nn.Sequential(nn.Conv2d(nf, channels, 1), nn.PReLU())
        self.conv_emb2 = nn.Sequential(nn.Conv2d(ref_nf, channels, self.patch_size, 1, self.patch_size // 2), nn.PReLU())
        self.conv_ass = nn.Conv2d(ref_nf, channels * 2, self.patch_size, 1, self.patch_size // 2)
        self.scale = channels**-0.5
        self.feat_fusion = nn.Conv2d(nf + channels * 2, nf, 1)
        self.spatial_attn = nn.Conv2d(nf + channels * 2, channels * 2, 1)
        self.spatial_attn_mul1 = nn.Conv2d(channels * 2, channels * 2, 3, padding=1)
        self.spatial_attn_mul2 = nn.Conv2d(channels * 2, channels * 2, 3, padding=1)
        self.spatial_attn_add1 = nn.Conv2d(channels * 2, channels * 2, 3, padding=1)
        self.spatial_attn_add2 = nn.Conv2d(channels * 2, channels * 2, 3, padding=1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def spatial_padding(self, feats):
        _, _, h, w = feats.size()
        pad_h, pad_w = (4 - h % 4) % 4, (4 - w % 4) % 4
        if pad_h == 0 and pad_w == 0:
            return feats
        return F.pad(feats, [0, pad_w, 0, pad_h], mode='reflect')

    def forward(self, target, refs):
        """target (n,nf,h,w); refs: list of t tensors (n,ref_nf,h,w)   (reference signature)"""
        t = len(refs)
        return self._fuse(target, torch.stack(refs, dim=1).flatten(0, 1), t, t_major=False)

    def forward_stacked(self, target, refs, t):
        """refs (t*n, ref_nf, h, w) stacked t-major (the batched path: no stack / permute copy)"""
        return self._fuse(target, refs, t, t_major=True)

    def forward_nhwc(self, target, refs, t):
        """channels-last inference form: target [n,H,W,nf], refs [t*n,H,W,ref_nf] t-major -> [n,H,W,nf].
        torch.cat of :339/:346 = two-source convolutions; H, W that are not multiples of 4 are reflect-padded
        and cropped back as in :306-311, :348 (CUFED5's 125 x 125 LR inputs need it at two scales)."""
        h_in, w_in = target.shape[1:3]
        if h_in % 4 or w_in % 4:
            target = nhwc.to_nhwc(self.spatial_padding(nhwc.as_nchw(target)))
            refs = nhwc.to_nhwc(self.spatial_padding(nhwc.as_nchw(refs)))
            return self.forward_nhwc(target, refs, t)[:, :h_in, :w_in, :].contiguous()
        q = nhwc.conv(self.conv_emb1[0], target, prelu=self.conv_emb1[1], amax=False)   # (q, emb, ass: attention operands)
        train = q.requires_grad   # a graph is being recorded (archs/nhwc_train.py): no in-place edits of saved tensors
        fold = not train and q.dtype == torch.float32 and not nhwc.BF16   # q * scale formed inside the attention kernel (same bits, no pass over q)
        if not fold:
            q = q * self.scale if train else nhwc.rnd_(q.mul_(self.scale))
        emb = nhwc.conv(self.conv_emb2[0], refs, prelu=self.conv_emb2[1], amax=False)
        ass = nhwc.conv(self.conv_ass, refs, amax=False)
        if fold:
            r = hip.mrattn_fwd_nhwc(q, emb, ass, t, q_scale=float(self.scale))
        else:
            r = nhwc_train.attention(q, emb, ass, t) if train else nhwc.rnd_(hip.mrattn_fwd_nhwc(q, emb, ass, t))
        del emb, ass
        attn = nhwc.conv(self.spatial_attn, target, x2=r, slope=0.1)
        attn_mul = nhwc.conv(self.spatial_attn_mul2, nhwc.conv(self.spatial_attn_mul1, attn, slope=0.1), amax=False)   # (modulation terms)
        attn_add = nhwc.conv(self.spatial_attn_add2, nhwc.conv(self.spatial_attn_add1, attn, slope=0.1), amax=False)
        # refs * sigmoid(mul) * 2 + add, one pass
        r = nhwc_train.modulate(r, attn_mul, attn_add) if train else nhwc.rnd_(hip.attn_modulate_(r, attn_mul, attn_add))
        return nhwc.conv(self.feat_fusion, target, x2=r, slope=0.1)

    def _fuse(self, target, refs, t, t_major):
        h_input, w_input = target.shape[-2:]
        target = self.spatial_padding(target)
        refs = self.spatial_padding(refs)
        q = self.conv_emb1(target) * self.scale
        emb = self.conv_emb2(refs)
        ass = conv_act(self.conv_ass, refs)
        refs = _MultiRefAttention.apply(q, emb, ass, t, t_major)
        # spatial attention
        attn = conv_act(self.spatial_attn, torch.cat([target, refs], dim=1), 0.1)
        attn_mul = conv_act(self.spatial_attn_mul2, conv_act(self.spatial_attn_mul1, attn, 0.1))
        attn_add = conv_act(self.spatial_attn_add2, conv_act(self.spatial_attn_add1, attn, 0.1))
        attn_mul = torch.sigmoid(attn_mul)
        refs = refs * attn_mul * 2 + attn_add
        feat = conv_act(self.feat_fusion, torch.cat([target, refs], dim=1), 0.1)
        if feat.shape[-2:] != (h_input, w_input):   # reflect-padded to a multiple of 4 (:306-311): crop back (:348)
            feat = feat[:, :, :h_input, :w_input].contiguous()
        return feat
