"""DCNv1 / DCNv2 autograd functions and modules with the interface of the reference's
basicsr/ops/dcn/deform_conv.py (DeformConvFunction :33-118, ModulatedDeformConvFunction :121-184,
modules :191-379) -- which is also the argument order of mmcv.ops.modulated_deform_conv2d used at
ref_mrapa_restoration_arch.py:74-76.

Native side: one fused HIP kernel for the forward (gather -> LDS -> MFMA, see csrc/dcn.hip); the backward of the
3x3 / stride 1 / one-group fp32 layers runs on the two fused kernels of csrc/dcn_bwd.hip (_backward_fused), every other
shape on the HIP im2col / col2im kernels plus two plain library GEMMs (hipBLASLt via torch).
CPU tensors raise NotImplementedError exactly like the reference (:61-62, :143-144).
"""
import math
import os

import torch
from torch import nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn import functional as F
from torch.nn.modules.utils import _pair, _single

from ... import hip

_COL_BYTES_LIMIT = 3 << 30  # backward column buffers are built per batch chunk of at most this size
FUSED_BWD = os.environ.get('MREFSR_DCN_FUSED_BWD', '1') != '0'   # 0: im2col + GEMMs + col2im for every shape (A/B, tests)


def _ones(*vs):
    return all((tuple(v) if isinstance(v, (tuple, list)) else (v, v)) == (1, 1) for v in vs)


def _backward_fused(grad_output, x, offset, mask, weight, with_bias, need_x, dg, need=(True, True, True)):
    """The DCNv2 backward of the layers of the path (3x3, stride / padding / dilation 1, one group, fp32) on the two fused kernels of
    csrc/dcn_bwd.hip -- d columns = W^T . g on the matrix pipe with the offset / mask / input gradients as its epilogue, d W with
    the columns re-gathered inside the GEMM -- instead of im2col + two library GEMMs + col2im (deform_conv.py:155-184,
    deform_conv_cuda.cpp:571-685).  Returns None where the kernels do not apply (the caller takes the generic route)."""
    co, cig, kh, kw = weight.shape
    c = x.shape[1]
    if not (mask is not None and (kh, kw) == (3, 3) and x.dtype == torch.float32 and c % 32 == 0 and c // dg in (8, 16, 32) and co % 16 == 0
            and cig == c and not hip.is_range_free()):
        return None
    from ...archs import nhwc_train as nt   # (the fp16 weight scale cache of the training engine: one readback per parameter storage)
    if torch.cuda.is_current_stream_capturing() and (weight.data_ptr(), weight.numel()) not in nt._scales:
        return None
    ws = nt._wscale(weight)
    if not ws:   # an (almost) all-zero weight: no fp16 scale
        return None
    # the kernels read channels-last activations: one transposition of x and g (the column buffer this replaces is 9x either)
    g = grad_output.permute(0, 2, 3, 1).contiguous()
    xl = x.permute(0, 2, 3, 1).contiguous()
    g, g_bias, _, amax = hip.act_bwd_nhwc(g, None, 0, want_bias=with_bias, want_amax=True)
    gx = goff = gm = gw = None
    if need_x or need[0] or need[1]:
        pk = hip.conv_pack_view(weight, None, 16, dgrad='T', wscale=ws)
        gx, goff, gm = hip.dcn_bwd_data(g, xl, offset, mask, pk, dg, g_amax=amax, need_grad_x=need_x)
    if need[2]:
        gw = hip.dcn_bwd_weight(g, xl, offset, mask, co, dg, g_amax=amax)
    return gx, goff, gm, gw, g_bias


def _backward(ctx, grad_output, x, offset, mask, weight, with_bias, need_x, fused=True):
    stride, padding, dilation, groups, dg = ctx.stride, ctx.padding, ctx.dilation, ctx.groups, ctx.deformable_groups
    if fused and FUSED_BWD and groups == 1 and _ones(stride, padding, dilation):
        r = _backward_fused(grad_output, x, offset, mask, weight, with_bias, need_x, dg)
        if r is not None:
            return r
    b, c, _, _ = x.shape
    co, cig, kh, kw = weight.shape
    cog = co // groups
    # a channels-last gradient (the training engine's storage, handed over as a permuted view) feeds the two GEMMs as it lies:
    # a transposed operand is a GEMM flag, not a copy
    if not (groups == 1 and grad_output.permute(0, 2, 3, 1).is_contiguous()):
        grad_output = grad_output.contiguous()
    ho, wo = grad_output.shape[2:]
    per_sample = c * kh * kw * ho * wo * 4
    step = max(1, min(b, _COL_BYTES_LIMIT // max(per_sample, 1)))
    grad_x = torch.zeros_like(x) if need_x and step < b else None
    grad_offset = torch.empty_like(offset) if step < b else None
    grad_mask = torch.empty_like(mask) if mask is not None and step < b else None
    grad_weight = torch.zeros(groups, cog, cig * kh * kw, device=x.device, dtype=x.dtype)
    wg = weight.view(groups, cog, cig * kh * kw)
    for b0 in range(0, b, step):
        sl = slice(b0, min(b, b0 + step))
        xs, offs = x[sl], offset[sl]
        ms = mask[sl] if mask is not None else None
        nb = grad_output[sl].shape[0]
        # d(weight): grad_out . columns^T        (deform_conv_cuda.cpp:640-657)
        col = hip.dcn_im2col(xs, offs, ms, weight.shape, stride, padding, dilation, groups, dg)
        if groups == 1:
            # batched GEMMs on the tensors as they lie (a transposed operand is a GEMM flag): the einsum forms below re-lay
            # the 1-GB column buffers of the 160^2 scale out twice per direction
            g2 = grad_output[sl].flatten(2)          # [nb, Co, Ho*Wo], a view for planar and channels-last storage alike
            grad_weight[0] += torch.bmm(g2, col.transpose(1, 2)).sum(0)
            del col
            gcol = torch.bmm(wg[0].t().unsqueeze(0).expand(nb, -1, -1), g2)   # d(columns) = W^T . grad_out   (:617-620), [nb, C*kh*kw, Ho*Wo]
        else:
            go = grad_output[sl].reshape(nb, groups, cog, ho * wo)
            grad_weight += torch.einsum('bgop,bgkp->gok', go, col.view(nb, groups, cig * kh * kw, ho * wo))
            del col
            gcol = torch.einsum('gok,bgop->bgkp', wg, go).reshape(nb, c * kh * kw, ho * wo).contiguous()
        gx, goff, gm = hip.dcn_col2im(gcol, xs, offs, ms, weight.shape, stride, padding, dilation, groups, dg,
                                      need_grad_x=need_x)
        del gcol
        if step >= b:   # one chunk: the kernel's outputs are the results
            grad_offset, grad_mask, grad_x = goff, gm, gx
            break
        grad_offset[sl] = goff
        if grad_mask is not None:
            grad_mask[sl] = gm
        if need_x:
            grad_x[sl] = gx
    grad_bias = grad_output.sum(dim=(0, 2, 3)) if with_bias else None
    return grad_x, grad_offset, grad_mask, grad_weight.view_as(weight), grad_bias


class DeformConvFunction(Function):
    """DCNv1: deform_conv(input, offset, weight, stride, padding, dilation, groups,
    deformable_groups, im2col_step)  (deform_conv.py:36-45)."""

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError(f'Expected 4D tensor as input, got {input.dim()}D tensor instead.')
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        if not input.is_cuda:
            raise NotImplementedError
        cur_im2col_step = min(im2col_step, input.shape[0])
        assert (input.shape[0] % cur_im2col_step) == 0, 'im2col step must divide batchsize'
        DeformConvFunction._output_size(input, weight, ctx.padding, ctx.dilation, ctx.stride)
        input, offset, weight = input.contiguous(), offset.contiguous(), weight.contiguous()
        ctx.save_for_backward(input, offset, weight)
        return hip.dcn_fwd(input, offset, None, weight, None, ctx.stride, ctx.padding, ctx.dilation, groups,
                           deformable_groups)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, weight = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        gx, goff, _, gw, _ = _backward(ctx, grad_output, input, offset, None, weight, False, need_x)
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            gx = goff = None
        if not ctx.needs_input_grad[2]:
            gw = None
        return (gx, goff, gw, None, None, None, None, None, None)

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        channels = weight.size(0)
        output_size = (input.size(0), channels)
        for d in range(input.dim() - 2):
            kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
            output_size += ((input.size(d + 2) + (2 * padding[d]) - kernel) // stride[d] + 1, )
        if not all(map(lambda s: s > 0, output_size)):
            raise ValueError(f'convolution input is too small (output would be {"x".join(map(str, output_size))})')
        return output_size


class ModulatedDeformConvFunction(Function):
    """DCNv2: modulated_deform_conv(input, offset, mask, weight, bias, stride, padding, dilation,
    groups, deformable_groups)  (deform_conv.py:124-134).  ``act_slope`` (extra, last) fuses the
    LeakyReLU that follows every DynAgg of the path into the kernel epilogue."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1, act_slope=1.0):
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.with_bias = bias is not None
        ctx.act_slope = float(act_slope)
        if not input.is_cuda:
            raise NotImplementedError
        # decided from the ORIGINAL arguments (bias included), before any .contiguous() copy: inside Function.forward grad
        # mode is off, so a copy would report requires_grad = False
        need_grad = any(ctx.needs_input_grad[:5])
        input, offset, mask, weight = input.contiguous(), offset.contiguous(), mask.contiguous(), weight.contiguous()
        output = hip.dcn_fwd(input, offset, mask, weight, bias.contiguous() if ctx.with_bias else None, stride,
                             padding, dilation, groups, deformable_groups, ctx.act_slope)
        if need_grad:
            ctx.save_for_backward(input, offset, mask, weight, output if ctx.act_slope != 1.0 else None)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, output = ctx.saved_tensors
        if output is not None:  # derivative of the fused LeakyReLU
            grad_output = torch.where(output > 0, grad_output, grad_output * ctx.act_slope)
        gx, goff, gm, gw, gb = _backward(ctx, grad_output, input, offset, mask, weight, ctx.with_bias,
                                         ctx.needs_input_grad[0])
        return (gx, goff, gm, gw, gb, None, None, None, None, None, None)

    @staticmethod
    def _infer_shape(ctx, input, weight):
        n, channels_out = input.size(0), weight.size(0)
        height, width = input.shape[2:4]
        kernel_h, kernel_w = weight.shape[2:4]
        height_out = (height + 2 * ctx.padding - (ctx.dilation * (kernel_h - 1) + 1)) // ctx.stride + 1
        width_out = (width + 2 * ctx.padding - (ctx.dilation * (kernel_w - 1) + 1)) // ctx.stride + 1
        return n, channels_out, height_out, width_out


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply


class DeformConv(nn.Module):
    """deform_conv.py:191-249"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, f'in_channels {in_channels} is not divisible by groups {groups}'
        assert out_channels % groups == 0, f'out_channels {out_channels} is not divisible by groups {groups}'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.transposed, self.output_padding = False, _single(0)  # nn.Conv2d compatibility
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // self.groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.in_channels * self.kernel_size[0] * self.kernel_size[1])
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        # inputs smaller than the kernel are zero-padded and the output cropped (:237-249)
        pad_h = max(self.kernel_size[0] - x.size(2), 0)
        pad_w = max(self.kernel_size[1] - x.size(3), 0)
        if pad_h or pad_w:
            x = F.pad(x, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
            offset = F.pad(offset, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
        out = deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                          self.deformable_groups)
        if pad_h or pad_w:
            out = out[:, :, :out.size(2) - pad_h, :out.size(3) - pad_w].contiguous()
        return out


class DeformConvPack(DeformConv):
    """deform_conv.py:252-292: offsets predicted by an ordinary conv initialised to zero."""
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels,
                                     self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride),
                                     padding=_pair(self.padding), dilation=_pair(self.dilation), bias=True)
        self.init_offset()

    def init_offset(self):
        with torch.no_grad():
            self.conv_offset.weight.zero_()
        with torch.no_grad():
            self.conv_offset.bias.zero_()

    def forward(self, x):
        offset = self.conv_offset(x)
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)


class ModulatedDeformConv(nn.Module):
    """deform_conv.py:295-337 (parameter names weight / bias, uniform(+-1/sqrt(C*k*k)) init)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups = groups, deformable_groups
        self.with_bias = bias
        self.transposed, self.output_padding = False, _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.init_weights()

    def init_weights(self):
        stdv = 1. / math.sqrt(self.in_channels * self.kernel_size[0] * self.kernel_size[1])
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    """deform_conv.py:340-379 (used by DCNv2Pack / EDVR-style consumers)."""
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels,
                                     self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride),
                                     padding=_pair(self.padding), dilation=_pair(self.dilation), bias=True)
        self.init_weights()

    def init_weights(self):
        super().init_weights()
        if hasattr(self, 'conv_offset'):
            with torch.no_grad():
                self.conv_offset.weight.zero_()
            with torch.no_grad():
                self.conv_offset.bias.zero_()

    def forward(self, x):
        out = self.conv_offset(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)
