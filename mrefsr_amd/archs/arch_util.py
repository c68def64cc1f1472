"""The helpers of basicsr/archs/arch_util.py that the MRAPA path uses: ResidualBlockNoBN :89-117,
make_layer :73-86, default_init_weights :42-70, srntt_init_weights :18-40, tensor_shift :386-410."""
import os
import torch
from torch import nn as nn
from torch.nn import functional as F
from torch.nn import init as init
from torch.nn.modules.batchnorm import _BatchNorm

from .. import hip


def conv_act(conv, x, slope=1.0, residual=None):
    """act(conv(x) + bias) [+ residual] with act = LeakyReLU(slope) (1 = identity, 0 = ReLU).

    Inference (no autograd graph needed): MIOpen convolution without bias, then ONE fused HIP
    pass for bias + activation + residual, in place on the convolution output
    (mrefsr_bias_act_res_f32) -- PyTorch-ROCm would launch a broadcast add, an activation and an
    add as three more full passes.  Same floating-point operations in the same order.
    With autograd enabled the plain torch ops run (training memory/graph semantics unchanged)."""
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        y = conv(x)
        if slope == 0.0:
            y = F.relu(y, inplace=True)
        elif slope != 1.0:
            y = F.leaky_relu(y, slope, inplace=True)
        return y if residual is None else residual + y
    y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return hip.bias_act_res_(y, conv.bias, slope, residual)


def run_conv_relu_stack(layers, x, taps=None):
    """nn.Sequential of Conv2d / ReLU / MaxPool2d (the VGG stacks) with every conv+ReLU pair fused
    through conv_act.  ``taps``: names whose output is returned in a dict (else the final tensor)."""
    out = {}
    items = list(layers._modules.items())
    i = 0
    no_graph = not (torch.is_grad_enabled() and x.requires_grad)
    while i < len(items):
        name, layer = items[i]
        # conv -> ReLU -> MaxPool2d(2,2) with neither intermediate tapped: one fused epilogue pass
        if (no_graph and isinstance(layer, nn.Conv2d) and i + 2 < len(items) and isinstance(items[i + 1][1], nn.ReLU)
                and isinstance(items[i + 2][1], nn.MaxPool2d) and items[i + 2][1].kernel_size in (2, (2, 2))
                and items[i + 2][1].stride in (2, (2, 2)) and items[i + 2][1].padding in (0, (0, 0))
                and not (taps and (name in taps or items[i + 1][0] in taps or items[i + 2][0] in taps))
                and not (torch.is_grad_enabled() and layer.weight.requires_grad)):
            y = F.conv2d(x, layer.weight, None, layer.stride, layer.padding, layer.dilation, layer.groups)
            x = hip.bias_relu_pool2(y, layer.bias)
            i += 3
            continue
        if isinstance(layer, nn.Conv2d) and i + 1 < len(items) and isinstance(items[i + 1][1], nn.ReLU) \
                and not (taps and name in taps):
            x = conv_act(layer, x, 0.0)
            name = items[i + 1][0]
            i += 2
        elif isinstance(layer, nn.Conv2d):
            x = conv_act(layer, x, 1.0)
            i += 1
        else:
            x = layer(x)
            i += 1
        if taps and name in taps:
            out[name] = x.clone()
    return out if taps else x


def srntt_init_weights(net, init_type='normal', init_gain=0.02):
    """Applies to every sub-module whose CLASS NAME contains 'Conv' or 'Linear' and that owns a
    ``weight`` (so nn.Conv2d yes, DynAgg no -- exactly the reference's name test)."""

    def init_func(m):
        name = m.__class__.__name__
        if hasattr(m, 'weight') and ('Conv' in name or 'Linear' in name):
            if init_type == 'normal':
                init.normal_(m.weight, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight, gain=init_gain)
            else:
                raise NotImplementedError(f'initialization method [{init_type}] is not implemented')
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias, 0.0)
        elif 'BatchNorm2d' in name:
            init.normal_(m.weight, 1.0, init_gain)
            init.constant_(m.bias, 0.0)

    net.apply(init_func)


@torch.no_grad()
def default_init_weights(module_list, scale=1, bias_fill=0, **kwargs):
    if not isinstance(module_list, list):
        module_list = [module_list]
    for module in module_list:
        for m in module.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                init.kaiming_normal_(m.weight, **kwargs)
                m.weight.mul_(scale)
                if m.bias is not None:
                    m.bias.fill_(bias_fill)
            elif isinstance(m, _BatchNorm):
                init.constant_(m.weight, 1)
                if m.bias is not None:
                    m.bias.fill_(bias_fill)


def make_layer(basic_block, num_basic_block, **kwarg):
    return nn.Sequential(*[basic_block(**kwarg) for _ in range(num_basic_block)])


class ResidualBlockNoBN(nn.Module):
    """x + res_scale * conv2(relu(conv1(x)))"""

    def __init__(self, num_feat=64, res_scale=1, pytorch_init=False):
        super().__init__()
        self.res_scale = res_scale
        self.conv1 = nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=True)
        self.relu = nn.ReLU(inplace=True)
        if not pytorch_init:
            default_init_weights([self.conv1, self.conv2], 0.1)

    def forward(self, x):
        if self.res_scale != 1:
            return x + self.conv2(self.relu(self.conv1(x))) * self.res_scale
        return conv_act(self.conv2, conv_act(self.conv1, x, 0.0), 1.0, residual=x)


def tensor_shift(x, shift=(2, 2), fill_val=0):
    """[b,h,w,c] shifted down/right by ``shift`` with ``fill_val`` elsewhere (negative shifts are
    NotImplementedError in the reference too).  The path itself does not call this: the HIP kernel
    mrefsr_offsets_from_idx_f32 writes all 27 shifted planes directly."""
    _, h, w, _ = x.size()
    sh, sw = shift
    if sh < 0 or sw < 0:
        raise NotImplementedError
    new = torch.full_like(x, fill_val)
    new[:, sh:, sw:, :] = x[:, :h - sh, :w - sw, :]
    return new


class DCNv2Pack(nn.Module):
    """Modulated deformable conv whose offsets / masks come from ANOTHER feature map
    (basicsr/archs/arch_util.py:291-318, used by EDVR / BasicVSR++ style alignment).  Built on
    mrefsr_amd.ops.dcn.ModulatedDeformConvPack's parameters (weight, bias, conv_offset).  The
    reference warns when mean|offset| > 50 through a host sync per call; here the check is left to
    the caller (``last_offset_absmean`` is a device scalar)."""

    def __new__(cls, *args, **kwargs):
        from ..ops.dcn import ModulatedDeformConvPack

        class _DCNv2Pack(ModulatedDeformConvPack):

            def forward(self, x, feat):
                from ..ops.dcn import modulated_deform_conv
                out = self.conv_offset(feat)
                o1, o2, mask = torch.chunk(out, 3, dim=1)
                offset = torch.cat((o1, o2), dim=1)
                mask = torch.sigmoid(mask)
                self.last_offset_absmean = offset.detach().abs().mean()
                return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                             self.dilation, self.groups, self.deformable_groups)

        _DCNv2Pack.__name__ = 'DCNv2Pack'
        return _DCNv2Pack(*args, **kwargs)
