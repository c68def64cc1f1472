"""The helpers of basicsr/archs/arch_util.py that the MRAPA path uses: ResidualBlockNoBN :89-117,
make_layer :73-86, default_init_weights :42-70, srntt_init_weights :18-40, tensor_shift :386-410."""
import torch
from torch import nn as nn
from torch.nn import init as init
from torch.nn.modules.batchnorm import _BatchNorm


def srntt_init_weights(net, init_type='normal', init_gain=0.02):
    """Applies to every sub-module whose CLASS NAME contains 'Conv' or 'Linear' and that owns a
    ``weight`` (so nn.Conv2d yes, DynAgg no -- exactly the reference's name test)."""

    def init_func(m):
        name = m.__class__.__name__
        if hasattr(m, 'weight') and ('Conv' in name or 'Linear' in name):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError(f'initialization method [{init_type}] is not implemented')
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)
        elif 'BatchNorm2d' in name:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    net.apply(init_func)


@torch.no_grad()
def default_init_weights(module_list, scale=1, bias_fill=0, **kwargs):
    if not isinstance(module_list, list):
        module_list = [module_list]
    for module in module_list:
        for m in module.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                init.kaiming_normal_(m.weight, **kwargs)
                m.weight.data *= scale
                if m.bias is not None:
                    m.bias.data.fill_(bias_fill)
            elif isinstance(m, _BatchNorm):
                init.constant_(m.weight, 1)
                if m.bias is not None:
                    m.bias.data.fill_(bias_fill)


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
        return x + self.conv2(self.relu(self.conv1(x))) * self.res_scale


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
