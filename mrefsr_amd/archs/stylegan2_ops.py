"""The StyleGAN2 building blocks that consume ``basicsr.ops.upfirdn2d`` and ``basicsr.ops.fused_act``
(basicsr/archs/stylegan2_arch.py:26-175: make_resample_kernel, UpFirDnUpsample, UpFirDnDownsample, UpFirDnSmooth,
EqualLinear) -- the one real call pattern of the two operators: FIR resampling around a convolution and
bias + leaky ReLU + gain after it, differentiated twice by the R1 / path-length regularisers.

Padding rules (per axis; k = FIR length, f = factor, s = size of the convolution the smoother sits next to), derived from
"output has exactly in*f (or in/f) samples and the FIR is centred":
    upsample by f            total pad p = k - f        split ((p + 1) // 2 + f - 1, p // 2)
    downsample by f          total pad p = k - f        split ((p + 1) // 2, p // 2)
    smooth after a stride-f transposed conv of size s:   p = (k - f) - (s - 1),  split ((p + 1) // 2 + f - 1, p // 2 + 1)
    smooth before a stride-f conv of size s:             p = (k - f) + (s - 1),  split ((p + 1) // 2, p // 2)
An upsampling FIR is scaled by f^2 so that zero-stuffing keeps the mean.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ..ops.fused_act import fused_leaky_relu
from ..ops.upfirdn2d import upfirdn2d


def make_resample_kernel(k):
    """1-D magnitudes (e.g. [1, 3, 3, 1]) -> normalised separable 2-D FIR; a 2-D input is only normalised"""
    k = torch.as_tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


def _split(total, head_extra=0, tail_extra=0):
    return ((total + 1) // 2 + head_extra, total // 2 + tail_extra)


class _FirResample(nn.Module):
    up, down = 1, 1

    def forward(self, x):
        return upfirdn2d(x, self.kernel.type_as(x), up=self.up, down=self.down, pad=self.pad)


class UpFirDnUpsample(_FirResample):

    def __init__(self, resample_kernel, factor=2):
        super().__init__()
        self.factor = self.up = factor
        self.kernel = make_resample_kernel(resample_kernel) * factor ** 2
        self.pad = _split(self.kernel.shape[0] - factor, head_extra=factor - 1)

    def __repr__(self):
        return f'{self.__class__.__name__}(factor={self.factor})'


class UpFirDnDownsample(_FirResample):

    def __init__(self, resample_kernel, factor=2):
        super().__init__()
        self.factor = self.down = factor
        self.kernel = make_resample_kernel(resample_kernel)
        self.pad = _split(self.kernel.shape[0] - factor)

    def __repr__(self):
        return f'{self.__class__.__name__}(factor={self.factor})'


class UpFirDnSmooth(_FirResample):

    def __init__(self, resample_kernel, upsample_factor=1, downsample_factor=1, kernel_size=1):
        super().__init__()
        self.upsample_factor, self.downsample_factor = upsample_factor, downsample_factor
        self.kernel = make_resample_kernel(resample_kernel)
        k = self.kernel.shape[0]
        if upsample_factor > 1:
            self.kernel = self.kernel * upsample_factor ** 2
            self.pad = _split((k - upsample_factor) - (kernel_size - 1), head_extra=upsample_factor - 1, tail_extra=1)
        elif downsample_factor > 1:
            self.pad = _split((k - downsample_factor) + (kernel_size - 1))
        else:
            raise NotImplementedError

    def __repr__(self):
        return f'{self.__class__.__name__}(upsample_factor={self.upsample_factor}, downsample_factor={self.downsample_factor})'


class EqualLinear(nn.Module):
    """equalised-learning-rate linear layer, optionally followed by the fused bias + leaky ReLU (stylegan2_arch.py:134-175):
    weight ~ N(0, 1) / lr_mul at rest, multiplied by lr_mul / sqrt(in) when used"""

    def __init__(self, in_channels, out_channels, bias=True, bias_init_val=0, lr_mul=1, activation=None):
        super().__init__()
        if activation not in ('fused_lrelu', None):
            raise ValueError(f"Wrong activation value in EqualLinear: {activation}Supported ones are: ['fused_lrelu', None].")
        self.in_channels, self.out_channels, self.lr_mul, self.activation = in_channels, out_channels, lr_mul, activation
        self.scale = lr_mul / math.sqrt(in_channels)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels) / lr_mul)
        if bias:
            self.bias = nn.Parameter(torch.full((out_channels,), float(bias_init_val)))
        else:
            self.register_parameter('bias', None)

    def forward(self, x):
        bias = None if self.bias is None else self.bias * self.lr_mul
        if self.activation == 'fused_lrelu':
            return fused_leaky_relu(F.linear(x, self.weight * self.scale), bias)
        return F.linear(x, self.weight * self.scale, bias=bias)

    def __repr__(self):
        return f'{self.__class__.__name__}(in_channels={self.in_channels}, out_channels={self.out_channels}, bias={self.bias is not None})'
