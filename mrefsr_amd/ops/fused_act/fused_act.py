"""``basicsr.ops.fused_act`` on csrc/fused_act.hip (reference: basicsr/ops/fused_act/fused_act.py:30-95, kernel
fused_bias_act_kernel.cu:19-50).

Definition:  y = leaky_relu(x + bias[channel], slope) * scale.
    dy/dx is the diagonal map  g -> g * (y > 0 ? 1 : slope) * scale  (the sign of y is the sign of x + bias), which the
    kernel evaluates as mode (act=3, grad=1) with ``refer = y``;  d/dbias = that, summed over all axes but the channel's.
The diagonal map is linear and its own transpose, so ops/_linear.LinearKernel gives the backward and every higher
derivative (the reference's double backward ``fused_bias_act(gg_input, gg_bias, out, 3, 1, ...)`` falls out of it: autograd
adds the broadcast bias cotangent to the input cotangent before the gate is applied).
"""
import torch
from torch import nn
from torch.autograd import Function

from ... import hip
from .._linear import LinearKernel


class _Ext:
    """name-compatible stand-in for the reference's pybind module ``fused_act_ext`` (fused_bias_act.cpp:14-26)"""

    @staticmethod
    def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
        if not input.is_cuda:
            raise RuntimeError('input must be a CUDA tensor')
        return hip.fused_bias_act(input, bias, refer, act, grad, alpha, scale)


fused_act_ext = _Ext()


class _SignGate:
    """g -> g * (out > 0 ? 1 : slope) * scale: the Jacobian of the activation at the saved output"""

    def __init__(self, out, slope, scale):
        self.out, self.slope, self.scale = out, slope, scale

    def __call__(self, g):
        return fused_act_ext.fused_bias_act(g, g.new_empty(0), self.out, 3, 1, self.slope, self.scale)

    @property
    def T(self):
        return self


def _channel_sum(t):
    return t.sum(dim=[d for d in range(t.ndim) if d != 1])


class FusedLeakyReLUFunctionBackward:
    """Call-compatible twin of the reference's Function of that name:
    ``apply(grad_output, out, negative_slope, scale) -> (grad_input, grad_bias)``, differentiable again."""

    @staticmethod
    def apply(grad_output, out, negative_slope, scale):
        grad_input = LinearKernel.apply(grad_output.contiguous(), _SignGate(out, negative_slope, scale))
        return grad_input, _channel_sum(grad_input)


class FusedLeakyReLUFunction(Function):

    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = fused_act_ext.fused_bias_act(input, bias, input.new_empty(0), 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.gate_args = (negative_slope, scale)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, ctx.saved_tensors[0], *ctx.gate_args)
        return grad_input, grad_bias, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2**0.5):
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    """learned per-channel bias + leaky ReLU + gain (state-dict key ``bias``, as the reference)"""

    def __init__(self, channel, negative_slope=0.2, scale=2**0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
