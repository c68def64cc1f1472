"""basicsr/ops/fused_act/fused_act.py:30-95 on the HIP kernel of csrc/fused_act.hip:
forward y = lrelu(x + b[c], slope) * scale; backward and double-backward reuse the same kernel
with (act=3, grad=1) on the sign of the saved output, as the reference does."""
import torch
from torch import nn
from torch.autograd import Function

from ... import hip


class _Ext:
    """name-compatible stand-in for the reference's pybind module ``fused_act_ext``"""

    @staticmethod
    def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
        if not input.is_cuda:
            raise RuntimeError('input must be a CUDA tensor')  # TORCH_CHECK of fused_bias_act.cpp:10,20
        return hip.fused_bias_act(input, bias, refer, act, grad, alpha, scale)


fused_act_ext = _Ext()


class FusedLeakyReLUFunctionBackward(Function):

    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        empty = grad_output.new_empty(0)
        grad_input = fused_act_ext.fused_bias_act(grad_output, empty, out, 3, 1, negative_slope, scale)
        dim = [0] + list(range(2, grad_input.ndim))
        grad_bias = grad_input.sum(dim).detach()
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        gradgrad_out = fused_act_ext.fused_bias_act(gradgrad_input, gradgrad_bias, out, 3, 1, ctx.negative_slope,
                                                    ctx.scale)
        return gradgrad_out, None, None, None


class FusedLeakyReLUFunction(Function):

    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        empty = input.new_empty(0)
        out = fused_act_ext.fused_bias_act(input, bias, empty, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.negative_slope, ctx.scale)
        return grad_input, grad_bias, None, None


class FusedLeakyReLU(nn.Module):

    def __init__(self, channel, negative_slope=0.2, scale=2**0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2**0.5):
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)
