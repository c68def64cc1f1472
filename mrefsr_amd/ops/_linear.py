"""Autograd plumbing shared by the StyleGAN2 operators of ``basicsr.ops`` (fused_act, upfirdn2d).

Both operators differentiate into *linear* device kernels: the gradient of upfirdn2d is another upfirdn2d (swapped
rates, flipped FIR), the gradient of the fused leaky ReLU is a sign-gated scaling of the incoming gradient.  A linear
map needs one rule only -- d(L x) = L dx, whose backward is the transposed map -- so a single Function serves every
order of differentiation: ``LinearKernel.apply(x, L)`` computes ``L(x)`` and differentiates into
``LinearKernel.apply(g, L.T)``.  Maps are small objects with ``__call__`` (launch the kernel) and ``.T``.
"""
from torch.autograd import Function


class LinearKernel(Function):

    @staticmethod
    def forward(ctx, x, linear_map):
        ctx.linear_map = linear_map
        return linear_map(x)

    @staticmethod
    def backward(ctx, grad):
        return LinearKernel.apply(grad.contiguous(), ctx.linear_map.T), None
