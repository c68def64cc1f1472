"""Name- and signature-compatible stand-in for the reference's pybind module ``deform_conv_ext``
(basicsr/ops/dcn/src/deform_conv_ext.cpp:52-147) over the C ABI of include/mrefsr_hip.h, for callers that kept the
reference's ``deform_conv.py`` and only want the native side replaced:

    import mrefsr_amd.ops.dcn.deform_conv_ext as deform_conv_ext     # instead of `from . import deform_conv_ext`

Contract kept from the reference: the CALLER allocates outputs / gradients and passes them in; the functions
write in place (`grad_input`, `grad_offset`, `grad_mask`, `output` are assigned; `gradWeight` / `grad_weight` / `grad_bias`
are ACCUMULATED into, as the per-sample loops of deform_conv_cuda.cpp:539-561, :640-685 do -- the reference's Python hands
them over as zeros); `columns` / `ones` are scratch the native side may ignore (it does: the forward is a fused gather+GEMM
without a column buffer).  Shape errors raise RuntimeError (TORCH_CHECK / AT_ERROR in the reference); CPU tensors raise.
Kernel-size / stride / pad / dilation arguments come in the reference's (W, H) order for DCNv1 and (h, w) for DCNv2.
"""
import torch

from ... import hip


def _cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('deform_conv_ext: input must be a CUDA tensor')


def _columns_grads(input, offset, mask, weight, grad_output, stride, padding, dilation, group, dg, need_x=True):
    """(grad_input, grad_offset, grad_mask, grad_weight): the three GEMM + col2im stages of deform_conv_cuda.cpp:571-685"""
    co, cig, kh, kw = weight.shape
    b, c = input.shape[:2]
    ho, wo = grad_output.shape[2:]
    cog = co // group
    go = grad_output.contiguous().view(b, group, cog, ho * wo)
    col = hip.dcn_im2col(input, offset, mask, weight.shape, stride, padding, dilation, group, dg)
    grad_weight = torch.einsum('bgop,bgkp->gok', go, col.view(b, group, cig * kh * kw, ho * wo)).reshape(weight.shape)
    del col
    gcol = torch.einsum('gok,bgop->bgkp', weight.view(group, cog, cig * kh * kw), go).reshape(b, c * kh * kw, ho * wo).contiguous()
    gx, goff, gm = hip.dcn_col2im(gcol, input, offset, mask, weight.shape, stride, padding, dilation, group, dg, need_grad_x=need_x)
    return gx, goff, gm, grad_weight


def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                        deformable_group, im2col_step):
    _cuda(input, weight, offset, output)
    if tuple(weight.shape[2:]) != (kH, kW):
        raise RuntimeError(f'kernel size should be consistent with weight, but got kH: {kH} kW: {kW} weight.size(2): '
                           f'{weight.size(2)}, weight.size(3): {weight.size(3)}')
    res = hip.dcn_fwd(input.contiguous(), offset.contiguous(), None, weight.contiguous(), None, (dH, dW), (padH, padW),
                      (dilationH, dilationW), group, deformable_group)
    if output.shape != res.shape:
        output.resize_(res.shape)   # the reference re-views / resizes the caller's tensor too (deform_conv_cuda.cpp:196-197)
    output.copy_(res)
    return 1


def deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH, padW, padH,
                               dilationW, dilationH, group, deformable_group, im2col_step):
    _cuda(input, offset, gradOutput, gradInput, gradOffset, weight)
    gx, goff, _, _ = _columns_grads(input.contiguous(), offset.contiguous(), None, weight.contiguous(), gradOutput, (dH, dW),
                                    (padH, padW), (dilationH, dilationW), group, deformable_group)
    gradInput.copy_(gx)
    gradOffset.copy_(goff)
    return 1


def deform_conv_backward_parameters(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH, dilationW,
                                    dilationH, group, deformable_group, scale, im2col_step):
    _cuda(input, offset, gradOutput, gradWeight)
    _, _, _, gw = _columns_grads(input.contiguous(), offset.contiguous(), None, _weight_like(gradWeight), gradOutput, (dH, dW),
                                 (padH, padW), (dilationH, dilationW), group, deformable_group, need_x=False)
    gradWeight.add_(gw, alpha=scale)
    return 1


def _weight_like(grad_weight):
    """a placeholder weight of the right shape for the stages that only need its geometry (d weight does not read W)"""
    return torch.zeros_like(grad_weight)


def modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w, stride_h, stride_w,
                                  pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    _cuda(input, weight, offset, mask, output)
    if tuple(weight.shape[2:]) != (kernel_h, kernel_w):
        raise RuntimeError(f'Input shape and kernel shape wont match: ({kernel_h} x {kernel_w} vs {weight.size(2)} x {weight.size(3)}).')
    if input.size(1) != weight.size(1) * group:
        raise RuntimeError(f"Input shape and kernel channels wont match: ({input.size(1)} vs {weight.size(1) * group}).")
    res = hip.dcn_fwd(input.contiguous(), offset.contiguous(), mask.contiguous(), weight.contiguous(),
                      bias.contiguous() if with_bias else None, (stride_h, stride_w), (pad_h, pad_w), (dilation_h, dilation_w), group,
                      deformable_group)
    if output.shape != res.shape:
        output.resize_(res.shape)
    output.copy_(res)


def modulated_deform_conv_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight, grad_bias, grad_offset,
                                   grad_mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                                   group, deformable_group, with_bias):
    _cuda(input, weight, offset, mask, grad_input, grad_weight, grad_offset, grad_mask, grad_output)
    gx, goff, gm, gw = _columns_grads(input.contiguous(), offset.contiguous(), mask.contiguous(), weight.contiguous(), grad_output,
                                      (stride_h, stride_w), (pad_h, pad_w), (dilation_h, dilation_w), group, deformable_group)
    grad_input.copy_(gx)
    grad_offset.copy_(goff)
    grad_mask.copy_(gm)
    grad_weight.add_(gw)
    if with_bias:
        grad_bias.add_(grad_output.sum(dim=(0, 2, 3)))
