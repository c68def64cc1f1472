"""``basicsr.ops.upfirdn2d`` on csrc/upfirdn2d.hip (reference: basicsr/ops/upfirdn2d/upfirdn2d.py:30-192).

Definition (per image plane, per axis; upfirdn2d.py:162-192):  zero-stuff by ``up``, pad by (pad0, pad1) (negative = crop),
correlate with the flipped FIR, keep every ``down``-th sample;  out = (in * up + pad0 + pad1 - k) // down + 1.
That is a linear map R(k, up, down, pad).  Its transpose is again a resampler: R(flip k, up' = down, down' = up, pad')
acting on planes of the output size, with per axis
        pad0' = k - pad0 - 1,        pad1' = in * up - out * down + pad0 - up + 1
(match "output sample o reads input sample i through tap o*down + t - pad0 = i*up" from the other side; the reference's
g_pad, :121-126).  ops/_linear.LinearKernel turns the pair (R, R^T) into gradients of every order.
"""
import torch

from ... import hip
from .._linear import LinearKernel


class _Ext:
    """name-compatible stand-in for the reference's pybind module ``upfirdn2d_ext`` (upfirdn2d.cpp:13-24)"""

    @staticmethod
    def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
        if not (input.is_cuda and kernel.is_cuda):
            raise RuntimeError('input must be a CUDA tensor')
        return hip.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)


upfirdn2d_ext = _Ext()


class _Resampler:
    """R(fir, up, down, pad) on tensors (..., in_h, in_w); ``.T`` is its transpose on (..., out_h, out_w)"""

    def __init__(self, fir, up, down, pad, in_hw, transpose_of=None):
        self.fir, self.up, self.down, self.pad, self.in_hw = fir, tuple(up), tuple(down), tuple(pad), tuple(in_hw)
        kh, kw = fir.shape
        (ux, uy), (dx, dy), (px0, px1, py0, py1) = self.up, self.down, self.pad
        self.out_hw = ((in_hw[0] * uy + py0 + py1 - kh) // dy + 1, (in_hw[1] * ux + px0 + px1 - kw) // dx + 1)
        self._t = transpose_of

    def __call__(self, x):
        lead = x.shape[:-2]
        planes = x.reshape(-1, self.in_hw[0], self.in_hw[1], 1)
        y = upfirdn2d_ext.upfirdn2d(planes, self.fir, *self.up, *self.down, *self.pad)
        return y.view(*lead, *self.out_hw)

    @property
    def T(self):
        if self._t is None:
            kh, kw = self.fir.shape
            (ux, uy), (dx, dy), (px0, _, py0, _) = self.up, self.down, self.pad
            (ih, iw), (oh, ow) = self.in_hw, self.out_hw
            t_pad = (kw - px0 - 1, iw * ux - ow * dx + px0 - ux + 1, kh - py0 - 1, ih * uy - oh * dy + py0 - uy + 1)
            self._t = _Resampler(torch.flip(self.fir, [0, 1]), self.down, self.up, t_pad, self.out_hw, transpose_of=self)
        return self._t


class UpFirDn2d:
    """Call-compatible twin of the reference's Function: ``apply(input (N,C,H,W), kernel, (up_x, up_y), (down_x, down_y),
    (pad_x0, pad_x1, pad_y0, pad_y1))``"""

    @staticmethod
    def apply(input, kernel, up, down, pad):
        return LinearKernel.apply(input.contiguous(), _Resampler(kernel, up, down, pad, input.shape[-2:]))


class UpFirDn2dBackward:
    """Call-compatible twin of the reference's backward Function (the transposed resampler applied to ``grad_output``);
    ``grad_kernel`` / ``g_pad`` are what the reference precomputes and are implied by the other arguments here."""

    @staticmethod
    def apply(grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        fwd = _Resampler(kernel, up, down, pad, in_size[-2:])
        g = grad_output.reshape(*in_size[:-2], *fwd.out_hw).contiguous()
        return LinearKernel.apply(g, fwd.T)


def upfirdn2d_native(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """The reference's CPU form (upfirdn2d.py:162-192) from the definition at the top of this file, in plain torch: zero-stuff,
    pad (negative = crop), correlate with the flipped FIR, decimate.  input (N, C, H, W) -> (N, C, out_h, out_w).  The reference
    ships this path itself (its `upfirdn2d` takes it for CPU tensors, :153-155); it never touches the HIP library."""
    n, c, in_h, in_w = input.shape
    kh, kw = kernel.shape
    planes = input.reshape(n * c, 1, in_h, in_w)
    stuffed = planes.new_zeros(n * c, 1, in_h * up_y, in_w * up_x)
    stuffed[:, :, ::up_y, ::up_x] = planes
    # torch's pad crops with negative widths: [left, right, top, bottom]
    padded = torch.nn.functional.pad(stuffed, [pad_x0, pad_x1, pad_y0, pad_y1])
    full = torch.nn.functional.conv2d(padded, torch.flip(kernel, [0, 1]).to(padded.dtype).view(1, 1, kh, kw))
    out = full[:, :, ::down_y, ::down_x]
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    return out[:, :, :out_h, :out_w].reshape(n, c, out_h, out_w)


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """Same call as the reference (:149-155), including its dispatch: CPU tensors take the plain-torch form above, GPU tensors
    the HIP kernels."""
    if input.device.type == 'cpu':
        return upfirdn2d_native(input, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
