"""Pure-torch DCNv2 (modulated deformable convolution).  TEST INFRASTRUCTURE ONLY.

The reference path calls ``mmcv.ops.modulated_deform_conv2d`` (ref_mrapa_restoration_arch.py:5,74-76)
-- third party, un-vendored, version unpinned.  This restates the arithmetic of the vendored
same-lineage spec: basicsr/ops/dcn/src/deform_conv_cuda_kernel.cu:467-497 (bilinear with zero
corners), :570-633 (sampling positions, validity window (-1,H)x(-1,W), value*mask) and the host
GEMM deform_conv_cuda.cpp:539-568 (out = W . columns + bias).  Differentiable through autograd, so
it is also the backward oracle (cross-checked against oracle/mrefsr_oracle.c:orc_dcnv2_bwd).
"""
import torch


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def deform_columns(x, offset, mask, kh, kw, stride, padding, dilation, deform_groups):
    """Returns columns [B, C*kh*kw, Ho*Wo] (channel-major, tap-minor like the spec's data_col)."""
    b, c, h, w = x.shape
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    ho = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    wo = (w + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    kk = kh * kw
    dg = deform_groups
    cpg = c // dg
    dev, dt = x.device, x.dtype
    base_h = (torch.arange(ho, device=dev, dtype=dt) * sh - ph).view(1, 1, 1, ho, 1)
    base_w = (torch.arange(wo, device=dev, dtype=dt) * sw - pw).view(1, 1, 1, 1, wo)
    tap_h = (torch.arange(kh, device=dev, dtype=dt) * dh).repeat_interleave(kw).view(1, 1, kk, 1, 1)
    tap_w = (torch.arange(kw, device=dev, dtype=dt) * dw).repeat(kh).view(1, 1, kk, 1, 1)
    off = offset.view(b, dg, kk, 2, ho, wo)
    pos_h = base_h + tap_h + off[:, :, :, 0]  # [b, dg, kk, ho, wo]
    pos_w = base_w + tap_w + off[:, :, :, 1]
    hl = torch.floor(pos_h)
    wl = torch.floor(pos_w)
    lh, lw = pos_h - hl, pos_w - wl
    uh, uw = 1 - lh, 1 - lw
    hl, wl = hl.long(), wl.long()
    hh, wh = hl + 1, wl + 1
    xg = x.view(b, dg, cpg, h * w)

    def corner(hi, wi):
        ok = (hi >= 0) & (hi <= h - 1) & (wi >= 0) & (wi <= w - 1)
        lin = (hi.clamp(0, h - 1) * w + wi.clamp(0, w - 1)).view(b, dg, 1, kk * ho * wo).expand(-1, -1, cpg, -1)
        v = torch.gather(xg, 3, lin).view(b, dg, cpg, kk, ho, wo)
        return v * ok.view(b, dg, 1, kk, ho, wo).to(dt)

    def wgt(t):
        return t.view(b, dg, 1, kk, ho, wo)

    val = (wgt(uh * uw) * corner(hl, wl) + wgt(uh * lw) * corner(hl, wh)
           + wgt(lh * uw) * corner(hh, wl) + wgt(lh * lw) * corner(hh, wh))
    if mask is not None:
        val = val * mask.view(b, dg, 1, kk, ho, wo)
    return val.reshape(b, c * kk, ho * wo), ho, wo


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                            groups=1, deform_groups=1):
    """Signature of mmcv.ops.modulated_deform_conv2d / basicsr.ops.dcn.modulated_deform_conv."""
    co, cig, kh, kw = weight.shape
    b = x.shape[0]
    cols, ho, wo = deform_columns(x, offset, mask, kh, kw, stride, padding, dilation, deform_groups)
    cols = cols.view(b, groups, cig * kh * kw, ho * wo)
    wg = weight.view(groups, co // groups, cig * kh * kw)
    out = torch.einsum('gok,bgkp->bgop', wg, cols).reshape(b, co, ho, wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def deform_conv2d(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deform_groups=1):
    """DCNv1 (no mask, no bias): basicsr/ops/dcn/deform_conv.py:33-118 semantics."""
    return modulated_deform_conv2d(x, offset, None, weight, None, stride, padding, dilation, groups,
                                   deform_groups)
