"""Channels-last inference engine shared by the arch mirrors.

Under ``torch.no_grad()`` on a GPU every 3x3 / 1x1 convolution of the path (VGG16 extractor, VGG19
taps, ContentExtractor, offset convolutions, MRAPAFusion heads, residual trunks, tails: ~38 TFLOP
of the ~40 per batch-8 step) runs on ``mrefsr_conv_nhwc_f32`` (csrc/conv_nhwc.hip): an implicit
GEMM on the 16-bit matrix pipe whose exact operand splits give fp32-equivalent results, with
bias / activation / residual / max-pool / pixel-shuffle / concat fused, activations kept
[N,H,W,C] end to end.  Tensors that cross the public (reference) API stay *logically* NCHW: they
are returned as ``permute(0,3,1,2)`` views of the NHWC storage (= torch channels_last), so
callers written against the reference see the same shapes and values.

With autograd enabled (training of net_g) the same launches are recorded as autograd nodes
(archs/nhwc_train.py).  There is no switch that routes the path to a library: what the engine cannot
take (non-fp32 tensors, training maps whose sides are not multiples of 4, module kinds it does not
know) falls to the generic torch forms of arch_util.py -- listed in INTEGRATION.md.  ``ENABLED`` is a
module attribute for the tests that compare the two (tests/ flip it; no environment variable).
"""
import os

import torch
from torch import nn as nn

from .. import hip

ENABLED = True
# Arithmetic of the convolution kernel, both as accurate against fp64 as an fp32 convolution:
#   16 (default) fp16 two-term split, 3 products; needs |activation| < 65504 -- guarded by a device
#                flag (hip.conv_range_tripped()): MultiRefRestorationModel.test() / optimize_parameters() read it once per
#                batch and re-run the batch with terms 6 (hip.range_free()) when it fired
#   6            bf16 three-term split, 6 products, no range limit, 1.5x slower
#   3            bf16 two-term split (~2^-16 relative): experiments only
#   1            bf16 ARITHMETIC (BASELINE configs[4]): operands rounded to bf16, one product, fp32 accumulate,
#                results rounded to bf16 (fp32 containers); selected by set_arithmetic('bf16') / MREFSR_DTYPE=bf16
TERMS = int(os.environ.get('MREFSR_CONV_TERMS', '16'))
BF16 = False
# Winograd F(2x2, 3x3) form of the terms-16 arithmetic (csrc/conv_wino.hip, descriptor terms 17: 2.25x fewer MFMAs per output,
# the same or a smaller error against fp64): 'auto' (default) takes it for the layer shapes on which it is faster than the direct
# kernel on an MI355X (tools/conv_wino_check.py, profiles/r4_conv_layers.txt), '1' wherever it applies, '0' never
WINO = os.environ.get('MREFSR_CONV_WINO', 'auto')


def wino_applies(n, h, w, cin, cout, ld_max, epilogue=0):
    """terms 17 instead of 16 for a 3x3 convolution of [n,h,w,cin] -> cout?  The kernels need >= 3 channel chunks and 32-bit
    byte offsets inside an image.  'auto': every launch the four-wave kernel serves (whole 16 x 16 tiles, whole cout blocks, plain or
    max-pool epilogue: 1.15-1.47x the direct kernels on every benchmark shape, profiles/r5_conv_wino4_check.txt); for the rest
    (eight-wave kernel) the direct kernel keeps what it wins: 64-channel inputs on mid-size launches, where its 8-row tiles run
    three blocks per CU, and 64 -> 256 layers."""
    if WINO == '0' or cin <= 32 or cout > 1024 or h * w * ld_max * 4 >= 0xffff0000:   # (the kernel stages <= 1024 biases in LDS; 32-bit offsets)
        return False
    if WINO != 'auto':
        return True
    if cin >= 128 or (h % 16 == 0 and w % 16 == 0 and cout % 64 == 0 and epilogue in (0, 1)):
        return True
    tiles = n * ((h + 15) // 16) * ((w + 15) // 16) * ((cout + 63) // 64)
    return cout <= 128 and (tiles >= 8192 or tiles <= 1024)
# In the bf16 arithmetic the activations also TRAVEL as bf16 (2-byte channels-last tensors: half the HBM bytes of every
# layer; the kernels take bf16 tensors directly).  MREFSR_BF16_STORE=0 keeps the bf16 values in fp32 containers instead
# (same bits: every tensor is a rounded bf16 value either way; tests compare the two).
STORE16 = os.environ.get('MREFSR_BF16_STORE', '1') != '0'


def set_arithmetic(kind):
    """'fp32' (default: fp32-equivalent results) or 'bf16' (weights and activations rounded to bf16, fp32
    accumulation -- BASELINE configs[4]; parity is then against oracle.pipeline with BF16 = True)"""
    global TERMS, BF16
    if kind == 'bf16':
        TERMS, BF16 = 1, True
    elif kind == 'fp32':
        TERMS, BF16 = int(os.environ.get('MREFSR_CONV_TERMS', '16')), False
    else:
        raise ValueError(f'set_arithmetic: {kind!r} (fp32 or bf16)')


if os.environ.get('MREFSR_DTYPE', 'fp32') == 'bf16':
    set_arithmetic('bf16')


def rnd_(t):
    """in the bf16 arithmetic: round an activation produced by a non-convolution op to bf16, in place (a bf16 tensor
    already is: torch computes its elementwise ops in fp32 and rounds once, the same value)"""
    if BF16 and t.dtype != torch.bfloat16:
        t.copy_(t.bfloat16())
    return t


def storing16():
    return BF16 and STORE16


def active(x):
    """the engine applies: GPU tensor and no autograd graph is being recorded"""
    return ENABLED and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()


def train_active(x):
    """the channels-last TRAINING engine applies (archs/nhwc_train.py): fp32 GPU tensor, autograd recording, fp32 arithmetic"""
    from . import nhwc_train
    return ENABLED and nhwc_train.ENABLED and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and not BF16


def is_nhwc_view(x):
    """logical NCHW tensor whose storage is contiguous [N,H,W,C]"""
    return x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()


AMAX_ATTR = '_mrefsr_amax'   # python attribute of an engine-produced tensor: the device word holding max |tensor| (hip.amax_slot)


def _keep_amax(src, dst):
    a = getattr(src, AMAX_ATTR, None)
    if a is not None:
        setattr(dst, AMAX_ATTR, a)
    return dst


def to_nhwc(x):
    """logical NCHW -> contiguous [N,H,W,C] (free for a channels_last tensor, one transposition otherwise)"""
    return _keep_amax(x, x.permute(0, 2, 3, 1).contiguous())


def as_nchw(x):
    """[N,H,W,C] storage -> logical NCHW view (no copy)"""
    return _keep_amax(x, x.permute(0, 3, 1, 2))


def image_to_nhwc4(img, mean=None, std=None, range_norm=False):
    """[N,3,H,W] image -> [N,H,W,4] (((img + 1) / 2 if range_norm, then (. - mean) / std if mean is given: vgg_arch.py:150-153 of the
    reference) in channels 0..2, zero in channel 3): the conv kernel reads 16-byte channel vectors, the packed weights carry a zero
    fourth input channel.  fp32 inference: one HIP pass (ATen's operations in ATen's order: the same bits)"""
    n, c, h, w = img.shape
    if (c == 3 and img.dtype == torch.float32 and img.is_cuda and img.is_contiguous() and not storing16() and not img.requires_grad
            and (mean is None or (mean.dtype == torch.float32 and mean.is_contiguous() and std.is_contiguous()))):
        return hip.image_to_nhwc4(img, mean, std, range_norm)
    if range_norm:
        img = (img + 1) / 2
    if mean is not None:
        img = (img - mean) / std
    if storing16():   # bf16 storage: 16-byte channel vectors are 8 channels
        out = torch.zeros((n, h, w, (c + 7) // 8 * 8), device=img.device, dtype=torch.bfloat16)
    else:
        out = torch.zeros((n, h, w, (c + 3) // 4 * 4), device=img.device, dtype=torch.float32)
    out[..., :c] = img.permute(0, 2, 3, 1)
    return out


def _conv_ok(conv):
    k = conv.kernel_size
    return (isinstance(conv, nn.Conv2d) and k in ((1, 1), (3, 3)) and conv.stride == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding == (k[0] // 2, k[0] // 2) and conv.padding_mode == 'zeros')


def conv(mod, x1, x2=None, slope=None, prelu=None, pre=None, residual=None, epilogue=0, cin_slice=None, bias=True, out=None, amax=True):
    """``mod`` (nn.Conv2d, 1x1 or 3x3 'same') applied to cat([x1, x2], channels), NHWC in / NHWC out.

    slope: LeakyReLU slope (0.0 = ReLU, None = no activation); prelu: nn.PReLU (single parameter) instead;
    pre [Np,H,W,Cout] is added before the activation (batch-broadcast), residual after it;
    epilogue 1 = MaxPool2d(2,2), 2 = PixelShuffle(2); cin_slice=(a, b) uses weight[:, a:b] only.
    amax=False: the caller knows that no 3x3 convolution reads the result (attention operands, modulation terms, the final image):
    the launch does not measure max |out| (a tensor without the word is measured on demand, so this is only ever a saving)."""
    if not _conv_ok(mod):
        raise NotImplementedError(f'nhwc.conv: unsupported convolution {mod}')
    if torch.is_grad_enabled():   # a graph is being recorded: one autograd node per fused launch (archs/nhwc_train.py)
        from . import nhwc_train
        if nhwc_train.recording(mod.weight, mod.bias, x1, x2, pre, residual):
            if out is not None:
                raise NotImplementedError('nhwc.conv: out= under autograd')
            return nhwc_train.conv(mod, x1, x2, slope, prelu, pre, residual, epilogue, cin_slice, bias)
    slope_ptr = None
    if prelu is not None:
        if prelu.weight.numel() != 1:
            raise NotImplementedError('nhwc.conv: per-channel PReLU')
        slope_ptr = prelu.weight.detach()
    terms = 6 if (TERMS == 16 and hip.is_range_free()) else TERMS   # re-run of a batch that left the fp16 range
    if terms == 16 and mod.kernel_size[0] == 3 and x1.dtype == torch.float32:
        nimg = max(x1.shape[0], x2.shape[0] if x2 is not None else 0, residual.shape[0] if residual is not None else 0)
        cin = x1.shape[3] + (x2.shape[3] if x2 is not None else 0)
        if wino_applies(nimg, x1.shape[1], x1.shape[2], cin, mod.out_channels, max(x1.stride(2), x2.stride(2) if x2 is not None else 0), epilogue):
            terms = 17
    packed = hip.packed_weight(mod.weight, cin_slice, terms)
    b = mod.bias.detach() if (bias and mod.bias is not None) else None
    # every fp32-equivalent launch measures max |out| in its epilogue (a zeroed device word of hip's pool); the Winograd launch that
    # reads the tensor takes that word as its input scale
    slot = hip.amax_slot(x1.device) if (amax and terms in (16, 17) and x1.dtype == torch.float32 and WINO_INSCALE) else None
    y = hip.conv_nhwc(x1, packed, b, mod.out_channels, mod.kernel_size[0], x2=x2, pre=pre, residual=residual,
                      act=slope is not None or prelu is not None, slope=0.0 if slope is None else slope,
                      slope_ptr=slope_ptr, epilogue=epilogue, out=out,
                      in_amax=wino_in_amax(x1, x2) if (terms == 17 and WINO_INSCALE) else None, out_amax=slot)
    if slot is not None:
        setattr(y, AMAX_ATTR, slot)
    return y


# Input scale of the Winograd launches.  The transform B^T d B is split into fp16 high + low terms AFTER it is formed; the low term
# of a value below 2^-3 is an fp16 subnormal (absolute error 2^-25), which shows against activations of 1e-2 and less.
# The kernels therefore multiply their input by the power of two that brings max |x| into [2^11, 2^12) and the result by its inverse
# (csrc/conv_wino.hip: in_amax) -- exact, and the low term is a normal number down to 2^-14 of the maximum.
# Round 6: max |x| is the CURRENT tensor's, measured by the kernel that produced it (every convolution and DCN launch of the engine
# writes max |out| into a device word in its epilogue -- hip.amax_slot -- and the word travels with the tensor as a python
# attribute): no reduction launches, no calibration that could go stale (round 5 measured once per layer on the first batch).  A
# tensor that did not come out of the engine (a caller's own tensor, the result of a torch op) is measured here, per call; the
# count of such measurements is AMAX_MEASURED (0 on the benchmark path: tests/test_archs_gpu.py).  MREFSR_WINO_INSCALE=0: no scaling.
WINO_INSCALE = os.environ.get('MREFSR_WINO_INSCALE', '1') != '0'
AMAX_MEASURED = [0]


def reset_wino_calibration():
    """(kept for callers of round 5's interface: there is no calibration state any more)"""


def amax_of(x):
    """the device word with max |x|: the producer's, or measured now (two torch launches) for a tensor the engine did not produce"""
    a = getattr(x, AMAX_ATTR, None)
    if a is None:
        AMAX_MEASURED[0] += 1
        a = x.detach().abs().amax().float().reshape(1)
        try:
            setattr(x, AMAX_ATTR, a)
        except AttributeError:
            pass
    return a


def wino_in_amax(x1, x2=None):
    """the input maximum of a launch over cat([x1, x2]) as a 1-element device tensor"""
    a = amax_of(x1)
    return a if x2 is None else torch.maximum(a, amax_of(x2))


def add_(h, x):
    """h += x in place (the skip connections around the trunks, ref_mrapa_restoration_arch.py:228-258) with the bound
    max |h + x| <= max |h| + max |x| as the sum's maximum (a 1-element add: the scale only has to be an upper bound within 2x)"""
    ah, ax = getattr(h, AMAX_ATTR, None), getattr(x, AMAX_ATTR, None)
    h.add_(x)
    if ah is not None and ax is not None:
        setattr(h, AMAX_ATTR, ah + ax)
    elif hasattr(h, AMAX_ATTR):
        delattr(h, AMAX_ATTR)
    return h


def res_chain(blocks, x):
    """nn.Sequential of ResidualBlockNoBN (res_scale 1): x + conv2(relu(conv1(x))), two launches per block"""
    if torch.is_grad_enabled() and all(_conv_ok(b.conv1) and _conv_ok(b.conv2) for b in blocks):
        from . import nhwc_train
        if nhwc_train.recording(x, *[b.conv1.weight for b in blocks]):
            y = nhwc_train.reschain(blocks, x)   # the whole trunk as one autograd node (batched weight gradients)
            if y is not None:
                return y
    for blk in blocks:
        if blk.res_scale != 1:
            raise NotImplementedError('nhwc.res_chain: res_scale != 1')
        y = None
        if torch.is_grad_enabled() and _conv_ok(blk.conv1) and _conv_ok(blk.conv2):
            from . import nhwc_train
            if nhwc_train.recording(blk.conv1.weight, blk.conv2.weight, x):
                y = nhwc_train.resblock(blk, x)   # one autograd node per block: the gradient add of the skip rides on conv1's dgrad
        x = y if y is not None else conv(blk.conv2, conv(blk.conv1, x, slope=0.0), residual=x)
    return x


def stack_ok(layers):
    """the VGG stacks the engine handles: 3x3 'same' convolutions, ReLU, MaxPool2d(2,2)"""
    for layer in layers.children():
        if isinstance(layer, nn.Conv2d):
            if not (_conv_ok(layer) and layer.kernel_size == (3, 3)):
                return False
        elif isinstance(layer, nn.MaxPool2d):
            if not (layer.kernel_size in (2, (2, 2)) and layer.stride in (2, (2, 2)) and layer.padding in (0, (0, 0))
                    and not layer.ceil_mode):
                return False
        elif not isinstance(layer, nn.ReLU):
            return False
    return True


def vgg_stack(layers, x, taps=None):
    """Conv2d / ReLU / MaxPool2d sequence on an NHWC tensor; conv+ReLU(+pool) fuse into one launch.
    Returns the final NHWC tensor, or {name: NHWC tensor} for the names in ``taps``."""
    out = {}
    items = list(layers._modules.items())
    i = 0
    while i < len(items):
        name, layer = items[i]
        if isinstance(layer, nn.Conv2d):
            relu = i + 1 < len(items) and isinstance(items[i + 1][1], nn.ReLU) and not (taps and name in taps)
            pool = (relu and i + 2 < len(items) and isinstance(items[i + 2][1], nn.MaxPool2d)
                    and not (taps and items[i + 1][0] in taps) and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0)
            x = conv(layer, x, slope=0.0 if relu else None, epilogue=1 if pool else 0)
            step = 3 if pool else (2 if relu else 1)
            name = items[i + step - 1][0]
            i += step
        elif isinstance(layer, nn.MaxPool2d):
            x = to_nhwc(torch.nn.functional.max_pool2d(as_nchw(x), 2, 2))
            i += 1
        else:  # a ReLU that could not be fused (its convolution output is tapped)
            x = torch.relu(x)
            i += 1
        if taps and name in taps:
            out[name] = x
    return out if taps else x
