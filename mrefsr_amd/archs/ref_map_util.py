"""feature_match_index with the signature of basicsr/archs/ref_map_util.py:26-86, on the fused
HIP correlation kernel (mrefsr_corr_top1_f32): no unfold, no (n_ref x n_query) correlation matrix,
no chunk loop.  Indices are bit-identical to oracle/mrefsr_oracle.c and equal to the reference's
on every golden vector."""
import os

import torch

from .. import hip

# MREFSR_CORR_EXACT=1 forces the single-pass exact fp32-MFMA kernel; default is a pre-filter on the 16-bit
# matrix pipe + exact re-scoring (same bits out; csrc/corr_prefilter.hip)
_EXACT_ONLY = os.environ.get('MREFSR_CORR_EXACT', '0') == '1'
# The pre-filter operand: one fp16 plane (one MFMA per term) with a window derived from the MEASURED rounding error
# of the feature maps (hip.prefilter_window: ~2x tighter than the worst case, which made this variant overflow its
# candidate lists on maps full of near-ties) -- the default for 256-channel features; MREFSR_CORR_FP16=0 selects
# the bf16 two-term split (three MFMAs per term, fixed window), MREFSR_CORR_WINDOW=worst the fp16 operand with the
# worst-case window (A/B measurements).
_BF16_PREFILTER = os.environ.get('MREFSR_CORR_FP16', '1') == '0'
_WORST_CASE_WINDOW = os.environ.get('MREFSR_CORR_WINDOW', '') == 'worst'


def sample_patches(inputs, patch_size=3, stride=1):
    """(c,h,w) -> (c, patch, patch, n_patches), row-major patches (ref_map_util.py:4-23).  Not used
    by the path (the kernel reads 3x3 windows in place); kept for API parity."""
    c, h, w = inputs.shape
    return inputs.unfold(1, patch_size, stride).unfold(2, patch_size, stride)\
        .reshape(c, -1, patch_size, patch_size).permute(0, 2, 3, 1)


def feature_match_index(feat_input, feat_ref, patch_size=3, input_stride=1, ref_stride=1, is_norm=True,
                        norm_input=False):
    """feat_input (c,h,w), feat_ref (c,h',w') -> (max_idx int64, max_val fp32), both ((h-p)/s_in+1, (w-p)/s_in+1); max_idx indexes the
    reference patches row-major.  The maps are used as given (the caller normalises them, corres_generation_arch.py:57-59).
    patch_size 3 / strides 1 / equal sizes (the path) run the fused MFMA kernels, anything else the general kernel."""
    if patch_size != 3 or input_stride != 1 or ref_stride != 1 or feat_input.shape != feat_ref.shape or feat_input.shape[0] > 256:
        # the general form (no shipped yml uses it): scalar kernel with the same defined operation order
        return hip.feature_match_index_generic(feat_input.contiguous(), feat_ref.contiguous(), patch_size, input_stride, ref_stride,
                                               is_norm, norm_input)
    c, h, w = feat_input.shape
    # un-normalised maps of unknown scale: the pre-filter's error window is proven for the path's
    # per-pixel-normalised maps; this general entry uses the single-pass exact kernel
    y_in, n2_in = hip.pixnorm(feat_input.unsqueeze(0).contiguous(), normalize=False)
    y_ref, n2_ref = hip.pixnorm(feat_ref.unsqueeze(0).contiguous(), normalize=False)
    nrm_in, _ = hip.patch_norm(n2_in)
    _, inv_ref = hip.patch_norm(n2_ref)
    if not is_norm:
        inv_ref = torch.ones_like(inv_ref)
    if not norm_input:
        nrm_in = torch.ones_like(nrm_in)
    idx, val = hip.corr_top1(y_in, y_ref, inv_ref, nrm_in, h, w)
    return idx[0], val[0]


def match_normalised_batch(feat_in, feat_ref):
    """The batched form the path uses: feat_in [B,C,h,w], feat_ref [K*B,C,h,w] (k-major), raw
    extractor outputs.  Per-pixel normalisation (corres_generation_arch.py:57-59) is fused into
    the layout pass.  Returns max_idx [K*B,h-2,w-2] int64."""
    h, w = feat_in.shape[2:]

    # pre-filter operand: one fp16 plane (256 channels, the path) or the bf16 hi|lo split
    fmt = 'fp16' if (hip.padded_channels(feat_in.shape[1]) == 256 and not _BF16_PREFILTER) else 'bf16'

    def prep(f, split):
        # channels-last extractor outputs (archs/nhwc.py) are read in place; NCHW ones as before
        err = split and fmt == 'fp16'
        if not f.is_contiguous() and f.permute(0, 2, 3, 1).is_contiguous():
            return hip.pixnorm(f.permute(0, 2, 3, 1), normalize=True, want_bf16_split=split, nhwc=True, split=fmt, want_err=err)
        return hip.pixnorm(f.contiguous(), normalize=True, want_bf16_split=split, split=fmt, want_err=err)

    tau = None
    if _EXACT_ONLY:
        y_in, n2_in = prep(feat_in, False)
        y_ref, n2_ref = prep(feat_ref, False)
        bf_in = bf_ref = None
    elif fmt == 'fp16':
        y_in, n2_in, bf_in, d2_in = prep(feat_in, True)
        y_ref, n2_ref, bf_ref, d2_ref = prep(feat_ref, True)
    else:
        y_in, n2_in, bf_in = prep(feat_in, True)
        y_ref, n2_ref, bf_ref = prep(feat_ref, True)
    nrm_in, _ = hip.patch_norm(n2_in)
    _, inv_ref = hip.patch_norm(n2_ref)
    if fmt == 'fp16' and not _EXACT_ONLY and not _WORST_CASE_WINDOW:
        tau = hip.prefilter_window(nrm_in, inv_ref, d2_in, d2_ref)   # data-dependent, proven window
    idx, _ = hip.corr_top1(y_in, y_ref, inv_ref, nrm_in, h, w, want_val=False, ybf_in=bf_in, ybf_ref=bf_ref, tau=tau)
    return idx
