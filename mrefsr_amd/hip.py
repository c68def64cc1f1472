"""torch-tensor level bindings of the C ABI (include/mrefsr_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the stream; the library
gets raw device pointers + ``torch.cuda.current_stream().cuda_stream``.  No CPU path: a non-GPU
tensor raises (the reference's native ops raise NotImplementedError on CPU tensors as well,
basicsr/ops/dcn/deform_conv.py:61-62).
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import DcnShape


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-kernel timing: HIP events recorded on the very stream the kernel is launched on
# (torch's current stream).  bench.py turns it on to measure the correlation kernel inside the
# timed end-to-end step; off by default (no events, no overhead).
_timing = {'on': False, 'detail': False, 'events': {}, 'work': {}}


def set_kernel_timing(on=True, detail=False):
    """on: time the correlation call; detail: also every conv_nhwc / dcn_fwd launch (hundreds of event
    pairs per step -- bench.py does that in an extra, untimed step)"""
    _timing['on'] = bool(on)
    _timing['detail'] = bool(on and detail)
    _timing['events'] = {}
    _timing['work'] = {}
    _timing['bytes'] = {}


def kernel_work():
    """{kernel: algorithmic FLOPs summed over the timed launches}"""
    return dict(_timing['work'])


def kernel_bytes():
    """{kernel: algorithmic bytes (every operand and result once) summed over the timed launches}"""
    return dict(_timing.get('bytes', {}))


def kernel_timings():
    """{kernel: [ms, ...]} -- call after torch.cuda.synchronize()."""
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in _timing['events'].items()}


class _timed:
    def __init__(self, name, flops=0.0, detail=False, nbytes=0.0):
        self.name, self.flops, self.nbytes = name, flops, nbytes
        # (an event recorded inside a hipGraph capture is a graph node, not a timestamp: elapsed_time on it is an invalid handle)
        self.active = _timing['on'] and (_timing['detail'] or not detail) and not torch.cuda.is_current_stream_capturing()

    def __enter__(self):
        if self.active:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.active:
            self.b.record()
            _timing['events'].setdefault(self.name, []).append((self.a, self.b))
            _timing['work'][self.name] = _timing['work'].get(self.name, 0.0) + self.flops
            _timing.setdefault('bytes', {})[self.name] = _timing.get('bytes', {}).get(self.name, 0.0) + self.nbytes


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk(name, *tensors, dtype=torch.float32):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NotImplementedError(f'{name}: tensor on {t.device}; mrefsr_amd has no CPU path (HIP kernels only)')
        if t.dtype != dtype:
            raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
        if not t.is_contiguous():
            raise ValueError(f'{name}: tensor must be contiguous')


# ------------------------------------------------------------------ correlation path
def padded_channels(c):
    r = _lib.load().mrefsr_corr_padded_channels(int(c))
    if r < 0:
        raise _lib.MrefsrHipError(_lib.load().mrefsr_last_error().decode())
    return r


def pixnorm(x, normalize=True, want_bf16_split=False, nhwc=False, split='bf16', want_err=False):
    """x [N,C,h,w] (or [N,h,w,C] with nhwc=True) -> (y [N,h*w,Cp] split layout, n2 [N,h,w][, pre-filter operand][, d2]).
    want_bf16_split: also return the pre-filter operand -- split='bf16': [N,h*w,2,Cp] bf16 hi|lo;
    split='fp16': [N,h*w,Cp] float16 (the single-plane pre-filter, Cp = 256 only).
    want_err: also d2 [N,h,w], the squared norm of each pixel vector's fp16 rounding error (prefilter_window)."""
    x16 = x.dtype == torch.bfloat16
    if x16 and not nhwc:
        raise TypeError('pixnorm: bf16 feature maps are accepted channels-last only')
    _chk('pixnorm', x, dtype=x.dtype if x16 else torch.float32)
    if nhwc:
        n, h, w, c = x.shape
    else:
        n, c, h, w = x.shape
    cp = padded_channels(c)
    y = torch.empty((n, h * w, cp), device=x.device, dtype=torch.float32)
    n2 = torch.empty((n, h, w), device=x.device, dtype=torch.float32)
    ybf = None
    if want_bf16_split:
        # + (6 image rows + 16 pixels) of slack: the pre-filter's LDS-DMA staging reads edge tiles
        # (8 rows x 16 pixels from an origin <= (h-3, w-3)) unclamped: include/mrefsr_hip.h,
        # mrefsr_corr_top1_prefilter_f32
        if split == 'fp16':
            flat = torch.empty(n * h * w * cp + (6 * w + 16) * cp, device=x.device, dtype=torch.float16)
            ybf = flat[:n * h * w * cp].view(n, h * w, cp)
        else:
            flat = torch.empty(n * h * w * 2 * cp + (6 * w + 16) * 2 * cp, device=x.device, dtype=torch.bfloat16)
            ybf = flat[:n * h * w * 2 * cp].view(n, h * w, 2, cp)
    d2 = torch.empty((n, h, w), device=x.device, dtype=torch.float32) if want_err else None
    _lib.call('mrefsr_pixnorm_f32', _p(x), _p(y), _p(n2), _p(ybf), n, c, h * w, 1 if normalize else 0, (2 if x16 else 1) if nhwc else 0,
              1 if split == 'fp16' else 0, _p(d2), _stream())
    out = (y, n2, ybf) if want_bf16_split else (y, n2)
    return out + (d2,) if want_err else out


def prefilter_window(nrm_in, inv_ref, d2_in, d2_ref):
    """Per-query window of the fp16 pre-filter (include/mrefsr_hip.h, mrefsr_corr_top1_prefilter_f32): twice a proven
    bound on |approximate - canonical score|.  nrm_in [n_in,P'], inv_ref [n_pair,P'] from patch_norm of the n2 maps,
    d2_* from pixnorm(want_err=True).  Returns tau [n_pair, P'] (pair p uses input p % n_in)."""
    d_in, _ = patch_norm(d2_in)                      # sqrt(3x3 sum of d2) + 1e-5  >=  D
    d_ref, _ = patch_norm(d2_ref)
    n_in, n_pair = nrm_in.shape[0], inv_ref.shape[0]
    rho = (d_ref * inv_ref).flatten(1).amax(1)       # [n_pair]: largest relative rounding error of a reference patch
    rho = torch.nan_to_num(rho, nan=1.0, posinf=1.0).view(n_pair // n_in, n_in, 1, 1)
    dq, nq = d_in.unsqueeze(0), nrm_in.unsqueeze(0)  # [1,n_in,h-2,w-2]
    tau = 2.02 * (dq + (nq + 3.0 * dq) * rho) + 2.0e-4 * nq
    return tau.reshape(n_pair, *nrm_in.shape[1:]).contiguous()


def patch_norm(n2):
    """n2 [N,h,w] -> (norm+1e-5 [N,h-2,w-2], 1/(norm+1e-5))."""
    _chk('patch_norm', n2)
    n, h, w = n2.shape
    ne = torch.empty((n, h - 2, w - 2), device=n2.device, dtype=torch.float32)
    inv = torch.empty_like(ne)
    _lib.call('mrefsr_patch_norm_f32', _p(n2), _p(ne), _p(inv), n, h, w, _stream())
    return ne, inv


def corr_top1(y_in, y_ref, inv_ref, nrm_in, h, w, want_val=True, ybf_in=None, ybf_ref=None, tau=None):
    """y_in [n_in,h*w,Cp], y_ref [n_pair,h*w,Cp] -> (max_idx int64 [n_pair,h-2,w-2], max_val|None).
    With the bf16 / fp16 pre-filter operands given, the pre-filter + exact re-scoring path runs (same bits out);
    tau (fp16 operands only): the per-query window from prefilter_window(), else the worst-case window."""
    _chk('corr_top1', y_in, y_ref, inv_ref, nrm_in)
    n_in, hw, cp = y_in.shape
    n_pair = y_ref.shape[0]
    if hw != h * w or y_ref.shape[1] != hw or y_ref.shape[2] != cp:
        raise ValueError('corr_top1: input / reference feature sizes differ '
                         '(the path assumes equal sizes: corres_generation_arch.py:33-35)')
    if n_pair % n_in:
        raise ValueError(f'corr_top1: n_pair={n_pair} not a multiple of n_in={n_in}')
    idx = torch.empty((n_pair, h - 2, w - 2), device=y_in.device, dtype=torch.int64)
    val = torch.empty((n_pair, h - 2, w - 2), device=y_in.device, dtype=torch.float32) if want_val else None
    if ybf_in is not None and ybf_ref is not None:
        fmt = 1 if ybf_in.dtype == torch.float16 else 0
        _chk('corr_top1', ybf_in, ybf_ref, dtype=torch.float16 if fmt else torch.bfloat16)
        if tau is not None:
            _chk('corr_top1', tau)
            if not fmt or tuple(tau.shape) != (n_pair, h - 2, w - 2):
                raise ValueError('corr_top1: tau is [n_pair,h-2,w-2] and belongs to the fp16 pre-filter operand')
        need = _lib.load().mrefsr_corr_workspace_bytes(n_pair, h, w)
        ws = _workspace(y_in.device, need)   # (pooled per device and stream with the DCN's: ~400 MB at the benchmark size, used
                                             #  and dropped within this call; a fresh torch.empty per call held a second copy
                                             #  of it alive in the allocator across the whole pass)
        _timing['last_corr_ws'] = (ws, n_pair, (h - 2) * (w - 2)) if _timing.get('keep_ws') else None
        with _timed('corr_top1'):
            _lib.call('mrefsr_corr_top1_prefilter_f32', _p(y_in), _p(y_ref), _p(ybf_in), _p(ybf_ref), _p(inv_ref),
                      _p(nrm_in), _p(idx), _p(val), _p(ws), C.c_int64(need), n_in, n_pair, cp, h, w, fmt, _p(tau), _stream())
        return idx, val
    with _timed('corr_top1'):
        _lib.call('mrefsr_corr_top1_f32', _p(y_in), _p(y_ref), _p(inv_ref), _p(nrm_in), _p(idx), _p(val), n_in, n_pair,
                  cp, h, w, _stream())
    return idx, val


def feature_match_index_generic(feat_in, feat_ref, patch_size, input_stride, ref_stride, is_norm, norm_input):
    """feat_in (C,h,w), feat_ref (C,hr,wr) -> (max_idx int64 (nqy,nqx), max_val fp32): any patch size / strides / sizes"""
    _chk('feature_match_index', feat_in, feat_ref)
    c, h, w = feat_in.shape
    cr, hr, wr = feat_ref.shape
    if c != cr:
        raise ValueError('feature_match_index: channel counts differ')
    if min(h, w, hr, wr) < patch_size:
        raise ValueError('feature_match_index: a map is smaller than the patch')
    nqy, nqx = (h - patch_size) // input_stride + 1, (w - patch_size) // input_stride + 1
    idx = torch.empty((nqy, nqx), device=feat_in.device, dtype=torch.int64)
    val = torch.empty((nqy, nqx), device=feat_in.device, dtype=torch.float32)
    need = _lib.load().mrefsr_feature_match_index_workspace_bytes(h, w, hr, wr)
    ws = torch.empty(need, device=feat_in.device, dtype=torch.uint8)
    _lib.call('mrefsr_feature_match_index_f32', _p(feat_in), _p(feat_ref), c, h, w, hr, wr, int(patch_size), int(input_stride), int(ref_stride),
              1 if is_norm else 0, 1 if norm_input else 0, _p(idx), _p(val), _p(ws), C.c_int64(need), _stream())
    return idx, val


def offsets_from_idx(idx, h, w, scales=(1, 2, 4)):
    """idx int64 [N,h-2,w-2] -> dict scale -> [N,9,s*h,s*w,2] fp32 ([x,y])."""
    _chk('offsets_from_idx', idx, dtype=torch.int64)
    n = idx.shape[0]
    outs = {s: torch.empty((n, 9, s * h, s * w, 2), device=idx.device, dtype=torch.float32) for s in scales}
    _lib.call('mrefsr_offsets_from_idx_f32', _p(idx), _p(outs.get(1)), _p(outs.get(2)), _p(outs.get(4)), n, h, w,
              _stream())
    return outs


# ------------------------------------------------------------------ DynAgg glue
def dynagg_prep(om, pre, dg, abs_sum=None, om_bias=None, om_nhwc=False):
    """om [B,27dg,H,W] (or [B,H,W,27dg] with om_nhwc) -> planar (offset [B,18dg,H,W], mask [B,9dg,H,W])"""
    _chk('dynagg_prep', om, pre, om_bias)
    if om_nhwc:
        b, h, w, ch = om.shape
    else:
        b, ch, h, w = om.shape
    if ch != 27 * dg or tuple(pre.shape) != (b, 9, h, w, 2):
        raise ValueError(f'dynagg_prep: om {tuple(om.shape)} / pre {tuple(pre.shape)} inconsistent with dg={dg}')
    offset = torch.empty((b, 18 * dg, h, w), device=om.device, dtype=torch.float32)
    mask = torch.empty((b, 9 * dg, h, w), device=om.device, dtype=torch.float32)
    if abs_sum is not None:
        _chk('dynagg_prep', abs_sum, dtype=torch.float64)
    _lib.call('mrefsr_dynagg_prep_f32', _p(om), _p(om_bias), _p(pre), _p(offset), _p(mask), _p(abs_sum), b, dg, h, w,
              1 if om_nhwc else 0, _stream())
    return offset, mask


def dynagg_prep_bwd(g_offset, g_mask, mask, dg):
    _chk('dynagg_prep_bwd', g_offset, g_mask, mask)
    b, _, h, w = mask.shape
    g_om = torch.empty((b, 27 * dg, h, w), device=mask.device, dtype=torch.float32)
    _lib.call('mrefsr_dynagg_prep_bwd_f32', _p(g_offset), _p(g_mask), _p(mask), _p(g_om), b, dg, h, w, _stream())
    return g_om


def dynagg_prep_bwd_nhwc(g_offset, g_mask, mask, dg, want_bias=True):
    """dynagg_prep_bwd with the result channels-last [B,H,W,27dg] and, from the same pass, (bias gradient [27dg] | None, max |g_om| [1])"""
    _chk('dynagg_prep_bwd_nhwc', g_offset, g_mask, mask)
    b, _, h, w = mask.shape
    nc = 27 * dg
    g_om = torch.empty((b, h, w, nc), device=mask.device, dtype=torch.float32)
    z = zeros_f32(mask.device, nc + 1)
    _lib.call('mrefsr_dynagg_prep_bwd_nhwc_f32', _p(g_offset), _p(g_mask), _p(mask), _p(g_om), _p(z[:nc]) if want_bias else None, _p(z[nc:]), b, dg, h, w,
              _stream())
    return g_om, (z[:nc] if want_bias else None), z[nc:]


# ------------------------------------------------------------------ DCN
def dcn_shape(x, weight, stride, padding, dilation, groups, dg):
    def pair(v):
        return (v, v) if isinstance(v, int) else tuple(v)
    (sh, sw), (ph, pw), (dh, dw) = pair(stride), pair(padding), pair(dilation)
    b, c, h, w = x.shape
    co, cig, kh, kw = weight.shape
    if cig * groups != c:
        raise RuntimeError(f"Input shape and kernel channels won't match: ({c} vs {cig * groups}).")
    s = DcnShape(b, c, h, w, co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg)
    ho = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    wo = (w + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    return s, ho, wo


_ws_cache = {}


def _workspace(device, nbytes):
    # one buffer per (device, stream): dcn_fwd re-packs its split weights into this buffer on every call and then reads them,
    # which is ordered on ONE stream only -- two eager streams sharing a buffer would overwrite each other's packed weights.
    # Buffers of streams that were under hipGraph capture are flagged so that they can be freed with their graphs.
    capturing = torch.cuda.is_current_stream_capturing()
    key = (device.index, torch.cuda.current_stream().cuda_stream, capturing)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1), device=device, dtype=torch.uint8)
        _ws_cache[key] = buf
    return buf


def release_capture_workspaces():
    """drop the DCN workspaces that belonged to capture streams (call when the graphs that used them are destroyed)"""
    for k in [k for k in _ws_cache if k[2]]:
        del _ws_cache[k]
    for k in [k for k in _zero_chunks if k[2]]:
        del _zero_chunks[k]


def capture_refs():
    """the cached buffers (DCN workspaces, zero-filled accumulator chunks) whose addresses a hipGraph captured now may have baked
    in: whoever owns the graph keeps this list as long as the graph lives"""
    # (only the buffers that were handed out UNDER a capture: an eager regrowth of some other stream's workspace -- validation at
    #  another size, the correlation call's scratch -- does not move anything a graph baked in, and must not force a recapture)
    return [[t for k, t in _ws_cache.items() if k[2]], [c[0] for k, c in _zero_chunks.items() if k[2]], list(_wgrad_ws.values())]


def capture_ptrs():
    """addresses of the cached buffers of capture_refs(): a graph owner compares them before a replay (a regrown workspace has moved)"""
    return tuple(t.data_ptr() for group in capture_refs() for t in group)


_cap_state = [False, 0]


def capture_epoch():
    """(capturing, epoch): the epoch moves every time this is called on the other side of a hipGraph capture boundary than
    the call before -- device state prepared on one side (zero-filled accumulators, packed weight copies) is not replayed with
    a graph captured on the other, so its users key it by this epoch"""
    cap = torch.cuda.is_current_stream_capturing()
    if cap != _cap_state[0]:
        _cap_state[0] = cap
        _cap_state[1] += 1
    return cap, _cap_state[1]


_zero_chunks = {}
_ZCHUNK = 1 << 18
ZERO_POOL = os.environ.get('MREFSR_ZERO_POOL', '1') != '0'
_ZALIGN = 128         # floats: slices start on 512-byte boundaries like allocations of their own (float atomics into a slice at
                      # 16-byte granularity ran 17 % slower: act_bwd_nhwc 5.66 -> 4.87 ms per training step)


def zeros_f32(device, n):
    """n zero float32 (512-byte aligned) carved from a chunk that ONE fill launch zeroed: the accumulators of act_bwd_nhwc (bias
    gradient, PReLU slope gradient, max |g|: ~70 floats, 174 times per training step) each had a fill launch of their own.  A
    slice is handed out once and never re-zeroed, so it may be kept (autograd adopts the bias gradient as ``.grad``); the chunk
    lives as long as any of its slices.  Chunks are per stream and per side of a capture boundary (the fill that zeroes a chunk
    has to be part of the graph whose kernels accumulate into it)."""
    if not ZERO_POOL:
        return torch.zeros(n, device=device, dtype=torch.float32)
    cap, epoch = capture_epoch()
    key = (device.index, torch.cuda.current_stream().cuda_stream, cap)
    n4 = (n + _ZALIGN - 1) // _ZALIGN * _ZALIGN
    ch = _zero_chunks.get(key)
    if ch is None or ch[2] != epoch or ch[1] + n4 > ch[0].numel():
        ch = _zero_chunks[key] = [torch.zeros(max(_ZCHUNK, n4), device=device, dtype=torch.float32), 0, epoch]
    out = ch[0][ch[1]:ch[1] + n]
    ch[1] += n4
    return out


def dcn_mfma_eligible(c, co, dg, k=3):
    """True when mrefsr_dcn_fwd_f32 runs its fused gather+MFMA kernel for a stride-1 'same' k x k DCN"""
    s = DcnShape(1, c, 8, 8, co, k, k, 1, 1, k // 2, k // 2, 1, 1, 1, dg)
    return _lib.load().mrefsr_dcn_fwd_workspace_bytes(C.byref(s)) > 0


def dcn_fwd(x, offset, mask, weight, bias, stride, padding, dilation, groups, dg, act_slope=1.0, nhwc_gather=True,
            channels_last=False, bf16_arith=False, range_free=False, out_amax=None):
    """x NCHW.  For MFMA-eligible shapes the input is re-laid out to NHWC once (one HBM pass) so the
    deformable gather reads 16-byte channel vectors instead of scalar corners (nhwc_gather=False
    keeps the NCHW gather).  channels_last=True: x is given [B,H,W,C] and the result is [B,Ho,Wo,Co]
    (the inference path; MFMA-eligible shapes only); offset / mask are planar either way."""
    if x.dtype in (torch.float16, torch.float64):   # the reference's other dtypes: portable kernels (mrefsr_dcn_fwd)
        if channels_last:
            raise TypeError('dcn_fwd: channels_last is an fp32 / bf16 fast path')
        _chk('dcn_fwd', x, offset, mask, weight, bias, dtype=x.dtype)
        s, ho, wo = dcn_shape(x, weight, stride, padding, dilation, groups, dg)
        kk = s.kh * s.kw
        if tuple(offset.shape) != (s.B, 2 * dg * kk, ho, wo) or (mask is not None and tuple(mask.shape) != (s.B, dg * kk, ho, wo)):
            raise RuntimeError(f'dcn_fwd: offset {tuple(offset.shape)} / mask shape mismatch')
        out = torch.empty((s.B, s.Co, ho, wo), device=x.device, dtype=x.dtype)
        _lib.call('mrefsr_dcn_fwd', _p(x), _p(offset), _p(mask), _p(weight), _p(bias), _p(out), C.byref(s), C.c_float(act_slope), _DT[x.dtype],
                  _stream())
        return out
    io16 = channels_last and x.dtype == torch.bfloat16
    _chk('dcn_fwd', x, dtype=x.dtype if io16 else torch.float32)
    _chk('dcn_fwd', offset, mask, weight, bias)
    range_free = range_free or _range_free[0]
    if io16 and not bf16_arith:
        raise TypeError('dcn_fwd: bf16 tensors go with bf16_arith=True')
    if channels_last:
        s, ho, wo = dcn_shape(x.permute(0, 3, 1, 2), weight, stride, padding, dilation, groups, dg)
        need = _lib.load().mrefsr_dcn_fwd_workspace_bytes(C.byref(s))
        if need <= 0:
            raise _lib.MrefsrHipError('dcn_fwd(channels_last): shape is not eligible for the fused MFMA kernel')
        if tuple(offset.shape) != (s.B, 2 * dg * 9, ho, wo) or (mask is not None and tuple(mask.shape) != (s.B, dg * 9, ho, wo)):
            raise RuntimeError(f'dcn_fwd: offset {tuple(offset.shape)} / mask shape mismatch')
        out = torch.empty((s.B, ho, wo, s.Co), device=x.device, dtype=x.dtype)
        with _timed('dcn_fwd', 2.0 * s.B * ho * wo * s.C * s.Co * 9, detail=True):
            _lib.call('mrefsr_dcn_fwd_amax_f32', _p(x), _p(offset), _p(mask), _p(weight), _p(bias), _p(out), C.byref(s),
                      C.c_float(act_slope), (23 if io16 else 7) if bf16_arith else (11 if range_free else 3), _p(_workspace(x.device, need)), C.c_int64(need),
                      _p(_range_flag(x.device)), _p(out_amax), _stream())
        return out
    s, ho, wo = dcn_shape(x, weight, stride, padding, dilation, groups, dg)
    kk = s.kh * s.kw
    if tuple(offset.shape) != (s.B, 2 * dg * kk, ho, wo):
        raise RuntimeError(f'dcn_fwd: offset shape {tuple(offset.shape)} != {(s.B, 2 * dg * kk, ho, wo)}')
    if mask is not None and tuple(mask.shape) != (s.B, dg * kk, ho, wo):
        raise RuntimeError(f'dcn_fwd: mask shape {tuple(mask.shape)} != {(s.B, dg * kk, ho, wo)}')
    out = torch.empty((s.B, s.Co, ho, wo), device=x.device, dtype=torch.float32)
    need = _lib.load().mrefsr_dcn_fwd_workspace_bytes(C.byref(s))
    ws = _workspace(x.device, need) if need > 0 else None
    nhwc = (9 if range_free else 1) if (need > 0 and nhwc_gather) else 0
    xin = x.permute(0, 2, 3, 1).contiguous() if nhwc else x
    _lib.call('mrefsr_dcn_fwd_f32', _p(xin), _p(offset), _p(mask), _p(weight), _p(bias), _p(out), C.byref(s),
              C.c_float(act_slope), nhwc, _p(ws), C.c_int64(need), _p(_range_flag(x.device)) if nhwc else None, _stream())
    return out


def dcn_im2col(x, offset, mask, weight_shape, stride, padding, dilation, groups, dg):
    _chk('dcn_im2col', x, offset, mask, dtype=x.dtype if x.dtype in (torch.float16, torch.float64) else torch.float32)
    fake_w = torch.empty(weight_shape, device='meta')
    s, ho, wo = dcn_shape(x, fake_w, stride, padding, dilation, groups, dg)
    col = torch.empty((s.B, s.C * s.kh * s.kw, ho * wo), device=x.device, dtype=x.dtype)
    if x.dtype == torch.float32:
        _lib.call('mrefsr_dcn_im2col_f32', _p(x), _p(offset), _p(mask), _p(col), C.byref(s), _stream())
    else:
        _lib.call('mrefsr_dcn_im2col', _p(x), _p(offset), _p(mask), _p(col), C.byref(s), _DT[x.dtype], _stream())
    return col


def dcn_col2im(grad_col, x, offset, mask, weight_shape, stride, padding, dilation, groups, dg, need_grad_x=True):
    if x.dtype == torch.float16:   # gradients of an f16 call are accumulated in f32 (atomics), then rounded
        gx, goff, gmask = dcn_col2im(grad_col.float(), x.float(), offset.float(), None if mask is None else mask.float(), weight_shape, stride,
                                     padding, dilation, groups, dg, need_grad_x)
        return (None if gx is None else gx.half()), goff.half(), (None if gmask is None else gmask.half())
    _chk('dcn_col2im', grad_col, x, offset, mask, dtype=x.dtype if x.dtype == torch.float64 else torch.float32)
    fake_w = torch.empty(weight_shape, device='meta')
    s, _, _ = dcn_shape(x, fake_w, stride, padding, dilation, groups, dg)
    gx = torch.zeros_like(x) if need_grad_x else None
    goff = torch.empty_like(offset)
    gmask = torch.empty_like(mask) if mask is not None else None
    if x.dtype == torch.float64:
        _lib.call('mrefsr_dcn_col2im', _p(grad_col), _p(x), _p(offset), _p(mask), _p(gx), _p(goff), _p(gmask), C.byref(s), _DT[x.dtype], _stream())
    else:
        _lib.call('mrefsr_dcn_col2im_f32', _p(grad_col), _p(x), _p(offset), _p(mask), _p(gx), _p(goff), _p(gmask),
                  C.byref(s), _stream())
    return gx, goff, gmask


def _dcn_bwd_shapes(name, g_out, x, offset, mask, dg):
    """the backward kernels index offset / mask / g_out from x's B, H, W and dg: a mismatch would read out of bounds"""
    b, h, w, c = x.shape
    if g_out.dim() != 4 or tuple(g_out.shape[:3]) != (b, h, w):
        raise ValueError(f'{name}: g_out {tuple(g_out.shape)} does not match x {tuple(x.shape)} (channels-last, stride 1)')
    if dg < 1 or c % dg or tuple(offset.shape) != (b, 18 * dg, h, w):
        raise ValueError(f'{name}: offset {tuple(offset.shape)} != {(b, 18 * dg, h, w)} for x {tuple(x.shape)}, deformable groups {dg}')
    if mask is not None and tuple(mask.shape) != (b, 9 * dg, h, w):
        raise ValueError(f'{name}: mask {tuple(mask.shape)} != {(b, 9 * dg, h, w)}')


def dcn_bwd_data(g_out, x, offset, mask, packed_wT, dg, g_amax=None, need_grad_x=True):
    """Fused backward of DCNv2 (3x3, stride 1, pad 1) w.r.t. offset, mask and input (mrefsr_dcn_bwd_data_f32): g_out [B,H,W,Co] and
    x [B,H,W,C] channels-last, offset / mask planar; packed_wT = conv_pack_view(weight, terms=16, dgrad='T', wscale=...);
    g_amax: device float max |g_out| (None: no scaling).  -> (grad_x planar [B,C,H,W] | None, grad_offset, grad_mask | None)"""
    _chk('dcn_bwd_data', g_out, x, offset, mask, g_amax)
    b, h, w, c = x.shape
    co = g_out.shape[3]
    _dcn_bwd_shapes('dcn_bwd_data', g_out, x, offset, mask, dg)
    if packed_wT.terms != 16:
        raise ValueError('dcn_bwd_data: the transposed weights packed with terms=16 expected')
    s = _lib.DcnShape(b, c, h, w, co, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg)
    gx = torch.zeros((b, c, h, w), device=x.device, dtype=torch.float32) if need_grad_x else None
    goff = torch.empty_like(offset)
    gmask = torch.empty_like(mask) if mask is not None else None
    _lib.call('mrefsr_dcn_bwd_data_f32', _p(g_out), _p(x), _p(offset), _p(mask), _p(packed_wT.data), C.c_float(packed_wT.wscale), _p(g_amax),
              _p(gx), _p(goff), _p(gmask), C.byref(s), _stream())
    return gx, goff, gmask


def conv_wgrad1x1(x, g, cin, cout, g_amax):
    """weight gradient [cout, cin, 1, 1] of a 1x1 convolution from channels-last storage (x [..., >= cin], g [..., >= cout], both
    pixel-contiguous views): mrefsr_conv_wgrad1x1_f32"""
    _chk('conv_wgrad1x1', g_amax)
    ld_x, ld_g = _nhwc_ld('x', x), _nhwc_ld('g', g)
    pixels = x.shape[0] * x.shape[1] * x.shape[2]
    nbytes = _lib.load().mrefsr_conv_wgrad1x1_workspace_bytes(pixels, cout, cin)
    ws = _wgrad_workspace(x.device, nbytes)
    gw = torch.empty((cout, cin, 1, 1), device=x.device, dtype=torch.float32)
    _lib.call('mrefsr_conv_wgrad1x1_f32', _p(g), _p(x), _p(g_amax), _p(gw), _p(ws), nbytes, pixels, cout, ld_g, cin, ld_x,
              _p(_range_flag(x.device)), _stream())
    return gw


def dcn_bwd_weight(g_out, x, offset, mask, cout, dg, g_amax=None):
    """d weight [Co,C,3,3] of DCNv2 (3x3, stride 1, pad 1) with the columns re-gathered inside the GEMM (mrefsr_dcn_bwd_weight_f32);
    tensors as dcn_bwd_data"""
    _chk('dcn_bwd_weight', g_out, x, offset, mask, g_amax)
    b, h, w, c = x.shape
    _dcn_bwd_shapes('dcn_bwd_weight', g_out, x, offset, mask, dg)
    s = _lib.DcnShape(b, c, h, w, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg)
    nbytes = _lib.load().mrefsr_dcn_bwd_weight_workspace_bytes(C.byref(s))
    ws = _wgrad_workspace(x.device, nbytes)
    gw = torch.empty((cout, c, 3, 3), device=x.device, dtype=torch.float32)
    _lib.call('mrefsr_dcn_bwd_weight_f32', _p(g_out), _p(x), _p(offset), _p(mask), _p(g_amax), _p(gw), _p(ws), nbytes, C.byref(s),
              _p(_range_flag(x.device)), _stream())
    return gw


# ------------------------------------------------------------------ attention core
def mrattn_fwd(q, emb, ass, t, want_prob=True, t_major=False):
    """q [N,c,H,W], emb [N*T,c,H,W], ass [N*T,c2,H,W] -> (out [N,c2,H,W], prob [N,T,H,W]|None)."""
    _chk('mrattn_fwd', q, emb, ass)
    n, c, h, w = q.shape
    c2 = ass.shape[1]
    if emb.shape[0] != n * t or ass.shape[0] != n * t or emb.shape[1] != c:
        raise ValueError('mrattn_fwd: inconsistent shapes')
    out = torch.empty((n, c2, h, w), device=q.device, dtype=torch.float32)
    prob = torch.empty((n, t, h, w), device=q.device, dtype=torch.float32) if want_prob else None
    _lib.call('mrefsr_mrattn_fwd_f32', _p(q), _p(emb), _p(ass), _p(out), _p(prob), n, t, c, c2, h * w, 1 if t_major else 0, _stream())
    return out, prob


def mrattn_fwd_nhwc(q, emb, ass, t, q_scale=None):
    """q [N,H,W,c], emb [t*N,H,W,c], ass [t*N,H,W,2c] (t-major) -> out [N,H,W,2c]; q_scale (fp32 tensors): the attention of
    q * q_scale, the products formed in the kernel as the separate pass would have rounded them"""
    b16 = q.dtype == torch.bfloat16
    if q_scale is not None and b16:
        raise TypeError('mrattn_fwd_nhwc: q_scale goes with fp32 tensors')
    _chk('mrattn_fwd_nhwc', q, emb, ass, dtype=q.dtype if b16 else torch.float32)
    n, h, w, c = q.shape
    if tuple(emb.shape) != (n * t, h, w, c) or tuple(ass.shape) != (n * t, h, w, 2 * c):
        raise ValueError('mrattn_fwd_nhwc: inconsistent shapes')
    out = torch.empty((n, h, w, 2 * c), device=q.device, dtype=q.dtype)
    with _timed('mrattn_fwd', (3.0 * t + 3.0) * c * h * w * q.element_size() * n, detail=True):   # "work" = algorithmic bytes
        if q_scale is not None:
            _lib.call('mrefsr_mrattn_fwd_nhwc_scaled_f32', _p(q), _p(emb), _p(ass), _p(out), n, t, c, h * w, C.c_float(q_scale), _stream())
        else:
            _lib.call('mrefsr_mrattn_fwd_nhwc_bf16' if b16 else 'mrefsr_mrattn_fwd_nhwc_f32', _p(q), _p(emb), _p(ass), _p(out), n, t, c, h * w, _stream())
    return out


def mrattn_bwd(q, emb, ass, prob, g_out, t, t_major=False):
    _chk('mrattn_bwd', q, emb, ass, prob, g_out)
    n, c, h, w = q.shape
    c2 = ass.shape[1]
    g_q, g_emb, g_ass = torch.empty_like(q), torch.empty_like(emb), torch.empty_like(ass)
    _lib.call('mrefsr_mrattn_bwd_f32', _p(q), _p(emb), _p(ass), _p(prob), _p(g_out), _p(g_q), _p(g_emb), _p(g_ass), n, t,
              c, c2, h * w, 1 if t_major else 0, _stream())
    return g_q, g_emb, g_ass


# ------------------------------------------------------------------ fused_act / upfirdn2d
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """basicsr.ops.fused_act ext entry: empty bias / ref tensors mean 'absent'
    (fused_bias_act_kernel.cu:63-64)."""
    if x.dtype not in _DT:
        raise TypeError(f'fused_bias_act: unsupported dtype {x.dtype}')
    x = x.contiguous()
    bias = bias.contiguous() if bias is not None and bias.numel() else None
    ref = ref.contiguous() if ref is not None and ref.numel() else None
    _chk('fused_bias_act', x, bias, ref, dtype=x.dtype)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    out = torch.empty_like(x)
    _lib.call('mrefsr_fused_bias_act', _p(x), _p(bias), _p(ref), _p(out), C.c_int64(x.numel()), step_b,
              0 if bias is None else bias.numel(), int(act), int(grad), C.c_float(alpha), C.c_float(scale), _DT[x.dtype],
              _stream())
    return out


def tail_bilinear_add(y_nhwc, x, scale=4):
    """y_nhwc [B,H,W,C] (a channel slice of a wider tensor is fine) + F.interpolate(x [B,C,h,w], None, scale, 'bilinear', False) -> [B,C,H,W]
    (ref_mrapa_restoration_arch.py:132-137): one pass, torch's interpolation bits"""
    x = x.contiguous()
    _chk('tail_bilinear_add', x)
    if not y_nhwc.is_cuda or y_nhwc.dtype != torch.float32:
        raise TypeError(f'tail_bilinear_add: y on {y_nhwc.device}, {y_nhwc.dtype} (a CUDA fp32 tensor, rows may be padded)')
    b, c, h, w = x.shape
    if tuple(y_nhwc.shape) != (b, h * scale, w * scale, c) or y_nhwc.stride(3) != 1 or y_nhwc.stride(1) != y_nhwc.stride(2) * w * scale \
            or y_nhwc.stride(0) != y_nhwc.stride(1) * h * scale:
        raise ValueError(f'tail_bilinear_add: y {tuple(y_nhwc.shape)} / strides {y_nhwc.stride()} against x {tuple(x.shape)}')
    out = torch.empty((b, c, h * scale, w * scale), device=x.device, dtype=torch.float32)
    _lib.call('mrefsr_tail_bilinear_add_f32', _p(y_nhwc), _p(x), _p(out), b, c, h, w, scale, y_nhwc.stride(2), _stream())
    return out


def bias_act_res_(x, bias, slope, residual=None, pre=None):
    """in place on x [N,C,H,W] fp32: x = lrelu(x + bias[c] + pre, slope) (+ residual); `pre` [Np,C,H,W]
    is broadcast over N / Np groups.  slope 1 = identity, 0 = ReLU."""
    _chk('bias_act_res', x, bias, residual, pre)
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    _lib.call('mrefsr_bias_act_res_f32', _p(x), _p(bias), _p(pre), C.c_int64(0 if pre is None else pre.shape[0]),
              _p(residual), _p(x), C.c_int64(n), c, C.c_int64(hw), C.c_float(slope), _stream())
    return x


_PACKED = {}  # (id(weight), slice, terms) -> (weakref, version, packed)


_range_flags = {}  # device index -> int32[1]: set by conv_nhwc (terms=16) when an activation leaves the fp16 range


def _range_flag(device):
    f = _range_flags.get(device.index)
    if f is None:
        f = _range_flags[device.index] = torch.zeros(1, device=device, dtype=torch.int32)
    return f


_range_free = [False]


class range_free:
    """``with hip.range_free():`` -- the fp16 two-term kernels (|activation| < 65504) are replaced by their bf16 three-term
    twins (no range limit, ~1.5x slower) inside the block: convolutions pack / run with terms 6, the DCN takes its
    six-product split.  The re-run path after conv_range_tripped()."""

    def __enter__(self):
        self.saved = _range_free[0]
        _range_free[0] = True

    def __exit__(self, *exc):
        _range_free[0] = self.saved


def is_range_free():
    return _range_free[0]


def conv_range_tripped(reset=True):
    """True if any fp16-split kernel (terms=16 convolution, channels-last DCN) launched since the last call met a value
    outside the fp16 range (|x| > 65000, Inf or NaN).  One 4-byte readback (host sync) per device that ran such a kernel.
    MultiRefRestorationModel.test() / optimize_parameters() call it once per batch and re-run the batch on the range-free
    bf16 three-term split when it fires (archs/nhwc.range_free()).  The same word carries the "packed weights stale" bit of
    verify_packed(): read it with packed_stale()."""
    hit = False
    for f in _range_flags.values():
        v = int(f.item())
        if v & _STALE_BIT:
            _stale_seen[0] = True
        if v & 1:
            hit = True
        if v and reset:
            f.zero_()
    return hit


# ---- parameters edited behind autograd's back (`.data` writes bump no version): a device-side fingerprint of every parameter
# whose packed copy is cached, compared once per forward pass; the verdict travels in the range flag's word (no extra readback)
_STALE_BIT = 2
_stale_seen = [False]
_FP = {'rows': {}, 'order': [], 'dirty': True, 'table': None, 'sums': None, 'done': None, 'ref': None, 'wref': {}}


def _fp_register(weight):
    """remember `weight` (a parameter whose packed copy has just been made) and take its reference fingerprint NOW -- one
    one-row checksum launch on the stream that has just packed it: a write that bumps no version (`.data`, an EMA, a
    hipGraph-replayed optimiser step) between this packing and the next verify_packed() is then a mismatch, not part of the
    reference"""
    import weakref
    wid = id(weight)
    key = (weight.data_ptr(), weight.numel())
    _FP['rows'][wid] = (weakref.ref(weight), key, weight._version)
    dev = weight.device
    tbl = torch.tensor(list(key), dtype=torch.int64).to(dev)
    ref = torch.zeros(1, dtype=torch.int64, device=dev)
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.call('mrefsr_weights_checksum', _p(tbl), 1, _p(ref), _p(done), None, None, 0, _stream())
    _FP.setdefault('wref', {})[wid] = (ref, tbl, done)   # (the table row and counter live as long as the launch may)
    _FP['dirty'] = True


def _fp_table(device):
    live = [(wid, r) for wid, r in _FP['rows'].items() if r[0]() is not None]
    if _FP['dirty'] or _FP['table'] is None or len(live) != len(_FP['order']):
        _FP['rows'] = dict(live)
        _FP['order'] = [wid for wid, _ in live]
        wref = _FP.setdefault('wref', {})
        for wid in [w for w in wref if w not in _FP['rows']]:
            del wref[wid]
        flat = [v for _, r in live for v in r[1]]
        n = len(live)
        _FP['table'] = torch.tensor(flat, dtype=torch.int64).view(-1, 2).to(device)
        _FP['sums'] = torch.zeros(n, dtype=torch.int64, device=device)
        _FP['done'] = torch.zeros(n, dtype=torch.int32, device=device)
        _FP['ref'] = torch.cat([wref[wid][0] for wid in _FP['order']]) if n else None   # the fingerprints taken at packing time
        _FP['dirty'] = False
    return len(_FP['order'])


def verify_packed(device=None):
    """One launch: fingerprint every parameter with a cached packed copy and raise the stale bit where it differs from the
    fingerprint taken at packing time.  No synchronisation; the bit is read with the range flag (conv_range_tripped(), then
    packed_stale()).  Call it once per forward pass, before the convolutions."""
    if not _FP['rows']:
        return
    # a parameter modified in a way autograd sees (its version moved: optimiser step, load_state_dict, an in-place op under no_grad)
    # is re-packed -- and re-fingerprinted -- by the pass that follows: it leaves the check instead of failing it
    moved = [wid for wid, r in _FP['rows'].items() if r[0]() is not None and (r[0]()._version != r[2] or r[0]().data_ptr() != r[1][0])]
    if moved:
        for wid in moved:
            del _FP['rows'][wid]
        _FP['dirty'] = True
        if not _FP['rows']:
            return
    device = device or torch.device('cuda', torch.cuda.current_device())
    n = _fp_table(device)
    if n == 0:
        return
    _lib.call('mrefsr_weights_checksum', _p(_FP['table']), n, _p(_FP['sums']), _p(_FP['done']), _p(_FP['ref']),
              _p(_range_flag(device)), _STALE_BIT, _stream())


def packed_stale(reset=True):
    """True if verify_packed() found a parameter that no longer matches its packed copy (seen at the last conv_range_tripped()
    readback); the caller drops the cache (invalidate_packed()) and repeats the pass."""
    seen = _stale_seen[0]
    if reset:
        _stale_seen[0] = False
    return seen


def check_conv_range(reset=True):
    """Raise if conv_range_tripped(): for callers that drive the kernels directly and want the event as an error."""
    if conv_range_tripped(reset):
            raise FloatingPointError('mrefsr_conv_nhwc_f32 (terms=16) / mrefsr_dcn_fwd_f32: an activation exceeded the fp16 range '
                                     '(|x| > 65000 or NaN); rerun with MREFSR_CONV_TERMS=6 MREFSR_DCN_TERMS=6 (bf16 three-term '
                                     'splits, no range limit)')


class PackedWeight:
    """packed split fragments + the power-of-two scale they carry (terms == 16; 1.0 otherwise)"""
    __slots__ = ('data', 'wscale', 'terms')

    def __init__(self, data, wscale, terms):
        self.data, self.wscale, self.terms = data, wscale, terms


def packed_weight(weight, cin_slice=None, terms=6):
    """cached conv_pack_weight(weight[:, a:b]): re-packed when the parameter is modified in place through autograd-visible
    ops (optimizer step, load_state_dict, ``with torch.no_grad(): p.copy_(..)``: they bump ``_version``), re-allocated
    (``data_ptr``) or replaced.  Writes through ``p.data`` bump nothing; they are caught on the device instead: every cached
    parameter is fingerprinted when packed and again by ``verify_packed()`` (one launch per forward pass of the model, verdict
    read with the range flag), and a mismatch makes the model drop the cache and repeat the pass."""
    import weakref
    key = (id(weight), cin_slice, terms)
    hit = _PACKED.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr()):
        return hit[2]
    if len(_PACKED) > 4096:   # entries of parameters that no longer exist
        for k in [k for k, v in _PACKED.items() if v[0]() is None]:
            del _PACKED[k]
    w = weight.detach()
    if cin_slice is not None:
        w = w[:, cin_slice[0]:cin_slice[1]]
    packed = conv_pack_weight(w.contiguous(), terms)
    _PACKED[key] = (weakref.ref(weight), (weight._version, weight.data_ptr()), packed)
    _fp_register(weight)
    return packed


_packed_epoch = [0]


def invalidate_packed():
    """drop every cached packed weight (and let hipGraph captures notice): call after editing parameters through ``.data``"""
    _PACKED.clear()
    _packed_epoch[0] += 1
    _FP.update(rows={}, order=[], dirty=True, table=None, ref=None, wref={})


def packed_epoch():
    return _packed_epoch[0]


def conv_pack_weight(weight, terms=6):
    """weight [Cout,Cin,k,k] fp32 (k = 1 or 3) -> PackedWeight (split fragments as a uint8 tensor) for conv_nhwc.
    terms=16 (fp16 two-term split) scales the weights by 2^s with max|w| * 2^s in [2^13, 2^14): one host
    sync per packing (the result is cached per parameter version by packed_weight)."""
    import math
    _chk('conv_pack_weight', weight)
    co, ci, kh, kw = weight.shape
    if kh != kw or kh not in (1, 3):
        raise ValueError('conv_pack_weight: 1x1 or 3x3 kernels only')
    wscale = 1.0
    if terms == 17 and kh != 3:
        raise ValueError('conv_pack_weight: terms=17 (Winograd F(2x2, 3x3)) is for 3x3 kernels')
    if terms in (16, 17):   # 17: the Winograd form of 16 -- |G g G^T| <= 2.25 max|w| stays inside fp16 under the same scale
        amax = float(weight.abs().max().item())
        if not math.isfinite(amax):
            raise ValueError('conv_pack_weight: non-finite weights')
        wscale = 2.0 ** (13 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    nbytes = _lib.load().mrefsr_conv_packed_bytes(co, ci, kh, terms)
    packed = torch.empty(nbytes, device=weight.device, dtype=torch.uint8)
    _lib.call('mrefsr_conv_pack_weight_f32', _p(weight), _p(packed), co, ci, kh, terms, C.c_float(wscale), _stream())
    return PackedWeight(packed, wscale, terms)


def conv_pack_view(weight, cin_slice=None, terms=6, dgrad=False, wscale=1.0):
    """Packed fragments of weight[:, a:b] (``cin_slice`` = (a, b), default all input channels) read in place -- no slicing
    copy -- or, with ``dgrad``, of the operator of the convolution's input gradient w.r.t. those channels (output channels
    b - a, input channels Cout, taps point-mirrored): ``conv_nhwc(g_out, that, None, b - a, k)`` is d loss / d x[..., a:b].
    No host synchronisation: terms 16 (fp16 two-term split) takes the power-of-two ``wscale`` from the caller (max|w| * wscale
    must stay below 65504; conv_pack_weight derives it from the weight's amax with a readback)."""
    _chk('conv_pack_view', weight)
    co, ci, kh, kw = weight.shape
    if kh != kw or kh not in (1, 3):
        raise ValueError('conv_pack_view: 1x1 or 3x3 kernels only')
    if terms not in (6, 1, 16):
        raise ValueError('conv_pack_view: terms 6, 1 or 16')
    pw, j = conv_pack_plan(weight, cin_slice, terms, dgrad, wscale)
    conv_pack_one(j)
    return pw


def conv_pack_one(j):
    """run one mrefsr_conv_pack_job by itself"""
    _lib.call('mrefsr_conv_pack_weight_view_f32', C.c_void_p(j.weight), C.c_void_p(j.packed), j.Cout, j.Cin, j.ksize, j.terms, C.c_float(j.wscale),
              C.c_int64(j.stride_o), C.c_int64(j.stride_i), j.flip, _stream())


def conv_pack_plan(weight, cin_slice=None, terms=6, dgrad=False, wscale=1.0):
    """(PackedWeight with an unfilled buffer, the mrefsr_conv_pack_job that fills it): conv_pack_view's arguments as a table entry
    for conv_pack_multi.  The job holds raw addresses: the caller keeps ``weight`` and the PackedWeight alive."""
    co, ci, kh, kw = weight.shape
    if not weight.is_contiguous():
        raise ValueError('conv_pack_plan: contiguous OIHW weight expected')
    if terms != 16:
        wscale = 1.0
    a, b = cin_slice if cin_slice is not None else (0, ci)
    taps = kh * kw
    po, pi, so, si = (b - a, co, taps, ci * taps) if dgrad else (co, b - a, ci * taps, taps)
    nbytes = _lib.load().mrefsr_conv_packed_bytes(po, pi, kh, terms)
    packed = torch.empty(nbytes, device=weight.device, dtype=torch.uint8)
    # dgrad = 'T': the transposed operator WITHOUT the point mirror (d columns = W^T . g_out of a deformable convolution: the taps
    # keep their places, mrefsr_dcn_bwd_data_f32)
    job = _lib.ConvPackJob(weight.data_ptr() + 4 * a * taps, packed.data_ptr(), so, si, po, pi, kh, terms, 1 if dgrad is True else 0, wscale)
    return PackedWeight(packed, wscale, terms), job


def conv_pack_table(jobs, device):
    """device copy of a list of mrefsr_conv_pack_job (a synchronous upload: build it when the set of weights changes, not per step)"""
    arr = (_lib.ConvPackJob * len(jobs))(*jobs)
    return torch.frombuffer(bytearray(arr), dtype=torch.uint8).to(device)


def conv_pack_multi(table, n_jobs):
    """run the n_jobs packings of a conv_pack_table in one launch; a weight that no longer fits the fp16 range under the scale
    of its job raises the range flag (conv_range_tripped())"""
    _lib.call('mrefsr_conv_pack_weights_multi_f32', _p(table), n_jobs, _p(_range_flag(table.device)), _stream())


def act_bwd_nhwc(g_out, out, act, slope=0.0, slope_ptr=None, want_bias=True, want_amax=False):
    """Backward of a fused convolution epilogue on [..., C] contiguous tensors: g_pre = g_out * act'(out) with act 0 none,
    1 LeakyReLU(slope) (0 = ReLU), 2 PReLU(slope_ptr).  Returns (g_pre [..., ld] with ld = C rounded up to 4 (extra channels
    zero: the dgrad convolution reads 16-byte channel vectors), bias gradient [C] | None, PReLU weight gradient [1] | None)."""
    _chk('act_bwd_nhwc', g_out, out, slope_ptr)
    c = g_out.shape[-1]
    npix = g_out.numel() // c
    blocks = _lib.load().mrefsr_act_bwd_blocks(npix, c)
    if blocks <= 0:
        raise ValueError(f'act_bwd_nhwc: unsupported channel count {c}')
    ld = (c + 3) // 4 * 4
    if act == 0 and ld == c:
        g_pre = None
        if not want_bias and not want_amax:
            return g_out, None, None
    elif ld == c:
        g_pre = torch.empty_like(g_out)
    else:
        g_pre = torch.zeros(g_out.shape[:-1] + (ld,), device=g_out.device, dtype=torch.float32)
    # the three zero-initialised accumulators of the kernel in ONE allocation (one fill launch instead of up to three)
    z = zeros_f32(g_out.device, c + 2) if (want_bias or act == 2 or want_amax) else None
    g_bias = z[:c] if want_bias else None
    g_slope = z[c:c + 1] if act == 2 else None
    amax = z[c + 1:c + 2] if want_amax else None
    _lib.call('mrefsr_act_bwd_nhwc_f32', _p(g_out), _p(out if act else None), _p(g_pre), ld, _p(g_bias), _p(g_slope), _p(amax), C.c_int64(npix), c,
              act, C.c_float(slope), _p(slope_ptr), _p(_range_flag(g_out.device)) if act == 2 else None, _stream())
    if want_amax:
        return (g_out if g_pre is None else g_pre), g_bias, g_slope, amax
    return (g_out if g_pre is None else g_pre), g_bias, g_slope


_wgrad_ws = {}


def _wgrad_workspace(device, nbytes):
    """one growing scratch buffer per device and stream (launches on a stream are ordered: the partials of one weight gradient
    are consumed by its reduction before the next one writes)"""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _wgrad_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _wgrad_ws[key] = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
    return buf


def conv_wgrad3x3(x, g, cin, cout, g_amax):
    """d loss / d weight [cout,cin,3,3] of a 3x3 'same' convolution from channels-last tensors x [N,H,W,>=cin], g [N,H,W,>=cout]
    (pixel-contiguous; extra trailing channels ignored); g_amax [1] = max |g| from act_bwd_nhwc(want_amax=True)"""
    n, h, w, _ = x.shape
    if tuple(g.shape[:3]) != (n, h, w):
        raise ValueError('conv_wgrad3x3: x / g sizes differ')
    ldx, ldg = _nhwc_ld('x', x), _nhwc_ld('g', g)
    _chk('conv_wgrad3x3', g_amax)
    dw = torch.empty((cout, cin, 3, 3), device=x.device, dtype=torch.float32)
    need = _lib.load().mrefsr_conv_wgrad3x3_workspace_bytes(n, h, w, cin, cout)
    ws = _wgrad_workspace(x.device, need)
    _lib.call('mrefsr_conv_wgrad3x3_f32', _p(x), ldx, cin, _p(g), ldg, cout, _p(dw), C.c_int64(cin * 9), C.c_int64(9), 0, _p(g_amax), n, h, w,
              _p(ws), C.c_int64(need), _p(_range_flag(x.device)), _stream())
    return dw


def conv_wgrad3x3_batch(xs, gs, cin, cout, g_amaxes):
    """conv_wgrad3x3 for len(xs) <= 32 convolutions of one geometry in ONE launch pair (the convolutions of a residual trunk:
    their weight gradients feed nothing inside backward, so they can wait for each other); -> [n, cout, cin, 3, 3], row j = job j"""
    nj = len(xs)
    if not (0 < nj <= 32 and len(gs) == nj and len(g_amaxes) == nj):
        raise ValueError('conv_wgrad3x3_batch: 1..32 jobs, one g / g_amax per x')
    n, h, w, _ = xs[0].shape
    ldx, ldg = _nhwc_ld('x', xs[0]), _nhwc_ld('g', gs[0])
    for x, g in zip(xs, gs):
        if tuple(x.shape[:3]) != (n, h, w) or tuple(g.shape[:3]) != (n, h, w) or _nhwc_ld('x', x) != ldx or _nhwc_ld('g', g) != ldg:
            raise ValueError('conv_wgrad3x3_batch: the jobs of a batch share one geometry')
    _chk('conv_wgrad3x3_batch', *g_amaxes)
    dw = torch.empty((nj, cout, cin, 3, 3), device=xs[0].device, dtype=torch.float32)
    need = _lib.load().mrefsr_conv_wgrad3x3_batch_workspace_bytes(nj, n, h, w, cin, cout)
    ws = _wgrad_workspace(xs[0].device, need)
    arr = C.c_void_p * nj
    _lib.call('mrefsr_conv_wgrad3x3_batch_f32', nj, arr(*[t.data_ptr() for t in xs]), ldx, cin, arr(*[t.data_ptr() for t in gs]), ldg, cout,
              arr(*[dw[j].data_ptr() for j in range(nj)]), C.c_int64(cin * 9), C.c_int64(9), 0, arr(*[t.data_ptr() for t in g_amaxes]), n, h, w,
              _p(ws), C.c_int64(need), _p(_range_flag(xs[0].device)), _stream())
    return dw


def mrattn_bwd_nhwc(q, emb, ass, g_out, t):
    """gradient of mrattn_fwd_nhwc: -> (g_q, g_emb, g_ass), same layouts"""
    _chk('mrattn_bwd_nhwc', q, emb, ass, g_out)
    n, h, w, c = q.shape
    if tuple(emb.shape) != (n * t, h, w, c) or tuple(ass.shape) != (n * t, h, w, 2 * c) or tuple(g_out.shape) != (n, h, w, 2 * c):
        raise ValueError('mrattn_bwd_nhwc: inconsistent shapes')
    g_q, g_emb, g_ass = torch.empty_like(q), torch.empty_like(emb), torch.empty_like(ass)
    _lib.call('mrefsr_mrattn_bwd_nhwc_f32', _p(q), _p(emb), _p(ass), _p(g_out), _p(g_q), _p(g_emb), _p(g_ass), n, t, c, h * w, _stream())
    return g_q, g_emb, g_ass


def attn_modulate_bwd(g, refs, mul):
    """gradient of refs * sigmoid(mul) * 2 + add w.r.t. (refs, mul); d/d add = g"""
    _chk('attn_modulate_bwd', g, refs, mul)
    g_refs, g_mul = torch.empty_like(refs), torch.empty_like(mul)
    _lib.call('mrefsr_attn_modulate_bwd_f32', _p(g), _p(refs), _p(mul), _p(g_refs), _p(g_mul), C.c_int64(mul.numel()), _stream())
    return g_refs, g_mul


# ---- max |out| words of the forward launches (mrefsr_conv_nhwc_amax_f32 / mrefsr_dcn_fwd_amax_f32): one zeroed float per producing
# launch, handed to the consumer as its in_amax.  A pool per device, two halves: a half is zeroed (one memset) when the slot counter
# enters it -- its words were handed out >= AMAX_POOL / 2 launches ago, their tensors have long been consumed.  amax_pool_reset()
# (start of a pass: MultiRefRestorationModel.test / optimize_parameters) zeroes everything and restarts at slot 0, so that a
# captured graph re-zeroes and re-uses the same words in every replay.
AMAX_POOL = 8192
_amax_pool = {}


def _amax_state(device):
    st = _amax_pool.get(device)
    if st is None:
        buf = torch.zeros(AMAX_POOL, device=device, dtype=torch.float32)
        st = _amax_pool[device] = [buf, 0, list(buf.split(1))]   # (the one-element views are made once: a slice per launch costs microseconds)
    return st


def amax_pool_reset():
    for st in _amax_pool.values():
        st[0].zero_()
        st[1] = 0


def amax_slot(device):
    """a zeroed 1-element float32 device tensor for one launch's max |out|"""
    st = _amax_state(device)
    i = st[1]
    if i == AMAX_POOL:
        i = 0
    if i == 0 or i == AMAX_POOL // 2:
        st[0][i:i + AMAX_POOL // 2].zero_()
    st[1] = i + 1
    return st[2][i]


def _nhwc_ld(name, t):
    """channel stride of a pixel for an [N,H,W,C] tensor that may be a channel slice of a wider one"""
    n, h, w, c = t.shape
    ld = t.stride(2)
    if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.stride(3) == 1 and t.stride(1) == w * ld and t.stride(0) == h * w * ld):
        raise ValueError(f'conv_nhwc: {name} must be a pixel-contiguous float32 / bfloat16 NHWC device tensor, got shape '
                         f'{tuple(t.shape)} strides {t.stride()} dtype {t.dtype}')
    return ld


def conv_nhwc(x1, packed, bias, cout, ksize, x2=None, pre=None, residual=None, act=False, slope=0.0, slope_ptr=None,
              epilogue=0, out=None, terms=None, in_amax=None, out_amax=None):
    """Convolution (k = 1 / 3, stride 1, same padding) of cat([x1, x2], channel) with fused epilogue; all NHWC.

    x1 [N1,H,W,C1], x2 [N2,H,W,C2] (batch-broadcast: image n reads x[n % N]); pre [Np,H,W,cout] added
    before the activation (broadcast n % Np); act -> LeakyReLU(slope | *slope_ptr); residual [N,H,W,cout]
    added after it; epilogue 0 plain / 1 MaxPool2d(2,2) / 2 PixelShuffle(2).  `out` may be a channel
    slice of a wider NHWC buffer.  Returns out."""
    n1, h, w, c1 = x1.shape
    if terms is not None and terms != packed.terms:
        raise ValueError(f'conv_nhwc: weights packed for terms={packed.terms}, asked for terms={terms}')
    terms = packed.terms
    io16 = x1.dtype == torch.bfloat16
    if io16:
        if terms != 1:
            raise TypeError('conv_nhwc: bfloat16 tensors go with the bf16 arithmetic (weights packed with terms=1)')
        for t_ in (x2, pre, residual, out):
            if t_ is not None and t_.dtype != torch.bfloat16:
                raise TypeError('conv_nhwc: with a bfloat16 x1 every activation tensor must be bfloat16')
        terms = 2   # descriptor code of "terms = 1 arithmetic on bf16 tensors"
    d = _lib.ConvDesc()
    d.wscale = packed.wscale
    d.H, d.W, d.ksize, d.C1, d.ld1, d.N1 = h, w, ksize, c1, _nhwc_ld('x1', x1), n1
    n = n1
    if x2 is not None:
        d.C2, d.ld2, d.N2 = x2.shape[3], _nhwc_ld('x2', x2), x2.shape[0]
        if tuple(x2.shape[1:3]) != (h, w):
            raise ValueError('conv_nhwc: x1 / x2 spatial size mismatch')
        n = max(n, x2.shape[0])
    if residual is not None:
        n = max(n, residual.shape[0])
        d.ld_res = _nhwc_ld('residual', residual)
    if pre is not None:
        _chk('conv_nhwc', pre, dtype=x1.dtype)
        d.pre_N = pre.shape[0]
    d.N, d.Cout, d.act, d.epilogue, d.terms, d.slope = n, cout, 1 if act else 0, epilogue, terms, slope
    oshape = {0: (n, h, w, cout), 1: (n, h // 2, w // 2, cout), 2: (n, 2 * h, 2 * w, cout // 4)}[epilogue]
    if out is None:
        out = torch.empty(oshape, device=x1.device, dtype=x1.dtype)
    elif tuple(out.shape) != oshape:
        raise ValueError(f'conv_nhwc: out shape {tuple(out.shape)} != {oshape}')
    d.ld_out = _nhwc_ld('out', out)
    _chk('conv_nhwc', bias, slope_ptr)
    es = x1.element_size()   # algorithmic bytes of the launch: every operand and the result once (broadcast operands once)
    nby = (x1.numel() + (x2.numel() if x2 is not None else 0) + (pre.numel() if pre is not None else 0) +
           (residual.numel() if residual is not None else 0) + out.shape[0] * out.shape[1] * out.shape[2] * oshape[3]) * es + \
        4.0 * (d.C1 + d.C2) * cout * ksize * ksize
    with _timed('conv_wino_k3' if terms == 17 else f'conv_nhwc_k{ksize}', 2.0 * n * h * w * (d.C1 + d.C2) * cout * ksize * ksize, detail=True, nbytes=nby):
        if out_amax is not None:   # + max |out| into out_amax[0] (a zeroed device word: amax_slot): the next Winograd layer's input scale
            _chk('conv_nhwc', in_amax, out_amax)
            _lib.call('mrefsr_conv_nhwc_amax_f32', C.byref(d), _p(x1), _p(x2), _p(packed.data), _p(bias), _p(slope_ptr), _p(pre),
                      _p(residual), _p(out), _p(_range_flag(x1.device)), _p(in_amax), _p(out_amax), _stream())
        elif in_amax is not None:   # inputs of unknown magnitude (gradients): scaled into the fp16 range by the kernel, terms 16 only
            _lib.call('mrefsr_conv_nhwc_scaled_f32', C.byref(d), _p(x1), _p(x2), _p(packed.data), _p(bias), _p(slope_ptr), _p(pre),
                      _p(residual), _p(out), _p(_range_flag(x1.device)), _p(in_amax), _stream())
        else:
            _lib.call('mrefsr_conv_nhwc_f32', C.byref(d), _p(x1), _p(x2), _p(packed.data), _p(bias), _p(slope_ptr), _p(pre), _p(residual),
                      _p(out), _p(_range_flag(x1.device) if terms in (16, 17) else None), _stream())
    return out


def conv_nhwc_bwd(x, packed, cout, ksize, residual=None, residual_is_mask=False, in_amax=None, want_stats=True):
    """Input-gradient convolution of a training step with the pass that would follow it folded in (mrefsr_conv_nhwc_bwd_f32):
    out = conv(x) + residual, or conv(x) where residual > 0 (``residual_is_mask``: the ReLU of the layer below); with
    ``want_stats`` also the per-channel sums of out (a bias gradient) and max |out| (the fp16 input scale of whatever reads out
    next).  terms-16 packed weights, fp32 channels-last tensors.  -> (out, sums [cout] | None, amax [1] | None)"""
    if packed.terms != 16:
        raise ValueError('conv_nhwc_bwd: weights packed with terms=16 expected')
    n, h, w, c1 = x.shape
    d = _lib.ConvDesc()
    d.wscale = packed.wscale
    d.N, d.H, d.W, d.ksize, d.C1, d.ld1, d.N1 = n, h, w, ksize, c1, _nhwc_ld('x', x), n
    d.Cout, d.act, d.epilogue, d.terms, d.slope = cout, 0, 0, 16, 0.0
    if residual is not None:
        if tuple(residual.shape) != (n, h, w, cout):
            raise ValueError('conv_nhwc_bwd: residual / mask source must have the output shape')
        d.ld_res = _nhwc_ld('residual', residual)
    out = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    d.ld_out = cout
    _chk('conv_nhwc_bwd', x, residual, in_amax)
    z = zeros_f32(x.device, cout + 1) if want_stats else None
    _lib.call('mrefsr_conv_nhwc_bwd_f32', C.byref(d), _p(x), _p(packed.data), _p(residual), 1 if residual_is_mask else 0, _p(out),
              _p(_range_flag(x.device)), _p(in_amax), _p(z[:cout]) if want_stats else None, _p(z[cout:]) if want_stats else None, _stream())
    return out, (z[:cout] if want_stats else None), (z[cout:] if want_stats else None)


def conv_dynagg(x, packed, bias, pre, dg, abs_sum=None):
    """conv_offset_mask (3x3, C -> 27 dg) of a DynAgg with the glue of ref :56-73 as its epilogue: x [N,H,W,C] channels-last
    -> planar (offset [N,18dg,H,W], mask [N,9dg,H,W]) for dcn_fwd; pre [N,9,H,W,2] ([x,y]); abs_sum float64[1] or None"""
    n, h, w, c = x.shape
    _chk('conv_dynagg', pre, bias)
    if tuple(pre.shape) != (n, 9, h, w, 2):
        raise ValueError(f'conv_dynagg: pre {tuple(pre.shape)} does not match x {tuple(x.shape)}')
    if abs_sum is not None:
        _chk('conv_dynagg', abs_sum, dtype=torch.float64)
    d = _lib.ConvDesc()
    d.wscale, d.terms = packed.wscale, (2 if (x.dtype == torch.bfloat16 and packed.terms == 1) else packed.terms)
    d.N, d.H, d.W, d.ksize, d.C1, d.ld1, d.N1, d.Cout = n, h, w, 3, c, _nhwc_ld('x', x), n, 27 * dg
    offset = torch.empty((n, 18 * dg, h, w), device=x.device, dtype=torch.float32)
    mask = torch.empty((n, 9 * dg, h, w), device=x.device, dtype=torch.float32)
    nby = x.numel() * x.element_size() + (pre.numel() + offset.numel() + mask.numel()) * 4.0 + 4.0 * c * 27 * dg * 9
    with _timed('conv_nhwc_k3', 2.0 * n * h * w * c * 27 * dg * 9, detail=True, nbytes=nby):
        _lib.call('mrefsr_conv_dynagg_f32', C.byref(d), _p(x), _p(packed.data), _p(bias), _p(pre), _p(offset), _p(mask), _p(abs_sum), dg,
                  _p(_range_flag(x.device) if packed.terms == 16 else None), _stream())
    return offset, mask


def image_to_nhwc4(img, mean=None, std=None, range_norm=False):
    """[N,3,H,W] float32 (contiguous) -> [N,H,W,4]: ((img + 1) / 2 if range_norm) then ((. - mean) / std if mean is given; 3-element
    device tensors) in channels 0..2, zero in channel 3 -- the extractors' input normalisation and channels-last packing in one pass"""
    _chk('image_to_nhwc4', img, mean, std)
    n, c, h, w = img.shape
    if c != 3:
        raise ValueError(f'image_to_nhwc4: 3 channels expected, got {c}')
    if mean is not None and (mean.numel() != 3 or std is None or std.numel() != 3):
        raise ValueError('image_to_nhwc4: mean / std must hold 3 values each')
    out = torch.empty((n, h, w, 4), device=img.device, dtype=torch.float32)
    _lib.call('mrefsr_image_to_nhwc4_f32', _p(img), _p(out), C.c_int64(n), C.c_int64(h * w), 1 if range_norm else 0, _p(mean), _p(std), _stream())
    return out


def attn_modulate_(refs, mul, add):
    """mul <- refs * sigmoid(mul) * 2 + add, in place on ``mul`` (all three contiguous, same shape)"""
    b16 = mul.dtype == torch.bfloat16
    _chk('attn_modulate', refs, mul, add, dtype=mul.dtype if b16 else torch.float32)
    if refs.shape != mul.shape or add.shape != mul.shape:
        raise ValueError('attn_modulate: shape mismatch')
    _lib.call('mrefsr_attn_modulate_bf16' if b16 else 'mrefsr_attn_modulate_f32', _p(refs), _p(mul), _p(add), C.c_int64(mul.numel()), _stream())
    return mul


def bias_relu_pool2(x, bias):
    """x [N,C,H,W] (conv output without bias) -> relu(maxpool2x2(x) + bias) [N,C,H/2,W/2]"""
    _chk('bias_relu_pool2', x, bias)
    n, c, h, w = x.shape
    out = torch.empty((n, c, h // 2, w // 2), device=x.device, dtype=torch.float32)
    _lib.call('mrefsr_bias_relu_pool2_f32', _p(x), _p(bias), _p(out), C.c_int64(n), c, h, w, _stream())
    return out


def upfirdn2d(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """x [major,in_h,in_w,minor] -> [major,out_h,out_w,minor]  (upfirdn2d.cpp:13-24)."""
    if x.dtype not in _DT:
        raise TypeError(f'upfirdn2d: unsupported dtype {x.dtype}')
    x, kernel = x.contiguous(), kernel.to(x.dtype).contiguous()
    _chk('upfirdn2d', x, kernel, dtype=x.dtype)
    mj, ih, iw, mn = x.shape
    kh, kw = kernel.shape
    oh = (ih * up_y + pad_y0 + pad_y1 - kh + down_y) // down_y
    ow = (iw * up_x + pad_x0 + pad_x1 - kw + down_x) // down_x
    out = torch.empty((mj, oh, ow, mn), device=x.device, dtype=x.dtype)
    _lib.call('mrefsr_upfirdn2d', _p(x), _p(kernel), _p(out), mj, ih, iw, mn, kh, kw, up_x, up_y, down_x, down_y,
              pad_x0, pad_x1, pad_y0, pad_y1, _DT[x.dtype], _stream())
    return out
