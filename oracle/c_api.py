"""ctypes binding of oracle/mrefsr_oracle.c.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every wrapper takes/returns C-contiguous numpy arrays.  The shared object is built on demand with
gcc (``make -C oracle``); it never touches /root/reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libmrefsr_oracle.so')
_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
_i64p = np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')


def build(force=False):
    src = os.path.join(_HERE, 'mrefsr_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = C.CDLL(_SO)
        except OSError:
            build(force=True)
            _lib = C.CDLL(_SO)
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def pixnorm(x):
    """x [C,h,w] -> (y [C,h,w], n2 [h,w]).  corres_generation_arch.py:57-59."""
    x = _f32(x)
    c, h, w = x.shape
    y = np.empty_like(x)
    n2 = np.empty((h, w), np.float32)
    lib().orc_pixnorm(_ptr(x), c, h * w, _ptr(y), _ptr(n2))
    return y, n2


def patch_norm(n2):
    """n2 [h,w] -> (norm+1e-5 [h-2,w-2], 1/(norm+1e-5))."""
    n2 = _f32(n2)
    h, w = n2.shape
    ne = np.empty((h - 2, w - 2), np.float32)
    inv = np.empty((h - 2, w - 2), np.float32)
    lib().orc_patch_norm(_ptr(n2), h, w, _ptr(ne), _ptr(inv))
    return ne, inv


def sumsq(y):
    y = _f32(y)
    c, h, w = y.shape
    n2 = np.empty((h, w), np.float32)
    lib().orc_sumsq(_ptr(y), c, h * w, _ptr(n2))
    return n2


def corr_top1_normalised(fin, fref):
    """fin, fref [C,h,w] already pixel-normalised -> (idx int64 [h-2,w-2], val fp32).
    ref_map_util.py:26-86 with patch 3, stride 1, is_norm=True, norm_input=True."""
    fin, fref = _f32(fin), _f32(fref)
    c, h, w = fin.shape
    ne_in, _ = patch_norm(sumsq(fin))
    _, inv_ref = patch_norm(sumsq(fref))
    idx = np.empty((h - 2, w - 2), np.int64)
    val = np.empty((h - 2, w - 2), np.float32)
    lib().orc_corr_top1(_ptr(fin), _ptr(fref), c, h, w, _ptr(inv_ref), _ptr(ne_in), _ptr(idx), _ptr(val))
    return idx, val


def feature_match_index(feat_in, feat_ref):
    """The path of corres_generation_arch.py:55-68: normalise both maps per pixel, then
    ref_map_util.py:26-86 with patch 3, stride 1, is_norm=True, norm_input=True.
    feat_* raw [C,h,w] -> (idx int64 [h-2,w-2], val fp32 [h-2,w-2])."""
    feat_in, feat_ref = _f32(feat_in), _f32(feat_ref)
    c, h, w = feat_in.shape
    yin, n2_in = pixnorm(feat_in)
    yref, n2_ref = pixnorm(feat_ref)
    ne_in, _ = patch_norm(n2_in)
    _, inv_ref = patch_norm(n2_ref)
    idx = np.empty((h - 2, w - 2), np.int64)
    val = np.empty((h - 2, w - 2), np.float32)
    lib().orc_corr_top1(_ptr(yin), _ptr(yref), c, h, w, _ptr(inv_ref), _ptr(ne_in), _ptr(idx), _ptr(val))
    return idx, val


def feature_match_index_rows(feat_in, feat_ref, rows):
    """feature_match_index for the patch rows ``rows`` only (same arithmetic: orc_corr_top1_rows shares its two helpers with
    orc_corr_top1) -> (idx int64 [len(rows), w-2], val fp32): tests at sizes where a whole map takes a minute per pair"""
    feat_in, feat_ref = _f32(feat_in), _f32(feat_ref)
    c, h, w = feat_in.shape
    yin, n2_in = pixnorm(feat_in)
    yref, n2_ref = pixnorm(feat_ref)
    ne_in, _ = patch_norm(n2_in)
    _, inv_ref = patch_norm(n2_ref)
    rows = np.ascontiguousarray(rows, np.int32)
    assert rows.ndim == 1 and rows.min() >= 0 and rows.max() < h - 2
    idx = np.empty((len(rows), w - 2), np.int64)
    val = np.empty((len(rows), w - 2), np.float32)
    lib().orc_corr_top1_rows(_ptr(yin), _ptr(yref), c, h, w, _ptr(inv_ref), _ptr(ne_in), rows.ctypes.data_as(C.c_void_p), len(rows),
                             _ptr(idx), _ptr(val))
    return idx, val


def feature_match_index_generic(feat_in, feat_ref, patch_size=3, input_stride=1, ref_stride=1, is_norm=True, norm_input=False):
    """ref_map_util.feature_match_index with any patch size / strides / map sizes (maps used as given)"""
    fin, fref = _f32(feat_in), _f32(feat_ref)
    c, h, w = fin.shape
    _, hr, wr = fref.shape
    nqy, nqx = (h - patch_size) // input_stride + 1, (w - patch_size) // input_stride + 1
    idx = np.empty((nqy, nqx), dtype=np.int64)
    val = np.empty((nqy, nqx), dtype=np.float32)
    lib().orc_feature_match_index(_ptr(fin), _ptr(fref), c, h, w, hr, wr, patch_size, input_stride, ref_stride, int(bool(is_norm)),
                                  int(bool(norm_input)), _ptr(idx), _ptr(val))
    return idx, val


def corr_pair_f64(fin_n, fref_n, q, r):
    fin_n, fref_n = _f32(fin_n), _f32(fref_n)
    c, h, w = fin_n.shape
    f = lib().orc_corr_pair_f64
    f.restype = C.c_double
    return float(f(_ptr(fin_n), _ptr(fref_n), c, h, w, int(q), int(r)))


def offsets_from_idx(idx, h, w):
    """idx int64 [h-2,w-2] -> three arrays [9,s*h,s*w,2] for s = 1, 2, 4 (last dim [x,y])."""
    idx = np.ascontiguousarray(idx, np.int64)
    outs = [np.empty((9, h * s, w * s, 2), np.float32) for s in (1, 2, 4)]
    lib().orc_offsets_from_idx(_ptr(idx), h, w, _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]))
    return outs


def dcnv2_fwd(x, offset, mask, weight, bias, stride=1, pad=1, dil=1, groups=1, dg=1):
    x, offset, weight = _f32(x), _f32(offset), _f32(weight)
    mask = None if mask is None else _f32(mask)
    bias = None if bias is None else _f32(bias)
    b, c, h, w = x.shape
    co, _, kh, kw = weight.shape
    ho = (h + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    out = np.empty((b, co, ho, wo), np.float32)
    lib().orc_dcnv2_fwd(_ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(bias), _ptr(out),
                        b, c, h, w, co, kh, kw, stride, pad, dil, groups, dg)
    return out


def dcnv2_im2col(x, offset, mask, kh=3, kw=3, stride=1, pad=1, dil=1, dg=1):
    """-> columns [B, C*kh*kw, Ho*Wo]"""
    x, offset = _f32(x), _f32(offset)
    mask = None if mask is None else _f32(mask)
    b, c, h, w = x.shape
    ho = (h + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    col = np.empty((b, c * kh * kw, ho * wo), np.float32)
    lib().orc_dcnv2_im2col(_ptr(x), _ptr(offset), _ptr(mask), _ptr(col), b, c, h, w, kh, kw, stride, pad, dil, dg)
    return col


def dcnv2_bwd(x, offset, mask, weight, gout, stride=1, pad=1, dil=1, groups=1, dg=1, with_bias=True):
    x, offset, weight, gout = _f32(x), _f32(offset), _f32(weight), _f32(gout)
    mask = None if mask is None else _f32(mask)
    b, c, h, w = x.shape
    co, _, kh, kw = weight.shape
    gx = np.zeros_like(x)
    goff = np.zeros_like(offset)
    gmask = None if mask is None else np.zeros_like(mask)
    gw = np.zeros_like(weight)
    gb = np.zeros((co,), np.float32) if with_bias else None
    lib().orc_dcnv2_bwd(_ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(gout), _ptr(gx),
                        _ptr(goff), _ptr(gmask), _ptr(gw), _ptr(gb), b, c, h, w, co, kh, kw,
                        stride, pad, dil, groups, dg)
    return gx, goff, gmask, gw, gb


def mrattn_fwd(q, emb, ass):
    """q [N,c,H,W], emb [N,T,c,H,W], ass [N,T,c2,H,W] -> (out [N,c2,H,W], prob [N,T,H,W])."""
    q, emb, ass = _f32(q), _f32(emb), _f32(ass)
    n, c, h, w = q.shape
    t, c2 = ass.shape[1], ass.shape[2]
    out = np.empty((n, c2, h, w), np.float32)
    prob = np.empty((n, t, h, w), np.float32)
    lib().orc_mrattn_fwd(_ptr(q), _ptr(emb), _ptr(ass), _ptr(out), _ptr(prob), n, t, c, c2, h * w)
    return out, prob


def mrattn_bwd(q, emb, ass, gout):
    q, emb, ass, gout = _f32(q), _f32(emb), _f32(ass), _f32(gout)
    n, c, h, w = q.shape
    t, c2 = ass.shape[1], ass.shape[2]
    gq, gemb, gass = np.empty_like(q), np.empty_like(emb), np.empty_like(ass)
    lib().orc_mrattn_bwd(_ptr(q), _ptr(emb), _ptr(ass), _ptr(gout), _ptr(gq), _ptr(gemb), _ptr(gass),
                         n, t, c, c2, h * w)
    return gq, gemb, gass


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    x = _f32(x)
    bias = None if bias is None or bias.size == 0 else _f32(bias)
    ref = None if ref is None or ref.size == 0 else _f32(ref)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    out = np.empty_like(x)
    lib().orc_fused_bias_act(_ptr(x), _ptr(bias), _ptr(ref), _ptr(out), C.c_int64(x.size), step_b,
                             0 if bias is None else bias.size, act, grad, C.c_float(alpha), C.c_float(scale))
    return out


def upfirdn2d(x, k, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    """x [major,in_h,in_w,minor], k [kh,kw] -> [major,out_h,out_w,minor]."""
    x, k = _f32(x), _f32(k)
    mj, ih, iw, mn = x.shape
    kh, kw = k.shape
    oh = (ih * up_y + py0 + py1 - kh + down_y) // down_y
    ow = (iw * up_x + px0 + px1 - kw + down_x) // down_x
    out = np.empty((mj, oh, ow, mn), np.float32)
    lib().orc_upfirdn2d(_ptr(x), _ptr(k), _ptr(out), mj, ih, iw, mn, kh, kw, int(up_x), int(up_y), int(down_x),
                        int(down_y), int(px0), int(px1), int(py0), int(py1))
    return out
