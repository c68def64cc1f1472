"""torch-CPU restatement of the whole MRefSR forward path.  TEST INFRASTRUCTURE ONLY.

Functional (state dicts in, tensors out), written to follow the reference module by module and
loop by loop -- per-reference python loops, cat / permute / bmm attention -- so it doubles as the
"port" CPU baseline of bench.py:
    extractor      contras_multi_extractor_arch.py:10-64
    correspondence corres_generation_arch.py:49-118  (matching via oracle/mrefsr_oracle.c)
    vgg19 taps     vgg_arch.py:54-161
    net_g          ref_mrapa_restoration_arch.py:45-348 (DCN via oracle/dcn_torch.py)
Pinned by tests/test_oracle_pipeline.py against the e2e golden vectors (outputs of the
reference's own MultiRefRestorationModel).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import c_api, dcn_torch

FAST_DCN = True  # False: pure-torch gather formulation (oracle/dcn_torch.py); tests run both
# BASELINE configs[4] restatement: weights and activations rounded to bf16 (round-to-nearest-even), fp32
# accumulation, every layer output rounded again -- the rounding points of mrefsr_amd.archs.nhwc.set_arithmetic('bf16')
BF16 = False


def _r(x):
    return x.bfloat16().float() if BF16 else x

_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
_STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def _conv(sd, name, x, padding=1, cin=None, bias=True):
    w = sd[name + '.weight']
    if cin is not None:
        w = w[:, cin[0]:cin[1]]
    return F.conv2d(_r(x), _r(w), sd.get(name + '.bias') if bias else None, 1, padding)


def vgg_to_conv3_1(sd, prefix, x, taps=None):
    """conv1_1 relu conv1_2 relu pool conv2_1 relu conv2_2 relu pool conv3_1 [relu]; `taps` collects
    relu{1,2,3}_1 (VGG19 use) else returns conv3_1 without ReLU (VGG16 extractor use)."""
    x = (x - _MEAN) / _STD
    out = {}
    x = _r(F.relu(_conv(sd, prefix + 'conv1_1', x))); out['relu1_1'] = x
    x = _r(F.relu(_conv(sd, prefix + 'conv1_2', x)))
    x = F.max_pool2d(x, 2, 2)
    x = _r(F.relu(_conv(sd, prefix + 'conv2_1', x))); out['relu2_1'] = x
    x = _r(F.relu(_conv(sd, prefix + 'conv2_2', x)))
    x = F.max_pool2d(x, 2, 2)
    x = _conv(sd, prefix + 'conv3_1', x)
    if taps is None:
        return _r(x)
    out['relu3_1'] = _r(F.relu(x))
    return {k: out[k] for k in taps}


def correspondence(f1, f2, idx_given=None):
    """f1 [B,256,h,w], f2 [B,256,h,w] -> (pre_offset dict of [B,9,sh,sw,2], idx [B,h-2,w-2]); idx_given [B,h-2,w-2]: take these
    match indices instead of matching (tests: net_g of the restatement on the product's own matches)"""
    b, _, h, w = f1.shape
    idxs, offs = [], {1: [], 2: [], 4: []}
    for i in range(b):
        if idx_given is not None:
            idx = np.ascontiguousarray(idx_given[i], dtype=np.int64)
        else:
            idx, _ = c_api.feature_match_index(f1[i].numpy(), f2[i].numpy())
        idxs.append(idx)
        for s, o in zip((1, 2, 4), c_api.offsets_from_idx(idx, h, w)):
            offs[s].append(o)
    pre = {'relu3_1': torch.from_numpy(np.stack(offs[1])), 'relu2_1': torch.from_numpy(np.stack(offs[2])),
           'relu1_1': torch.from_numpy(np.stack(offs[4]))}
    return pre, np.stack(idxs)


def dyn_agg(sd, prefix, ref_feat, feat, pre_offset, dg=8):
    """DynAgg.forward, ref_mrapa_restoration_arch.py:45-76"""
    out = _r(_conv(sd, prefix + 'conv_offset_mask', feat))
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    pre = pre_offset.repeat([1, dg, 1, 1, 1])
    reorder = torch.zeros_like(offset)
    reorder[:, 0::2] = pre[..., 1]
    reorder[:, 1::2] = pre[..., 0]
    offset = offset + reorder
    mask = torch.sigmoid(mask)
    w, bias = sd[prefix + 'weight'], sd[prefix + 'bias']
    if FAST_DCN or BF16:  # im2col (C, OpenMP) + one GEMM per sample: deform_conv_cuda.cpp:539-561 on the CPU
        col = torch.from_numpy(c_api.dcnv2_im2col(ref_feat.numpy(), offset.numpy(), mask.numpy(), 3, 3, 1, 1, 1, dg))
        out = torch.matmul(_r(w).flatten(1), _r(col)) + bias.view(1, -1, 1)
        return out.view(ref_feat.shape[0], w.shape[0], *ref_feat.shape[2:])
    return dcn_torch.modulated_deform_conv2d(ref_feat, offset, mask, w, bias, 1, 1, 1, 1, dg)


def _prelu(x, w):
    return F.prelu(x, w)


def fusion(sd, prefix, target, refs):
    """MRAPAFusion.forward, ref_mrapa_restoration_arch.py:313-348 (literal permute/bmm form)"""
    n, _, h_in, w_in = target.shape
    t = len(refs)

    def pad(x):
        ph, pw = (4 - x.shape[2] % 4) % 4, (4 - x.shape[3] % 4) % 4
        return F.pad(x, [0, pw, 0, ph], mode='reflect') if (ph or pw) else x

    target = pad(target)
    refs = pad(torch.stack(refs, dim=1).flatten(0, 1))
    c = sd[prefix + 'conv_emb1.0.weight'].shape[0]
    et = _r(_r(_prelu(_conv(sd, prefix + 'conv_emb1.0', target, 0), sd[prefix + 'conv_emb1.1.weight'])) * c ** -0.5)
    et = et.permute(0, 2, 3, 1).unsqueeze(3).contiguous().flatten(0, 2)
    emb = _r(_prelu(_conv(sd, prefix + 'conv_emb2.0', refs), sd[prefix + 'conv_emb2.1.weight'])).unflatten(0, (n, t))
    emb = emb.permute(0, 3, 4, 2, 1).contiguous().flatten(0, 2)
    ass = _r(_conv(sd, prefix + 'conv_ass', refs)).unflatten(0, (n, t)).permute(0, 3, 4, 1, 2).contiguous().flatten(0, 2)
    prob = F.softmax(torch.matmul(et, emb), dim=2)
    r = _r(torch.matmul(prob, ass).squeeze(1).unflatten(0, (n, *target.shape[-2:])).permute(0, 3, 1, 2).contiguous())
    attn = _r(F.leaky_relu(_conv(sd, prefix + 'spatial_attn', torch.cat([target, r], 1), 0), 0.1))
    mul = _r(_conv(sd, prefix + 'spatial_attn_mul2', _r(F.leaky_relu(_conv(sd, prefix + 'spatial_attn_mul1', attn), 0.1))))
    add = _r(_conv(sd, prefix + 'spatial_attn_add2', _r(F.leaky_relu(_conv(sd, prefix + 'spatial_attn_add1', attn), 0.1))))
    r = _r(r * torch.sigmoid(mul) * 2 + add)
    feat = _r(F.leaky_relu(_conv(sd, prefix + 'feat_fusion', torch.cat([target, r], 1), 0), 0.1))
    return feat[:, :, :h_in, :w_in]


def _res_blocks(sd, prefix, x, n):
    for i in range(n):
        x = _r(x + _conv(sd, f'{prefix}{i}.conv2', _r(F.relu(_conv(sd, f'{prefix}{i}.conv1', x)))))
    return x


def net_g(sd, x, pre_list, feat_list, n_blocks=16, trace=None):
    """MRAPARestorationNet.forward, ref_mrapa_restoration_arch.py:123-137, :213-259"""
    x = _r(x)
    base = _r(F.interpolate(x, None, 4, 'bilinear', False))
    h = _r(F.leaky_relu(_conv(sd, 'content_extractor.conv_first', x), 0.1))
    x = _res_blocks(sd, 'content_extractor.body.', h, n_blocks)
    p = 'dyn_agg_restore.'
    for scale, key in (('small', 'relu3_1'), ('medium', 'relu2_1'), ('large', 'relu1_1')):
        swapped = []
        for pre, feat in zip(pre_list, feat_list):
            if BF16:  # the build's split form: x half (rounded) + reference half, see _swap_nhwc
                nx = x.shape[1]
                ox = _r(_conv(sd, f'{p}{scale}_offset_conv1', x, cin=(0, nx), bias=False))
                off = _r(F.leaky_relu(_conv(sd, f'{p}{scale}_offset_conv1', feat[key], cin=(nx, nx + feat[key].shape[1])) + ox, 0.1))
            else:
                off = torch.cat([x, feat[key]], 1)
                off = F.leaky_relu(_conv(sd, f'{p}{scale}_offset_conv1', off), 0.1)
            off = _r(F.leaky_relu(_conv(sd, f'{p}{scale}_offset_conv2', off), 0.1))
            swapped.append(_r(F.leaky_relu(dyn_agg(sd, f'{p}{scale}_dyn_agg.', feat[key], off, pre[key]), 0.1)))
        if trace is not None:
            trace[f'{scale}_swapped'] = torch.cat(swapped, 0)
        hh = fusion(sd, f'{p}head_{scale}.', x, swapped)
        if trace is not None:
            trace[f'{scale}_head'] = hh
        hh = _r(_res_blocks(sd, f'{p}body_{scale}.', hh, n_blocks) + x)
        if scale == 'large':
            x = _r(_conv(sd, f'{p}tail_large.2', _r(F.leaky_relu(_conv(sd, f'{p}tail_large.0', hh), 0.1))))
        else:
            x = _r(F.leaky_relu(F.pixel_shuffle(_conv(sd, f'{p}tail_{scale}.0', hh), 2), 0.1))
        if trace is not None:
            trace[f'{scale}_out'] = x
    return _r(x + base)


@torch.no_grad()
def forward(sd_g, sd_extractor, sd_map, data, trace=None, max_idx=None):
    """data: dict of CPU tensors img_in_lq (B,3,h,w), img_in_up (B,3,4h,4w), img_ref_list (B,K,3,4h,4w).
    Returns (output (B,3,4h,4w), max_idx [K,B,h-2,w-2]).  multi_ref_restoration_model.py:281-294.
    max_idx [K,B,h-2,w-2] given: the matching step is skipped and these indices feed index_to_flow (the extractor is then
    not needed either)."""
    sd_g = {k: torch.as_tensor(v) for k, v in sd_g.items()}
    sd_e = {k: torch.as_tensor(v) for k, v in sd_extractor.items()}
    sd_m = {k: torch.as_tensor(v) for k, v in sd_map.items()}
    refs = list(torch.unbind(data['img_ref_list'], dim=1))
    pre_list, feat_list, idxs = [], [], []
    if max_idx is None:
        f1 = vgg_to_conv3_1(sd_e, 'feature_extraction_image1.model.', data['img_in_up'])
    else:   # only the map size is needed: conv3_1 of a (4h, 4w) image is (h, w)
        f1 = torch.empty(data['img_in_up'].shape[0], 256, data['img_in_up'].shape[2] // 4, data['img_in_up'].shape[3] // 4)
    for kk, r in enumerate(refs):
        f2 = vgg_to_conv3_1(sd_e, 'feature_extraction_image2.model.', r) if max_idx is None else None
        pre, idx = correspondence(f1, f2, None if max_idx is None else max_idx[kk])
        pre_list.append(pre)
        idxs.append(idx)
        feat_list.append(vgg_to_conv3_1(sd_m, 'vgg.vgg_net.', r, taps=('relu1_1', 'relu2_1', 'relu3_1')))
    if trace is not None and max_idx is None:
        trace['f1'] = f1
    out = net_g(sd_g, data['img_in_lq'], pre_list, feat_list, trace=trace)
    return out, np.stack(idxs)
