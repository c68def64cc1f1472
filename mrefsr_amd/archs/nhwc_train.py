"""Channels-last TRAINING engine: the autograd side of archs/nhwc.py.

The reference trains net_g with torch.autograd over NCHW tensors (multi_ref_restoration_model.py:197-279: forward,
``l_pix.backward()``, Adam step); on ROCm that is one MIOpen call per convolution and direction plus separate bias /
activation / residual / concatenation kernels (2 200 launches per step at BASELINE configs[2]'s per-GPU shape).  Here the
same graph is recorded over [N,H,W,C] tensors with one autograd node per FUSED launch of the inference engine:

  _Conv          conv_nhwc (bias, broadcast pre-activation term, LeakyReLU / ReLU / PReLU, residual, PixelShuffle, two-source
                 concatenation fused) -- backward: ONE pass for the activation derivative + bias gradient (+ PReLU weight
                 gradient) (mrefsr_act_bwd_nhwc_f32), the input gradients as conv_nhwc launches on the point-mirrored,
                 transposed weights (mrefsr_conv_pack_weight_view_f32: dgrad of a stride-1 'same' convolution IS such a
                 convolution), the weight gradient by MIOpen's channels-last wgrad kernels on the same storage (no transposes)
  _ConvDynAgg    conv_offset_mask + DynAgg glue (mrefsr_conv_dynagg_f32) -- backward through mrefsr_dynagg_prep_bwd_f32
  _Dcn           fused gather + MFMA deformable convolution on channels-last features (mrefsr_dcn_fwd_f32) with its LeakyReLU
                 -- backward: HIP im2col / col2im + two library GEMMs (ops/dcn/deform_conv.py)
  _Attention     mrefsr_mrattn_fwd_nhwc_f32 / mrefsr_mrattn_bwd_nhwc_f32 (softmax recomputed, nothing extra saved)
  _Modulate      refs * sigmoid(mul) * 2 + add, one pass each way

Arithmetic: forward and input-gradient convolutions run the bf16 three-term split (terms 6: fp32-equivalent, no range
limit -- gradients of 1e-8 would be flushed by the fp16 two-term split -- and no host synchronisation when the weights are
re-packed after every optimiser step).  Gradients match the reference's own optimisation step to the fingerprints of
tests/golden/e2e_c2.npz (tests/test_configs_gpu.py).  ``ENABLED`` is flipped by tests/ only (generic NCHW autograd forms as the comparison).
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import hip

# MIOpen's channels-last kernels for the weight gradients (read once by torch, at its first MIOpen convolution)
os.environ.setdefault('PYTORCH_MIOPEN_SUGGEST_NHWC', '1')

ENABLED = True
TERMS = 6
# Forward convolutions on the fp16 two-term split (3 products instead of 6: the small maps of a training step are bound by
# the serial MFMA chain of one block).  Its weight scale 2^s (max|w| 2^s in [2^13, 2^14)) comes from a readback of the
# weight's amax when a parameter is first seen -- not per step: weights drift slowly, and check_scales() (one multi-tensor
# launch per step, no synchronisation) raises the library's range flag if a weight has grown past the cached headroom (4x),
# upon which the model re-runs the step on the range-free kernels and the scales are taken afresh.
FWD_TERMS = int(os.environ.get('MREFSR_TRAIN_FWD_TERMS', '16'))
_scales = {}          # (data_ptr, numel) of a weight -> (weakref, scale, limit = 60000 / scale)
_scale_epoch = [0]


def _scale_of(amax):
    """2^s with amax * 2^s in [2^13, 2^14); None for an (almost) all-zero weight (conv_offset_mask right after its zero
    initialisation): that convolution runs the range-free split until a refresh finds it grown"""
    import math
    if not (amax > 2.0 ** -40 and math.isfinite(amax)):
        return None
    return 2.0 ** (13 - math.floor(math.log2(amax)))


def _wscale(weight):
    import weakref
    key = (weight.data_ptr(), weight.numel())   # the parameter's storage: the same for every Python handle autograd hands back
    hit = _scales.get(key)
    if hit is not None and hit[0]() is not None:
        return hit[1]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError('nhwc_train: the fp16 weight scale of a parameter is not cached yet (first use, or its holder was freed) and '
                           'deriving it needs a readback, which a hipGraph capture cannot contain: run one eager step first')
    s = _scale_of(float(weight.detach().abs().max().item()))
    _scales[key] = (weakref.ref(weight), s, 60000.0 / s if s else float('inf'))
    _scale_epoch[0] += 1
    return s


def reset_scales():
    _scales.clear()
    _scale_epoch[0] += 1


def scale_epoch():
    return _scale_epoch[0]


_lim_cache = {}
REFRESH = 50          # steps between full refreshes of the cached scales (one readback of all amaxes)
_checks = [0]


def check_scales():
    """once per training step, before the forward pass: every cached weight scale still leaves max|w| * scale inside the fp16
    range, else the range flag of hip.conv_range_tripped() is raised (device side: no synchronisation).  Every REFRESH-th
    call re-derives all scales from the current weights (one readback)."""
    import weakref
    live = [(k, v[0]()) for k, v in _scales.items() if v[0]() is not None]
    for k in [k for k, v in _scales.items() if v[0]() is None]:
        del _scales[k]
    if not live:
        return
    amax = torch.stack(torch._foreach_norm([w.detach() for _, w in live], float('inf')))
    _checks[0] += 1
    if _checks[0] % REFRESH == 0:
        changed = False
        for (k, w), a in zip(live, amax.tolist()):
            s = _scale_of(a)
            changed |= s != _scales[k][1]
            _scales[k] = (weakref.ref(w), s, 60000.0 / s if s else float('inf'))
        if changed:
            _scale_epoch[0] += 1
        return
    if _lim_cache.get('key') != (_scale_epoch[0], len(live)):
        _lim_cache.update(key=(_scale_epoch[0], len(live)), lim=torch.tensor([_scales[k][2] for k, _ in live], device=amax.device))
    hip._range_flag(amax.device).bitwise_or_((amax > _lim_cache['lim']).any().to(torch.int32))


def _fwd_terms():
    return TERMS if (FWD_TERMS != 16 or hip.is_range_free()) else 16


# Input-gradient convolutions on the same three-product mode: the output gradients of an L1-trained network sit around
# 1e-6..1e-9, far below the fp16 normal range, so the backward glue kernel also returns max |g| and the convolution kernel
# scales its input by the power of two that brings that maximum to [2^13, 2^14) (mrefsr_conv_nhwc_scaled_f32; exact both ways)
BWD_TERMS = int(os.environ.get('MREFSR_TRAIN_BWD_TERMS', '16'))


# Packed copies of the training weights.  A parameter changes once per step (optimizer_g.step(), ref
# multi_ref_restoration_model.py:277) and is used by two launches (forward operator, input-gradient operator): 346 packing
# launches of ~4 us per step.  The copies live in persistent buffers instead, and ONE multi-tensor launch
# (mrefsr_conv_pack_weights_multi_f32, a job table in device memory) refreshes all of them: at begin_step() -- called by the
# model at the top of every optimisation step, also inside the hipGraph capture -- or at the first lookup that finds a copy
# older than its parameter (``_version``; callers that drive net_g without the model).  A new (weight, slice, arithmetic)
# is packed by itself once and joins the table of the following refresh.
_packs = {}        # device index -> {'entries': {key: [weakref, PackedWeight, stamp, job, last refresh it was used in]}, 'table', 'rows', 'dirty', 'refresh'}
PACK_MULTI = os.environ.get('MREFSR_TRAIN_PACK_MULTI', '1') != '0'
_PACK_KEEP = 4     # refreshes an unused entry survives (validation passes, a second network)


def _pack_state(device):
    st = _packs.get(device.index)
    if st is None:
        st = _packs[device.index] = dict(entries={}, table=None, rows=[], dirty=True, refresh=0)
    return st


def _refresh_packs(device):
    """repack every live entry of ``device`` in one launch; False when the job table would have to be rebuilt (an upload)
    inside a hipGraph capture"""
    st = _pack_state(device)
    cap, epoch = hip.capture_epoch()
    ent = st['entries']
    # dead: the parameter is gone, its storage was replaced (`p.data = ...`, `.to()`: the job row still holds the old address),
    # or nothing has asked for the copy for _PACK_KEEP refreshes
    dead = [k for k, e in ent.items() if e[0]() is None or e[0]().data_ptr() != k[0] or st['refresh'] - e[4] > _PACK_KEEP]
    if (dead or st['dirty']) and cap:
        return False
    for k in dead:
        del ent[k]
    st['refresh'] += 1
    if dead or st['dirty']:
        st['rows'] = list(ent.values())
        st['table'] = hip.conv_pack_table([e[3] for e in st['rows']], device) if st['rows'] else None
        st['dirty'] = False
    if st['rows']:
        hip.conv_pack_multi(st['table'], len(st['rows']))
        for e in st['rows']:
            e[2] = (e[0]()._version, epoch)
    return True


def begin_step(device=None):
    """top of an optimisation step: refresh every packed weight copy (one launch per device that has any)"""
    if not PACK_MULTI:
        return
    for idx in list(_packs) if device is None else [torch.device(device).index]:
        if idx in _packs and _packs[idx]['entries']:
            _refresh_packs(torch.device('cuda', idx))


def _packed(weight, cin_slice, terms, dgrad=False, wscale=1.0):
    """hip.conv_pack_view(weight, cin_slice, terms, dgrad, wscale) from the step's refreshed copies"""
    if not PACK_MULTI:
        return hip.conv_pack_view(weight, cin_slice, terms, dgrad=dgrad, wscale=wscale)
    st = _pack_state(weight.device)
    cap, epoch = hip.capture_epoch()
    key = (weight.data_ptr(), weight.numel(), tuple(weight.shape), cin_slice, terms, dgrad, wscale)
    e = st['entries'].get(key)
    stamp = (weight._version, epoch)
    if e is not None and e[0]() is not None:
        e[4] = st['refresh']
        if e[2] != stamp:
            if not (_refresh_packs(weight.device) and e[2] == stamp):
                hip.conv_pack_one(e[3])   # (table being rebuilt under capture, or the entry is not in the table yet): this copy alone
                e[2] = stamp
        return e[1]
    import weakref
    pw, job = hip.conv_pack_plan(weight, cin_slice, terms, dgrad, wscale)
    hip.conv_pack_one(job)
    if not cap:   # copies of the same (weight, slice, arithmetic, operator) at a superseded scale: nobody reads them again, and a
        for k in [k for k in st['entries'] if k[:6] == key[:6] and k[6] != wscale]:   # repack could raise the range flag for them
            del st['entries'][k]
    st['entries'][key] = [weakref.ref(weight), pw, stamp, job, st['refresh']]
    st['dirty'] = True
    return pw


def reset_packs():
    _packs.clear()


def capture_state():
    """(objects, addresses) a hipGraph captured now has baked in from this module's caches: the job table of the multi-tensor pack
    launch, every packed copy and the parameter each one is made from.  The model keeps the objects alive as long as the graph and
    compares the addresses before every replay (a table rebuilt by an eager pass in between, a parameter whose storage was
    replaced: the graph would read freed memory)."""
    objs, ptrs = [], []
    for idx in sorted(_packs):
        st = _packs[idx]
        objs.append((st['table'], list(st['rows'])))
        ptrs.append(st['table'].data_ptr() if st['table'] is not None else 0)
        for e in st['rows']:
            w = e[0]()
            ptrs.append((e[1].data.data_ptr(), w.data_ptr() if w is not None else -1))
    ptrs.append(hip.capture_ptrs())   # workspaces (DCN, zero chunks, weight-gradient partials): a regrown one has moved
    return objs, tuple(ptrs)


def _bwd_pack(weight, cin_slice):
    """(packed dgrad operator, terms): fp16 two-term with the weight's cached scale, else the range-free split"""
    if BWD_TERMS == 16 and not hip.is_range_free():
        ws = _wscale(weight)
        if ws is not None:
            return _packed(weight, cin_slice, 16, dgrad=True, wscale=ws), 16
    return _packed(weight, cin_slice, TERMS, dgrad=True), TERMS


def _unshuffle(t):
    """inverse of PixelShuffle(2) on channels-last storage: [N,2H,2W,C] -> [N,H,W,4C] (channel 4c + 2i + j)"""
    n, h2, w2, c = t.shape
    return t.view(n, h2 // 2, 2, w2 // 2, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(n, h2 // 2, w2 // 2, 4 * c)


DCN_FUSED_BWD = os.environ.get('MREFSR_TRAIN_DCN_FUSED', '1') != '0'   # 0: im2col / library GEMMs / col2im (the round-3 structure)

# weight gradients on the library's own kernels (csrc/wgrad.hip).  The generic forms below serve what those refuse: a batch that is
# being re-run on the range-free path (an activation left the fp16 range) -- see INTEGRATION.md
WGRAD_HIP = True


def _wgrad(g_pre, cout, x, cin, k, amax=None):
    """d loss / d weight [cout,cin,k,k] from channels-last storage (g_pre [N,H,W,>=cout], x [N,H,W,>=cin])"""
    if k == 3 and WGRAD_HIP and amax is not None and not hip.is_range_free():
        return hip.conv_wgrad3x3(x, g_pre, cin, cout, amax)
    if k == 1 and WGRAD_HIP and amax is not None and not hip.is_range_free():
        return hip.conv_wgrad1x1(x, g_pre, cin, cout, amax)
    if k == 1:
        # (range-free re-runs) a plain GEMM over the pixels, g^T [cout, P] . x [P, cin], on the tensors as they lie; K = P is ~10^5 against M, N of a
        # few hundred, so it is split into S batches (a library GEMM has no split-K for this shape: 4x slower) and the
        # S partial results are added
        g2, x2 = g_pre.reshape(-1, g_pre.shape[3]), x.reshape(-1, x.shape[3])
        p_ = g2.shape[0]
        split = next((d for d in (128, 100, 64, 50, 40, 32, 25, 20, 16, 10, 8, 5, 4, 2) if p_ % d == 0 and p_ // d >= 256), 1)
        g3 = g2.view(split, p_ // split, g2.shape[1])[:, :, :cout]
        x3 = x2.view(split, p_ // split, x2.shape[1])[:, :, :cin]
        return torch.bmm(g3.transpose(1, 2), x3).sum(0).view(cout, cin, 1, 1)
    g = g_pre.permute(0, 3, 1, 2)
    xi = x.permute(0, 3, 1, 2)
    if g.shape[1] != cout:
        g = g[:, :cout]
    if xi.shape[1] != cin:
        xi = xi[:, :cin]
    w = torch.empty((cout, cin, k, k), device=x.device, dtype=x.dtype).contiguous(memory_format=torch.channels_last)
    _, gw, _ = torch.ops.aten.convolution_backward(g, xi, w, None, [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [False, True, False])
    return gw


class _Conv(Function):
    """out = act(conv(cat[x1, x2]; weight[:, a:b]) + bias + pre) + residual, optionally through PixelShuffle(2)"""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, prelu_w, pre, residual, slope, epilogue, cin_slice):
        co, ci, k, _ = weight.shape
        act = 2 if prelu_w is not None else (1 if slope is not None else 0)
        if act and residual is not None:
            raise NotImplementedError('nhwc_train: activation followed by a residual add in one launch has no fused backward')
        if cin_slice is not None and x2 is not None:
            raise NotImplementedError('nhwc_train: cin_slice with a second source')
        weight = weight.contiguous()
        a, b = cin_slice if cin_slice is not None else (0, ci)
        terms, wscale = _fwd_terms(), 1.0
        if terms == 16:
            wscale = _wscale(weight)
            if wscale is None:
                terms, wscale = TERMS, 1.0
        packed = _packed(weight, (a, b), terms, wscale=wscale)
        out = hip.conv_nhwc(x1, packed, bias, co, k, x2=x2, pre=pre, residual=residual, act=act != 0, slope=slope if act == 1 else 0.0,
                            slope_ptr=prelu_w, epilogue=epilogue)
        ctx.meta = (act, slope, epilogue, (a, b), k, bias is not None, 0 if pre is None else pre.shape[0])
        ctx.save_for_backward(x1, x2, weight, prelu_w, out if act else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x1, x2, weight, prelu_w, out = ctx.saved_tensors
        act, slope, epilogue, (a, b), k, has_bias, pre_n = ctx.meta
        need = ctx.needs_input_grad
        co = weight.shape[0]
        g = g.contiguous()
        g_res = g if need[6] else None
        if epilogue == 2:
            g = _unshuffle(g)
            if act:
                out = _unshuffle(out)
        dgrad = need[0] or (x2 is not None and need[1])
        g_pre, g_bias, g_slope, amax = hip.act_bwd_nhwc(g, out, act, slope if act == 1 else 0.0, prelu_w, want_bias=has_bias and need[3],
                                                        want_amax=True) if (dgrad or need[2]) else \
            hip.act_bwd_nhwc(g, out, act, slope if act == 1 else 0.0, prelu_w, want_bias=has_bias and need[3]) + (None,)
        n, h, w, _ = g_pre.shape
        g_x1 = g_x2 = g_w = g_p = None
        c1 = x1.shape[3] if x2 is not None else b - a
        if need[0]:
            if x1.shape[3] != c1 or x1.shape[0] != n:
                raise NotImplementedError('nhwc_train: gradient of a channel-padded / batch-broadcast input')
            pk, terms = _bwd_pack(weight, (a, a + c1))
            g_x1 = hip.conv_nhwc(g_pre, pk, None, c1, k, in_amax=amax if terms == 16 else None)
        if x2 is not None and need[1]:
            if x2.shape[0] != n:
                raise NotImplementedError('nhwc_train: gradient of a batch-broadcast input')
            pk, terms = _bwd_pack(weight, (c1, c1 + x2.shape[3]))
            g_x2 = hip.conv_nhwc(g_pre, pk, None, x2.shape[3], k, in_amax=amax if terms == 16 else None)
        if need[2]:
            gw1 = _wgrad(g_pre, co, x1, c1, k, amax)
            if x2 is not None:
                g_w = torch.cat([gw1, _wgrad(g_pre, co, x2, x2.shape[3], k, amax)], 1)
            elif (a, b) == (0, weight.shape[1]):
                g_w = gw1
            else:
                g_w = torch.zeros_like(weight)
                g_w[:, a:b] = gw1
        if pre_n and need[5]:
            gp = g_pre if g_pre.shape[3] == co else g_pre[..., :co]
            g_p = gp.reshape(n // pre_n, pre_n, h, w, co).sum(0) if pre_n != n else gp
        return g_x1, g_x2, g_w, g_bias, g_slope, g_p, g_res, None, None, None


class _ResBlock(Function):
    """x + conv2(relu(conv1(x))) -- ResidualBlockNoBN with res_scale 1 (arch_util.py:45-70) -- as ONE autograd node: recorded as
    two _Conv nodes, autograd adds the two gradients of x (identity path and conv1's input gradient) in a launch of its own, 48
    times per trunk; here conv1's input-gradient convolution takes the incoming gradient as its fused residual term."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        c = w1.shape[0]
        w1, w2 = w1.contiguous(), w2.contiguous()
        pks = []
        for w in (w1, w2):
            terms, wscale = _fwd_terms(), 1.0
            if terms == 16:
                wscale = _wscale(w)
                if wscale is None:
                    terms, wscale = TERMS, 1.0
            pks.append(_packed(w, (0, w.shape[1]), terms, wscale=wscale))
        t = hip.conv_nhwc(x, pks[0], b1, c, 3, act=True, slope=0.0)
        out = hip.conv_nhwc(t, pks[1], b2, w2.shape[0], 3, residual=x)
        ctx.save_for_backward(x, t, w1, w2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, t, w1, w2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        c = w1.shape[0]
        g = g.contiguous()
        _, g_b2, _, amax2 = hip.act_bwd_nhwc(g, None, 0, want_bias=ctx.has_bias[1] and need[4], want_amax=True)
        pk, terms = _bwd_pack(w2, (0, c))
        g_t = hip.conv_nhwc(g, pk, None, c, 3, in_amax=amax2 if terms == 16 else None)
        g_w2 = _wgrad(g, w2.shape[0], t, c, 3, amax2) if need[3] else None
        g_pre, g_b1, _, amax1 = hip.act_bwd_nhwc(g_t, t, 1, 0.0, want_bias=ctx.has_bias[0] and need[2], want_amax=True)
        g_x = None
        if need[0]:
            pk, terms = _bwd_pack(w1, (0, w1.shape[1]))
            g_x = hip.conv_nhwc(g_pre, pk, None, w1.shape[1], 3, residual=g, in_amax=amax1 if terms == 16 else None)
        g_w1 = _wgrad(g_pre, c, x, w1.shape[1], 3, amax1) if need[1] else None
        return g_x, g_w1, g_b1, g_w2, g_b2


class _ResChain(Function):
    """A whole trunk of residual blocks (nn.Sequential of ResidualBlockNoBN, 16 per scale in MRAPARestorationNet) as ONE node:
    the input gradients run block by block as in _ResBlock, the 2 * n_blocks weight gradients -- which feed nothing inside
    backward -- are left for the end and run as one batched launch pair (mrefsr_conv_wgrad3x3_batch_f32): 2 launches instead of
    64 per trunk, and on the small maps the blocks of all 32 jobs fill the chip together instead of 80 at a time."""

    @staticmethod
    def forward(ctx, x, *params):
        nb = len(params) // 4
        c = params[0].shape[0]
        saved = [x]
        for i in range(nb):
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            pks = []
            for w in (w1, w2):
                terms, wscale = _fwd_terms(), 1.0
                if terms == 16:
                    wscale = _wscale(w)
                    if wscale is None:
                        terms, wscale = TERMS, 1.0
                pks.append(_packed(w, (0, c), terms, wscale=wscale))
            t = hip.conv_nhwc(x, pks[0], b1, c, 3, act=True, slope=0.0)
            x = hip.conv_nhwc(t, pks[1], b2, c, 3, residual=x)
            saved += [t, x]
        ctx.nb = nb
        ctx.save_for_backward(*saved[:-1], *params)     # x_0, (t_i, x_i+1) ... without the output
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        nb = ctx.nb
        acts, params = ctx.saved_tensors[:2 * nb], ctx.saved_tensors[2 * nb:]
        c = params[0].shape[0]
        g = g.contiguous()
        grads = [None] * (4 * nb)
        jobs_x, jobs_g, jobs_a, slots = [], [], [], []
        stats = None   # (bias gradient of conv2, max |g|) of the current g when the launch that produced it has already made them
        for i in reversed(range(nb)):
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            x, t = acts[2 * i], acts[2 * i + 1]
            if stats is None:
                _, g_b2, _, amax2 = hip.act_bwd_nhwc(g, None, 0, want_bias=True, want_amax=True)
            else:
                g_b2, amax2 = stats
            pk2, terms2 = _bwd_pack(w2, (0, c))
            pk1, terms1 = _bwd_pack(w1, (0, c))
            if FUSED_BWD and terms1 == 16 and terms2 == 16:
                # both element-wise passes of the block ride on the input-gradient launches: conv2's takes the ReLU mask from t and
                # leaves conv1's bias gradient and the scale of its own output; conv1's adds the skip and leaves the same for the
                # block above (mrefsr_conv_nhwc_bwd_f32) -- two launches per block instead of four, no extra pass over g
                g_pre, g_b1, amax1 = hip.conv_nhwc_bwd(g, pk2, c, 3, residual=t, residual_is_mask=True, in_amax=amax2)
                g_in, sb, sa = hip.conv_nhwc_bwd(g_pre, pk1, c, 3, residual=g, in_amax=amax1, want_stats=i > 0)
                stats = (sb, sa) if i > 0 else None
            else:
                g_t = hip.conv_nhwc(g, pk2, None, c, 3, in_amax=amax2 if terms2 == 16 else None)
                g_pre, g_b1, _, amax1 = hip.act_bwd_nhwc(g_t, t, 1, 0.0, want_bias=True, want_amax=True)
                g_in = hip.conv_nhwc(g_pre, pk1, None, c, 3, residual=g, in_amax=amax1 if terms1 == 16 else None)
                stats = None
            grads[4 * i + 1], grads[4 * i + 3] = g_b1, g_b2
            jobs_x += [t, x]
            jobs_g += [g, g_pre]
            jobs_a += [amax2, amax1]
            slots += [4 * i + 2, 4 * i]
            g = g_in
        if hip.is_range_free():
            for x_, g_, a_, s_ in zip(jobs_x, jobs_g, jobs_a, slots):
                grads[s_] = _wgrad(g_, c, x_, c, 3, a_)
        else:
            for k in range(0, len(jobs_x), 32):
                dw = hip.conv_wgrad3x3_batch(jobs_x[k:k + 32], jobs_g[k:k + 32], c, c, jobs_a[k:k + 32])
                for j, s_ in enumerate(slots[k:k + 32]):
                    grads[s_] = dw[j]
        return (g if ctx.needs_input_grad[0] else None, *grads)


class _ConvDynAgg(Function):
    """(feat [N,H,W,C], conv_offset_mask weight / bias, pre_offset) -> planar (offset, mask) for the DCN   ref :56-73"""

    @staticmethod
    def forward(ctx, feat, weight, bias, pre_offset, dg, abs_sum):
        weight = weight.contiguous()
        terms, wscale = _fwd_terms(), 1.0
        if terms == 16:
            wscale = _wscale(weight)
            if wscale is None:
                terms, wscale = TERMS, 1.0
        offset, mask = hip.conv_dynagg(feat, _packed(weight, None, terms, wscale=wscale), bias, pre_offset, dg, abs_sum)
        ctx.dg = dg
        ctx.save_for_backward(feat, weight, mask)
        return offset, mask

    @staticmethod
    @once_differentiable
    def backward(ctx, g_offset, g_mask):
        feat, weight, mask = ctx.saved_tensors
        co = weight.shape[0]
        if 27 * ctx.dg <= 256:   # channels-last result, bias gradient and max |g| from ONE pass (no transposing copy, no reduction pass)
            g_om, g_bias, amax = hip.dynagg_prep_bwd_nhwc(g_offset.contiguous(), g_mask.contiguous(), mask, ctx.dg, want_bias=ctx.needs_input_grad[2])
        else:
            g_om = hip.dynagg_prep_bwd(g_offset.contiguous(), g_mask.contiguous(), mask, ctx.dg)     # [N,27dg,H,W]
            g_om = g_om.permute(0, 2, 3, 1).contiguous()
            _, g_bias, _, amax = hip.act_bwd_nhwc(g_om, None, 0, want_bias=ctx.needs_input_grad[2], want_amax=True)
        g_feat = g_w = None
        if ctx.needs_input_grad[0]:
            pk, terms = _bwd_pack(weight, None)
            g_feat = hip.conv_nhwc(g_om, pk, None, feat.shape[3], 3, in_amax=amax if terms == 16 else None)
        if ctx.needs_input_grad[1]:
            g_w = _wgrad(g_om, co, feat, feat.shape[3], 3, amax)
        return g_feat, g_w, g_bias, None, None, None


class _Dcn(Function):
    """lrelu(DCNv2(x; offset, mask), act_slope) on channels-last x [N,H,W,C] -> [N,H,W,Co]; offset / mask planar"""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, dg, act_slope):
        out = hip.dcn_fwd(x, offset, mask, weight, bias, 1, 1, 1, 1, dg, act_slope, channels_last=True)
        ctx.dg, ctx.act_slope = dg, act_slope
        ctx.save_for_backward(x, offset, mask, weight, out if act_slope != 1.0 else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from ..ops.dcn.deform_conv import _backward
        x, offset, mask, weight, out = ctx.saved_tensors
        g = g.contiguous()
        c, co = x.shape[3], weight.shape[0]
        ws = _wscale(weight) if (DCN_FUSED_BWD and not hip.is_range_free()) else None
        if ws is not None and c % 32 == 0 and c // ctx.dg in (8, 16, 32) and co % 16 == 0:
            # fused: d columns = W^T . g on the matrix pipe with the offset / mask / input gradients as its epilogue, d W with the
            # columns re-gathered inside the GEMM -- no C*9*H*W buffer, no library GEMM (deform_conv_cuda.cpp:571-685 as two launches)
            g_pre, g_bias, _, amax = hip.act_bwd_nhwc(g, out, 0 if out is None else 1, ctx.act_slope, want_bias=ctx.needs_input_grad[4], want_amax=True)
            need_x = ctx.needs_input_grad[0]
            gx = goff = gm = gw = None
            if need_x or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                pk = _packed(weight, None, 16, dgrad='T', wscale=ws)
                gx, goff, gm = hip.dcn_bwd_data(g_pre, x, offset, mask, pk, ctx.dg, g_amax=amax, need_grad_x=need_x)
                if gx is not None:
                    gx = gx.permute(0, 2, 3, 1).contiguous()
            if ctx.needs_input_grad[3]:
                gw = hip.dcn_bwd_weight(g_pre, x, offset, mask, co, ctx.dg, g_amax=amax)
            return gx, goff, gm, gw, g_bias, None, None
        g_pre, g_bias, _ = hip.act_bwd_nhwc(g, out, 0 if out is None else 1, ctx.act_slope, want_bias=ctx.needs_input_grad[4])
        ctx.stride, ctx.padding, ctx.dilation, ctx.groups, ctx.deformable_groups = 1, 1, 1, 1, ctx.dg
        gx, goff, gm, gw, _ = _backward(ctx, g_pre.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2).contiguous(), offset, mask, weight,
                                        False, ctx.needs_input_grad[0], fused=False)   # (the fused kernels were refused above)
        if gx is not None:
            gx = gx.permute(0, 2, 3, 1).contiguous()
        return gx, goff, gm, gw, g_bias, None, None


class _Attention(Function):
    """softmax_t(<q, emb_t>) . ass_t per pixel on channels-last tensors (t-major references)   ref :321-335"""

    @staticmethod
    def forward(ctx, q, emb, ass, t):
        ctx.t = t
        ctx.save_for_backward(q, emb, ass)
        return hip.mrattn_fwd_nhwc(q, emb, ass, t)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        q, emb, ass = ctx.saved_tensors
        return hip.mrattn_bwd_nhwc(q, emb, ass, g.contiguous(), ctx.t) + (None,)


class _Modulate(Function):
    """refs * sigmoid(mul) * 2 + add   ref :343-345"""

    @staticmethod
    def forward(ctx, refs, mul, add):
        ctx.save_for_backward(refs, mul)
        return hip.attn_modulate_(refs, mul.clone(), add)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        refs, mul = ctx.saved_tensors
        g = g.contiguous()
        return hip.attn_modulate_bwd(g, refs, mul) + (g,)


def recording(*tensors):
    """autograd is on and one of the tensors / parameters is part of a graph"""
    return ENABLED and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def conv(mod, x1, x2, slope, prelu, pre, residual, epilogue, cin_slice, bias):
    prelu_w = None
    if prelu is not None:
        if prelu.weight.numel() != 1:
            raise NotImplementedError('nhwc.conv: per-channel PReLU')
        if hip.is_range_free():   # re-run after a PReLU slope <= 0 was met: unfused activation (its backward needs the sign of x)
            y = _Conv.apply(x1, x2, mod.weight, mod.bias if bias else None, None, pre, residual, None, epilogue, cin_slice)
            return torch.nn.functional.prelu(y, prelu.weight)
        prelu_w = prelu.weight
    return _Conv.apply(x1, x2, mod.weight, mod.bias if bias else None, prelu_w, pre, residual, slope, epilogue, cin_slice)


RESBLOCK = os.environ.get('MREFSR_TRAIN_RESBLOCK', '1') != '0'


def resblock(blk, x):
    """one residual block of a trunk as a single node, or None when it does not have that form (the caller records two convolutions)"""
    c1, c2 = blk.conv1, blk.conv2
    if not (RESBLOCK and blk.res_scale == 1 and c1.kernel_size == (3, 3) and c2.kernel_size == (3, 3) and x.shape[3] == c1.in_channels
            and c1.in_channels % 4 == 0 and c1.out_channels % 4 == 0 and c2.out_channels == c1.in_channels):
        return None
    return _ResBlock.apply(x, c1.weight, c1.bias, c2.weight, c2.bias)


RESCHAIN = os.environ.get('MREFSR_TRAIN_RESCHAIN', '1') != '0'
FUSED_BWD = os.environ.get('MREFSR_TRAIN_FUSED_BWD', '1') != '0'


def reschain(blocks, x):
    """a trunk of residual blocks as a single node, or None when the blocks do not all have the plain form (3x3, one width, biases,
    res_scale 1, every parameter trained): the caller records them one by one"""
    blocks = list(blocks)
    if not (RESCHAIN and RESBLOCK and len(blocks) >= 2):
        return None
    c = x.shape[3]
    params = []
    for blk in blocks:
        for cv in (blk.conv1, blk.conv2):
            if not (cv.kernel_size == (3, 3) and cv.in_channels == c and cv.out_channels == c and c % 4 == 0 and cv.bias is not None
                    and cv.weight.requires_grad and cv.bias.requires_grad):
                return None
            params += [cv.weight, cv.bias]
        if blk.res_scale != 1:
            return None
    return _ResChain.apply(x, *params)


conv_dynagg = _ConvDynAgg.apply
dcn = _Dcn.apply
attention = _Attention.apply
modulate = _Modulate.apply
