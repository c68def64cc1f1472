"""GPU: the channels-last training engine (mrefsr_amd/archs/nhwc_train.py) node by node against torch.autograd in fp64 on
the CPU (the reference trains with plain autograd: multi_ref_restoration_model.py:197-279), and the whole training step on
both engines.  Tolerances: fp32-equivalent convolutions, 2e-5 relative to the tensor's largest magnitude."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, want, tol=2e-5):
    got, want = got.detach().double().cpu(), want.detach().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = float(want.abs().max()) + 1e-30
    err = float((got - want).abs().max()) / scale
    assert err <= tol, err


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


CASES = [
    # cin1, cin2, cout, k, act, extras
    dict(c1=64, c2=0, co=64, k=3, act='lrelu'),
    dict(c1=64, c2=0, co=64, k=3, act='relu'),
    dict(c1=64, c2=0, co=64, k=3, act=None, residual=True),
    dict(c1=64, c2=128, co=64, k=1, act='lrelu'),
    dict(c1=64, c2=64, co=64, k=3, act='lrelu'),
    dict(c1=64, c2=256, co=64, k=3, act='lrelu'),
    dict(c1=32, c2=0, co=3, k=3, act=None),
    dict(c1=3, c2=0, co=64, k=3, act='lrelu', pad4=True),
    dict(c1=64, c2=0, co=128, k=3, act='prelu'),
    dict(c1=64, c2=0, co=256, k=3, act='lrelu', shuffle=True),
    dict(c1=128, c2=0, co=128, k=3, act='lrelu', pre=2, cin_slice=(64, 192), bias=True),
    dict(c1=64, c2=0, co=128, k=3, act=None, cin_slice=(0, 64), bias=False),
    dict(c1=64, c2=0, co=216, k=3, act=None),
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: '-'.join(f'{k}{v}' for k, v in c.items()))
def test_conv_node_matches_fp64_autograd(case):
    from mrefsr_amd.archs import nhwc
    torch.manual_seed(5)
    n, h, w = 4, 20, 36
    c1, c2, co, k = case['c1'], case['c2'], case['co'], case['k']
    a, b = case.get('cin_slice', (0, c1 + c2))
    ci_w = max(b, c1 + c2) if 'cin_slice' not in case else 192
    conv = nn.Conv2d(ci_w, co, k, 1, k // 2).cuda()
    prelu = nn.PReLU().cuda() if case['act'] == 'prelu' else None
    x1 = torch.randn(n, c1, h, w)
    x2 = torch.randn(n, c2, h, w) if c2 else None
    pre = torch.randn(case['pre'], co, h, w) if case.get('pre') else None
    res = torch.randn(n, co, h, w) if case.get('residual') else None
    use_bias = case.get('bias', True)

    # fp64 CPU reference with the literal ops
    W, B = conv.weight.detach().double().cpu().requires_grad_(), conv.bias.detach().double().cpu().requires_grad_()
    r1 = x1.double().requires_grad_(not case.get('pad4'))
    r2 = x2.double().requires_grad_() if c2 else None
    rp = pre.double().requires_grad_() if pre is not None else None
    rr = res.double().requires_grad_() if res is not None else None
    rw = prelu.weight.detach().double().cpu().requires_grad_() if prelu is not None else None
    y = F.conv2d(torch.cat([r1, r2], 1) if c2 else r1, W[:, a:b], B if use_bias else None, 1, k // 2)
    if rp is not None:
        y = (y.view(n // rp.shape[0], rp.shape[0], co, h, w) + rp.unsqueeze(0)).view(n, co, h, w)
    y = {'lrelu': lambda t: F.leaky_relu(t, 0.1), 'relu': F.relu, 'prelu': lambda t: F.prelu(t, rw), None: lambda t: t}[case['act']](y)
    if rr is not None:
        y = y + rr
    if case.get('shuffle'):
        y = F.pixel_shuffle(y, 2)
    gy = torch.randn(y.shape, dtype=torch.float64)
    y.backward(gy)

    def gpu_in(t, req=True):
        return None if t is None else _nhwc(t).cuda().requires_grad_(req)
    g1 = gpu_in(F.pad(x1, (0, 0, 0, 0, 0, 1)) if case.get('pad4') else x1, not case.get('pad4'))
    g2, gp, gr = gpu_in(x2), gpu_in(pre), gpu_in(res)
    slope = {'lrelu': 0.1, 'relu': 0.0}.get(case['act'])
    out = nhwc.conv(conv, g1, x2=g2, slope=slope, prelu=prelu, pre=gp, residual=gr, epilogue=2 if case.get('shuffle') else 0,
                    cin_slice=case.get('cin_slice'), bias=use_bias)
    assert out.requires_grad
    _close(out.permute(0, 3, 1, 2), y)
    out.backward(_nhwc(gy).float().cuda())
    _close(conv.weight.grad, W.grad)
    if use_bias:
        _close(conv.bias.grad, B.grad)
    if not case.get('pad4'):
        _close(g1.grad.permute(0, 3, 1, 2), r1.grad)
    for gt, rt in ((g2, r2), (gp, rp), (gr, rr)):
        if gt is not None:
            _close(gt.grad.permute(0, 3, 1, 2), rt.grad)
    if prelu is not None:
        _close(prelu.weight.grad, rw.grad)


@pytest.mark.parametrize('c', [64, 128, 256])
def test_attention_and_modulation_nodes_match_fp64_autograd(c):
    from mrefsr_amd.archs import nhwc_train
    torch.manual_seed(c)
    n, t, h, w = 2, 5, 12, 20
    q, emb, ass = torch.randn(n, h, w, c) * c ** -0.5, torch.randn(t * n, h, w, c), torch.randn(t * n, h, w, 2 * c)
    rq, re, ra = (v.double().requires_grad_() for v in (q, emb, ass))
    logit = torch.einsum('nhwc,tnhwc->nhwt', rq, re.view(t, n, h, w, c))
    want = torch.einsum('nhwt,tnhwc->nhwc', torch.softmax(logit, -1), ra.view(t, n, h, w, 2 * c))
    g = torch.randn_like(want)
    want.backward(g)
    gq, ge, ga = (v.cuda().requires_grad_() for v in (q, emb, ass))
    got = nhwc_train.attention(gq, ge, ga, t)
    _close(got, want)
    got.backward(g.float().cuda())
    for a, b in ((gq, rq), (ge, re), (ga, ra)):
        _close(a.grad, b.grad)
    # modulation
    r, m, a = (torch.randn(n, h, w, 2 * c) for _ in range(3))
    rr, rm, ra = (v.double().requires_grad_() for v in (r, m, a))
    want = rr * torch.sigmoid(rm) * 2 + ra
    want.backward(g)
    gr, gm, ga = (v.cuda().requires_grad_() for v in (r, m, a))
    got = nhwc_train.modulate(gr, gm, ga)
    _close(got, want)
    got.backward(g.float().cuda())
    for x, y in ((gr, rr), (gm, rm), (ga, ra)):
        _close(x.grad, y.grad)


def test_training_step_on_both_engines(golden, monkeypatch):
    """the channels-last engine (default) and the MIOpen / NCHW autograd path (MREFSR_NHWC_TRAIN=0) produce the same loss
    and the same gradients, parameter by parameter, at the small golden shape"""
    from test_configs_gpu import _golden_model
    from mrefsr_amd.archs import nhwc_train
    g = golden('e2e_c0')
    grads, losses = [], []
    for enabled in (True, False):
        monkeypatch.setattr(nhwc_train, 'ENABLED', enabled)
        model, data, _ = _golden_model(g, True)
        model.feed_data(data)
        model.optimize_parameters(1)
        losses.append(float(model.get_current_log()['l_g_pix']))
        grads.append({n: p.grad.detach().double().cpu() for n, p in model.get_bare_model(model.net_g).named_parameters()})
    assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[1])
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-9, n


def test_graph_replayed_training_steps_equal_eager_steps(golden, monkeypatch):
    """MREFSR_TRAIN_GRAPH=1: forward + backward and the Adam update replayed as hipGraphs (after three eager steps) leave
    the same parameters as six eager steps of the same model (same kernels, same capturable Adam)"""
    from test_configs_gpu import _golden_model
    monkeypatch.setenv('MREFSR_TRAIN_GRAPH', '1')
    g = golden('e2e_c0')
    finals, losses = [], []
    for graphed in (True, False):
        model, data, _ = _golden_model(g, True)
        if not graphed:
            monkeypatch.setattr(type(model), '_optimize_graphed', lambda self, step: False)
        for it in range(1, 7):
            model.feed_data(data)
            model.optimize_parameters(it)
        if graphed:
            assert model._tgraph['fb'] is not None
        losses.append(float(model.get_current_log()['l_g_pix']))
        finals.append({n: p.detach().double().cpu() for n, p in model.get_bare_model(model.net_g).named_parameters()})
    # Adam moves every element by ~lr whatever the gradient's size, so the summation-order noise of the float atomics in the
    # DCN / split-K weight gradients can flip single near-zero elements by 2 lr per step: compare the bulk, not the worst element
    assert abs(losses[0] - losses[1]) <= 1e-3 * abs(losses[1]), losses
    bad = tot = 0
    for n in finals[0]:
        a, b = finals[0][n], finals[1][n]
        bad += int(((a - b).abs() > 2e-5 * float(b.abs().max()) + 1e-7).sum())
        tot += a.numel()
        assert float((a - b).abs().max()) <= 6 * 2.5e-4, n          # never more than the six steps could move an element apart
    print(f'graph vs eager after 6 steps: {bad} of {tot} elements differ, losses {losses}')
    assert bad <= 0.05 * tot, (bad, tot)


def test_graph_replay_survives_a_per_iteration_learning_rate_schedule(golden, monkeypatch):
    """update_learning_rate with warm-up (base_model.py:172-193) moves the learning rates every iteration: the captured Adam update
    reads them from device tensors, so the graphs are kept (no recapture) and the parameters follow the schedule -- the same
    steps taken eagerly give the same parameters"""
    from test_configs_gpu import _golden_model
    monkeypatch.setenv('MREFSR_TRAIN_GRAPH', '1')
    g = golden('e2e_c0')
    finals = []
    for graphed in (True, False):
        model, data, _ = _golden_model(g, True)
        if not graphed:
            monkeypatch.setattr(type(model), '_optimize_graphed', lambda self, step: False)
        base = [pg['lr'] for pg in model.optimizer_g.param_groups]
        fbs = []
        for it in range(1, 11):
            for pg, b in zip(model.optimizer_g.param_groups, base):
                pg['lr'] = b * it / 10.0                      # a linear warm-up: a different value every step
            model.feed_data(data)
            model.optimize_parameters(it)
            fbs.append(getattr(model, '_tgraph', {}).get('fb'))
        if graphed:   # (the weight scales may settle once during the first steps; from then on ONE capture serves every learning rate)
            assert fbs[-1] is not None and all(f is fbs[-1] for f in fbs[-4:]), [id(f) for f in fbs]
        finals.append({n: p.detach().double().cpu() for n, p in model.get_bare_model(model.net_g).named_parameters()})
    bad = tot = 0
    for n in finals[0]:
        a, b = finals[0][n], finals[1][n]
        bad += int(((a - b).abs() > 2e-5 * float(b.abs().max()) + 1e-7).sum())
        tot += a.numel()
        assert float((a - b).abs().max()) <= 10 * 2.5e-4, n
    # (ten steps of +-lr moves with order-dependent float atomics in the gradients: 6 % of the elements end more than 2e-5 of the
    # tensor's range apart; a learning rate frozen at its capture-time value would move ALL of them apart)
    print(f'graph vs eager under a moving learning rate: {bad} of {tot} elements differ')
    assert bad <= 0.10 * tot, (bad, tot)


def test_prelu_with_a_non_positive_slope_and_outgrown_weight_scales_raise_the_range_flag():
    """the two conditions the fused training kernels cannot handle are reported through the library's range flag (the model
    then re-runs the step under hip.range_free()): a PReLU slope <= 0 (the fused backward recovers x from out / slope) and a
    weight that has outgrown its cached fp16 scale; under range_free() the unfused / range-free forms give torch's gradients"""
    from mrefsr_amd import hip
    from mrefsr_amd.archs import nhwc, nhwc_train
    torch.manual_seed(3)
    conv, prelu = nn.Conv2d(32, 32, 3, 1, 1).cuda(), nn.PReLU(init=-0.2).cuda()
    x = torch.randn(2, 10, 12, 32, device='cuda', requires_grad=True)
    hip.conv_range_tripped()
    nhwc.conv(conv, x, prelu=prelu).sum().backward()
    assert hip.conv_range_tripped()                                   # slope <= 0 seen by the fused backward
    for p in (conv.weight, conv.bias, prelu.weight, x):
        p.grad = None
    with hip.range_free():
        out = nhwc.conv(conv, x, prelu=prelu)
        g = torch.randn_like(out)
        out.backward(g)
    assert not hip.conv_range_tripped()
    xr = x.detach().double().cpu().permute(0, 3, 1, 2).requires_grad_()
    W, B, S = (t.detach().double().cpu().requires_grad_() for t in (conv.weight, conv.bias, prelu.weight))
    want = F.prelu(F.conv2d(xr, W, B, 1, 1), S)
    want.backward(g.double().cpu().permute(0, 3, 1, 2))
    _close(out.permute(0, 3, 1, 2), want)
    _close(prelu.weight.grad, S.grad)
    _close(conv.weight.grad, W.grad)
    _close(x.grad.permute(0, 3, 1, 2), xr.grad)
    # a weight grows 8x after its scale was cached: check_scales() raises the flag without a host synchronisation
    nhwc_train.reset_scales()
    conv2 = nn.Conv2d(32, 32, 3, 1, 1).cuda()
    nhwc.conv(conv2, x.detach().requires_grad_(), slope=0.1).sum().backward()
    nhwc_train.check_scales()
    assert not hip.conv_range_tripped()
    with torch.no_grad():
        conv2.weight.mul_(8.0)
    nhwc_train.check_scales()
    assert hip.conv_range_tripped()
    nhwc_train.reset_scales()


def test_single_reference_network_trains_on_both_engines(monkeypatch):
    """RestorationNet (ref_restoration_arch.py:101-259, the single-reference path): forward and every parameter gradient
    on the channels-last training engine equal the NCHW / MIOpen autograd path"""
    from mrefsr_amd.archs import nhwc_train
    from mrefsr_amd.archs.ref_restoration_arch import RestorationNet
    torch.manual_seed(11)
    net = RestorationNet(ngf=64, n_blocks=2, groups=8).cuda()
    with torch.no_grad():
        for m in net.modules():   # non-trivial offsets / masks (conv_offset_mask is zero-initialised)
            if hasattr(m, 'conv_offset_mask'):
                m.conv_offset_mask.weight.normal_(0, 1e-5)
                m.conv_offset_mask.bias.normal_(0, 0.02)
    b, h, w = 2, 12, 16
    x = torch.rand(b, 3, h, w, device='cuda')
    # sampling positions well inside their bilinear cells (integer shift + 0.37 +- small learned offsets): the two engines' forward
    # results differ by fp32 rounding, and a position within that distance of a cell border would flip its corner set
    pre = {k: torch.randint(-3, 4, (b, 9, s * h, s * w, 2), device='cuda').float() + 0.37 for k, s in (('relu3_1', 1), ('relu2_1', 2), ('relu1_1', 4))}
    feat = {k: torch.randn(b, c, s * h, s * w, device='cuda') for k, c, s in (('relu3_1', 256, 1), ('relu2_1', 128, 2), ('relu1_1', 64, 4))}
    target = torch.rand(b, 3, 4 * h, 4 * w, device='cuda')
    outs, grads = [], []
    for enabled in (True, False):
        monkeypatch.setattr(nhwc_train, 'ENABLED', enabled)
        net.zero_grad(set_to_none=True)
        out = net(x, pre, feat)
        F.l1_loss(out, target).backward()
        outs.append(out.detach())
        grads.append({n: p.grad.detach().clone() for n, p in net.named_parameters()})
    _close(outs[0], outs[1].cpu(), 1e-5)
    worst = {}
    for n in grads[0]:
        a, b_ = grads[0][n].double().cpu(), grads[1][n].double().cpu()
        worst[n] = float((a - b_).abs().max()) / (float(b_.abs().max()) + 1e-30)
    bad = {n: round(v, 6) for n, v in worst.items() if v > 1e-4}
    print('largest relative gradient differences:', sorted(worst.items(), key=lambda kv: -kv[1])[:4])
    assert not bad, bad


@pytest.mark.parametrize('c', [64, 128])
def test_dcn_node_matches_the_nchw_autograd_function(c):
    """_Dcn (channels-last forward kernel + act_bwd + im2col / col2im backward) against ModulatedDeformConvFunction on NCHW
    tensors: same outputs, same gradients for offset, mask, weight and bias"""
    from mrefsr_amd.archs import nhwc_train
    from mrefsr_amd.ops.dcn import modulated_deform_conv
    torch.manual_seed(c)
    b, h, w, dg = 2, 24, 40, 8
    x = torch.randn(b, h, w, c, device='cuda')
    off = (torch.randint(-3, 4, (b, 18 * dg, h, w), device='cuda').float() + 0.37 + 0.05 * torch.randn(b, 18 * dg, h, w, device='cuda'))
    msk = torch.rand(b, 9 * dg, h, w, device='cuda')
    wgt = (torch.randn(c, c, 3, 3, device='cuda') * 0.05)
    bias = torch.randn(c, device='cuda') * 0.1
    g = torch.randn(b, h, w, c, device='cuda')
    res = []
    for nhwc_path in (True, False):
        o, m, wt, bs = (t.clone().requires_grad_() for t in (off, msk, wgt, bias))
        if nhwc_path:
            out = nhwc_train.dcn(x, o, m, wt, bs, dg, 0.1)
            out.backward(g)
            out = out.permute(0, 3, 1, 2)
        else:
            out = modulated_deform_conv(x.permute(0, 3, 1, 2).contiguous(), o, m, wt, bs, 1, 1, 1, 1, dg, 0.1)
            out.backward(g.permute(0, 3, 1, 2).contiguous())
        res.append((out.detach(), o.grad, m.grad, wt.grad, bs.grad))
    for a, b_ in zip(*res):
        _close(a, b_.cpu(), 2e-5)


@pytest.mark.parametrize('shape', [(2, 20, 36, 64, 64), (1, 17, 37, 3, 64), (2, 9, 70, 32, 3), (1, 33, 40, 128, 216), (3, 40, 40, 80, 48)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_wgrad_kernel_matches_fp64(shape):
    """mrefsr_conv_wgrad3x3_f32 (16-bit matrix pipe, transposed LDS staging, shifted gradient copies) against the fp64 weight
    gradient of F.conv2d: channel counts that are not multiples of 64 / 4, widths that are not multiples of 32, tiny gradients"""
    from mrefsr_amd import hip
    n, h, w, ci, co = shape
    torch.manual_seed(ci * 7 + co)
    x = torch.randn(n, ci, h, w)
    g = torch.randn(n, co, h, w) * 3e-7            # the magnitude of an L1-trained network's gradients: far below fp16's range
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, None, 1, 1).backward(g.double())
    ldx, ldg = (ci + 3) // 4 * 4, (co + 3) // 4 * 4
    xd = torch.zeros(n, h, w, ldx, device='cuda')
    xd[..., :ci] = x.permute(0, 2, 3, 1)
    gd = torch.zeros(n, h, w, ldg, device='cuda')
    gd[..., :co] = g.permute(0, 2, 3, 1)
    amax = gd.abs().max().view(1)
    got = hip.conv_wgrad3x3(xd, gd, ci, co, amax)
    _close(got, wt.grad, 2e-5)


def test_residual_block_node_matches_fp64_autograd_and_the_two_node_recording(monkeypatch):
    """x + conv2(relu(conv1(x))) (arch_util.py:45-70) as one node: gradients against fp64 autograd of the literal ops, and against
    the recording with two convolution nodes plus autograd's own add"""
    from mrefsr_amd.archs import arch_util, nhwc, nhwc_train
    torch.manual_seed(11)
    blocks = nn.Sequential(*[arch_util.ResidualBlockNoBN(num_feat=64) for _ in range(3)]).cuda()
    for p in blocks.parameters():
        p.data.normal_(0, 0.05)
    x = torch.randn(4, 64, 24, 40)
    gout = torch.randn(4, 64, 24, 40)
    # fp64 reference
    ps = [p.detach().double().cpu().requires_grad_() for p in blocks.parameters()]
    xd = x.double().requires_grad_()
    y = xd
    for i in range(3):
        w1, b1, w2, b2 = ps[4 * i:4 * i + 4]
        y = y + F.conv2d(F.relu(F.conv2d(y, w1, b1, padding=1)), w2, b2, padding=1)
    (y * gout.double()).sum().backward()
    got = {}
    for fused in ('ResChain', 'ResBlock', ''):   # the trunk as one node (batched weight gradients), one node per block, two per block
        monkeypatch.setattr(nhwc_train, 'RESCHAIN', fused == 'ResChain')
        monkeypatch.setattr(nhwc_train, 'RESBLOCK', fused != '')
        for p in blocks.parameters():
            p.grad = None
        xg = _nhwc(x).cuda().requires_grad_()
        out = nhwc.res_chain(blocks, xg)
        node = type(out.grad_fn).__name__
        assert (fused in node) if fused else ('Res' not in node), node
        (out * _nhwc(gout).cuda()).sum().backward()
        _close(out.permute(0, 3, 1, 2), y)
        _close(xg.grad.permute(0, 3, 1, 2), xd.grad)
        for p, q in zip(blocks.parameters(), ps):
            _close(p.grad, q.grad)
        got[fused] = [out.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in blocks.parameters()]
    # the fused add (inside the epilogue of conv1's input-gradient launch) and autograd's own add are the same single fp32 addition
    for k in ('ResChain', 'ResBlock'):
        for a, b in zip(got[k], got['']):
            _close(a, b.cpu(), 1e-6)


def test_multi_tensor_weight_pack_equals_single_packs_and_follows_the_parameters():
    """mrefsr_conv_pack_weights_multi_f32 (one launch for every packed copy of a step) writes the bytes of
    mrefsr_conv_pack_weight_view_f32 per tensor; the training engine's cache re-packs after an in-place parameter update
    (``_version``) and at begin_step() even when the update went through .data"""
    from mrefsr_amd import hip
    from mrefsr_amd.archs import nhwc_train
    torch.manual_seed(3)
    ws = [torch.randn(64, 64, 3, 3, device='cuda') * 0.05, torch.randn(216, 64, 3, 3, device='cuda') * 0.02,
          torch.randn(64, 192, 1, 1, device='cuda') * 0.1, torch.randn(3, 32, 3, 3, device='cuda'), torch.randn(128, 256, 3, 3, device='cuda') * 0.01]
    plans, singles = [], []
    for i, w in enumerate(ws):
        for terms, dgrad, sl in ((16, False, None), (16, True, None), (6, False, None), (6, True, (0, w.shape[1] // 2 or 1))):
            scale = 2.0 ** 10 if terms == 16 else 1.0
            plans.append(hip.conv_pack_plan(w, sl, terms, dgrad, scale))
            singles.append(hip.conv_pack_view(w, sl, terms, dgrad=dgrad, wscale=scale))
    for pw, _ in plans:
        pw.data.fill_(0x5a)
    table = hip.conv_pack_table([j for _, j in plans], ws[0].device)
    hip.conv_pack_multi(table, len(plans))
    for (pw, _), ref in zip(plans, singles):
        assert torch.equal(pw.data, ref.data)
    assert not hip.conv_range_tripped()
    # a weight that has outgrown the power-of-two scale of its job (|w * scale| > 65000) raises the range flag from the device
    _, job = hip.conv_pack_plan(ws[3], None, 16, False, 2.0 ** 16)
    hip.conv_pack_multi(hip.conv_pack_table([job], ws[3].device), 1)
    assert hip.conv_range_tripped()
    # the cache of the training engine
    nhwc_train.reset_packs()
    w = torch.nn.Parameter(ws[0].clone())
    a = nhwc_train._packed(w, None, 6)
    first = a.data.clone()
    assert nhwc_train._packed(w, None, 6) is a
    with torch.no_grad():
        w.mul_(2.0)                        # in-place update: version bump -> refreshed at the next lookup, same buffer
    b = nhwc_train._packed(w, None, 6)
    assert b is a and torch.equal(a.data, hip.conv_pack_view(w.detach(), None, 6).data) and not torch.equal(a.data, first)
    w.data.mul_(0.5)                       # no version bump: the model's begin_step() refreshes unconditionally
    nhwc_train.begin_step()
    assert torch.equal(nhwc_train._packed(w, None, 6).data, first)
    nhwc_train.reset_packs()


def test_zero_pool_slices_are_zero_disjoint_and_never_rezeroed():
    from mrefsr_amd import hip
    dev = torch.device('cuda', 0)
    a = hip.zeros_f32(dev, 66)
    a.fill_(1.0)
    b = hip.zeros_f32(dev, 66)
    assert float(b.abs().sum()) == 0.0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
    assert b.data_ptr() >= a.data_ptr() + 66 * 4 or b.data_ptr() + 66 * 4 <= a.data_ptr()
    big = hip.zeros_f32(dev, (1 << 18) + 5)    # larger than a chunk: its own allocation
    assert big.numel() == (1 << 18) + 5 and float(big.abs().sum()) == 0.0
    assert float(a.sum()) == 66.0


@pytest.mark.parametrize('geom', [(4, 40, 40, 64, 32), (2, 23, 37, 64, 7), (1, 80, 80, 128, 3), (3, 16, 70, 48, 1)],
                         ids=lambda g: 'x'.join(map(str, g)))
def test_batched_wgrad_equals_the_single_launches(geom):
    """mrefsr_conv_wgrad3x3_batch_f32: the jobs of a batch against one mrefsr_conv_wgrad3x3_f32 call each (same products, another
    split of the rows over blocks: fp32 summation order differs) and against fp64"""
    from mrefsr_amd import hip
    n, h, w, c, nj = geom
    torch.manual_seed(21)
    xs = [torch.randn(n, h, w, c, device='cuda') for _ in range(nj)]
    gs = [torch.randn(n, h, w, c, device='cuda') * 10.0 ** (-3 - j % 5) for j in range(nj)]
    am = [g.abs().max().reshape(1) for g in gs]
    dw = hip.conv_wgrad3x3_batch(xs, gs, c, c, am)
    assert tuple(dw.shape) == (nj, c, c, 3, 3)
    for j in (0, nj // 2, nj - 1):
        one = hip.conv_wgrad3x3(xs[j], gs[j], c, c, am[j])
        _close(dw[j], one.cpu(), 2e-6)
        want = torch.nn.grad.conv2d_weight(xs[j].permute(0, 3, 1, 2).double().cpu(), (c, c, 3, 3), gs[j].permute(0, 3, 1, 2).double().cpu(), padding=1)
        _close(dw[j], want, 2e-5)
    hip.check_conv_range()


@pytest.mark.parametrize('shape', [(3, 8, 13, 17), (2, 8, 40, 40), (1, 4, 5, 7), (20, 8, 80, 80)], ids=lambda s: 'x'.join(map(str, s)))
def test_dynagg_prep_bwd_channels_last_equals_the_planar_kernel_and_its_reductions(shape):
    """mrefsr_dynagg_prep_bwd_nhwc_f32: the planar kernel's values (bit for bit) in [B,H,W,27dg] order, the per-channel sums (bias
    gradient of conv_offset_mask) and max |g_om| of the same pass"""
    from mrefsr_amd import hip
    b, dg, h, w = shape
    torch.manual_seed(4)
    g_off = torch.randn(b, 18 * dg, h, w, device='cuda') * 1e-4
    g_m = torch.randn(b, 9 * dg, h, w, device='cuda') * 1e-3
    mask = torch.rand(b, 9 * dg, h, w, device='cuda')
    want = hip.dynagg_prep_bwd(g_off, g_m, mask, dg).permute(0, 2, 3, 1).contiguous()
    got, bias, amax = hip.dynagg_prep_bwd_nhwc(g_off, g_m, mask, dg)
    assert torch.equal(got, want)
    _close(bias, want.double().sum((0, 1, 2)).cpu(), 1e-5)
    assert float(amax) == float(want.abs().max())
    got2, bias2, _ = hip.dynagg_prep_bwd_nhwc(g_off, g_m, mask, dg, want_bias=False)
    assert bias2 is None and torch.equal(got2, want)


@pytest.mark.parametrize('geom', [(4, 40, 40, 64), (2, 23, 37, 64), (4, 160, 160, 64), (1, 16, 70, 128)], ids=lambda g: 'x'.join(map(str, g)))
def test_fused_input_gradient_convolution_equals_convolution_plus_act_bwd(geom):
    """mrefsr_conv_nhwc_bwd_f32 (ReLU mask / skip add and the per-channel sums + max |out| in the epilogue of the input-gradient
    convolution) against mrefsr_conv_nhwc_scaled_f32 followed by mrefsr_act_bwd_nhwc_f32: the same output bits, sums to fp32
    summation order, the same maximum"""
    from mrefsr_amd import hip
    n, h, w, c = geom
    torch.manual_seed(17)
    g = torch.randn(n, h, w, c, device='cuda') * 1e-5
    t = torch.randn(n, h, w, c, device='cuda')          # the forward activation whose sign is the mask
    skip = torch.randn(n, h, w, c, device='cuda') * 1e-5
    wt = torch.randn(c, c, 3, 3, device='cuda') * 0.05
    amax = g.abs().max().reshape(1)
    pk = hip.conv_pack_view(wt, None, 16, dgrad=True, wscale=2.0 ** 12)
    # ReLU mask + statistics
    ref = hip.conv_nhwc(g, pk, None, c, 3, in_amax=amax)
    ref_pre, ref_b, _, ref_a = hip.act_bwd_nhwc(ref, t, 1, 0.0, want_bias=True, want_amax=True)
    out, sb, sa = hip.conv_nhwc_bwd(g, pk, c, 3, residual=t, residual_is_mask=True, in_amax=amax)
    assert torch.equal(out, ref_pre)
    _close(sb, ref_pre.double().sum((0, 1, 2)).cpu(), 2e-6)
    _close(sb, ref_b.cpu(), 2e-6)
    assert float(sa) == float(ref_a) == float(ref_pre.abs().max())
    # skip add + statistics, and without statistics
    ref2 = hip.conv_nhwc(g, pk, None, c, 3, residual=skip, in_amax=amax)
    out2, sb2, sa2 = hip.conv_nhwc_bwd(g, pk, c, 3, residual=skip, in_amax=amax)
    assert torch.equal(out2, ref2)
    _close(sb2, ref2.double().sum((0, 1, 2)).cpu(), 2e-6)
    assert float(sa2) == float(ref2.abs().max())
    out3, sb3, sa3 = hip.conv_nhwc_bwd(g, pk, c, 3, residual=skip, in_amax=amax, want_stats=False)
    assert sb3 is None and sa3 is None and torch.equal(out3, ref2)
    hip.check_conv_range()
