"""GPU: the ``basicsr.ops`` surface beyond the MRefSR path -- the StyleGAN2 call pattern of upfirdn2d / fused_act
(forward, backward, double backward), their dtypes, the deform_conv_ext stand-in, the range-flag recovery of the model.
Oracles: upfirdn2d_native outputs generated from the reference (metrics_ops.npz) pin a torch restatement of the operator,
which autograd then differentiates twice; the fused activation is its 3-line formula (fused_bias_act_kernel.cu:29-47: the
reference has no CPU path for it)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# the native entry takes alpha / scale as C floats (fused_bias_act.cpp:14-16: `float alpha, float scale`), also for double tensors
SLOPE, GAIN = float(np.float32(0.2)), float(np.float32(2 ** 0.5))


def upfirdn2d_torch(x, k, up, down, pad):
    """the operator from its definition (upfirdn2d.py:162-192) with differentiable torch ops, any dtype: x (N,C,H,W)"""
    n, c, h, w = x.shape
    z = x.new_zeros(n, c, h, up, w, up)
    z[:, :, :, 0, :, 0] = x
    z = z.view(n, c, h * up, w * up)
    p0, p1 = pad
    z = F.pad(z, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    z = z[:, :, max(-p0, 0):z.shape[2] - max(-p1, 0), max(-p0, 0):z.shape[3] - max(-p1, 0)]
    wgt = torch.flip(k, [0, 1]).to(x.dtype)[None, None].expand(c, 1, *k.shape)
    return F.conv2d(z, wgt, groups=c)[:, :, ::down, ::down]


def test_torch_restatement_of_upfirdn2d_equals_the_reference_native(golden):
    g = golden('metrics_ops')
    for i, (u, d, p0, p1, ks) in enumerate(g['up_cases']):
        x, k = torch.from_numpy(g[f'up_x{i}']), torch.from_numpy(g[f'up_k{i}'])
        np.testing.assert_allclose(upfirdn2d_torch(x, k, int(u), int(d), (int(p0), int(p1))).numpy(), g[f'up_out{i}'], rtol=0, atol=2e-6)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-6), (torch.float64, 1e-12), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_upfirdn2d_dtypes_and_tiled_kernel(dtype, tol):
    """every dtype of the reference's dispatch (+ bf16), sizes that take the tiled LDS kernel with ragged tile edges, negative pads"""
    from mrefsr_amd.ops.upfirdn2d import upfirdn2d
    gen = torch.Generator().manual_seed(0)
    for (n, c, h, w, ks, u, d, pad) in [(2, 3, 37, 71, 4, 2, 1, (2, 1)), (1, 2, 64, 130, 4, 1, 2, (1, 1)), (1, 1, 20, 20, 3, 1, 1, (-1, 2)),
                                         (2, 2, 45, 77, 4, 1, 1, (2, 1)), (1, 2, 31, 66, 4, 1, 1, (-1, 3)), (1, 1, 19, 23, 4, 2, 1, (3, 0)),
                                         (2, 2, 33, 65, 5, 3, 2, (2, 2)), (1, 1, 5, 7, 2, 1, 1, (0, 0)),
                                         # up x2 quad path: odd output widths (rows lose their 8-byte alignment), tiles wider than one block, both pad parities
                                         (1, 2, 18, 35, 4, 2, 1, (2, 2)), (1, 1, 40, 150, 4, 2, 1, (1, 2)), (2, 1, 70, 140, 4, 2, 1, (2, 1)), (1, 1, 33, 129, 4, 2, 1, (3, 2))]:
        x = torch.randn(n, c, h, w, generator=gen, dtype=torch.float64)
        k = torch.rand(ks, ks, generator=gen, dtype=torch.float64)
        want = upfirdn2d_torch(x.to(dtype).double(), k.to(dtype).double(), u, d, pad)
        got = upfirdn2d(x.to(dtype).cuda(), k.to(dtype).cuda(), up=u, down=d, pad=pad)
        assert got.dtype == dtype and tuple(got.shape) == tuple(want.shape)
        scale = float(want.abs().max())
        assert (got.double().cpu() - want).abs().max().item() <= tol * max(scale, 1.0), (dtype, n, c, h, w)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-6), (torch.float64, 1e-14), (torch.float16, 2e-3)])
def test_fused_leaky_relu_dtypes(dtype, tol):
    from mrefsr_amd.ops.fused_act import fused_leaky_relu
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(3, 6, 5, 7, generator=gen, dtype=torch.float64).to(dtype)
    b = torch.randn(6, generator=gen, dtype=torch.float64).to(dtype)
    want = F.leaky_relu(x.double() + b.double().view(1, -1, 1, 1), SLOPE) * GAIN
    got = fused_leaky_relu(x.cuda(), b.cuda())
    assert got.dtype == dtype
    assert (got.double().cpu() - want).abs().max().item() <= tol * float(want.abs().max())


def test_stylegan2_pattern_forward_backward_double_backward():
    """blur -> upsample -> bias + leaky ReLU + gain (stylegan2_arch.py:65-172), through mrefsr_amd.archs.stylegan2_ops, against
    the torch restatement: outputs, first derivatives w.r.t. input and bias, and a gradient-penalty style second
    derivative (d/dx of |d out / d x|^2 -- the R1 / path-length regulariser pattern that needs the double backward of both
    operators)"""
    from mrefsr_amd.archs.stylegan2_ops import EqualLinear, UpFirDnDownsample, UpFirDnSmooth, UpFirDnUpsample, make_resample_kernel
    from mrefsr_amd.ops.fused_act import FusedLeakyReLU
    gen = torch.Generator().manual_seed(2)
    x0 = torch.randn(2, 4, 10, 12, generator=gen, dtype=torch.float64)
    b0 = 0.3 * torch.randn(4, generator=gen, dtype=torch.float64)
    k1 = [1, 3, 3, 1]
    smooth, upsm, down = UpFirDnSmooth(k1, upsample_factor=2, kernel_size=3), UpFirDnUpsample(k1, 2), UpFirDnDownsample(k1, 2)
    act = FusedLeakyReLU(4).cuda().double()
    with torch.no_grad():
        act.bias.copy_(b0)

    def ours(x):
        return down(act(upsm(smooth(x))))

    def theirs(x, b):
        kk = make_resample_kernel(k1).double()
        y = upfirdn2d_torch(x, kk * 4, 1, 1, smooth.pad)
        y = upfirdn2d_torch(y, kk * 4, 2, 1, upsm.pad)
        y = F.leaky_relu(y + b.view(1, -1, 1, 1), SLOPE) * GAIN
        return upfirdn2d_torch(y, kk, 1, 2, down.pad)

    assert smooth.pad == (1, 1) and upsm.pad == (2, 1) and down.pad == (1, 1)      # the reference's values for [1,3,3,1]
    xg = x0.clone().cuda().requires_grad_(True)
    xc, bc = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    yo, yt = ours(xg), theirs(xc, bc)
    assert (yo.cpu() - yt).abs().max().item() <= 1e-12
    w = torch.randn(yt.shape, generator=gen, dtype=torch.float64)
    (gxo, gbo) = torch.autograd.grad((yo * w.cuda()).sum(), [xg, act.bias], create_graph=True)
    (gxt, gbt) = torch.autograd.grad((yt * w).sum(), [xc, bc], create_graph=True)
    assert (gxo.cpu() - gxt).abs().max().item() <= 1e-12 and (gbo.cpu() - gbt).abs().max().item() <= 1e-11
    # second order.  The block is piecewise linear, so curvature has to come from the loss: L = sum(w * y^2) gives
    # grad_x L = J^T (2 w y), and differentiating <grad_x L, v> + sum(grad_b L) again runs J v forward through the
    # double-backward paths of both operators (the R1 / path-length regularisers of StyleGAN2 do exactly this)
    v = torch.randn(gxt.shape, generator=gen, dtype=torch.float64)
    (hxo, hbo) = torch.autograd.grad((ours(xg) ** 2 * w.cuda()).sum(), [xg, act.bias], create_graph=True)
    (hxt, hbt) = torch.autograd.grad((theirs(xc, bc) ** 2 * w).sum(), [xc, bc], create_graph=True)
    assert (hxo.cpu() - hxt).abs().max().item() <= 1e-11
    (ggo, gbo2) = torch.autograd.grad((hxo * v.cuda()).sum() + hbo.sum(), [xg, act.bias])
    (ggt, gbt2) = torch.autograd.grad((hxt * v).sum() + hbt.sum(), [xc, bc])
    assert (ggo.cpu() - ggt).abs().max().item() <= 1e-10 and (gbo2.cpu() - gbt2).abs().max().item() <= 1e-9
    # gradient w.r.t. the cotangent (the other half of "double backward"): d/dw of <grad_x(w), v>
    wq = w.clone().cuda().requires_grad_(True)
    wc = w.clone().requires_grad_(True)
    (g1,) = torch.autograd.grad((ours(xg) * wq).sum(), [xg], create_graph=True)
    (g2,) = torch.autograd.grad((theirs(xc, bc) * wc).sum(), [xc], create_graph=True)
    (d1,) = torch.autograd.grad((g1 * v.cuda()).sum(), [wq])
    (d2,) = torch.autograd.grad((g2 * v).sum(), [wc])
    assert (d1.cpu() - d2).abs().max().item() <= 1e-11
    lin = EqualLinear(8, 5, bias=True, bias_init_val=0.1, lr_mul=0.5, activation='fused_lrelu').cuda()
    z = torch.randn(3, 8, generator=gen).cuda()
    want = F.leaky_relu(F.linear(z, lin.weight * lin.scale) + lin.bias * 0.5, 0.2) * 2 ** 0.5
    assert (lin(z) - want).abs().max().item() <= 1e-6


def test_deform_conv_ext_stand_in_matches_the_module_path():
    """the five entry points of deform_conv_ext.cpp:150-164, called the way the reference's deform_conv.py calls them
    (caller-allocated outputs and zero-initialised gradients), against the autograd Functions of mrefsr_amd.ops.dcn"""
    from mrefsr_amd.ops.dcn import deform_conv, modulated_deform_conv
    from mrefsr_amd.ops.dcn import deform_conv_ext as ext
    gen = torch.Generator().manual_seed(3)
    b, c, h, w, co, dg = 2, 16, 9, 11, 8, 4
    x = torch.randn(b, c, h, w, generator=gen).cuda().requires_grad_(True)
    off = (2 * torch.randn(b, dg * 18, h, w, generator=gen)).cuda().requires_grad_(True)
    msk = torch.rand(b, dg * 9, h, w, generator=gen).cuda().requires_grad_(True)
    wt = (0.1 * torch.randn(co, c, 3, 3, generator=gen)).cuda().requires_grad_(True)
    bias = torch.randn(co, generator=gen).cuda().requires_grad_(True)
    go = torch.randn(b, co, h, w, generator=gen).cuda()
    # DCNv2
    want = modulated_deform_conv(x, off, msk, wt, bias, 1, 1, 1, 1, dg)
    want.backward(go)
    out = x.new_empty(b, co, h, w)
    ext.modulated_deform_conv_forward(x.detach(), wt.detach(), bias.detach(), x.new_empty(0), off.detach(), msk.detach(), out, x.new_empty(0),
                                      3, 3, 1, 1, 1, 1, 1, 1, 1, dg, True)
    assert torch.equal(out, want.detach())
    gi, gw, gb, goff, gm = (torch.zeros_like(t) for t in (x, wt, bias, off, msk))
    ext.modulated_deform_conv_backward(x.detach(), wt.detach(), bias.detach(), x.new_empty(0), off.detach(), msk.detach(), x.new_empty(0), gi, gw,
                                       gb, goff, gm, go, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, True)
    for name, got, ref in (('input', gi, x.grad), ('weight', gw, wt.grad), ('bias', gb, bias.grad), ('offset', goff, off.grad), ('mask', gm, msk.grad)):
        assert (got - ref).abs().max().item() <= 2e-4 * max(float(ref.abs().max()), 1.0), name   # float atomics: order differs run to run
    # DCNv1 ((W, H) argument order)
    for t in (x, off, wt):
        t.grad = None
    want1 = deform_conv(x, off, wt, 1, 1, 1, 1, dg)
    want1.backward(go)
    out1 = x.new_empty(b, co, h, w)
    assert ext.deform_conv_forward(x.detach(), wt.detach(), off.detach(), out1, x.new_empty(0), x.new_empty(0), 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, 2) == 1
    assert torch.equal(out1, want1.detach())
    gi, goff, gw = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(wt)
    ext.deform_conv_backward_input(x.detach(), off.detach(), go, gi, goff, wt.detach(), x.new_empty(0), 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, 2)
    ext.deform_conv_backward_parameters(x.detach(), off.detach(), go, gw, x.new_empty(0), x.new_empty(0), 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, 1.0, 2)
    for name, got, ref in (('input', gi, x.grad), ('offset', goff, off.grad), ('weight', gw, wt.grad)):
        assert (got - ref).abs().max().item() <= 2e-4 * max(float(ref.abs().max()), 1.0), name
    with pytest.raises(RuntimeError):
        ext.modulated_deform_conv_forward(x.detach().cpu(), wt.detach(), bias.detach(), None, off.detach(), msk.detach(), out, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, True)


def test_range_flag_recovery_reruns_the_batch(golden, tmp_path):
    """an activation beyond the fp16 range of the split kernels: test() re-runs the batch on the range-free kernels (counted,
    logged, no exception) and the result equals what the range-free mode gives directly; validation writes its images"""
    from test_archs_gpu import _model
    from mrefsr_amd import hip
    g = golden('e2e')
    model, data = _model(g, False)
    model.feed_data(data)
    model.test()
    assert model.range_fallbacks == 0
    base = model.output.clone()
    net = model.get_bare_model(model.net_g)
    with torch.no_grad():   # blow one trunk activation past 65504 and undo it in the next layer's weights: same function, huge intermediate
        blk = net.content_extractor.body[3]
        blk.conv1.weight.mul_(1e6)
        blk.conv1.bias.mul_(1e6)
        blk.conv2.weight.mul_(1e-6)
    from mrefsr_amd.archs import nhwc
    saved = nhwc.WINO_INSCALE
    try:
        # (1) without the Winograd input scale the 1e6-fold activation meets the fp16 split unscaled: the flag fires, the batch is re-run
        nhwc.WINO_INSCALE = False
        model.test()
        assert model.range_fallbacks == 1 and torch.isfinite(model.output).all()
        with hip.range_free():
            model.test()
        direct = model.output.clone()
        assert model.range_fallbacks == 1                      # the range-free kernels do not raise the flag
        model.test()
        assert model.range_fallbacks == 2 and torch.equal(model.output, direct)   # the default kernels again: the same trip, the same re-run
        assert (direct - base).abs().max().item() <= 5e-3      # same function up to the rounding of the 1e6 / 1e-6 detour
        # (2) with it (the default, round 6: every layer is handed the maximum its producer measured for this very batch) the next
        # layer scales the huge tensor into the fp16 range: no trip, no re-run, the same function
        nhwc.WINO_INSCALE = True
        model.test()
        assert model.range_fallbacks == 2 and torch.isfinite(model.output).all()
        assert (model.output - direct).abs().max().item() <= 5e-3
    finally:
        nhwc.WINO_INSCALE = saved
    model.check_numeric_range()                            # flag left clear


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-11), (torch.float16, 3e-2)])
def test_dcn_other_dtypes_of_the_reference_dispatch(dtype, tol):
    """f64 and f16 instantiations (AT_DISPATCH_FLOATING_TYPES_AND_HALF, deform_conv_cuda_kernel.cu:781,813,846) through the
    basicsr.ops.dcn functions: forward and every gradient against the torch restatement of the vendored spec in fp64
    (oracle/dcn_torch.py), DCNv2 with stride / groups and DCNv1"""
    from mrefsr_amd.ops.dcn import deform_conv, modulated_deform_conv
    from oracle import dcn_torch
    gen = torch.Generator().manual_seed(9)
    b, c, h, w, co, dg, groups, stride = 2, 8, 9, 10, 6, 2, 2, 2
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    x0 = torch.randn(b, c, h, w, generator=gen, dtype=torch.float64)
    off0 = 1.5 * torch.randn(b, dg * 18, ho, wo, generator=gen, dtype=torch.float64)
    m0 = torch.rand(b, dg * 9, ho, wo, generator=gen, dtype=torch.float64)
    w0 = 0.3 * torch.randn(co, c // groups, 3, 3, generator=gen, dtype=torch.float64)
    b0 = torch.randn(co, generator=gen, dtype=torch.float64)
    go = torch.randn(b, co, ho, wo, generator=gen, dtype=torch.float64)
    q = lambda t: t.detach().to(dtype).double().clone()      # noqa: E731  the values the kernel actually sees
    ref_in = [q(t).requires_grad_(True) for t in (x0, off0, m0, w0, b0)]
    want = dcn_torch.modulated_deform_conv2d(*ref_in, stride, 1, 1, groups, dg)
    want.backward(go)
    got_in = [t.detach().to(dtype).cuda().requires_grad_(True) for t in (x0, off0, m0, w0, b0)]
    got = modulated_deform_conv(*got_in, stride, 1, 1, groups, dg)
    assert got.dtype == dtype
    got.backward(go.to(dtype).cuda())
    scale = float(want.detach().abs().max())
    assert (got.detach().double().cpu() - want.detach()).abs().max().item() <= tol * scale
    for name, a, r in zip(('input', 'offset', 'mask', 'weight', 'bias'), got_in, ref_in):
        assert a.grad.dtype == dtype
        assert (a.grad.double().cpu() - r.grad).abs().max().item() <= 4 * tol * max(float(r.grad.abs().max()), 1.0), name
    # DCNv1
    xr, orf, wr = (q(t).requires_grad_(True) for t in (x0, off0, w0))
    want1 = dcn_torch.modulated_deform_conv2d(xr, orf, torch.ones_like(m0), wr, None, stride, 1, 1, groups, dg)
    xg, og, wg = (t.detach().to(dtype).cuda().requires_grad_(True) for t in (x0, off0, w0))
    got1 = deform_conv(xg, og, wg, stride, 1, 1, groups, dg)
    assert (got1.detach().double().cpu() - want1.detach()).abs().max().item() <= tol * float(want1.detach().abs().max())
