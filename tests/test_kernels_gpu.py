"""GPU parity tests: every HIP kernel, called through the C ABI (mrefsr_amd.hip -> ctypes ->
libmrefsr_hip.so), against the CPU oracle and the reference-generated golden vectors.

Bars (north_star): correlation indices bit-exact; integer / index work bit-exact; floating point
within the tolerance written at each assert.
"""
import numpy as np
import pytest
import torch

import cases
import synth
from oracle import c_api as orc

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


@pytest.fixture(scope='module')
def hip():
    from mrefsr_amd import hip as h
    return h


def unsplit(y, c):
    """[N, HW, Cp] split layout -> [N, C, HW]"""
    n, hw, cp = y.shape
    half = cp // 2
    out = np.empty((n, cp, hw), np.float32)
    out[:, 0::2] = y[:, :, :half].transpose(0, 2, 1)
    out[:, 1::2] = y[:, :, half:].transpose(0, 2, 1)
    assert (out[:, c:] == 0).all()
    return out[:, :c]


# --------------------------------------------------------------------------------- correlation
@pytest.mark.parametrize('c,h,w', [(256, 12, 14), (64, 16, 16), (100, 9, 21), (256, 40, 40)])
def test_pixnorm_and_patch_norm_bit_exact(hip, c, h, w):
    x = synth.randn(f'pn/{c}x{h}x{w}', (2, c, h, w)) * 3
    x[1, :, 0, 0] = 0  # zero pixel -> the 1e-12 clamp
    y, n2 = hip.pixnorm(dev(x))
    ne, inv = hip.patch_norm(n2)
    for b in range(2):
        yo, n2o = orc.pixnorm(x[b])
        np.testing.assert_array_equal(unsplit(y.cpu().numpy(), c)[b].reshape(c, h, w), yo)  # IEEE div / sqrt
        np.testing.assert_array_equal(n2[b].cpu().numpy(), n2o)
        neo, invo = orc.patch_norm(n2o)
        np.testing.assert_array_equal(ne[b].cpu().numpy(), neo)
        np.testing.assert_array_equal(inv[b].cpu().numpy(), invo)
    # layout-only mode
    y0, n20 = hip.pixnorm(dev(x), normalize=False)
    np.testing.assert_array_equal(unsplit(y0.cpu().numpy(), c).reshape(2, c, h, w), x)
    np.testing.assert_array_equal(n20[0].cpu().numpy(), orc.sumsq(x[0]))


def _gpu_fmi(hip, fin, fref, prefilter=False):
    """prefilter: False = exact kernel, True = bf16 two-term pre-filter, 'fp16' = fp16 single-plane pre-filter with the
    worst-case window, 'fp16w' = the same with the data-dependent window of hip.prefilter_window (the path's default)
    (256-channel maps; other channel counts use the bf16 operand, as the path does)"""
    f16 = prefilter in ('fp16', 'fp16w') and hip.padded_channels(fin.shape[0]) == 256
    split = 'fp16' if f16 else 'bf16'
    tau = None
    if f16 and prefilter == 'fp16w':
        yi, n2i, bi, d2i = hip.pixnorm(dev(fin[None]), want_bf16_split=True, split=split, want_err=True)
        yr, n2r, br, d2r = hip.pixnorm(dev(fref[None]), want_bf16_split=True, split=split, want_err=True)
    else:
        yi, n2i, bi = hip.pixnorm(dev(fin[None]), want_bf16_split=True, split=split)
        yr, n2r, br = hip.pixnorm(dev(fref[None]), want_bf16_split=True, split=split)
    nei, _ = hip.patch_norm(n2i)
    _, invr = hip.patch_norm(n2r)
    if f16 and prefilter == 'fp16w':
        tau = hip.prefilter_window(nei, invr, d2i, d2r)
    h, w = fin.shape[1:]
    if prefilter:
        idx, val = hip.corr_top1(yi, yr, invr, nei, h, w, ybf_in=bi, ybf_ref=br, tau=tau)
    else:
        idx, val = hip.corr_top1(yi, yr, invr, nei, h, w)
    return idx[0].cpu().numpy(), val[0].cpu().numpy()


@pytest.mark.parametrize('kind', ['gauss', 'heavy_tail', 'smooth', 'near_duplicates', 'sparse'])
def test_fp16_window_prefilter_equals_exact_kernel_on_hostile_statistics(hip, kind):
    """the default path (fp16 operand + data-dependent window) against the exact single-pass kernel, bit for bit, on
    feature statistics chosen to stress the window: heavy tails, smooth maps full of near-ties, near-duplicate
    reference pixels, sparse (post-ReLU-like) channels"""
    rng = np.random.default_rng({'gauss': 1, 'heavy_tail': 2, 'smooth': 3, 'near_duplicates': 4, 'sparse': 5}[kind])
    b, k, c, h, w = 2, 2, 256, 45, 52
    fin = rng.standard_normal((b, c, h, w)).astype(np.float32)
    fref = rng.standard_normal((k * b, c, h, w)).astype(np.float32)
    if kind == 'heavy_tail':
        fin, fref = fin ** 3 * 5, fref ** 3 * 5
    elif kind == 'smooth':
        import scipy.ndimage as ndi
        fin = ndi.gaussian_filter(fin, (0, 0, 4, 4)).astype(np.float32)
        fref = ndi.gaussian_filter(fref, (0, 0, 4, 4)).astype(np.float32)
    elif kind == 'near_duplicates':
        fref = (np.repeat(fref[:, :, :1, :], h, axis=2) + 1e-4 * rng.standard_normal(fref.shape)).astype(np.float32)
    elif kind == 'sparse':
        fin, fref = np.maximum(fin - 1.0, 0), np.maximum(fref - 1.0, 0)
    fin, fref = dev(np.ascontiguousarray(fin, np.float32)), dev(np.ascontiguousarray(fref, np.float32))
    yi, n2i, bi, d2i = hip.pixnorm(fin, want_bf16_split=True, split='fp16', want_err=True)
    yr, n2r, br, d2r = hip.pixnorm(fref, want_bf16_split=True, split='fp16', want_err=True)
    nei, _ = hip.patch_norm(n2i)
    _, invr = hip.patch_norm(n2r)
    tau = hip.prefilter_window(nei, invr, d2i, d2r)
    idx0, val0 = hip.corr_top1(yi, yr, invr, nei, h, w)
    idx1, val1 = hip.corr_top1(yi, yr, invr, nei, h, w, ybf_in=bi, ybf_ref=br, tau=tau)
    assert torch.equal(idx0, idx1) and torch.equal(val0, val1)


def test_prefilter_window_bounds_the_measured_error(hip):
    """hip.prefilter_window against fp64: for random queries and references |fp16 score - exact score| stays below
    tau / 2, and the window is the tight one (about half of the worst-case 2.02 * 1.1e-3 * nrm)"""
    c, h, w = 256, 20, 23
    fin = synth.randn('tau/in', (c, h, w))
    fref = (0.6 * np.roll(fin, (3, -2), axis=(1, 2)) + 0.4 * synth.randn('tau/ref', (c, h, w))).astype(np.float32)
    yi, n2i, bi, d2i = hip.pixnorm(dev(fin[None]), want_bf16_split=True, split='fp16', want_err=True)
    yr, n2r, br, d2r = hip.pixnorm(dev(fref[None]), want_bf16_split=True, split='fp16', want_err=True)
    nei, _ = hip.patch_norm(n2i)
    _, invr = hip.patch_norm(n2r)
    tau = hip.prefilter_window(nei, invr, d2i, d2r)[0].double().cpu()
    ya = torch.nn.functional.normalize(torch.from_numpy(fin).double(), dim=0)
    yb = torch.nn.functional.normalize(torch.from_numpy(fref).double(), dim=0)
    # d2 is what it says: squared norm of y - fp16(y)
    err = (ya.float() - ya.float().half().float()).double()
    np.testing.assert_allclose(d2i[0].cpu().numpy(), (err ** 2).sum(0).numpy(), rtol=2e-2, atol=1e-12)
    ha, hb = ya.float().half().double(), yb.float().half().double()
    unf = lambda t: torch.nn.functional.unfold(t[None], 3)[0]           # [c*9, P]
    inv = invr[0].double().cpu().flatten()
    exact = (unf(yb).T @ unf(ya)) * inv[:, None]                         # [P_ref, P_in]
    approx = (unf(hb).T @ unf(ha)) * inv[:, None]
    worst = (exact - approx).abs().amax(0)                               # per query
    assert (worst <= 0.5 * tau.flatten()).all()
    rel = (tau.flatten() / nei[0].double().cpu().flatten())
    assert 6e-4 < rel.mean() < 1.4e-3                                    # worst-case window: 2.2e-3


def test_bf16_split_is_exact_two_term_expansion(hip):
    x = synth.randn('split/x', (1, 100, 7, 9)) * 2
    y, _, ybf = hip.pixnorm(dev(x), want_bf16_split=True)
    yn = unsplit(y.cpu().numpy(), 100)[0]                    # [C, HW]
    hi = ybf[0, :, 0, :].float().cpu().numpy().T              # [Cp, HW]
    lo = ybf[0, :, 1, :].float().cpu().numpy().T
    assert (hi[100:] == 0).all() and (lo[100:] == 0).all()
    err = np.abs(yn - (hi[:100] + lo[:100]))
    assert (err <= np.abs(yn) * 2.0 ** -16).all()             # |y - hi - lo| <= 2^-18 |y| in theory
    assert (np.abs(yn - hi[:100]) <= np.abs(yn) * 2.0 ** -8).all()


@pytest.mark.parametrize('prefilter', [False, True, 'fp16', 'fp16w'])
def test_corr_top1_bit_exact_vs_oracle_and_reference(hip, golden, prefilter):
    """both device paths -- the exact fp32-MFMA kernel and the bf16x3 pre-filter + exact re-scoring
    (+ brute force on candidate overflow: the 'ties' case) -- return the oracle's bits"""
    g = golden('corr_fmi')
    for name, fin, fref in cases.corr_cases():
        assert str(g[name + '/chk']) == synth.checksum(fin, fref)
        idx, val = _gpu_fmi(hip, fin, fref, prefilter)
        oidx, oval = orc.feature_match_index(fin, fref)
        assert idx.dtype == np.int64
        np.testing.assert_array_equal(idx, oidx, err_msg=f'{name}: HIP vs oracle indices')
        np.testing.assert_array_equal(val, oval, err_msg=f'{name}: HIP vs oracle values (bitwise)')
        np.testing.assert_array_equal(idx, g[name + '/idx'], err_msg=f'{name}: HIP vs reference indices')
        np.testing.assert_allclose(val, g[name + '/val'], rtol=0, atol=5e-6)


def test_corr_prefilter_block_geometries_and_the_index_only_shortcut(hip, golden, monkeypatch):
    """round 6: (1) the pre-filter with FOUR waves per block (two independent blocks per CU, MREFSR_CORR_W=4: measured slower, kept
    selectable) returns the eight-wave kernel's -- the oracle's -- indices on every golden case and at 160 x 160, where the last
    block row of both geometries takes two column tiles per block; (2) without `want_val` the re-scoring kernel does not evaluate a
    query whose window kept one candidate: the same indices as with the values asked for"""
    g = golden('corr_fmi')
    for w in ('8', '4'):
        monkeypatch.setenv('MREFSR_CORR_W', w)
        for name, fin, fref in cases.corr_cases():
            idx, val = _gpu_fmi(hip, fin, fref, 'fp16')
            np.testing.assert_array_equal(idx, g[name + '/idx'], err_msg=f'{name}: W={w} vs reference indices')
    rng = np.random.default_rng(7)
    fin = rng.standard_normal((1, 256, 160, 160)).astype(np.float32)
    fin = fin + 2.0 * torch.nn.functional.avg_pool2d(torch.from_numpy(fin), 5, 1, 2).numpy()
    fref = np.roll(fin, (17, -23), (2, 3)) + 0.3 * rng.standard_normal(fin.shape).astype(np.float32)
    yi, n2i, hi, d2i = hip.pixnorm(dev(fin), want_bf16_split=True, split='fp16', want_err=True)
    yr, n2r, hr, d2r = hip.pixnorm(dev(fref), want_bf16_split=True, split='fp16', want_err=True)
    nei, _ = hip.patch_norm(n2i)
    _, invr = hip.patch_norm(n2r)
    tau = hip.prefilter_window(nei, invr, d2i, d2r)
    exact, _ = hip.corr_top1(yi, yr, invr, nei, 160, 160)
    for w in ('8', '4'):
        monkeypatch.setenv('MREFSR_CORR_W', w)
        with_val, _ = hip.corr_top1(yi, yr, invr, nei, 160, 160, ybf_in=hi, ybf_ref=hr, tau=tau)
        only_idx, _ = hip.corr_top1(yi, yr, invr, nei, 160, 160, want_val=False, ybf_in=hi, ybf_ref=hr, tau=tau)
        assert torch.equal(with_val, exact) and torch.equal(only_idx, exact), w


def test_corr_rescore_seven_evaluations_per_wave_returns_the_quad_kernel_s_bits(hip, golden, monkeypatch):
    """corr_rescore_lds_kernel (round 6: one lane per (evaluation, tap), operands through a wave-private LDS tile) against
    corr_rescore_kernel (MREFSR_CORR_RESCORE=quad: one evaluation per wave): the same indices AND the same value bits, on every golden
    case (= the reference's indices) and on 160 x 160 maps whose windows keep many candidates (near-duplicate reference patches:
    queries with 2 .. 16 candidates, groups of 32 queries that straddle the end of a pair), with and without `want_val`"""
    g = golden('corr_fmi')
    for name, fin, fref in cases.corr_cases():
        outs = []
        for flag in ('quad', 'lds'):
            monkeypatch.setenv('MREFSR_CORR_RESCORE', flag)
            outs.append(_gpu_fmi(hip, fin, fref, 'fp16'))
        np.testing.assert_array_equal(outs[1][0], g[name + '/idx'], err_msg=f'{name}: vs reference indices')
        np.testing.assert_array_equal(outs[0][0], outs[1][0], err_msg=name)
        np.testing.assert_array_equal(outs[0][1], outs[1][1], err_msg=f'{name}: value bits')
    rng = np.random.default_rng(11)
    seen = set()
    for n_pair, c, hw, dup in ((3, 256, 160, 0.02), (2, 256, 45, 0.0), (2, 256, 70, 0.05)):
        fin = rng.standard_normal((1, c, hw, hw)).astype(np.float32)
        fin = fin + 2.0 * torch.nn.functional.avg_pool2d(torch.from_numpy(fin), 5, 1, 2).numpy()
        # references: shifted copies of the input with a little noise; `dup` > 0 repeats a band of rows: many near-ties per window
        fref = np.concatenate([np.roll(fin, (5 * i + 3, -7 * i - 2), (2, 3)) + (0.3 if dup == 0 else dup) * rng.standard_normal(fin.shape).astype(np.float32)
                               for i in range(n_pair)], 0)
        if dup > 0:
            for r0 in (hw // 2, hw // 2 + 14):
                fref[:, :, r0:r0 + 12] = fref[:, :, 8:20] + 1e-4 * rng.standard_normal(fref[:, :, 8:20].shape).astype(np.float32)
        yi, n2i, hi, d2i = hip.pixnorm(dev(fin), want_bf16_split=True, split='fp16', want_err=True)
        yr, n2r, hr, d2r = hip.pixnorm(dev(fref), want_bf16_split=True, split='fp16', want_err=True)
        nei, _ = hip.patch_norm(n2i)
        _, invr = hip.patch_norm(n2r)
        tau = hip.prefilter_window(nei, invr, d2i, d2r)
        exact_i, exact_v = hip.corr_top1(yi, yr, invr, nei, hw, hw)
        res = {}
        hip._timing['keep_ws'] = True
        for flag in ('quad', 'lds'):
            monkeypatch.setenv('MREFSR_CORR_RESCORE', flag)
            res[flag] = hip.corr_top1(yi, yr, invr, nei, hw, hw, ybf_in=hi, ybf_ref=hr, tau=tau)
            only_idx, _ = hip.corr_top1(yi, yr, invr, nei, hw, hw, want_val=False, ybf_in=hi, ybf_ref=hr, tau=tau)
            assert torch.equal(only_idx, exact_i), (flag, c, hw)
        assert torch.equal(res['lds'][0], exact_i) and torch.equal(res['quad'][0], exact_i), (c, hw)
        assert torch.equal(res['lds'][1], res['quad'][1]) and torch.equal(res['lds'][1], exact_v), (c, hw)
        ws, npair, pp = hip._timing['last_corr_ws']
        cand_n = ws.view(torch.int32)[npair * pp * 16: npair * pp * 17]
        seen.update(int(v) for v in torch.unique(cand_n).tolist())
    hip._timing['keep_ws'] = False
    assert {1, 2, 3, 4}.issubset(seen) and max(seen) >= 5, f'candidate counts met by the test: {sorted(seen)}'
    print('candidate counts met:', sorted(seen))


@pytest.mark.parametrize('c,co,dg,sizes', [(64, 64, 8, ((3, 64, 48), (1, 9, 11), (2, 40, 56))), (128, 128, 8, ((2, 32, 24), (1, 17, 9))),
                                           (64, 128, 4, ((2, 24, 24),)), (64, 64, 1, ((1, 20, 28),))])
def test_dcn_chunk_outer_kernel_returns_the_one_tile_kernel_s_bits(hip, monkeypatch, c, co, dg, sizes):
    """dcn_fwd_pt_kernel (round 6: T tiles per block chunk-outer, offsets / masks staged by LDS-DMA, paired-lane gather through DPP
    operands) against round 5's one-tile kernel (MREFSR_DCN_PT=0) on the same inputs: the SAME BITS -- both tile counts, tile
    groups that run past the map, masks present and absent, deformable groups of 8 / 16 / 64 channels"""
    torch.manual_seed(c + dg)
    for n, h, w in sizes:
        x = torch.randn(n, h, w, c, device='cuda')
        wgt = torch.randn(co, c, 3, 3, device='cuda') * 0.03
        bias = torch.randn(co, device='cuda')
        off = torch.randn(n, 18 * dg, h, w, device='cuda') * 3
        off[:, :, 0, 0] = 0.0
        off[:, 0, 1, 1] = -60.0
        for msk in (torch.rand(n, 9 * dg, h, w, device='cuda'), None):
            monkeypatch.setenv('MREFSR_DCN_PT', '0')
            ref = hip.dcn_fwd(x, off, msk, wgt, bias, 1, 1, 1, 1, dg, 0.1, channels_last=True)
            for t in ('2', '4'):
                monkeypatch.setenv('MREFSR_DCN_PT', '1')
                monkeypatch.setenv('MREFSR_DCN_T', t)
                got = hip.dcn_fwd(x, off, msk, wgt, bias, 1, 1, 1, 1, dg, 0.1, channels_last=True)
                assert torch.equal(got, ref), (c, co, dg, n, h, w, msk is None, t)
    hip.check_conv_range()


def test_image_to_nhwc4_is_aten_s_normalisation_and_packing_bit_for_bit(hip):
    """mrefsr_image_to_nhwc4_f32 (the extractors' input normalisation, vgg_arch.py:150-153 / contras_multi_extractor_arch.py:41 of the
    reference, fused with the channels-last packing) against the ATen operations it replaces -- (x + 1) / 2, (x - mean) / std, zero
    fill, strided copy -- bit for bit, every combination of the two normalisations, odd sizes"""
    from mrefsr_amd.archs import nhwc
    torch.manual_seed(3)
    mean = torch.tensor([0.485, 0.456, 0.406], device='cuda').view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device='cuda').view(1, 3, 1, 1)
    for n, h, w in ((3, 37, 53), (2, 160, 160), (1, 1, 5)):
        x = torch.randn(n, 3, h, w, device='cuda') * 2
        for rn in (False, True):
            for norm in (False, True):
                got = nhwc.image_to_nhwc4(x, mean if norm else None, std if norm else None, rn)
                y = (x + 1) / 2 if rn else x
                y = (y - mean) / std if norm else y
                ref = torch.zeros(n, h, w, 4, device='cuda')
                ref[..., :3] = y.permute(0, 2, 3, 1)
                assert got.shape == ref.shape and torch.equal(got, ref), (n, h, w, rn, norm)


def test_conv1x1_kernel_returns_the_direct_kernel_s_bits(hip, monkeypatch):
    """conv1x1_kernel (round 6: 8 x 32 tiles, two chunks of input in flight, double-buffered split tile, hand-counted waits) against
    conv_nhwc_kernel (MREFSR_CONV1X1=0) on the same inputs: the SAME BITS -- launches large enough for the throughput shapes, ragged
    right / bottom edges, a concatenated (and batch-broadcast) second input with a channel count that is no multiple of 16, one chunk only, an odd
    and an even number of chunks, residual, pre-activation term, input scale, Cout tails above 32; a tail of <= 32 stays on the direct kernel"""
    torch.manual_seed(11)
    cases = [dict(n=2, h=160, w=160, c1=64, cout=128), dict(n=3, h=150, w=139, c1=48, cout=64, res=True),
             dict(n=3, h=133, w=161, c1=48, c2=24, cout=64, pre=True), dict(n=4, h=144, w=160, c1=16, cout=64),
             dict(n=2, h=136, w=160, c1=32, c2=12, cout=128, n2=1), dict(n=1, h=320, w=320, c1=80, cout=104, scaled=True),
             dict(n=2, h=160, w=144, c1=32, cout=96, res=True, scaled=True), dict(n=2, h=160, w=160, c1=64, cout=72),
             dict(n=1, h=157, w=163, c1=48, c2=32, cout=256, res=True), dict(n=4, h=96, w=128, c1=16, cout=128, pre=True)]
    for c in cases:
        n, h, w, c1, cout = c['n'], c['h'], c['w'], c['c1'], c['cout']
        c2 = c.get('c2', 0)
        x1 = torch.randn(n, h, w, c1, device='cuda')
        x2 = torch.randn(c.get('n2', n), h, w, c2, device='cuda') if c2 else None
        pk = hip.conv_pack_weight(torch.randn(cout, c1 + c2, 1, 1, device='cuda') * 0.05, 16)
        bias = torch.randn(cout, device='cuda')
        res = torch.randn(n, h, w, cout, device='cuda') if c.get('res') else None
        pre = torch.randn(1, h, w, cout, device='cuda') if c.get('pre') else None
        amax = None
        if c.get('scaled'):
            x1 = x1 * 3.0e-4
            amax = x1.abs().max().reshape(1)
        outs = []
        for flag in ('0', '1'):
            monkeypatch.setenv('MREFSR_CONV1X1', flag)
            slot = hip.amax_slot(x1.device) if amax is not None else None
            outs.append((hip.conv_nhwc(x1, pk, bias, cout, 1, x2=x2, pre=pre, residual=res, act=True, slope=0.2, in_amax=amax, out_amax=slot), slot))
        assert torch.equal(outs[0][0], outs[1][0]), c
        if amax is not None:
            assert torch.equal(outs[0][1], outs[1][1]) and outs[1][1].item() == outs[1][0].abs().max().item(), c
    hip.check_conv_range()
    # the range guard of the new kernel: one value beyond the fp16 range anywhere in a tile raises the flag
    x = torch.randn(4, 160, 160, 64, device='cuda')
    x[3, 77, 131, 9] = 7.0e4
    pk = hip.conv_pack_weight(torch.randn(64, 64, 1, 1, device='cuda') * 0.05, 16)
    monkeypatch.setenv('MREFSR_CONV1X1', '1')
    hip.conv_nhwc(x, pk, None, 64, 1)
    with pytest.raises(Exception):
        hip.check_conv_range()


def test_general_feature_match_index_vs_oracle_and_reference(hip, golden):
    """feature_match_index with other patch sizes / strides / map sizes (ref_map_util.py:26-86): the general HIP kernel returns
    the oracle's bits and the reference's indices; at patch 3 / stride 1 it returns the bits of the fused MFMA path"""
    from mrefsr_amd.archs.ref_map_util import feature_match_index
    g = golden('fmi_general')
    for name, fin, fref, kw in cases.fmi_general_cases():
        idx, val = feature_match_index(dev(fin), dev(fref), **kw)
        oidx, oval = orc.feature_match_index_generic(fin, fref, **kw)
        assert idx.dtype == torch.int64
        np.testing.assert_array_equal(idx.cpu().numpy(), oidx, err_msg=f'{name}: HIP vs oracle indices')
        np.testing.assert_array_equal(val.cpu().numpy(), oval, err_msg=f'{name}: HIP vs oracle values (bitwise)')
        np.testing.assert_array_equal(idx.cpu().numpy(), g[name + '/idx'], err_msg=f'{name}: HIP vs reference indices')
        np.testing.assert_allclose(val.cpu().numpy(), g[name + '/val'], rtol=2e-6, atol=1e-7)
    fin, fref = dev(synth.randn('fmi/eq/in', (256, 20, 23))), dev(synth.randn('fmi/eq/ref', (256, 20, 23)))
    for is_norm, norm_input in ((True, True), (True, False), (False, False)):
        a = hip.feature_match_index_generic(fin, fref, 3, 1, 1, is_norm, norm_input)
        b = feature_match_index(fin, fref, 3, 1, 1, is_norm, norm_input)     # exact fp32-MFMA kernel
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_corr_top1_batched_pairs(hip):
    """refs stacked [K][B]: pair p uses input p % B."""
    b, k, c, h, w = 2, 3, 256, 14, 17
    fin = synth.randn('cb/in', (b, c, h, w))
    fref = synth.randn('cb/ref', (k * b, c, h, w))
    yi, n2i = hip.pixnorm(dev(fin))
    yr, n2r = hip.pixnorm(dev(fref))
    nei, _ = hip.patch_norm(n2i)
    _, invr = hip.patch_norm(n2r)
    idx, val = hip.corr_top1(yi, yr, invr, nei, h, w)
    _, _, bi = hip.pixnorm(dev(fin), want_bf16_split=True)
    _, _, br = hip.pixnorm(dev(fref), want_bf16_split=True)
    idx2, val2 = hip.corr_top1(yi, yr, invr, nei, h, w, ybf_in=bi, ybf_ref=br)
    for p in range(k * b):
        oidx, oval = orc.feature_match_index(fin[p % b], fref[p])
        np.testing.assert_array_equal(idx[p].cpu().numpy(), oidx)
        np.testing.assert_array_equal(val[p].cpu().numpy(), oval)
        np.testing.assert_array_equal(idx2[p].cpu().numpy(), oidx)
        np.testing.assert_array_equal(val2[p].cpu().numpy(), oval)


@pytest.mark.parametrize('prefilter', [False, True, 'fp16', 'fp16w'])
def test_corr_top1_full_size_properties(hip, prefilter):
    """BASELINE config-2 size (C=256, 160x160): planted correspondences are recovered, and the
    returned index is the fp64 arg-max among sampled candidates (size-independent properties; the
    oracle itself also runs this size in ~10 s and must agree bit-exactly)."""
    c, h, w = 256, 160, 160
    fin = synth.randn('full/in', (c, h, w))
    shift = (17, -23)
    fref = (np.roll(fin, shift, axis=(1, 2)) + synth.randn('full/n', (c, h, w), 0, 0.1)).astype(np.float32)
    idx, val = _gpu_fmi(hip, fin, fref, prefilter)
    ph, pw = h - 2, w - 2
    qy, qx = np.meshgrid(np.arange(ph), np.arange(pw), indexing='ij')
    ry, rx = idx // pw, idx % pw
    # fref[:, y, x] = fin[:, y-17, x+23] (mod size): the query patch at (qy, qx) reappears at
    # (qy+17, qx-23) mod size whenever that 3x3 window does not straddle the wrap-around seam
    ey, ex = (qy + shift[0]) % h, (qx + shift[1]) % w
    nowrap = (ey + 2 < h) & (ex + 2 < w)
    assert nowrap.sum() > 20000
    assert (ry[nowrap] == ey[nowrap]).all() and (rx[nowrap] == ex[nowrap]).all()
    oidx, oval = orc.feature_match_index(fin, fref)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(val, oval)
    yin, _ = orc.pixnorm(fin)
    yref, _ = orc.pixnorm(fref)
    rng = np.random.default_rng(0)
    for q in rng.integers(0, ph * pw, 20):
        best = orc.corr_pair_f64(yin, yref, int(q), int(idx.flat[q]))
        for r in rng.integers(0, ph * pw, 200):
            assert orc.corr_pair_f64(yin, yref, int(q), int(r)) <= best + 1e-6


def test_offsets_from_idx_bit_exact(hip, golden):
    g = golden('corrgen')
    f1 = synth.randn('corrgen/f1', (2, 256, 10, 12))
    f2 = synth.randn('corrgen/f2', (2, 256, 10, 12))
    idx = np.stack([orc.feature_match_index(f1[b], f2[b])[0] for b in range(2)])
    outs = hip.offsets_from_idx(dev(idx), 10, 12)
    np.testing.assert_array_equal(outs[1].cpu().numpy(), g['pre_relu3_1'])
    np.testing.assert_array_equal(outs[2].cpu().numpy(), g['pre_relu2_1'])
    np.testing.assert_array_equal(outs[4].cpu().numpy(), g['pre_relu1_1'])
    # ragged / minimum sizes against the oracle
    for (h, w) in [(3, 3), (3, 9), (7, 5)]:
        rng = np.random.default_rng(h * 100 + w)
        idx = rng.integers(0, (h - 2) * (w - 2), (1, h - 2, w - 2)).astype(np.int64)
        outs = hip.offsets_from_idx(dev(idx), h, w)
        for s, o in zip((1, 2, 4), orc.offsets_from_idx(idx[0], h, w)):
            np.testing.assert_array_equal(outs[s][0].cpu().numpy(), o)


# --------------------------------------------------------------------------------- DynAgg / DCN
def test_dynagg_prep_matches_reference_glue(hip, golden):
    """offset / mask handed to the DCN by the reference's DynAgg.forward (captured call args)."""
    from conftest import spec_from
    g = golden('dynagg')
    sd = synth.state_dict(spec_from(g))
    x1 = synth.randn('dynagg/x1', (2, 64, 9, 11))
    pre = (synth.randn('dynagg/pre', (2, 9, 9, 11, 2)) * 3).round().astype(np.float32)
    om = torch.nn.functional.conv2d(dev(x1), dev(sd['conv_offset_mask.weight']), dev(sd['conv_offset_mask.bias']), padding=1)
    acc = torch.zeros(1, dtype=torch.float64, device='cuda')
    offset, mask = hip.dynagg_prep(om.contiguous(), dev(pre), 8, acc)
    om_nb = torch.nn.functional.conv2d(dev(x1), dev(sd['conv_offset_mask.weight']), None, padding=1)
    off2, mask2 = hip.dynagg_prep(om_nb.contiguous(), dev(pre), 8, None, dev(sd['conv_offset_mask.bias']))  # bias folded in
    np.testing.assert_allclose(off2.cpu().numpy(), g['dcn_offset'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(mask2.cpu().numpy(), g['dcn_mask'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(offset.cpu().numpy(), g['dcn_offset'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(mask.cpu().numpy(), g['dcn_mask'], rtol=0, atol=2e-6)
    want = np.abs(om[:, :144].cpu().numpy().astype(np.float64)).sum()
    assert abs(acc.item() - want) <= 1e-4 * want
    # backward of the glue
    go, gm = torch.randn_like(offset), torch.randn_like(mask)
    g_om = hip.dynagg_prep_bwd(go, gm, mask, 8)
    np.testing.assert_array_equal(g_om[:, :144].cpu().numpy(), go.cpu().numpy())
    np.testing.assert_allclose(g_om[:, 144:].cpu().numpy(), (gm * mask * (1 - mask)).cpu().numpy(), rtol=1e-6, atol=1e-7)


DCN_CASES = [
    # (B, C, H, W, Co, dg, groups, stride, pad, dil, with_mask)      path
    (2, 64, 9, 11, 64, 8, 1, 1, 1, 1, True),      # mfma <1,1>, ragged pixel tile
    (1, 128, 12, 16, 128, 8, 1, 1, 1, 1, True),   # mfma <1,2>
    (1, 256, 10, 13, 256, 8, 1, 1, 1, 1, True),   # mfma <2,2>
    (1, 64, 17, 9, 64, 1, 1, 1, 1, 1, False),     # mfma, DCNv1 (no mask), dg = 1
    (1, 64, 11, 10, 128, 4, 1, 2, 1, 1, True),    # mfma with stride 2
    (2, 8, 7, 6, 8, 4, 1, 1, 1, 1, True),         # generic
    (1, 8, 9, 8, 4, 2, 2, 2, 1, 1, True),         # generic, groups 2, stride 2
    (1, 12, 8, 8, 20, 3, 1, 1, 2, 2, True),       # generic, dilation 2, odd channel counts
]


@pytest.mark.parametrize('case', DCN_CASES)
def test_dcn_forward_vs_oracle(hip, case):
    b, c, h, w, co, dg, groups, stride, pad, dil, with_mask = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((b, c, h, w)).astype(np.float32)
    wgt = (rng.standard_normal((co, c // groups, 3, 3)) * (2.0 / (c * 9)) ** 0.5).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    ho = (h + 2 * pad - (dil * 2 + 1)) // stride + 1
    wo = (w + 2 * pad - (dil * 2 + 1)) // stride + 1
    off = (rng.standard_normal((b, dg * 18, ho, wo)) * 4).astype(np.float32)
    off[:, :, 0, 0] = 0.0           # integer positions
    off[:, 0, 1, 1] = -50.0         # far outside
    msk = rng.random((b, dg * 9, ho, wo)).astype(np.float32) if with_mask else None
    want = orc.dcnv2_fwd(x, off, msk, wgt, bias, stride, pad, dil, groups, dg)
    for nhwc in (True, False):  # NHWC vector gather (default for MFMA shapes) and the NCHW scalar gather
        got = hip.dcn_fwd(dev(x), dev(off), None if msk is None else dev(msk), dev(wgt), dev(bias), stride, pad, dil, groups,
                          dg, nhwc_gather=nhwc)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)  # fp32 GEMM order vs fp64 oracle
    # fused LeakyReLU(0.1) epilogue, no bias
    want2 = orc.dcnv2_fwd(x, off, msk, wgt, None, stride, pad, dil, groups, dg)
    want2 = np.where(want2 > 0, want2, 0.1 * want2)
    got2 = hip.dcn_fwd(dev(x), dev(off), None if msk is None else dev(msk), dev(wgt), None, stride, pad, dil, groups, dg, 0.1)
    np.testing.assert_allclose(got2.cpu().numpy(), want2, rtol=1e-4, atol=1e-4)


def test_dcn_forward_vs_oracle_at_a_chip_filling_grid(hip):
    """the block order per XCD band, the 8-row tiles and the epilogue slab of the fused forward only exist on launches that fill
    the chip: one 64-channel 320 x 320 map (1600 pixel tiles) against the oracle directly, channels-last gather and output"""
    b, c, h, w, co, dg = 1, 64, 320, 320, 64, 8
    rng = np.random.default_rng(321)
    x = rng.standard_normal((b, c, h, w)).astype(np.float32)
    wgt = (rng.standard_normal((co, c, 3, 3)) * (2.0 / (c * 9)) ** 0.5).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    off = (rng.standard_normal((b, dg * 18, h, w)) * 3).astype(np.float32)
    msk = rng.random((b, dg * 9, h, w)).astype(np.float32)
    want = orc.dcnv2_fwd(x, off, msk, wgt, bias, 1, 1, 1, 1, dg)
    want = np.where(want > 0, want, 0.1 * want)
    got = hip.dcn_fwd(_nhwc(x), dev(off), dev(msk), dev(wgt), dev(bias), 1, 1, 1, 1, dg, 0.1, channels_last=True)
    hip.check_conv_range()
    np.testing.assert_allclose(got.permute(0, 3, 1, 2).cpu().numpy(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('case', DCN_CASES)
def test_dcn_backward_pieces_vs_oracle(hip, case):
    b, c, h, w, co, dg, groups, stride, pad, dil, with_mask = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 1)
    x = rng.standard_normal((b, c, h, w)).astype(np.float32)
    wgt = (rng.standard_normal((co, c // groups, 3, 3)) * (2.0 / (c * 9)) ** 0.5).astype(np.float32)
    ho = (h + 2 * pad - (dil * 2 + 1)) // stride + 1
    wo = (w + 2 * pad - (dil * 2 + 1)) // stride + 1
    off = (rng.standard_normal((b, dg * 18, ho, wo)) * 3).astype(np.float32)
    msk = rng.random((b, dg * 9, ho, wo)).astype(np.float32) if with_mask else None
    gout = rng.standard_normal((b, co, ho, wo)).astype(np.float32)
    gx, goff, gm, gw, gb = orc.dcnv2_bwd(x, off, msk, wgt, gout, stride, pad, dil, groups, dg)
    dx, doff, dm, dw, dgo = dev(x), dev(off), None if msk is None else dev(msk), dev(wgt), dev(gout)
    col = hip.dcn_im2col(dx, doff, dm, wgt.shape, stride, pad, dil, groups, dg)  # [B, C*9, HoWo]
    cig, cog = c // groups, co // groups
    # weight / column gradients: plain library GEMMs on the host side (hipBLASLt through torch)
    go_g = dgo.view(b, groups, cog, ho * wo)
    col_g = col.view(b, groups, cig * 9, ho * wo)
    gw_hip = torch.einsum('bgop,bgkp->gok', go_g, col_g).reshape(co, cig, 3, 3)
    np.testing.assert_allclose(gw_hip.cpu().numpy(), gw, rtol=2e-4, atol=2e-4)
    gcol = torch.einsum('gok,bgop->bgkp', dw.view(groups, cog, cig * 9), go_g).reshape(b, c * 9, ho * wo).contiguous()
    gx_h, goff_h, gm_h = hip.dcn_col2im(gcol, dx, doff, dm, wgt.shape, stride, pad, dil, groups, dg)
    np.testing.assert_allclose(goff_h.cpu().numpy(), goff, rtol=2e-4, atol=2e-4)
    if with_mask:
        np.testing.assert_allclose(gm_h.cpu().numpy(), gm, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(gx_h.cpu().numpy(), gx, rtol=2e-4, atol=2e-4)  # atomics: order-free within tol


@pytest.mark.parametrize('n,h,w,cin,cout,ldx,ldg', [(2, 20, 24, 64, 64, 64, 64), (1, 33, 17, 320, 256, 320, 256), (3, 9, 11, 36, 20, 40, 24), (1, 40, 40, 576, 64, 576, 64)])
def test_conv_wgrad1x1_vs_fp64(hip, n, h, w, cin, cout, ldx, ldg):
    """mrefsr_conv_wgrad1x1_f32 (pixel-K GEMM on the matrix pipe, gradients of magnitude 1e-6 scaled by their maximum) against an
    fp64 contraction; channel counts off the 4 / 64 grid and tensors that are channel slices of wider ones"""
    rng = np.random.default_rng(n * 100 + cin)
    x = rng.standard_normal((n, h, w, ldx)).astype(np.float32)
    g = (rng.standard_normal((n, h, w, ldg)) * 1e-6).astype(np.float32)
    want = np.einsum('nhwo,nhwi->oi', g[..., :cout].astype(np.float64), x[..., :cin].astype(np.float64))
    amax = dev(np.array([np.abs(g).max()], np.float32))
    got = hip.conv_wgrad1x1(dev(x)[..., :cin], dev(g)[..., :cout], cin, cout, amax).view(cout, cin).cpu().numpy()
    hip.check_conv_range()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6 * float(np.abs(want).max()))


DCN_FUSED_BWD_CASES = [(2, 64, 9, 11, 64, 8, True), (1, 128, 12, 16, 128, 8, True), (1, 256, 10, 13, 256, 8, True), (1, 64, 17, 9, 64, 2, False),
                        (1, 64, 160, 160, 64, 8, True)]


@pytest.mark.parametrize('case', DCN_FUSED_BWD_CASES)
def test_dcn_fused_backward_vs_oracle(hip, case):
    """mrefsr_dcn_bwd_data_f32 / mrefsr_dcn_bwd_weight_f32 (no column buffer, no library GEMM) against the oracle's
    restatement of deform_conv_cuda_kernel.cu:635-767 + the two GEMMs of deform_conv_cuda.cpp:571-685, incl. one benchmark-size map"""
    b, c, h, w, co, dg, with_mask = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 7)
    x = rng.standard_normal((b, c, h, w)).astype(np.float32)
    wgt = (rng.standard_normal((co, c, 3, 3)) * (2.0 / (c * 9)) ** 0.5).astype(np.float32)
    off = (rng.standard_normal((b, dg * 18, h, w)) * 3).astype(np.float32)
    msk = rng.random((b, dg * 9, h, w)).astype(np.float32) if with_mask else None
    gout = (rng.standard_normal((b, co, h, w)) * 1e-6).astype(np.float32)     # the magnitude of an L1 loss's gradients
    gx, goff, gm, gw, gb = orc.dcnv2_bwd(x, off, msk, wgt, gout, 1, 1, 1, 1, dg)
    dw = dev(wgt)
    amax = dev(np.array([np.abs(gout).max()], np.float32))
    ws = 2.0 ** (13 - int(np.floor(np.log2(np.abs(wgt).max()))))
    pk = hip.conv_pack_view(dw, None, 16, dgrad='T', wscale=ws)
    gx_h, goff_h, gm_h = hip.dcn_bwd_data(_nhwc(gout), _nhwc(x), dev(off), None if msk is None else dev(msk), pk, dg, g_amax=amax)
    hip.check_conv_range()
    tol = dict(rtol=2e-4, atol=2e-4 * 1e-6)
    np.testing.assert_allclose(goff_h.cpu().numpy(), goff, **tol)
    if with_mask:
        np.testing.assert_allclose(gm_h.cpu().numpy(), gm, **tol)
    np.testing.assert_allclose(gx_h.cpu().numpy(), gx, **tol)   # atomics: order-free within the tolerance
    if hasattr(hip, 'dcn_bwd_weight'):
        gw_h = hip.dcn_bwd_weight(_nhwc(gout), _nhwc(x), dev(off), None if msk is None else dev(msk), co, dg, g_amax=amax)
        np.testing.assert_allclose(gw_h.cpu().numpy(), gw, rtol=2e-4, atol=2e-4 * float(np.abs(gw).max()))


# --------------------------------------------------------------------------------- attention
@pytest.mark.parametrize('n,t,c,h,w', [(2, 3, 64, 12, 16), (1, 5, 256, 8, 12), (1, 1, 32, 5, 7), (1, 10, 64, 6, 6)])
def test_mrattn_fwd_bwd_vs_oracle(hip, n, t, c, h, w):
    rng = np.random.default_rng(n * 1000 + t * 100 + c)
    q = (rng.standard_normal((n, c, h, w)) * c ** -0.5).astype(np.float32)
    emb = rng.standard_normal((n, t, c, h, w)).astype(np.float32)
    ass = rng.standard_normal((n, t, 2 * c, h, w)).astype(np.float32)
    want, wprob = orc.mrattn_fwd(q, emb, ass)
    out, prob = hip.mrattn_fwd(dev(q), dev(emb.reshape(n * t, c, h, w)), dev(ass.reshape(n * t, 2 * c, h, w)), t)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(prob.cpu().numpy(), wprob, rtol=1e-5, atol=1e-6)
    g = rng.standard_normal(want.shape).astype(np.float32)
    gq, gemb, gass = orc.mrattn_bwd(q, emb, ass, g)
    hq, hemb, hass = hip.mrattn_bwd(dev(q), dev(emb.reshape(n * t, c, h, w)), dev(ass.reshape(n * t, 2 * c, h, w)), prob,
                                    dev(g), t)
    np.testing.assert_allclose(hq.cpu().numpy(), gq, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(hemb.cpu().numpy().reshape(emb.shape), gemb, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(hass.cpu().numpy().reshape(ass.shape), gass, rtol=1e-4, atol=1e-5)


# --------------------------------------------------------------------------------- fused_act / upfirdn2d
@pytest.mark.parametrize('shape', [(2, 8, 5, 7), (3, 16, 8, 8), (4, 6)])
@pytest.mark.parametrize('act,grad', [(3, 0), (3, 1), (1, 0), (3, 2)])
def test_fused_bias_act_vs_oracle(hip, shape, act, grad):
    rng = np.random.default_rng(len(shape) * 10 + act + grad)
    x = rng.standard_normal(shape).astype(np.float32)
    bias = rng.standard_normal(shape[1]).astype(np.float32)
    ref = rng.standard_normal(shape).astype(np.float32)
    for b_, r_ in ((bias, None), (None, ref), (bias, ref)):
        want = orc.fused_bias_act(x, b_, r_, act, grad, 0.2, 2 ** 0.5)
        got = hip.fused_bias_act(dev(x), None if b_ is None else dev(b_), None if r_ is None else dev(r_), act, grad, 0.2,
                                 2 ** 0.5)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    # reduced-precision storage (the reference dispatches half too)
    for dt in (torch.float16, torch.bfloat16):
        xt, bt = dev(x, dt), dev(bias, dt)
        got = hip.fused_bias_act(xt, bt, None, 3, 0, 0.2, 1.0).float().cpu().numpy()
        xb = (xt.float() + bt.float().view(*([1, -1] + [1] * (x.ndim - 2)))).cpu().numpy()
        want = np.where(xb > 0, xb, 0.2 * xb)
        np.testing.assert_allclose(got, want, rtol=1e-2, atol=1e-2)


def test_fused_bias_act_c_entry_with_a_partial_last_plane(hip):
    """the C entry has no 'size_x is a whole number of planes' precondition (fused_bias_act_kernel.cu:29-33: bias index
    (i / step_b) % size_b of a flat index): a trailing partial plane and size_x < step_b, for the three storage types"""
    import ctypes as C
    from mrefsr_amd import _lib
    rng = np.random.default_rng(7)
    for dt, code, tol in ((torch.float32, 0, 1e-6), (torch.float16, 1, 1e-2), (torch.bfloat16, 2, 2e-2)):
        for size_x, step_b, size_b in ((2 * 64 + 24, 64, 3), (40, 64, 3), (5 * 32 + 8, 32, 2)):
            x = dev(rng.standard_normal(size_x + 64).astype(np.float32), dt)   # (64 guard elements behind the tensor)
            b = dev(rng.standard_normal(size_b).astype(np.float32), dt)
            out = torch.full_like(x, 9.0)
            _lib.call('mrefsr_fused_bias_act', C.c_void_p(x.data_ptr()), C.c_void_p(b.data_ptr()), None, C.c_void_p(out.data_ptr()),
                      C.c_int64(size_x), step_b, size_b, 3, 0, C.c_float(0.2), C.c_float(1.5), code, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            idx = (torch.arange(size_x, device='cuda') // step_b) % size_b
            v = x[:size_x].float() + b.float()[idx]
            want = torch.where(v > 0, v, 0.2 * v) * 1.5
            assert (out[:size_x].float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
            assert (out[size_x:] == 9.0).all()


def test_upfirdn2d_vs_reference_native_and_oracle(hip, golden):
    g = golden('metrics_ops')
    for i, (u, d, p0, p1, ks) in enumerate(g['up_cases']):
        u, d, p0, p1 = int(u), int(d), int(p0), int(p1)
        x, k, ref = g[f'up_x{i}'], g[f'up_k{i}'], g[f'up_out{i}']
        n, c, h, w = x.shape
        out = hip.upfirdn2d(dev(x.reshape(n * c, h, w, 1)), dev(k), u, u, d, d, p0, p1, p0, p1)
        np.testing.assert_allclose(out.cpu().numpy().reshape(ref.shape), ref, rtol=1e-5, atol=1e-5)
    # minor > 1 and anisotropic factors against the oracle
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 6, 7, 3)).astype(np.float32)
    k = rng.random((4, 3)).astype(np.float32)
    want = orc.upfirdn2d(x, k, 2, 1, 1, 2, 1, 2, 0, 3)
    got = hip.upfirdn2d(dev(x), dev(k), 2, 1, 1, 2, 1, 2, 0, 3)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
def test_upfirdn2d_two_byte_blur_and_down2_column_pair_kernel(hip, dt):
    """StyleGAN2's blur / down x2 on 2-byte tensors (upfirdn2d_pair_kernel: a lane owns a column pair, packed fp32 FMAs, 4-byte loads
    and stores) against the fp32 kernels on the same (2-byte-valued) inputs and taps, rounded once: the tap order per output is
    the same, so the bits are; sizes with ragged tiles, more than one tile per row, both pad parities, and shapes that fall
    back to the one-column path (odd widths)"""
    rng = np.random.default_rng(11)
    k1 = np.array([1., 3., 3., 1.])
    k = torch.tensor(np.outer(k1, k1) / 64.0 * 4.0, device='cuda').to(dt)
    for (mj, h, w), (down, pads) in [((6, 70, 300), (1, (2, 1, 2, 1))), ((3, 32, 256), (1, (2, 1, 2, 1))), ((5, 41, 130), (1, (1, 2, 1, 2))),
                                     ((4, 64, 260), (2, (1, 1, 1, 1))), ((4, 20, 258), (2, (1, 1, 1, 1))), ((3, 50, 300), (2, (2, 0, 1, 1))), ((2, 16, 512), (2, (1, 1, 1, 1))),
                                     ((2, 33, 131), (1, (2, 1, 2, 1))), ((2, 34, 129), (2, (1, 1, 1, 1)))]:
        x = torch.tensor(rng.standard_normal((mj, h, w, 1)).astype(np.float32), device='cuda').to(dt)
        got = hip.upfirdn2d(x, k, 1, 1, down, down, *pads)
        want = hip.upfirdn2d(x.float(), k.float(), 1, 1, down, down, *pads).to(dt)
        assert got.shape == want.shape
        assert torch.equal(got, want), (mj, h, w, down, pads, float((got.float() - want.float()).abs().max()))
        ref = orc.upfirdn2d(x.float().cpu().numpy(), k.float().cpu().numpy(), 1, 1, down, down, *pads)
        np.testing.assert_allclose(got.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)


def test_ops_refuse_cpu_tensors(hip):
    with pytest.raises(NotImplementedError):
        hip.pixnorm(torch.zeros(1, 8, 4, 4))
    with pytest.raises(NotImplementedError):
        hip.fused_bias_act(torch.zeros(2, 3), torch.zeros(3), None, 3, 0, 0.2, 1.0)


def test_tail_bilinear_add_returns_torch_bits(hip):
    """the network's tail (ref_mrapa_restoration_arch.py:132-137: F.interpolate(x, None, 4, 'bilinear', False) + the last convolution's
    output) as one pass: the bits of the literal torch ops on the GPU -- interpolation restated, one fp32 add -- for the path's 3-channel
    image, odd sizes, a padded channels-last source (the convolution writes Cout = 3 into a wider row), scale 2"""
    import torch.nn.functional as F
    torch.manual_seed(9)
    for b, c, h, w, ld, scale in ((2, 3, 40, 40, 3, 4), (1, 3, 37, 53, 4, 4), (3, 5, 16, 20, 8, 2), (1, 1, 1, 7, 1, 4), (2, 3, 160, 160, 3, 4)):
        x = torch.rand(b, c, h, w, device='cuda')
        wide = torch.randn(b, h * scale, w * scale, ld, device='cuda')
        y = wide[..., :c]
        got = hip.tail_bilinear_add(y, x, scale)
        base = F.interpolate(x, None, scale, 'bilinear', False)
        want = (y.permute(0, 3, 1, 2).float() + base).contiguous()
        assert torch.equal(hip.tail_bilinear_add(torch.zeros_like(y), x, scale), base), (b, c, h, w, 'interpolation bits')
        assert torch.equal(got, want), (b, c, h, w, ld, scale)


# --------------------------------------------------------------------------------- conv epilogue
@pytest.mark.parametrize('shape', [(2, 8, 6, 10), (3, 5, 7, 9), (1, 64, 32, 32)])
@pytest.mark.parametrize('slope', [1.0, 0.0, 0.1])
def test_bias_act_res_matches_torch_ops_bitwise(hip, shape, slope):
    """(x + b[c]) -> LeakyReLU(slope) -> + residual: same fp32 operations in the same order as the
    separate torch kernels it replaces, so the result is bit-identical"""
    rng = np.random.default_rng(int(slope * 10) + shape[1])
    x = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape[1]).astype(np.float32)
    r = rng.standard_normal(shape).astype(np.float32)
    for use_b, use_r in ((True, False), (True, True), (False, True)):
        xt = dev(x)
        want = xt + dev(b).view(1, -1, 1, 1) if use_b else xt.clone()
        want = torch.where(want > 0, want, want * slope)
        if use_r:
            want = want + dev(r)
        got = hip.bias_act_res_(dev(x), dev(b) if use_b else None, slope, dev(r) if use_r else None)
        np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy())
    # pre-activation addend broadcast over groups of images (K references sharing one x-half)
    x3 = np.concatenate([x, x * 0.5, x - 1.0])
    pre = rng.standard_normal(shape).astype(np.float32)
    want = dev(x3) + dev(b).view(1, -1, 1, 1) + dev(pre).repeat(3, 1, 1, 1)
    want = torch.where(want > 0, want, want * slope)
    got = hip.bias_act_res_(dev(x3), dev(b), slope, pre=dev(pre))
    np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy())


def test_conv_act_fused_path_equals_module_path(hip):
    """archs.arch_util.conv_act: inference (fused epilogue) == autograd-enabled (plain torch ops)"""
    from mrefsr_amd.archs.arch_util import ResidualBlockNoBN, conv_act
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(16, 24, 3, 1, 1).cuda()
    x = torch.randn(2, 16, 20, 28, device='cuda')
    res = torch.randn(2, 24, 20, 28, device='cuda')
    with torch.no_grad():
        fused = conv_act(conv, x, 0.1, residual=res)
    plain = conv_act(conv, x.requires_grad_(True), 0.1, residual=res)
    np.testing.assert_allclose(fused.cpu().numpy(), plain.detach().cpu().numpy(), rtol=0, atol=1e-6)
    blk = ResidualBlockNoBN(16).cuda()
    x2 = torch.randn(2, 16, 12, 12, device='cuda')
    with torch.no_grad():
        a = blk(x2)
    bb = blk(x2.clone().requires_grad_(True))
    np.testing.assert_allclose(a.cpu().numpy(), bb.detach().cpu().numpy(), rtol=0, atol=1e-6)
    bb.sum().backward()  # the training path keeps a graph


@pytest.mark.parametrize('variant,mode', [('MREFSR_CORR_PREFILTER_WS16', 'fp16w'), ('MREFSR_CORR_PREFILTER_WS', 'bf16'),
                                          ('MREFSR_CORR_PREFILTER_STREAM', 'bf16')])
def test_ab_build_prefilter_generations_bit_exact(variant, mode):
    """the earlier pre-filter generations live only in the -DMREFSR_AB_KERNELS build (mrefsr_amd/lib_ab, made by
    __graft_entry__.build()); through MREFSR_HIP_LIB they still return the oracle's bits (tools/ab_check.py)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'mrefsr_amd', 'lib_ab', 'libmrefsr_hip.so')
    if not os.path.exists(lib):
        pytest.skip('A/B build of the library not present (python -c "import __graft_entry__ as g; g.build()")')
    env = dict(os.environ, MREFSR_HIP_LIB=lib, **{variant: '1'})
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'ab_check.py'), mode], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_prefilter_degenerate_inputs_fall_back_to_brute_force(hip):
    """every reference patch identical (constant map): every query overflows its candidate list
    and is brute-forced; the tie rule must give index 0 everywhere.  Also a map with a zero pixel."""
    c, h, w = 256, 20, 24
    fin = synth.randn('deg/in', (c, h, w))
    fref = np.ones((c, h, w), np.float32)
    for prefilter in (False, True, 'fp16', 'fp16w'):
        idx, val = _gpu_fmi(hip, fin, fref, prefilter)
        oidx, oval = orc.feature_match_index(fin, fref)
        np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_array_equal(val, oval)
    assert (oidx == 0).all()
    fref2 = synth.randn('deg/ref', (c, h, w))
    fref2[:, 3, 4] = 0.0
    fin2 = fin.copy()
    fin2[:, 7, 7] = 0.0
    oidx, oval = orc.feature_match_index(fin2, fref2)
    for prefilter in (True, 'fp16', 'fp16w'):
        idx, val = _gpu_fmi(hip, fin2, fref2, prefilter)
        np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_array_equal(val, oval)


@pytest.mark.parametrize('shape', [(2, 5, 8, 12), (1, 3, 9, 7), (2, 4, 6, 10)])
def test_bias_relu_pool2_matches_torch_bitwise(hip, shape):
    rng = np.random.default_rng(shape[2] * 10 + shape[3])
    x = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape[1]).astype(np.float32)
    want = torch.nn.functional.max_pool2d(torch.relu(dev(x) + dev(b).view(1, -1, 1, 1)), 2, 2)
    got = hip.bias_relu_pool2(dev(x), dev(b))
    np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy())


def _nhwc(a):
    return dev(a).permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('shape', [(2, 16, 64, 20, 40), (1, 64, 64, 33, 70), (1, 32, 216, 16, 32), (1, 12, 8, 9, 11), (2, 4, 30, 18, 34)])
@pytest.mark.parametrize('terms', [6, 16])
@pytest.mark.parametrize('ksize', [3, 1])
def test_conv_nhwc_fp32_equivalent(hip, shape, terms, ksize):
    """conv on the bf16 pipe with the 6-product split is as close to the fp64 result as an fp32 convolution is"""
    import torch.nn.functional as F
    n, ci, co, h, w = shape
    rng = np.random.default_rng(ci * 1000 + co)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, ksize, ksize)) / np.sqrt(ksize * ksize * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    res = rng.standard_normal((n, co, h, w)).astype(np.float32)
    tx, tw, tb, tr = (torch.from_numpy(a) for a in (x, wt, b, res))
    want64 = F.leaky_relu(F.conv2d(tx.double(), tw.double(), tb.double(), 1, ksize // 2), 0.1) + tr.double()
    f32 = F.leaky_relu(F.conv2d(tx, tw, tb, 1, ksize // 2), 0.1) + tr
    packed = hip.conv_pack_weight(dev(wt), terms)
    got = hip.conv_nhwc(_nhwc(x), packed, dev(b), co, ksize, residual=_nhwc(res), act=True, slope=0.1, terms=terms)
    got = got.permute(0, 3, 1, 2).cpu().double()
    err = (got - want64).abs().max().item()
    err32 = (f32.double() - want64).abs().max().item()
    rms = (got - want64).pow(2).mean().sqrt().item()
    rms32 = (f32.double() - want64).pow(2).mean().sqrt().item()
    print(f'conv k={ksize} terms={terms} shape={shape}: max err {err:.3e} rms {rms:.3e}  (fp32 CPU conv: {err32:.3e} rms {rms32:.3e})')
    if terms in (6, 16):
        # "as accurate as an fp32 convolution": the error's RMS within 1.75x of oneDNN's fp32 result on the same inputs (the
        # robust statistic), its maximum within 3x (a maximum over 10^4..10^5 outputs is itself noisy)
        assert rms <= 1.75 * rms32, (rms, rms32)
        assert err <= max(3 * err32, 1.5e-6), (err, err32)
    else:
        assert err <= 2e-4


# ---- conv_wino_kernel: the Winograd F(2x2, 3x3) form of the terms-16 convolution (descriptor terms 17)
def _wino_waves(n):
    import os
    if n is None:
        os.environ.pop('MREFSR_WINO_WAVES', None)
    else:
        os.environ['MREFSR_WINO_WAVES'] = str(n)


@pytest.mark.parametrize('shape', [(2, 48, 64, 20, 40), (1, 64, 64, 33, 70), (1, 80, 216, 16, 32), (1, 256, 40, 9, 11), (2, 36, 30, 18, 34),
                                   (1, 512, 64, 16, 16)])
def test_conv_wino_fp32_equivalent(hip, shape):
    """the Winograd form of the fp16 two-term split is as close to the fp64 result as an fp32 direct convolution is (the bar of
    test_conv_nhwc_fp32_equivalent), with bias + LeakyReLU + residual fused"""
    import torch.nn.functional as F
    n, ci, co, h, w = shape
    rng = np.random.default_rng(ci * 1000 + co)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    res = rng.standard_normal((n, co, h, w)).astype(np.float32)
    tx, tw, tb, tr = (torch.from_numpy(a) for a in (x, wt, b, res))
    want64 = F.leaky_relu(F.conv2d(tx.double(), tw.double(), tb.double(), 1, 1), 0.1) + tr.double()
    f32 = F.leaky_relu(F.conv2d(tx, tw, tb, 1, 1), 0.1) + tr
    packed = hip.conv_pack_weight(dev(wt), 17)
    got = hip.conv_nhwc(_nhwc(x), packed, dev(b), co, 3, residual=_nhwc(res), act=True, slope=0.1, terms=17)
    hip.check_conv_range()
    got = got.permute(0, 3, 1, 2).cpu().double()
    err, err32 = (got - want64).abs().max().item(), (f32.double() - want64).abs().max().item()
    rms, rms32 = (got - want64).pow(2).mean().sqrt().item(), (f32.double() - want64).pow(2).mean().sqrt().item()
    print(f'conv_wino shape={shape}: max err {err:.3e} rms {rms:.3e}  (fp32 CPU conv: {err32:.3e} rms {rms32:.3e})')
    assert rms <= 1.75 * rms32, (rms, rms32)
    assert err <= max(3 * err32, 1.5e-6), (err, err32)


def test_conv_wino_small_activations(hip):
    """the low term of an activation is not scaled into its binade (conv_wino.hip: split_pair): below |x| = 2^-3 it is an fp16
    subnormal with an absolute error <= 2^-25.  Activations of magnitude 1e-2 / 1e-3: the error relative to the output's RMS stays
    within 2e-6 / 2e-5 (an fp32 convolution: ~2e-7); stated, not hidden"""
    import torch.nn.functional as F
    rng = np.random.default_rng(77)
    n, ci, co, h, w = 1, 64, 64, 24, 24
    wt = (rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32)
    packed = hip.conv_pack_weight(dev(wt), 17)
    for scale, bar in ((1e-2, 2e-6), (1e-3, 2e-5)):
        x = (rng.standard_normal((n, ci, h, w)) * scale).astype(np.float32)
        want = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), None, 1, 1)
        got = hip.conv_nhwc(_nhwc(x), packed, None, co, 3, terms=17).permute(0, 3, 1, 2).cpu().double()
        rel = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
        print(f'conv_wino activations ~{scale:g}: rms error / rms output = {rel:.2e}')
        assert rel <= bar, (scale, rel)


def test_conv_wino_input_scale_makes_small_activations_fp32_equivalent(hip):
    """with the layer's input maximum handed over (in_amax: the producing launch's out_amax, archs/nhwc.py) both Winograd kernels
    scale their input by a power of two into the fp16 normal range: activations of magnitude 1e-2 ... 1e-6 -- and a trained-like mix
    (post-LeakyReLU values with 1e-3 typical size and rare outliers 30 times larger) -- come out as accurate as an fp32 convolution
    (error RMS within 1.75x of oneDNN's fp32 result against fp64, the bar of test_conv_wino_fp32_equivalent), the two kernels agree
    to the bit, and the scale is exact: a tensor scaled by 2^-7 with in_amax scaled alike gives 2^-7 times the same bits"""
    import torch.nn.functional as F
    rng = np.random.default_rng(78)
    for (n, ci, co, h, w) in ((1, 64, 64, 32, 32), (1, 48, 64, 24, 40)):       # four-wave kernel / eight-wave kernel (ragged tiles)
        wt = (rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32)
        packed = hip.conv_pack_weight(dev(wt), 17)
        for scale in (1e-2, 1e-3, 1e-6, 'mix'):
            if scale == 'mix':
                x = rng.standard_normal((n, ci, h, w)) * 1e-3
                x = np.where(x > 0, x, 0.1 * x) * np.where(rng.random((n, ci, h, w)) < 1e-3, 30.0, 1.0)
            else:
                x = rng.standard_normal((n, ci, h, w)) * scale
            x = x.astype(np.float32)
            tx = torch.from_numpy(x)
            want = F.conv2d(tx.double(), torch.from_numpy(wt).double(), None, 1, 1)
            f32 = F.conv2d(tx, torch.from_numpy(wt), None, 1, 1).double()
            xd = _nhwc(x)
            am = xd.abs().amax().reshape(1)
            outs = []
            try:
                for nw in (8, 4):
                    _wino_waves(nw)
                    outs.append(hip.conv_nhwc(xd, packed, None, co, 3, terms=17, in_amax=am))
            finally:
                _wino_waves(None)
            hip.check_conv_range()
            assert torch.equal(outs[0], outs[1])
            got = outs[1].permute(0, 3, 1, 2).cpu().double()
            rms, rms32 = (got - want).pow(2).mean().sqrt().item(), (f32 - want).pow(2).mean().sqrt().item()
            plain = hip.conv_nhwc(xd, packed, None, co, 3, terms=17).permute(0, 3, 1, 2).cpu().double()
            print(f'conv_wino {h}x{w} activations ~{scale}: rms error {rms:.2e} with the input scale, {(plain - want).pow(2).mean().sqrt().item():.2e} '
                  f'without, fp32 convolution {rms32:.2e}')
            assert rms <= 1.75 * rms32, (scale, rms, rms32)
            small = hip.conv_nhwc(xd * 2.0 ** -7, packed, None, co, 3, terms=17, in_amax=am * 2.0 ** -7)
            assert torch.equal(small, outs[1] * 2.0 ** -7)


def test_out_amax_of_every_producing_kernel_is_the_tensor_maximum(hip):
    """mrefsr_conv_nhwc_amax_f32 / mrefsr_dcn_fwd_amax_f32 (round 6): every forward launch of the engine writes max |out| into a
    zeroed device word from its epilogue -- the direct kernels (1x1, 3x3, pooled, pixel-shuffled, ragged Cout), both Winograd kernels
    (plain / residual / pre / pooled / pixel-shuffled) and the DCN kernels at the three channel counts: the word equals
    out.abs().max() exactly (a maximum is exact), and a second launch into the same word keeps the larger value"""
    torch.manual_seed(21)
    cases = [  # n, h, w, cin, cout, k, terms, residual, pre, epilogue
        (2, 20, 24, 32, 40, 1, 16, 0, 0, 0), (2, 32, 32, 4, 64, 3, 16, 0, 0, 0), (2, 32, 32, 16, 24, 3, 16, 0, 0, 1),
        (2, 16, 16, 32, 64, 3, 16, 1, 0, 2), (3, 48, 48, 64, 128, 3, 17, 0, 0, 0), (3, 48, 48, 64, 64, 3, 17, 1, 0, 0),
        (2, 32, 32, 64, 64, 3, 17, 0, 1, 0), (2, 32, 32, 64, 64, 3, 17, 0, 0, 1), (2, 24, 40, 48, 40, 3, 17, 0, 0, 0),
        (2, 16, 16, 64, 256, 3, 17, 0, 0, 2), (9, 64, 64, 128, 256, 3, 16, 0, 0, 0)]
    for n, h, w, ci, co, k, terms, res, pre, ep in cases:
        x = torch.randn(n, h, w, ci, device='cuda')
        wt = torch.randn(co, ci, k, k, device='cuda') / (k * ci ** 0.5)
        pk = hip.conv_pack_weight(wt, terms)
        oshape = {0: (n, h, w, co), 1: (n, h // 2, w // 2, co), 2: (n, 2 * h, 2 * w, co // 4)}[ep]
        r = torch.randn(oshape, device='cuda') if res and ep == 0 else None
        p_ = torch.randn(1, h, w, co, device='cuda') if pre else None
        slot = hip.amax_slot(x.device)
        out = hip.conv_nhwc(x, pk, torch.randn(co, device='cuda'), co, k, residual=r, pre=p_, act=True, slope=0.1, epilogue=ep, out_amax=slot)
        assert slot.item() == out.abs().max().item(), (n, h, w, ci, co, k, terms, res, pre, ep)
        big = hip.conv_nhwc(x * 3, pk, None, co, k, epilogue=ep, out_amax=slot)
        assert slot.item() == max(out.abs().max().item(), big.abs().max().item())
    hip.check_conv_range()
    for c, hw in ((64, 40), (128, 24), (256, 16)):
        x = torch.randn(2, hw, hw, c, device='cuda')
        off = torch.randn(2, 144, hw, hw, device='cuda') * 2
        msk = torch.rand(2, 72, hw, hw, device='cuda')
        wgt = torch.randn(c, c, 3, 3, device='cuda') * 0.02
        slot = hip.amax_slot(x.device)
        out = hip.dcn_fwd(x, off, msk, wgt, torch.randn(c, device='cuda'), 1, 1, 1, 1, 8, 0.1, channels_last=True, out_amax=slot)
        assert slot.item() == out.abs().max().item(), c


def test_winograd_input_scale_is_the_current_batch_s(hip):
    """archs/nhwc.conv: the input scale of a Winograd layer is the maximum its PRODUCER measured for this very tensor -- a batch 2^16
    times smaller than the one before it comes out at fp32 level (round 5 measured once per layer: such a batch lost its low terms
    until the next refresh), a batch 2^10 times larger does not trip the range flag, and no reduction launch is involved (the
    caller's own input tensor is the only one measured)"""
    import torch.nn.functional as F
    from torch import nn
    from mrefsr_amd.archs import nhwc
    torch.manual_seed(3)
    c1, c2 = nn.Conv2d(64, 64, 3, 1, 1).cuda(), nn.Conv2d(64, 128, 3, 1, 1).cuda()
    base = torch.randn(2, 32, 32, 64, device='cuda')
    with torch.no_grad():
        for scale in (1.0, 2.0 ** -16, 2.0 ** 10, 2.0 ** -16):
            x = base * scale
            before = nhwc.AMAX_MEASURED[0]
            y1 = nhwc.conv(c1, x, bias=False)
            y2 = nhwc.conv(c2, y1, bias=False)
            assert nhwc.AMAX_MEASURED[0] - before == 1          # x is the caller's tensor; y1 carries its producer's word
            assert getattr(y1, nhwc.AMAX_ATTR).item() == y1.abs().max().item()
            hip.check_conv_range()
            xd = x.permute(0, 3, 1, 2).double().cpu()
            w1, w2 = c1.weight.double().cpu(), c2.weight.double().cpu()
            want = F.conv2d(F.conv2d(xd, w1, None, 1, 1), w2, None, 1, 1)
            f32 = F.conv2d(F.conv2d(xd.float(), w1.float(), None, 1, 1), w2.float(), None, 1, 1).double()
            got = y2.permute(0, 3, 1, 2).double().cpu()
            rms, rms32 = (got - want).pow(2).mean().sqrt().item(), (f32 - want).pow(2).mean().sqrt().item()
            print(f'two Winograd layers at input scale {scale:g}: rms error {rms:.2e}, fp32 convolutions {rms32:.2e}')
            assert rms <= 1.75 * rms32, (scale, rms, rms32)


def test_conv_wino_epilogues_sources_slices(hip):
    """everything conv_nhwc fuses, through the Winograd kernel: cat([x broadcast over K, ref]) + bias + broadcast pre-activation
    term + PReLU into a channel slice; MaxPool2d(2,2); PixelShuffle(2); ragged sizes, channel counts off the 16 / 64 grid"""
    import torch.nn.functional as F
    rng = np.random.default_rng(15)
    b, k, c1, c2, co, h, w = 2, 3, 16, 24, 40, 10, 37
    x = rng.standard_normal((b, c1, h, w)).astype(np.float32)
    r = rng.standard_normal((k * b, c2, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, c1 + c2, 3, 3)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    pre = rng.standard_normal((b, co, h, w)).astype(np.float32)
    a = np.float32(0.25)
    want = F.prelu(F.conv2d(torch.cat([torch.from_numpy(x).repeat(k, 1, 1, 1), torch.from_numpy(r)], 1).double(),
                            torch.from_numpy(wt).double(), torch.from_numpy(bias).double(), 1, 1)
                   + torch.from_numpy(pre).double().repeat(k, 1, 1, 1), torch.tensor([a], dtype=torch.float64))
    wide_in = torch.zeros(k * b, h, w, c2 + 8, device='cuda')
    wide_in[..., 4:4 + c2] = _nhwc(r)
    wide_out = torch.full((k * b, h, w, co + 12), 7.0, device='cuda')
    hip.conv_nhwc(_nhwc(x), hip.conv_pack_weight(dev(wt), 17), dev(bias), co, 3, x2=wide_in[..., 4:4 + c2], pre=_nhwc(pre), act=True,
                  slope_ptr=dev(np.array([a])), out=wide_out[..., 8:8 + co])
    got = wide_out[..., 8:8 + co].permute(0, 3, 1, 2).cpu().double()
    assert (got - want).abs().max().item() < 1e-5
    assert (wide_out[..., :8] == 7.0).all() and (wide_out[..., 8 + co:] == 7.0).all()
    # pooled / pixel-shuffled outputs
    n, ci, co, h, w = 2, 40, 24, 12, 40   # (an odd number of 16-channel chunks: the pair is padded with a zero chunk)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    conv = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(bias).double(), 1, 1)
    pk = hip.conv_pack_weight(dev(wt), 17)
    got = hip.conv_nhwc(_nhwc(x), pk, dev(bias), co, 3, act=True, slope=0.0, epilogue=1).permute(0, 3, 1, 2).cpu().double()
    assert (got - F.max_pool2d(torch.relu(conv), 2, 2)).abs().max().item() < 1e-5
    got = hip.conv_nhwc(_nhwc(x), pk, dev(bias), co, 3, act=True, slope=0.1, epilogue=2).permute(0, 3, 1, 2).cpu().double()
    assert (got - F.pixel_shuffle(F.leaky_relu(conv, 0.1), 2)).abs().max().item() < 1e-5
    hip.check_conv_range()


def test_conv_wino_many_tiles_per_block_equals_direct_kernel(hip):
    """more tiles than persistent blocks (every block walks several tiles, the chunk stream runs across their boundaries, the
    last tiles of a band are ragged): the Winograd kernel against the direct one on the same inputs, and bit-reproducible"""
    torch.manual_seed(3)
    for n, h, w, ci, co in ((3, 250, 330, 64, 64), (2, 100, 92, 144, 192)):
        x = torch.randn(n, h, w, ci, device='cuda')
        wt = torch.randn(co, ci, 3, 3, device='cuda') / (3.0 * ci ** 0.5)
        bias = torch.randn(co, device='cuda')
        r = torch.randn(n, h, w, co, device='cuda')
        ref = hip.conv_nhwc(x, hip.conv_pack_weight(wt, 16), bias, co, 3, residual=r)
        pk = hip.conv_pack_weight(wt, 17)
        got = hip.conv_nhwc(x, pk, bias, co, 3, residual=r)
        again = hip.conv_nhwc(x, pk, bias, co, 3, residual=r)
        hip.check_conv_range()
        assert torch.equal(got, again)
        assert (got - ref).abs().max().item() < 2e-5, (n, h, w, ci, co)
    # the pre-activation term (batch-broadcast) on interior tiles: the kernel's third instantiation
    n, h, w, ci, co = 6, 72, 56, 64, 64
    x = torch.randn(n, h, w, ci, device='cuda')
    wt = torch.randn(co, ci, 3, 3, device='cuda') / (3.0 * ci ** 0.5)
    bias, pre = torch.randn(co, device='cuda'), torch.randn(2, h, w, co, device='cuda')
    ref = hip.conv_nhwc(x, hip.conv_pack_weight(wt, 16), bias, co, 3, pre=pre, act=True, slope=0.1)
    got = hip.conv_nhwc(x, hip.conv_pack_weight(wt, 17), bias, co, 3, pre=pre, act=True, slope=0.1)
    hip.check_conv_range()
    assert (got - ref).abs().max().item() < 2e-5
    want = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(
        x.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), bias.double().cpu(), 1, 1) + pre.permute(0, 3, 1, 2).double().cpu().repeat(3, 1, 1, 1), 0.1)
    assert (got.permute(0, 3, 1, 2).double().cpu() - want).abs().max().item() < 1e-5


def test_conv_wino_large_launch_is_reproducible_and_equals_direct_kernel(hip):
    """launches of the benchmark's size (tensors far beyond the caches: the waves of a block drift apart, which is what a
    missing barrier between a tile's output exchange and the next tile's first patch stores, and a counted wait of a block's first
    sub-step that had nothing younger to count, needed to show -- about one run in four had wrong first tiles): repeated runs into
    NaN-filled outputs at different addresses are bit-identical, complete, and agree with the direct kernel"""
    torch.manual_seed(5)
    for n, h, w, ci, co, act in ((10, 640, 640, 64, 64, True), (10, 320, 320, 128, 128, True), (10, 640, 640, 64, 128, False)):
        x = torch.randn(n, h, w, ci, device='cuda')
        wt = torch.randn(co, ci, 3, 3, device='cuda') / (3.0 * ci ** 0.5)
        bias = torch.randn(co, device='cuda')
        pk = hip.conv_pack_weight(wt, 17)
        outs, pad = [], []
        for rep in range(5 if h == 320 else 3):
            pad.append(torch.empty(1 + 5000 * (rep + 1), device='cuda'))
            o = torch.full((n, h, w, co), float('nan'), device='cuda')
            hip.conv_nhwc(x, pk, bias, co, 3, act=act, slope=0.1, out=o)
            outs.append(o)
        hip.check_conv_range()
        assert not torch.isnan(outs[0]).any()
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (n, h, w, ci, co)
        ref = hip.conv_nhwc(x, hip.conv_pack_weight(wt, 16), bias, co, 3, act=act, slope=0.1)
        assert (outs[0] - ref).abs().max().item() < 2e-5
        del outs, ref, x


def test_conv_wino_counted_waits_cover_their_loads():
    """the Winograd kernel's hand-counted s_waitcnt protocol under the checking build (mrefsr_amd/lib_ab): a software shadow of the
    in-order vmcnt counter (loads issued / proven returned / the issue count behind what a wait guards) counts every patch piece and
    weight fragment that its wait's count does not reach -- none may, at launches of every instantiation
    (tools/conv_wino_arrival_check.py; a build with the first-wait fault put back counts 1024 per launch)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'mrefsr_amd', 'lib_ab', 'libmrefsr_hip.so')
    if not os.path.exists(lib):
        pytest.skip('no A/B build of the library (python -c "import __graft_entry__ as g; g.build()")')
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'conv_wino_arrival_check.py')], env=dict(os.environ, MREFSR_HIP_LIB=lib),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count(' ok') == 8, out.stdout


def test_conv_wino4_is_bit_identical_to_the_eight_wave_kernel(hip):
    """the four-wave Winograd kernel (csrc/conv_wino4.hip: the default for whole 16 x 16 tiles and whole cout blocks) against the
    eight-wave one (MREFSR_WINO_WAVES=8) on the same inputs -- same arithmetic in the same order: the SAME BITS -- for every
    instantiation: plain / residual / broadcast pre-activation term / max-pool epilogue, one and two sources (batch-broadcast first
    source), channel counts off the 16 grid (a ragged last chunk of the only or the second source), outputs into a channel slice, several
    tiles per block across tile boundaries, streamed (beyond-the-cache) outputs; and against fp64"""
    import torch.nn.functional as F
    torch.manual_seed(11)
    cases = [  # n, h, w, c1, c2, co, residual, pre, act, epilogue
        (1, 16, 16, 48, 0, 64, 0, 0, 0, 0), (2, 48, 48, 64, 0, 64, 0, 1, 1, 0), (9, 64, 64, 64, 0, 128, 1, 0, 1, 0),
        (4, 96, 80, 128, 0, 64, 0, 0, 1, 1), (2, 32, 48, 36, 0, 64, 1, 0, 1, 0), (3, 32, 32, 32, 36, 192, 0, 0, 1, 0),
        (6, 48, 32, 64, 64, 64, 1, 0, 0, 0), (2, 64, 32, 40, 0, 128, 0, 1, 1, 0), (30, 320, 320, 64, 0, 64, 1, 0, 1, 0),
        (40, 160, 160, 64, 0, 64, 0, 0, 1, 1)]
    try:
        for n, h, w, c1, c2, co, res, pre, act, ep in cases:
            nb = 2 if (c2 and n % 2 == 0) else n                     # the first source is batch-broadcast where the batch allows
            x1 = torch.randn(nb if c2 else n, h, w, c1, device='cuda')
            x2 = torch.randn(n, h, w, c2, device='cuda') if c2 else None
            wt = torch.randn(co, c1 + c2, 3, 3, device='cuda') / (3.0 * (c1 + c2) ** 0.5)
            bias = torch.randn(co, device='cuda')
            r = torch.randn(n, h, w, co, device='cuda') if res else None
            p = torch.randn(2 if n % 2 == 0 else 1, h, w, co, device='cuda') if pre else None
            pk = hip.conv_pack_weight(wt, 17)
            outs = []
            for nw in (8, 4, None):
                _wino_waves(nw)
                ho, wo = (h // 2, w // 2) if ep == 1 else (h, w)
                wide = torch.full((n, ho, wo, co + 8), 7.0, device='cuda')
                hip.conv_nhwc(x1, pk, bias, co, 3, x2=x2, residual=r, pre=p, act=bool(act), slope=0.1, epilogue=ep, out=wide[..., 4:4 + co])
                assert (wide[..., :4] == 7.0).all() and (wide[..., 4 + co:] == 7.0).all()
                outs.append(wide[..., 4:4 + co].clone())
            hip.check_conv_range()
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (n, h, w, c1, c2, co, res, pre, act, ep)
            if n * h * w <= 40000:
                xin = x1.repeat(n // x1.shape[0], 1, 1, 1) if c2 else x1
                xin = torch.cat([xin, x2], -1) if c2 else xin
                y = F.conv2d(xin.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), bias.double().cpu(), 1, 1)
                if pre:
                    y = y + p.permute(0, 3, 1, 2).double().cpu().repeat(n // p.shape[0], 1, 1, 1)
                if act:
                    y = F.leaky_relu(y, 0.1)
                if res:
                    y = y + r.permute(0, 3, 1, 2).double().cpu()
                if ep == 1:
                    y = F.max_pool2d(y, 2, 2)
                assert (outs[1].permute(0, 3, 1, 2).double().cpu() - y).abs().max().item() < 1e-5
            del outs
    finally:
        _wino_waves(None)


def test_conv_wino4_repeated_large_launches_are_reproducible(hip):
    """launches of the benchmark's size on the four-wave kernel into NaN-filled outputs at shifting addresses: complete, bit-identical
    from run to run (its waits are counted by hand: a short count shows as a run that differs), equal to the direct kernel's"""
    torch.manual_seed(6)
    for n, h, w, ci, co, res in ((10, 640, 640, 64, 64, True), (12, 320, 320, 256, 256, False), (16, 160, 160, 512, 512, False)):
        x = torch.randn(n, h, w, ci, device='cuda')
        wt = torch.randn(co, ci, 3, 3, device='cuda') / (3.0 * ci ** 0.5)
        bias = torch.randn(co, device='cuda')
        r = torch.randn(n, h, w, co, device='cuda') if res else None
        pk = hip.conv_pack_weight(wt, 17)
        outs, pad = [], []
        for rep in range(4):
            pad.append(torch.empty(1 + 7000 * (rep + 1), device='cuda'))
            o = torch.full((n, h, w, co), float('nan'), device='cuda')
            hip.conv_nhwc(x, pk, bias, co, 3, residual=r, act=True, slope=0.1, out=o)
            outs.append(o)
        hip.check_conv_range()
        assert not torch.isnan(outs[0]).any()
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (n, h, w, ci, co)
        ref = hip.conv_nhwc(x, hip.conv_pack_weight(wt, 16), bias, co, 3, residual=r, act=True, slope=0.1)
        assert (outs[0] - ref).abs().max().item() < 3e-5
        del outs, ref, x


def test_conv_wino4_random_launches_equal_the_eight_wave_kernel():
    """tools/conv_wino4_stress.py: 48 random launches (shapes up to 128 x 128, 1-12 images, one or two sources with batch broadcast,
    ragged channel counts, every served epilogue, with and without the input scale, activations of 1e-4 ... 10) -- the four-wave
    kernel three times into NaN-filled channel slices, bit for bit against the eight-wave kernel, every fourth case against fp64"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'conv_wino4_stress.py'), '48', '7'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and '48 of 48 cases ok' in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_conv_wino4_range_flag(hip):
    """the four-wave kernel's fp16 range guard: a transform value beyond the fp16 range is an inf in the high term and makes the
    outputs it enters non-finite -- the flag is raised from the outputs (the eight-wave kernel compares the raw activations with
    16000); activations inside the range leave it down"""
    pk = hip.conv_pack_weight(torch.randn(64, 64, 3, 3, device='cuda') * 0.03, 17)
    x = torch.randn(2, 48, 48, 64, device='cuda')
    hip.conv_nhwc(x, pk, None, 64, 3)
    assert not hip.conv_range_tripped()
    x[1, 17, 33, 5] = 7.0e4
    hip.conv_nhwc(x, pk, None, 64, 3)
    assert hip.conv_range_tripped()
    hip.conv_nhwc(torch.randn(2, 48, 48, 64, device='cuda'), pk, None, 64, 3)
    assert not hip.conv_range_tripped()
    # max-pool epilogue: two activations of 4e4 two pixels apart on a tile's diagonal make ONE transform value
    # (V[0][0] = d00 - d02 - d20 + d22 = 8e4) leave the fp16 range; it enters one output of the 2 x 2 tile only, which the
    # maximum would drop as a NaN -- the guard reads the tile before the maximum
    for ep in (0, 1):
        x = torch.randn(2, 48, 48, 64, device='cuda').relu_()
        x[1, 15, 31, 5] = 4.0e4
        x[1, 17, 33, 5] = 4.0e4
        hip.conv_nhwc(x, pk, None, 64, 3, act=True, slope=0.0, epilogue=ep)
        assert hip.conv_range_tripped(), ep
        hip.conv_nhwc(torch.randn(2, 48, 48, 64, device='cuda'), pk, None, 64, 3, act=True, slope=0.0, epilogue=ep)
        assert not hip.conv_range_tripped(), ep


def test_conv_wino_range_flag_and_argument_checks(hip):
    x = torch.randn(2, 40, 40, 64, device='cuda')
    x[1, 17, 33, 5] = 2.0e4   # |B^T d B| <= 4 max|x| must stay inside fp16: the guard fires at |x| > 16000
    pk = hip.conv_pack_weight(torch.randn(64, 64, 3, 3, device='cuda') * 0.03, 17)
    hip.conv_nhwc(x, pk, None, 64, 3)
    assert hip.conv_range_tripped()
    hip.conv_nhwc(torch.randn(2, 40, 40, 64, device='cuda'), pk, None, 64, 3)
    assert not hip.conv_range_tripped()
    from mrefsr_amd._lib import MrefsrHipError
    with pytest.raises(ValueError):
        hip.conv_pack_weight(torch.randn(8, 32, 1, 1, device='cuda'), 17)            # 3x3 only
    with pytest.raises(MrefsrHipError):
        hip.conv_nhwc(torch.randn(1, 16, 16, 32, device='cuda'), hip.conv_pack_weight(torch.randn(8, 32, 3, 3, device='cuda'), 17), None, 8, 3)  # fewer than three K chunks


def test_conv_nhwc_two_sources_pre_prelu_slices(hip):
    """cat([x broadcast over K, ref]) -> conv3x3 + bias + pre (broadcast) -> PReLU, written into a channel slice"""
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    b, k, c1, c2, co, h, w = 2, 3, 16, 24, 40, 10, 37
    x = rng.standard_normal((b, c1, h, w)).astype(np.float32)
    r = rng.standard_normal((k * b, c2, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, c1 + c2, 3, 3)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    pre = rng.standard_normal((b, co, h, w)).astype(np.float32)
    a = np.float32(0.25)
    want = F.prelu(F.conv2d(torch.cat([torch.from_numpy(x).repeat(k, 1, 1, 1), torch.from_numpy(r)], 1).double(),
                            torch.from_numpy(wt).double(), torch.from_numpy(bias).double(), 1, 1)
                   + torch.from_numpy(pre).double().repeat(k, 1, 1, 1), torch.tensor([a], dtype=torch.float64))
    wide_in = torch.zeros(k * b, h, w, c2 + 8, device='cuda')
    wide_in[..., 4:4 + c2] = _nhwc(r)
    wide_out = torch.full((k * b, h, w, co + 12), 7.0, device='cuda')
    hip.conv_nhwc(_nhwc(x), hip.conv_pack_weight(dev(wt)), dev(bias), co, 3, x2=wide_in[..., 4:4 + c2], pre=_nhwc(pre), act=True,
                  slope_ptr=dev(np.array([a])), out=wide_out[..., 8:8 + co])
    got = wide_out[..., 8:8 + co].permute(0, 3, 1, 2).cpu().double()
    assert (got - want).abs().max().item() < 1e-5
    assert (wide_out[..., :8] == 7.0).all() and (wide_out[..., 8 + co:] == 7.0).all()


def test_conv_nhwc_pool_and_pixel_shuffle_epilogues(hip):
    import torch.nn.functional as F
    rng = np.random.default_rng(6)
    n, ci, co, h, w = 2, 8, 24, 12, 40
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    conv = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(bias).double(), 1, 1)
    pk = hip.conv_pack_weight(dev(wt))
    got = hip.conv_nhwc(_nhwc(x), pk, dev(bias), co, 3, act=True, slope=0.0, epilogue=1).permute(0, 3, 1, 2).cpu().double()
    assert (got - F.max_pool2d(F.relu(conv), 2, 2)).abs().max().item() < 1e-5
    got = hip.conv_nhwc(_nhwc(x), pk, dev(bias), co, 3, act=True, slope=0.1, epilogue=2).permute(0, 3, 1, 2).cpu().double()
    assert (got - F.leaky_relu(F.pixel_shuffle(conv, 2), 0.1)).abs().max().item() < 1e-5


def test_channels_last_variants_match_planar_kernels(hip):
    """pixnorm / dynagg_prep / DCN / attention core read and write [N,H,W,C] with the same results as their NCHW forms"""
    rng = np.random.default_rng(11)
    # pixnorm: identical arithmetic per pixel -> identical bits
    x = dev(rng.standard_normal((3, 256, 9, 13)).astype(np.float32))
    a = hip.pixnorm(x, want_bf16_split=True)
    b = hip.pixnorm(x.permute(0, 2, 3, 1).contiguous(), want_bf16_split=True, nhwc=True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # dynagg_prep
    bsz, dg, h, w = 3, 8, 7, 21
    om = dev(rng.standard_normal((bsz, 27 * dg, h, w)).astype(np.float32))
    bias = dev(rng.standard_normal(27 * dg).astype(np.float32))
    pre = dev(rng.standard_normal((bsz, 9, h, w, 2)).astype(np.float32))
    s1 = torch.zeros(1, dtype=torch.float64, device='cuda')
    s2 = torch.zeros(1, dtype=torch.float64, device='cuda')
    o1, m1 = hip.dynagg_prep(om, pre, dg, s1, bias)
    o2, m2 = hip.dynagg_prep(om.permute(0, 2, 3, 1).contiguous(), pre, dg, s2, bias, om_nhwc=True)
    assert torch.equal(o1, o2) and torch.equal(m1, m2)
    assert abs(s1.item() - s2.item()) <= 1e-6 * abs(s1.item())
    # DCN, channels-last in and out
    c, co = 64, 64
    xx = dev(rng.standard_normal((bsz, c, h, w)).astype(np.float32))
    wt = dev((rng.standard_normal((co, c, 3, 3)) * 0.05).astype(np.float32))
    bb = dev(rng.standard_normal(co).astype(np.float32))
    y1 = hip.dcn_fwd(xx, o1, m1, wt, bb, 1, 1, 1, 1, dg, 0.1)
    y2 = hip.dcn_fwd(xx.permute(0, 2, 3, 1).contiguous(), o1, m1, wt, bb, 1, 1, 1, 1, dg, 0.1, channels_last=True)
    assert torch.equal(y1, y2.permute(0, 3, 1, 2))
    # attention core
    for ch in (64, 128, 256):
        n, t = 2, 5
        q = dev(rng.standard_normal((n, ch, h, w)).astype(np.float32))
        emb = dev(rng.standard_normal((t * n, ch, h, w)).astype(np.float32))
        ass = dev(rng.standard_normal((t * n, 2 * ch, h, w)).astype(np.float32))
        want, _ = hip.mrattn_fwd(q, emb, ass, t, want_prob=False, t_major=True)
        got = hip.mrattn_fwd_nhwc(*(v.permute(0, 2, 3, 1).contiguous() for v in (q, emb, ass)), t)
        torch.testing.assert_close(got.permute(0, 3, 1, 2), want, rtol=1e-5, atol=2e-5)
        # q * scale formed inside the kernel (mrefsr_mrattn_fwd_nhwc_scaled_f32) = the separate pass over q, bit for bit
        ql, el, al = (v.permute(0, 2, 3, 1).contiguous() for v in (q, emb, ass))
        scale = float(ch) ** -0.5
        assert torch.equal(hip.mrattn_fwd_nhwc(ql, el, al, t, q_scale=scale), hip.mrattn_fwd_nhwc(ql * scale, el, al, t))


@pytest.mark.parametrize('c,hw', [(256, 160), (128, 320), (64, 640)])
def test_dcn_fwd_is_bit_reproducible_at_benchmark_sizes(hip, c, hw):
    """The same launch six times at the three scales of the benchmark (8 samples): identical bits.  Small
    problems never showed the gfx950 packed-multiply problem (it needs several workgroups per CU drifting
    out of phase), full-size ones showed it on 0.2-2 % of the pixels in every repetition (tools/dcn_repro.py)."""
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(8, hw, hw, c, device='cuda', generator=g)
    off = torch.randn(8, 144, hw, hw, device='cuda', generator=g) * 4
    msk = torch.rand(8, 72, hw, hw, device='cuda', generator=g)
    wgt = torch.randn(c, c, 3, 3, device='cuda', generator=g) * 0.02
    bias = torch.randn(c, device='cuda', generator=g) * 0.1
    first = hip.dcn_fwd(x, off, msk, wgt, bias, 1, 1, 1, 1, 8, 0.1, channels_last=True).clone()
    for _ in range(5):
        again = hip.dcn_fwd(x, off, msk, wgt, bias, 1, 1, 1, 1, 8, 0.1, channels_last=True)
        assert torch.equal(again, first)
    # and the values are the deformable convolution's: a strided sample of output pixels against the planar fp32 path
    ref = hip.dcn_fwd(x[:1, :, :, :].permute(0, 3, 1, 2).contiguous(), off[:1], msk[:1], wgt, bias, 1, 1, 1, 1, 8, 0.1, nhwc_gather=False)
    torch.testing.assert_close(first[:1].permute(0, 3, 1, 2), ref, rtol=1e-4, atol=2e-4)


def test_conv_nhwc_fp16_range_guard(hip):
    """terms=16 flags activations outside the fp16 range instead of silently returning inf"""
    x = torch.ones(1, 8, 8, 16, device='cuda')
    wt = dev(np.full((16, 16, 3, 3), 0.01, np.float32))
    pk = hip.conv_pack_weight(wt, 16)
    hip.conv_nhwc(x, pk, None, 16, 3)
    hip.check_conv_range()                     # in range: silent
    x[0, 3, 3, 5] = 1.0e5
    hip.conv_nhwc(x, pk, None, 16, 3)
    with pytest.raises(FloatingPointError):
        hip.check_conv_range()
    hip.check_conv_range()                     # flag was reset
    out = hip.conv_nhwc(x, hip.conv_pack_weight(wt, 6), None, 16, 3)   # the bf16 mode has no range limit
    assert torch.isfinite(out).all() and abs(out[0, 3, 3, 0].item() - (1.0e5 * 0.01 + (16 * 9 - 1) * 0.01)) < 1e-2


def test_dcn_fp16_split_range_guard(hip):
    """the channels-last DCN (fp16 two-term split) raises the same device flag as the convolution when a sampled
    column leaves the fp16 range"""
    rng = np.random.default_rng(3)
    x = dev(rng.standard_normal((1, 9, 11, 64)).astype(np.float32))
    off = dev((rng.standard_normal((1, 144, 9, 11)) * 0.5).astype(np.float32))
    msk = torch.ones(1, 72, 9, 11, device='cuda')
    wt = dev((rng.standard_normal((64, 64, 3, 3)) * 0.05).astype(np.float32))
    hip.dcn_fwd(x, off, msk, wt, None, 1, 1, 1, 1, 8, 0.1, channels_last=True)
    hip.check_conv_range()
    x[0, 4, 5, :] = 3.0e5
    hip.dcn_fwd(x, off, msk, wt, None, 1, 1, 1, 1, 8, 0.1, channels_last=True)
    with pytest.raises(FloatingPointError):
        hip.check_conv_range()
    hip.check_conv_range()


def test_conv_nhwc_randomised_shapes(hip):
    """40 seeded random configurations (odd sizes, channel tails, two sources with batch broadcast, channel-sliced
    inputs / outputs, every epilogue, both arithmetic modes) against torch's fp64 convolution"""
    import torch.nn.functional as F
    rng = np.random.default_rng(2024)
    for case in range(40):
        ks = int(rng.choice([1, 3]))
        terms = int(rng.choice([16, 6]))
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        k = int(rng.integers(1, 4))
        b = int(rng.integers(1, 3))
        two = bool(rng.integers(0, 2))
        c1 = int(rng.choice([16, 32, 48])) if two else int(rng.choice([4, 8, 12, 20, 64]))
        c2 = int(rng.choice([4, 8, 24, 40])) if two else 0
        co = int(rng.choice([3, 8, 30, 32, 64, 70, 130]))
        epi = int(rng.choice([0, 0, 1, 2]))
        if epi == 1:
            h, w = 2 * max(h // 2, 1), 2 * max(w // 2, 1)
        if epi == 2:
            co = 4 * max(co // 4, 1)
        use_pre = epi != 1 and bool(rng.integers(0, 2))
        use_res = epi == 0 and bool(rng.integers(0, 2))
        slope = float(rng.choice([0.0, 0.1]))
        act = bool(rng.integers(0, 2))
        n = k * b
        x1 = rng.standard_normal((b if two else n, c1, h, w)).astype(np.float32)            # first source batch-broadcast when two
        x2 = rng.standard_normal((n, c2, h, w)).astype(np.float32) if two else None
        wt = (rng.standard_normal((co, c1 + c2, ks, ks)) / np.sqrt(ks * ks * (c1 + c2))).astype(np.float32)
        bias = rng.standard_normal(co).astype(np.float32)
        pre = rng.standard_normal((b, co, h, w)).astype(np.float32) if use_pre else None
        res = rng.standard_normal((n, co, h, w)).astype(np.float32) if use_res else None
        tin = torch.from_numpy(x1).double()
        if two:
            tin = torch.cat([tin.repeat(k, 1, 1, 1), torch.from_numpy(x2).double()], 1)
        want = F.conv2d(tin, torch.from_numpy(wt).double(), torch.from_numpy(bias).double(), 1, ks // 2)
        if use_pre:
            want = want + torch.from_numpy(pre).double().repeat(k, 1, 1, 1)
        if act:
            want = F.leaky_relu(want, slope)
        if use_res:
            want = want + torch.from_numpy(res).double()
        if epi == 1:
            want = F.max_pool2d(want, 2, 2)
        elif epi == 2:
            want = F.pixel_shuffle(want, 2)
        # inputs / output as channel slices of wider NHWC buffers
        wide1 = torch.zeros(x1.shape[0], h, w, c1 + 4, device='cuda')
        wide1[..., :c1] = _nhwc(x1)
        xin2 = None
        if two:
            wide2 = torch.zeros(n, h, w, c2 + 8, device='cuda')
            wide2[..., 4:4 + c2] = _nhwc(x2)
            xin2 = wide2[..., 4:4 + c2]
        oshape = tuple(want.shape[i] for i in (0, 2, 3, 1))
        wide_out = torch.full(oshape[:3] + (oshape[3] + 5,), -7.0, device='cuda')
        out_view = wide_out[..., 1:1 + oshape[3]]          # unaligned channel offset and stride: the scalar store path
        got = hip.conv_nhwc(wide1[..., :c1], hip.conv_pack_weight(dev(wt), terms), dev(bias), co, ks, x2=xin2,
                            pre=_nhwc(pre) if use_pre else None, residual=_nhwc(res) if use_res else None, act=act, slope=slope,
                            epilogue=epi, out=out_view)
        err = (got.permute(0, 3, 1, 2).cpu().double() - want).abs().max().item()
        scale = max(1.0, want.abs().max().item())
        assert err <= 2e-5 * scale, (case, ks, terms, (n, c1, c2, co, h, w), epi, use_pre, use_res, err)
        assert (wide_out[..., 0] == -7.0).all() and (wide_out[..., 1 + oshape[3]:] == -7.0).all(), case
    hip.check_conv_range()


def test_attn_modulate_matches_torch(hip):
    rng = np.random.default_rng(9)
    r, m, a = (dev(rng.standard_normal((2, 6, 10, 8)).astype(np.float32)) for _ in range(3))
    want = r * torch.sigmoid(m) * 2 + a
    got = hip.attn_modulate_(r, m.clone(), a)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('terms', [16, 6, 1])
@pytest.mark.parametrize('c,dg,h,w', [(64, 8, 20, 40), (32, 4, 17, 33), (128, 8, 9, 70)])
def test_conv_dynagg_equals_convolution_then_glue(hip, terms, c, dg, h, w):
    """mrefsr_conv_dynagg_f32 (conv_offset_mask with the DynAgg glue of ref :56-73 as its epilogue, planar outputs) is
    bit-identical to mrefsr_conv_nhwc_f32 followed by mrefsr_dynagg_prep_f32 -- which test_dynagg_prep_matches_reference_glue
    pins to the reference's own DynAgg.forward"""
    gen = torch.Generator().manual_seed(c + dg + h)
    n = 3
    x = torch.randn(n, h, w, c, generator=gen).cuda()
    wt = (0.2 * torch.randn(27 * dg, c, 3, 3, generator=gen)).cuda()
    bias = torch.randn(27 * dg, generator=gen).cuda()
    pre = (4 * torch.randn(n, 9, h, w, 2, generator=gen)).round().cuda()
    packed = hip.conv_pack_weight(wt, terms)
    om = hip.conv_nhwc(x, packed, bias, 27 * dg, 3)
    s1 = torch.zeros(1, dtype=torch.float64, device='cuda')
    off1, m1 = hip.dynagg_prep(om, pre, dg, s1, None, om_nhwc=True)
    s2 = torch.zeros(1, dtype=torch.float64, device='cuda')
    off2, m2 = hip.conv_dynagg(x, packed, bias, pre, dg, s2)
    assert torch.equal(off1, off2) and torch.equal(m1, m2)
    assert abs(float(s1) - float(s2)) <= 1e-6 * float(s1)        # the same terms, summed in another order


def test_bf16_storage_kernels_equal_the_fp32_container_arithmetic(hip):
    """terms = 1 (bf16 arithmetic) on bf16 TENSORS (descriptor terms 2, DCN nhwc bit 4, *_bf16 entry points) returns exactly
    the bf16 values the fp32-container versions return: two-source convolution with pre / PReLU-slope / residual, the pooled
    and pixel-shuffled epilogues, conv_offset_mask + glue, the DCN, the attention core and the modulation"""
    gen = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=gen).bfloat16().float().cuda()      # noqa: E731  bf16-valued fp32 tensors
    b = lambda t: t.bfloat16()                                                    # noqa: E731
    n, h, w = 3, 18, 40
    x1, x2 = r(n, h, w, 32), r(1, h, w, 16)
    wt = (0.1 * torch.randn(40, 48, 3, 3, generator=gen)).cuda()
    bias, pre, res = torch.randn(40, generator=gen).cuda(), r(1, h, w, 40), r(n, h, w, 40)
    slope = torch.tensor([0.2], device='cuda')
    pk = hip.conv_pack_weight(wt, 1)
    want = hip.conv_nhwc(x1, pk, bias, 40, 3, x2=x2, pre=pre, residual=res, act=True, slope_ptr=slope)
    got = hip.conv_nhwc(b(x1), pk, bias, 40, 3, x2=b(x2), pre=b(pre), residual=b(res), act=True, slope_ptr=slope)
    assert got.dtype == torch.bfloat16 and torch.equal(got.float(), want)
    for ep in (1, 2):
        assert torch.equal(hip.conv_nhwc(b(x1), pk, bias, 40, 3, x2=b(x2), act=True, slope=0.1, epilogue=ep).float(),
                           hip.conv_nhwc(x1, pk, bias, 40, 3, x2=x2, act=True, slope=0.1, epilogue=ep))
    w1 = (0.1 * torch.randn(24, 32, 1, 1, generator=gen)).cuda()
    p1 = hip.conv_pack_weight(w1, 1)
    assert torch.equal(hip.conv_nhwc(b(x1), p1, None, 24, 1).float(), hip.conv_nhwc(x1, p1, None, 24, 1))
    img = torch.zeros(2, h, w, 8, device='cuda')
    img[..., :3] = r(2, h, w, 3)
    w3 = (0.3 * torch.randn(16, 3, 3, 3, generator=gen)).cuda()
    p3 = hip.conv_pack_weight(w3, 1)
    assert torch.equal(hip.conv_nhwc(b(img), p3, None, 16, 3).float(), hip.conv_nhwc(img[..., :4].contiguous(), p3, None, 16, 3))
    # conv_offset_mask + glue, then the DCN on its planar outputs
    c, dg = 64, 8
    feat, xin = r(n, h, w, c), r(n, h, w, c)
    wom = (0.05 * torch.randn(27 * dg, c, 3, 3, generator=gen)).cuda()
    bom = (0.1 * torch.randn(27 * dg, generator=gen)).cuda()
    prof = (3 * torch.randn(n, 9, h, w, 2, generator=gen)).round().cuda()
    pom = hip.conv_pack_weight(wom, 1)
    o32, m32 = hip.conv_dynagg(feat, pom, bom, prof, dg)
    o16, m16 = hip.conv_dynagg(b(feat), pom, bom, prof, dg)
    assert torch.equal(o32, o16) and torch.equal(m32, m16)
    wd = (0.05 * torch.randn(c, c, 3, 3, generator=gen)).cuda()
    bd = torch.randn(c, generator=gen).cuda()
    d32 = hip.dcn_fwd(xin, o32, m32, wd, bd, 1, 1, 1, 1, dg, 0.1, channels_last=True, bf16_arith=True)
    d16 = hip.dcn_fwd(b(xin), o32, m32, wd, bd, 1, 1, 1, 1, dg, 0.1, channels_last=True, bf16_arith=True)
    assert d16.dtype == torch.bfloat16 and torch.equal(d16.float(), d32)
    # attention core + modulation (their fp32 versions are followed by a rounding pass on the host)
    t = 5
    q, emb, ass = r(2, 8, 12, 64), r(2 * t, 8, 12, 64), r(2 * t, 8, 12, 128)
    a32 = hip.mrattn_fwd_nhwc(q, emb, ass, t)
    a16 = hip.mrattn_fwd_nhwc(b(q), b(emb), b(ass), t)
    assert torch.equal(a16.float(), a32.bfloat16().float())
    rf, mul, add = r(2, 8, 12, 128), r(2, 8, 12, 128), r(2, 8, 12, 128)
    m32_ = hip.attn_modulate_(rf, mul.clone(), add)
    m16_ = hip.attn_modulate_(b(rf), b(mul), b(add))
    assert torch.equal(m16_.float(), m32_.bfloat16().float())
    # pixnorm reads bf16 channels-last features
    f = r(2, 10, 12, 256)
    y32 = hip.pixnorm(f, nhwc=True, want_bf16_split=True, split='fp16', want_err=True)
    y16 = hip.pixnorm(b(f), nhwc=True, want_bf16_split=True, split='fp16', want_err=True)
    assert all(torch.equal(u, v) for u, v in zip(y32, y16))
