"""CPU: pin the oracle against vectors produced by the reference's own Python
(tests/golden/gen_golden.py).  These run everywhere (no GPU)."""
import numpy as np
import pytest

import cases
import synth
from oracle import c_api as orc


def test_corr_oracle_matches_reference_indices(golden):
    g = golden('corr_fmi')
    seen = 0
    for name, fin, fref in cases.corr_cases():
        assert str(g[name + '/chk']) == synth.checksum(fin, fref), f'RNG drift in {name}'
        idx, val = orc.feature_match_index(fin, fref)
        ref_idx, ref_val = g[name + '/idx'], g[name + '/val']
        assert idx.dtype == np.int64 and idx.shape == ref_idx.shape
        np.testing.assert_array_equal(idx, ref_idx, err_msg=name)  # bit-exact indices
        np.testing.assert_allclose(val, ref_val, rtol=0, atol=5e-6, err_msg=name)  # fp32 summation-order noise
        seen += 1
    assert seen == len(g['names'])


def test_corr_oracle_row_subset_matches_reference_indices(golden):
    """orc.feature_match_index_rows (the patch rows a test asks for, at sizes where the whole map takes a minute per pair) against the
    reference's own indices on every golden correlation case -- it shares its two helpers with the whole-map function, and is pinned
    on its own all the same"""
    g = golden('corr_fmi')
    for name, fin, fref in cases.corr_cases():
        ph = fin.shape[1] - 2
        rows = np.unique(np.array([0, ph // 3, ph // 2, ph - 1], np.int32))
        idx, val = orc.feature_match_index_rows(fin, fref, rows)
        np.testing.assert_array_equal(idx, g[name + '/idx'][rows], err_msg=name)
        np.testing.assert_allclose(val, g[name + '/val'][rows], rtol=0, atol=5e-6, err_msg=name)


def test_corr_oracle_matches_reference_at_benchmark_size(golden):
    """BASELINE configs[1] feature size (256 x 160 x 160, P = 24 964): the reference's feature_match_index (its chunked
    conv2d + max, run by tests/golden/gen_golden.py) vs the oracle -- the pin of the oracle where the benchmark runs"""
    g = golden('corr_fmi_160')
    for name, fin, fref in cases.corr160_cases():
        assert str(g[name + '/chk']) == synth.checksum(fin, fref), f'RNG drift in {name}'
        idx, val = orc.feature_match_index(fin, fref)
        np.testing.assert_array_equal(idx, g[name + '/idx'].astype(np.int64), err_msg=name)
        np.testing.assert_allclose(val[::8, ::8], g[name + '/val_sub'], rtol=0, atol=5e-6, err_msg=name)


def test_corr_oracle_exact_ties_pick_lowest_index(golden):
    g = golden('corr_fmi')
    idx = g['ties_c256_16x20/idx']
    # the ref map is 4x4 tiles of a 4x5 base: a patch at (y, x) recurs at (y+4a, x+5b); the
    # winner must be the first occurrence, i.e. inside the top-left period
    pw = 18
    assert (idx // pw < 4).all() and (idx % pw < 5).all()


def test_corr_oracle_is_argmax_of_exact_correlation():
    """independent check of the Gram restatement: fp64 correlation in the reference's own
    operation order (normalise ref patch, then dot) has the oracle's pick as its maximum, up to
    fp32 rounding."""
    name, fin, fref = next(c for c in cases.corr_cases() if c[0] == 'rand_c256_12x14')
    yin, _ = orc.pixnorm(fin)
    yref, _ = orc.pixnorm(fref)
    idx, _ = orc.feature_match_index(fin, fref)
    P = idx.size
    for q in range(0, P, 7):
        best = orc.corr_pair_f64(yin, yref, q, int(idx.flat[q]))
        allv = [orc.corr_pair_f64(yin, yref, q, r) for r in range(P)]
        assert best >= max(allv) - 1e-6


def test_general_feature_match_index_oracle_matches_reference(golden):
    """patch sizes 1..7, strides 1..3, input / reference maps of different sizes, with and without the two normalisations:
    the oracle's indices equal the reference's own feature_match_index (ref_map_util.py:26-86) on every case"""
    g = golden('fmi_general')
    for name, fin, fref, kw in cases.fmi_general_cases():
        assert str(g[name + '/chk']) == synth.checksum(fin, fref)
        idx, val = orc.feature_match_index_generic(fin, fref, **kw)
        np.testing.assert_array_equal(idx, g[name + '/idx'], err_msg=name)
        np.testing.assert_allclose(val, g[name + '/val'], rtol=2e-6, atol=1e-7, err_msg=name)
    # patch 3 / stride 1 / equal sizes: the general restatement returns the bits of the path's restatement
    fin, fref = synth.randn('fmi/eq/in', (64, 11, 13)), synth.randn('fmi/eq/ref', (64, 11, 13))
    a = orc.feature_match_index_generic(fin, fref, 3, 1, 1, True, True)
    yin, _ = orc.pixnorm(fin)
    yref, _ = orc.pixnorm(fref)
    a = orc.feature_match_index_generic(yin, yref, 3, 1, 1, True, True)
    b = orc.feature_match_index(fin, fref)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_offsets_oracle_matches_reference(golden):
    g = golden('corrgen')
    f1 = synth.randn('corrgen/f1', (2, 256, 10, 12))
    f2 = synth.randn('corrgen/f2', (2, 256, 10, 12))
    for b in range(2):
        idx, _ = orc.feature_match_index(f1[b], f2[b])
        o1, o2, o4 = orc.offsets_from_idx(idx, 10, 12)
        np.testing.assert_array_equal(o1, g['pre_relu3_1'][b])
        np.testing.assert_array_equal(o2, g['pre_relu2_1'][b])
        np.testing.assert_array_equal(o4, g['pre_relu1_1'][b])


def test_upfirdn2d_oracle_matches_reference_native(golden):
    g = golden('metrics_ops')
    for i, (u, d, p0, p1, ks) in enumerate(g['up_cases']):
        x, k, ref = g[f'up_x{i}'], g[f'up_k{i}'], g[f'up_out{i}']
        n, c, h, w = x.shape
        out = orc.upfirdn2d(x.reshape(n * c, h, w, 1), k, u, u, d, d, p0, p1, p0, p1)
        np.testing.assert_allclose(out.reshape(ref.shape), ref, rtol=1e-5, atol=1e-5)


def test_dcn_oracle_c_vs_torch_formulation():
    """two independent restatements of the vendored DCN spec agree (fwd + every gradient)."""
    import torch
    from oracle import dcn_torch
    rng = np.random.default_rng(0)
    for (b, c, h, w, co, dg, groups, stride, pad, dil) in [(2, 8, 7, 6, 8, 4, 1, 1, 1, 1), (1, 8, 9, 8, 4, 2, 2, 2, 1, 1),
                                                          (1, 4, 8, 8, 4, 1, 1, 1, 2, 2)]:
        x = rng.standard_normal((b, c, h, w)).astype(np.float32)
        wgt = rng.standard_normal((co, c // groups, 3, 3)).astype(np.float32) * 0.3
        bias = rng.standard_normal(co).astype(np.float32)
        ho = (h + 2 * pad - (dil * 2 + 1)) // stride + 1
        wo = (w + 2 * pad - (dil * 2 + 1)) // stride + 1
        off = (rng.standard_normal((b, dg * 18, ho, wo)) * 3).astype(np.float32)
        msk = rng.random((b, dg * 9, ho, wo)).astype(np.float32)
        out_c = orc.dcnv2_fwd(x, off, msk, wgt, bias, stride, pad, dil, groups, dg)
        tx, toff, tm, tw, tb = (torch.tensor(a, requires_grad=True) for a in (x, off, msk, wgt, bias))
        out_t = dcn_torch.modulated_deform_conv2d(tx, toff, tm, tw, tb, stride, pad, dil, groups, dg)
        np.testing.assert_allclose(out_c, out_t.detach().numpy(), rtol=1e-4, atol=1e-4)
        gout = rng.standard_normal(out_c.shape).astype(np.float32)
        out_t.backward(torch.tensor(gout))
        gx, goff, gm, gw, gb = orc.dcnv2_bwd(x, off, msk, wgt, gout, stride, pad, dil, groups, dg)
        for a, t in ((gx, tx), (goff, toff), (gm, tm), (gw, tw), (gb, tb)):
            np.testing.assert_allclose(a, t.grad.numpy(), rtol=2e-4, atol=2e-4)


def test_mrattn_oracle_vs_reference_formulation():
    """orc_mrattn_* vs the literal permute/matmul/softmax formulation of
    ref_mrapa_restoration_arch.py:321-335 in torch."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(1)
    n, t, c, h, w = 2, 3, 8, 5, 6
    q = torch.tensor(rng.standard_normal((n, c, h, w)).astype(np.float32), requires_grad=True)
    emb = torch.tensor(rng.standard_normal((n, t, c, h, w)).astype(np.float32), requires_grad=True)
    ass = torch.tensor(rng.standard_normal((n, t, 2 * c, h, w)).astype(np.float32), requires_grad=True)
    et = q.permute(0, 2, 3, 1).unsqueeze(3).contiguous().flatten(0, 2)
    e2 = emb.permute(0, 3, 4, 2, 1).contiguous().flatten(0, 2)
    a2 = ass.permute(0, 3, 4, 1, 2).contiguous().flatten(0, 2)
    prob = F.softmax(torch.matmul(et, e2), dim=2)
    refs = torch.matmul(prob, a2).squeeze(1).unflatten(0, (n, h, w)).permute(0, 3, 1, 2).contiguous()
    out, _ = orc.mrattn_fwd(q.detach().numpy(), emb.detach().numpy(), ass.detach().numpy())
    np.testing.assert_allclose(out, refs.detach().numpy(), rtol=1e-5, atol=1e-5)
    g = rng.standard_normal(out.shape).astype(np.float32)
    refs.backward(torch.tensor(g))
    gq, gemb, gass = orc.mrattn_bwd(q.detach().numpy(), emb.detach().numpy(), ass.detach().numpy(), g)
    np.testing.assert_allclose(gq, q.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gemb, emb.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gass, ass.grad.numpy(), rtol=1e-4, atol=1e-5)
