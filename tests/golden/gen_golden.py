#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON (imported read-only from
/root/reference through tests/golden/_refimport.py) on synthetic inputs.

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
The fixtures are data (inputs or their checksums + expected outputs); no reference source travels.
Weights are a pure function of the reference's state-dict keys/shapes (tests/golden/synth.py), so
the tests also pin state-dict compatibility.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport as R  # noqa: E402
import synth  # noqa: E402
import cases as cases_mod  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB')


def spec_of(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


def load_synth(module, seed=0):
    spec = spec_of(module)
    sd = synth.state_dict(spec, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return spec


def spec_arrays(spec):
    return dict(spec_keys=np.array([k for k, _ in spec]),
                spec_shapes=np.array([' '.join(map(str, s)) for _, s in spec]))


# ---------------------------------------------------------------------------------------------
def gen_corr():
    """ref_map_util.feature_match_index (+ the F.normalize of corres_generation_arch.py:57-59)."""
    rmu = R.ref_module('basicsr.archs.ref_map_util')
    cases = OrderedDict()

    def run(name, fin, fref):
        c, h, w = fin.shape
        a, b = torch.from_numpy(fin), torch.from_numpy(fref)
        an = F.normalize(a.reshape(c, -1), dim=0).view(c, h, w)
        bn = F.normalize(b.reshape(c, -1), dim=0).view(c, h, w)
        idx, val = rmu.feature_match_index(an, bn, patch_size=3, input_stride=1, ref_stride=1,
                                           is_norm=True, norm_input=True)
        cases[name] = (fin, fref, idx.numpy(), val.numpy())

    for name, fin, fref in cases_mod.corr_cases():
        run(name, fin, fref)
    out = {}
    for name, (fin, fref, idx, val) in cases.items():
        out[name + '/chk'] = np.array(synth.checksum(fin, fref))
        out[name + '/idx'] = idx
        out[name + '/val'] = val
        if fin.size <= 256 * 12 * 14:  # keep the raw inputs of the smallest cases too
            out[name + '/fin'], out[name + '/fref'] = fin, fref
    out['names'] = np.array(list(cases.keys()))
    save('corr_fmi', **out)


def gen_corr160():
    """feature_match_index at the benchmark's full feature size (configs[1]: 160 x 160, 256 channels, P = 24 964) through the
    REFERENCE (its own chunked conv2d + max, ref_map_util.py:54-84): indices stored as int32 (values < 2^15), inputs as
    checksums of the synthetic recipe (cases.corr160_cases)."""
    rmu = R.ref_module('basicsr.archs.ref_map_util')
    out, names = {}, []
    for name, fin, fref in cases_mod.corr160_cases():
        c, h, w = fin.shape
        a, b = torch.from_numpy(fin), torch.from_numpy(fref)
        an = F.normalize(a.reshape(c, -1), dim=0).view(c, h, w)
        bn = F.normalize(b.reshape(c, -1), dim=0).view(c, h, w)
        idx, val = rmu.feature_match_index(an, bn, patch_size=3, input_stride=1, ref_stride=1, is_norm=True, norm_input=True)
        out[name + '/chk'] = np.array(synth.checksum(fin, fref))
        out[name + '/idx'] = idx.numpy().astype(np.int32)
        out[name + '/val_sub'] = val.numpy()[::8, ::8].copy()
        names.append(name)
        print('   ', name, 'distinct matches', len(np.unique(idx.numpy())))
    out['names'] = np.array(names)
    save('corr_fmi_160', **out)


def gen_fmi_general():
    """ref_map_util.feature_match_index with patch sizes / strides / map sizes other than the shipped 3 / 1 / equal"""
    rmu = R.ref_module('basicsr.archs.ref_map_util')
    out, names = {}, []
    for name, fin, fref, kw in cases_mod.fmi_general_cases():
        idx, val = rmu.feature_match_index(torch.from_numpy(fin), torch.from_numpy(fref), **kw)
        out[name + '/chk'] = np.array(synth.checksum(fin, fref))
        out[name + '/idx'] = idx.numpy()
        out[name + '/val'] = val.numpy()
        names.append(name)
    out['names'] = np.array(names)
    save('fmi_general', **out)


def gen_corrgen():
    """CorrespondenceGenerationArch.forward: idx -> flow -> 27 shifted planes (+ VGG19 taps)."""
    m = R.ref_module('basicsr.archs.corres_generation_arch')
    net = m.CorrespondenceGenerationArch(patch_size=3, stride=1,
                                         vgg_layer_list=['relu1_1', 'relu2_1', 'relu3_1'], vgg_type='vgg19')
    spec = load_synth(net)
    b, h, w = 2, 10, 12
    f1 = synth.randn('corrgen/f1', (b, 256, h, w))
    f2 = synth.randn('corrgen/f2', (b, 256, h, w))
    img = np.stack([synth.image(f'corrgen/img{i}', 3, 4 * h, 4 * w) for i in range(b)])
    pre, feat = net({'dense_features1': torch.from_numpy(f1), 'dense_features2': torch.from_numpy(f2)},
                    torch.from_numpy(img))
    save('corrgen', chk=np.array(synth.checksum(f1, f2, img)), **spec_arrays(spec),
         pre_relu3_1=pre['relu3_1'].numpy(), pre_relu2_1=pre['relu2_1'].numpy(),
         pre_relu1_1=pre['relu1_1'].numpy(),
         feat_relu3_1=feat['relu3_1'].numpy(), feat_relu2_1_b0=feat['relu2_1'][0].numpy(),
         feat_relu1_1_b0c8=feat['relu1_1'][0, :8].numpy())


def gen_extractor():
    m = R.ref_module('basicsr.archs.contras_multi_extractor_arch')
    net = m.ContrasMultiExtractorSep()
    spec = load_synth(net)
    img1 = synth.image('extr/img1', 3, 16, 24)[None]
    refs = [synth.image(f'extr/ref{k}', 3, 16, 24)[None] for k in range(2)]
    out = net(torch.from_numpy(img1), [torch.from_numpy(r) for r in refs])
    save('extractor', chk=np.array(synth.checksum(img1, *refs)), **spec_arrays(spec),
         dense_features1=out[0]['dense_features1'].numpy(),
         dense_features2_0=out[0]['dense_features2'].numpy(),
         dense_features2_1=out[1]['dense_features2'].numpy())
    v = R.ref_module('basicsr.archs.vgg_arch')
    vg = v.VGGFeatureExtractor(layer_name_list=['relu1_1', 'relu2_1', 'relu3_1'], vgg_type='vgg19')
    vspec = load_synth(vg)
    o = vg(torch.from_numpy(img1))
    save('vggfeat', chk=np.array(synth.checksum(img1)), **spec_arrays(vspec),
         relu1_1=o['relu1_1'].numpy(), relu2_1=o['relu2_1'].numpy(), relu3_1=o['relu3_1'].numpy())


def gen_dynagg():
    """DynAgg.forward glue (ref_mrapa_restoration_arch.py:45-76) around the DCN call; the DCN
    arithmetic itself is oracle/dcn_torch.py (mmcv absent) -- the call arguments are captured."""
    m = R.ref_module('basicsr.archs.ref_mrapa_restoration_arch')
    captured = {}
    orig = m.modulated_deform_conv2d

    def spy(x, offset, mask, *a):
        captured['offset'], captured['mask'] = offset.numpy().copy(), mask.numpy().copy()
        return orig(x, offset, mask, *a)

    m.modulated_deform_conv2d = spy
    try:
        net = m.DynAgg(64, 64, 3, stride=1, padding=1, dilation=1, deform_groups=8, extra_offset_mask=True)
        spec = load_synth(net)
        b, h, w = 2, 9, 11
        x0 = synth.randn('dynagg/x0', (b, 64, h, w))
        x1 = synth.randn('dynagg/x1', (b, 64, h, w))
        pre = (synth.randn('dynagg/pre', (b, 9, h, w, 2)) * 3).round().astype(np.float32)
        out = net([torch.from_numpy(x0), torch.from_numpy(x1)], torch.from_numpy(pre))
    finally:
        m.modulated_deform_conv2d = orig
    save('dynagg', chk=np.array(synth.checksum(x0, x1, pre)), **spec_arrays(spec), out=out.numpy(),
         dcn_offset=captured['offset'], dcn_mask=captured['mask'])


def gen_fusion():
    m = R.ref_module('basicsr.archs.ref_mrapa_restoration_arch')
    net = m.MRAPAFusion(nf=64, ref_nf=64)
    spec = load_synth(net)
    n, t, h, w = 2, 3, 10, 13
    target = synth.randn('fusion/target', (n, 64, h, w))
    refs = [synth.randn(f'fusion/ref{k}', (n, 64, h, w)) for k in range(t)]
    out = net(torch.from_numpy(target), [torch.from_numpy(r) for r in refs])
    save('mrapa_fusion', chk=np.array(synth.checksum(target, *refs)), **spec_arrays(spec), out=out.numpy())


def gen_singleref():
    """RestorationNet (single-reference C2-Matching net, ref_restoration_arch.py:101-259) fed with
    the reference's own CorrespondenceGenerationArch outputs."""
    m = R.ref_module('basicsr.archs.ref_restoration_arch')
    cg = R.ref_module('basicsr.archs.corres_generation_arch')
    ex = R.ref_module('basicsr.archs.contras_extractor_arch')
    net, mp, ext = m.RestorationNet(ngf=64, n_blocks=16, groups=8), cg.CorrespondenceGenerationArch(
        patch_size=3, stride=1, vgg_layer_list=['relu1_1', 'relu2_1', 'relu3_1'], vgg_type='vgg19'), ex.ContrasExtractorSep()
    spec, mspec, espec = load_synth(net), load_synth(mp), load_synth(ext)
    s = synth.sr_sample('singleref', 1, 12, 16)
    lq, up, ref = (torch.from_numpy(s[k][None]) for k in ('img_in_lq', 'img_in_up', 'img_ref_list'))
    ref = ref[:, 0]
    feats = ext(up, ref)
    pre, rf = mp(feats, ref)
    out = net(lq, pre, rf)
    arrays = dict(out=out.numpy(), img_in_lq=lq.numpy(), img_in_up=up.numpy(), img_ref=ref.numpy())
    for name, sp in (('net', spec), ('map', mspec), ('ext', espec)):
        sa = spec_arrays(sp)
        arrays[f'{name}_spec_keys'], arrays[f'{name}_spec_shapes'] = sa['spec_keys'], sa['spec_shapes']
    save('singleref', **arrays)


def gen_datasets():
    """the reference's own MultiRefCUFEDSet / MultiRefMegaDepthDataset on synthetic PNG / csv files
    (cv2.imread / cvtColor / flip and mmcv.impad restated in _refimport.py: both are absent here)"""
    import random
    import tempfile
    import make_dataset_files as mk
    m = R.ref_module('basicsr.data.multi_ref_dataset')
    arrays = {}
    with tempfile.TemporaryDirectory() as td:
        opt = mk.make_cufed(os.path.join(td, 'cufed'))
        ds = m.MultiRefCUFEDSet(opt)
        arrays['cufed_len'] = np.array(len(ds))
        for i in range(len(ds)):
            d = ds[i]
            for k in ('img_in', 'img_in_lq', 'img_in_up', 'img_ref_list', 'img_ref_lq_list', 'img_ref_up_list'):
                t = d[k].numpy()
                arrays[f'cufed{i}/{k}/shape'] = np.array(t.shape)
                arrays[f'cufed{i}/{k}/chk'] = np.array(synth.checksum(t))
                arrays[f'cufed{i}/{k}/corner'] = t[..., :6, :6].copy()
            arrays[f'cufed{i}/original_size'] = np.array(d['original_size'])
            arrays[f'cufed{i}/lq_name'] = np.array(os.path.basename(d['lq_path']))
        opt = mk.make_megadepth(os.path.join(td, 'mega'))
        ds = m.MultiRefMegaDepthDataset(opt)
        arrays['mega_len'] = np.array(len(ds))
        for i in range(len(ds)):
            for seed in (0, 1, 2, 3):
                random.seed(100 * i + seed)
                d = ds[i]
                for k in ('img_in', 'img_in_lq', 'img_in_up', 'img_ref_list', 'img_ref_lq_list', 'img_ref_up_list'):
                    arrays[f'mega{i}s{seed}/{k}'] = d[k].numpy()
    save('datasets', **arrays)


def gen_singleref_dataset():
    """the reference's own SingleRefMegaDepthDataset on the same synthetic csv / PNG files"""
    import random
    import tempfile
    import make_dataset_files as mk
    m = R.ref_module('basicsr.data.single_ref_dataset')
    arrays = {}
    with tempfile.TemporaryDirectory() as td:
        opt = mk.make_megadepth(os.path.join(td, 'mega'))
        opt['type'] = 'SingleRefMegaDepthDataset'
        ds = m.SingleRefMegaDepthDataset(opt)
        arrays['len'] = np.array(len(ds))
        for i in range(len(ds)):
            for seed in (0, 1, 2, 3):
                random.seed(100 * i + seed)
                np.random.seed(100 * i + seed)
                d = ds[i]
                for k in ('img_in', 'img_in_lq', 'img_in_up', 'img_ref', 'img_ref_lq', 'img_ref_up'):
                    arrays[f's{i}s{seed}/{k}'] = d[k].numpy()
    save('singleref_dataset', **arrays)


def _build_model(is_train, b, k, lr_h, lr_w, key='e2e'):
    """The reference's own MultiRefRestorationModel on CPU (num_gpu 0; the hard-coded .cuda() of
    multi_ref_restoration_model.py:27 made a no-op), synthetic weights in all three nets."""
    mm = R.ref_module('basicsr.models.multi_ref_restoration_model')
    opt = OrderedDict(
        name='golden', model_type='MultiRefRestorationModel', scale=4, crop_border=4, num_gpu=0,
        manual_seed=10, is_train=is_train, dist=False, rank=0,
        network_g=dict(type='MRAPARestorationNet', ngf=64, n_blocks=16, groups=8),
        network_map=dict(type='CorrespondenceGenerationArch', patch_size=3, stride=1,
                         vgg_layer_list=['relu1_1', 'relu2_1', 'relu3_1'], vgg_type='vgg19'),
        network_extractor=dict(type='ContrasMultiExtractorSep'),
        path=dict(pretrain_network_g=None, pretrain_network_feature_extractor=None, strict_load=True),
        train=dict(lr_g=1e-4, lr_offset=1e-4, lr_relu2_offset=1e-5, lr_relu3_offset=1e-6, weight_decay_g=0,
                   beta_g=[0.9, 0.999], scheduler=dict(type='MultiStepLR', milestones=[300000, 400000], gamma=0.5),
                   total_iter=255000, warmup_iter=-1, net_g_pretrain_steps=0, pixel_criterion='L1Loss',
                   pixel_weight=1.0),
        val=dict(save_img=False))
    saved = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, *a, **kw: self
    try:
        model = mm.MultiRefRestorationModel(opt)
    finally:
        torch.nn.Module.cuda = saved
    specs = {}
    for name in ('net_g', 'net_extractor', 'net_map'):
        specs[name] = load_synth(getattr(model, name))
    samples = [synth.sr_sample(f'{key}/s{i}', k, lr_h, lr_w) for i in range(b)]
    data = {key: torch.from_numpy(np.stack([s[key] for s in samples])) for key in samples[0]}
    return model, specs, data


def gen_e2e(b=2, k=2, lr_h=12, lr_w=16, name='e2e', store_inputs=True, key='e2e'):
    torch.set_grad_enabled(True)
    model, specs, data = _build_model(True, b, k, lr_h, lr_w, key)
    model.feed_data(data)
    # forward (test(): eval, no_grad) -- multi_ref_restoration_model.py:281-294
    model.test()
    out_test = model.output.detach().numpy().copy()
    # matching indices per ref (pin of the index path inside the full pipeline)
    rmu = R.ref_module('basicsr.archs.ref_map_util')
    with torch.no_grad():
        feats = model.net_extractor(model.match_img_in, model.img_ref_list)
        idxs = []
        for f in feats:
            per_b = []
            for i in range(b):
                a, r = f['dense_features1'][i], f['dense_features2'][i]
                c, h, w = a.shape
                a = F.normalize(a.reshape(c, -1), dim=0).view(c, h, w)
                r = F.normalize(r.reshape(c, -1), dim=0).view(c, h, w)
                per_b.append(rmu.feature_match_index(a, r, 3, 1, 1, True, True)[0].numpy())
            idxs.append(np.stack(per_b))
    # PSNR protocol of nondist_validation (:339-365) on sample 0, rgb2bgr off (cv2 absent; PSNR is
    # invariant to the channel swap)
    iu = R.ref_module('basicsr.utils.img_util')
    ps = R.ref_module('basicsr.metrics.psnr_ssim')
    # NB tensor2img clamps its argument IN PLACE (img_util.py:66 clamp_): hand it copies
    sr_img = iu.tensor2img(torch.from_numpy(out_test[:1].copy()), rgb2bgr=False)
    gt_img = iu.tensor2img(data['img_in'][:1].clone(), rgb2bgr=False)
    psnr = ps.calculate_psnr(sr_img, gt_img, crop_border=4, test_y_channel=False)
    # one optimisation step -- :197-279 with net_g_pretrain_steps = 0 -> L1 branch
    model.optimize_parameters(1)
    loss = float(model.log_dict['l_g_pix'])
    net_g = model.net_g
    names, gsum, gabs, psum = [], [], [], []
    for n, p in net_g.named_parameters():
        names.append(n)
        g = p.grad.detach().double() if p.grad is not None else torch.zeros(1, dtype=torch.float64)
        gsum.append(float(g.sum()))
        gabs.append(float(g.abs().sum()))
        psum.append(float(p.detach().double().sum()))
    groups = [[float(g['lr']), len(g['params'])] for g in model.optimizer_g.param_groups]
    torch.set_grad_enabled(False)
    arrays = dict(b=np.array(b), k=np.array(k), lr_hw=np.array([lr_h, lr_w]), key=np.array(key),
                  max_idx=np.stack(idxs).astype(np.int32 if not store_inputs else np.int64), psnr_s0=np.array(psnr), sr_img_s0=sr_img,
                  loss=np.array(loss), param_names=np.array(names), grad_sum=np.array(gsum),
                  grad_abs=np.array(gabs), param_sum_after=np.array(psum), opt_groups=np.array(groups))
    if store_inputs:
        arrays.update(img_in_lq=data['img_in_lq'].numpy(), img_in_up=data['img_in_up'].numpy(),
                      img_ref_list=data['img_ref_list'].numpy(), img_in=data['img_in'].numpy(), out_test=out_test)
    else:
        # larger cases: the inputs are the synthetic recipe synth.sr_sample(f'{key}/s{i}', ...) (checksum kept), the output is kept
        # in full for sample 0 and on a stride-4 grid + per-sample sums for the rest
        arrays.update(chk=np.array(synth.checksum(*[data[n].numpy() for n in ('img_in_lq', 'img_in_up', 'img_ref_list', 'img_in')])),
                      out_test_s0=out_test[0], out_test_sub=out_test[:, :, ::4, ::4].copy(),
                      out_test_sum=out_test.astype(np.float64).sum(axis=(1, 2, 3)),
                      out_test_abs=np.abs(out_test.astype(np.float64)).sum(axis=(1, 2, 3)),
                      base_resid_max=np.array(float((torch.from_numpy(out_test) - F.interpolate(
                          data['img_in_lq'], None, 4, 'bilinear', False)).abs().max())))
    for nm, spec in specs.items():
        sa = spec_arrays(spec)
        arrays[f'{nm}_spec_keys'], arrays[f'{nm}_spec_shapes'] = sa['spec_keys'], sa['spec_shapes']
    save(name, **arrays)


def gen_e2e_c0():
    """BASELINE configs[0]: 1-ref 4x SR, LR 40 x 40, B = 1, the reference's CPU forward (+ one training step for free)"""
    gen_e2e(1, 1, 40, 40, 'e2e_c0', store_inputs=False, key='e2e_c0')


def gen_e2e_c2():
    """BASELINE configs[2], per-GPU shape: B = 4, K = 5, LR 40 x 40 -- test() output, loss, per-parameter gradient
    fingerprints and post-Adam parameter sums of the reference's own optimize_parameters (multi_ref_restoration_model.py:197-279)"""
    gen_e2e(4, 5, 40, 40, 'e2e_c2', store_inputs=False, key='e2e_c2')


def gen_metrics_ops():
    ps = R.ref_module('basicsr.metrics.psnr_ssim')
    iu = R.ref_module('basicsr.utils.img_util')
    a = synth.rand('psnr/a', (1, 3, 24, 20))
    b_ = np.clip(a + synth.randn('psnr/n', a.shape, 0, 0.05), -0.2, 1.2).astype(np.float32)
    ia = iu.tensor2img(torch.from_numpy(a), rgb2bgr=False)
    ib = iu.tensor2img(torch.from_numpy(b_), rgb2bgr=False)
    vals = [ps.calculate_psnr(ia, ib, crop_border=cb) for cb in (0, 4)]
    up = R.ref_module('basicsr.ops.upfirdn2d.upfirdn2d')
    cases = [(1, 1, (0, 0), 3), (2, 1, (2, 1), 4), (1, 2, (1, 1), 4), (2, 2, (1, 2), 3), (1, 1, (-1, 2), 2),
             (3, 2, (2, 2), 5)]
    arrays = dict(psnr_a=a, psnr_b=b_, img_a=ia, img_b=ib, psnr_vals=np.array(vals),
                  up_cases=np.array([[u, d, p[0], p[1], ks] for u, d, p, ks in cases]))
    for i, (u, d, p, ks) in enumerate(cases):
        x = synth.randn(f'upfirdn/x{i}', (2, 3, 9, 11))
        k = synth.rand(f'upfirdn/k{i}', (ks, ks))
        arrays[f'up_x{i}'], arrays[f'up_k{i}'] = x, k
        arrays[f'up_out{i}'] = up.upfirdn2d_native(torch.from_numpy(x), torch.from_numpy(k), u, u, d, d,
                                                   p[0], p[1], p[0], p[1]).numpy()
    save('metrics_ops', **arrays)


if __name__ == '__main__':
    assert R.available(), 'reference tree not present: run in the build container'
    R.install()
    which = sys.argv[1:] or ['corr', 'corr160', 'e2e_c0', 'e2e_c2', 'corrgen', 'extractor', 'dynagg', 'fusion', 'e2e', 'singleref', 'datasets', 'singleref_dataset',
                             'metrics_ops', 'fmi_general']
    for w in which:
        print(f'[{w}]')
        globals()['gen_' + w]()
