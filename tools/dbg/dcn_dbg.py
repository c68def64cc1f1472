import os, sys, torch
sys.path.insert(0, '.')
from mrefsr_amd import hip
torch.manual_seed(0)
for (n, c, h, w) in ((1, 64, 16, 16), (1, 64, 32, 32), (1, 64, 9, 11), (1, 128, 16, 16)):
    x = torch.randn(n, h, w, c, device='cuda')
    wgt = torch.randn(c, c, 3, 3, device='cuda') * 0.02
    bias = torch.randn(c, device='cuda') * 0.1
    msk = torch.rand(n, 72, h, w, device='cuda')
    off = torch.randn(n, 144, h, w, device='cuda') * 2
    outs = {}
    for name, env in (('old', {'MREFSR_DCN_PT': '0'}), ('t2', {'MREFSR_DCN_PT': '1', 'MREFSR_DCN_T': '2'}), ('t4', {'MREFSR_DCN_PT': '1', 'MREFSR_DCN_T': '4'})):
        os.environ.update(env)
        outs[name] = hip.dcn_fwd(x, off, msk, wgt, bias, 1, 1, 1, 1, 8, 0.1, channels_last=True).clone()
    for k in ('t2', 't4'):
        d = (outs[k] - outs['old']).abs()
        px = (d.amax(dim=3) > 0)[0]
        print((n, c, h, w), k, 'max', float(d.max()), 'pixels differing', int(px.sum()), 'of', h * w, 'channels differing', int((d.amax(dim=(0, 1, 2)) > 0).sum()))
        if px.any():
            idx = px.flatten().nonzero().flatten()
            print('   first/last differing pixel index', int(idx[0]), int(idx[-1]), ' count per 64-tile:', torch.bincount(idx // 64).tolist()[:12])
    # which taps matter: zero all but one tap of weights
    for tap in range(9):
        keep = torch.zeros_like(wgt); keep[:, :, tap // 3, tap % 3] = 1
        os.environ.update({'MREFSR_DCN_PT': '0'}); a = hip.dcn_fwd(x, off, msk, wgt * keep, bias, 1, 1, 1, 1, 8, 0.1, channels_last=True).clone()
        os.environ.update({'MREFSR_DCN_PT': '1', 'MREFSR_DCN_T': '2'}); b2 = hip.dcn_fwd(x, off, msk, wgt * keep, bias, 1, 1, 1, 1, 8, 0.1, channels_last=True)
        print('   tap', tap, 'max diff', float((a - b2).abs().max()))
