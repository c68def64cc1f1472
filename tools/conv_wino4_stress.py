#!/usr/bin/env python3
"""random launches of the four-wave Winograd kernel against the eight-wave one (same arithmetic: the same bits are required) and, every
fourth case, against an fp64 convolution: shapes, sources, epilogues, channel raggedness, batch broadcast, channel-slice outputs, the
input scale; several launches per case into NaN-filled outputs (a hand-counted wait that is one short shows as a run that differs).
    python tools/conv_wino4_stress.py [cases=80] [seed=0]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from mrefsr_amd import hip  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    h, w = 16 * int(rng.integers(1, 9)), 16 * int(rng.integers(1, 9))
    n = int(rng.integers(1, 13))
    two = rng.random() < 0.3
    c1 = 16 * int(rng.integers(1, 6)) if two else 4 * int(rng.integers(9, 50))
    c2 = 4 * int(rng.integers(5, 30)) if two else 0
    co = 64 * int(rng.integers(1, 5))
    ep = 1 if rng.random() < 0.2 else 0
    res = ep == 0 and rng.random() < 0.35
    pre = ep == 0 and not res and rng.random() < 0.25
    act = rng.random() < 0.7
    scaled = rng.random() < 0.5
    n1 = 1 if (two and rng.random() < 0.5) else n
    mag = float(10.0 ** rng.uniform(-4, 1))
    g = torch.Generator(device='cuda').manual_seed(1000 + it)
    x1 = torch.randn(n1, h, w, c1, device='cuda', generator=g) * mag
    x2 = torch.randn(n, h, w, c2, device='cuda', generator=g) * mag if two else None
    wt = torch.randn(co, c1 + c2, 3, 3, device='cuda', generator=g) / (3.0 * (c1 + c2) ** 0.5)
    bias = torch.randn(co, device='cuda', generator=g) * mag
    r = torch.randn(n, h, w, co, device='cuda', generator=g) * mag if res else None
    p = torch.randn(1 if n % 2 else 2, h, w, co, device='cuda', generator=g) * mag if pre else None
    am = torch.maximum(x1.abs().amax(), x2.abs().amax() if two else x1.new_zeros(())).reshape(1) if scaled else None
    pk = hip.conv_pack_weight(wt, 17)
    ho, wo = (h // 2, w // 2) if ep else (h, w)
    outs = {}
    for nw in ('8', '4', '4', '4'):
        os.environ['MREFSR_WINO_WAVES'] = nw
        wide = torch.full((n, ho, wo, co + 4), float('nan'), device='cuda')
        hip.conv_nhwc(x1, pk, bias, co, 3, x2=x2, residual=r, pre=p, act=act, slope=0.2, epilogue=ep, out=wide[..., :co], in_amax=am)
        outs.setdefault(nw, []).append(wide)
    os.environ.pop('MREFSR_WINO_WAVES')
    ok = all(torch.equal(outs['8'][0][..., :co], o[..., :co]) and bool(torch.isnan(o[..., co:]).all()) for o in outs['4'])
    ok = ok and not bool(torch.isnan(outs['8'][0][..., :co]).any())
    note = ''
    if it % 4 == 0 and n * h * w <= 60000:
        xin = x1.repeat(n // n1, 1, 1, 1) if n1 != n else x1
        xin = torch.cat([xin, x2], -1) if two else xin
        y = F.conv2d(xin.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), bias.double().cpu(), 1, 1)
        if pre:
            y = y + p.permute(0, 3, 1, 2).double().cpu().repeat(n // p.shape[0], 1, 1, 1)
        if act:
            y = F.leaky_relu(y, 0.2)
        if res:
            y = y + r.permute(0, 3, 1, 2).double().cpu()
        if ep:
            y = F.max_pool2d(y, 2, 2)
        err = (outs['4'][0][..., :co].permute(0, 3, 1, 2).double().cpu() - y).abs().max().item()
        scale = float(y.abs().max())
        note = f' | vs fp64 {err:.2e} of {scale:.2e} (activations ~{mag:.1e})'
        ok = ok and err <= (3e-6 if scaled else 3e-5) * max(scale, 1e-30) + (0 if scaled else 2.0 ** -22)
    bad += 0 if ok else 1
    print(f'{it:3d} N={n}({n1}) {h}x{w} {c1}+{c2}->{co} ep={ep} res={int(res)} pre={int(pre)} act={int(act)} scaled={int(scaled)}: '
          f'{"ok" if ok else "WRONG"}{note}', flush=True)
hip.check_conv_range()
print(f'{cases - bad} of {cases} cases ok')
sys.exit(1 if bad else 0)
