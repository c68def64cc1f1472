#!/usr/bin/env python3
"""A/B of the DCN forward kernels at the benchmark's three scales (K*B = 40 images): the T-tile chunk-outer kernel
(dcn_fwd_pt_kernel, MREFSR_DCN_T = 2 / 4) against the one-tile kernel (MREFSR_DCN_PT=0) -- same bits required -- and the
time per launch of each, on coherent offsets (a global shift per image + small learned part: what the matching produces)
and on random ones.   python tools/dcn_ab.py [images]"""
import os
import sys

import torch

sys.path.insert(0, '.')
from mrefsr_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)


def run(env, *a):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        out = hip.dcn_fwd(*a, 1, 1, 1, 1, 8, 0.1, channels_last=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            hip.dcn_fwd(*a, 1, 1, 1, 1, 8, 0.1, channels_last=True)
        e1.record()
        torch.cuda.synchronize()
        return out, e0.elapsed_time(e1) / 3
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


bad = 0
for c, hw in ((64, 640), (128, 320), (256, 160)):
    x = torch.randn(n, hw, hw, c, device='cuda')
    wgt = torch.randn(c, c, 3, 3, device='cuda') * 0.02
    bias = torch.randn(c, device='cuda') * 0.1
    msk = torch.rand(n, 72, hw, hw, device='cuda')
    for kind in ('coherent', 'random'):
        if kind == 'coherent':
            s = hw // 160
            sh = torch.tensor([[17.0 * (i % 5 + 1) * s, -23.0 * (i % 5 + 1) * s] for i in range(n)], device='cuda')   # (y, x) per image
            off = torch.randn(n, 144, hw, hw, device='cuda') * 0.3
            off[:, 0::2] += sh[:, 0].view(n, 1, 1, 1)
            off[:, 1::2] += sh[:, 1].view(n, 1, 1, 1)
        else:
            off = torch.randn(n, 144, hw, hw, device='cuda') * 4
        ref, t_ref = run({'MREFSR_DCN_PT': '0'}, x, off, msk, wgt, bias)
        line = f'C={c:3d} {hw}x{hw} x{n} {kind:8s}: one-tile {t_ref:6.2f} ms'
        for t in ('2', '4'):
            if t == '4' and c != 64:
                continue
            out, ms = run({'MREFSR_DCN_PT': '1', 'MREFSR_DCN_T': t}, x, off, msk, wgt, bias)
            same = torch.equal(out, ref)
            bad += not same
            line += f' | T={t} {ms:6.2f} ms {"same bits" if same else "DIFFERENT (max %.3e)" % float((out - ref).abs().max())}'
        print(line, flush=True)
        del off
    del x, msk
print('FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
