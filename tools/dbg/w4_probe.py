#!/usr/bin/env python3
"""where the four-wave kernel differs from the eight-wave one: per channel chunk of the input, and the spatial pattern"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mrefsr_amd import hip

def run(x, wt, co, nw):
    os.environ['MREFSR_WINO_WAVES'] = nw
    pk = hip.conv_pack_weight(wt, 17)
    return hip.conv_nhwc(x, pk, None, co, 3, act=False)

torch.manual_seed(0)
n, h, w, ci, co = 1, 32, 32, 48, 64
wt = torch.randn(co, ci, 3, 3, device='cuda') / (3.0 * ci ** 0.5)
for lo in range(0, ci, 16):
    x = torch.zeros(n, h, w, ci, device='cuda')
    x[..., lo:lo + 16] = torch.randn(n, h, w, 16, device='cuda')
    a, b = run(x, wt, co, '8'), run(x, wt, co, '4')
    d = (a - b).abs().amax(dim=-1)[0]
    print(f'chunk {lo // 16}: max diff {d.max().item():.3e}; wrong pixels {(d > 1e-4).sum().item()} of {h * w}')
    if d.max() > 1e-4:
        for y in range(h):
            print(''.join('#' if d[y, xx] > 1e-4 else '.' for xx in range(w)))
