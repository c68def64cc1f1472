#!/usr/bin/env python3
"""time three layer shapes on the four-wave Winograd kernel of the loaded library (ablation builds: -DW4_ABL=n, results wrong)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrefsr_amd import hip  # noqa: E402
os.environ['MREFSR_WINO_WAVES'] = '4'


def t(n, h, w, ci, co, res):
    x = torch.randn(n, h, w, ci, device='cuda')
    wt = torch.randn(co, ci, 3, 3, device='cuda') / (3.0 * ci ** 0.5)
    bias = torch.randn(co, device='cuda')
    r = torch.randn(n, h, w, co, device='cuda') if res else None
    pk = hip.conv_pack_weight(wt, 17)
    out = torch.empty(n, h, w, co, device='cuda')
    f = lambda: hip.conv_nhwc(x, pk, bias, co, 3, residual=r, act=True, slope=0.1, out=out)  # noqa: E731
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for a, b in evs:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


print(sys.argv[1] if len(sys.argv) > 1 else '', ' '.join(f'{t(*s):.3f}' for s in ((8, 640, 640, 64, 64, True), (8, 320, 320, 256, 256, False), (8, 160, 160, 512, 512, False))), 'ms  (640^2 64->64 res | 320^2 256->256 | 160^2 512->512)', flush=True)
