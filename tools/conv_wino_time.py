#!/usr/bin/env python3
"""time of the Winograd convolution (terms 17) on a few benchmark shapes, one process per library build:
   python tools/conv_wino_time.py [lib.so ...]"""
import os
import subprocess
import sys

if len(sys.argv) == 1 or sys.argv[1] != '--child':
    for lib in (sys.argv[1:] or ['mrefsr_amd/lib/libmrefsr_hip.so']):
        env = dict(os.environ, MREFSR_HIP_LIB=lib)
        r = subprocess.run([sys.executable, __file__, '--child', lib], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-600:], flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrefsr_amd import hip  # noqa: E402


def run(n, h, w, cin, cout, residual=False, terms=17, iters=5):
    x = torch.randn(n, h, w, cin, device='cuda')
    wgt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    pw = hip.conv_pack_weight(wgt, terms=terms)
    res = torch.randn(n, h, w, cout, device='cuda') if residual else None
    out = torch.empty(n, h, w, cout, device='cuda')
    for _ in range(2):
        hip.conv_nhwc(x, pw, b, cout, 3, residual=res, act=True, slope=0.1, out=out)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        hip.conv_nhwc(x, pw, b, cout, 3, residual=res, act=True, slope=0.1, out=out)
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / iters


shapes = [(8, 640, 640, 64, 64, True), (8, 320, 320, 64, 64, True), (40, 640, 640, 64, 64, False), (8, 640, 640, 128, 128, False),
          (8, 320, 320, 256, 256, False), (8, 160, 160, 512, 512, False), (40, 640, 640, 64, 128, False)]
terms = int(os.environ.get('WINO_TERMS', '17'))
ts = [run(*s, terms=terms) for s in shapes]
print(f'{sys.argv[2]:40s}: ' + '  '.join(f'{t:.3f}' for t in ts) + f'   sum {sum(ts):.3f} ms')
