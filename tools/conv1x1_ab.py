#!/usr/bin/env python3
"""1 x 1 convolution launches of the benchmark step through conv1x1_kernel and through conv_nhwc_kernel (MREFSR_CONV1X1=0): bits and time.
    python tools/conv1x1_ab.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrefsr_amd import hip  # noqa: E402

SHAPES = [(8, 640, 640, 192, 128), (8, 320, 320, 320, 256), (8, 640, 640, 192, 64), (8, 160, 160, 576, 512), (8, 640, 640, 64, 64),
          (8, 320, 320, 320, 64), (8, 320, 320, 64, 128), (8, 160, 160, 576, 64), (8, 160, 160, 64, 256)]


def run(x, pk, bias, cout, flag, iters):
    os.environ['MREFSR_CONV1X1'] = flag
    out = hip.conv_nhwc(x, pk, bias, cout, 1, act=True, slope=0.1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hip.conv_nhwc(x, pk, bias, cout, 1, act=True, slope=0.1, out=out)
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / iters


tot = [0.0, 0.0]
for n, h, w, cin, cout in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(n, h, w, cin, device='cuda')
    pk = hip.conv_pack_weight(torch.randn(cout, cin, 1, 1, device='cuda') * 0.05, 16)
    bias = torch.randn(cout, device='cuda')
    o0, t0 = run(x, pk, bias, cout, '0', 10)
    o1, t1 = run(x, pk, bias, cout, '1', 10)
    o0b, t0b = run(x, pk, bias, cout, '0', 10)
    gb = (x.numel() + o1.numel()) * 4 / 1e9
    tot[0] += min(t0, t0b)
    tot[1] += t1
    print(f'{n}x{h}x{w} {cin:4d}->{cout:4d}  direct {min(t0, t0b):7.3f} ms  1x1 {t1:7.3f} ms ({gb / t1:5.2f} TB/s)  same bits: {bool(torch.equal(o0, o1))}', flush=True)
print(f'sum: direct {tot[0]:.3f} ms, 1x1 {tot[1]:.3f} ms')
