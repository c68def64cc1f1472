#!/usr/bin/env python3
"""conv_nhwc8_kernel (8 waves, two cout blocks per halo tile) against conv_nhwc_kernel on the benchmark's Cout >= 128 shapes:
bit-identical outputs (same accumulation order) and the time of each.   python tools/conv8_check.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrefsr_amd import hip  # noqa: E402


def timeit(fn, warm=2, iters=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


torch.manual_seed(0)
shapes = [  # n, h, w, cin, cin2, cout, residual, act, epilogue
    (8, 640, 640, 128, 0, 128, False, True, 0), (40, 320, 320, 128, 0, 128, False, True, 0), (8, 320, 320, 256, 0, 256, False, True, 0),
    (40, 160, 160, 256, 0, 256, True, False, 0), (8, 160, 160, 512, 0, 512, False, True, 0), (40, 640, 640, 64, 0, 128, False, True, 0),
    (40, 320, 320, 128, 0, 256, False, True, 0), (40, 320, 320, 128, 0, 128, False, True, 1), (8, 320, 320, 64, 64, 128, False, True, 0),
    (8, 160, 160, 64, 0, 256, False, True, 2), (3, 100, 92, 144, 0, 192, True, False, 0)]
tot = [0.0, 0.0]
for n, h, w, c1, c2, co, res, act, ep in shapes:
    x1 = torch.randn(n, h, w, c1, device='cuda')
    x2 = torch.randn(n, h, w, c2, device='cuda') if c2 else None
    wt = torch.randn(co, c1 + c2, 3, 3, device='cuda') * 0.03
    bias = torch.randn(co, device='cuda')
    r = torch.randn(n, h, w, co, device='cuda') if res else None
    pk = hip.conv_pack_weight(wt, 16)
    outs, ts = [], []
    for flag in ('0', '1'):
        os.environ['MREFSR_CONV8'] = flag
        f = lambda: hip.conv_nhwc(x1, pk, bias, co, 3, x2=x2, residual=r, act=act, slope=0.1, epilogue=ep)  # noqa: E731
        outs.append(f())
        ts.append(timeit(f, warm=2, iters=5))
    same = torch.equal(outs[0], outs[1])
    fl = 2.0 * n * h * w * (c1 + c2) * co * 9
    tot[0] += ts[0]
    tot[1] += ts[1]
    print(f'N={n:2d} {h}x{w} {c1}+{c2}->{co} res={int(res)} ep={ep}: 4-wave {ts[0]:7.3f} ms {fl/ts[0]/1e9:6.1f} TF/s | 8-wave {ts[1]:7.3f} ms {fl/ts[1]/1e9:6.1f} TF/s '
          f'({ts[0]/ts[1]:.3f}x) | bit-identical {same}', flush=True)
    if not same:
        d = (outs[0] - outs[1]).abs()
        print('   max diff', float(d.max()), 'mismatching', int((d > 0).sum()), 'of', d.numel())
    hip.check_conv_range()
print(f'sum: 4-wave {tot[0]:.2f} ms, 8-wave {tot[1]:.2f} ms ({tot[0]/tot[1]:.3f}x)')
# NaN / out-of-range input raises the range flag in both kernels
x = torch.randn(8, 160, 160, 128, device='cuda')
x[3, 17, 33, 5] = 7.0e4
pk = hip.conv_pack_weight(torch.randn(128, 128, 3, 3, device='cuda') * 0.03, 16)
for flag in ('0', '1'):
    os.environ['MREFSR_CONV8'] = flag
    hip.conv_nhwc(x, pk, None, 128, 3)
    assert hip.conv_range_tripped(), flag
print('range flag ok')
