#!/usr/bin/env python3
"""conv_wino4_kernel (four 512-register waves, csrc/conv_wino4.hip) against conv_wino_kernel (eight waves, MREFSR_WINO_WAVES=8) and the
direct kernel (terms 16): (1) small shapes with every epilogue -- same bits as the eight-wave kernel, error against an fp64 convolution;
(2) the benchmark's layer shapes: time of the three kernels, bits.   python tools/conv_wino4_check.py [--quick]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from mrefsr_amd import hip  # noqa: E402


def timeit(fn, warm=2, iters=7):
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


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def waves(n):
    os.environ['MREFSR_WINO_WAVES'] = str(n)


torch.manual_seed(0)
small = [  # n, h, w, c1, c2, cout, residual, pre, act, epilogue
    (1, 16, 16, 48, 0, 64, False, False, False, 0), (2, 20, 36, 64, 0, 64, True, False, True, 0), (1, 33, 19, 32, 16, 40, False, True, True, 0),
    (2, 12, 40, 40, 0, 24, False, False, True, 1), (2, 12, 40, 36, 0, 24, False, False, True, 2), (1, 17, 31, 64, 64, 128, True, False, False, 0),
    (3, 8, 8, 256, 0, 64, False, False, True, 0), (5, 40, 72, 48, 0, 64, False, False, True, 0), (1, 24, 24, 36, 0, 3, False, False, False, 0),
    (2, 48, 48, 64, 0, 64, False, True, True, 0), (9, 64, 64, 64, 0, 128, True, False, True, 0), (4, 96, 80, 128, 0, 64, False, False, True, 1)]
bad = 0
for n, h, w, c1, c2, co, res, pre, act, ep in small:
    x1, x2 = torch.randn(n, c1, h, w), (torch.randn(n, c2, h, w) if c2 else None)
    wt = torch.randn(co, c1 + c2, 3, 3) / (3.0 * (c1 + c2) ** 0.5)
    bias = torch.randn(co)
    r = torch.randn(n, co, h, w) if res else None
    pr = torch.randn(n, co, h, w) if pre else None
    xin = torch.cat([x1, x2], 1) if c2 else x1
    want = F.conv2d(xin.double(), wt.double(), bias.double(), 1, 1)
    if pre:
        want = want + pr.double()
    if act:
        want = F.leaky_relu(want, 0.1)
    if res:
        want = want + r.double()
    if ep == 1:
        want = F.max_pool2d(want, 2, 2)
    elif ep == 2:
        want = F.pixel_shuffle(want, 2)
    pk = hip.conv_pack_weight(wt.cuda(), 17)
    outs = []
    for nw in (8, 4):
        waves(nw)
        out = torch.full_like(nhwc(want.float()), float('nan'))
        got = hip.conv_nhwc(nhwc(x1), pk, bias.cuda(), co, 3, x2=nhwc(x2) if c2 else None, pre=nhwc(pr) if pre else None,
                            residual=nhwc(r) if res else None, act=act, slope=0.1, epilogue=ep)
        outs.append(got)
    hip.check_conv_range()
    err = (outs[1].permute(0, 3, 1, 2).cpu().double() - want).abs().max().item()
    same = torch.equal(outs[0], outs[1])
    ok = err < 2e-5 and same
    bad += 0 if ok else 1
    print(f'N={n} {h}x{w} {c1}+{c2}->{co} res={int(res)} pre={int(pre)} act={int(act)} ep={ep}: four-wave max err {err:.2e}, '
          f'bits {"equal" if same else "DIFFER (max %.3e)" % float((outs[0] - outs[1]).abs().max())} {"ok" if ok else "WRONG"}', flush=True)
if bad:
    sys.exit(1)
if '--quick' in sys.argv:
    sys.exit(0)

shapes = [  # n, h, w, cin, cout, residual, act
    (8, 640, 640, 64, 64, True, False), (8, 640, 640, 64, 64, False, True), (40, 640, 640, 64, 64, False, True), (8, 320, 320, 64, 64, True, False),
    (8, 160, 160, 64, 64, True, False), (8, 640, 640, 128, 128, False, True), (40, 320, 320, 128, 128, False, True),
    (8, 320, 320, 256, 256, False, True), (40, 160, 160, 256, 256, False, True), (8, 160, 160, 512, 512, False, True),
    (40, 640, 640, 64, 128, False, True), (40, 320, 320, 128, 256, False, True), (40, 160, 160, 256, 512, False, True)]
tot = [0.0, 0.0, 0.0]
for n, h, w, ci, co, res, act in shapes:
    x = torch.randn(n, h, w, ci, device='cuda')
    wt = torch.randn(co, ci, 3, 3, device='cuda') / (3.0 * ci ** 0.5)
    bias = torch.randn(co, device='cuda')
    r = torch.randn(n, h, w, co, device='cuda') if res else None
    outs, ts = [], []
    for terms, nw in ((16, 0), (17, 8), (17, 4)):
        waves(nw)
        pk = hip.conv_pack_weight(wt, terms)
        f = lambda: hip.conv_nhwc(x, pk, bias, co, 3, residual=r, act=act, slope=0.1)  # noqa: E731
        outs.append(f())
        ts.append(timeit(f))
    hip.check_conv_range()
    fl = 2.0 * n * h * w * ci * co * 9
    for i in range(3):
        tot[i] += ts[i]
    same = torch.equal(outs[1], outs[2])
    print(f'N={n:2d} {h}x{w} {ci}->{co} res={int(res)}: direct {ts[0]:7.3f} ms {fl/ts[0]/1e9:6.1f} | eight-wave {ts[1]:7.3f} ms {fl/ts[1]/1e9:6.1f} | '
          f'four-wave {ts[2]:7.3f} ms {fl/ts[2]/1e9:6.1f} TF/s ({ts[1]/ts[2]:.2f}x of eight, {ts[0]/ts[2]:.2f}x of direct) | '
          f'bits {"equal" if same else "DIFFER %.2e" % float((outs[1] - outs[2]).abs().max())}', flush=True)
print(f'sum: direct {tot[0]:.2f} ms, eight-wave {tot[1]:.2f} ms, four-wave {tot[2]:.2f} ms')
