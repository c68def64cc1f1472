#!/usr/bin/env python3
"""Per-shape table of the convolution launches of one benchmark step (N, H, W, Cin, Cout, k, launches, ms, TFLOP/s
fp32-equivalent): where the convolution time goes.    python tools/conv_layers.py"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from mrefsr_amd import hip  # noqa: E402


class A:
    batch = 8; refs = 5; lr = 160; mode = 'infer'; dtype = 'fp32'; graph = False; miopen_find = False  # noqa: E702


model = bench.build(A, False)
bench.seeded_weights(model)
model.feed_data(bench.synth_batch(8, 5, 160, seed=10))
for _ in range(2):
    model.test()
torch.cuda.synchronize()
rec = []
det = []
orig = hip.conv_nhwc


def wrapped(x1, packed, bias, cout, ksize, x2=None, **kw):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = orig(x1, packed, bias, cout, ksize, x2=x2, **kw)
    b.record()
    n, h, w = out.shape[0], x1.shape[1], x1.shape[2]
    cin = x1.shape[3] + (x2.shape[3] if x2 is not None else 0)
    rec.append(((n, h, w, cin, cout, ksize, kw.get('epilogue', 0)), a, b))
    if '--detail' in sys.argv:
        det.append((rec[-1][0], tuple(x1.stride()), None if x2 is None else tuple(x2.shape), tuple(out.stride()),
                    [k for k in ('pre', 'residual', 'act', 'slope_ptr', 'out') if kw.get(k) is not None and kw.get(k) is not False], a, b))
    return out


orig_dyn = hip.conv_dynagg


def wrapped_dyn(x, packed, bias, pre, dg, abs_sum=None):
    """conv_offset_mask + DynAgg glue (mrefsr_conv_dynagg_f32, epilogue 3): the same kernel, three launches per step"""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = orig_dyn(x, packed, bias, pre, dg, abs_sum)
    b.record()
    n, h, w, c = x.shape
    rec.append(((n, h, w, c, 27 * dg, 3, 3), a, b))
    return out


hip.conv_nhwc = wrapped
hip.conv_dynagg = wrapped_dyn
model.test()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, a, b in rec:
    ms = a.elapsed_time(b)
    c, t = agg.get(key, (0, 0.0))
    agg[key] = (c + 1, t + ms)
tot = sum(t for _, t in agg.values())
print(f'{len(rec)} launches, {tot:.1f} ms')
print('    N    H    W  Cin Cout k ep  launches       ms    %   TFLOP/s')
for (n, h, w, cin, cout, k, ep), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * n * h * w * cin * cout * k * k * c
    print(f'{n:5d} {h:4d} {w:4d} {cin:4d} {cout:4d} {k} {ep:2d}  {c:8d} {t:8.2f} {100 * t / tot:4.1f}  {fl / t / 1e9:8.1f}')
if '--detail' in sys.argv:   # one line per launch, in launch order
    for key, xs, x2s, os_, flags, a, b in det:
        print(key, 'x1 stride', xs, 'x2', x2s, 'out stride', os_, flags, f'{a.elapsed_time(b):.3f} ms')
