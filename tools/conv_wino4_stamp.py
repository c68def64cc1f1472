#!/usr/bin/env python3
"""Where a wave of conv_wino4_kernel spends its life (instrumentation build: make ... EXTRA=-DWINO_STAMP into lib_wstamp): shader-clock
totals per phase as shares of the summed wave lifetimes, and clocks per chunk step / per tile.
    MREFSR_HIP_LIB=mrefsr_amd/lib_wstamp/libmrefsr_hip.so python tools/conv_wino4_stamp.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrefsr_amd import hip, _lib  # noqa: E402

NAMES = ['A: wait j0', 'A: slots 0-5', 'A: slots 6-11 (+stores)', 'A: slots 12-17 (+requests)', 'A: slots 18-23', 'B: MFMA 0 + barrier',
         'B: slots 0-11', 'B: slots 12-23', 'cursors', 'exchange+epilogue', 'bookkeeping']
lib = _lib.load()
fn = lib.mrefsr_dbg_wino4_stamps
fn.restype = C.c_int
buf = (C.c_ulonglong * 12)()
os.environ['MREFSR_WINO_WAVES'] = '4'


def run(n, h, w, cin, cout, residual=False, iters=3):
    x = torch.randn(n, h, w, cin, device='cuda')
    wgt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    pw = hip.conv_pack_weight(wgt, terms=17)
    res = torch.randn(n, h, w, cout, device='cuda') if residual else None
    out = torch.empty(n, h, w, cout, device='cuda')
    hip.conv_nhwc(x, pw, b, cout, 3, residual=res, act=True, slope=0.1, out=out)
    fn(buf)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        hip.conv_nhwc(x, pw, b, cout, 3, residual=res, act=True, slope=0.1, out=out)
    e.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(e) / iters
    fn(buf)
    v = [float(buf[i]) for i in range(12)]
    tot = sum(v[:11])
    chunks = (cin + 15) // 16
    tiles = ((h + 15) // 16) * ((w + 15) // 16) * ((cout + 63) // 64) * n
    steps = tiles * chunks * 4.0 * iters   # wave-steps
    print(f'N={n} {h}x{w} {cin}->{cout} res={int(residual)}: {ms:.3f} ms, {2.0*n*h*w*cin*cout*9/ms/1e9:.0f} TFLOP/s-eq, {tot/v[11]:.0f} clk per wave, '
          f'{sum(v[:9])/steps:.0f} clk per chunk step')
    print('   shares: ' + '  '.join(f'{nm} {100*x_/tot:.1f}%' for nm, x_ in zip(NAMES, v[:11])))
    print('   clk per wave and chunk step: ' + '  '.join(f'{nm} {x_/steps:.0f}' for nm, x_ in zip(NAMES[:9], v[:9])) +
          ' | per tile: ' + '  '.join(f'{nm} {x_/(tiles*4.0*iters):.0f}' for nm, x_ in zip(NAMES[9:], v[9:11])))


run(8, 640, 640, 64, 64, residual=True)
run(40, 640, 640, 64, 64)
run(8, 320, 320, 256, 256)
run(8, 160, 160, 512, 512)
