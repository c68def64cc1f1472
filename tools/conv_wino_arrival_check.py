#!/usr/bin/env python3
"""conv_wino_kernel's hand-counted waits under the checking build (the A/B library, -DMREFSR_AB_KERNELS: after every counted wait
the guarded registers are snapshotted, everything is drained and the registers are compared -- a register that changed was not
covered by the wait).  Runs every instantiation at benchmark-size launches (slow loads, many tiles per block, tile transitions, first
sub-steps) and prints the late-arrival counters; exits 1 if any is non-zero or a result differs from the direct kernel.
    MREFSR_HIP_LIB=mrefsr_amd/lib_ab/libmrefsr_hip.so python tools/conv_wino_arrival_check.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrefsr_amd import hip, _lib  # noqa: E402

lib = _lib.load()
if not hasattr(lib, 'mrefsr_dbg_wino_late'):
    sys.exit('the loaded library is not a checking build (MREFSR_HIP_LIB=mrefsr_amd/lib_ab/libmrefsr_hip.so)')
fn = lib.mrefsr_dbg_wino_late
fn.restype = C.c_int
late = (C.c_uint * 2)()
fn(late)
torch.manual_seed(2)
bad = 0
for n, h, w, ci, co, kind, ep in ((10, 640, 640, 64, 64, 'act', 0), (10, 640, 640, 64, 64, 'res', 0), (10, 640, 640, 64, 64, 'pre', 0), (4, 320, 320, 48, 64, 'act', 1),
                                 (10, 320, 320, 128, 128, 'act', 0), (8, 160, 160, 512, 512, 'act', 0), (2, 50, 70, 40, 24, 'act', 2), (1, 16, 16, 64, 64, 'plain', 0)):
    x = torch.randn(n, h, w, ci, device='cuda')
    wt = torch.randn(co, ci, 3, 3, device='cuda') / (3.0 * ci ** 0.5)
    bias = torch.randn(co, device='cuda')
    kw = dict(act=kind in ('act', 'res', 'pre'), slope=0.1, epilogue=ep)
    if kind == 'res':
        kw['residual'] = torch.randn(n, h, w, co, device='cuda')
    if kind == 'pre':
        kw['pre'] = torch.randn(2, h, w, co, device='cuda')
    ref = hip.conv_nhwc(x, hip.conv_pack_weight(wt, 16), bias, co, 3, **kw)
    pk = hip.conv_pack_weight(wt, 17)
    err = 0.0
    for _ in range(3):
        err = max(err, float((hip.conv_nhwc(x, pk, bias, co, 3, **kw) - ref).abs().max()))
    hip.check_conv_range()
    fn(late)
    ok = late[0] == 0 and late[1] == 0 and err < 2e-5
    bad += 0 if ok else 1
    print(f'N={n} {h}x{w} {ci}->{co} {kind} ep={ep}: late patch pieces {late[0]}, late weight fragments {late[1]}, max |wino - direct| {err:.2e} '
          f'{"ok" if ok else "WRONG"}', flush=True)
sys.exit(1 if bad else 0)
