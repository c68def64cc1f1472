#!/usr/bin/env python3
"""Run-to-run reproducibility of mrefsr_dcn_fwd_f32 at the three benchmark scales (B=8): the same
inputs REPS times, every output compared bit for bit with the first, and the first compared with the
fp32-MFMA kernel.  MREFSR_DCN_BF16=1 selects the bf16-split kernel; MREFSR_HIP_LIB another build.
    python tools/dcn_repro.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, '.')
from mrefsr_amd import hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
only = os.environ.get('DCN_REPRO_CHUNK')   # "tap,cb": zero every weight outside that 32-channel chunk of that tap
shapes = ((256, 160), (128, 320), (64, 640))
if only or os.environ.get('DCN_REPRO_C128'):
    shapes = ((128, 320),)
torch.manual_seed(0)
bad = 0
for c, hw in shapes:
    x = torch.randn(8, hw, hw, c, device='cuda')
    off = torch.randn(8, 144, hw, hw, device='cuda') * 4
    msk = torch.rand(8, 72, hw, hw, device='cuda')
    wgt = torch.randn(c, c, 3, 3, device='cuda') * 0.02
    bias = torch.randn(c, device='cuda') * 0.1
    if only:
        tap, cb = (int(v) for v in only.split(','))
        keep = torch.zeros_like(wgt)
        keep[:, 32 * cb:32 * cb + 32, tap // 3, tap % 3] = 1
        wgt = wgt * keep
    first = None
    for r in range(reps):
        out = hip.dcn_fwd(x, off, msk, wgt, bias, 1, 1, 1, 1, 8, 0.1, channels_last=True)
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
            continue
        d = (out != first)
        nbad = int(d.sum())
        if nbad:
            bad += 1
            px = d.any(dim=3)
            where = px.nonzero()[:6].tolist()
            print(f'C={c} {hw}x{hw} rep {r}: {nbad} values differ on {int(px.sum())} pixels, max {float((out - first).abs().max()):.3e}; first {where}')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        hip.dcn_fwd(x, off, msk, wgt, bias, 1, 1, 1, 1, 8, 0.1, channels_last=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f'C={c} {hw}x{hw}: {reps} repetitions done, {ms:.2f} ms per call, {2.0 * 8 * hw * hw * c * c * 9 / ms / 1e9:.0f} TFLOP/s '
          f'(MREFSR_DCN_BF16={os.environ.get("MREFSR_DCN_BF16", "unset")})', flush=True)
print('NOT REPRODUCIBLE' if bad else 'reproducible')
