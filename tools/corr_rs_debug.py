#!/usr/bin/env python3
"""Debug aid for the row-stationary pre-filter (csrc/corr_rowstream.hip): dumps the approximate score of every
(query, reference) patch pair it forms and compares it with the fp32 value computed from the same fp16 operands.
Needs the debug build:   make -C mrefsr_amd/csrc OBJDIR=_obj_dbg OUTDIR=../lib_dbg EXTRA=-DMREFSR_CORR_DEBUG
    MREFSR_HIP_LIB=mrefsr_amd/lib_dbg/libmrefsr_hip.so python tools/corr_rs_debug.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrefsr_amd import _lib, hip  # noqa: E402

lib = _lib.load()
fn = lib.mrefsr_dbg_corr_rs16_scores
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p]
bad = 0
for (h, w) in ((12, 14), (9, 21), (20, 33), (40, 40), (7, 50)):
    torch.manual_seed(h * 100 + w)
    fin = torch.randn(1, 256, h, w, device='cuda')
    fref = torch.roll(fin, (3, -2), (2, 3)) + 0.3 * torch.randn(1, 256, h, w, device='cuda')
    yi, n2i, hi = hip.pixnorm(fin, want_bf16_split=True, split='fp16')
    yr, n2r, hr = hip.pixnorm(fref, want_bf16_split=True, split='fp16')
    nei, _ = hip.patch_norm(n2i)
    _, invr = hip.patch_norm(n2r)
    ph, pw = h - 2, w - 2
    P = ph * pw
    scores = torch.full((P, P), float('nan'), device='cuda')
    ws = torch.zeros(lib.mrefsr_corr_workspace_bytes(1, h, w), dtype=torch.uint8, device='cuda')
    rc = fn(hi.data_ptr(), hr.data_ptr(), invr.data_ptr(), nei.data_ptr(), None, scores.data_ptr(), ws.data_ptr(), h, w,
            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0, lib.mrefsr_last_error()
    # reference: the fp16 operands (split layout is a channel permutation: irrelevant for dot products), fp32 math
    a = hi.float().view(h, w, 256)
    b = hr.float().view(h, w, 256)
    g = torch.einsum('yxc,vuc->yxvu', a, b)                       # pixel Gram [h,w,h,w]
    s = sum(g[dy:dy + ph, dx:dx + pw, dy:dy + ph, dx:dx + pw] for dy in range(3) for dx in range(3))
    s = (s * invr.view(1, 1, ph, pw)).reshape(P, P)
    missing = int(torch.isnan(scores).sum())
    err = float((scores - s).abs().nan_to_num(0).max())
    wi = ws.view(torch.int32)
    cand_n = wi[P * 16: P * 17]
    print(f'{h}x{w}: P={P} never scored={missing} max|approx - fp32|={err:.3e}  cand_n min/max {int(cand_n.min())}/{int(cand_n.max())} '
          f'flagged {int(wi[P * 18])}')
    if missing or err > 2e-3:
        bad += 1
        q, r = (torch.isnan(scores) | ((scores - s).abs() > 2e-3)).nonzero()[0].tolist()
        print(f'   first bad entry: query ({q // pw},{q % pw}) ref ({r // pw},{r % pw}) got {float(scores[q, r])} want {float(s[q, r])}')
        d = (torch.isnan(scores) | ((scores - s).abs() > 2e-3)).view(ph, pw, ph, pw)
        print('   bad by query row', d.any(3).any(2).any(1).int().tolist())
        print('   bad by query col', d.any(3).any(2).any(0).int().tolist())
        print('   bad by ref row  ', d.any(3).any(1).any(0).int().tolist())
        print('   bad by ref col  ', d.any(2).any(1).any(0).int().tolist())
print('FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
