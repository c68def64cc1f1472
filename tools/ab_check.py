#!/usr/bin/env python3
"""Correctness of the A/B kernel generations that are compiled only into a -DMREFSR_AB_KERNELS build of the library
(mrefsr_amd/lib_ab, built by __graft_entry__.build()): every pre-filter variant must return the oracle's bits.
    MREFSR_HIP_LIB=mrefsr_amd/lib_ab/libmrefsr_hip.so MREFSR_CORR_PREFILTER_WS16=1 python tools/ab_check.py fp16w
    MREFSR_HIP_LIB=... MREFSR_CORR_PREFILTER_WS=1 | MREFSR_CORR_PREFILTER_STREAM=1 python tools/ab_check.py bf16"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import cases  # noqa: E402
from mrefsr_amd import hip  # noqa: E402
from oracle import c_api as orc  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
bad = 0
for name, fin, fref in cases.corr_cases():
    if fin.shape[0] != 256 or fin.shape[1] < 12:
        continue
    a, b = torch.from_numpy(fin[None]).cuda(), torch.from_numpy(fref[None]).cuda()
    split = 'fp16' if mode == 'fp16w' else 'bf16'
    tau = None
    if mode == 'fp16w':
        yi, n2i, bi, d2i = hip.pixnorm(a, want_bf16_split=True, split=split, want_err=True)
        yr, n2r, br, d2r = hip.pixnorm(b, want_bf16_split=True, split=split, want_err=True)
    else:
        yi, n2i, bi = hip.pixnorm(a, want_bf16_split=True, split=split)
        yr, n2r, br = hip.pixnorm(b, want_bf16_split=True, split=split)
    nei, _ = hip.patch_norm(n2i)
    _, invr = hip.patch_norm(n2r)
    if mode == 'fp16w':
        tau = hip.prefilter_window(nei, invr, d2i, d2r)
    idx, val = hip.corr_top1(yi, yr, invr, nei, fin.shape[1], fin.shape[2], ybf_in=bi, ybf_ref=br, tau=tau)
    oidx, oval = orc.feature_match_index(fin, fref)
    ok = np.array_equal(idx[0].cpu().numpy(), oidx) and np.array_equal(val[0].cpu().numpy(), oval)
    print(name, 'ok' if ok else 'MISMATCH')
    bad += not ok
sys.exit(1 if bad else 0)
