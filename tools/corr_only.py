#!/usr/bin/env python3
"""Runs only the correlation kernel at the bench shape (B=8, K=5, C=256, 160x160 -> 40 pairs per
launch) for rocprofv3 counter collection:  python3 tools/corr_only.py [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrefsr_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
b, k, h, w = 8, 5, 160, 160
g = torch.Generator().manual_seed(0)
fin = torch.randn(b, 256, h, w, generator=g).cuda()
fref = torch.randn(k * b, 256, h, w, generator=g).cuda()
exact = os.environ.get('MREFSR_CORR_EXACT', '0') == '1'
yi, n2i, bi = hip.pixnorm(fin, want_bf16_split=True)
yr, n2r, br = hip.pixnorm(fref, want_bf16_split=True)
nei, _ = hip.patch_norm(n2i)
_, invr = hip.patch_norm(n2r)
for _ in range(n):
    if exact:
        idx, _ = hip.corr_top1(yi, yr, invr, nei, h, w, want_val=False)
    else:
        idx, _ = hip.corr_top1(yi, yr, invr, nei, h, w, want_val=False, ybf_in=bi, ybf_ref=br)
torch.cuda.synchronize()
print('done', int(idx.sum()))
