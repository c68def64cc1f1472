#!/usr/bin/env python3
"""The product's correlation call (fp16 pre-filter with the data-dependent window + exact re-scoring) on synthetic maps of
the benchmark's size (B=8, K=5, 256 x 160 x 160 -> 40 pairs): HIP-event time per call and candidate statistics.
    [MREFSR_HIP_LIB=...] python tools/corr_time.py [calls] [lr]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrefsr_amd import hip  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h = w = int(sys.argv[2]) if len(sys.argv) > 2 else 160
b, k = 8, 5
torch.manual_seed(0)
fin = torch.randn(b, 256, h, w, device='cuda')
fin = fin + 2.0 * torch.nn.functional.avg_pool2d(fin, 5, 1, 2)      # spatially correlated, like real feature maps
fref = torch.cat([torch.roll(fin, (17 * (kk + 1), -23 * (kk + 1)), (2, 3)) + 0.3 * torch.randn_like(fin) for kk in range(k)])
yi, n2i, hi, d2i = hip.pixnorm(fin, want_bf16_split=True, split='fp16', want_err=True)
yr, n2r, hr, d2r = hip.pixnorm(fref, want_bf16_split=True, split='fp16', want_err=True)
nei, _ = hip.patch_norm(n2i)
_, invr = hip.patch_norm(n2r)
tau = hip.prefilter_window(nei, invr, d2i, d2r)
hip._timing['keep_ws'] = True
hip.corr_top1(yi, yr, invr, nei, h, w, want_val=False, ybf_in=hi, ybf_ref=hr, tau=tau)
torch.cuda.synchronize()
hip.set_kernel_timing(True)
for _ in range(calls):
    idx, _ = hip.corr_top1(yi, yr, invr, nei, h, w, want_val=False, ybf_in=hi, ybf_ref=hr, tau=tau)
torch.cuda.synchronize()
ms = hip.kernel_timings()['corr_top1']
ws, n_pair, P = hip._timing['last_corr_ws']
wi = ws.view(torch.int32)
cand_n = wi[n_pair * P * 16: n_pair * P * 17]
print(f'{os.environ.get("MREFSR_HIP_LIB", "default lib")}: corr call ms {[round(x, 2) for x in ms]}  checksum {int(idx.sum())}  '
      f'flagged {int(wi[n_pair * P * 18])}  mean candidates {cand_n[cand_n >= 0].float().mean().item():.3f}')
