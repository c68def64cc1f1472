#!/usr/bin/env python3
"""a few launches of ONE convolution shape through the Winograd kernel (terms 17) and the direct one (terms 16), for rocprofv3
--pmc / --kernel-trace passes:    python tools/conv_wino_one.py N H W CIN COUT [iters] [res]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrefsr_amd import hip  # noqa: E402

n, h, w, cin, cout = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
res = len(sys.argv) > 7 and sys.argv[7] == 'res'
torch.manual_seed(0)
x = torch.randn(n, h, w, cin, device='cuda')
wt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.03
bias = torch.randn(cout, device='cuda')
r = torch.randn(n, h, w, cout, device='cuda') if res else None
for terms in (17, 16):
    pk = hip.conv_pack_weight(wt, terms)
    for _ in range(iters):
        hip.conv_nhwc(x, pk, bias, cout, 3, residual=r, act=not res, slope=0.1)
torch.cuda.synchronize()
