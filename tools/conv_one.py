#!/usr/bin/env python3
"""a few launches of ONE convolution shape through both kernels (for rocprofv3 --pmc / --kernel-trace passes):
    python tools/conv_one.py N H W CIN COUT [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrefsr_amd import hip  # noqa: E402

n, h, w, cin, cout = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
torch.manual_seed(0)
x = torch.randn(n, h, w, cin, device='cuda')
pk = hip.conv_pack_weight(torch.randn(cout, cin, 3, 3, device='cuda') * 0.03, 16)
bias = torch.randn(cout, device='cuda')
for flag in ('0', '1'):
    os.environ['MREFSR_CONV8'] = flag
    for _ in range(iters):
        hip.conv_nhwc(x, pk, bias, cout, 3, act=True, slope=0.1)
torch.cuda.synchronize()
