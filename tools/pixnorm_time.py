#!/usr/bin/env python3
"""time of mrefsr_pixnorm_f32 on the benchmark's reference maps (40 x 160 x 160 x 256 channels-last, fp16 operand + error norms)"""
import sys
import torch
sys.path.insert(0, '.')
from mrefsr_amd import hip  # noqa: E402
x = torch.randn(40, 160, 160, 256, device='cuda')
for _ in range(2):
    hip.pixnorm(x, normalize=True, want_bf16_split=True, nhwc=True, split='fp16', want_err=True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    hip.pixnorm(x, normalize=True, want_bf16_split=True, nhwc=True, split='fp16', want_err=True)
b.record()
torch.cuda.synchronize()
print('pixnorm nhwc 40x160x160x256 fp16+err: %.3f ms' % (a.elapsed_time(b) / 5))
