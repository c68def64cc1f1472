#!/bin/bash
cat > /tmp/pn.py <<'PY'
import torch, sys, os
sys.path.insert(0, '.')
from mrefsr_amd import hip
x = torch.randn(40, 160, 160, 256, device='cuda')
for _ in range(3):
    hip.pixnorm(x, want_bf16_split=True, nhwc=True, split='fp16', want_err=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    hip.pixnorm(x, want_bf16_split=True, nhwc=True, split='fp16', want_err=True)
e1.record(); torch.cuda.synchronize()
print(os.environ.get('MREFSR_HIP_LIB', 'default'), 'pixnorm 40x160x160x256 channels-last:', e0.elapsed_time(e1) / 20, 'ms')
PY
python /tmp/pn.py; MREFSR_HIP_LIB=mrefsr_amd/lib_old/libmrefsr_hip.so python /tmp/pn.py; python /tmp/pn.py; MREFSR_HIP_LIB=mrefsr_amd/lib_old/libmrefsr_hip.so python /tmp/pn.py
