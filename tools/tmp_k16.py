import os, sys, json, subprocess
sys.path.insert(0, '.')
import torch
import bench
class A: batch=8; refs=5; lr=160; mode='infer'; dtype='fp32'; graph=False; miopen_find=False
model = bench.build(A, False); bench.seeded_weights(model)
model.feed_data(bench.synth_batch(8, 5, 160, seed=10))
from mrefsr_amd import hip
for _ in range(2): model.test()
torch.cuda.synchronize()
hip.set_kernel_timing(True)
for _ in range(3): model.test()
torch.cuda.synchronize()
t = hip.kernel_timings().get('corr_top1', [])
print('corr ms', [round(x, 1) for x in t], 'checksum', float(model.output.double().sum()))
