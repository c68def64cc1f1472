#!/usr/bin/env python3
"""Time the correlation call inside the benchmark step (its real feature maps): three steps, per-call HIP-event time
and an output checksum (identical across kernel variants: the re-scoring makes every variant exact).
    [MREFSR_HIP_LIB=...] python tools/corr_bench_step.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


class A:
    batch = 8; refs = 5; lr = 160; mode = 'infer'; dtype = 'fp32'; graph = False; miopen_find = False  # noqa: E702


model = bench.build(A, False)
bench.seeded_weights(model)
model.feed_data(bench.synth_batch(8, 5, 160, seed=10))
from mrefsr_amd import hip  # noqa: E402
hip._timing['keep_ws'] = True
for _ in range(2):
    model.test()
torch.cuda.synchronize()
hip.set_kernel_timing(True)
for _ in range(3):
    model.test()
torch.cuda.synchronize()
print('corr ms', [round(x, 1) for x in hip.kernel_timings().get('corr_top1', [])], 'checksum', float(model.output.double().sum()))
ws, n_pair, P = hip._timing['last_corr_ws']
wi = ws.view(torch.int32)
cand_n = wi[n_pair * P * 16: n_pair * P * 17]
print(f'flagged queries {int(wi[n_pair * P * 18])} of {n_pair * P}; mean candidates {cand_n[cand_n >= 0].float().mean().item():.3f}; '
      f'hist(-1..16) {[int((cand_n == i).sum()) for i in range(-1, 17)]}')
