#!/usr/bin/env python3
"""ATen kernels left in one inference step of the benchmark workload, with the python line that called them.
    python tools/infer_aten.py"""
import argparse
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
import bench  # noqa: E402

args = argparse.Namespace(mode='infer', batch=8, refs=5, lr=160)
model = bench.build(args, False)
bench.seeded_weights(model)
model.feed_data(bench.synth_batch(8, 5, 160, seed=100))
for _ in range(2):
    model.test()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.test()
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith('aten::') or e.device_time_total <= 0:
        continue
    if e.cpu_children and any(c.name.startswith('aten::') and c.device_time_total > 0 for c in e.cpu_children):
        continue
    st = [s for s in (e.stack or []) if 'mrefsr_amd' in s or 'bench.py' in s]
    rows[(e.name, st[0].split('/root/repo/')[-1][:90] if st else '?', str(e.input_shapes)[:60] if e.input_shapes else '')][0] += 1
    rows[(e.name, st[0].split('/root/repo/')[-1][:90] if st else '?', str(e.input_shapes)[:60] if e.input_shapes else '')][1] += e.device_time_total
for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f'{t / 1e3:8.3f} ms {c:4d}  {k[0]:28s} {k[1]}')
