#!/usr/bin/env python3
"""Where the kernel launches of one training step (BASELINE configs[2] per-GPU shape: B=4, K=5, LR 40x40) come from:
device kernels by name, and the ATen glue (fill / copy / add / cat ...) by the Python line that called it.
    python tools/train_launches.py [--top 40]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--top', type=int, default=40)
ap.add_argument('--stacks', type=int, default=0)
ap.add_argument('--by-time', action='store_true')
o = ap.parse_args()
args = argparse.Namespace(mode='train', batch=4, refs=5, lr=40)
model = bench.build(args, False)
bench.seeded_weights(model)
model.feed_data(bench.synth_batch(4, 5, 40, seed=100))
for i in range(4):
    model.optimize_parameters(i + 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=bool(o.stacks)) as prof:
    model.optimize_parameters(5)
    torch.cuda.synchronize()
evs = prof.events()
kern = [e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA and 'emcpy' not in e.name and 'emset' not in e.name]
print(f'{len(kern)} kernel launches, {sum(e.device_time_total for e in kern) / 1e3:.2f} ms of kernel time')
by = collections.defaultdict(lambda: [0, 0.0])
for e in kern:
    n = e.name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:110]
    by[n][0] += 1
    by[n][1] += e.device_time_total
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1 if o.by_time else 0])[:o.top]:
    print(f'{c:5d} {t / 1e3:8.3f} ms  {n}')
if o.stacks:
    # ATen ops that launch glue kernels, by the first repo frame of their Python stack
    sites = collections.defaultdict(lambda: [0, 0.0])
    glue = ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::cat', 'aten::sum', 'aten::mul', 'aten::clone',
            'aten::contiguous', 'aten::zeros', 'aten::bmm', 'aten::abs', 'aten::sub', 'aten::mean', 'aten::div', 'aten::index', 'aten::stack')
    for e in evs:
        if e.device_type != torch.autograd.DeviceType.CPU or e.name not in glue:
            continue
        dev = sum(k.device_time_total if hasattr(k, 'device_time_total') else 0 for k in e.kernels) if e.kernels else 0
        if not e.kernels:
            continue
        frame = next((s for s in (e.stack or []) if '/root/repo' in s or 'mrefsr_amd' in s or 'bench.py' in s), (e.stack or ['?'])[0] if e.stack else '<autograd engine>')
        sites[(e.name, frame.strip()[-110:])][0] += len(e.kernels)
        sites[(e.name, frame.strip()[-110:])][1] += sum(k.duration for k in e.kernels)
    print('\nATen glue launches by call site')
    for (n, f), (c, t) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:o.top]:
        print(f'{c:5d} {t / 1e3:8.3f} ms  {n:16s} {f}')
