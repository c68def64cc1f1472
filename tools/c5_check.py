#!/usr/bin/env python3
"""BASELINE configs[4] geometry (K=10, LR 320x320) through the fp32 path: pre-filter == exact kernel, end-to-end runs"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrefsr_amd import hip
h = w = 320
torch.manual_seed(0)
fin = torch.randn(1, 256, h, w, device='cuda'); fref = torch.randn(2, 256, h, w, device='cuda')
fref[1] = torch.roll(fin[0], (37, -51), (1, 2)) + 0.3 * torch.randn_like(fin[0])
yi, n2i, bi = hip.pixnorm(fin, want_bf16_split=True); yr, n2r, br = hip.pixnorm(fref, want_bf16_split=True)
nei, _ = hip.patch_norm(n2i); _, invr = hip.patch_norm(n2r)
t0 = time.time(); i1, v1 = hip.corr_top1(yi, yr, invr, nei, h, w, ybf_in=bi, ybf_ref=br); torch.cuda.synchronize(); t1 = time.time()
i2, v2 = hip.corr_top1(yi, yr, invr, nei, h, w); torch.cuda.synchronize(); t2 = time.time()
print(f'320x320: prefilter {1e3*(t1-t0):.0f} ms, exact {1e3*(t2-t1):.0f} ms, equal idx {bool((i1==i2).all())} val {bool((v1==v2).all())}; '
      f'planted recovered {(i1[1].view(-1) == i2[1].view(-1)).float().mean().item():.3f}')
import bench
class A: pass
a = A(); a.batch, a.refs, a.lr, a.mode, a.miopen_find = 1, 10, 320, 'infer', False
model = bench.build(a, False); bench.seeded_weights(model)
model.feed_data(bench.synth_batch(1, 10, 320, seed=3)); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.time(); model.test(); model.check_numeric_range(); torch.cuda.synchronize(); t1 = time.time()
print(f'C5 geometry (B=1, K=10, LR 320 -> 1280) fp32: {1e3*(t1-t0):.0f} ms/step = {1280*1280/1e6/(t1-t0):.2f} Mpix/s, output finite {bool(torch.isfinite(model.output).all())}, '
      f'peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
