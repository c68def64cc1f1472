#!/usr/bin/env python3
"""What this GPU sustains for pure writes, pure reads and a copy: the ceiling of the write-dominated operators (upfirdn2d up x2 writes
four bytes for every byte it reads).   python tools/hbm_write_bw.py"""
import torch


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


n = 1 << 28   # 1 GiB of float32: four times the last-level cache
x = torch.empty(n, device='cuda')
y = torch.empty(n, device='cuda')
t = timeit(lambda: x.fill_(1.0))
print(f'fill   (write 1 GiB):          {t:.3f} ms  {4 * n / t / 1e9:7.2f} TB/s')
t = timeit(lambda: y.copy_(x))
print(f'copy   (read 1 + write 1 GiB): {t:.3f} ms  {8 * n / t / 1e9:7.2f} TB/s')
t = timeit(lambda: x.sum())
print(f'sum    (read 1 GiB):           {t:.3f} ms  {4 * n / t / 1e9:7.2f} TB/s')
z = torch.empty(n // 4, device='cuda')
t = timeit(lambda: torch.add(z, 1.0, out=y[:n // 4]))
print(f'1:1 elementwise on 256 MiB:    {t:.3f} ms  {2 * n / t / 1e9:7.2f} TB/s')
