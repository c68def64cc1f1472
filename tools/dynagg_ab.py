#!/usr/bin/env python3
"""conv_offset_mask (64 -> 216, 40 x 640 x 640) with its DynAgg epilogue against the same convolution with the plain NHWC epilogue"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrefsr_amd import hip  # noqa: E402


def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for n, hw, c in ((40, 640, 64), (40, 320, 128), (40, 160, 256)):
    torch.manual_seed(0)
    x = torch.randn(n, hw, hw, c, device='cuda')
    w = torch.randn(216, c, 3, 3, device='cuda') * 0.02
    b = torch.randn(216, device='cuda') * 0.1
    pre = torch.randn(n, 9, hw, hw, 2, device='cuda')
    pk = hip.conv_pack_weight(w, 16)
    t_dyn = timeit(lambda: hip.conv_dynagg(x, pk, b, pre, 8))
    out = torch.empty(n, hw, hw, 216, device='cuda')
    t_plain = timeit(lambda: hip.conv_nhwc(x, pk, b, 216, 3, out=out))
    w2 = torch.randn(256, c, 3, 3, device='cuda') * 0.02
    pk2 = hip.conv_pack_weight(w2, 16)
    out2 = torch.empty(n, hw, hw, 256, device='cuda')
    t_256 = timeit(lambda: hip.conv_nhwc(x, pk2, None, 256, 3, out=out2))
    pk17 = hip.conv_pack_weight(w2, 17)
    t_w = timeit(lambda: hip.conv_nhwc(x, pk17, None, 256, 3, out=out2))
    print(f'{n}x{hw}x{hw} {c}->216: DynAgg epilogue {t_dyn:.2f} ms | plain NHWC epilogue {t_plain:.2f} ms | direct {c}->256 {t_256:.2f} ms | Winograd {c}->256 {t_w:.2f} ms', flush=True)
    del x, pre, out, out2
