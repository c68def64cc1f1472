#!/usr/bin/env python3
"""Per-kernel micro-benchmark at BASELINE config-2 sizes (5-ref, LR 160x160, B=8, fp32).
Run on the GPU box:  python tools/kbench.py [corr|dcn|attn|conv|all]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from mrefsr_amd import hip  # noqa: E402


def timeit(fn, warm=2, iters=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def bench_corr(h=160, w=160, b=8, k=5):
    fin = torch.randn(b, 256, h, w, device='cuda')
    fref = torch.randn(k * b, 256, h, w, device='cuda')
    t = timeit(lambda: hip.pixnorm(fref))
    byts = fref.numel() * 4 * 2
    print(f'pixnorm      N={k*b}: {t:8.3f} ms  {byts/t/1e6:8.1f} GB/s')
    for split in ('fp16', 'bf16'):
        yi, n2i, bfi = hip.pixnorm(fin, want_bf16_split=True, split=split)
        yr, n2r, bfr = hip.pixnorm(fref, want_bf16_split=True, split=split)
        nei, _ = hip.patch_norm(n2i)
        _, invr = hip.patch_norm(n2r)
        for name, data in (('random', None), ('planted', 1)):
            if data:
                fr2 = torch.cat([torch.roll(fin, (17 * (kk + 1), -23 * (kk + 1)), (2, 3)) + 0.3 * torch.randn_like(fin) for kk in range(k)])
                yr2, n2r2, bfr2 = hip.pixnorm(fr2, want_bf16_split=True, split=split)
                _, invr2 = hip.patch_norm(n2r2)
            else:
                yr2, bfr2, invr2 = yr, bfr, invr
            t = timeit(lambda: hip.corr_top1(yi, yr2, invr2, nei, h, w, ybf_in=bfi, ybf_ref=bfr2), warm=1, iters=3)
            P = (h - 2) * (w - 2)
            print(f'corr_top1 PREFILTER[{split}] {name:8s} pairs={k*b}: {t:8.2f} ms  {t/(k*b):6.2f} ms/pair  algorithmic {2.0*P*P*2304*k*b/t/1e9:7.1f} TF/s')
            i1, v1 = hip.corr_top1(yi, yr2, invr2, nei, h, w)
            i2, v2 = hip.corr_top1(yi, yr2, invr2, nei, h, w, ybf_in=bfi, ybf_ref=bfr2)
            print('   equal to exact kernel:', bool((i1 == i2).all()), bool((v1 == v2).all()))
    for npair in (8, k * b):
        t = timeit(lambda: hip.corr_top1(yi, yr[:npair], invr[:npair], nei, h, w), warm=1, iters=3)
        P = (h - 2) * (w - 2)
        alg = 2.0 * P * P * 2304 * npair
        tiles = ((h - 2 + 5) // 6) * ((w - 2 + 13) // 14)
        exe = 2.0 * 128 * 128 * 256 * tiles * tiles * npair
        print(f'corr_top1 pairs={npair:3d}: {t:8.2f} ms  {t/npair:6.2f} ms/pair  algorithmic {alg/t/1e9:7.1f} TF/s  '
              f'executed-mfma {exe/t/1e9:6.1f} TF/s ({exe/t/1e9/157.3*100:.0f}% of fp32 matrix peak)')
    idx, _ = hip.corr_top1(yi, yr, invr, nei, h, w)
    t = timeit(lambda: hip.offsets_from_idx(idx, h, w))
    byts = idx.numel() * 8 + k * b * 9 * 2 * 4 * 21 * h * w
    print(f'offsets      N={k*b}: {t:8.3f} ms  {byts/t/1e6:8.1f} GB/s')


def bench_dcn(b=8):
    for c, hw in ((256, 160), (128, 320), (64, 640)):
        x = torch.randn(b, c, hw, hw, device='cuda')
        off = torch.randn(b, 144, hw, hw, device='cuda') * 5
        msk = torch.rand(b, 72, hw, hw, device='cuda')
        wgt = torch.randn(c, c, 3, 3, device='cuda') * 0.02
        bias = torch.zeros(c, device='cuda')
        fl = 2.0 * c * c * 9 * hw * hw * b
        byts = ((2 * c + 216) * hw * hw * 4) * b
        for nhwc in (False, True):
            t = timeit(lambda: hip.dcn_fwd(x, off, msk, wgt, bias, 1, 1, 1, 1, 8, 0.1, nhwc_gather=nhwc))
            print(f'dcn_fwd C={c:3d} {hw}x{hw} B={b} nhwc={int(nhwc)}: {t:8.2f} ms  {fl/t/1e9:6.1f} TF/s  alg-bytes {byts/t/1e6:7.1f} GB/s')
        om = torch.randn(b, 216, hw, hw, device='cuda')
        pre = torch.randn(b, 9, hw, hw, 2, device='cuda')
        t = timeit(lambda: hip.dynagg_prep(om, pre, 8))
        print(f'  dynagg_prep: {t:8.3f} ms  {(om.numel()*2+pre.numel())*4/t/1e6:8.1f} GB/s')
        del x, off, msk, om, pre


def bench_attn(b=8, k=5):
    for c, hw in ((256, 160), (128, 320), (64, 640)):
        q = torch.randn(b, c, hw, hw, device='cuda')
        emb = torch.randn(b * k, c, hw, hw, device='cuda')
        ass = torch.randn(b * k, 2 * c, hw, hw, device='cuda')
        t = timeit(lambda: hip.mrattn_fwd(q, emb, ass, k, want_prob=False))
        byts = (3 * k + 3) * c * hw * hw * 4 * b
        print(f'mrattn_fwd c={c:3d} {hw}x{hw} B={b} T={k}: {t:8.2f} ms  {byts/t/1e6:8.1f} GB/s ({byts/t/1e6/8000*100:.0f}% of 8 TB/s)')
        del q, emb, ass


def bench_conv(b=8):
    import torch.nn.functional as F
    torch.backends.cudnn.benchmark = True
    for cin, cout, hw, n in ((64, 64, 640, b), (64, 64, 160, b), (320, 256, 160, b), (256, 256, 160, b), (512, 512, 160, b),
                             (256, 256, 320, b), (128, 128, 640, b), (128, 128, 320, b), (3, 64, 640, 6 * b),
                             (64, 64, 640, 6 * b), (128, 128, 320, 6 * b), (256, 216, 160, b), (64, 216, 640, b)):
        x = torch.randn(n, cin, hw, hw, device='cuda')
        wt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.02
        t = timeit(lambda: F.conv2d(x, wt, padding=1), warm=3, iters=5)
        fl = 2.0 * cin * cout * 9 * hw * hw * n
        print(f'conv3x3 {cin:3d}->{cout:3d} {hw}x{hw} N={n:2d}: {t:8.2f} ms  {fl/t/1e9:6.1f} TF/s')
        del x


def bench_conv3x3(b=8):
    """own bf16-split implicit-GEMM conv (NHWC) against MIOpen fp32 (NCHW, default find mode as in bench.py)"""
    import torch.nn.functional as F
    for cin, cout, hw, n in ((64, 64, 640, b), (64, 64, 320, b), (64, 64, 160, b), (128, 128, 320, 5 * b), (256, 256, 160, 5 * b),
                             (64, 216, 640, b), (128, 256, 160, 5 * b)):
        x = torch.randn(n, cin, hw, hw, device='cuda')
        wt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.02
        bias = torch.randn(cout, device='cuda')
        fl = 2.0 * cin * cout * 9 * hw * hw * n
        t0 = timeit(lambda: F.conv2d(x, wt, bias, padding=1), warm=2, iters=5)
        xn = x.permute(0, 2, 3, 1).contiguous()
        line = f'conv3x3 {cin:3d}->{cout:3d} {hw}x{hw} N={n:2d}: MIOpen {t0:7.2f} ms {fl/t0/1e9:6.1f} TF/s'
        for terms in (6, 16):
            pk = hip.conv_pack_weight(wt, terms)
            t = timeit(lambda: hip.conv_nhwc(xn, pk, bias, cout, 3, terms=terms), warm=2, iters=5)
            line += f' | x{terms}: {t:7.2f} ms {fl/t/1e9:6.1f} TF/s'
        ref = F.conv2d(x, wt, bias, padding=1)
        got = hip.conv_nhwc(xn, hip.conv_pack_weight(wt, 6), bias, cout, 3).permute(0, 3, 1, 2)
        line += f' | maxdiff {(got - ref).abs().max().item():.2e}'
        got16 = hip.conv_nhwc(xn, hip.conv_pack_weight(wt, 16), bias, cout, 3, terms=16).permute(0, 3, 1, 2)
        line += f' x16 vs x6 {(got16 - got).abs().max().item():.2e}'
        print(line, flush=True)
        del x, xn, ref, got


def bench_stylegan_ops():
    """the two HBM-bound StyleGAN2 operators of basicsr.ops at StyleGAN2-like sizes: algorithmic bytes (fused_act: read + write;
    upfirdn2d: input + output) / time against the 8 TB/s HBM peak"""
    from mrefsr_amd.ops.upfirdn2d import upfirdn2d
    for dt in (torch.float32, torch.float16):
        x = torch.randn(8, 512, 256, 256, device='cuda', dtype=dt)
        b = torch.randn(512, device='cuda', dtype=dt)
        e = x.new_empty(0)
        t = timeit(lambda: hip.fused_bias_act(x, b, e, 3, 0, 0.2, 2 ** 0.5))
        byts = 2 * x.numel() * x.element_size()
        print(f'fused_bias_act fwd {str(dt)[6:]:8s} {tuple(x.shape)}: {t:7.3f} ms  {byts/t/1e6:7.1f} GB/s  ({byts/t/1e6/8000*100:.0f} % of 8 TB/s)')
        t = timeit(lambda: hip.fused_bias_act(x, e, x, 3, 1, 0.2, 2 ** 0.5))
        byts = 3 * x.numel() * x.element_size()
        print(f'fused_bias_act bwd {str(dt)[6:]:8s} {tuple(x.shape)}: {t:7.3f} ms  {byts/t/1e6:7.1f} GB/s  ({byts/t/1e6/8000*100:.0f} % of 8 TB/s)')
        del x
        k = torch.tensor([1., 3., 3., 1.], device='cuda')
        k = (k[:, None] * k[None, :] / 64).to(dt)
        for name, shape, up, down, pad in (('blur', (8, 256, 256, 256), 1, 1, (2, 1)), ('up 2x', (8, 256, 128, 128), 2, 1, (2, 1)),
                                           ('down 2x', (8, 256, 256, 256), 1, 2, (1, 1))):
            x = torch.randn(*shape, device='cuda', dtype=dt)
            kk = k * (up * up)
            out = upfirdn2d(x, kk, up=up, down=down, pad=pad)
            t = timeit(lambda: upfirdn2d(x, kk, up=up, down=down, pad=pad))
            byts = (x.numel() + out.numel()) * x.element_size()
            print(f'upfirdn2d {name:7s} {str(dt)[6:]:8s} {tuple(x.shape)} -> {tuple(out.shape)}: {t:7.3f} ms  {byts/t/1e6:7.1f} GB/s  '
                  f'({byts/t/1e6/8000*100:.0f} % of 8 TB/s)')
            del x, out


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    print(torch.cuda.get_device_name(0))
    t0 = time.time()
    if what in ('corr', 'all'):
        bench_corr()
    if what in ('dcn', 'all'):
        bench_dcn()
    if what in ('attn', 'all'):
        bench_attn()
    if what in ('conv', 'all'):
        bench_conv()
    if what in ('conv3x3', 'all'):
        bench_conv3x3()
    if what in ('stylegan', 'fused_act', 'upfirdn2d', 'all'):
        bench_stylegan_ops()
    print(f'total {time.time()-t0:.1f}s')
