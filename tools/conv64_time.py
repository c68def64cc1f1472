import os, sys
sys.path.insert(0, '/root/repo')
import torch
from mrefsr_amd import hip
sys.path.insert(0, '/root/repo/tools')
def timeit(fn, warm=2, iters=7):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    evs=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a,b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts=sorted(a.elapsed_time(b) for a,b in evs); return ts[len(ts)//2]
torch.manual_seed(0)
for n,h,res in ((8,640,False),(8,640,True),(40,640,False),(8,320,True),(8,320,False),(8,160,True)):
    x=torch.randn(n,h,h,64,device='cuda'); r=torch.randn(n,h,h,64,device='cuda') if res else None
    pk=hip.conv_pack_weight(torch.randn(64,64,3,3,device='cuda')*0.03,16); b=torch.randn(64,device='cuda')
    t=timeit(lambda: hip.conv_nhwc(x,pk,b,64,3,residual=r,act=not res,slope=0.0))
    print(f'N={n} {h}x{h} 64->64 res={int(res)}: {t:.3f} ms  {2.0*n*h*h*64*64*9/t/1e9:.1f} TF/s', flush=True)
