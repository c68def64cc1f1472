#!/usr/bin/env python3
"""candidate statistics of the pre-filter pass (random vs planted data, bf16 vs fp16 operand)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrefsr_amd import hip
hip._timing['keep_ws'] = True
b, k, h, w = 8, 5, 160, 160
torch.manual_seed(0)
fin = torch.randn(b, 256, h, w, device='cuda'); fref = torch.randn(k * b, 256, h, w, device='cuda')
fpl = torch.cat([torch.roll(fin, (17 * (kk + 1), -23 * (kk + 1)), (2, 3)) + 0.3 * torch.randn_like(fin) for kk in range(k)])
for split in ('bf16', 'fp16'):
    yi, n2i, bi = hip.pixnorm(fin, want_bf16_split=True, split=split)
    nei, _ = hip.patch_norm(n2i)
    for name, fr in (('random', fref), ('planted', fpl)):
        yr, n2r, br = hip.pixnorm(fr, want_bf16_split=True, split=split)
        _, invr = hip.patch_norm(n2r)
        hip.corr_top1(yi, yr, invr, nei, h, w, ybf_in=bi, ybf_ref=br)
        torch.cuda.synchronize()
        ws, n_pair, P = hip._timing['last_corr_ws']
        wi = ws.view(torch.int32)
        cand_n = wi[n_pair * P * 16: n_pair * P * 17]
        flag_count = int(wi[n_pair * P * 18].item())
        cn = cand_n.float()
        print(f'{split} {name}: flagged(brute force)={flag_count} of {n_pair*P}  mean candidates={cn[cn>=0].mean().item():.3f}  '
              f'max={int(cand_n.max())}  hist={[int((cand_n==i).sum()) for i in range(-1,17)]}')
