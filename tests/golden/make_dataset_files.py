"""Synthetic on-disk datasets (PNG + csv) in the two formats of multi_ref_dataset.py, written from
synth.image so the generator (reference run) and the tests (our run) read identical files."""
import os

import numpy as np
from PIL import Image

import synth


def _png(path, key, h, w):
    img = (synth.image(key, 3, h, w) * 255).round().astype(np.uint8).transpose(1, 2, 0)
    Image.fromarray(img).save(path)


def make_cufed(root):
    os.makedirs(root, exist_ok=True)
    for name, (h, w) in (('000', (45, 38)), ('001', (33, 50))):
        _png(os.path.join(root, f'{name}_0.png'), f'cufed/{name}/0', h, w)
        for k in range(1, 6):
            _png(os.path.join(root, f'{name}_{k}.png'), f'cufed/{name}/{k}', h - k, w + k)
    return dict(name='CUFED5', type='MultiRefCUFEDSet', dataroot_in=root, dataroot_ref=root, scale=4)


def make_megadepth(root):
    scene = os.path.join(root, '0001')
    os.makedirs(scene, exist_ok=True)
    rows = []
    for i in range(2):
        names = [f't{i}.png'] + [f'r{i}_{k}.png' for k in range(5)]
        for n in names:
            _png(os.path.join(scene, n), f'mega/{n}', 72, 80)
        pts = [(30 + 2 * k, 34 + k) for k in range(6)]
        rows.append(names + [str(p) for p in pts] + ['0001'])
    ann = os.path.join(root, 'ann.csv')
    with open(ann, 'w') as f:
        f.write('target,H,M1,M2,L1,L2,p0,p1,p2,p3,p4,p5,scene\n')
        for r in rows:
            f.write(','.join(f'"{c}"' if c.startswith('(') else c for c in r) + '\n')
    return dict(name='Mega', type='MultiRefMegaDepthDataset', dataroot_in=root, dataroot_ref=root, ann_file=ann,
                gt_size=32, use_flip=True, use_rot=True, scale=4)
