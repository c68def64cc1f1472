"""On-disk formats of the path's two datasets, without cv2 / mmcv (PIL + numpy + pandas only).

Mirror of basicsr/data/multi_ref_dataset.py:
  MultiRefMegaDepthDataset :19-139  csv columns target,H,M1,M2,L1,L2,p0..p5,scene; gt_size crops centred
                                    on the matched key-points; reference order shuffled; flip / rot
                                    augmentation shared by all images; PIL-bicubic LR and up-sampled LR
  MultiRefCUFEDSet        :142-227  <name>_0.png input, <name>_1..5.png references; mod_crop, zero pad to
                                    500x500 (bottom / right), PIL-bicubic LR / up; returns RGB CHW fp32
Returned dict keys and tensor layouts are the reference's (:127-134, :215-225).
Random draws are consumed in the reference's order (random.shuffle, then the three
`flag and random.random() < 0.5` of transforms.augment :116-118), so seeding `random` reproduces it.
"""
import glob
import os.path as osp
import random
from ast import literal_eval

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

from . import DATASET_REGISTRY


def _to_tensor(img):
    """HWC float32 -> CHW tensor (img2tensor, img_util.py:22-30, without the cv2 colour swap)"""
    return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).float()


def _bicubic_pair(img_u8, scale):
    """uint8 HWC -> (LR, up-sampled LR) with PIL's bicubic kernel, as fp32 [0,1] arrays"""
    h, w = img_u8.shape[:2]
    lq = Image.fromarray(img_u8).resize((w // scale, h // scale), Image.BICUBIC)
    up = lq.resize((w, h), Image.BICUBIC)
    return np.array(lq).astype(np.float32) / 255., np.array(up).astype(np.float32) / 255.


def mod_crop(img, scale):
    """transforms.py:6-23"""
    if img.ndim not in (2, 3):
        raise ValueError(f'Wrong img ndim: {img.ndim}.')
    h, w = img.shape[:2]
    return img[:h - h % scale, :w - w % scale, ...].copy()


def impad(img, shape, pad_val=0):
    """mmcv.impad(img, shape=(h, w), pad_val): pad at the bottom / right up to `shape`"""
    h, w = img.shape[:2]
    out = np.full((max(shape[0], h), max(shape[1], w)) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:h, :w] = img
    return out


def augment(imgs, hflip=True, rotation=True):
    """transforms.py:94-154 for images: same random draws, numpy instead of cv2.flip"""
    hflip = hflip and random.random() < 0.5
    vflip = rotation and random.random() < 0.5
    rot90 = rotation and random.random() < 0.5

    def _augment(img):
        if hflip:
            img = img[:, ::-1]
        if vflip:
            img = img[::-1]
        if rot90:
            img = img.transpose(1, 0, 2)
        return np.ascontiguousarray(img)

    return [_augment(img) for img in imgs]


@DATASET_REGISTRY.register()
class MultiRefMegaDepthDataset(data.Dataset):

    def __init__(self, opt):
        super().__init__()
        import pandas as pd
        self.opt = opt
        self.in_folder, self.ref_folder = opt['dataroot_in'], opt['dataroot_ref']
        self.ann_file = opt['ann_file']
        self.samples = []
        df = pd.read_csv(self.ann_file, dtype={'scene': 'string'})
        for i in range(len(df)):
            target, H, M1, M2, L1, L2, p0, p1, p2, p3, p4, p5, scene = df.loc[i].tolist()
            refs = [osp.join(self.in_folder, scene, r) for r in (H, M1, M2, L1, L2)]
            self.samples.append((osp.join(self.in_folder, scene, target), refs, np.array(literal_eval(p0)),
                                 [np.array(literal_eval(p)) for p in (p1, p2, p3, p4, p5)]))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        scale = self.opt['scale']
        in_path, ref_paths, p0, p_refs = self.samples[index]
        img_in = np.array(Image.open(in_path).convert('RGB')).astype(np.float32) / 255.
        refs = [np.array(Image.open(p).convert('RGB')).astype(np.float32) / 255. for p in ref_paths]
        g = self.opt['gt_size']
        img_in = img_in[p0[1] - g // 2:p0[1] + g // 2, p0[0] - g // 2:p0[0] + g // 2]
        refs = [r[p[1] - g // 2:p[1] + g // 2, p[0] - g // 2:p[0] + g // 2] for r, p in zip(refs, p_refs)]
        random.shuffle(refs)
        imgs = augment([img_in] + refs, self.opt['use_flip'], self.opt['use_rot'])
        img_in, refs = imgs[0], imgs[1:]
        in_lq, in_up = _bicubic_pair((img_in * 255).astype(np.uint8), scale)
        pairs = [_bicubic_pair((r * 255).astype(np.uint8), scale) for r in refs]
        return {
            'img_in': _to_tensor(img_in),
            'img_in_lq': _to_tensor(in_lq),
            'img_in_up': _to_tensor(in_up),
            'img_ref_list': torch.stack([_to_tensor(r) for r in refs]),
            'img_ref_lq_list': torch.stack([_to_tensor(p[0]) for p in pairs]),
            'img_ref_up_list': torch.stack([_to_tensor(p[1]) for p in pairs]),
        }


@DATASET_REGISTRY.register()
class MultiRefCUFEDSet(data.Dataset):

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.input_list = sorted(glob.glob(osp.join(opt['dataroot_in'], '*_0.png')))
        self.ref_lists = [sorted(glob.glob(osp.join(opt['dataroot_ref'], f'*_{k}.png'))) for k in range(1, 6)]

    def __len__(self):
        return len(self.input_list)

    def __getitem__(self, idx):
        scale = self.opt['scale']
        load = lambda p: np.array(Image.open(p).convert('RGB'))  # noqa: E731  (RGB; the reference loads BGR and swaps at the end)
        img_in = mod_crop(load(self.input_list[idx]), scale)
        refs = [load(lst[idx]) for lst in self.ref_lists]
        img_in_gt = img_in.copy()
        img_in_h, img_in_w = img_in.shape[:2]
        img_in = impad(img_in, shape=(500, 500), pad_val=0)
        refs = [impad(r, shape=(500, 500), pad_val=0) for r in refs]
        in_lq, in_up = _bicubic_pair(img_in, scale)
        pairs = [_bicubic_pair(r, scale) for r in refs]
        return {
            'img_in': _to_tensor(img_in_gt.astype(np.float32) / 255.),
            'img_in_lq': _to_tensor(in_lq),
            'img_in_up': _to_tensor(in_up),
            'img_ref_list': torch.stack([_to_tensor(r.astype(np.float32) / 255.) for r in refs]),
            'img_ref_lq_list': torch.stack([_to_tensor(p[0]) for p in pairs]),
            'img_ref_up_list': torch.stack([_to_tensor(p[1]) for p in pairs]),
            'lq_path': self.ref_lists[0][idx].replace('_1.png', '_multi.png'),
            'padding': True,
            'original_size': (img_in_h, img_in_w),
        }
