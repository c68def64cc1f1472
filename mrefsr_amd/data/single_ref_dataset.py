"""Mirror of basicsr/data/single_ref_dataset.py:19-123 (SingleRefMegaDepthDataset): the csv layout of the
multi-reference MegaDepth set, one of the five references per sample drawn with
``np.random.permutation(5)[0]`` (:69), keypoint-centred gt_size crops, flip / rot90 augmentation, PIL
bicubic down / up.  Returns the six tensors of :111-118."""
import os.path as osp
from ast import literal_eval

import numpy as np
from PIL import Image
from torch.utils import data as data

from . import DATASET_REGISTRY
from .multi_ref_dataset import _bicubic_pair, _to_tensor, augment


@DATASET_REGISTRY.register()
class SingleRefMegaDepthDataset(data.Dataset):

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
        pick = np.random.permutation(5)[0]
        img_in = np.array(Image.open(in_path).convert('RGB')).astype(np.float32) / 255.
        img_ref = np.array(Image.open(ref_paths[pick]).convert('RGB')).astype(np.float32) / 255.
        g = self.opt['gt_size']
        pr = p_refs[pick]
        img_in = img_in[p0[1] - g // 2:p0[1] + g // 2, p0[0] - g // 2:p0[0] + g // 2]
        img_ref = img_ref[pr[1] - g // 2:pr[1] + g // 2, pr[0] - g // 2:pr[0] + g // 2]
        img_in, img_ref = augment([img_in, img_ref], self.opt['use_flip'], self.opt['use_rot'])
        in_lq, in_up = _bicubic_pair((img_in * 255).astype(np.uint8), scale)
        ref_lq, ref_up = _bicubic_pair((img_ref * 255).astype(np.uint8), scale)
        return {'img_in': _to_tensor(img_in), 'img_in_lq': _to_tensor(in_lq), 'img_in_up': _to_tensor(in_up),
                'img_ref': _to_tensor(img_ref), 'img_ref_lq': _to_tensor(ref_lq), 'img_ref_up': _to_tensor(ref_up)}
