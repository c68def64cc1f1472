"""VGG feature taps without torchvision.

Mirror of basicsr/archs/vgg_arch.py:54-161 (VGGFeatureExtractor, layer names of NAMES :10-33,
state-dict keys ``vgg_net.<name>.{weight,bias}`` + buffers ``mean`` / ``std``).  The reference
takes the layers from ``torchvision.models.vgg*(pretrained=True)``; torchvision and the download
are unavailable here, so the stack is built from the public VGG configuration and weights come
from (a) ``experiments/pretrained_models/vgg19-dcbb9e9d.pth`` (torchvision key layout
``features.N.*``) when that file exists, as in the reference (:9,103-108), or (b) a later
``load_state_dict``; otherwise they stay at their random init (a warning is logged).
These are plain 3x3 convolutions: MIOpen via PyTorch-ROCm ("stays PyTorch").
"""
import logging
import os
from collections import OrderedDict

import torch
from torch import nn as nn

from ..utils.registry import ARCH_REGISTRY

VGG_PRETRAIN_PATH = 'experiments/pretrained_models/vgg19-dcbb9e9d.pth'

_CFG = {
    'vgg11': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'vgg13': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'vgg16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
    'vgg19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M'],
}


def layer_names(vgg_type):
    """conv{b}_{i}, relu{b}_{i}, pool{b} in network order (== NAMES[vgg_type] of the reference)."""
    names, block, idx = [], 1, 1
    for v in _CFG[vgg_type]:
        if v == 'M':
            names.append(f'pool{block}')
            block, idx = block + 1, 1
        else:
            names += [f'conv{block}_{idx}', f'relu{block}_{idx}']
            idx += 1
    return names


NAMES = {k: layer_names(k) for k in _CFG}


def build_vgg_layers(vgg_type, last_name, use_bn=False, remove_pooling=False, pooling_stride=2):
    """OrderedDict name -> layer up to and including ``last_name``; also the torchvision
    ``features`` index of every conv (for loading torchvision checkpoints)."""
    layers, tv_index = OrderedDict(), {}
    cin, block, idx, tv_i = 3, 1, 1, 0
    for v in _CFG[vgg_type]:
        if v == 'M':
            name = f'pool{block}'
            if not remove_pooling:
                layers[name] = nn.MaxPool2d(kernel_size=2, stride=pooling_stride)
            block, idx, tv_i = block + 1, 1, tv_i + 1
            if name == last_name:
                break
            continue
        cname, rname = f'conv{block}_{idx}', f'relu{block}_{idx}'
        layers[cname] = nn.Conv2d(cin, v, 3, padding=1)
        tv_index[cname] = tv_i
        tv_i += 1
        if cname == last_name:
            break
        if use_bn:
            layers[f'bn{block}_{idx}'] = nn.BatchNorm2d(v)
            tv_i += 1
            if f'bn{block}_{idx}' == last_name:
                break
        layers[rname] = nn.ReLU(inplace=True)
        tv_i += 1
        cin, idx = v, idx + 1
        if rname == last_name:
            break
    return layers, tv_index


def load_torchvision_vgg(layers, tv_index, path):
    state = torch.load(path, map_location='cpu')
    for name, i in tv_index.items():
        with torch.no_grad():
            layers[name].weight.copy_(state[f'features.{i}.weight'])
        with torch.no_grad():
            layers[name].bias.copy_(state[f'features.{i}.bias'])


@ARCH_REGISTRY.register()
class VGGFeatureExtractor(nn.Module):

    def __init__(self, layer_name_list, vgg_type='vgg19', use_input_norm=True, range_norm=False, requires_grad=False,
                 remove_pooling=False, pooling_stride=2):
        super().__init__()
        self.layer_name_list = layer_name_list
        self.use_input_norm = use_input_norm
        self.range_norm = range_norm
        use_bn = 'bn' in vgg_type
        base = vgg_type.replace('_bn', '')
        self.names = NAMES[base]
        if use_bn:  # insert_bn of the reference (:36-51)
            self.names = [n2 for n in self.names for n2 in ([n, 'bn' + n[4:]] if n.startswith('conv') else [n])]
        last = max(layer_name_list, key=self.names.index)
        layers, tv_index = build_vgg_layers(base, last, use_bn, remove_pooling, pooling_stride)
        if os.path.exists(VGG_PRETRAIN_PATH) and not use_bn:
            load_torchvision_vgg(layers, tv_index, VGG_PRETRAIN_PATH)
        else:
            logging.getLogger('basicsr').warning(
                f'VGGFeatureExtractor: {VGG_PRETRAIN_PATH} not found and no download possible; '
                'weights are randomly initialised until a state dict is loaded.')
        self.vgg_net = nn.Sequential(layers)
        if not requires_grad:
            self.vgg_net.eval()
            for p in self.parameters():
                p.requires_grad = False
        else:
            self.vgg_net.train()
            for p in self.parameters():
                p.requires_grad = True
        if self.use_input_norm:
            self.register_buffer('mean', torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
            self.register_buffer('std', torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, x):
        from . import nhwc
        from .arch_util import run_conv_relu_stack
        if nhwc.active(x) and nhwc.stack_ok(self.vgg_net):
            # channels-last engine; taps are logical NCHW views of [N,h,w,C] storage; the input normalisation (:150-153 of the
            # reference) rides on the channels-last packing of the image
            x4 = nhwc.image_to_nhwc4(x, self.mean if self.use_input_norm else None, self.std if self.use_input_norm else None, self.range_norm)
            feats = nhwc.vgg_stack(self.vgg_net, x4, taps=self.layer_name_list)
            return {k: nhwc.as_nchw(v) for k, v in feats.items()}
        if self.range_norm:
            x = (x + 1) / 2
        if self.use_input_norm:
            x = (x - self.mean) / self.std
        return run_conv_relu_stack(self.vgg_net, x, taps=self.layer_name_list)
