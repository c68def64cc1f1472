"""Mirror of basicsr/archs/contras_multi_extractor_arch.py:10-64 (and of the single-reference
twin contras_extractor_arch.py): two VGG16 stacks cut at conv3_1 (no ReLU after it), ImageNet
mean / std buffers; state-dict keys ``feature_extraction_image{1,2}.model.<name>.*``.

Beyond the reference: ``forward_stacked`` runs the shared reference stack ONCE over all K
references stacked on the batch axis (same arithmetic per image, one launch sequence instead
of K)."""
import logging

import torch
import torch.nn as nn

from ..utils.registry import ARCH_REGISTRY
from .vgg_arch import build_vgg_layers


class ContrasExtractorLayer(nn.Module):

    def __init__(self):
        super().__init__()
        layers, _ = build_vgg_layers('vgg16', 'conv3_1')
        self.model = nn.Sequential(layers)
        self.register_buffer('mean', torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, batch):
        from . import nhwc
        from .arch_util import run_conv_relu_stack
        if nhwc.active(batch) and nhwc.stack_ok(self.model):
            # channels-last engine; the result is a logical NCHW view of [N,h,w,256] storage
            return nhwc.as_nchw(nhwc.vgg_stack(self.model, nhwc.image_to_nhwc4(batch, self.mean, self.std)))
        return run_conv_relu_stack(self.model, (batch - self.mean) / self.std)


@ARCH_REGISTRY.register()
class ContrasMultiExtractorSep(nn.Module):

    def __init__(self):
        super().__init__()
        logging.getLogger('basicsr').info('ContrasMultiExtractorSep: VGG16 weights are random until '
                                          'pretrain_network_feature_extractor is loaded (no download possible).')
        self.feature_extraction_image1 = ContrasExtractorLayer()
        self.feature_extraction_image2 = ContrasExtractorLayer()

    def forward(self, image1, image_list):
        dense_features1 = self.feature_extraction_image1(image1)
        return [{'dense_features1': dense_features1, 'dense_features2': self.feature_extraction_image2(image2)}
                for image2 in image_list]

    def forward_stacked(self, image1, images2):
        """image1 [B,3,H,W], images2 [K*B,3,H,W] (k-major) -> (feat1 [B,256,h,w], feat2 [K*B,256,h,w])."""
        return self.feature_extraction_image1(image1), self.feature_extraction_image2(images2)


@ARCH_REGISTRY.register()
class ContrasExtractorSep(nn.Module):
    """single-reference twin (contras_extractor_arch.py:46-62)"""

    def __init__(self):
        super().__init__()
        self.feature_extraction_image1 = ContrasExtractorLayer()
        self.feature_extraction_image2 = ContrasExtractorLayer()

    def forward(self, image1, image2):
        return {'dense_features1': self.feature_extraction_image1(image1),
                'dense_features2': self.feature_extraction_image2(image2)}
