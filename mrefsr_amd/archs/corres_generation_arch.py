"""Mirror of basicsr/archs/corres_generation_arch.py:14-118.

forward(dense_features, img_ref_hr) keeps the reference's contract:
    pre_offset  = {'relu3_1': (B,9,h,w,2), 'relu2_1': (B,9,2h,2w,2), 'relu1_1': (B,9,4h,4w,2)}, last dim [x,y]
    img_ref_feat = VGG19 taps of the HR reference image
but the per-sample python loop (:53), the two F.normalize, feature_match_index, index_to_flow and
the 27 tensor_shift / repeat_interleave calls are three HIP launches for the whole batch.
"""
import torch
import torch.nn as nn

from .. import hip
from ..utils.registry import ARCH_REGISTRY
from .ref_map_util import match_normalised_batch
from .vgg_arch import VGGFeatureExtractor


@ARCH_REGISTRY.register()
class CorrespondenceGenerationArch(nn.Module):

    def __init__(self, patch_size=3, stride=1, vgg_layer_list=['relu3_1', 'relu2_1', 'relu1_1'], vgg_type='vgg19'):
        super().__init__()
        if patch_size != 3 or stride != 1:
            raise NotImplementedError('CorrespondenceGenerationArch: HIP path implements patch_size=3, stride=1')
        self.patch_size = patch_size
        self.stride = stride
        self.vgg_layer_list = vgg_layer_list
        self.vgg = VGGFeatureExtractor(layer_name_list=vgg_layer_list, vgg_type=vgg_type)

    def index_to_flow(self, max_idx):
        """(h-2, w-2) int64 -> (1, h, w, 2) fp32 [x, y], zero-padded bottom/right (:30-47)."""
        ph, pw = max_idx.shape
        return hip.offsets_from_idx(max_idx.unsqueeze(0).contiguous(), ph + 2, pw + 2, scales=(1,))[1][:, 0]

    @torch.no_grad()
    def offsets(self, feat_in, feat_ref):
        """feat_in [B,256,h,w], feat_ref [K*B,256,h,w] k-major -> dict of [K*B,9,sh,sw,2]."""
        h, w = feat_in.shape[2:]
        idx = match_normalised_batch(feat_in, feat_ref)
        offs = hip.offsets_from_idx(idx, h, w)
        return {'relu3_1': offs[1], 'relu2_1': offs[2], 'relu1_1': offs[4]}, idx

    def forward(self, dense_features, img_ref_hr):
        pre_offset, _ = self.offsets(dense_features['dense_features1'], dense_features['dense_features2'])
        img_ref_feat = self.vgg(img_ref_hr)
        return pre_offset, img_ref_feat
