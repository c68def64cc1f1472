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
        if self.patch_size == 3 and self.stride == 1:
            idx = match_normalised_batch(feat_in, feat_ref)
        else:
            # any other patch size / stride (the reference's ctor takes them, :14-28; no shipped yml does): the general
            # kernel (mrefsr_feature_match_index_f32) pair by pair on the per-pixel-normalised maps (:57-68).  As in the
            # reference, the index grid is ((h - p) / s + 1, (w - p) / s + 1) and index_to_flow (:30-47) pads it by 2:
            # only p = 3, s = 1 gives offsets of the feature maps' size.
            from .ref_map_util import feature_match_index
            b = feat_in.shape[0]
            nrm = [torch.nn.functional.normalize(f.reshape(f.shape[0], f.shape[1], -1).float(), dim=1).view_as(f) for f in (feat_in, feat_ref)]
            idx = torch.stack([feature_match_index(nrm[0][i % b], nrm[1][i], self.patch_size, self.stride, self.stride, True, True)[0]
                               for i in range(feat_ref.shape[0])])
            h, w = idx.shape[1] + 2, idx.shape[2] + 2
        offs = hip.offsets_from_idx(idx.contiguous(), h, w)
        return {'relu3_1': offs[1], 'relu2_1': offs[2], 'relu1_1': offs[4]}, idx

    def forward(self, dense_features, img_ref_hr):
        pre_offset, _ = self.offsets(dense_features['dense_features1'], dense_features['dense_features2'])
        img_ref_feat = self.vgg(img_ref_hr)
        return pre_offset, img_ref_feat
