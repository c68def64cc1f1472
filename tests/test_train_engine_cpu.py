"""CPU: the host-side pieces of the channels-last training engine (mrefsr_amd/archs/nhwc_train.py) that need no GPU."""
import math

import torch
import torch.nn.functional as F


def test_unshuffle_inverts_pixel_shuffle_on_channels_last_storage():
    from mrefsr_amd.archs.nhwc_train import _unshuffle
    x = torch.randn(2, 12, 5, 7)                                  # NCHW, 12 = 3 * 2 * 2 channels
    y = F.pixel_shuffle(x, 2)                                     # [2, 3, 10, 14]
    back = _unshuffle(y.permute(0, 2, 3, 1).contiguous())         # NHWC in, NHWC out
    assert torch.equal(back, x.permute(0, 2, 3, 1))


def test_weight_scale_puts_the_largest_weight_into_the_fp16_sweet_spot():
    from mrefsr_amd.archs.nhwc_train import _scale_of
    for amax in (3e-5, 0.02, 0.7, 1.0, 13.0, 4096.0):
        s = _scale_of(amax)
        assert s == 2.0 ** round(math.log2(s))                    # a power of two: scaling is exact
        assert 2.0 ** 13 <= amax * s < 2.0 ** 14
        assert amax * s * 4 < 65504                                # the 4x headroom check_scales() relies on
    assert _scale_of(0.0) is None and _scale_of(float('nan')) is None and _scale_of(float('inf')) is None


def test_training_engine_is_not_engaged_without_a_graph_or_on_the_cpu():
    from mrefsr_amd.archs import nhwc, nhwc_train
    w = torch.nn.Parameter(torch.randn(4, 4, 3, 3))
    x = torch.randn(1, 8, 8, 4)
    assert nhwc_train.recording(w, x)
    with torch.no_grad():
        assert not nhwc_train.recording(w, x)
    assert not nhwc_train.recording(w.detach(), x)
    assert not nhwc.train_active(x)                               # CPU tensor: the engine never applies (no CPU path)
