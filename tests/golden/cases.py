"""Input recipes shared by the golden generator and the tests (pure numpy, no reference)."""
import numpy as np

import synth

CORR_RAND = [(256, 12, 14, 0), (256, 20, 24, 1), (256, 40, 40, 2), (256, 33, 47, 3), (64, 16, 16, 4), (128, 9, 21, 5),
             (256, 3, 3, 6), (256, 3, 40, 7)]


def corr_cases():
    """yield (name, feat_in [C,h,w], feat_ref [C,h,w]) -- raw (un-normalised) feature maps."""
    for (c, h, w, seed) in CORR_RAND:
        name = f'rand_c{c}_{h}x{w}'
        yield name, synth.randn(name + '/in', (c, h, w), seed), synth.randn(name + '/ref', (c, h, w), seed)
    # planted correspondences: ref = rolled input + noise (what the synthetic benchmark uses)
    name = 'planted_c256_24x28'
    fin = synth.randn(name + '/in', (256, 24, 28), 0)
    fref = np.roll(fin, (5, -7), axis=(1, 2)) + synth.randn(name + '/n', (256, 24, 28), 0, 0.05)
    yield name, fin, fref.astype(np.float32)
    # exact ties: the ref map is periodic, so identical ref patches recur; the lowest index must
    # win (torch CPU max semantics)
    name = 'ties_c256_16x20'
    fin = synth.randn(name + '/in', (256, 16, 20), 0)
    base = synth.randn(name + '/ref', (256, 4, 5), 0)
    yield name, fin, np.tile(base, (1, 4, 4))
    # non-negative (post-ReLU-like) features: small margins
    name = 'relu_c256_24x24'
    yield (name, np.maximum(synth.randn(name + '/in', (256, 24, 24), 0), 0),
           np.maximum(synth.randn(name + '/ref', (256, 24, 24), 0), 0))


def corr160_cases():
    """BASELINE configs[1] feature size (256 x 160 x 160): a planted-correspondence pair with noise (the benchmark's structure)
    and a smooth pair with many near-ties (low-pass filtered noise: neighbouring patches are similar)"""
    name = 'planted_c256_160x160'
    fin = synth.randn(name + '/in', (256, 160, 160), 0)
    fref = np.roll(fin, (17, -23), axis=(1, 2)) + synth.randn(name + '/n', (256, 160, 160), 0, 0.1)
    yield name, fin, fref.astype(np.float32)
    name = 'smooth_c256_160x160'
    a = synth.randn(name + '/in', (256, 160, 160), 0)
    b = synth.randn(name + '/ref', (256, 160, 160), 0)

    def lowpass(x):
        for ax in (1, 2):
            x = (np.roll(x, 1, ax) + 2 * x + np.roll(x, -1, ax)) * 0.25
        return x.astype(np.float32)
    yield name, lowpass(lowpass(a)) + 0.5 * a.mean(0, keepdims=True), (lowpass(lowpass(0.7 * a + 0.3 * b))).astype(np.float32)


# feature_match_index in its general form: (name, C, (h, w), (hr, wr), patch, input_stride, ref_stride, is_norm, norm_input)
FMI_GENERAL = [('p5_s1', 32, (14, 17), (14, 17), 5, 1, 1, True, True),
               ('p3_s2_1', 64, (21, 24), (21, 24), 3, 2, 1, True, False),
               ('p4_s2_3_sizes', 48, (20, 26), (17, 29), 4, 2, 3, True, True),
               ('p1_s1', 40, (9, 11), (12, 7), 1, 1, 1, True, True),
               ('p3_s1_nonorm', 256, (10, 12), (10, 12), 3, 1, 1, False, False),
               ('p7_s3_2', 16, (25, 23), (22, 31), 7, 3, 2, True, True)]


def fmi_general_cases():
    """yield (name, feat_in, feat_ref, kwargs): maps used as given (feature_match_index does not normalise pixels)"""
    for name, c, (h, w), (hr, wr), p, si, sr, is_norm, norm_input in FMI_GENERAL:
        fin = synth.randn(f'fmi/{name}/in', (c, h, w), 0)
        fref = synth.randn(f'fmi/{name}/ref', (c, hr, wr), 0)
        yield name, fin, fref, dict(patch_size=p, input_stride=si, ref_stride=sr, is_norm=is_norm, norm_input=norm_input)
