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
