"""Deterministic synthetic tensors / weights shared by the golden generator (build container, with the
reference imported) and by the tests (anywhere).  numpy PCG64 streams keyed by name: no torch RNG,
no files.  Fixtures store a checksum of every synthesised input so RNG drift is detected, not
silently mis-compared."""
import hashlib
import zlib

import numpy as np


def _rng(key, seed=0):
    return np.random.default_rng([zlib.crc32(key.encode()), seed])


def randn(key, shape, seed=0, scale=1.0):
    return (_rng(key, seed).standard_normal(shape) * scale).astype(np.float32)


def rand(key, shape, seed=0):
    return _rng(key, seed).random(shape).astype(np.float32)


def checksum(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def image(key, c, h, w, seed=0):
    """Smooth-ish image in [0,1]: a few random sinusoids + 3% noise (so bicubic/VGG see structure)."""
    r = _rng(key, seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing='ij')
    img = np.zeros((c, h, w))
    for ch in range(c):
        for _ in range(6):
            fy, fx = r.uniform(0.02, 0.45, 2)
            ph = r.uniform(0, 2 * np.pi)
            img[ch] += r.uniform(0.3, 1.0) * np.sin(2 * np.pi * (fy * yy + fx * xx) + ph)
    img = 0.5 + img / 6.0 + 0.03 * r.standard_normal((c, h, w))
    return np.clip(img, 0.0, 1.0).astype(np.float32)


def sr_sample(key, k_refs, lr_h, lr_w, seed=0, noise=0.02):
    """One synthetic (LR, up, K refs, GT) sample of the shape MultiRef*Dataset yields
    (multi_ref_dataset.py:127-134): refs are rolled copies of GT + noise so real correspondences
    exist; LR is the 4x4 box average of GT; 'up' is LR repeated 4x (nearest) then 3x3-box smoothed."""
    h, w = 4 * lr_h, 4 * lr_w
    gt = image(key + '/gt', 3, h, w, seed)
    lq = gt.reshape(3, lr_h, 4, lr_w, 4).mean(axis=(2, 4)).astype(np.float32)
    up = np.repeat(np.repeat(lq, 4, axis=1), 4, axis=2)
    pad = np.pad(up, ((0, 0), (1, 1), (1, 1)), mode='edge')
    up = sum(pad[:, i:i + h, j:j + w] for i in range(3) for j in range(3)) / 9.0
    r = _rng(key + '/refs', seed)
    refs = []
    for k in range(1, k_refs + 1):
        rr = np.roll(gt, shift=((5 * k) % h, (-7 * k) % w), axis=(1, 2))
        rr = np.clip(rr + noise * r.standard_normal(rr.shape), 0, 1).astype(np.float32)
        refs.append(rr)
    return dict(img_in_lq=lq, img_in_up=up.astype(np.float32), img_ref_list=np.stack(refs), img_in=gt)


_IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
_IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)


def state_dict(spec, seed=0):
    """spec: iterable of (key, shape).  Returns {key: float32 ndarray}; a pure function of
    (key, shape, seed).  Scales keep activations O(1) through the 48 residual blocks."""
    out = {}
    for key, shape in spec:
        shape = tuple(int(s) for s in shape)
        if key.endswith('mean') and shape == (1, 3, 1, 1):
            out[key] = _IMAGENET_MEAN.copy()
        elif key.endswith('std') and shape == (1, 3, 1, 1):
            out[key] = _IMAGENET_STD.copy()
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.0
            if '.body' in key or 'body_' in key:
                gain = 0.1
            if 'conv_offset_mask' in key:
                gain = 0.3
            if 'tail_large.2' in key:
                gain = 0.1  # keeps the SR residual O(0.5): pixel tolerances then mean what they say
            out[key] = randn(key, shape, seed, gain * (2.0 / fan_in) ** 0.5)
        elif len(shape) == 1 and shape[0] == 1:  # PReLU slope
            out[key] = np.full(shape, 0.25, np.float32)
        elif len(shape) == 1:
            out[key] = randn(key, shape, seed, 0.02)
        else:
            out[key] = randn(key, shape, seed, 0.05)
    return out
