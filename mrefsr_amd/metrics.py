"""PSNR / SSIM protocol of the reference's validation loop, numpy only (no cv2):
tensor2img (basicsr/utils/img_util.py:38-94: squeeze, clamp to [0,1], HWC, *255, round, uint8) and
calculate_psnr (basicsr/metrics/psnr_ssim.py:11-48: crop_border, float64 MSE, 10 log10(255^2/mse)).
The reference swaps RGB->BGR with cv2 before the metric; PSNR over all channels is invariant to
that permutation, so it is not reproduced (Y-channel metrics use explicit BT.601 weights)."""
import numpy as np
import torch


def tensor2img(tensor, out_type=np.uint8, min_max=(0, 1)):
    """(1|-,3|1,H,W) tensor -> HWC (or HW) ndarray, RGB order kept."""
    t = tensor.squeeze(0).float().detach().cpu().clamp_(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 3:
        img = t.numpy().transpose(1, 2, 0)
        if img.shape[2] == 1:
            img = np.squeeze(img, axis=2)
    elif t.dim() == 2:
        img = t.numpy()
    else:
        raise TypeError(f'Only support 3D or 2D tensor per image. But received with dimension: {t.dim()}')
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return img.astype(out_type)


def rgb_to_y(img):
    """uint8/float [0,255] RGB HWC -> Y of YCbCr in [0,255] (float, not rounded):
    metric_util.py:31-45 via color_util bgr2ycbcr(y_only=True) coefficients."""
    img = img.astype(np.float32) / 255.
    y = np.dot(img, [65.481, 128.553, 24.966]) + 16.0
    return (y / 255.)[..., None] * 255.


def calculate_psnr(img, img2, crop_border, test_y_channel=False):
    assert img.shape == img2.shape, f'Image shapes are different: {img.shape}, {img2.shape}.'
    if img.ndim == 2:
        img, img2 = img[..., None], img2[..., None]
    if crop_border != 0:
        img = img[crop_border:-crop_border, crop_border:-crop_border, ...]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        img, img2 = rgb_to_y(img), rgb_to_y(img2)
    mse = np.mean((img.astype(np.float64) - img2.astype(np.float64))**2)
    if mse == 0:
        return float('inf')
    return 10. * np.log10(255. * 255. / mse)


def _gaussian_window(size=11, sigma=1.5):
    """the 1-D taps cv2.getGaussianKernel(11, 1.5) returns: exp(-(i - c)^2 / (2 sigma^2)), normalised to sum 1"""
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def _filter_valid(img, k):
    """separable correlation with taps k along both axes, 'valid' region only (= cv2.filter2D(...)[5:-5, 5:-5] with the
    outer-product window of psnr_ssim.py:187-190; the border handling of filter2D never reaches that region)"""
    n = len(k)
    rows = sum(k[i] * img[i:img.shape[0] - n + 1 + i, :] for i in range(n))
    return sum(k[j] * rows[:, j:rows.shape[1] - n + 1 + j] for j in range(n))


def _ssim(img, img2):
    """single-channel SSIM map mean (psnr_ssim.py:172-200): 11 x 11 Gaussian window sigma 1.5, c1 = (0.01*255)^2, c2 = (0.03*255)^2"""
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    k = _gaussian_window()
    mu1, mu2 = _filter_valid(img, k), _filter_valid(img2, k)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _filter_valid(img * img, k) - mu1_sq
    s2 = _filter_valid(img2 * img2, k) - mu2_sq
    s12 = _filter_valid(img * img2, k) - mu1_mu2
    return float((((2 * mu1_mu2 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))).mean())


def calculate_ssim(img, img2, crop_border, test_y_channel=False):
    """psnr_ssim.py:85-129 (HWC uint8 / [0,255] images): per-channel SSIM averaged; Y channel as in calculate_psnr"""
    assert img.shape == img2.shape, f'Image shapes are different: {img.shape}, {img2.shape}.'
    if img.ndim == 2:
        img, img2 = img[..., None], img2[..., None]
    if crop_border != 0:
        img = img[crop_border:-crop_border, crop_border:-crop_border, ...]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        img, img2 = rgb_to_y(img), rgb_to_y(img2)
    img, img2 = img.astype(np.float64), img2.astype(np.float64)
    return float(np.mean([_ssim(img[..., i], img2[..., i]) for i in range(img.shape[2])]))


def imwrite(img, path):
    """HWC uint8 RGB image -> PNG (the reference converts to BGR and lets cv2.imwrite convert back: same file)"""
    import os
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(img).save(path)


def batch_psnr(output, gt, crop_border=4):
    """mean PSNR over a batch of (B,3,H,W) tensors with the protocol above."""
    vals = [calculate_psnr(tensor2img(output[i:i + 1]), tensor2img(gt[i:i + 1]), crop_border) for i in range(output.shape[0])]
    return float(np.mean(vals)), vals
