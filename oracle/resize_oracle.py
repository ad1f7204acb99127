"""CPU ORACLE (test infrastructure) for the input pipeline's resize — NOT product code.

The reference resizes with torchvision's ``T.Resize`` on PIL images (reference enhancing/dataloader/imagenet.py:31,49), i.e. ``PIL.Image.resize(size,
BILINEAR)``: Pillow's antialiased separable resampler on 8-bit pixels.  Pillow is a third-party dependency (not vendored under /root/reference; this image
ships Pillow 12.2.0); its published algorithm (src/libImaging/Resample.c) is restated here in numpy:
  * ``precompute_coeffs``: per output index the support window [xmin, xmin + n) and the normalised triangle weights, filter scale = max(in / out, 1)
    (antialiasing when shrinking), centre = (x + 0.5) * in / out;
  * ``normalize_coeffs_8bpc``: weights to fixed point with PRECISION_BITS = 32 - 8 - 2 = 22, rounded half away from zero;
  * ``ImagingResampleHorizontal_8bpc`` then ``ImagingResampleVertical_8bpc``: ss = 2^21 + sum pixel * k, result = clip8(ss >> 22); the horizontal pass
    runs first and its uint8 result feeds the vertical pass.
PINNED: tests/test_oracle_cpu.py compares this restatement with Pillow itself (bit-exact) on ragged sizes, up- and down-scaling."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_coeffs(in_size: int, out_size: int):
    """-> (bounds int32 [out, 2] = (first input index, tap count), weights int32 [out, ksize] in 22-bit fixed point)"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                      # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.float64)
    bounds = np.zeros((out_size, 2), np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * inv)
            w = 1.0 - a if a < 1.0 else 0.0
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, kk * (1 << PRECISION_BITS) - 0.5, kk * (1 << PRECISION_BITS) + 0.5).astype(np.int64).astype(np.int32)
    return bounds, fixed


def resize_u8(img: np.ndarray, out_hw) -> np.ndarray:
    """img uint8 [H, W, C] -> uint8 [Ho, Wo, C], PIL.Image.resize((Wo, Ho), BILINEAR)"""
    H, W, C = img.shape
    Ho, Wo = out_hw
    a = img.astype(np.int64)
    if Wo != W:
        b, k = bilinear_coeffs(W, Wo)
        t = np.empty((H, Wo, C), np.int64)
        for xx in range(Wo):
            x0, n = b[xx]
            t[:, xx] = (1 << (PRECISION_BITS - 1)) + (a[:, x0:x0 + n] * k[xx, :n].astype(np.int64)[None, :, None]).sum(1)
        a = np.clip(t >> PRECISION_BITS, 0, 255)
    if Ho != H:
        b, k = bilinear_coeffs(H, Ho)
        t = np.empty((Ho, a.shape[1], C), np.int64)
        for yy in range(Ho):
            y0, n = b[yy]
            t[yy] = (1 << (PRECISION_BITS - 1)) + (a[y0:y0 + n] * k[yy, :n].astype(np.int64)[:, None, None]).sum(0)
        a = np.clip(t >> PRECISION_BITS, 0, 255)
    return a.astype(np.uint8)


def torchvision_resize_size(w: int, h: int, size) -> tuple:
    """torchvision.transforms.Resize output (w, h): an int = the SHORTER side goes to `size`, the other to int(size * long / short) (truncation); a pair =
    exactly that (h, w) (reference imagenet.py:31 train: int ; imagenet.py:44-49 validation: (R, R))"""
    if isinstance(size, (tuple, list)):
        return int(size[1]), int(size[0])
    if w <= h:
        return size, int(size * h / w)
    return int(size * w / h), size
