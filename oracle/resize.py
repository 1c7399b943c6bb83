"""CPU restatement of the mapper's ResizeShortestEdge for uint8 frames.  TEST INFRASTRUCTURE ONLY.

Detectron2's ``ResizeTransform.apply_image`` (un-vendored) resizes uint8 images
with ``PIL.Image.resize((w, h), BILINEAR)``.  The arithmetic therefore lives in Pillow (third party,
12.2.0 in this image) -- ``src/libImaging/Resample.c``: ``precompute_coeffs`` (support-scaled triangle
filter, coefficients normalised in float64), ``normalize_coeffs_8bpc`` (22-bit fixed point, round half
away from zero), ``ImagingResampleHorizontal_8bpc`` then ``ImagingResampleVertical_8bpc`` with a uint8
clip after EACH pass.  Pinned: ``tests/test_oracle_ops.py::test_resize_restatement_equals_pillow`` runs
Pillow itself on random frames (down- and up-scaling, odd sizes) and requires bit equality.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for BILINEAR over the whole axis."""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            v = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(v - 0.5) if k[x] < 0 else int(v + 0.5)
    return bounds, kk


def resize_bilinear_u8(img_chw, newh, neww):
    """uint8 [C,H,W] -> [C,newh,neww]: horizontal pass, clip, vertical pass, clip."""
    img = np.asarray(img_chw)
    C, H, W = img.shape
    hb, hk = precompute_coeffs(W, neww)
    vb, vk = precompute_coeffs(H, newh)
    src = img.astype(np.int64)
    tmp = np.zeros((C, H, neww), dtype=np.int64)
    for x in range(neww):
        x0, n = hb[x]
        acc = (1 << (PRECISION_BITS - 1)) + (src[:, :, x0:x0 + n] * hk[x, :n]).sum(-1)
        tmp[:, :, x] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.zeros((C, newh, neww), dtype=np.int64)
    for y in range(newh):
        y0, n = vb[y]
        acc = (1 << (PRECISION_BITS - 1)) + (tmp[:, y0:y0 + n, :] * vk[y, :n][None, :, None]).sum(1)
        out[:, y, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out.astype(np.uint8)
