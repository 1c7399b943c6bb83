"""CPU restatement of the strong augmentation -- TEST INFRASTRUCTURE ONLY.

Reference: ``daod/data/detection_utils.py:7-36`` (``build_strong_augmentation``: RandomApply(ColorJitter(.4, .4,
.4, .1), p=.8), RandomGrayscale(p=.2), RandomApply(GaussianBlur([.1, 2.]), p=.5), ToTensor, 3 x RandomErasing(
value="random"), ToPILImage), ``daod/data/transforms/augmentations.py:6-21`` (``GaussianBlur`` = Pillow's
``ImageFilter.GaussianBlur(radius=sigma)``), applied to the weakly augmented uint8 HWC array in
``daod/data/mappers/two_crop_augmentation_mapper.py:141-146`` (the BGR array is handed to Pillow as "RGB": the
colour formulas run on the channels in stored order, and so do these).

The arithmetic lives in torchvision (un-pinned, not installed) on top of Pillow (installed here: 12.2.0).  The
torchvision layer is thin and restated from its published code (``functional_pil``: ``ImageEnhance`` for
brightness / contrast / saturation, an HSV round trip with a wrapped uint8 shift for hue, ``convert("L")`` for
grayscale; ``RandomErasing.get_params``; ``ToPILImage`` = ``mul(255).byte()``); the Pillow layer is pinned
bit-exactly against Pillow itself in ``tests/test_augment.py`` (RGB<->HSV exhaustively over all 2^24 inputs).
Every function here works on uint8 HWC numpy arrays.
"""
import math

import numpy as np

f32, f64 = np.float32, np.float64

BRIGHTNESS, CONTRAST, SATURATION, HUE, GRAYSCALE = 0, 1, 2, 3, 4


def to_L(a):
    """Pillow ``convert("L")`` (ITU-R 601-2 luma, 16-bit fixed point with rounding)."""
    a = a.astype(np.int64)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg, img, alpha):
    """Pillow ``Image.blend(deg, img, alpha)`` on uint8: float32 arithmetic; truncation for alpha in [0, 1],
    clipping (then truncation) outside."""
    a = f32(alpha)
    t = (deg.astype(f32) + a * (img.astype(f32) - deg.astype(f32))).astype(f32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def adjust_brightness(img, factor):
    """``ImageEnhance.Brightness``: blend with black."""
    return blend(np.zeros_like(img), img, factor)


def contrast_mean(img):
    """``int(ImageStat.Stat(img.convert("L")).mean[0] + 0.5)``."""
    L = to_L(img)
    return int(int(L.astype(np.int64).sum()) / L.size + 0.5)


def adjust_contrast(img, factor):
    """``ImageEnhance.Contrast``: blend with the flat image of the mean luma."""
    return blend(np.full_like(img, contrast_mean(img)), img, factor)


def adjust_saturation(img, factor):
    """``ImageEnhance.Color``: blend with the luma image."""
    return blend(np.repeat(to_L(img)[..., None], 3, -1), img, factor)


def rgb_to_grayscale3(img):
    """torchvision ``rgb_to_grayscale(img, num_output_channels=3)`` on a PIL image."""
    return np.repeat(to_L(img)[..., None], 3, -1)


def rgb2hsv(a):
    """Pillow ``convert("HSV")`` (Convert.c rgb2hsv_row): float variables, double-literal expressions."""
    a = a.astype(np.int32)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(f32)
    safe = np.where(cr == 0, f32(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(f32)
    rc = ((maxc - r).astype(f32) / safe).astype(f64)
    gc = ((maxc - g).astype(f32) / safe).astype(f64)
    bc = ((maxc - b).astype(f32) / safe).astype(f64)
    h = np.where(r == maxc, bc - gc, np.where(g == maxc, 2.0 + rc - bc, 4.0 + gc - rc)).astype(f32)
    h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
    uh = np.clip((h.astype(f64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(f64) * 255.0).astype(np.int32), 0, 255)
    uh = np.where(cr == 0, 0, uh)
    us = np.where(cr == 0, 0, us)
    return np.stack([uh, us, maxc], -1).astype(np.uint8)


def hsv2rgb(a):
    """Pillow HSV -> RGB (Convert.c hsv2rgb_row); C ``round`` = half away from zero."""
    h, s, v = a[..., 0].astype(f32), a[..., 1].astype(f32), a[..., 2].astype(f32)
    hh = h.astype(f64) * 6.0 / 255.0
    i = np.floor(hh).astype(np.int32)
    f = (hh - i).astype(f32)
    fs = (s.astype(f64) / 255.0).astype(f32)
    vd, fsd, fd = v.astype(f64), fs.astype(f64), f.astype(f64)
    p = np.clip(np.floor(vd * (1.0 - fsd) + 0.5), 0, 255).astype(np.uint8)
    q = np.clip(np.floor(vd * (1.0 - fsd * fd) + 0.5), 0, 255).astype(np.uint8)
    t = np.clip(np.floor(vd * (1.0 - fsd * (1.0 - fd)) + 0.5), 0, 255).astype(np.uint8)
    vu = a[..., 2]
    k = i % 6
    r = np.choose(k, [vu, q, p, p, t, vu])
    g = np.choose(k, [t, vu, vu, q, p, p])
    b = np.choose(k, [p, p, t, vu, vu, q])
    z = a[..., 1] == 0
    return np.stack([np.where(z, vu, r), np.where(z, vu, g), np.where(z, vu, b)], -1).astype(np.uint8)


def hue_shift(hue_factor):
    """``np.uint8(hue_factor * 255)``: truncation toward zero, then wrap-around."""
    return int(hue_factor * 255) & 255


def adjust_hue(img, hue_factor):
    """torchvision ``functional_pil.adjust_hue``: HSV round trip with the H channel shifted modulo 256."""
    hsv = rgb2hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift(hue_factor)).astype(np.uint8)
    return hsv2rgb(hsv)


def gaussian_box_radius(sigma, passes=3):
    """Pillow BoxBlur.c ``_gaussian_blur_radius`` (all float32)."""
    f = f32
    radius = f(sigma)
    sigma2 = f(f(radius * radius) / f(passes))
    L = f(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(math.floor((float(L) - 1.0) / 2.0))
    a = f(f(f(2) * l + f(1)) * f(f(l * f(l + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(l + f(1)) * f(l + f(1))))))
    return f(l + a)


def box_weights(fr):
    """``ImagingHorizBoxBlur``: integer radius, 8.24 fixed-point weight of the inner taps (float32 division) and of
    the two fractional outer taps."""
    radius = int(fr)
    ww = int(f32(1 << 24) / (f32(fr) * f32(2) + f32(1)))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw


def box_pass(ch, fr):
    """One extended-box pass along the last axis of a [H, W] uint8 plane; edges repeat the border pixel."""
    radius, ww, fw = box_weights(fr)
    W = ch.shape[1]
    x = np.arange(W)
    a = ch.astype(np.int64)
    acc = np.zeros(ch.shape, np.int64)
    for k in range(-radius, radius + 1):
        acc += a[:, np.clip(x + k, 0, W - 1)]
    far = a[:, np.clip(x - radius - 1, 0, W - 1)] + a[:, np.clip(x + radius + 1, 0, W - 1)]
    return ((acc * ww + far * fw + (1 << 23)) >> 24).astype(np.uint8)


def gaussian_blur(img, sigma):
    """``ImageFilter.GaussianBlur(radius=sigma)``: 3 box passes along x, then 3 along y, per channel."""
    fr = gaussian_box_radius(sigma)
    out = img.copy()
    for c in range(img.shape[2]):
        ch = out[..., c]
        for _ in range(3):
            ch = box_pass(ch, fr)
        ch = np.ascontiguousarray(ch.T)
        for _ in range(3):
            ch = box_pass(ch, fr)
        out[..., c] = ch.T
    return out


def noise_to_u8(v):
    """``ToPILImage`` on the float tensor: ``v.mul(255).byte()`` -- float32 product, truncation toward zero,
    low 8 bits (torch's CPU cast of out-of-range values)."""
    return (np.trunc((np.asarray(v, f32) * f32(255)).astype(f64)).astype(np.int64) & 255).astype(np.uint8)


def erase(img, i, j, h, w, noise_chw):
    """``F.erase(img, i, j, h, w, v)`` between ToTensor and ToPILImage (the /255 * 255 round trip of the other
    pixels is the identity on uint8, checked in tests/test_augment.py)."""
    out = img.copy()
    out[i:i + h, j:j + w, :] = noise_to_u8(noise_chw).transpose(1, 2, 0)
    return out


def erasing_params(img_h, img_w, scale, ratio, draws):
    """``RandomErasing.get_params`` given its uniform draws ``[(u_area, u_logratio, u_i, u_j)] * 10`` in [0, 1):
    -> (i, j, h, w) or None when no attempt fits (the transform is then a no-op)."""
    area = img_h * img_w
    lr0, lr1 = math.log(ratio[0]), math.log(ratio[1])
    for ua, ur, ui, uj in draws:
        erase_area = area * (scale[0] + (scale[1] - scale[0]) * ua)
        aspect = math.exp(lr0 + (lr1 - lr0) * ur)
        h = int(round(math.sqrt(erase_area * aspect)))
        w = int(round(math.sqrt(erase_area / aspect)))
        if not (h < img_h and w < img_w):
            continue
        return int(ui * (img_h - h + 1)), int(uj * (img_w - w + 1)), h, w
    return None


def apply_ops(img, ops):
    """``ops``: [(code, factor)] in application order (ColorJitter's permutation, then grayscale)."""
    for code, factor in ops:
        if code == BRIGHTNESS:
            img = adjust_brightness(img, factor)
        elif code == CONTRAST:
            img = adjust_contrast(img, factor)
        elif code == SATURATION:
            img = adjust_saturation(img, factor)
        elif code == HUE:
            img = adjust_hue(img, factor)
        elif code == GRAYSCALE:
            img = rgb_to_grayscale3(img)
        else:
            raise ValueError(code)
    return img


def strong_augment(img, params):
    """The whole pipeline for given parameters: ``params = {"ops": [...], "sigma": float | None,
    "erase": [(i, j, h, w, noise [C,h,w])]}``."""
    img = apply_ops(img, params.get("ops", []))
    if params.get("sigma") is not None:
        img = gaussian_blur(img, params["sigma"])
    for i, j, h, w, noise in params.get("erase", []):
        img = erase(img, i, j, h, w, noise)
    return img
