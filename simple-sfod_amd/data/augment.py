"""Strong augmentation of the two-crop mapper on the device.

``build_strong_augmentation(cfg, is_train)`` (``daod/data/detection_utils.py:7-36``), applied to the weakly
augmented frame by ``DatasetMapperTwoCropSeparate.__call__``
(``daod/data/mappers/two_crop_augmentation_mapper.py:141-146``):

    RandomApply([ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8) -> RandomGrayscale(p=0.2) ->
    RandomApply([GaussianBlur([0.1, 2.0])], p=0.5) -> ToTensor -> RandomErasing(p=0.7, scale=(0.05, 0.2),
    ratio=(0.3, 3.3), "random") -> RandomErasing(p=0.5, (0.02, 0.2), (0.1, 6)) -> RandomErasing(p=0.3,
    (0.02, 0.2), (0.05, 8)) -> ToPILImage

The parameter draws follow torchvision's ``get_params`` logic (which values are drawn, their ranges, the 10
attempts of RandomErasing) on this loader's own generator -- the reference's stream of random numbers (torch's
global CPU generator inside worker processes, Python's ``random`` for sigma) is not reproducible anyway; the
pixel arithmetic for given parameters is bit-exact (``native.aug_*``, pinned to Pillow through the oracle).
"""
import math

import torch

from .. import native

ERASINGS = ((0.7, (0.05, 0.2), (0.3, 3.3)), (0.5, (0.02, 0.2), (0.1, 6.0)), (0.3, (0.02, 0.2), (0.05, 8.0)))


def erasing_params(img_h, img_w, scale, ratio, draws):
    """torchvision ``RandomErasing.get_params`` for given uniform draws [(u_area, u_logratio, u_i, u_j)] * 10:
    -> (i, j, h, w), or None when none of the attempts fits (the transform then returns the image unchanged)."""
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


class StrongAugmentation:
    """Callable on uint8 [3,H,W] device frames; ``sample`` / ``apply`` are split so that tests can replay the
    parameters through the oracle."""

    def __init__(self, generator=None, brightness=0.4, contrast=0.4, saturation=0.4, hue=0.1,
                 p_jitter=0.8, p_gray=0.2, p_blur=0.5, sigma=(0.1, 2.0)):
        self.gen = generator or torch.Generator().manual_seed(0)
        self.b, self.c, self.s, self.h = brightness, contrast, saturation, hue
        self.p_jitter, self.p_gray, self.p_blur, self.sigma = p_jitter, p_gray, p_blur, sigma

    def _u(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * torch.rand(1, generator=self.gen).item()

    def sample(self, height, width):
        ops = []
        if not (self.p_jitter < self._u()):                    # RandomApply: ``if self.p < torch.rand(1): return img``
            order = torch.randperm(4, generator=self.gen).tolist()      # ColorJitter.get_params
            f = {native.AUG_BRIGHTNESS: self._u(max(0.0, 1 - self.b), 1 + self.b),
                 native.AUG_CONTRAST: self._u(max(0.0, 1 - self.c), 1 + self.c),
                 native.AUG_SATURATION: self._u(max(0.0, 1 - self.s), 1 + self.s),
                 native.AUG_HUE: self._u(-self.h, self.h)}
            ops = [(k, f[k]) for k in order]
        if self._u() < self.p_gray:                            # RandomGrayscale
            ops.append((native.AUG_GRAYSCALE, 0.0))
        sigma = None
        if not (self.p_blur < self._u()):
            sigma = self._u(self.sigma[0], self.sigma[1])      # random.uniform(0.1, 2.0)
        erase = []
        for p, scale, ratio in ERASINGS:                       # RandomErasing.forward: ``if torch.rand(1) < self.p``
            if self._u() < p:
                draws = [tuple(self._u() for _ in range(4)) for _ in range(10)]
                rect = erasing_params(height, width, scale, ratio, draws)
                if rect is not None:
                    erase.append(rect)
        return {"ops": ops, "sigma": sigma, "erase": erase}

    @staticmethod
    def apply(img, params, noises=None):
        """``noises``: optional list of [3,h,w] fp32 device tensors (one per rectangle); drawn with
        ``torch.randn`` on the frame's device otherwise (``torch.empty(...).normal_()`` in the reference)."""
        out = native.aug_color(img, params["ops"]) if params["ops"] else img
        if params["sigma"] is not None:
            out = native.aug_gaussian_blur(out, params["sigma"])
        if params["erase"]:
            if out is img:
                out = img.clone()
            for k, (i, j, h, w) in enumerate(params["erase"]):
                noise = noises[k] if noises is not None else torch.randn(3, h, w, device=img.device)
                native.aug_erase_(out, i, j, h, w, noise)
        return out

    def __call__(self, img):
        return self.apply(img, self.sample(int(img.shape[1]), int(img.shape[2])))
