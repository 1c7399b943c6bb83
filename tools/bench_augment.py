"""Strong-augmentation throughput on one MI355X: per-kernel time on a 3x600x1200 uint8 frame and
frames/s of the sampled pipeline (the mix of transforms build_strong_augmentation draws).  One JSON line."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3          # us


def main():
    sfod = importlib.import_module("simple-sfod_amd")
    nat = sfod.native
    dev = torch.device("cuda:0")
    H, W = 600, 1200
    img = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, device=dev)
    nbytes = 3 * H * W
    res = {}
    res["jitter4_us"] = timed(lambda: nat.aug_color(img, [(0, 1.1), (1, 0.8), (2, 1.3), (3, 0.05)]))
    res["jitter_no_contrast_us"] = timed(lambda: nat.aug_color(img, [(0, 1.1), (2, 1.3), (3, 0.05)]))
    res["grayscale_us"] = timed(lambda: nat.aug_color(img, [(4, 0.0)]))
    res["blur_us"] = timed(lambda: nat.aug_gaussian_blur(img, 1.5))
    noise = torch.randn(3, 250, 400, device=dev)
    tmp = img.clone()
    res["erase_250x400_us"] = timed(lambda: nat.aug_erase_(tmp, 10, 20, 250, 400, noise))
    aug = sfod.data.StrongAugmentation(torch.Generator().manual_seed(0))
    params = [aug.sample(H, W) for _ in range(64)]
    it = iter(range(10 ** 9))
    us = timed(lambda: aug.apply(img, params[next(it) % 64]), n=256)
    res = {k: round(v, 1) for k, v in res.items()}
    print(json.dumps({"metric": "strong augmentation frames/s (sampled transform mix, 3x600x1200 uint8)",
                      "value": round(1e6 / us, 1), "unit": "frames/s", "us_per_frame": round(us, 1), "n_gpus": 1,
                      "kernels": res,
                      "hbm_GBps_jitter4": round(2 * nbytes * (1 + 0.5) / res["jitter4_us"] / 1e3, 1),
                      "hbm_GBps_blur": round(2 * nbytes * 6 / res["blur_us"] / 1e3, 1), "data": "synthetic"}))


if __name__ == "__main__":
    main()
