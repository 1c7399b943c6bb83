"""COCO-json datasets behind the loader contract.

The reference registers its datasets in code (``register_all_datasets(cfg)``, ``train_net_mt.py:71``; the
Cityscapes-to-COCO converter writes the json files) and reads them with Detectron2's
``load_coco_json`` / ``utils.read_image(file_name, format="BGR")``.  This module is the same thing without
Detectron2 / pycocotools: ``register_coco_instances(name, json_file, image_root)``, ``load_coco_json`` with d2's
conventions (category ids sorted and mapped to contiguous 0..K-1, ``bbox`` kept XYWH_ABS in the records,
``iscrowd`` carried, images without annotations kept), and ``CocoTargetDataset``: frames decoded once with Pillow
(by a pool of ``DATALOADER.NUM_WORKERS`` threads working ahead of the ordered consumer), kept resident on the device as uint8 CHW BGR tensors at their native size (288 GB of HBM hold the Cityscapes
training set ~15 times), resized / flipped / augmented per iteration by the same device mapper as the
synthetic set.  Frames of different sizes are handled per item; the loaders group batches by aspect ratio like
``AspectRatioGroupedSemiSupDatasetTwoCropSourceFree`` (``daod/data/common.py:199-228``).
"""
import json
import os

import numpy as np
import torch

from .synthetic import _resize_u8, resize_shortest_edge_shape

DATASETS = {}      # name -> (json_file, image_root)


def register_coco_instances(name, json_file, image_root):
    """d2 ``register_coco_instances`` (metadata-free): make ``name`` resolvable from ``DATASETS.*``."""
    DATASETS[name] = (json_file, image_root)


def register_from_file(path):
    """``{"name": {"json_file": ..., "image_root": ...}, ...}`` -> registrations (``SFOD.DATASETS_FILE``)."""
    with open(path) as f:
        for name, d in json.load(f).items():
            register_coco_instances(name, d["json_file"], d["image_root"])


def register_datasets(datasets, root=None):
    """``daod/data/datasets.py:41-108`` for its COCO-format families: the reference's dataset NAMES resolve to the
    same files under ``$DETECTRON2_DATASETS`` (foggy Cityscapes ``cityscapes_instancesonly_foggy_<split>_<fog>``,
    Cityscapes ``cityscapes_instancesonly_<split>``, ``sim10k_<split>``, ``kitti_<split>``).  A name whose json file
    is absent stays unregistered (-> synthetic stand-in); the Pascal-VOC style sets (clipart / comic / watercolor)
    are not supported."""
    import re
    root = root or os.getenv("DETECTRON2_DATASETS", "/scratch/username/datasets")
    for name in datasets:
        if name in DATASETS:
            continue
        if name.startswith("cityscapes_instancesonly_foggy_"):
            m = re.match(r"cityscapes_instancesonly_foggy_(.*)_(.*)", name)
            if m is None:
                raise ValueError(f"Error parsing dataset {name}.")
            split, fog = m.groups()
            base = os.path.join(root, "cityscapes_foggy")
            jf = os.path.join(base, "annotations", f"instancesonly_filtered_gtFine_{split}_{fog}.json")
        elif name.startswith("cityscapes_instancesonly"):
            split = re.match(r"cityscapes_instancesonly_(.*)", name).groups()[0]
            base = os.path.join(root, "cityscapes")
            jf = os.path.join(base, "annotations", f"instancesonly_filtered_gtFine_{split}.json")
        elif name.startswith("sim10k") or name.startswith("kitti"):
            m = re.match(r"(.*)_(.*)", name)
            if m is None:
                raise ValueError(f"Error parsing dataset {name}.")
            ds, split = m.groups()
            base = os.path.join(root, ds)
            jf = os.path.join(base, f"sim10k_coco_{split}.json" if ds == "sim10k" else f"kitti_{split}_coco_format.json")
        else:
            continue
        if os.path.isfile(jf):
            register_coco_instances(name, jf, base)


def register_all_datasets(cfg):
    """``register_all_datasets(cfg)`` (``daod/data/datasets.py:17-23``, called at ``train_net_mt.py:71``)."""
    for d in (cfg.DATASETS.TRAIN, cfg.DATASETS.TEST, cfg.DATASETS.get("TRAIN_TARGET", ())):
        register_datasets(d)


def load_coco_json(json_file, image_root):
    """d2 ``load_coco_json`` for boxes -> (dataset dicts, class names).  Category ids are sorted and mapped to
    0..K-1; every annotation keeps ``bbox`` (XYWH_ABS, ``bbox_mode`` 1), ``category_id`` (contiguous) and
    ``iscrowd``; an annotation with ``ignore`` != 0 is rejected like in d2."""
    with open(json_file) as f:
        data = json.load(f)
    cats = sorted(data["categories"], key=lambda c: c["id"])
    id_map = {c["id"]: i for i, c in enumerate(cats)}
    names = [c["name"] for c in cats]
    anns = {}
    seen = set()
    for a in data.get("annotations", []):
        assert a["id"] not in seen, "Annotation ids in '{}' are not unique!".format(json_file)
        seen.add(a["id"])
        assert a.get("ignore", 0) == 0, '"ignore" in COCO json file is not supported.'
        anns.setdefault(a["image_id"], []).append(a)
    dicts = []
    for img in sorted(data["images"], key=lambda i: i["id"]):
        rec = {"file_name": os.path.join(image_root, img["file_name"]), "height": img["height"],
               "width": img["width"], "image_id": img["id"], "annotations": []}
        for a in anns.get(img["id"], []):
            obj = {"bbox": [float(v) for v in a["bbox"]], "bbox_mode": 1, "category_id": id_map[a["category_id"]],
                   "iscrowd": int(a.get("iscrowd", 0))}
            if "area" in a:
                obj["area"] = float(a["area"])
            rec["annotations"].append(obj)
        dicts.append(rec)
    return dicts, names


def read_image_bgr(path):
    """d2 ``utils.read_image(path, format="BGR")``: EXIF-oriented RGB decode, channels reversed.  -> uint8 HWC."""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        im = ImageOps.exif_transpose(im).convert("RGB")
        return np.asarray(im)[:, :, ::-1].copy()


class CocoTargetDataset:
    """Same item layout as ``SyntheticTargetDataset`` (image / boxes at the mapper's output scale / classes /
    native height, width / ids) plus a per-item output ``size``; ``dataset_dicts`` are the loaded records."""

    def __init__(self, cfg, device, names, train=True):
        short = cfg.INPUT.MIN_SIZE_TRAIN[0] if train else cfg.INPUT.MIN_SIZE_TEST
        max_size = cfg.INPUT.MAX_SIZE_TRAIN if train else cfg.INPUT.MAX_SIZE_TEST
        self.device_resize = bool(cfg.SFOD.SYNTHETIC.DEVICE_RESIZE and torch.device(device).type == "cuda")
        self.items, self._dicts, self.class_names = [], [], None
        for name in names:
            dicts, cls = load_coco_json(*DATASETS[name])
            assert self.class_names in (None, cls), "datasets of one loader must share their categories"
            self.class_names = cls
            self._dicts += dicts
        # Decode off the constructing thread: the reference decodes in DATALOADER.NUM_WORKERS loader processes
        # (daod/data/build.py:289-367 -> torch DataLoader; two_crop_augmentation_mapper.py:73-157 reads the file there).  Here
        # a frame is decoded ONCE and stays resident, so the pool works ahead of the (ordered) consumer below: Pillow
        # releases the GIL inside its decoders, a few threads keep as many cores busy, at most 2 x workers decoded frames
        # wait in host memory, and on a CUDA device each goes through a pinned staging buffer so that the upload is
        # asynchronous to the next decode.  NUM_WORKERS 0: decode inline (the order and every value are the same).
        workers = int(cfg.DATALOADER.NUM_WORKERS) if "DATALOADER" in cfg else 0
        on_cuda = torch.device(device).type == "cuda"

        def decode(rec):
            img = torch.from_numpy(read_image_bgr(rec["file_name"])).permute(2, 0, 1).contiguous()
            return img.pin_memory() if on_cuda else img

        def decoded_in_order():
            if workers <= 0 or len(self._dicts) < 2:
                for rec in self._dicts:
                    yield rec, decode(rec)
                return
            from collections import deque
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="sfod-decode") as pool:
                pending, it = deque(), iter(self._dicts)
                for rec in it:
                    pending.append((rec, pool.submit(decode, rec)))
                    if len(pending) >= 2 * workers:
                        break
                while pending:
                    rec, fut = pending.popleft()
                    nxt = next(it, None)
                    if nxt is not None:
                        pending.append((nxt, pool.submit(decode, nxt)))
                    yield rec, fut.result()

        for rec, img in decoded_in_order():
            h, w = int(img.shape[1]), int(img.shape[2])
            newh, neww = resize_shortest_edge_shape(h, w, short, max_size)
            if (newh, neww) != (h, w) and not self.device_resize:
                img = _resize_u8(img, newh, neww)
            # the train mapper drops crowd annotations (two_crop_augmentation_mapper.py:124-131)
            keep = [a for a in rec["annotations"] if a.get("iscrowd", 0) == 0]
            b = torch.tensor([a["bbox"] for a in keep], dtype=torch.float32).reshape(-1, 4)
            boxes = torch.cat([b[:, :2], b[:, :2] + b[:, 2:]], 1) * torch.tensor([neww / w, newh / h] * 2)
            boxes[:, 0::2].clamp_(0, neww)          # transform_instance_annotations clips to the image
            boxes[:, 1::2].clamp_(0, newh)
            classes = torch.tensor([a["category_id"] for a in keep], dtype=torch.int64)
            nonempty = ((boxes[:, 2] - boxes[:, 0]) > 1e-5) & ((boxes[:, 3] - boxes[:, 1]) > 1e-5)   # filter_empty_instances
            self.items.append({"image": img.to(device, non_blocking=on_cuda), "boxes": boxes[nonempty].to(device),
                               "classes": classes[nonempty].to(device), "height": h, "width": w,
                               "image_id": rec["image_id"], "file_name": rec["file_name"], "size": (newh, neww)})
        if on_cuda:
            torch.cuda.synchronize(device)          # the pinned staging buffers may go once every upload has finished
        self.size = self.items[0]["size"] if self.items else (0, 0)

    def __len__(self):
        return len(self.items)

    def dataset_dicts(self, cfg=None):
        return self._dicts


def build_dataset(cfg, device, names, train=True, num_images=None):
    """The registered COCO-json datasets named in the config, else the synthetic stand-in."""
    from .synthetic import SyntheticTargetDataset
    if "DATASETS_FILE" in cfg.SFOD and cfg.SFOD.DATASETS_FILE:
        register_from_file(cfg.SFOD.DATASETS_FILE)
    register_datasets(names)
    real = [n for n in names if n in DATASETS]
    if real:
        return CocoTargetDataset(cfg, device, real, train=train)
    return SyntheticTargetDataset(cfg, device, num_images=num_images, train=train)
