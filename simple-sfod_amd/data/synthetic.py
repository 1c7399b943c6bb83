"""Synthetic COCO-format target data + the source-free two-crop loader contract.

The reference's loader (``daod/data/build.py:289-367``, ``daod/data/common.py:199-228``,
``daod/data/mappers/two_crop_augmentation_mapper.py:73-157``) yields, per iteration and per rank,
``(strong_list, weak_list)`` of ``IMS_PER_BATCH_TARGET // world_size`` dicts; each dict carries
``image`` (uint8 ``3xHxW`` BGR tensor, post ResizeShortestEdge + RandomFlip), ``instances``,
``height``, ``width``, ``file_name``, ``image_id``.  This module reproduces that contract on
synthetic 8-class Cityscapes-shape frames (there is no dataset / network in the build
environment); registered COCO-json datasets take their place through ``data/coco.py``, the strong
augmentation is ``data/augment.py``.

Frames are generated once and kept resident on the device at their native size (1024x2048 by
default); every iteration the mapper's ResizeShortestEdge (shortest edge 600, max 1333,
Pillow bilinear on uint8) and RandomFlip run on the device in one launch per frame
(``native.resize_bilinear_u8``, bit-exact with Pillow; SFOD.SYNTHETIC.DEVICE_RESIZE).  On a CPU device
(host-logic tests) or with DEVICE_RESIZE off the frames are resized once with Pillow at start-up.
"""
import numpy as np
import torch

from .. import native

from ..structures import Boxes, Instances

CITYSCAPES_CLASSES = ["person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle"]


def resize_shortest_edge_shape(h, w, short, max_size):
    """d2 ResizeShortestEdge.get_output_shape."""
    scale = short * 1.0 / min(h, w)
    if h < w:
        newh, neww = short, scale * w
    else:
        newh, neww = scale * h, short
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def make_frame(index, height, width, num_boxes, seed=42, num_classes=8):
    """uint8 [3,H,W] ``randint(0,256)`` seed 42+i; boxes w,h in [32,256]."""
    g = torch.Generator().manual_seed(seed + index)
    img = torch.randint(0, 256, (3, height, width), generator=g, dtype=torch.uint8)
    wh = torch.rand(num_boxes, 2, generator=g) * (256 - 32) + 32
    xy = torch.rand(num_boxes, 2, generator=g) * torch.tensor([width, height], dtype=torch.float32)
    boxes = torch.cat([xy, xy + wh], dim=1)
    boxes[:, 0::2].clamp_(0, width)
    boxes[:, 1::2].clamp_(0, height)
    classes = torch.randint(0, num_classes, (num_boxes,), generator=g)
    return img, boxes, classes


def coco_record(index, height, width, boxes, classes):
    """The dataset-dict / COCO annotation schema of cityscapes-to-coco-conversion (main.py:203-209)."""
    return {
        "file_name": f"synthetic/{index:06d}.png", "image_id": index, "height": height, "width": width,
        "annotations": [{"bbox": [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])],
                         "bbox_mode": 1, "category_id": int(c), "iscrowd": 0}
                        for b, c in zip(boxes.tolist(), classes.tolist())],
    }


def _resize_u8(img_chw, newh, neww):
    from PIL import Image
    arr = img_chw.permute(1, 2, 0).numpy()
    out = np.asarray(Image.fromarray(arr).resize((neww, newh), Image.BILINEAR))
    return torch.from_numpy(out.copy()).permute(2, 0, 1).contiguous()


class SyntheticTargetDataset:
    """N frames, mapped once (resize) and kept on `device` as uint8 CHW tensors."""

    def __init__(self, cfg, device, num_images=None, train=True):
        s = cfg.SFOD.SYNTHETIC
        n = num_images or s.NUM_IMAGES
        short = cfg.INPUT.MIN_SIZE_TRAIN[0] if train else cfg.INPUT.MIN_SIZE_TEST
        max_size = cfg.INPUT.MAX_SIZE_TRAIN if train else cfg.INPUT.MAX_SIZE_TEST
        newh, neww = resize_shortest_edge_shape(s.HEIGHT, s.WIDTH, short, max_size)
        resize = (newh, neww) != (s.HEIGHT, s.WIDTH)
        # resize per iteration on the device (the mapper's job), or once here with Pillow
        self.device_resize = bool(resize and s.DEVICE_RESIZE and torch.device(device).type == "cuda")
        self.items = []
        for i in range(n):
            img, boxes, classes = make_frame(i, s.HEIGHT, s.WIDTH, s.BOXES_PER_IMAGE, seed=max(cfg.SEED, 0))
            if resize:
                if not self.device_resize:
                    img = _resize_u8(img, newh, neww)
                boxes = boxes * torch.tensor([neww / s.WIDTH, newh / s.HEIGHT] * 2)
            host = bool(s.get("HOST_FRAMES", False)) and self.device_resize and train      # (the training mapper uploads; section 6)
            self.items.append({"image": img.pin_memory() if host else img.to(device), "boxes": boxes.to(device), "classes": classes.to(device),
                               "height": s.HEIGHT, "width": s.WIDTH, "image_id": i,
                               "file_name": f"synthetic/{i:06d}.png"})
        self.size = (newh, neww)

    def __len__(self):
        return len(self.items)

    def dataset_dicts(self, cfg):
        """d2 dataset records (annotations at the frames' native size) -- the evaluator's ground truth."""
        s = cfg.SFOD.SYNTHETIC
        out = []
        for i in range(len(self.items)):
            _, boxes, classes = make_frame(i, s.HEIGHT, s.WIDTH, s.BOXES_PER_IMAGE, seed=max(cfg.SEED, 0))
            out.append(coco_record(i, s.HEIGHT, s.WIDTH, boxes, classes))
        return out


class InferenceSampler:
    """d2 InferenceSampler: rank r takes one contiguous share of range(size) (shares differ by at most 1)."""

    def __init__(self, size, rank=0, world=1):
        shard = size // world
        left = size % world
        sizes = [shard + int(r < left) for r in range(world)]
        begin = sum(sizes[:rank])
        self.indices = range(begin, min(begin + sizes[rank], size))

    def __iter__(self):
        return iter(self.indices)

    def __len__(self):
        return len(self.indices)


class TestLoader:
    """``build_detection_test_loader(cfg, name, batch_size=TEST.IMS_PER_BATCH, mapper=DatasetMapperAnnotation(
    cfg, is_train=False))`` (``base.py:163-171``): dataset order, ResizeShortestEdge(MIN_SIZE_TEST,
    MAX_SIZE_TEST), no flip; ``height`` / ``width`` are the frame's native size (what ``detector_postprocess``
    rescales the detections to)."""

    def __init__(self, cfg, device, rank=0, world=1, dataset=None, dataset_name=None):
        if dataset is None:
            from .coco import build_dataset
            dataset = build_dataset(cfg, device, [dataset_name] if dataset_name else list(cfg.DATASETS.TEST),
                                    train=False, num_images=cfg.SFOD.SYNTHETIC.NUM_TEST_IMAGES)
        self.dataset = dataset
        self.batch = max(int(cfg.TEST.IMS_PER_BATCH), 1)
        self.sampler = InferenceSampler(len(self.dataset), rank, world)

    def __len__(self):
        return (len(self.sampler) + self.batch - 1) // self.batch

    def _map(self, item):
        img = item["image"]
        newh, neww = item.get("size", self.dataset.size)
        if getattr(self.dataset, "device_resize", False) and (newh, neww) != tuple(img.shape[1:]):
            img = native.resize_bilinear_u8(img, newh, neww, flip=False)
        inst = Instances((int(img.shape[1]), int(img.shape[2])))
        inst.gt_boxes = Boxes(item["boxes"])
        inst.gt_classes = item["classes"]
        return {"image": img, "instances": inst, "height": item["height"], "width": item["width"],
                "image_id": item["image_id"], "file_name": item["file_name"]}

    def __iter__(self):
        batch = []
        for idx in self.sampler:
            batch.append(self._map(self.dataset.items[idx]))
            if len(batch) == self.batch:
                yield batch
                batch = []
        if batch:
            yield batch


class TrainingSampler:
    """d2 TrainingSampler: one shared-seed infinite permutation stream, rank r takes r, r+W, ..."""

    def __init__(self, size, seed, rank=0, world=1, shuffle=True):
        self.size, self.seed, self.rank, self.world, self.shuffle = size, int(seed), rank, world, shuffle

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        i = 0
        while True:
            perm = torch.randperm(self.size, generator=g) if self.shuffle else torch.arange(self.size)
            for idx in perm.tolist():
                if i % self.world == self.rank:
                    yield idx
                i += 1


class TwoCropLoader:
    """``build_detection_semisup_train_loader_two_crops_source_free``: yields (strong, weak) lists.

    With ``WEAK_STRONG_AUGMENT: False`` (hot yaml) the trainer discards the strong copy
    (source_free_adaptive_teacher.py:351-352), so both lists reference the same weakly augmented
    tensors; with it on (the R101 yaml) the strong list carries ``StrongAugmentation`` of the weak frame,
    computed on the device (data/augment.py)."""

    def __init__(self, cfg, device, rank=0, world=1, labeled=False, dataset=None):
        total = cfg.SOLVER.IMS_PER_BATCH if labeled else cfg.SOLVER.IMS_PER_BATCH_TARGET
        # build.py:337-343 (the message is the reference's: tests/golden/glue_ref.npz ``lb_assert_msg``; d2's labelled loader
        # says "Total batch size")
        assert total > 0 and total % world == 0, \
            "Total {}batch size ({}) must be divisible by the number of gpus ({}).".format("" if labeled else "unlabel ", total, world)
        if "ASPECT_RATIO_GROUPING" in cfg.DATALOADER and not cfg.DATALOADER.ASPECT_RATIO_GROUPING and not labeled:
            raise NotImplementedError("ASPECT_RATIO_GROUPING = False is not supported yet")        # build.py:367
        self.batch = total // world
        if dataset is None:
            from .coco import build_dataset
            names = cfg.DATASETS.TRAIN if labeled else (cfg.DATASETS.TRAIN_TARGET if "TRAIN_TARGET" in cfg.DATASETS
                                                        else cfg.DATASETS.TRAIN)
            dataset = build_dataset(cfg, device, list(names), train=True)
        self.dataset = dataset
        self.sampler = iter(TrainingSampler(len(self.dataset), max(cfg.SEED, 0), rank, world))
        self._buckets = [[], []]       # aspect-ratio grouping (daod/data/common.py:199-228): w > h | otherwise
        self.flip = cfg.INPUT.RANDOM_FLIP == "horizontal"
        self.gen = torch.Generator().manual_seed(max(cfg.SEED, 0) + rank)
        self.labeled = labeled
        # the strong copy of the two-crop mapper (two_crop_augmentation_mapper.py:141-146) when the config uses it
        self.strong_aug = None
        if not labeled and "WEAK_STRONG_AUGMENT" in cfg and cfg.WEAK_STRONG_AUGMENT and \
                torch.device(device).type == "cuda" and cfg.SFOD.SYNTHETIC.STRONG_AUGMENT:
            from .augment import StrongAugmentation
            self.strong_aug = StrongAugmentation(torch.Generator().manual_seed(max(cfg.SEED, 0) + 7919 * (rank + 1)))

    def _map(self, item):
        img, boxes = item["image"], item["boxes"]
        do_flip = bool(self.flip and torch.rand(1, generator=self.gen).item() < 0.5)
        newh, neww = item.get("size", self.dataset.size)
        if getattr(self.dataset, "device_resize", False) and (newh, neww) != tuple(img.shape[1:]):
            if not img.is_cuda:        # SFOD.SYNTHETIC.HOST_FRAMES: the frame comes from pinned host memory, on the loader's stream
                img = img.to(boxes.device, non_blocking=True)
            img = native.resize_bilinear_u8(img, newh, neww, flip=do_flip)     # resize (+ flip) in one launch
        elif do_flip:
            img = native.hflip_u8(img) if img.is_cuda else torch.flip(img, dims=[2])
        if do_flip:
            w = img.shape[2]
            boxes = torch.stack([w - boxes[:, 2], boxes[:, 1], w - boxes[:, 0], boxes[:, 3]], dim=1)
        inst = Instances((int(img.shape[1]), int(img.shape[2])))
        inst.gt_boxes = Boxes(boxes)
        inst.gt_classes = item["classes"]
        return {"image": img, "instances": inst, "height": item["height"], "width": item["width"],
                "image_id": item["image_id"], "file_name": item["file_name"]}

    def __iter__(self):
        return self

    def _produce(self):
        # a batch holds images of ONE aspect class; the other bucket keeps what it has for a later batch
        while True:
            d = self._map(self.dataset.items[next(self.sampler)])
            bucket = self._buckets[0 if d["width"] > d["height"] else 1]
            bucket.append(d)
            if len(bucket) == self.batch:
                weak = bucket[:]
                del bucket[:]
                break
        if self.labeled:
            return weak
        strong = [dict(d) for d in weak]
        if self.strong_aug is not None:
            for d in strong:
                d["image"] = self.strong_aug(d["image"])
        return strong, weak

    def __next__(self):
        """One batch.  On a GPU the mapper's device work (resize + flip launches) for batch t+1 is enqueued on
        a separate low-priority stream while step t runs -- what the reference's loader worker processes do
        on the host -- and the consumer's stream waits for its event."""
        if not getattr(self.dataset, "device_resize", False):
            return self._produce()
        main = torch.cuda.current_stream()
        st = self.__dict__.get("_stream")
        if st is None:
            st = self._stream = torch.cuda.Stream(priority=0)
            self._ahead = None
        if self._ahead is None:
            self._ahead = self._enqueue(st)
        batch, ev = self._ahead
        main.wait_event(ev)
        for lst in (batch if isinstance(batch, tuple) else (batch,)):
            for d in lst:
                d["image"].record_stream(main)     # allocated on the loader stream, consumed elsewhere
        self._ahead = self._enqueue(st)            # next batch: overlaps the step that consumes this one
        return batch

    def _enqueue(self, st):
        with torch.cuda.stream(st):
            batch = self._produce()
            ev = torch.cuda.Event()
            ev.record(st)
        return batch, ev


class FourWayLoader:
    """``build_detection_semisup_train_loader_two_crops`` (daod/data/build.py:145-213) behind
    ``AdaptiveTeacherTrainer.build_train_loader`` (daod/engine/trainers/adaptive_teacher.py:78-81): yields
    ``(label_strong, label_weak, unlabel_strong, unlabel_weak)`` -- labelled source frames from ``DATASETS.TRAIN``
    (``SOLVER.IMS_PER_BATCH`` per step) beside unlabelled target frames from ``DATASETS.TRAIN_TARGET``
    (``SOLVER.IMS_PER_BATCH_TARGET``), each as the two crops of ``DatasetMapperTwoCropSeparate``.

    Batch formation is ``AspectRatioGroupedSemiSupDatasetTwoCrop.__iter__`` (daod/data/common.py:119-160) statement for
    statement, including what it does when one side's batch is complete and the other's is not: BOTH streams advance every
    iteration, and the element drawn for the side whose current bucket is already full is dropped."""

    def __init__(self, cfg, device, rank=0, world=1, label_dataset=None, unlabel_dataset=None):
        nl, nu = cfg.SOLVER.IMS_PER_BATCH, cfg.SOLVER.IMS_PER_BATCH_TARGET
        # build.py:229-239 (the second message reports the LABEL batch size: the reference's own slip, kept)
        assert nl > 0 and nl % world == 0, \
            "Total label batch size ({}) must be divisible by the number of gpus ({}).".format(nl, world)
        assert nu > 0 and nu % world == 0, \
            "Total unlabel batch size ({}) must be divisible by the number of gpus ({}).".format(nl, world)
        self.lab = TwoCropLoader(cfg, device, rank, world, labeled=True, dataset=label_dataset)
        self.unl = TwoCropLoader(cfg, device, rank, world, labeled=False, dataset=unlabel_dataset)
        self.batch_size_label, self.batch_size_unlabel = nl // world, nu // world
        # the labelled frames' strong crop (two_crop_augmentation_mapper.py:141-146: both datasets go through the same mapper)
        self.strong_aug = None
        if torch.device(device).type == "cuda" and cfg.SFOD.SYNTHETIC.STRONG_AUGMENT:
            from .augment import StrongAugmentation
            self.strong_aug = StrongAugmentation(torch.Generator().manual_seed(max(cfg.SEED, 0) + 104729 * (rank + 1)))
        self._label_buckets, self._label_buckets_key = [[], []], [[], []]
        self._unlabel_buckets, self._unlabel_buckets_key = [[], []], [[], []]
        self._label_bucket, self._unlabel_bucket = [], []          # common.py:120: the buckets last appended to
        self._label_key, self._unlabel_key = [], []

    def _two_crops(self, src):
        weak = src._map(src.dataset.items[next(src.sampler)])
        strong = dict(weak)
        if self.strong_aug is not None:
            strong["image"] = self.strong_aug(weak["image"])
        return strong, weak                                        # d[0] strong, d[1] weak (common.py:123-124)

    def __iter__(self):
        return self

    def __next__(self):
        while True:
            d_label, d_unlabel = self._two_crops(self.lab), self._two_crops(self.unl)
            if len(self._label_bucket) != self.batch_size_label:
                i = 0 if d_label[0]["width"] > d_label[0]["height"] else 1
                self._label_bucket, self._label_key = self._label_buckets[i], self._label_buckets_key[i]
                self._label_bucket.append(d_label[0])
                self._label_key.append(d_label[1])
            if len(self._unlabel_bucket) != self.batch_size_unlabel:
                i = 0 if d_unlabel[0]["width"] > d_unlabel[0]["height"] else 1
                self._unlabel_bucket, self._unlabel_key = self._unlabel_buckets[i], self._unlabel_buckets_key[i]
                self._unlabel_bucket.append(d_unlabel[0])
                self._unlabel_key.append(d_unlabel[1])
            if len(self._label_bucket) == self.batch_size_label and len(self._unlabel_bucket) == self.batch_size_unlabel:
                out = (self._label_bucket[:], self._label_key[:], self._unlabel_bucket[:], self._unlabel_key[:])
                del self._label_bucket[:], self._label_key[:], self._unlabel_bucket[:], self._unlabel_key[:]
                return out
