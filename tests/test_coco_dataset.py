"""COCO-json datasets behind the loader contract: d2 ``load_coco_json`` conventions, BGR decode,
per-image ResizeShortestEdge, crowd / empty-box filtering of the train mapper, aspect-ratio grouped batches
(daod/data/common.py:199-228), and the evaluator's ground truth from the same records."""
import json
import os

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = os.path.join(ROOT, "configs", "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml")


def _make_dataset(tmp_path, sizes):
    rng = np.random.default_rng(0)
    images, anns = [], []
    aid = 1
    for i, (h, w) in enumerate(sizes):
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(arr, "RGB").save(tmp_path / f"img{i}.png")
        images.append({"id": 100 + i, "file_name": f"img{i}.png", "height": h, "width": w})
        anns.append({"id": aid, "image_id": 100 + i, "category_id": 24, "bbox": [2, 3, w / 2, h / 2], "iscrowd": 0,
                     "area": w * h / 4}); aid += 1
        anns.append({"id": aid, "image_id": 100 + i, "category_id": 7, "bbox": [1, 1, 5, 4], "iscrowd": 1}); aid += 1
    anns.append({"id": aid, "image_id": 100, "category_id": 33, "bbox": [sizes[0][1], 0, 30, 10], "iscrowd": 0})  # clips to nothing
    cats = [{"id": 33, "name": "bicycle"}, {"id": 7, "name": "crowdish"}, {"id": 24, "name": "person"}]
    jf = tmp_path / "ann.json"
    jf.write_text(json.dumps({"images": images, "annotations": anns, "categories": cats}))
    return str(jf), arr


def test_coco_json_dataset_loader_and_evaluator(sfod, tmp_path):
    D = sfod.data
    sizes = [(40, 80), (60, 30), (48, 96), (90, 50), (32, 64), (64, 32)]
    jf, last = _make_dataset(tmp_path, sizes)
    D.register_coco_instances("tiny_train", jf, str(tmp_path))
    dicts, names = D.load_coco_json(jf, str(tmp_path))
    assert names == ["crowdish", "person", "bicycle"]                     # sorted by category id: 7, 24, 33
    assert [d["image_id"] for d in dicts] == [100, 101, 102, 103, 104, 105]
    assert dicts[0]["annotations"][0]["category_id"] == 1 and dicts[0]["annotations"][1]["iscrowd"] == 1
    cfg = sfod.config.setup_cfg(HOT, ["MODEL.DEVICE", "cpu", "DATASETS.TRAIN_TARGET", "('tiny_train',)",
                                      "DATASETS.TEST", "('tiny_train',)", "INPUT.MIN_SIZE_TRAIN", "(32,)",
                                      "INPUT.MAX_SIZE_TRAIN", "64", "INPUT.MIN_SIZE_TEST", "32", "INPUT.MAX_SIZE_TEST", "64",
                                      "SOLVER.IMS_PER_BATCH_TARGET", "2", "TEST.IMS_PER_BATCH", "4",
                                      "INPUT.RANDOM_FLIP", "none", "MODEL.ROI_HEADS.NUM_CLASSES", "3"])
    loader = D.TwoCropLoader(cfg, torch.device("cpu"))
    ds = loader.dataset
    assert isinstance(ds, D.CocoTargetDataset) and len(ds) == 6
    # BGR decode of the last frame, ResizeShortestEdge per image (short 32, max 64)
    it = ds.items[5]
    ref = np.asarray(Image.fromarray(last, "RGB").resize((32, 64), Image.BILINEAR))[:, :, ::-1]
    assert it["size"] == (64, 32) and np.array_equal(it["image"].permute(1, 2, 0).numpy(), ref)
    assert ds.items[0]["size"] == (32, 64) and ds.items[3]["size"] == (58, 32)   # 90x50 -> 57.6 x 32 -> (58, 32)
    # crowd annotation dropped, the box clipped to (almost) nothing dropped, the real one scaled
    assert ds.items[0]["classes"].tolist() == [1]
    torch.testing.assert_close(ds.items[0]["boxes"][0], torch.tensor([2 * 0.8, 3 * 0.8, (2 + 40) * 0.8, (3 + 20) * 0.8]))
    # batches never mix landscape and portrait frames; every frame shows up
    seen = set()
    for _ in range(12):
        strong, weak = next(loader)
        kinds = {d["width"] > d["height"] for d in weak}
        assert len(weak) == 2 and len(kinds) == 1
        seen |= {d["image_id"] for d in weak}
        for d in weak:
            assert d["image"].shape[1:] == tuple(ds.items[d["image_id"] - 100]["size"])
    assert seen == {100, 101, 102, 103, 104, 105}
    # evaluation on the same records: ground truth as detections -> AP 100 for the classes that have non-crowd boxes
    T = sfod.engine.BaseTrainer
    tl = T.build_test_loader(cfg, "tiny_train")
    ev = T.build_evaluator(cfg, "tiny_train", data_loader=tl)
    assert ev.class_names == names
    S = sfod.structures
    for batch in tl:
        outs = []
        for d in batch:
            inst = S.Instances((d["height"], d["width"]))
            sx = d["width"] / d["image"].shape[2]
            inst.pred_boxes = S.Boxes(d["instances"].gt_boxes.tensor * sx)
            inst.scores = torch.full((len(d["instances"]),), 0.9)
            inst.pred_classes = d["instances"].gt_classes
            outs.append({"instances": inst})
        ev.process(batch, outs)
    r = ev.evaluate()["bbox"]
    assert abs(r["AP-person"] - 100.0) < 1e-6 and abs(r["AP50"] - 50.0) < 1.0    # "bicycle": its only box was filtered
    # unregistered names fall back to the synthetic set
    cfg2 = sfod.config.setup_cfg(HOT, ["MODEL.DEVICE", "cpu", "SFOD.SYNTHETIC.HEIGHT", "32", "SFOD.SYNTHETIC.WIDTH", "64",
                                       "SFOD.SYNTHETIC.NUM_IMAGES", "2", "INPUT.MIN_SIZE_TRAIN", "(32,)",
                                       "SOLVER.IMS_PER_BATCH_TARGET", "1"])
    assert isinstance(D.TwoCropLoader(cfg2, torch.device("cpu")).dataset, D.SyntheticTargetDataset)


def test_reference_dataset_names_resolve_under_the_datasets_root(sfod, tmp_path, monkeypatch):
    """daod/data/datasets.py:41-108: name -> files under $DETECTRON2_DATASETS; absent files leave the name unregistered."""
    D = sfod.data
    root = tmp_path / "ds"
    (root / "cityscapes_foggy" / "annotations").mkdir(parents=True)
    (root / "kitti").mkdir()
    jf, _ = _make_dataset(root / "cityscapes_foggy", [(20, 40), (30, 20)])
    os.replace(jf, root / "cityscapes_foggy" / "annotations" / "instancesonly_filtered_gtFine_train_foggy_beta_0.02.json")
    (root / "kitti" / "kitti_train_coco_format.json").write_text(json.dumps({"images": [], "annotations": [], "categories": []}))
    monkeypatch.setenv("DETECTRON2_DATASETS", str(root))
    names = ["cityscapes_instancesonly_foggy_train_foggy_beta_0.02", "cityscapes_instancesonly_val", "kitti_train",
             "sim10k_train", "something_else"]
    D.register_datasets(names)
    reg = D.coco.DATASETS
    assert reg[names[0]] == (str(root / "cityscapes_foggy" / "annotations" / "instancesonly_filtered_gtFine_train_foggy_beta_0.02.json"),
                             str(root / "cityscapes_foggy"))
    assert reg["kitti_train"][0].endswith("kitti/kitti_train_coco_format.json")
    assert "cityscapes_instancesonly_val" not in reg and "sim10k_train" not in reg and "something_else" not in reg
    cfg = sfod.config.setup_cfg(HOT, ["MODEL.DEVICE", "cpu", "DATASETS.TRAIN_TARGET", "('%s',)" % names[0],
                                      "INPUT.MIN_SIZE_TRAIN", "(16,)", "SOLVER.IMS_PER_BATCH_TARGET", "1"])
    D.register_all_datasets(cfg)
    ds = D.TwoCropLoader(cfg, torch.device("cpu")).dataset
    assert isinstance(ds, D.CocoTargetDataset) and len(ds) == 2


def test_decode_pool_keeps_order_and_values(sfod, tmp_path):
    """The frames are decoded by DATALOADER.NUM_WORKERS threads working ahead of an ordered consumer (the reference: loader
    worker processes, daod/data/build.py:289-367): whatever the worker count, the items come out in the json's image-id
    order with identical tensors, boxes, classes and dict keys."""
    D = sfod.data
    sizes = [(40 + 3 * i, 80 - 2 * i) for i in range(13)]
    jf, _ = _make_dataset(tmp_path, sizes)
    D.register_coco_instances("tiny_pool", jf, str(tmp_path))
    got = {}
    for workers in (0, 1, 4):
        cfg = sfod.config.setup_cfg(HOT, ["MODEL.DEVICE", "cpu", "DATASETS.TRAIN_TARGET", "('tiny_pool',)",
                                          "INPUT.MIN_SIZE_TRAIN", "(32,)", "INPUT.MAX_SIZE_TRAIN", "64",
                                          "DATALOADER.NUM_WORKERS", str(workers), "MODEL.ROI_HEADS.NUM_CLASSES", "3"])
        got[workers] = D.CocoTargetDataset(cfg, torch.device("cpu"), ["tiny_pool"], train=True)
    ref = got[0]
    assert [it["image_id"] for it in ref.items] == [100 + i for i in range(13)]
    for workers in (1, 4):
        ds = got[workers]
        assert len(ds) == len(ref)
        for a, b in zip(ref.items, ds.items):
            assert a.keys() == b.keys() and a["image_id"] == b["image_id"] and a["size"] == b["size"]
            assert torch.equal(a["image"], b["image"]) and torch.equal(a["boxes"], b["boxes"]) and torch.equal(a["classes"], b["classes"])
