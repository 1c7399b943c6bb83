"""Evaluation path: COCO box AP restatement against hand-computed cases, the
NewCOCOEvaluator result keys (new_cocoevaluator.py:34-112), samplers / loader / Trainer.test plumbing."""
import math
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = os.path.join(ROOT, "configs", "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml")


def _gt(img, cat, box, crowd=0):
    return {"image_id": img, "category_id": cat, "bbox": list(box), "iscrowd": crowd}


def _dt(img, cat, box, score):
    return {"image_id": img, "category_id": cat, "bbox": list(box), "score": score}


def _run(sfod, gts, dts, cats, imgs):
    ev = sfod.evaluation.COCOevalBBox(gts, dts, cats, imgs)
    ev.evaluate()
    ev.accumulate()
    ev.summarize()
    return ev


def test_ap_hand_case_tp_fp_tp(sfod):
    """Detections (by score) TP, FP, TP(IoU 0.82): per-threshold AP from the 101-point interpolation by hand."""
    gts = [_gt(1, 0, (0, 0, 10, 10)), _gt(1, 0, (20, 20, 10, 10))]
    dts = [_dt(1, 0, (20, 20, 10, 8.2), .7), _dt(1, 0, (0, 0, 10, 10), .9), _dt(1, 0, (50, 50, 10, 10), .8)]
    ev = _run(sfod, gts, dts, [0, 1], [1])
    hi = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101        # third detection is a TP: thresholds .50 .. .80
    lo = 51.0 / 101                                 # ... a FP: thresholds .85 .. .95
    p = ev.eval["precision"]
    assert p.shape == (10, 101, 2, 4, 3)
    for t in range(10):
        assert abs(p[t, :, 0, 0, 2].mean() - (hi if t < 7 else lo)) < 1e-12
    assert (p[:, :, 1] == -1).all()                 # class 1: no ground truth, no detections
    assert abs(ev.stats[0] - (7 * hi + 3 * lo) / 10) < 1e-12
    assert abs(ev.stats[1] - hi) < 1e-12 and abs(ev.stats[2] - hi) < 1e-12
    assert abs(ev.stats[3] - ev.stats[0]) < 1e-12   # all boxes are "small"
    assert ev.stats[4] == -1 and ev.stats[5] == -1
    # maxDets = 1: only the .9 detection counts -> recall .5, AR@1 = .5; AR@100: 1.0 up to .80, .5 above
    assert abs(ev.stats[6] - 0.5) < 1e-12
    assert abs(ev.stats[8] - (7 * 1.0 + 3 * 0.5) / 10) < 1e-12


def test_ap_crowd_is_ignored_and_scores_rank_across_images(sfod):
    # detections inside a crowd region are neither TP nor FP, a crowd can absorb several
    gts = [_gt(1, 0, (0, 0, 10, 10)), _gt(1, 0, (100, 100, 50, 50), crowd=1)]
    dts = [_dt(1, 0, (0, 0, 10, 10), .9), _dt(1, 0, (110, 110, 10, 10), .8), _dt(1, 0, (120, 120, 10, 10), .7)]
    ev = _run(sfod, gts, dts, [0], [1])
    assert abs(ev.stats[0] - 1.0) < 1e-12
    # ranking is global: image 1's FP (.9) precedes image 2's TP (.8); image 1's GT stays unmatched
    gts = [_gt(1, 0, (0, 0, 10, 10)), _gt(2, 0, (0, 0, 10, 10))]
    dts = [_dt(2, 0, (0, 0, 10, 10), .8), _dt(1, 0, (40, 40, 10, 10), .9)]
    ev = _run(sfod, gts, dts, [0], [1, 2])
    assert abs(ev.stats[1] - 51 * 0.5 / 101) < 1e-12
    # one ground truth can be claimed once: the lower-scored duplicate is a FP
    gts = [_gt(1, 0, (0, 0, 10, 10))]
    dts = [_dt(1, 0, (0, 0, 10, 10), .9), _dt(1, 0, (0, 0, 10, 10), .8)]
    ev = _run(sfod, gts, dts, [0], [1])
    assert abs(ev.stats[0] - 1.0) < 1e-12           # precision envelope: recall 1 is reached at precision 1
    # area ranges: a 40x40 box is "medium"; detections outside the range and unmatched are ignored
    gts = [_gt(1, 0, (0, 0, 40, 40))]
    dts = [_dt(1, 0, (0, 0, 40, 40), .9), _dt(1, 0, (200, 200, 5, 5), .95)]
    ev = _run(sfod, gts, dts, [0], [1])
    assert abs(ev.stats[4] - 1.0) < 1e-12 and ev.stats[3] == -1 and ev.stats[5] == -1
    assert abs(ev.stats[1] - 0.5) < 1e-12       # over all areas the small FP ranks first: envelope .5 everywhere


def test_new_coco_evaluator_protocol_and_keys(sfod):
    S = sfod.structures
    names = ["person", "car", "bus"]
    recs = [{"image_id": 7, "height": 100, "width": 200, "annotations": [
                {"bbox": [10, 10, 50, 40], "bbox_mode": 1, "category_id": 0, "iscrowd": 0},
                {"bbox": [100, 20, 160, 80], "bbox_mode": 0, "category_id": 1, "iscrowd": 0}]},      # XYXY_ABS
            {"image_id": 9, "height": 100, "width": 200, "annotations": [
                {"bbox": [0, 0, 30, 30], "bbox_mode": 1, "category_id": 1, "iscrowd": 0}]}]
    ev = sfod.evaluation.NewCOCOEvaluator("synthetic_test", recs, names, distributed=False)
    assert ev.evaluate() == {}                      # nothing processed

    def out(boxes, scores, classes):
        inst = S.Instances((100, 200))
        inst.pred_boxes = S.Boxes(torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4))
        inst.scores = torch.tensor(scores, dtype=torch.float32)
        inst.pred_classes = torch.tensor(classes, dtype=torch.int64)
        return {"instances": inst}

    # ground truth fed back as detections -> 100 everywhere it is defined
    ev.reset()
    ev.process([{"image_id": 7}], [out([[10, 10, 60, 50], [100, 20, 160, 80]], [.9, .8], [0, 1])])
    ev.process([{"image_id": 9}], [out([[0, 0, 30, 30]], [.7], [1])])
    res = ev.evaluate()
    assert list(res.keys()) == ["bbox"]
    r = res["bbox"]
    for k in ("AP", "AP50", "AP75", "AP-person", "AP-person_AP50", "AP-car", "AP-car_AP50"):
        assert abs(r[k] - 100.0) < 1e-9, k
    assert math.isnan(r["AP-bus"]) and math.isnan(r["AP-bus_AP50"]) and math.isnan(r["APl"])
    assert abs(r["APs"] - 100.0) < 1e-9 and abs(r["APm"] - 100.0) < 1e-9
    # detections present but none for any image's classes -> zeros; no instances at all -> NaN table
    ev.reset()
    ev.process([{"image_id": 7}], [out([[0, 0, 5, 5]], [.9], [2])])
    r = ev.evaluate()["bbox"]
    assert r["AP"] == 0.0 and r["AP-person"] == 0.0
    ev.reset()
    ev.process([{"image_id": 7}], [out([], [], [])])
    r = ev.evaluate()["bbox"]
    assert all(math.isnan(v) for v in r.values()) and list(r.keys()) == ["AP", "AP50", "AP75", "APs", "APm", "APl"]
    lines = sfod.evaluation.print_csv_format({"bbox": {"AP": 12.5, "AP50": 30.0, "AP-car": 1.0}})
    assert lines == ["copypaste: Task: bbox", "copypaste: AP,AP50", "copypaste: 12.5000,30.0000"]


def test_inference_sampler_and_test_loader(sfod):
    D = sfod.data
    shares = [list(D.InferenceSampler(10, r, 4)) for r in range(4)]
    assert shares == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    cfg = sfod.config.setup_cfg(HOT, ["MODEL.DEVICE", "cpu", "SFOD.SYNTHETIC.HEIGHT", "64", "SFOD.SYNTHETIC.WIDTH",
                                      "128", "SFOD.SYNTHETIC.NUM_TEST_IMAGES", "5", "TEST.IMS_PER_BATCH", "2",
                                      "INPUT.MIN_SIZE_TEST", "32", "SFOD.SYNTHETIC.BOXES_PER_IMAGE", "3"])
    loader = D.TestLoader(cfg, torch.device("cpu"))
    batches = list(loader)
    assert [len(b) for b in batches] == [2, 2, 1] and len(loader) == 3
    assert [d["image_id"] for b in batches for d in b] == [0, 1, 2, 3, 4]
    d = batches[0][0]
    assert d["image"].shape == (3, 32, 64) and d["image"].dtype == torch.uint8 and (d["height"], d["width"]) == (64, 128)
    recs = loader.dataset.dataset_dicts(cfg)
    assert len(recs) == 5 and len(recs[0]["annotations"]) == 3 and recs[0]["height"] == 64
    # the records are at native size; the loader's boxes are at the resized size (x0.5)
    b0 = torch.tensor(recs[0]["annotations"][0]["bbox"])
    torch.testing.assert_close(d["instances"].gt_boxes.tensor[0, :2], b0[:2] * 0.5)


def test_trainer_test_with_a_stand_in_model(sfod):
    """DefaultTrainer.test plumbing without a GPU: a stand-in model that returns the ground truth, rescaled to
    the native frame size as detector_postprocess does -> AP 100; training mode is restored."""
    S = sfod.structures
    cfg = sfod.config.setup_cfg(HOT, ["MODEL.DEVICE", "cpu", "SFOD.SYNTHETIC.HEIGHT", "64", "SFOD.SYNTHETIC.WIDTH",
                                      "128", "SFOD.SYNTHETIC.NUM_TEST_IMAGES", "4", "TEST.IMS_PER_BATCH", "3",
                                      "INPUT.MIN_SIZE_TEST", "32", "SFOD.SYNTHETIC.BOXES_PER_IMAGE", "5"])

    class Echo(torch.nn.Module):
        def forward(self, batched_inputs):
            assert not self.training
            outs = []
            for d in batched_inputs:
                inst = S.Instances((d["height"], d["width"]))
                sx = d["width"] / d["image"].shape[2]
                inst.pred_boxes = S.Boxes(d["instances"].gt_boxes.tensor * sx)
                inst.scores = torch.linspace(0.9, 0.5, len(d["instances"]))
                inst.pred_classes = d["instances"].gt_classes
                outs.append({"instances": inst})
            return outs

    model = Echo().train()
    res = sfod.engine.BaseTrainer.test(cfg, model)
    assert model.training
    assert list(res.keys()) == list(cfg.DATASETS.TEST) and len(res) == 2       # one entry per DATASETS.TEST name
    res = res[cfg.DATASETS.TEST[0]]
    assert abs(res["bbox"]["AP"] - 100.0) < 1e-6 and abs(res["bbox"]["AP50"] - 100.0) < 1e-6
    assert any(k.startswith("AP-") and k.endswith("_AP50") for k in res["bbox"])
    # a single dataset is flattened (DefaultTrainer.test)
    cfg1 = sfod.config.setup_cfg(HOT, ["MODEL.DEVICE", "cpu", "SFOD.SYNTHETIC.HEIGHT", "64", "SFOD.SYNTHETIC.WIDTH",
                                       "128", "SFOD.SYNTHETIC.NUM_TEST_IMAGES", "2", "INPUT.MIN_SIZE_TEST", "32",
                                       "DATASETS.TEST", "('synthetic_cityscapes_foggy_val',)"])
    assert "bbox" in sfod.engine.BaseTrainer.test(cfg1, model)


def test_ap50_equals_an_independent_computation_from_scikit_learns_pr_curve(sfod):
    """A second route to the same number on a randomised scene whose matching is unambiguous (every detection overlaps at most one
    ground truth, with IoU ~0.8 or 0): precision / recall points from scikit-learn's ``precision_recall_curve`` (its own sort and
    cumulative sums; recall rescaled to COCO's denominator = ALL ground truths of the class), then COCOeval's published rule --
    precision made non-increasing from the right, sampled at recall 0:0.01:1 at the first point with recall >= r, 0 beyond the last
    -- written here in numpy.  Per class and overall AP50 of ``COCOevalBBox`` must equal it."""
    from sklearn.metrics import precision_recall_curve
    rng = np.random.default_rng(5)
    K, n_img = 3, 6
    gts, dts, truth = [], [], {k: ([], []) for k in range(K)}            # class -> (is_tp flags, scores)
    npig = {k: 0 for k in range(K)}
    used_scores = set()

    def score():
        while True:
            s = round(float(rng.uniform(0.05, 0.99)), 4)
            if s not in used_scores:
                used_scores.add(s)
                return s
    for img in range(1, n_img + 1):
        cell = 0
        for k in range(K):
            for _ in range(int(rng.integers(2, 5))):
                x, y = 120.0 * (cell % 8), 120.0 * (cell // 8)            # one 120 x 120 cell per object: nothing else overlaps it
                cell += 1
                w, h = float(rng.uniform(40, 90)), float(rng.uniform(40, 90))
                gts.append(_gt(img, k, (x, y, w, h)))
                npig[k] += 1
                r = rng.uniform()
                if r < 0.7:                                              # found: same box shrunk by 10 % in w (IoU 0.9)
                    s = score()
                    dts.append(_dt(img, k, (x, y, 0.9 * w, h), s))
                    truth[k][0].append(1), truth[k][1].append(s)
                    if rng.uniform() < 0.3:                              # ... and a lower-scored duplicate: a false positive
                        s2 = round(s * 0.5, 5)
                        dts.append(_dt(img, k, (x, y, 0.9 * w, h), s2))
                        truth[k][0].append(0), truth[k][1].append(s2)
                # else: missed
            for _ in range(int(rng.integers(0, 3))):                      # false positives in empty cells
                x, y = 120.0 * (cell % 8), 120.0 * (cell // 8)
                cell += 1
                s = score()
                dts.append(_dt(img, k, (x, y, 50.0, 50.0), s))
                truth[k][0].append(0), truth[k][1].append(s)
    ev = _run(sfod, gts, dts, list(range(K)), list(range(1, n_img + 1)))
    rec_thrs = np.linspace(0.0, 1.0, 101)
    aps = []
    for k in range(K):
        y, s = np.array(truth[k][0]), np.array(truth[k][1])
        assert y.sum() > 0 and (1 - y).sum() > 0 and len(set(s.tolist())) == len(s)
        prec, rec, _ = precision_recall_curve(y, s)            # decreasing recall, last point (recall 0, precision 1) appended
        prec, rec = prec[:-1][::-1], rec[:-1][::-1]            # by descending score threshold = COCO's cumulative order
        rec = rec * y.sum() / npig[k]                          # COCO divides by every ground truth, found or not
        env = np.maximum.accumulate(prec[::-1])[::-1]          # non-increasing from the right
        idx = np.searchsorted(rec, rec_thrs, side="left")
        q = np.where(idx < len(env), env[np.minimum(idx, len(env) - 1)], 0.0)
        aps.append(q.mean())
        got = ev.eval["precision"][0, :, k, 0, 2]              # IoU 0.50, all areas, maxDets 100
        np.testing.assert_allclose(got, q, rtol=0, atol=1e-12)
    assert abs(ev.stats[1] - float(np.mean(aps))) < 1e-12
