"""Evaluation path: ``Trainer.test`` -> ``inference_on_dataset`` -> COCO-style
box AP with the per-class AP / AP50 table of ``daod/evaluation/new_cocoevaluator.py:33-112``.

The reference's evaluator is Detectron2's ``COCOEvaluator`` (``process`` / ``evaluate``) around pycocotools'
``COCOeval``; neither is vendored nor installed here, so the scoring arithmetic below restates pycocotools'
published algorithm (``cocoeval.py``: ``computeIoU``, ``evaluateImg``, ``accumulate``, ``summarize``; bbox
IoU type, default parameters: IoU .50:.05:.95, 101 recall points, maxDets 1/10/100, areas all / small <32^2 /
medium / large >96^2, crowd ground truth matched as "ignore" with intersection-over-detection-area).
Parity unpinned: checked against hand-computed cases (``tests/test_evaluation.py``), not against pycocotools.
Host-side numpy like the reference's (the model's forward is the device work); not on the timed path.
The reference also appends an ``F1Evaluator`` (``base.py:149``); that one is not built.
"""
import copy
import itertools
import logging
from collections import OrderedDict, defaultdict

import numpy as np
import torch

logger = logging.getLogger("sfod")

IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
MAX_DETS = [1, 10, 100]
AREA_RNG = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]


def bbox_iou_xywh(dt, gt, iscrowd):
    """pycocotools ``maskUtils.iou`` for boxes: [D,4] x [G,4] (x,y,w,h) -> [D,G]; for a crowd ground truth
    the union is the detection's area."""
    dt = np.asarray(dt, dtype=np.float64).reshape(-1, 4)
    gt = np.asarray(gt, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((len(dt), len(gt)))
    for j in range(len(gt)):
        ga = gt[j, 2] * gt[j, 3]
        for i in range(len(dt)):
            da = dt[i, 2] * dt[i, 3]
            w = min(dt[i, 0] + dt[i, 2], gt[j, 0] + gt[j, 2]) - max(dt[i, 0], gt[j, 0])
            h = min(dt[i, 1] + dt[i, 3], gt[j, 1] + gt[j, 3]) - max(dt[i, 1], gt[j, 1])
            if w <= 0 or h <= 0:
                continue
            inter = w * h
            out[i, j] = inter / (da if iscrowd[j] else da + ga - inter)
    return out


class COCOevalBBox:
    """``COCOeval(cocoGt, cocoDt, "bbox")``: ``evaluate(); accumulate(); summarize()`` -> ``stats`` [12],
    ``eval["precision"]`` [T,R,K,A,M] and ``eval["recall"]`` [T,K,A,M]."""

    def __init__(self, gt_anns, dt_anns, cat_ids, img_ids):
        self.cat_ids, self.img_ids = sorted(set(cat_ids)), sorted(set(img_ids))
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for i, g in enumerate(gt_anns):
            g = dict(g)
            g.setdefault("iscrowd", 0)
            g.setdefault("area", g["bbox"][2] * g["bbox"][3])
            g.setdefault("id", i + 1)
            g["ignore"] = g.get("ignore", 0) or g["iscrowd"]
            self._gts[g["image_id"], g["category_id"]].append(g)
        for i, d in enumerate(dt_anns):            # COCO.loadRes: area = w*h, id = running index
            d = dict(d)
            d["area"] = d["bbox"][2] * d["bbox"][3]
            d["id"] = i + 1
            self._dts[d["image_id"], d["category_id"]].append(d)
        self.eval, self.stats = {}, None

    def _evaluate_img(self, img, cat, arng, max_det, ious_cache):
        gt, dt = self._gts[img, cat], self._dts[img, cat]
        if len(gt) == 0 and len(dt) == 0:
            return None
        g_ignore = np.array([1 if (g["ignore"] or g["area"] < arng[0] or g["area"] > arng[1]) else 0 for g in gt])
        gtind = np.argsort(g_ignore, kind="mergesort")
        gt = [gt[i] for i in gtind]
        dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in dtind[0:max_det]]
        iscrowd = [int(o["iscrowd"]) for o in gt]
        ious = ious_cache[img, cat]
        ious = ious[:, gtind] if len(ious) > 0 else ious
        T, G, D = len(IOU_THRS), len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        gt_ig = np.array([g_ignore[i] for i in gtind]) if G else np.zeros(0)
        dt_ig = np.zeros((T, D))
        if len(ious) != 0:
            for tind, t in enumerate(IOU_THRS):
                for dind in range(D):
                    iou = min([t, 1 - 1e-10])
                    m = -1
                    for gind in range(G):
                        if gtm[tind, gind] > 0 and not iscrowd[gind]:
                            continue
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gind] == 1:
                            break
                        if ious[dind, gind] < iou:
                            continue
                        iou = ious[dind, gind]
                        m = gind
                    if m == -1:
                        continue
                    dt_ig[tind, dind] = gt_ig[m]
                    dtm[tind, dind] = gt[m]["id"]
                    gtm[tind, m] = dt[dind]["id"]
        a = np.array([d["area"] < arng[0] or d["area"] > arng[1] for d in dt]).reshape((1, len(dt)))
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
        return {"dtMatches": dtm, "dtScores": [d["score"] for d in dt], "gtIgnore": gt_ig, "dtIgnore": dt_ig}

    def evaluate(self):
        ious = {}
        for img in self.img_ids:
            for cat in self.cat_ids:
                gt, dt = self._gts[img, cat], self._dts[img, cat]
                if len(gt) == 0 and len(dt) == 0:
                    ious[img, cat] = []
                    continue
                inds = np.argsort([-d["score"] for d in dt], kind="mergesort")
                dt = [dt[i] for i in inds][: MAX_DETS[-1]]
                ious[img, cat] = bbox_iou_xywh([d["bbox"] for d in dt], [g["bbox"] for g in gt],
                                               [int(g["iscrowd"]) for g in gt])
        self._eval_imgs = [self._evaluate_img(img, cat, arng, MAX_DETS[-1], ious)
                           for cat in self.cat_ids for arng in AREA_RNG for img in self.img_ids]

    def accumulate(self):
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cat_ids), len(AREA_RNG), len(MAX_DETS)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        I0 = len(self.img_ids)
        for k in range(K):
            for a in range(A):
                for m, max_det in enumerate(MAX_DETS):
                    E = [self._eval_imgs[k * A * I0 + a * I0 + i] for i in range(I0)]
                    E = [e for e in E if e is not None]
                    if len(E) == 0:
                        continue
                    dt_scores = np.concatenate([e["dtScores"][0:max_det] for e in E])
                    inds = np.argsort(-dt_scores, kind="mergesort")
                    dtm = np.concatenate([e["dtMatches"][:, 0:max_det] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e["dtIgnore"][:, 0:max_det] for e in E], axis=1)[:, inds]
                    gt_ig = np.concatenate([e["gtIgnore"] for e in E])
                    npig = np.count_nonzero(gt_ig == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
                    fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        q = np.zeros((R,))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        inds_r = np.searchsorted(rc, REC_THRS, side="left")
                        for ri, pi in enumerate(inds_r):
                            if pi >= nd:
                                break
                            q[ri] = pr[pi]
                        precision[t, :, k, a, m] = q
        self.eval = {"precision": precision, "recall": recall}

    def _summarize(self, ap, iou_thr=None, area=0, max_det=100):
        mind = MAX_DETS.index(max_det)
        s = self.eval["precision"] if ap else self.eval["recall"]
        if iou_thr is not None:
            s = s[np.where(np.isclose(IOU_THRS, iou_thr))[0]]
        s = s[:, :, :, area, mind] if ap else s[:, :, area, mind]
        return -1.0 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))

    def summarize(self):
        z = self._summarize
        self.stats = np.array([z(1), z(1, .5), z(1, .75), z(1, area=1), z(1, area=2), z(1, area=3),
                               z(0, max_det=1), z(0, max_det=10), z(0), z(0, area=1), z(0, area=2), z(0, area=3)])
        return self.stats


def instances_to_coco_json(instances, img_id):
    """d2 ``instances_to_coco_json`` for boxes: XYXY -> XYWH, one dict per detection."""
    if len(instances) == 0:
        return []
    boxes = instances.pred_boxes.tensor.detach().float().cpu().numpy().copy()
    boxes[:, 2] -= boxes[:, 0]
    boxes[:, 3] -= boxes[:, 1]
    scores = instances.scores.detach().float().cpu().tolist()
    classes = instances.pred_classes.detach().cpu().tolist()
    return [{"image_id": img_id, "category_id": int(classes[k]), "bbox": boxes[k].tolist(), "score": scores[k]}
            for k in range(len(scores))]


class NewCOCOEvaluator:
    """``NewCOCOEvaluator(dataset_name, output_dir=...)`` (``new_cocoevaluator.py:33``): d2's ``reset`` /
    ``process`` / ``evaluate`` protocol; ``evaluate`` returns ``OrderedDict(bbox={AP, AP50, AP75, APs, APm,
    APl, "AP-<class>", "AP-<class>_AP50"})`` in percent, NaN where undefined.  ``dataset_dicts`` are d2
    dataset records (``annotations`` with ``bbox`` XYWH_ABS / ``category_id`` / ``iscrowd``)."""

    METRICS = ["AP", "AP50", "AP75", "APs", "APm", "APl"]

    def __init__(self, dataset_name, dataset_dicts, class_names, distributed=True, output_dir=None):
        self.dataset_name, self.class_names = dataset_name, list(class_names)
        self._distributed, self._output_dir = distributed, output_dir
        self._gt = []
        self._img_ids = []
        for rec in dataset_dicts:
            self._img_ids.append(rec["image_id"])
            for ann in rec.get("annotations", []):
                x, y, w, h = ann["bbox"]
                if ann.get("bbox_mode", 1) == 0:          # XYXY_ABS -> XYWH_ABS
                    w, h = w - x, h - y
                self._gt.append({"image_id": rec["image_id"], "category_id": int(ann["category_id"]),
                                 "bbox": [x, y, w, h], "iscrowd": int(ann.get("iscrowd", 0)),
                                 "area": float(ann.get("area", w * h))})
        self.reset()

    def reset(self):
        self._predictions = []

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            pred = {"image_id": inp["image_id"]}
            if "instances" in out:
                pred["instances"] = instances_to_coco_json(out["instances"], inp["image_id"])
            self._predictions.append(pred)

    def evaluate(self):
        preds = self._predictions
        if self._distributed and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            gathered = [None] * torch.distributed.get_world_size()
            torch.distributed.all_gather_object(gathered, preds)
            preds = list(itertools.chain(*gathered))
            if torch.distributed.get_rank() != 0:
                return {}
        if len(preds) == 0:
            logger.warning("[COCOEvaluator] Did not receive valid predictions.")
            return {}
        results = OrderedDict()
        if "instances" in preds[0]:
            coco_results = list(itertools.chain(*[p["instances"] for p in preds]))
            coco_eval = None
            if len(coco_results) > 0:
                coco_eval = COCOevalBBox(self._gt, coco_results, range(len(self.class_names)), self._img_ids)
                coco_eval.evaluate()
                coco_eval.accumulate()
                coco_eval.summarize()
            results["bbox"] = self._derive_coco_results(coco_eval, "bbox", self.class_names)
        return copy.deepcopy(results)

    def _derive_coco_results(self, coco_eval, iou_type, class_names=None):
        """``new_cocoevaluator.py:34-112``: the six standard numbers + per-category AP and AP50."""
        metrics = self.METRICS
        if coco_eval is None:
            logger.warning("No predictions from the model!")
            return {metric: float("nan") for metric in metrics}
        results = {metric: float(coco_eval.stats[idx] * 100 if coco_eval.stats[idx] >= 0 else "nan")
                   for idx, metric in enumerate(metrics)}
        if class_names is None or len(class_names) <= 1:
            return results
        precisions = coco_eval.eval["precision"]           # (iou, recall, cls, area range, max dets)
        assert len(class_names) == precisions.shape[2]
        per_category = []
        for idx, name in enumerate(class_names):
            precision = precisions[:, :, idx, 0, -1]
            p50 = precision[0]
            p50 = p50[p50 > -1]
            ap50 = np.mean(p50) if p50.size else float("nan")
            precision = precision[precision > -1]
            ap = np.mean(precision) if precision.size else float("nan")
            per_category.append((name, float(ap * 100)))
            per_category.append((name + "_AP50", float(ap50 * 100)))
        results.update({"AP-" + name: ap for name, ap in per_category})
        return results


def inference_on_dataset(model, data_loader, evaluator):
    """d2 ``inference_on_dataset``: eval mode + no_grad, ``evaluator.process`` per batch, ``evaluate`` at the
    end; the model's previous training mode is restored."""
    evaluator.reset()
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            for inputs in data_loader:
                outputs = model(inputs)
                evaluator.process(inputs, outputs)
    finally:
        model.train(was_training)
    results = evaluator.evaluate()
    return {} if results is None else results


def print_csv_format(results):
    """d2 ``print_csv_format``: one ``copypaste:`` block per task, first the metric names, then the values."""
    lines = []
    for task, res in results.items():
        if isinstance(res, dict):
            important = [(k, v) for k, v in res.items() if "-" not in k]
            lines.append("copypaste: Task: {}".format(task))
            lines.append("copypaste: " + ",".join(k for k, _ in important))
            lines.append("copypaste: " + ",".join("{0:.4f}".format(v) for _, v in important))
        else:
            lines.append("copypaste: {}={}".format(task, res))
    for ln in lines:
        logger.info(ln)
    return lines
