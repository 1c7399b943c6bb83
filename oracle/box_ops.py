"""Oracle (test infrastructure): anchors, box coding, IoU, Matcher, sampling, NMS.

CPU restatement in fp32 torch/numpy of the Detectron2 / torchvision semantics the
reference reaches through ``daod/modeling/proposal_generator/rpn.py:25,45,54`` and
``daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py:165-215``
(SURVEY.md Appendix A.3, A.5-A.9).  Detectron2 / torchvision are not installed: these follow their published
algorithms and are pinned by Detectron2's / torchvision's own unit-test vectors (anchors, Matcher, pairwise_iou, box
transform: tests/test_oracle_ops.py, tests/test_oracle_d2_golden.py) and, for ``nms`` / ``batched_nms``, by the independent
greedy NMS inside HuggingFace ``transformers`` (tests/test_oracle_hf_nms.py); oracle/__init__.py lists what is still unpinned.
"""
import math

import numpy as np
import torch

SCALE_CLAMP = math.log(1000.0 / 16)


# ---------------------------------------------------------------------------------------------
# A.3  DefaultAnchorGenerator (offset 0.0)
# ---------------------------------------------------------------------------------------------
def cell_anchors(sizes, aspect_ratios):
    """detectron2 DefaultAnchorGenerator.generate_cell_anchors: size outer, ratio inner."""
    out = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(hf, wf, stride, cell):
    """Anchors for one feature level, order (y, x, a) with a fastest -> [hf*wf*A, 4] fp32."""
    shifts_x = torch.arange(0, wf * stride, step=stride, dtype=torch.float32)
    shifts_y = torch.arange(0, hf * stride, step=stride, dtype=torch.float32)
    sy, sx = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
    sx = sx.reshape(-1)
    sy = sy.reshape(-1)
    shifts = torch.stack((sx, sy, sx, sy), dim=1)
    return (shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4)


# ---------------------------------------------------------------------------------------------
# A.5  Box2BoxTransform
# ---------------------------------------------------------------------------------------------
def get_deltas(src, tgt, weights):
    wx, wy, ww, wh = weights
    sw = src[:, 2] - src[:, 0]
    sh = src[:, 3] - src[:, 1]
    scx = src[:, 0] + 0.5 * sw
    scy = src[:, 1] + 0.5 * sh
    tw = tgt[:, 2] - tgt[:, 0]
    th = tgt[:, 3] - tgt[:, 1]
    tcx = tgt[:, 0] + 0.5 * tw
    tcy = tgt[:, 1] + 0.5 * th
    dx = wx * (tcx - scx) / sw
    dy = wy * (tcy - scy) / sh
    dw = ww * torch.log(tw / sw)
    dh = wh * torch.log(th / sh)
    return torch.stack((dx, dy, dw, dh), dim=1)


def apply_deltas(deltas, boxes, weights):
    """deltas [N, k*4], boxes [N, 4] -> [N, k*4]."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    wx, wy, ww, wh = weights
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww
    dh = deltas[:, 3::4] / wh
    dw = torch.clamp(dw, max=SCALE_CLAMP)
    dh = torch.clamp(dh, max=SCALE_CLAMP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    x1 = pcx - 0.5 * pw
    y1 = pcy - 0.5 * ph
    x2 = pcx + 0.5 * pw
    y2 = pcy + 0.5 * ph
    return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def clip_boxes(boxes, image_size):
    """Boxes.clip: x in [0, w], y in [0, h]; image_size = (h, w)."""
    h, w = image_size
    x1 = boxes[..., 0].clamp(min=0, max=w)
    y1 = boxes[..., 1].clamp(min=0, max=h)
    x2 = boxes[..., 2].clamp(min=0, max=w)
    y2 = boxes[..., 3].clamp(min=0, max=h)
    return torch.stack((x1, y1, x2, y2), dim=-1)


def nonempty(boxes, threshold=0.0):
    w = boxes[:, 2] - boxes[:, 0]
    h = boxes[:, 3] - boxes[:, 1]
    return (w > threshold) & (h > threshold)


# ---------------------------------------------------------------------------------------------
# A.6  pairwise IoU  (detectron2.structures.pairwise_iou, no +1)
# ---------------------------------------------------------------------------------------------
def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def pairwise_iou(b1, b2):
    a1 = box_area(b1)
    a2 = box_area(b2)
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh[..., 0] * wh[..., 1]
    iou = torch.where(
        inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype)
    )
    return iou


# ---------------------------------------------------------------------------------------------
# A.8  Matcher
# ---------------------------------------------------------------------------------------------
def matcher(M, thresholds, labels, allow_low_quality):
    """M [G, N] -> (matches int64 [N], match_labels int8 [N])."""
    n = M.shape[1]
    if M.numel() == 0:
        return (
            torch.zeros(n, dtype=torch.int64),
            torch.full((n,), labels[0], dtype=torch.int8),
        )
    thr = [-float("inf")] + list(thresholds) + [float("inf")]
    vals, matches = M.max(dim=0)
    out = torch.full((n,), 1, dtype=torch.int8)
    for l, lo, hi in zip(labels, thr[:-1], thr[1:]):
        out[(vals >= lo) & (vals < hi)] = l
    if allow_low_quality:
        best, _ = M.max(dim=1)
        _, pred_idx = torch.nonzero(M == best[:, None], as_tuple=True)
        out[pred_idx] = 1
    return matches, out


# ---------------------------------------------------------------------------------------------
# A.9  subsample_labels.  The reference draws torch.randperm on the model's device, a stream
# no other implementation can reproduce; the random choice is therefore an INPUT here: one
# uint32 key per element, the sample is the n candidates with the smallest (key, index).
# ---------------------------------------------------------------------------------------------
def subsample_labels(labels, num_samples, positive_fraction, bg_label, keys):
    labels = labels.to(torch.int64)
    pos = torch.nonzero((labels != -1) & (labels != bg_label)).squeeze(1)
    neg = torch.nonzero(labels == bg_label).squeeze(1)
    num_pos = min(pos.numel(), int(num_samples * positive_fraction))
    num_neg = min(neg.numel(), num_samples - num_pos)
    keys = keys.to(torch.int64)

    def pick(cand, n):
        k = keys[cand] * (1 << 32) + cand
        order = torch.argsort(k)
        return cand[order[:n]].sort().values

    return pick(pos, num_pos), pick(neg, num_neg)


# ---------------------------------------------------------------------------------------------
# A.6  NMS (torchvision.ops.nms greedy, strict '>', fp32, stable descending score order)
# ---------------------------------------------------------------------------------------------
def nms(boxes, scores, thr):
    """Returns kept original indices in descending-score order (int64 tensor)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    b = boxes.detach().to(torch.float32).numpy()
    s = scores.detach().to(torch.float32).numpy()
    order = np.argsort(-s, kind="stable")
    b = b[order]
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr32 = np.float32(thr)
    zero = np.float32(0)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 == n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(zero, xx2 - xx1)
        h = np.maximum(zero, yy2 - yy1)
        inter = (w * h).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / ((areas[i] + areas[i + 1:]) - inter)
        suppressed[i + 1:] |= ovr > thr32
    return torch.from_numpy(order[np.asarray(keep, dtype=np.int64)].astype(np.int64))


def batched_nms(boxes, scores, idxs, thr, numel_limit=20000):
    """torchvision.ops.batched_nms as reached via detectron2.layers.batched_nms.

    ``numel_limit`` is torchvision's strategy switch: 20000 for a GPU tensor (the
    reference's device), 4000 on CPU.  Above it: per-class loop on raw coordinates
    (``_batched_nms_vanilla``), else the coordinate-offset trick.
    """
    boxes = boxes.float()
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.int64)
    if boxes.numel() > numel_limit:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for cid in torch.unique(idxs):
            cur = torch.where(idxs == cid)[0]
            k = nms(boxes[cur], scores[cur], thr)
            keep_mask[cur[k]] = True
        keep_idx = torch.where(keep_mask)[0]
        order = torch.sort(scores[keep_idx], descending=True, stable=True)[1]
        return keep_idx[order]
    max_coord = boxes.max()
    offsets = idxs.to(boxes) * (max_coord + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, thr)
