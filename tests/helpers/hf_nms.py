"""An INDEPENDENT greedy NMS to pin the oracle's (and the device's) against: HuggingFace ``transformers``'
``OwlViTImageProcessorPil.post_process_image_guided_detection`` (transformers/models/owlvit/image_processing_pil_owlvit.py) --
"for i in argsort(-scores): if scores[i] == 0: continue; scores[box_iou(box_i, boxes) > nms_threshold] = 0" with the IoU as
inter / (area_i + area_j - inter) in fp32 and a strict '>': the semantics of torchvision's ``nms`` that Detectron2 reaches
(SURVEY A.6), written by other people for another model.  Nothing of it is copied: it is CALLED, on inputs it accepts.

The method takes boxes in centre format and converts them itself (``center_to_corners_format``): the corner boxes it works
on are what ``hf_corners`` returns, and the oracle / device get exactly those.  It returns the surviving boxes whose "alpha"
(score rescaled against the best score) is positive: with all scores in [0.5, 1) every survivor qualifies; survivors are
mapped back to indices by exact row match."""
import types

import torch


def hf_corners(centers):
    from transformers.image_transforms import center_to_corners_format
    return center_to_corners_format(centers)


def hf_greedy_nms(centers, scores, thr):
    """centers [N,4] (cx, cy, w, h) fp32, scores [N] in [0.5, 1) all distinct -> kept indices in descending score order"""
    from transformers.models.owlvit.image_processing_pil_owlvit import OwlViTImageProcessorPil
    assert float(scores.min()) >= 0.5 and float(scores.max()) < 1.0 and len(torch.unique(scores)) == len(scores)
    logits = torch.logit(scores.double()).float()          # sigmoid(logit) orders like the scores (monotone)
    assert len(torch.unique(torch.sigmoid(logits))) == len(scores), "scores too close for the sigmoid round trip"
    proc = OwlViTImageProcessorPil()
    out = types.SimpleNamespace(logits=logits.view(1, -1, 1).clone(), target_pred_boxes=centers.view(1, -1, 4).clone())
    res = proc.post_process_image_guided_detection(out, threshold=0.0, nms_threshold=float(thr), target_sizes=None)
    kept_boxes = res[0]["boxes"]
    import transformers.models.owlvit.image_processing_pil_owlvit as _mod
    corners = _mod.center_to_corners_format(centers)          # (the identity while hf_greedy_nms_corners has it bypassed)
    # rows are distinct: exact match gives the index
    idx = []
    for kb in kept_boxes:
        m = (corners == kb).all(dim=1).nonzero().flatten()
        assert len(m) == 1
        idx.append(int(m[0]))
    idx = torch.tensor(idx, dtype=torch.int64)
    order = torch.argsort(torch.sigmoid(logits)[idx], descending=True)
    return idx[order]


def random_centers(n, g, span=600.0, size=160.0):
    c = torch.rand(n, 2, generator=g) * span
    wh = torch.rand(n, 2, generator=g) * size + 4.0
    return torch.cat([c, wh], dim=1)


def distinct_scores(n, g):
    s = 0.5 + 0.49 * (torch.randperm(n, generator=g).float() + torch.rand(n, generator=g) * 0.5) / n
    return s


def hf_greedy_nms_corners(corners, scores, thr):
    """The same independent NMS on CORNER boxes with arbitrary distinct scores: the method's centre -> corner conversion is
    bypassed for the call (its module-level ``center_to_corners_format`` replaced by the identity), and the scores are replaced
    by their ranks mapped into [0.5, 1) -- greedy NMS depends on the score ORDER only.  -> kept indices, best first."""
    import transformers.models.owlvit.image_processing_pil_owlvit as mod
    n = len(scores)
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = torch.argsort(scores, descending=True, stable=True)
    ranked = torch.empty(n)
    ranked[order] = 0.5 + 0.49 * (n - torch.arange(n, dtype=torch.float32)) / (n + 1)
    saved = mod.center_to_corners_format
    mod.center_to_corners_format = lambda x: x
    try:
        return hf_greedy_nms(corners, ranked, thr)
    finally:
        mod.center_to_corners_format = saved
