"""The oracle's greedy NMS and class-wise ``batched_nms`` (oracle/box_ops.py; SURVEY A.6: torchvision ``nms`` / ``batched_nms`` as
Detectron2 reaches them from ``find_top_rpn_proposals`` and ``fast_rcnn_inference``, reference call sites
daod/modeling/proposal_generator/rpn.py:54 and daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py:161) against an
independent implementation that IS in this image: HuggingFace transformers' OwlViT post-processing (tests/helpers/hf_nms.py).
Until round 6 these two were the oracle's unpinned residue "no second implementation of NMS exists here"."""
import pytest
import torch

from helpers import hf_nms as H
from oracle import box_ops as OB


@pytest.mark.parametrize("n,thr,span", [(64, 0.7, 300.0), (65, 0.5, 200.0), (129, 0.7, 150.0), (600, 0.5, 400.0), (600, 0.3, 250.0)])
def test_greedy_nms_keep_set_and_order_equal_the_independent_implementation(n, thr, span):
    g = torch.Generator().manual_seed(n + int(100 * thr))
    centers = H.random_centers(n, g, span=span)
    scores = H.distinct_scores(n, g)
    corners = H.hf_corners(centers)
    ref = H.hf_greedy_nms(centers, scores, thr)
    got = OB.nms(corners, scores, thr)
    assert 0 < len(ref) < n, "the case must suppress something and keep something"
    assert got.tolist() == ref.tolist()


def test_threshold_is_strict_in_both():
    """IoU exactly at the threshold survives (strict '>'): two boxes with IoU 0.5 exactly, a duplicate is removed"""
    centers = torch.tensor([[5.0, 5.0, 10.0, 10.0], [5.0, 2.5, 10.0, 5.0], [5.0, 5.0, 10.0, 10.0 + 2 ** -18]])
    scores = torch.tensor([0.9, 0.8, 0.7])
    corners = H.hf_corners(centers)
    assert OB.nms(corners, scores, 0.5).tolist() == H.hf_greedy_nms(centers, scores, 0.5).tolist() == [0, 1]


@pytest.mark.parametrize("n,K,thr", [(300, 8, 0.5), (900, 8, 0.5), (400, 3, 0.7)])
def test_class_wise_batched_nms_equals_independent_nms_run_per_class(n, K, thr):
    """``batched_nms`` = boxes of different classes never suppress each other.  The independent NMS is run once per class
    (the definition); the oracle's coordinate-offset form (``idxs * (max + 1)`` added to the boxes) and its
    per-class form above the 20 000-element switch must both give that keep set.  (The offsets perturb an fp32 IoU in its last
    bits for classes > 0; the cases are seeded so that no IoU lies within 1e-5 of the threshold -- asserted.)"""
    g = torch.Generator().manual_seed(n * K)
    centers = H.random_centers(n, g, span=220.0)
    scores = H.distinct_scores(n, g)
    idxs = torch.randint(0, K, (n,), generator=g)
    corners = H.hf_corners(centers)
    iou = OB.pairwise_iou(corners, corners)
    same = idxs[:, None] == idxs[None, :]
    assert ((iou - thr).abs()[same] > 1e-5).all(), "seed puts an IoU on the threshold: pick another"
    ref = []
    for c in range(K):
        m = (idxs == c).nonzero().flatten()
        if len(m):
            ref.append(m[H.hf_greedy_nms(centers[m], scores[m], thr)])
    ref = torch.cat(ref)
    ref = ref[torch.argsort(scores[ref], descending=True)]
    trick = OB.batched_nms(corners, scores, idxs, thr, numel_limit=10 ** 9)       # coordinate trick
    vanilla = OB.batched_nms(corners, scores, idxs, thr, numel_limit=0)           # per-class loop
    assert 0 < len(ref) < n
    assert trick.tolist() == ref.tolist() and vanilla.tolist() == ref.tolist()


def test_fast_rcnn_inference_equals_a_composition_around_the_independent_nms():
    """``fast_rcnn_inference`` (Detectron2, reached from source_free_adaptive_teacher_roi_heads.py:161) end to end: the
    oracle's function against a composition whose NMS stage is the independent one -- softmax (torch), per-class decode
    (``apply_deltas``: pinned by Detectron2's test_rpn / test_fast_rcnn vectors), clip to the image, ``score > 0.05``, the
    HuggingFace greedy NMS run once per class at 0.5, then the best 100 by score.  Same detections, same order."""
    from oracle import model as om
    cfg = om.Cfg()
    K = cfg.num_classes
    g = torch.Generator().manual_seed(4)
    sizes = [(300, 500), (260, 420)]
    props, n_per = [], [180, 140]
    for (h, w), n in zip(sizes, n_per):
        xy = torch.rand(n, 2, generator=g) * torch.tensor([w * 0.8, h * 0.8])
        wh = torch.rand(n, 2, generator=g) * torch.tensor([w * 0.3, h * 0.3]) + 8
        props.append(torch.cat([xy, xy + wh], 1))
    R = sum(n_per)
    scores = torch.randn(R, K + 1, generator=g) * 2.5
    scores[:, K] += 1.0
    deltas = torch.randn(R, 4 * K, generator=g) * 0.6
    got = om.fast_rcnn_inference(scores, deltas, props, sizes, cfg)
    probs = torch.softmax(scores, -1)
    start = 0
    for i, ((h, w), n) in enumerate(zip(sizes, n_per)):
        p, d, pb = probs[start:start + n, :K], deltas[start:start + n], props[i]
        start += n
        boxes = OB.apply_deltas(d, pb, cfg.roi_bbox_weights).view(n, K, 4)
        boxes = torch.stack([boxes[..., 0].clamp(0, w), boxes[..., 1].clamp(0, h), boxes[..., 2].clamp(0, w), boxes[..., 3].clamp(0, h)], -1)
        det_b, det_s, det_c = [], [], []
        for c in range(K):
            m = (p[:, c] > 0.05).nonzero().flatten()
            if len(m) == 0:
                continue
            keep = m[H.hf_greedy_nms_corners(boxes[m, c], p[m, c], 0.5)]
            det_b.append(boxes[keep, c]); det_s.append(p[keep, c]); det_c.append(torch.full((len(keep),), c))
        det_b, det_s, det_c = torch.cat(det_b), torch.cat(det_s), torch.cat(det_c)
        order = torch.argsort(det_s, descending=True, stable=True)[:100]
        assert len(got[i]["scores"]) == len(order) == 100
        assert torch.equal(got[i]["classes"], det_c[order]) and torch.equal(got[i]["scores"], det_s[order])
        assert torch.equal(got[i]["boxes"], det_b[order])
