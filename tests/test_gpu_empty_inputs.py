"""Empty inputs through the C ABI: an image without proposals / candidates / ground truth beside a normal one, zero ROIs, zero
rows.  Detectron2's behaviour on them (the reference reaches it at rpn.py:45-56, source_free_adaptive_teacher_roi_heads.py:161-205,
source_free_adaptive_teacher.py:150-183): an empty image contributes nothing and never disturbs its neighbour in the batch --
NMS of nothing keeps nothing, ``fast_rcnn_inference`` of zero proposals (or of rows that all fail ``score > 0.05``) returns zero
detections and zero pseudo labels, the Matcher without ground truth labels every anchor / proposal background (no "ignore"),
ROIAlign of zero boxes is a [0, ...] tensor."""
import pytest
import torch

from oracle import box_ops as OB
from oracle import model as om

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _boxes(n, g, span=400.0):
    xy = torch.rand(n, 2, generator=g) * span
    wh = torch.rand(n, 2, generator=g) * 80 + 4
    return torch.cat([xy, xy + wh], 1)


def test_nms_of_an_empty_image_keeps_nothing_and_leaves_its_neighbour_alone(native):
    g = torch.Generator().manual_seed(0)
    n = 300
    b1 = _boxes(n, g)
    s1 = torch.rand(n, generator=g)
    order = torch.argsort(s1, descending=True)
    boxes = torch.stack([torch.full((n, 4), float("nan")), b1[order]]).to(DEV)        # image 0: garbage behind a count of 0
    cnt = torch.tensor([0, n], dtype=torch.int32, device=DEV)
    keep_idx, keep_cnt = native.nms(boxes, 0.5, n, n_per_image=cnt)
    assert keep_cnt.tolist()[0] == 0
    ref = OB.nms(b1, s1, 0.5)
    assert order[keep_idx[1, : keep_cnt[1].item()].cpu().long()].tolist() == ref.tolist()
    # both empty
    keep_idx, keep_cnt = native.nms(boxes, 0.5, n, n_per_image=torch.zeros(2, dtype=torch.int32, device=DEV))
    assert keep_cnt.tolist() == [0, 0]


def test_teacher_post_processing_with_an_image_without_proposals_and_one_without_candidates(native):
    cfg = om.Cfg()
    K, P = cfg.num_classes, 256
    g = torch.Generator().manual_seed(1)
    sizes = [(300, 500)] * 3
    props = torch.stack([_boxes(P, g) for _ in range(3)])
    scores = torch.randn(3 * P, K + 1, generator=g) * 3.0
    scores[2 * P:, :K] = -30.0                       # image 2: every class probability below 0.05 -> no candidates
    scores[2 * P:, K] = 30.0
    deltas = torch.randn(3 * P, 4 * K, generator=g) * 0.5
    pred = torch.cat([scores, deltas], 1).contiguous().to(DEV)
    count = torch.tensor([0, P, P], dtype=torch.int32, device=DEV)       # image 0: no proposals at all
    sizes_dev = torch.tensor(sizes, dtype=torch.int32, device=DEV)
    d = native.frcnn_inference(pred, K, props.to(DEV), count, sizes_dev, 0.05, 0.5, 100, 0.8)
    torch.cuda.synchronize()
    assert d["cand_count"].tolist()[0] == 0 and d["cand_count"].tolist()[2] == 0
    assert d["det_count"].tolist()[0] == 0 and d["det_count"].tolist()[2] == 0
    assert d["gt_count"].tolist()[0] == 0 and d["gt_count"].tolist()[2] == 0
    ref = om.fast_rcnn_inference(scores[P:2 * P], deltas[P:2 * P], [props[1]], [sizes[1]], cfg)[0]
    c = d["det_count"][1].item()
    assert c == len(ref["scores"]) > 0
    assert torch.equal(d["det_classes"][1, :c].cpu().long(), ref["classes"])
    torch.testing.assert_close(d["det_scores"][1, :c].cpu(), ref["scores"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(d["det_boxes"][1, :c].cpu(), ref["boxes"], rtol=1e-5, atol=1e-3)
    n_pseudo = int((ref["scores"] > 0.8).sum())
    assert d["gt_count"][1].item() == n_pseudo
    # the logged scalars of such a batch stay finite (mean over images, empty ones count as zero)
    rl = torch.zeros(3, 8, device=DEV)
    m = native.teacher_metrics(d["det_scores"], d["det_count"], rl, torch.zeros(3, dtype=torch.int32, device=DEV), d["gt_count"], 0.8)
    assert torch.isfinite(m).all()


def test_matchers_without_ground_truth_label_everything_background(native):
    g = torch.Generator().manual_seed(2)
    cell = torch.as_tensor(OB.cell_anchors([32, 64, 128, 256, 512], [0.5, 1.0, 2.0]), dtype=torch.float32).to(DEV)
    B, Hf, Wf, G = 2, 6, 9, 8
    gt = torch.zeros(B, G, 4)
    gt[1, :3] = _boxes(3, g, span=150.0)
    gcnt = torch.tensor([0, 3], dtype=torch.int32, device=DEV)
    matched, labels = native.anchor_match(cell, B, Hf, Wf, 32, gt.to(DEV), gcnt, 0.3, 0.7)
    assert (labels[0] == 0).all()                                    # d2 Matcher on an empty match matrix: all 0, none -1
    assert (labels[1] == 1).any()                                    # the neighbour still gets its low-quality matches
    P = 64
    boxes = torch.stack([_boxes(P, g, span=150.0) for _ in range(B)]).to(DEV)
    pcnt = torch.tensor([P, P], dtype=torch.int32, device=DEV)
    gcls = torch.randint(0, 8, (B, G), generator=g).to(torch.int32).to(DEV)
    m2, cls = native.roi_match(boxes, pcnt, gt.to(DEV), gcls, gcnt, 0.5, 8)
    assert (cls[0] == 8).all()                                       # roi_heads.py:179-190: no GT -> every proposal is background


def test_roi_align_of_zero_boxes_and_softmax_of_zero_rows(native):
    feat = native.cast(torch.randn(1, 12, 20, 64, device=DEV), native.SPLIT_DTYPE)
    rois = torch.zeros(0, 5, device=DEV)
    out = native.roi_align_fwd(feat, rois, 7, 1.0 / 32)
    assert tuple(out.shape[:2]) == (0, 49)
    dfeat = native.roi_align_bwd(torch.zeros(0, 49, 64, device=DEV), rois, (1, 12, 20, 64), 7, 1.0 / 32)
    torch.cuda.synchronize()
    assert float(dfeat.abs().max()) == 0.0
    p = native.predict_probs(torch.zeros(0, 9, device=DEV))
    assert tuple(p.shape) == (0, 9)
