"""Seeded inputs of the UPSTREAM pinning cases (test infrastructure).

The arithmetic of anchors, box coding, IoU matching, NMS, ROIAlign and the RPN / Fast R-CNN losses and inference lives in
Detectron2 / torchvision, which are absent from /root/reference and from this image: that half of the oracle is PARITY
UNPINNED here.  ``python -m oracle.gen_golden --upstream`` -- on any machine where ``detectron2`` and ``torchvision``
import -- runs THEIR implementations on exactly these inputs and records the outputs in
``tests/golden/upstream_ref.npz``; ``tests/test_oracle_upstream.py`` then checks the oracle against that file.  This
module is pure torch (no upstream import) and is shared by the generator and the test, so both see the same inputs.
"""
import torch


def _boxes(n, g, span=900.0, size=300.0, min_size=2.0):
    xy = torch.rand(n, 2, generator=g) * span
    wh = torch.rand(n, 2, generator=g) * size + min_size
    return torch.cat([xy, xy + wh], dim=1)


def cases():
    """-> dict name -> dict of input tensors (all fp32 / int64, CPU)."""
    c = {}
    g = torch.Generator().manual_seed(20260)
    # anchors: the two feature-map sizes of the benchmark (600x1200 -> 18x37, 1024x2048 -> 32x64)
    c["anchors"] = {"sizes": torch.tensor([[18, 37], [32, 64]])}
    # Box2BoxTransform, both weightings, incl. deltas beyond the scale clamp
    src, tgt = _boxes(300, g), _boxes(300, g)
    deltas = torch.randn(300, 4, generator=g) * torch.tensor([0.5, 0.5, 1.5, 1.5])
    deltas[:5, 2:] = 6.0
    c["box2box"] = {"src": src, "tgt": tgt, "deltas": deltas, "deltas_k": torch.randn(300, 32, generator=g) * 0.7}
    # pairwise IoU + Matcher: RPN thresholds with low-quality matches, ROI threshold without; exact ties by duplicates
    gt = _boxes(7, g, span=600.0, size=250.0, min_size=30.0)
    gt[3] = gt[1]                                              # identical ground-truth boxes: arg-max tie (first index)
    cand = torch.cat([_boxes(500, g, span=700.0), gt + 1e-3, gt[:2]])
    c["matcher"] = {"gt": gt, "cand": cand}
    # nms: random boxes, heavy overlap, score ties; thresholds 0.7 (RPN) / 0.5 (ROI)
    nb = _boxes(1200, g, span=400.0, size=200.0)
    ns = torch.rand(1200, generator=g)
    ns[100:140] = ns[100]                                      # ties in the scores
    c["nms"] = {"boxes": nb, "scores": ns}
    # batched_nms: both strategies on the same input
    c["batched_nms"] = {"boxes": nb[:800], "scores": ns[:800], "idxs": torch.randint(0, 8, (800,), generator=g)}
    # roi_align aligned=True, adaptive grid, rois partly outside the map
    feat = torch.randn(2, 16, 19, 38, generator=g)
    rb = _boxes(40, g, span=1100.0, size=500.0, min_size=4.0)
    rb[:4] -= 150.0                                            # partly outside
    rois = torch.cat([torch.randint(0, 2, (40, 1), generator=g).float(), rb], dim=1)
    c["roi_align"] = {"feat": feat, "rois": rois, "grad": torch.randn(40, 16, 7, 7, generator=g)}
    # RPN losses / proposals on one 9x11 map, 2 images
    A, Hf, Wf = 15, 9, 11
    NA = Hf * Wf * A
    logits = torch.randn(2, NA, generator=g)
    rdeltas = torch.randn(2, NA, 4, generator=g) * 0.3
    labels = torch.randint(-1, 2, (2, NA), generator=g).to(torch.int8)
    matched = torch.stack([_boxes(NA, g, span=250.0, size=120.0), _boxes(NA, g, span=250.0, size=120.0)])
    c["rpn"] = {"logits": logits, "deltas": rdeltas, "labels": labels, "matched_gt": matched,
                "image_sizes": torch.tensor([[288, 352], [270, 340]]), "hw": torch.tensor([Hf, Wf])}
    # Fast R-CNN losses and inference
    R, K = 300, 8
    scores = torch.randn(R, K + 1, generator=g) * 2
    bdeltas = torch.randn(R, 4 * K, generator=g) * 0.5
    props = _boxes(R, g, span=250.0, size=100.0, min_size=8.0)
    gtc = torch.randint(0, K + 1, (R,), generator=g)
    gtb = props + torch.randn(R, 4, generator=g) * 4
    gtb[:, 2:] = torch.max(gtb[:, 2:], gtb[:, :2] + 1)
    c["fast_rcnn"] = {"scores": scores, "deltas": bdeltas, "proposals": props, "gt_classes": gtc, "gt_boxes": gtb,
                      "split": torch.tensor([170, 130]), "image_sizes": torch.tensor([[288, 352], [270, 340]])}
    return c
