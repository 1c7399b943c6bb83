"""The oracle's Detectron2 / torchvision half against upstream-recorded fixtures.

``tests/golden/upstream_ref.npz`` is written by ``python -m oracle.gen_golden --upstream`` on a machine where
detectron2 and torchvision import (they are absent from the build image and from /root/reference, so until someone runs
that command the comparisons below are skipped -- the oracle is still run on every case here, so the cases themselves
cannot rot).  Tolerances: indices / labels bit-exact, floats 1e-6.

Since round 5 a large part of that half IS pinned another way: Detectron2's own unit-test goldens for the RPN, the ROI
heads' training path, the losses, anchors, Matcher, IoU, ROIAlign and the LR schedule (tests/test_oracle_d2_golden.py,
tests/test_oracle_ops.py).  What only this file would pin: ``fast_rcnn_inference``, multi-class ``batched_nms`` in both
strategies, ``roi_align``'s backward, random-input sweeps of everything.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import box_ops as OB
from oracle import model as om
from oracle.roi_align import roi_align
from oracle.upstream_cases import cases

PATH = os.path.join(GOLDEN, "upstream_ref.npz")
FX = np.load(PATH, allow_pickle=False) if os.path.exists(PATH) else None


def check(name, got, exact=False, rtol=1e-6, atol=1e-6):
    """compare with the upstream fixture when it is there; always make sure the oracle's value is sane"""
    g = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert np.isfinite(g.astype(np.float64)).all() or "get" in name, name
    if FX is None or name not in FX:
        return False
    ref = FX[name]
    assert g.shape == ref.shape, (name, g.shape, ref.shape)
    if exact:
        assert np.array_equal(g.astype(np.int64), ref.astype(np.int64)), name
    else:
        np.testing.assert_allclose(g, ref, rtol=rtol, atol=atol, err_msg=name)
    return True


def test_oracle_on_the_upstream_cases():
    c = cases()
    cfg = om.Cfg()
    pinned = []
    cell = OB.cell_anchors(cfg.anchor_sizes, cfg.anchor_ratios)
    for i, (h, w) in enumerate(c["anchors"]["sizes"].tolist()):
        pinned.append(check(f"anchors/{i}", OB.grid_anchors(h, w, cfg.stride, cell)))
    d = c["box2box"]
    for tag, wts in (("rpn", cfg.rpn_bbox_weights), ("roi", cfg.roi_bbox_weights)):
        pinned.append(check(f"box2box/{tag}/get", OB.get_deltas(d["src"], d["tgt"], wts), rtol=1e-5))
        pinned.append(check(f"box2box/{tag}/apply", OB.apply_deltas(d["deltas"], d["src"], wts), rtol=1e-5, atol=1e-3))
        pinned.append(check(f"box2box/{tag}/apply_k", OB.apply_deltas(d["deltas_k"], d["src"], wts), rtol=1e-5, atol=1e-3))
    d = c["matcher"]
    M = OB.pairwise_iou(d["gt"], d["cand"])
    pinned.append(check("matcher/iou", M))
    for tag, (thr, labels, lq) in (("rpn", (list(cfg.rpn_iou_thr), [0, -1, 1], True)), ("roi", (list(cfg.roi_iou_thr), [0, 1], False))):
        idx, lab = OB.matcher(M, thr, labels, lq)
        pinned.append(check(f"matcher/{tag}/idx", idx, exact=True))
        pinned.append(check(f"matcher/{tag}/labels", lab, exact=True))
    d = c["nms"]
    for thr in (0.7, 0.5):
        keep = OB.nms(d["boxes"], d["scores"], thr)
        assert len(keep) > 0 and len(torch.unique(keep)) == len(keep)
        pinned.append(check(f"nms/{thr}", keep, exact=True))
    d = c["batched_nms"]
    trick = OB.batched_nms(d["boxes"], d["scores"], d["idxs"], 0.5, numel_limit=10 ** 9)
    vanilla = OB.batched_nms(d["boxes"], d["scores"], d["idxs"], 0.5, numel_limit=0)
    assert torch.equal(torch.sort(trick).values, torch.sort(vanilla).values)      # separated classes: same keep set
    pinned.append(check("batched_nms/coordinate_trick", trick, exact=True))
    pinned.append(check("batched_nms/vanilla", vanilla, exact=True))
    d = c["roi_align"]
    x = d["feat"].clone().requires_grad_(True)
    y = roi_align(x, d["rois"], 7, 1.0 / 32, 0, True)
    y.backward(d["grad"])
    pinned.append(check("roi_align/out", y, rtol=1e-5, atol=1e-5))
    pinned.append(check("roi_align/grad_input", x.grad, rtol=1e-5, atol=1e-5))
    d = c["rpn"]
    Hf, Wf = d["hw"].tolist()
    anchors = OB.grid_anchors(Hf, Wf, cfg.stride, cell)
    losses = om.rpn_losses(anchors, d["logits"], d["deltas"], d["labels"], d["matched_gt"], cfg)
    pinned.append(check("rpn/loss_rpn_cls", losses["loss_rpn_cls"], rtol=1e-5))
    pinned.append(check("rpn/loss_rpn_loc", losses["loss_rpn_loc"], rtol=1e-5))
    sizes = [tuple(s) for s in d["image_sizes"].tolist()]
    props = om.rpn_proposals(anchors, d["logits"], d["deltas"], sizes, cfg, training=True)
    for i, (pb, pl) in enumerate(props):
        pinned.append(check(f"rpn/proposals/{i}/logits", pl, rtol=0, atol=0))
        pinned.append(check(f"rpn/proposals/{i}/boxes", pb, rtol=1e-5, atol=1e-3))
    d = c["fast_rcnn"]
    n0, n1 = d["split"].tolist()
    losses = om.fast_rcnn_losses(d["scores"], d["deltas"], d["proposals"], d["gt_classes"], d["gt_boxes"], cfg)
    pinned.append(check("fast_rcnn/loss_cls", losses["loss_cls"], rtol=1e-5))
    pinned.append(check("fast_rcnn/loss_box_reg", losses["loss_box_reg"], rtol=1e-5))
    sizes = [tuple(s) for s in d["image_sizes"].tolist()]
    dets = om.fast_rcnn_inference(d["scores"], d["deltas"], [d["proposals"][:n0], d["proposals"][n0:]], sizes, cfg)
    for i, det in enumerate(dets):
        assert len(det["scores"]) <= cfg.test_dets
        pinned.append(check(f"fast_rcnn/det/{i}/classes", det["classes"], exact=True))
        pinned.append(check(f"fast_rcnn/det/{i}/roi_idx", det["roi_idx"], exact=True))
        pinned.append(check(f"fast_rcnn/det/{i}/scores", det["scores"], rtol=1e-6))
        pinned.append(check(f"fast_rcnn/det/{i}/boxes", det["boxes"], rtol=1e-5, atol=1e-3))
    if FX is None:
        pytest.skip("PARITY UNPINNED for the Detectron2 / torchvision half: tests/golden/upstream_ref.npz is absent -- run "
                    "`python -m oracle.gen_golden --upstream` on a machine with detectron2 + torchvision and commit the file "
                    f"({len(pinned)} quantities of the oracle were computed and are sane)")
    assert all(pinned), "fixture file lacks some sections: regenerate it"
