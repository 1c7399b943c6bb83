"""``python -m oracle.gen_golden --upstream``: record what Detectron2 / torchvision compute on ``oracle.upstream_cases``.

Needs ``detectron2`` (0.6 / main) and ``torchvision`` importable; neither is in the build image, so this script has
never run there -- it is the one command that turns the oracle's "parity unpinned" half into a pinned one on any
machine that has them ("keep the fixture generator runnable against D2").  Every section is
independent: a section whose upstream API is missing is reported and skipped, the others are still recorded.
Only data is written (``tests/golden/upstream_ref.npz``: upstream outputs + the versions that produced them).
"""
import os

import numpy as np
import torch

from .upstream_cases import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "upstream_ref.npz")


def main():
    import detectron2
    import torchvision
    from detectron2.layers import ShapeSpec
    from detectron2.modeling.anchor_generator import DefaultAnchorGenerator
    from detectron2.modeling.box_regression import Box2BoxTransform
    from detectron2.modeling.matcher import Matcher
    from detectron2.structures import Boxes, Instances, pairwise_iou
    c = cases()
    out = {"versions": np.array([f"detectron2 {detectron2.__version__}", f"torchvision {torchvision.__version__}",
                                 f"torch {torch.__version__}"])}
    done, failed = [], []

    def section(name):
        def deco(fn):
            try:
                fn()
                done.append(name)
            except Exception as e:      # noqa: BLE001 -- report and go on with the other sections
                failed.append((name, repr(e)))
            return fn
        return deco

    @section("anchors")
    def _():
        gen = DefaultAnchorGenerator(sizes=[[32, 64, 128, 256, 512]], aspect_ratios=[[0.5, 1.0, 2.0]], strides=[32], offset=0.0)
        for i, (h, w) in enumerate(c["anchors"]["sizes"].tolist()):
            out[f"anchors/{i}"] = gen([torch.zeros(1, 1, h, w)])[0].tensor.numpy()

    @section("box2box")
    def _():
        d = c["box2box"]
        for tag, wts in (("rpn", (1.0, 1.0, 1.0, 1.0)), ("roi", (10.0, 10.0, 5.0, 5.0))):
            t = Box2BoxTransform(weights=wts)
            out[f"box2box/{tag}/get"] = t.get_deltas(d["src"], d["tgt"]).numpy()
            out[f"box2box/{tag}/apply"] = t.apply_deltas(d["deltas"], d["src"]).numpy()
            out[f"box2box/{tag}/apply_k"] = t.apply_deltas(d["deltas_k"], d["src"]).numpy()

    @section("matcher")
    def _():
        d = c["matcher"]
        M = pairwise_iou(Boxes(d["gt"]), Boxes(d["cand"]))
        out["matcher/iou"] = M.numpy()
        for tag, m in (("rpn", Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)),
                       ("roi", Matcher([0.5], [0, 1], allow_low_quality_matches=False))):
            idx, lab = m(M)
            out[f"matcher/{tag}/idx"], out[f"matcher/{tag}/labels"] = idx.numpy(), lab.numpy()

    @section("nms")
    def _():
        from torchvision.ops import nms
        d = c["nms"]
        for thr in (0.7, 0.5):
            out[f"nms/{thr}"] = nms(d["boxes"], d["scores"], thr).numpy()

    @section("batched_nms")
    def _():
        from torchvision.ops import boxes as tvb
        d = c["batched_nms"]
        out["batched_nms/coordinate_trick"] = tvb._batched_nms_coordinate_trick(d["boxes"], d["scores"], d["idxs"], 0.5).numpy()
        out["batched_nms/vanilla"] = tvb._batched_nms_vanilla(d["boxes"], d["scores"], d["idxs"], 0.5).numpy()

    @section("roi_align")
    def _():
        from torchvision.ops import roi_align
        d = c["roi_align"]
        x = d["feat"].clone().requires_grad_(True)
        y = roi_align(x, d["rois"], (7, 7), 1.0 / 32, 0, True)
        y.backward(d["grad"])
        out["roi_align/out"], out["roi_align/grad_input"] = y.detach().numpy(), x.grad.numpy()

    @section("rpn")
    def _():
        from detectron2.modeling.proposal_generator.proposal_utils import find_top_rpn_proposals
        from detectron2.modeling.proposal_generator.rpn import RPN, StandardRPNHead
        d = c["rpn"]
        Hf, Wf = d["hw"].tolist()
        gen = DefaultAnchorGenerator(sizes=[[32, 64, 128, 256, 512]], aspect_ratios=[[0.5, 1.0, 2.0]], strides=[32], offset=0.0)
        anchors = gen([torch.zeros(1, 1, Hf, Wf)])
        rpn = RPN(in_features=["vgg4"], head=StandardRPNHead(in_channels=8, num_anchors=15), anchor_generator=gen,
                  anchor_matcher=Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True),
                  box2box_transform=Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0)), batch_size_per_image=256,
                  positive_fraction=0.5, pre_nms_topk=(12000, 6000), post_nms_topk=(2000, 1000), nms_thresh=0.7,
                  min_box_size=0.0, smooth_l1_beta=0.0)
        losses = rpn.losses(anchors, [d["logits"]], list(d["labels"]), [d["deltas"]], list(d["matched_gt"]))
        out["rpn/loss_rpn_cls"], out["rpn/loss_rpn_loc"] = losses["loss_rpn_cls"].numpy(), losses["loss_rpn_loc"].numpy()
        rpn.train()
        sizes = [tuple(s) for s in d["image_sizes"].tolist()]
        props = rpn.predict_proposals(anchors, [d["logits"]], [d["deltas"]], sizes)
        for i, p in enumerate(props):
            out[f"rpn/proposals/{i}/boxes"] = p.proposal_boxes.tensor.numpy()
            out[f"rpn/proposals/{i}/logits"] = p.objectness_logits.numpy()
        del find_top_rpn_proposals

    @section("fast_rcnn")
    def _():
        from detectron2.modeling.roi_heads.fast_rcnn import FastRCNNOutputLayers
        d = c["fast_rcnn"]
        layer = FastRCNNOutputLayers(ShapeSpec(channels=1024), box2box_transform=Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0)),
                                     num_classes=8, test_score_thresh=0.05, test_nms_thresh=0.5, test_topk_per_image=100,
                                     smooth_l1_beta=0.0)
        n0, n1 = d["split"].tolist()
        sizes = [tuple(s) for s in d["image_sizes"].tolist()]
        plist = []
        for (a, b), sz in zip(((0, n0), (n0, n0 + n1)), sizes):
            inst = Instances(sz)
            inst.proposal_boxes = Boxes(d["proposals"][a:b])
            inst.gt_classes = d["gt_classes"][a:b]
            inst.gt_boxes = Boxes(d["gt_boxes"][a:b])
            plist.append(inst)
        losses = layer.losses((d["scores"], d["deltas"]), plist)
        out["fast_rcnn/loss_cls"], out["fast_rcnn/loss_box_reg"] = losses["loss_cls"].numpy(), losses["loss_box_reg"].numpy()
        dets, kept = layer.inference((d["scores"], d["deltas"]), plist)
        for i, (det, k) in enumerate(zip(dets, kept)):
            out[f"fast_rcnn/det/{i}/boxes"] = det.pred_boxes.tensor.numpy()
            out[f"fast_rcnn/det/{i}/scores"] = det.scores.numpy()
            out[f"fast_rcnn/det/{i}/classes"] = det.pred_classes.numpy()
            out[f"fast_rcnn/det/{i}/roi_idx"] = k.numpy()

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("recorded:", done)
    for n, e in failed:
        print("FAILED  :", n, e)
    print("->", OUT)
    return 0 if not failed else 1
