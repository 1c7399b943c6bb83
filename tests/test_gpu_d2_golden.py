"""The HIP path against the golden vectors of Detectron2's OWN unit tests (tests/helpers/d2_published.py), no oracle in
between: the product's ``RPN`` and ``StandardROIHeads`` modules, built from a config, loaded with the weights those tests'
seeds produce, run on ``cuda:0``.

  test_rpn        -> loss_rpn_cls / loss_rpn_loc, proposal boxes and objectness logits after clip + NMS(0.7)
  test_roi_heads  -> loss_cls / loss_box_reg of StandardROIHeads (ROIAlignV2 14 x 14 on a 1 x 2 map, 80 classes, gt appended)
  test_fast_rcnn  -> FastRCNNOutputLayers.losses on two boxes

Tolerances: fp32 mode 2e-5 relative on the losses (another summation order than torch's CPU convolution / linear), boxes 1e-4
pixels; the split-precision modes 2e-4 / 2e-3 pixels.
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import d2_published as D2

pytestmark = pytest.mark.gpu
DEV = "cuda"
HOT = os.path.join(ROOT, "configs", "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml")
TOL = {"fp32": (2e-5, 1e-4, 2e-6), "f16x3": (5e-5, 3e-4, 5e-6), "bf16x3": (2e-4, 2e-3, 3e-5)}    # losses rel, boxes px, logits abs


def _cfg(sfod, dtype, extra=()):
    return sfod.config.setup_cfg(HOT, ["SFOD.COMPUTE_DTYPE", dtype, "MODEL.DEVICE", DEV, "OUTPUT_DIR", "",
                                       "MODEL.RPN.IN_FEATURES", "['res4']", "MODEL.ROI_HEADS.IN_FEATURES", "['res4']",
                                       "MODEL.ANCHOR_GENERATOR.SIZES", "[[32, 64, 128, 256, 512]]",
                                       "MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS", "[[0.5, 1.0, 2.0]]",
                                       "MODEL.RPN.PRE_NMS_TOPK_TRAIN", "12000", "MODEL.RPN.POST_NMS_TOPK_TRAIN", "2000",
                                       "MODEL.RPN.LOSS_WEIGHT", "1.0", "MODEL.RPN.BBOX_REG_LOSS_WEIGHT", "1.0"] + list(extra))


def _instances(S, boxes, classes=None):
    out = []
    for i, b in enumerate(boxes):
        inst = S.Instances((15, 15))
        inst.gt_boxes = S.Boxes(b)
        inst.gt_classes = classes[i] if classes is not None else torch.zeros(len(b), dtype=torch.long)
        out.append(inst)
    return out


def _rpn(sfod, cfg, head):
    S = sfod.structures
    rpn = sfod.modeling.rpn.RPN(cfg, {"res4": S.ShapeSpec(channels=1024, stride=16)})
    rpn.rpn_head.load_state_dict(head)
    return rpn.to(DEV).train()


@pytest.mark.parametrize("dtype", ["fp32", "f16x3", "bf16x3"])
def test_hip_rpn_equals_detectron2s_test_rpn(sfod, native, dtype):
    c = D2.rpn_case()
    S = sfod.structures
    rpn = _rpn(sfod, _cfg(sfod, dtype), c["head"])
    images = S.ImageList(torch.zeros(2, 3, 20, 30), c["image_sizes"])
    with torch.no_grad():
        props, losses = rpn(images, {"res4": c["feat"].to(DEV)}, _instances(S, c["gt_boxes"]))
    rel, px, lg = TOL[dtype]
    np.testing.assert_allclose(losses["loss_rpn_cls"].item(), c["loss_rpn_cls"], rtol=rel)
    np.testing.assert_allclose(losses["loss_rpn_loc"].item(), c["loss_rpn_loc"], rtol=rel)
    assert len(props) == 2
    for p, eb, el in zip(props, c["proposal_boxes"], c["objectness_logits"]):
        assert len(p) == len(eb)
        np.testing.assert_allclose(p.proposal_boxes.tensor.cpu().numpy(), np.array(eb, dtype=np.float32), rtol=0, atol=px)
        np.testing.assert_allclose(p.objectness_logits.cpu().numpy(), np.array(el, dtype=np.float32), rtol=0, atol=lg)
    rpn.check_finite()


@pytest.mark.parametrize("dtype", ["fp32", "f16x3", "bf16x3"])
def test_hip_roi_heads_equal_detectron2s_test_roi_heads(sfod, native, dtype):
    c = D2.roi_heads_case()
    S = sfod.structures
    cfg = _cfg(sfod, dtype, ["MODEL.ROI_HEADS.NUM_CLASSES", "80", "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", "14",
                             "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", "512", "MODEL.ROI_HEADS.POSITIVE_FRACTION", "0.25",
                             "MODEL.ROI_HEADS.NAME", "StandardROIHeads", "MODEL.ROI_BOX_HEAD.FC_DIM", "1024"])
    rpn = _rpn(sfod, cfg, c["head"])
    heads = sfod.modeling.roi_heads.StandardROIHeads(cfg, {"res4": S.ShapeSpec(channels=1024, stride=16)})
    heads.box_head.load_state_dict({k: v for k, v in c["box"].items() if k.startswith("fc")})
    heads.box_predictor.load_state_dict({k: v for k, v in c["box"].items() if not k.startswith("fc")})
    heads = heads.to(DEV).train()
    images = S.ImageList(torch.zeros(2, 3, 20, 30), c["image_sizes"])
    gts = _instances(S, c["gt_boxes"], c["gt_classes"])
    feats = {"res4": c["feat"].to(DEV)}
    with torch.no_grad():
        props, _ = rpn(images, feats, gts, as_instances=False)
        samples, losses, _, _ = heads(images, feats, props, gts)
    assert samples["count"].tolist() == [4, 7]              # 2 / 5 proposals + the appended ground truth
    rel = TOL[dtype][0]
    np.testing.assert_allclose(losses["loss_cls"].item(), c["loss_cls"], rtol=rel)
    np.testing.assert_allclose(losses["loss_box_reg"].item(), c["loss_box_reg"], rtol=max(rel, 1e-4))


def test_hip_fast_rcnn_losses_equal_detectron2s_test_fast_rcnn(native):
    c = D2.fast_rcnn_case()
    K, R, ld = c["num_classes"], 2, 32
    pred = torch.zeros(R, ld)
    pred[:, :K + 1], pred[:, K + 1:5 * K + 1] = c["scores"], c["deltas"]
    rois = torch.cat([torch.zeros(R, 1), c["proposal_boxes"]], 1)
    nv = torch.tensor([R], dtype=torch.int32)
    loss, _ = native.frcnn_loss(pred.to(DEV), K, rois.to(DEV), c["gt_classes"].int().to(DEV), c["gt_boxes"].to(DEV), nv.to(DEV))
    np.testing.assert_allclose(loss[0].item(), c["loss_cls"], rtol=1e-6)
    np.testing.assert_allclose(loss[1].item(), c["loss_box_reg"], rtol=1e-6)
