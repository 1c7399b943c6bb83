"""Golden vectors of Detectron2's OWN unit tests, rebuilt here from their recipes.

Detectron2 is not installed in the build image and is not part of /root/reference, so nothing can be imported or run; but
three of its unit tests are fully specified by a torch seed, a construction order and literal inputs, and they assert
literal outputs:

  tests/modeling/test_rpn.py::RPNTest::test_rpn                     (v0.1 - v0.2 values)
  tests/modeling/test_roi_heads.py::ROIHeadsTest::test_roi_heads    (v0.1 - v0.2 values)
  tests/modeling/test_fast_rcnn.py::FastRCNNTest::test_fast_rcnn

torch's CPU generator stream (mt19937; uniform_ / normal_ / rand) has not changed since, so building THE SAME MODULES IN THE
SAME ORDER under the same seed reproduces the weights and inputs those tests saw -- only the amount of randomness the
ResNet-50-C4 backbone's constructor draws matters (it is never run: the tests feed ``torch.rand`` features), and that is
reproduced by constructing convolutions of the same shapes in build_resnet_backbone's order.  The expected values below are
quoted from those test files; that the recipes hit them to 7 digits (tests/test_oracle_d2_golden.py) is what shows both the
quotes and the recipes are right -- a wrong digit or a wrong construction order lands nowhere near.

(Later Detectron2 releases re-recorded test_rpn / test_roi_heads with other literals -- 0.08011703193 / 0.101470276 and
4.5253729820 / 0.0097857201 -- after a change in what their set-up draws from the generator; those were not reproduced here
and are not used.  The v0.1 - v0.2 literals are: every one of the 2 + 2 losses, 7 boxes and 7 logits is hit.)

These pin, against Detectron2 itself: DefaultAnchorGenerator, Matcher (+ low-quality matches), Box2BoxTransform (both
weightings), the RPN losses and their normaliser, find_top_rpn_proposals (decode, clip, non-empty, NMS 0.7, top-k),
ROIAlignV2, proposal_append_gt + label_and_sample_proposals, the box head, FastRCNNOutputLayers.losses.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

# ---- expected values, as asserted by the Detectron2 tests ---------------------------------------------------------------
RPN_EXPECTED = {
    "loss_rpn_cls": 0.0804563984, "loss_rpn_loc": 0.0990132466,
    "image_sizes": [(10, 10), (20, 30)],
    "proposal_boxes": [
        [[0, 0, 10, 10], [7.3365392685, 0, 10, 10]],
        [[0, 0, 30, 20], [0, 0, 16.7862777710, 13.1362524033], [0, 0, 30, 13.3173446655], [0, 0, 10.8602609634, 20],
         [7.7165775299, 0, 27.3875980377, 20]]],
    "objectness_logits": [[0.1225359365, -0.0133192837],
                          [0.1415634006, 0.0989848152, 0.0565387346, -0.0072308783, -0.0428492837]],
}
ROI_HEADS_EXPECTED = {"loss_cls": 4.4236516953, "loss_box_reg": 0.0091214813}
FAST_RCNN_EXPECTED = {"loss_cls": 1.7951188087, "loss_box_reg": 4.0357131958}


def _msra(m):      # fvcore c2_msra_fill (no bias in the backbone's convolutions)
    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


def draw_resnet50_c4_backbone():
    """What ``build_backbone(get_cfg())`` draws from the generator: BasicStem, then res2..res4 (3 / 4 / 6 BottleneckBlocks;
    per block: shortcut (when the width changes), conv1, conv2, conv3 constructed -- nn.Conv2d's default init draws -- then
    c2_msra_fill of conv1, conv2, conv3, shortcut in that order).  FrozenBN draws nothing."""
    stem = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
    _msra(stem)
    cin, bott, cout = 64, 64, 256
    for blocks in (3, 4, 6):
        for _ in range(blocks):
            sc = nn.Conv2d(cin, cout, 1, bias=False) if cin != cout else None
            c1 = nn.Conv2d(cin, bott, 1, bias=False)
            c2 = nn.Conv2d(bott, bott, 3, padding=1, bias=False)
            c3 = nn.Conv2d(bott, cout, 1, bias=False)
            for layer in (c1, c2, c3, sc):
                if layer is not None:
                    _msra(layer)
            cin = cout
        cout, bott = cout * 2, bott * 2


def draw_rpn_head(C=1024, A=15):
    """StandardRPNHead: conv 3x3, objectness 1x1, deltas 1x1 constructed, then normal_(0.01) / zero bias each."""
    conv, obj, dl = nn.Conv2d(C, C, 3, padding=1), nn.Conv2d(C, A, 1), nn.Conv2d(C, 4 * A, 1)
    for layer in (conv, obj, dl):
        nn.init.normal_(layer.weight, std=0.01)
        nn.init.constant_(layer.bias, 0)
    return {"conv.weight": conv.weight.detach(), "conv.bias": conv.bias.detach(),
            "objectness_logits.weight": obj.weight.detach(), "objectness_logits.bias": obj.bias.detach(),
            "anchor_deltas.weight": dl.weight.detach(), "anchor_deltas.bias": dl.bias.detach()}


def rpn_case():
    """test_rpn: seed 121, backbone, RPN, then the data.  -> dict(head weights, features [2,1024,1,2], gt boxes per image)."""
    torch.manual_seed(121)
    draw_resnet50_c4_backbone()
    head = draw_rpn_head()
    torch.rand(2, 20, 30)                                    # images_tensor
    feat = torch.rand(2, 1024, 1, 2)
    gt = [torch.tensor([[1.0, 1, 3, 3]]), torch.tensor([[2.0, 2, 6, 6]])]      # gt_instances[0], gt_instances[1]
    return {"head": head, "feat": feat, "gt_boxes": gt, "stride": 16, **RPN_EXPECTED}


def roi_heads_case():
    """test_roi_heads: seed 121, backbone, the data, RPN, StandardROIHeads (FastRCNNConvFCHead with 2 FC, ROIAlignV2 14 x 14,
    80 classes, box weights (10, 10, 5, 5))."""
    torch.manual_seed(121)
    draw_resnet50_c4_backbone()
    torch.rand(2, 20, 30)
    feat = torch.rand(2, 1024, 1, 2)
    head = draw_rpn_head()
    P, K = 14, 80
    fc1, fc2 = nn.Linear(1024 * P * P, 1024), nn.Linear(1024, 1024)
    for layer in (fc1, fc2):                                 # c2_xavier_fill
        nn.init.kaiming_uniform_(layer.weight, a=1)
        nn.init.constant_(layer.bias, 0)
    cls, bbox = nn.Linear(1024, K + 1), nn.Linear(1024, 4 * K)
    nn.init.normal_(cls.weight, std=0.01)
    nn.init.normal_(bbox.weight, std=0.001)
    for layer in (cls, bbox):
        nn.init.constant_(layer.bias, 0)
    box = {"fc1.weight": fc1.weight.detach(), "fc1.bias": fc1.bias.detach(), "fc2.weight": fc2.weight.detach(),
           "fc2.bias": fc2.bias.detach(), "cls_score.weight": cls.weight.detach(), "cls_score.bias": cls.bias.detach(),
           "bbox_pred.weight": bbox.weight.detach(), "bbox_pred.bias": bbox.bias.detach()}
    gtb = [torch.tensor([[1.0, 1, 3, 3], [2, 2, 6, 6]]), torch.tensor([[1.0, 5, 2, 8], [7, 3, 10, 5]])]
    gtc = [torch.tensor([2, 1]), torch.tensor([1, 2])]
    return {"head": head, "box": box, "feat": feat, "gt_boxes": gtb, "gt_classes": gtc, "stride": 16, "pooler": P,
            "num_classes": K, "image_sizes": [(10, 10), (20, 30)], **ROI_HEADS_EXPECTED}


def fast_rcnn_case():
    """test_fast_rcnn: seed 132, FastRCNNOutputLayers(8 -> 5 classes, box weights (10, 10, 5, 5)), torch.rand(2, 8) features."""
    torch.manual_seed(132)
    K = 5
    cls, bbox = nn.Linear(8, K + 1), nn.Linear(8, 4 * K)
    nn.init.normal_(cls.weight, std=0.01)
    nn.init.normal_(bbox.weight, std=0.001)
    for layer in (cls, bbox):
        nn.init.constant_(layer.bias, 0)
    feat = torch.rand(2, 8)
    with torch.no_grad():
        scores, deltas = cls(feat), bbox(feat)
    return {"scores": scores, "deltas": deltas, "num_classes": K,
            "proposal_boxes": torch.tensor([[0.8, 1.1, 3.2, 2.8], [2.3, 2.5, 7, 8]]),
            "gt_boxes": torch.tensor([[1.0, 1, 3, 3], [2, 2, 6, 6]]), "gt_classes": torch.tensor([1, 2]),
            **FAST_RCNN_EXPECTED}


def rpn_head_outputs(head, feat):
    """the head on the CPU (plain torch): -> objectness [N, A, H, W], deltas [N, 4A, H, W]"""
    with torch.no_grad():
        t = F.relu(F.conv2d(feat, head["conv.weight"], head["conv.bias"], padding=1))
        return (F.conv2d(t, head["objectness_logits.weight"], head["objectness_logits.bias"]),
                F.conv2d(t, head["anchor_deltas.weight"], head["anchor_deltas.bias"]))
