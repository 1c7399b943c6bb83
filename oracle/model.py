"""Oracle (test infrastructure): CPU fp32 restatement of the teacher-student Faster R-CNN step.

Functional PyTorch-CPU model over a flat ``state`` dict that uses the REFERENCE's state-dict
key names, so the same weights load into this oracle and into the HIP product model.

Follows (reference file:line)
  backbone            daod/modeling/meta_arch/vgg.py:10-24 (layers), :70-74 (stages), :102-113 (init); or, for
                      configs/r101_c4_cs_foggy_adaptive_teacher_source_free.yaml:1-28 (``Cfg.r101_c4()``), Detectron2's
                      ResNet-C4 trunk restated in oracle/resnet.py (res4, stride 16, 1024 channels)
  RPN forward         daod/modeling/proposal_generator/rpn.py:16-58   (+ d2 RPN, Appendix A.4/A.7/A.10)
  ROI heads           daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py:68-215
  meta-arch branches  daod/modeling/meta_arch/source_free_adaptive_teacher_rcnn.py:212-225,259-339
  pseudo-labelling    daod/engine/trainers/source_free_adaptive_teacher.py:150-183,256-280
  step / EMA          daod/engine/trainers/source_free_adaptive_teacher.py:335-603
  SGD / LR            detectron2 build_optimizer / WarmupMultiStepLR
Gradients come from torch.autograd over these plain ops, i.e. they are independent of the
hand-written HIP backward they check.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import box_ops as B
from . import resnet
from .roi_align import roi_align

VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
STAGE_SLICES = [(0, 7), (7, 14), (14, 24), (24, 34), (34, 44)]  # vgg.py:70-74
PIXEL_MEAN = (103.530, 116.280, 123.675)
PIXEL_STD = (1.0, 1.0, 1.0)


class Cfg:
    """The handful of effective config values the path reads."""

    num_classes = 8
    anchor_sizes = (32, 64, 128, 256, 512)
    anchor_ratios = (0.5, 1.0, 2.0)
    stride = 32
    feat_channels = 512
    rpn_pre_topk_train = 12000
    rpn_post_topk_train = 2000
    rpn_pre_topk_test = 6000
    rpn_post_topk_test = 1000
    rpn_nms_thresh = 0.7
    rpn_batch = 256
    rpn_pos_frac = 0.5
    rpn_iou_thr = (0.3, 0.7)
    rpn_bbox_weights = (1.0, 1.0, 1.0, 1.0)
    roi_batch = 512
    roi_pos_frac = 0.25
    roi_iou_thr = (0.5,)
    roi_bbox_weights = (10.0, 10.0, 5.0, 5.0)
    pooler_res = 7
    fc_dim = 1024
    test_score_thresh = 0.05
    test_nms_thresh = 0.5
    test_dets = 100
    bbox_threshold = 0.8
    bn_momentum = 0.1
    bn_eps = 1e-5
    nms_numel_limit = 20000  # torchvision batched_nms strategy switch for GPU tensors
    backbone = "vgg"         # "vgg" (build_vgg_backbone, feature vgg4) | "resnet" (d2 build_resnet_backbone, feature res4)
    resnet_depth = 101
    freeze_at = 2

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(self, k):
                raise KeyError(k)
            setattr(self, k, v)

    @property
    def num_anchors(self):
        return len(self.anchor_sizes) * len(self.anchor_ratios)

    @classmethod
    def r101_c4(cls, **kw):
        """Effective values of configs/r101_c4_cs_foggy_adaptive_teacher_source_free.yaml (last
        column): no BACKBONE.NAME -> build_resnet_backbone, RESNETS.DEPTH 101 ``:5-7``, RPN / ROI heads on ``res4``
        (stride 16, 1024 channels), ANCHOR_GENERATOR.SIZES [[64, 128, 256, 512]] ``:19`` x 3 ratios = 12 anchors per
        location, RPN.BATCH_SIZE_PER_IMAGE 256 ``:23``, ROI_HEADS.BATCH_SIZE_PER_IMAGE 256 ``:11``, FastRCNNConvFCHead
        with NUM_FC 2 / FC_DIM 2048 ``:14-17``, ROIAlignV2 7x7 at scale 1/16."""
        vals = dict(backbone="resnet", resnet_depth=101, freeze_at=2, anchor_sizes=(64, 128, 256, 512), stride=16,
                    feat_channels=1024, rpn_batch=256, roi_batch=256, fc_dim=2048)
        vals.update(kw)
        return cls(**vals)


# ---------------------------------------------------------------------------------------------
# state
# ---------------------------------------------------------------------------------------------
def vgg_layer_names():
    """[(stage, local_idx, kind, cin, cout)] in execution order."""
    out = []
    cin = 3
    flat = []
    for v in VGG16:
        if v == "M":
            flat.append(("pool", None, None))
        else:
            flat += [("conv", cin, v), ("bn", v, v), ("relu", None, None)]
            cin = v
    for s, (a, b) in enumerate(STAGE_SLICES):
        for local, item in enumerate(flat[a:b]):
            out.append((s, local) + item)
    return out


def init_state(cfg, seed=0, with_dc=False):
    """Random init following the reference/d2 initialisers; returns OrderedDict name->tensor."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def normal(shape, std):
        return torch.randn(shape, generator=g) * std

    if cfg.backbone == "resnet":
        sd.update(resnet.init_state(cfg.resnet_depth, seed + 1000, cfg.freeze_at))
    for s, i, kind, cin, cout in (vgg_layer_names() if cfg.backbone == "vgg" else ()):
        p = f"backbone.vgg{s}.{i}"
        if kind == "conv":
            fan_out = cout * 9
            sd[p + ".weight"] = normal((cout, cin, 3, 3), math.sqrt(2.0 / fan_out))
            sd[p + ".bias"] = torch.zeros(cout)
        elif kind == "bn":
            sd[p + ".weight"] = torch.ones(cout)
            sd[p + ".bias"] = torch.zeros(cout)
            sd[p + ".running_mean"] = torch.zeros(cout)
            sd[p + ".running_var"] = torch.ones(cout)
            sd[p + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    C, A = cfg.feat_channels, cfg.num_anchors
    r = "proposal_generator.rpn_head."
    sd[r + "conv.weight"] = normal((C, C, 3, 3), 0.01)
    sd[r + "conv.bias"] = torch.zeros(C)
    sd[r + "objectness_logits.weight"] = normal((A, C, 1, 1), 0.01)
    sd[r + "objectness_logits.bias"] = torch.zeros(A)
    sd[r + "anchor_deltas.weight"] = normal((4 * A, C, 1, 1), 0.01)
    sd[r + "anchor_deltas.bias"] = torch.zeros(4 * A)
    fin = C * cfg.pooler_res ** 2

    def xavier(o, i):  # c2_xavier_fill = kaiming_uniform_(a=1)
        bound = math.sqrt(3.0 / i)
        return (torch.rand((o, i), generator=g) * 2 - 1) * bound

    h = "roi_heads.box_head."
    sd[h + "fc1.weight"] = xavier(cfg.fc_dim, fin)
    sd[h + "fc1.bias"] = torch.zeros(cfg.fc_dim)
    sd[h + "fc2.weight"] = xavier(cfg.fc_dim, cfg.fc_dim)
    sd[h + "fc2.bias"] = torch.zeros(cfg.fc_dim)
    q = "roi_heads.box_predictor."
    K = cfg.num_classes
    sd[q + "cls_score.weight"] = normal((K + 1, cfg.fc_dim), 0.01)
    sd[q + "cls_score.bias"] = torch.zeros(K + 1)
    sd[q + "bbox_pred.weight"] = normal((4 * K, cfg.fc_dim), 0.001)
    sd[q + "bbox_pred.bias"] = torch.zeros(4 * K)
    return sd


FROZEN_PREFIXES = ("backbone.stem.", "backbone.res2.")   # FREEZE_AT 2 of the ResNet-C4 config: FrozenBatchNorm2d buffers
                                                          # + requires_grad False conv weights (not optimised, no grad)


def is_param(name):
    """a trainable parameter (what d2's optimizer sees): not a buffer, not in a frozen ResNet stage"""
    return not (name.endswith("running_mean") or name.endswith("running_var")
                or name.endswith("num_batches_tracked") or name.startswith(FROZEN_PREFIXES))


def is_norm_param(name):
    """BN affine params (weight_decay_norm = 0 in d2's optimizer)."""
    if name.startswith("backbone.res"):
        return ".norm." in name and is_param(name)
    if not name.startswith("backbone.vgg"):
        return False
    idx = int(name.split(".")[2])
    return idx in (1, 4, 7) and is_param(name)


def clone_state(sd, requires_grad=False):
    out = OrderedDict()
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and is_param(k):
            t.requires_grad_(True)
        out[k] = t
    return out


# ---------------------------------------------------------------------------------------------
# forward pieces
# ---------------------------------------------------------------------------------------------
def preprocess(images_u8):
    """list of uint8 [3,H,W] (BGR) -> (padded fp32 [N,3,Hmax,Wmax], image_sizes).  A.1"""
    mean = torch.tensor(PIXEL_MEAN).view(3, 1, 1)
    std = torch.tensor(PIXEL_STD).view(3, 1, 1)
    sizes = [(int(im.shape[1]), int(im.shape[2])) for im in images_u8]
    hm = max(s[0] for s in sizes)
    wm = max(s[1] for s in sizes)
    out = torch.zeros(len(images_u8), 3, hm, wm)
    for i, im in enumerate(images_u8):
        out[i, :, : im.shape[1], : im.shape[2]] = (im.float() - mean) / std
    return out, sizes


def vgg_forward(sd, x, cfg, training=True, return_all=False):
    feats = {}
    for s, i, kind, cin, cout in vgg_layer_names():
        p = f"backbone.vgg{s}.{i}"
        if kind == "conv":
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif kind == "bn":
            if training:
                sd[p + ".num_batches_tracked"] += 1
            x = F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                             sd[p + ".bias"], training, cfg.bn_momentum, cfg.bn_eps)
        elif kind == "relu":
            x = F.relu(x)
        else:
            x = F.max_pool2d(x, 2, 2)
        feats[f"vgg{s}"] = x
    return feats if return_all else feats["vgg4"]


def backbone_forward(sd, x, cfg, training=True):
    """normalised image batch -> the feature map the heads read (``vgg4`` / ``res4``); train mode refreshes the live
    BatchNorm statistics and counters in ``sd`` (the AdaBN side effect, also under no_grad)."""
    if cfg.backbone == "resnet":
        return resnet.forward(sd, x, depth=cfg.resnet_depth, training=training, freeze_at=cfg.freeze_at)
    return vgg_forward(sd, x, cfg, training=training)


def rpn_head(sd, feat):
    """-> logits [N, H*W*A] and deltas [N, H*W*A, 4] in (y, x, a) order (rpn.py:28-41)."""
    r = "proposal_generator.rpn_head."
    t = F.relu(F.conv2d(feat, sd[r + "conv.weight"], sd[r + "conv.bias"], padding=1))
    obj = F.conv2d(t, sd[r + "objectness_logits.weight"], sd[r + "objectness_logits.bias"])
    dlt = F.conv2d(t, sd[r + "anchor_deltas.weight"], sd[r + "anchor_deltas.bias"])
    return rpn_flatten(obj, dlt)


def rpn_flatten(obj, dlt):
    """(N,A,H,W) -> (N,H*W*A) and (N,4A,H,W) -> (N,H*W*A,4) (rpn.py:28-41; pinned by tests/golden/glue_ref.npz)."""
    n, a, h, w = obj.shape
    logits = obj.permute(0, 2, 3, 1).flatten(1)
    deltas = dlt.view(n, a, 4, h, w).permute(0, 3, 4, 1, 2).flatten(1, -2)
    return logits, deltas


def anchors_for(feat_hw, cfg):
    cell = B.cell_anchors(cfg.anchor_sizes, cfg.anchor_ratios)
    return B.grid_anchors(feat_hw[0], feat_hw[1], cfg.stride, cell)


def rpn_proposals(anchors, logits, deltas, image_sizes, cfg, training=True):
    """find_top_rpn_proposals, single level.  A.7.  -> list of (boxes [n,4], logits [n])."""
    pre = cfg.rpn_pre_topk_train if training else cfg.rpn_pre_topk_test
    post = cfg.rpn_post_topk_train if training else cfg.rpn_post_topk_test
    out = []
    with torch.no_grad():
        for n in range(logits.shape[0]):
            props = B.apply_deltas(deltas[n], anchors, cfg.rpn_bbox_weights)
            k = min(pre, logits.shape[1])
            order = torch.sort(logits[n], descending=True, stable=True)[1][:k]
            boxes = props[order]
            scores = logits[n][order]
            valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
            if not valid.all():
                if training:
                    raise FloatingPointError(
                        "Predicted boxes or scores contain Inf/NaN. Training has diverged.")
                boxes, scores = boxes[valid], scores[valid]
            boxes = B.clip_boxes(boxes, image_sizes[n])
            keep = B.nonempty(boxes, 0.0)
            boxes, scores = boxes[keep], scores[keep]
            keep = B.batched_nms(boxes, scores, torch.zeros(len(boxes), dtype=torch.int64),
                                 cfg.rpn_nms_thresh, cfg.nms_numel_limit)[:post]
            out.append((boxes[keep], scores[keep]))
    return out


def rpn_label_anchors(anchors, gt_boxes_list, keys_list, cfg):
    """label_and_sample_anchors.  A.10.  -> labels int8 [N, A], matched gt boxes [N, A, 4]."""
    labels, matched = [], []
    for gt, keys in zip(gt_boxes_list, keys_list):
        M = B.pairwise_iou(gt, anchors)
        idx, lab = B.matcher(M, cfg.rpn_iou_thr, [0, -1, 1], True)
        pos, neg = B.subsample_labels(lab, cfg.rpn_batch, cfg.rpn_pos_frac, 0, keys)
        lab = torch.full_like(lab, -1)
        lab[pos] = 1
        lab[neg] = 0
        labels.append(lab)
        matched.append(gt[idx] if len(gt) > 0 else torch.zeros_like(anchors))
    return torch.stack(labels), torch.stack(matched)


def rpn_losses(anchors, logits, deltas, labels, matched_gt, cfg):
    n = logits.shape[0]
    pos = labels == 1
    gt_deltas = torch.stack([B.get_deltas(anchors, m, cfg.rpn_bbox_weights) for m in matched_gt])
    loc = (deltas[pos] - gt_deltas[pos]).abs().sum()
    valid = labels >= 0
    obj = F.binary_cross_entropy_with_logits(logits[valid], labels[valid].float(), reduction="sum")
    norm = cfg.rpn_batch * n
    return {"loss_rpn_cls": obj / norm, "loss_rpn_loc": loc / norm}


def roi_label_and_sample(proposals, gt_boxes_list, gt_classes_list, keys_list, cfg):
    """label_and_sample_proposals (roi_heads.py:165-215) with proposal_append_gt=True.

    -> per image dict(boxes, gt_classes, gt_boxes, sampled_idxs)."""
    out = []
    K = cfg.num_classes
    for (boxes, _), gtb, gtc, keys in zip(proposals, gt_boxes_list, gt_classes_list, keys_list):
        if len(gtb) > 0:
            boxes = torch.cat([boxes, gtb])
        M = B.pairwise_iou(gtb, boxes)
        midx, mlab = B.matcher(M, cfg.roi_iou_thr, [0, 1], False)
        if len(gtc) > 0:
            cls = gtc[midx].clone()
            cls[mlab == 0] = K
            cls[mlab == -1] = -1
        else:
            cls = torch.zeros_like(midx) + K
        fg, bg = B.subsample_labels(cls, cfg.roi_batch, cfg.roi_pos_frac, K, keys[: len(cls)])
        sidx = torch.cat([fg, bg])
        out.append({
            "boxes": boxes[sidx],
            "gt_classes": cls[sidx],
            "gt_boxes": gtb[midx[sidx]] if len(gtb) > 0 else torch.zeros(len(sidx), 4),
            "sampled_idxs": sidx,
        })
    return out


def box_head(sd, feat, boxes_per_image, cfg):
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b], dim=1)
                      for i, b in enumerate(boxes_per_image)])
    pooled = roi_align(feat, rois, cfg.pooler_res, 1.0 / cfg.stride, 0, True)
    h = "roi_heads.box_head."
    x = pooled.flatten(1)
    x = F.relu(F.linear(x, sd[h + "fc1.weight"], sd[h + "fc1.bias"]))
    x = F.relu(F.linear(x, sd[h + "fc2.weight"], sd[h + "fc2.bias"]))
    q = "roi_heads.box_predictor."
    scores = F.linear(x, sd[q + "cls_score.weight"], sd[q + "cls_score.bias"])
    deltas = F.linear(x, sd[q + "bbox_pred.weight"], sd[q + "bbox_pred.bias"])
    return scores, deltas, pooled


def fast_rcnn_losses(scores, deltas, boxes, gt_classes, gt_boxes, cfg):
    K = cfg.num_classes
    if len(gt_classes) == 0:
        return {"loss_cls": scores.sum() * 0.0, "loss_box_reg": deltas.sum() * 0.0}
    loss_cls = F.cross_entropy(scores, gt_classes, reduction="mean")
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < K)).squeeze(1)
    fg_deltas = deltas.view(-1, K, 4)[fg, gt_classes[fg]]
    tgt = B.get_deltas(boxes[fg], gt_boxes[fg], cfg.roi_bbox_weights)
    loss_box = (fg_deltas - tgt).abs().sum() / max(gt_classes.numel(), 1.0)
    return {"loss_cls": loss_cls, "loss_box_reg": loss_box}


def fast_rcnn_inference(scores, deltas, boxes_per_image, image_sizes, cfg):
    """d2 fast_rcnn_inference (A.13) -> list of dict(boxes, scores, classes, roi_idx)."""
    K = cfg.num_classes
    out = []
    off = 0
    with torch.no_grad():
        for pb, size in zip(boxes_per_image, image_sizes):
            n = len(pb)
            sc = F.softmax(scores[off:off + n], dim=-1)
            bx = B.apply_deltas(deltas[off:off + n], pb, cfg.roi_bbox_weights)
            off += n
            valid = torch.isfinite(bx).all(dim=1) & torch.isfinite(sc).all(dim=1)
            roi_ids = torch.arange(n)
            if not valid.all():
                bx, sc, roi_ids = bx[valid], sc[valid], roi_ids[valid]
            sc = sc[:, :-1]
            bx = B.clip_boxes(bx.reshape(-1, 4), size).view(-1, K, 4)
            mask = sc > cfg.test_score_thresh
            inds = mask.nonzero()
            cb = bx[mask]
            cs = sc[mask]
            keep = B.batched_nms(cb, cs, inds[:, 1], cfg.test_nms_thresh, cfg.nms_numel_limit)
            keep = keep[: cfg.test_dets]
            out.append({"boxes": cb[keep], "scores": cs[keep], "classes": inds[keep, 1],
                        "roi_idx": roi_ids[inds[keep, 0]]})
    return out


def threshold_bbox(det, thr):
    """threshold_bbox 'roih' (source_free_adaptive_teacher.py:167-181): strict '>'."""
    m = det["scores"] > thr
    return {"gt_boxes": det["boxes"][m], "gt_classes": det["classes"][m], "scores": det["scores"][m]}


def predict_boxes_for_gt_classes(deltas, boxes, gt_classes, cfg):
    """d2 ``FastRCNNOutputLayers.predict_boxes_for_gt_classes`` (A.12): decode every row with the deltas of
    its own gt class, background clamped to K-1; not clipped.  The reference overwrites the sampled proposals'
    boxes with the result (source_free_adaptive_teacher_roi_heads.py:136-143)."""
    K = cfg.num_classes
    pb = B.apply_deltas(deltas, boxes, cfg.roi_bbox_weights).view(-1, K, 4)
    return pb[torch.arange(len(boxes)), gt_classes.clamp(0, K - 1)]


def convert_bbox_scores(scores, deltas, boxes_per_image, image_sizes, cfg):
    """``SourceFreeFastRCNNOutputLayers.convert_bbox_scores`` (source_free_fast_rcnn.py:15-36,82-147): softmax,
    per-class decode, non-finite rows dropped, background column dropped, clip, keep ``score > 0``, NO NMS;
    row-major (roi, class) order.  -> list of dict(boxes, scores, classes, roi_idx)."""
    K = cfg.num_classes
    out, off = [], 0
    with torch.no_grad():
        for pb, size in zip(boxes_per_image, image_sizes):
            n = len(pb)
            sc = F.softmax(scores[off:off + n], dim=-1)
            bx = B.apply_deltas(deltas[off:off + n], pb, cfg.roi_bbox_weights)
            off += n
            out.append(frcnn_inference_new_single(bx, sc, size))
    return out


def frcnn_inference_new_single(bx, sc, size):
    """``fast_rcnn_inference_single_image_new`` (source_free_fast_rcnn.py:82-147; pinned by tests/golden/glue_ref.npz):
    per-class boxes [R, 4K] (or class-agnostic [R, 4]) and probabilities [R, K+1] -> every (row, class) with score > 0."""
    valid = torch.isfinite(bx).all(dim=1) & torch.isfinite(sc).all(dim=1)
    if not valid.all():
        bx, sc = bx[valid], sc[valid]
    sc = sc[:, :-1]
    nreg = bx.shape[1] // 4
    bx = B.clip_boxes(bx.reshape(-1, 4), size).view(-1, nreg, 4)
    mask = sc > 0
    inds = mask.nonzero()
    # roi_idx: ``filter_inds[:, 0]`` (:147) -- an index into the rows that survived the finite filter
    boxes = bx[inds[:, 0], 0] if nreg == 1 else bx[mask]
    return {"boxes": boxes, "scores": sc[mask], "classes": inds[:, 1], "roi_idx": inds[:, 0]}


def bpc_loss(class_number, gts, dets, iou_thresh=0.5):
    """``bpc_loss`` (daod/loss/bpc_loss.py:10-262), vectorised.  ``gts``: list of (boxes [G,4], classes [G]);
    ``dets``: list of dict(boxes, scores, classes).  Per class (``evaluate_output`` :137-196): no ground truth
    of the class -> every detection of it is a false positive; otherwise (``count_confusions`` :87-134, IoU with
    the legacy +1 extents :62-85) a detection is a true positive when its best IoU over the class's ground truth
    is > 0.5 -- once per ground-truth box attaining that maximum (``np.where(mask)[1]`` repeats the column on
    ties) -- else a false positive.  ``loss_forward`` :200-262: AC / AN over true positives with score >= / < 0.5,
    IC / IN over false positives, per image log(1 + (AN + IC) / (AC + IN)) when the denominator is > 0, mean."""
    per_image = []
    for (gb, gc), d in zip(gts, dets):
        db, ds, dc = d["boxes"], d["scores"], d["classes"]
        nAC = nAN = nIC = nIN = torch.zeros(())
        for k in range(class_number):
            gk, dk = gc == k, dc == k
            s = ds[dk]
            if int(gk.sum()) == 0:
                tp_s, fp_s = torch.zeros(1), s
            elif int(dk.sum()) > 0:
                e, o = gb[gk], db[dk]
                ea = (e[:, 2] - e[:, 0] + 1) * (e[:, 3] - e[:, 1] + 1)
                oa = (o[:, 2] - o[:, 0] + 1) * (o[:, 3] - o[:, 1] + 1)
                w = (torch.min(e[:, None, 2], o[None, :, 2]) - torch.max(e[:, None, 0], o[None, :, 0]) + 1).clamp(min=0)
                h = (torch.min(e[:, None, 3], o[None, :, 3]) - torch.max(e[:, None, 1], o[None, :, 1]) + 1).clamp(min=0)
                inter = w * h
                ious = inter / (ea[:, None] + oa[None, :] - inter)
                tpm = (ious > iou_thresh) & (ious == ious.max(dim=0, keepdim=True).values)
                tp_cols = tpm.nonzero()[:, 1]                 # row-major: a column appears once per tied row
                fpm = torch.ones(len(o), dtype=torch.bool)
                fpm[tp_cols] = False
                tp_s, fp_s = s[tp_cols], s[fpm]
            else:
                continue
            t = torch.tanh
            nAC = nAC + (tp_s[tp_s >= 0.5] * t(tp_s[tp_s >= 0.5])).sum()
            nAN = nAN + (tp_s[tp_s < 0.5] * (1 - t(tp_s[tp_s < 0.5]))).sum()
            nIC = nIC + ((1 - fp_s[fp_s >= 0.5]) * t(fp_s[fp_s >= 0.5])).sum()
            nIN = nIN + ((1 - fp_s[fp_s < 0.5]) * (1 - t(fp_s[fp_s < 0.5]))).sum()
        numr, denom = nAN + nIC, nAC + nIN
        if denom > 0.0:
            per_image.append(torch.log(1 + numr / denom))
    return torch.stack(per_image).mean() if per_image else torch.tensor(0.0)


class AdaptiveThreshold:
    """Class-wise adaptive confidence threshold: ``AdaptiveConfidenceBasedSelfTrainingLoss``
    (adaptive_thresh/adaptive_confidence.py:6-34, the "convex" curve) + the trainer's bookkeeping
    (source_free_adaptive_teacher.py:116-120 state, :282-295 ``count_label_prediction``, :297-309
    ``update_adaptive_threshold``, :393-404 per-step update, :461-466 selection after WARM_UP)."""

    def __init__(self, threshold, num_classes, reserve):
        self.threshold, self.num_classes = threshold, num_classes
        self.classwise_acc = torch.ones(num_classes)
        self.reserve_matrix = torch.zeros(reserve, num_classes)

    def mask(self, confidence, pseudo_labels):
        a = self.classwise_acc[pseudo_labels]
        return confidence >= self.threshold * (a / (2. - a))

    def update(self, dets, it, fixed_thr):
        """:393-400: detections passing the CURRENT mask ("prediction_thresholding", :230-254), of which
        those with ``score > BBOX_THRESHOLD`` are counted per class into row ``it % RESERVE``; then :297-309."""
        K = self.num_classes
        count = torch.zeros(K)
        for d in dets:
            m = self.mask(d["scores"], d["classes"])
            sc, cl = d["scores"][m], d["classes"][m]
            count += cl[sc > fixed_thr].bincount(minlength=K)
        self.reserve_matrix[it % len(self.reserve_matrix)] = count
        counter = self.reserve_matrix.sum(dim=0)
        counter[0] = 0
        counter[2] = 0
        self.classwise_acc = counter / max(counter.max(), 1)
        self.classwise_acc[0] = 1
        self.classwise_acc[2] = 1

    def select(self, det):
        """``adaptive_threshold_bbox`` 'roih' (:204-226)."""
        idx = torch.nonzero(self.mask(det["scores"], det["classes"])).flatten()
        return {"gt_boxes": det["boxes"][idx], "gt_classes": det["classes"][idx], "scores": det["scores"][idx]}


# ---------------------------------------------------------------------------------------------
# branches
# ---------------------------------------------------------------------------------------------
def teacher_forward(sd, images_u8, cfg):
    """branch='unsup_data_weak' on the train-mode teacher under no_grad (trainer :385-390)."""
    with torch.no_grad():
        x, sizes = preprocess(images_u8)
        feat = backbone_forward(sd, x, cfg, training=True)
        logits, deltas = rpn_head(sd, feat)
        anchors = anchors_for(feat.shape[-2:], cfg)
        props = rpn_proposals(anchors, logits, deltas, sizes, cfg, training=True)
        scores, bdeltas, _ = box_head(sd, feat, [p[0] for p in props], cfg)
        dets = fast_rcnn_inference(scores, bdeltas, [p[0] for p in props], sizes, cfg)
    return props, dets


def eval_inference(sd, images_u8, cfg, out_sizes=None):
    """d2 ``GeneralizedRCNN.inference`` in eval mode (BN on running statistics, TEST top-k 6000 / 1000) +
    ``detector_postprocess``: rescale the boxes to ``out_sizes`` [(h, w)], clip, drop empty (reached from
    ``source_free_adaptive_teacher_rcnn.py:129-130`` and ``DefaultTrainer.test``)."""
    with torch.no_grad():
        x, sizes = preprocess(images_u8)
        feat = backbone_forward(sd, x, cfg, training=False)
        logits, deltas = rpn_head(sd, feat)
        anchors = anchors_for(feat.shape[-2:], cfg)
        props = rpn_proposals(anchors, logits, deltas, sizes, cfg, training=False)
        scores, bdeltas, _ = box_head(sd, feat, [p[0] for p in props], cfg)
        dets = fast_rcnn_inference(scores, bdeltas, [p[0] for p in props], sizes, cfg)
        out = []
        for n, det in enumerate(dets):
            oh, ow = out_sizes[n] if out_sizes is not None else sizes[n]
            sx, sy = ow / sizes[n][1], oh / sizes[n][0]
            bx = det["boxes"] * torch.tensor([sx, sy, sx, sy])
            bx = B.clip_boxes(bx, (oh, ow))
            keep = B.nonempty(bx, 0.0)
            out.append({"boxes": bx[keep], "scores": det["scores"][keep], "classes": det["classes"][keep]})
    return out


def student_losses(sd, images_u8, gt_boxes_list, gt_classes_list, rpn_keys, roi_keys, cfg,
                   return_aux=False, proposals=None):
    """branch='supervised_target' (rcnn.py:259-312) without the dead 2nd ROI pass; ``loss_bpc`` included.

    ``proposals`` (list of (boxes, logits)) replaces the RPN's own proposals: parity tests use it
    to give both implementations the same discrete proposal set, because a 1e-7 difference in a
    logit can legitimately flip an NMS decision (``given_proposals`` in the reference's signature)."""
    x, sizes = preprocess(images_u8)
    feat = backbone_forward(sd, x, cfg, training=True)
    logits, deltas = rpn_head(sd, feat)
    anchors = anchors_for(feat.shape[-2:], cfg)
    labels, matched = rpn_label_anchors(anchors, gt_boxes_list, rpn_keys, cfg)
    losses = rpn_losses(anchors, logits, deltas, labels, matched, cfg)
    props = rpn_proposals(anchors, logits.detach(), deltas.detach(), sizes, cfg, training=True)
    own_props = props
    if proposals is not None:
        props = proposals
    samp = roi_label_and_sample(props, gt_boxes_list, gt_classes_list, roi_keys, cfg)
    scores, bdeltas, _ = box_head(sd, feat, [s["boxes"] for s in samp], cfg)
    losses.update(fast_rcnn_losses(
        scores, bdeltas, torch.cat([s["boxes"] for s in samp]),
        torch.cat([s["gt_classes"] for s in samp]), torch.cat([s["gt_boxes"] for s in samp]), cfg))
    # rcnn.py:293: BPC of the training pass's per-class predictions (convert_bbox_scores on the sampled proposals
    # AFTER their boxes were overwritten by predict_boxes_for_gt_classes, roi_heads.py:136-158) against the
    # (pseudo) ground truth; logged, weighted by 0, no gradient
    with torch.no_grad():
        nb = predict_boxes_for_gt_classes(bdeltas, torch.cat([s["boxes"] for s in samp]),
                                          torch.cat([s["gt_classes"] for s in samp]), cfg)
        nb = list(nb.split([len(s["boxes"]) for s in samp]))
        inst = convert_bbox_scores(scores, bdeltas, nb, sizes, cfg)
        losses["loss_bpc"] = bpc_loss(cfg.num_classes, list(zip(gt_boxes_list, gt_classes_list)), inst)
    if return_aux:
        return losses, {"feat": feat, "logits": logits, "deltas": deltas, "labels": labels, "inst": inst,
                        "props": props, "own_props": own_props, "samp": samp, "scores": scores, "bdeltas": bdeltas}
    return losses


# ---------------------------------------------------------------------------------------------
# optimiser / EMA / schedule
# ---------------------------------------------------------------------------------------------
def lr_at(it, base_lr, steps=(60000, 80000, 90000), gamma=0.1, warmup_iters=1000,
          warmup_factor=0.001):
    """d2 WarmupMultiStepLR value used for iteration ``it`` (A.15)."""
    mult = gamma ** sum(1 for s in steps if it >= s)
    if it < warmup_iters:
        alpha = it / warmup_iters
        mult *= warmup_factor * (1 - alpha) + alpha
    return base_lr * mult


def sgd_step(sd, grads, bufs, lr, momentum=0.9, weight_decay=1e-4, weight_decay_norm=0.0):
    """torch.optim.SGD step with d2's per-group weight decay; grads: name->tensor or None."""
    with torch.no_grad():
        for name, p in sd.items():
            if not is_param(name) or grads.get(name) is None:
                continue
            wd = weight_decay_norm if is_norm_param(name) else weight_decay
            g = grads[name]
            if wd != 0:
                g = g + wd * p
            if name not in bufs:
                bufs[name] = g.clone()
            else:
                bufs[name].mul_(momentum).add_(g)
            p.add_(bufs[name], alpha=-lr)


def ema_update(teacher, student, keep_rate=0.9996):
    """_update_teacher_model (:583-603): every key of the teacher state dict, buffers included."""
    with torch.no_grad():
        for k, v in teacher.items():
            if k not in student:
                raise Exception("{} is not found in student model".format(k))
            new = student[k] * (1 - keep_rate) + v * keep_rate
            v.copy_(new)  # load_state_dict copy: int64 buffers truncate
