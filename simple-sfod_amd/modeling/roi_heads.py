"""ROI heads on the HIP kernels behind ``StandardROIHeads`` /
``SourceFreeAdaptiveTeacherStandardROIHeads`` / ``AdaptiveTeacherStandardROIHeads``.

Mirrors ``/root/reference/daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py``:
``_init_box_head`` ``:27-66``, ``forward`` ``:68-106`` (training+compute_loss -> label & sample,
4-tuple; otherwise inference, 2-tuple), ``_forward_box`` ``:108-163``,
``label_and_sample_proposals`` ``:165-215``; and the Detectron2 pieces it inherits (SURVEY.md
A.11-A.13): add_ground_truth_to_proposals, Matcher [0.5], subsample 512 @ 0.25, ROIAlignV2 7x7,
FastRCNNConvFCHead (2 FC), FastRCNNOutputLayers losses / inference.

State-dict keys: ``roi_heads.box_head.fc{1,2}.*``, ``roi_heads.box_predictor.{cls_score,bbox_pred}.*``.
"""
import math

import torch
import torch.nn as nn

from .. import native
from ..registry import ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY
from ..structures import Boxes, Instances, ShapeSpec
from .offchain import OffChain as _OffChain
from .batched import BatchedDetections, BatchedGT, BatchedProposals


class ROIPooler(nn.Module):
    """Single-level ROIAlignV2 pooler (aligned=True).  Attributes read from outside
    (source_free_adaptive_teacher_rcnn.py:190-194) are kept."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type):
        super().__init__()
        assert pooler_type == "ROIAlignV2" and sampling_ratio == 0 and len(scales) == 1
        self.output_size = (output_size, output_size)
        self.scale = scales[0]
        min_level = -(math.log2(scales[0]))
        self.min_level = self.max_level = int(min_level)
        self.canonical_level = 4
        self.canonical_box_size = 224


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Module):
    """NUM_CONV 0 + NUM_FC fully connected layers with ReLU (c2_xavier_fill)."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        assert cfg.MODEL.ROI_BOX_HEAD.NUM_CONV == 0, "conv layers in the box head are not on the hot path"
        num_fc, fc_dim = cfg.MODEL.ROI_BOX_HEAD.NUM_FC, cfg.MODEL.ROI_BOX_HEAD.FC_DIM
        assert num_fc == 2, "the named configs use NUM_FC 2"
        self._in = input_shape.channels * input_shape.height * input_shape.width
        self.fcs = []
        d = self._in
        for k in range(num_fc):
            fc = nn.Linear(d, fc_dim)
            nn.init.kaiming_uniform_(fc.weight, a=1)
            nn.init.constant_(fc.bias, 0)
            self.add_module("fc{}".format(k + 1), fc)
            self.fcs.append(fc)
            d = fc_dim
        self._out = d

    @property
    def output_shape(self):
        return ShapeSpec(channels=self._out)


class FastRCNNOutputLayers(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        d = input_shape.channels
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        assert not cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG
        self.cls_score = nn.Linear(d, self.num_classes + 1)
        self.bbox_pred = nn.Linear(d, self.num_classes * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)
        assert tuple(cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS) == (10.0, 10.0, 5.0, 5.0)
        assert cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA == 0.0
        self.test_score_thresh = cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST
        self.test_nms_thresh = cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE


_SCALE_CLAMP = math.log(1000.0 / 16)


def _apply_deltas(deltas, boxes, weights=(10.0, 10.0, 5.0, 5.0)):
    """d2 Box2BoxTransform.apply_deltas (A.5) in torch ops: the Instances-level API twins below use it; the
    step itself decodes inside the HIP kernels (detect.hip)."""
    deltas, boxes = deltas.float(), boxes.float()
    wx, wy, ww, wh = weights
    widths, heights = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    ctr_x, ctr_y = boxes[:, 0] + 0.5 * widths, boxes[:, 1] + 0.5 * heights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=_SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=_SCALE_CLAMP)
    pcx, pcy = dx * widths[:, None] + ctr_x[:, None], dy * heights[:, None] + ctr_y[:, None]
    pw, ph = torch.exp(dw) * widths[:, None], torch.exp(dh) * heights[:, None]
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1).reshape(deltas.shape)


class _PredictorAPI:
    """d2 ``FastRCNNOutputLayers.{predict_boxes, predict_probs, predict_boxes_for_gt_classes}`` on
    ``predictions = (scores [R,K+1], proposal_deltas [R,4K])`` and ``list[Instances]`` proposals (A.12)."""

    def predict_boxes(self, predictions, proposals):
        _, deltas = predictions
        if not len(proposals):
            return []
        pb = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        return _apply_deltas(deltas, pb).split([len(p) for p in proposals])

    def predict_probs(self, predictions, proposals):
        scores, _ = predictions
        # the library's row softmax (sfod_predict_probs: the fused teacher post-processing's own operation order)
        return native.predict_probs(scores.float()).split([len(p) for p in proposals], dim=0)

    def predict_boxes_for_gt_classes(self, predictions, proposals):
        if not len(proposals):
            return []
        _, deltas = predictions
        pb = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        N, K = pb.shape[0], deltas.shape[1] // 4
        boxes = _apply_deltas(deltas, pb)
        if K > 1:
            gt = torch.cat([p.gt_classes for p in proposals], dim=0).clamp(0, K - 1)
            boxes = boxes.view(N, K, 4)[torch.arange(N, device=boxes.device), gt]
        return boxes.split([len(p) for p in proposals])


for _name in ("predict_boxes", "predict_probs", "predict_boxes_for_gt_classes"):
    setattr(FastRCNNOutputLayers, _name, getattr(_PredictorAPI, _name))


class SourceFreeFastRCNNOutputLayers(FastRCNNOutputLayers):
    """source_free_fast_rcnn.py:14.  ``convert_bbox_scores`` (:15-36 -> ``fast_rcnn_inference_single_image_new``
    :82-147) only feeds the zero-weighted, logged BPC loss; this is its Instances-level twin in torch ops --
    the training step computes BPC from the same predictions in one fused kernel (``sfod_bpc_loss``)."""

    def convert_bbox_scores(self, predictions, proposals):
        boxes = self.predict_boxes(predictions, proposals)
        scores = self.predict_probs(predictions, proposals)
        image_shapes = [x.image_size for x in proposals]
        return self.fast_rcnn_inference_new(boxes, scores, image_shapes, self.test_score_thresh, self.test_nms_thresh,
                                            self.test_topk_per_image, proposals)

    def fast_rcnn_inference_new(self, boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image, proposals):
        """:38-80 -- per image ``fast_rcnn_inference_single_image_new``; -> (list[Instances], list[row index])."""
        per_image = [self.fast_rcnn_inference_single_image_new(b, s, shape, score_thresh, nms_thresh, topk_per_image, p)
                     for s, b, shape, p in zip(scores, boxes, image_shapes, proposals)]
        return [x[0] for x in per_image], [x[1] for x in per_image]

    def fast_rcnn_inference_single_image_new(self, boxes, scores, image_shape, score_thresh=0.0, nms_thresh=0.0,
                                             topk_per_image=-1, proposal=None):
        """:82-147 -- rows with a non-finite box or score are dropped first (the returned row index counts the
        survivors), the background column goes, boxes are clipped, every (row, class) with ``score > 0`` is kept in
        row-major order; the thresholds are accepted and unused like the reference's (its NMS is commented out, :132-138).
        ``boxes`` [R, 4K] per-class or [R, 4] class-agnostic; ``scores`` [R, K+1] probabilities.
        Pinned by the reference-run fixture tests/golden/glue_ref.npz."""
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
        if not valid.all():
            boxes, scores = boxes[valid], scores[valid]
        scores = scores[:, :-1]
        nreg = boxes.shape[1] // 4
        b = Boxes(boxes.reshape(-1, 4))
        b.clip(image_shape)
        boxes = b.tensor.view(-1, nreg, 4)
        mask = scores > 0                     # "We do not filter out anything here"
        inds = mask.nonzero()
        res = Instances(image_shape)
        res.pred_boxes = Boxes(boxes[inds[:, 0], 0] if nreg == 1 else boxes[mask])
        res.scores = scores[mask]
        res.pred_classes = inds[:, 1]
        return res, inds[:, 0]


class InstanceProposals:
    """4th value of the training-mode ROI heads (``instance_proposals``,
    source_free_adaptive_teacher_roi_heads.py:101,158): the per-class decoded predictions of the sampled
    proposals, kept as references to the pass's prediction matrix and sample arrays.  ``bpc_loss(gt)`` runs the
    fused kernel; ``to_instances()`` materialises the reference's ``list[Instances]`` (host API, syncs)."""

    def __init__(self, heads, pred, samples):
        self.heads, self.pred, self.samples = heads, pred, samples

    def bpc_loss(self, targets, iou_thresh=0.5):
        h, sm = self.heads, self.samples
        sizes = native.dev_const(tuple((int(s[0]), int(s[1])) for s in sm["image_sizes"]), torch.int32,
                                 self.pred.device)
        return native.bpc_loss(self.pred, h.num_classes, sm["rois"], sm["gt_cls"], sizes, targets.boxes,
                               targets.classes, targets.count, iou_thresh)

    def to_instances(self):
        h, sm = self.heads, self.samples
        K = h.num_classes
        rois, cls = sm["rois"], sm["gt_cls"]
        props = []
        for b, size in enumerate(sm["image_sizes"]):
            m = rois[:, 0] == b
            p = Instances(size)
            p.proposal_boxes = Boxes(rois[m, 1:5])
            p.gt_classes = cls[m].long()
            props.append(p)
        order = torch.cat([torch.nonzero(rois[:, 0] == b).flatten() for b in range(len(props))])
        predictions = (self.pred[order, : K + 1], self.pred[order, K + 1: 5 * K + 1])
        bp = h.box_predictor
        for p, nb in zip(props, bp.predict_boxes_for_gt_classes(predictions, props)):     # :136-143
            p.proposal_boxes = Boxes(nb)
        return bp.convert_bbox_scores(predictions, props)[0]


class _ROILossFn(torch.autograd.Function):
    """features (+ sampled rois) -> (loss_cls, loss_box_reg) with a hand-written backward."""

    @staticmethod
    def forward(ctx, heads, feat_nchw, samples, *params):
        st = heads._box_forward(feat_nchw, samples["rois"])
        loss, _ = native.frcnn_loss(st["pred"], heads.num_classes, samples["rois"], samples["gt_cls"],
                                    samples["gt_box"], samples["n_valid"])
        ctx.heads, ctx.st, ctx.samples = heads, st, samples
        ctx.feat_shape = feat_nchw.shape
        heads._last_pred = st["pred"]          # for InstanceProposals (BPC); alive until the backward anyway
        return loss[0].clone(), loss[1].clone()

    @staticmethod
    def backward(ctx, g_cls, g_box):
        heads, st, samples = ctx.heads, ctx.st, ctx.samples
        gs = torch.stack([g_cls.reshape(()), g_box.reshape(())]).float().contiguous()
        _, d_pred = native.frcnn_loss(st["pred"], heads.num_classes, samples["rois"], samples["gt_cls"],
                                      samples["gt_box"], samples["n_valid"], grad_scale=gs)
        dfeat, pgrads = heads._box_backward(st, samples["rois"], d_pred, ctx.feat_shape)
        ctx.st = None
        return (None, dfeat, None) + tuple(pgrads)


@ROI_HEADS_REGISTRY.register()
class StandardROIHeads(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        r = cfg.MODEL.ROI_HEADS
        self.num_classes = r.NUM_CLASSES
        self.batch_size_per_image = r.BATCH_SIZE_PER_IMAGE
        self.positive_fraction = r.POSITIVE_FRACTION
        self.proposal_append_gt = r.PROPOSAL_APPEND_GT
        assert list(r.IOU_THRESHOLDS) == [r.IOU_THRESHOLDS[0]] and list(r.IOU_LABELS) == [0, 1]
        self.iou_threshold = r.IOU_THRESHOLDS[0]
        self.box_in_features = self.in_features = r.IN_FEATURES
        assert len(self.in_features) == 1
        shape = input_shape[self.in_features[0]]
        res = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.box_pooler = ROIPooler(res, (1.0 / shape.stride,), cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO,
                                    cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE)
        self.box_head = ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(
            cfg, ShapeSpec(channels=shape.channels, height=res, width=res))
        self.box_predictor = self._make_predictor(cfg, self.box_head.output_shape)
        self.pooled = res
        self.channels = shape.channels
        self.compute_dtype = native.mode_dtype(cfg.SFOD.COMPUTE_DTYPE)
        K = self.num_classes
        self.pred_ld = (5 * K + 1 + 7) // 8 * 8
        self.bbox_threshold = cfg.SEMISUPNET.BBOX_THRESHOLD if "SEMISUPNET" in cfg else 0.7
        self.nms_numel_limit = 20000  # torchvision batched_nms switch for GPU tensors (A.6)
        self.train_on_pred_boxes = cfg.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES

    def _make_predictor(self, cfg, shape):
        return FastRCNNOutputLayers(cfg, shape)

    def _params(self):
        bh, bp = self.box_head, self.box_predictor
        return [bh.fc1.weight, bh.fc1.bias, bh.fc2.weight, bh.fc2.bias, bp.cls_score.weight, bp.cls_score.bias,
                bp.bbox_pred.weight, bp.bbox_pred.bias]

    # ---- the training pass's weights, packed ahead of time ---------------------------------------------------------------
    @torch.no_grad()
    def prefetch_weights(self):
        """Pack the box head's and predictor's weights in the forms the training pass multiplies with -- forward and
        backward -- NOW; the next ``_box_forward`` takes them instead of packing (fc1 is 103 MB: four launches of 60-90 us
        that otherwise sit between the pseudo labels and the losses).  The trainer calls it (``model.prefetch_features``)
        while the teacher is still labelling; nothing changes the weights before this step's update, and what is not
        taken by then is dropped at the start of the next step (``drop_prefetched``)."""
        dt = native.dt_of_dtype(self.compute_dtype)
        gdt = native.dt_of_dtype(native.grad_dtype_of(self.compute_dtype))
        bh, bp, C = self.box_head, self.box_predictor, self.channels
        wp = torch.cat([bp.cls_score.weight.detach(), bp.bbox_pred.weight.detach()])
        self._packed = {
            "w1": native.pack_fc_weight(bh.fc1.weight.detach(), dt, chw_c=C),
            "w2": native.pack_fc_weight(bh.fc2.weight.detach(), dt),
            "wp": wp, "bpb": torch.cat([bp.cls_score.bias.detach(), bp.bbox_pred.bias.detach()]),
            "wpp": native.pack_fc_weight(wp, dt),
            "wpt": native.pack_fc_weight(wp, gdt, transpose=True, ld=self.pred_ld),
            "w2t": native.pack_fc_weight(bh.fc2.weight.detach(), gdt, transpose=True),
            "w1t": native.pack_fc_weight(bh.fc1.weight.detach(), gdt, chw_c=C, transpose=True)}

    # ---- box branch: ROIAlign -> fc1 -> fc2 -> fused (cls_score | bbox_pred) -------------------------
    def _box_forward(self, feat_nchw, rois):
        dtype = self.compute_dtype
        dt = native.dt_of_dtype(dtype)
        feat = native.nhwc_operand(feat_nchw, dtype)
        bh, bp = self.box_head, self.box_predictor
        C, PP = self.channels, self.pooled * self.pooled
        pk = self.__dict__.pop("_packed", None) or {}
        pooled = native.roi_align_fwd(feat, rois, self.pooled, self.box_pooler.scale)
        R = pooled.shape[0]
        x0 = pooled.view(R, PP * C)
        w1 = pk["w1"] if pk else native.pack_fc_weight(bh.fc1.weight.detach(), dt, chw_c=C)
        h1 = native.conv_fwd(x0, w1, bh.fc1.bias.detach(), bh.fc1.out_features, 1, act=1)
        w2 = pk["w2"] if pk else native.pack_fc_weight(bh.fc2.weight.detach(), dt)
        h2 = native.conv_fwd(h1, w2, bh.fc2.bias.detach(), bh.fc2.out_features, 1, act=1)
        wp = pk["wp"] if pk else torch.cat([bp.cls_score.weight.detach(), bp.bbox_pred.weight.detach()])
        bpb = pk["bpb"] if pk else torch.cat([bp.cls_score.bias.detach(), bp.bbox_pred.bias.detach()])
        wpp = pk["wpp"] if pk else native.pack_fc_weight(wp, dt)
        pred = native.conv_fwd(h2, wpp, bpb, wp.shape[0], 1, out_dtype=torch.float32, ldy=self.pred_ld)
        st = {"x0": x0, "h1": h1, "h2": h2, "pred": pred, "wp": wp, "feat_shape": tuple(feat.shape)}
        for k in ("wpt", "w2t", "w1t"):           # the backward's forms ride with the pass's state
            if k in pk:
                st[k] = pk[k]
        return st

    def _box_backward(self, st, rois, d_pred, feat_shape_nchw):
        dtype = native.grad_dtype_of(self.compute_dtype)      # operands of the backward products ("f16x3": bf16 pairs)
        dt = native.dt_of_dtype(dtype)
        bh = self.box_head
        K = self.num_classes
        NP = 5 * K + 1
        d_pred_c = native.cast(d_pred, dtype)
        # predictor: its parameter gradients beside the data-gradient path (``_OffChain``), like fc2's and fc1's below
        off = _OffChain(self, d_pred.is_cuda)
        dwp, dbp = off.run(lambda: (native.conv_wgrad(st["h2"], d_pred_c, NP, 1, operand=dtype).view(NP, -1),
                                    native.bias_grad(d_pred, NP)), st["h2"], d_pred_c, d_pred)
        wpt = st.get("wpt")
        if wpt is None:
            wpt = native.pack_fc_weight(st["wp"], dt, transpose=True, ld=self.pred_ld)
        dh2 = native.conv_fwd(d_pred_c, wpt, None, bh.fc2.out_features, 1)
        dfeat, (dw1, db1, dw2, db2) = self._box_head_backward(st, rois, dh2)     # (joins the side stream)
        off.join(dwp, dbp)
        pgrads = [dw1, db1, dw2, db2, dwp[: K + 1].contiguous(), dbp[: K + 1].contiguous(),
                  dwp[K + 1:].contiguous(), dbp[K + 1:].contiguous()]
        return dfeat, pgrads

    def _box_head_backward(self, st, rois, dh2):
        """grad wrt the box-head output (``box_features`` = relu(fc2), consumed in place) -> (grad wrt the feature
        map as an NCHW view, [dw1, db1, dw2, db2]); dw1 is None when it went straight into the flat gradient."""
        dtype = native.grad_dtype_of(self.compute_dtype)      # operands of the backward products ("f16x3": bf16 pairs)
        dt = native.dt_of_dtype(dtype)
        bh = self.box_head
        C = self.channels
        native.act_bwd_(dh2, st["h2"], 1)
        # fc2
        off = _OffChain(self, dh2.is_cuda)
        dw2, db2 = off.run(lambda: (native.conv_wgrad(st["h1"], dh2, bh.fc2.out_features, 1, operand=dtype).view(bh.fc2.out_features, -1),
                                    native.bias_grad(dh2, bh.fc2.out_features)), st["h1"], dh2)
        w2t = st.get("w2t")
        if w2t is None:
            w2t = native.pack_fc_weight(bh.fc2.weight.detach(), dt, transpose=True)
        dh1 = native.conv_fwd(dh2, w2t, None, bh.fc2.in_features, 1)
        native.act_bwd_(dh1, st["h1"], 1)
        # fc1 (K axis in (p, c) order inside the kernels, (c, p) in the state dict).  Its weight gradient (the largest of the
        # heads: 26-420 GFLOP + the 103-411 MB un-packing pass) does not feed the data-gradient path: it runs on a side
        # stream beside dx0 and the ROIAlign backward -- a latency-bound gather that leaves most of the chip idle -- like the
        # backbone's weight gradients beside its data-gradient chain (SFOD_HEAD_WGRAD_STREAM=0: one stream)
        def fc1_param_grads():
            dw1p = native.conv_wgrad(st["x0"], dh1, bh.fc1.out_features, 1, operand=dtype).view(bh.fc1.out_features, -1)
            dw1_ = native.grad_sink(bh.fc1.weight)
            if dw1_ is not None:     # 103 MB: accumulate straight into the flat gradient, nothing for autograd to add
                native.unpack_fc_wgrad(dw1p, dw1_, chw_c=C, accumulate=True)
                dw1_ = None
            else:
                dw1_ = torch.empty_like(bh.fc1.weight)
                native.unpack_fc_wgrad(dw1p, dw1_, chw_c=C)
            return dw1_, native.bias_grad(dh1, bh.fc1.out_features)

        dw1, db1 = off.run(fc1_param_grads, st["x0"], dh1)     # dh1 (after its ReLU mask) and x0 are complete on the main stream
        w1t = st.get("w1t")
        if w1t is None:
            w1t = native.pack_fc_weight(bh.fc1.weight.detach(), dt, chw_c=C, transpose=True)
        dx0 = native.conv_fwd(dh1, w1t, None, bh.fc1.in_features, 1)
        B, H, W, _ = st["feat_shape"]
        dfeat = native.roi_align_bwd(dx0.view(-1, self.pooled * self.pooled, C), rois, (B, H, W, C), self.pooled,
                                     self.box_pooler.scale)
        off.join(dw1, db1, dw2, db2)     # the heads' gradients are final before anyone (all-reduce, SGD, autograd) reads them
        return dfeat.permute(0, 3, 1, 2), [dw1, db1, dw2, db2]

    # ---- label_and_sample_proposals (roi_heads.py:165-215) ----------------------------------------
    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets, branch="", keys=None):
        props = proposals
        B, P, _ = props.boxes.shape
        if self.proposal_append_gt:
            boxes, count = native.append_gt(props.boxes, props.count, targets.boxes, targets.count)
        else:
            boxes, count = props.boxes, props.count
        matched, cls = native.roi_match(boxes, count, targets.boxes, targets.classes, targets.count,
                                        self.iou_threshold, self.num_classes)
        keys = keys if keys is not None else getattr(self, "_forced_keys", None)
        if keys is None:
            keys = torch.randint(0, 2 ** 31 - 1, (B, boxes.shape[1]), dtype=torch.int32, device=boxes.device)
        sidx, scnt = native.subsample_roi(cls, keys, self.batch_size_per_image, self.positive_fraction,
                                          self.num_classes)
        rois, gt_cls, gt_box, n_valid = native.roi_build_samples(boxes, cls, matched, sidx, scnt, targets.boxes,
                                                                 targets.count)
        return {"rois": rois, "gt_cls": gt_cls, "gt_box": gt_box, "n_valid": n_valid, "count": scnt,
                "sampled_idxs": sidx, "image_sizes": props.image_sizes}

    @torch.no_grad()
    def _inference(self, feat, proposals, sizes_dev=None):
        rois = native.make_rois(proposals.boxes, proposals.count)
        st = self._box_forward(feat, rois)
        bp = self.box_predictor
        if sizes_dev is None:
            sizes_dev = native.dev_const(tuple((int(s[0]), int(s[1])) for s in proposals.image_sizes), torch.int32,
                                         feat.device)
        out = native.frcnn_inference(st["pred"], self.num_classes, proposals.boxes, proposals.count, sizes_dev,
                                     bp.test_score_thresh, bp.test_nms_thresh, bp.test_topk_per_image,
                                     self.bbox_threshold, self.nms_numel_limit)
        return BatchedDetections(out, proposals.image_sizes), st["pred"]

    # ---- module surface (roi_heads.py:68-106) -------------------------------------------------------
    def forward(self, images, features, proposals, targets=None, compute_loss=True, branch="",
                compute_val_loss=False, as_instances=True, keys=None):
        del images
        feat = features[self.in_features[0]]
        if not isinstance(proposals, BatchedProposals):
            proposals = _proposals_from_instances(proposals, feat.device)
        if (self.training and compute_loss) or compute_val_loss:
            assert targets is not None
            if self.pooled > 8 and torch.is_grad_enabled():
                # sfod_roi_align_bwd's register tiles stop at 8 (the forward serves <= 16: Detectron2's unit-test size); say so
                # BEFORE the losses, not in the first backward.  config.py's default is d2's 14: the named yamls set 7.
                # (Checked here, not in _box_forward: inside an autograd.Function's forward grad mode is always off.)
                raise ValueError(f"MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION {self.pooled}: the ROIAlign backward serves <= 8 "
                                 "(forward-only passes: <= 16)")
            if not isinstance(targets, BatchedGT):
                targets = BatchedGT.from_instances(targets, feat.device)
            append = self.proposal_append_gt
            if compute_val_loss and not (self.training and compute_loss):
                self.proposal_append_gt = False
            samples = self.label_and_sample_proposals(proposals, targets, branch=branch, keys=keys)
            self.proposal_append_gt = append
            l_cls, l_box = _ROILossFn.apply(self, feat, samples, *self._params())
            losses = {"loss_cls": l_cls, "loss_box_reg": l_box}
            instance_proposals = None
            if isinstance(self.box_predictor, SourceFreeFastRCNNOutputLayers):
                instance_proposals = InstanceProposals(self, self._last_pred, samples)
            self._last_pred = None
            return samples, losses, None, instance_proposals
        pred_instances, predictions = self._inference(feat, proposals)
        return (pred_instances.to_instances() if as_instances else pred_instances), predictions


def _proposals_from_instances(instances, device):
    B = len(instances)
    P = max(1, max(len(i) for i in instances))
    boxes = torch.zeros(B, P, 4, dtype=torch.float32, device=device)
    logits = torch.zeros(B, P, dtype=torch.float32, device=device)
    cnt = torch.zeros(B, dtype=torch.int32)
    for i, inst in enumerate(instances):
        n = len(inst)
        boxes[i, :n] = inst.proposal_boxes.tensor.to(device)
        if inst.has("objectness_logits"):
            logits[i, :n] = inst.objectness_logits.to(device)
        cnt[i] = n
    return BatchedProposals(boxes, logits, cnt.to(device), [i.image_size for i in instances])


@ROI_HEADS_REGISTRY.register()
class SourceFreeAdaptiveTeacherStandardROIHeads(StandardROIHeads):
    def _make_predictor(self, cfg, shape):
        if cfg.MODEL.ROI_HEADS.LOSS == "CrossEntropy":
            return SourceFreeFastRCNNOutputLayers(cfg, shape)
        raise ValueError("Unknown ROI head loss.")


@ROI_HEADS_REGISTRY.register()
class AdaptiveTeacherStandardROIHeads(StandardROIHeads):
    def _make_predictor(self, cfg, shape):
        if cfg.MODEL.ROI_HEADS.LOSS == "CrossEntropy":
            return FastRCNNOutputLayers(cfg, shape)
        raise ValueError("Unknown ROI head loss.")
