"""Meta-architectures: ``GeneralizedRCNN`` (source-only training, configs #1/#2) and
``SourceFreeAdaptiveTeacherGeneralizedRCNN`` (teacher / student of the adaptation step).

Mirrors ``/root/reference/daod/modeling/meta_arch/source_free_adaptive_teacher_rcnn.py``:
``__init__`` ``:28-71`` (DC_img / DC_ins always constructed), ``from_config`` ``:75-90``,
branch-dispatched ``forward`` ``:106-339``:
  * ``unsup_data_weak``   ``:314-339`` -> ``({}, proposals_rpn, proposals_roih)``
  * ``supervised_target`` ``:259-312`` -> ``(losses, proposals_roih, [], [])``
  * ``supervised``        ``:227-257`` -> ``(losses, [], [])``
  * ``domain_classifier`` ``:137-210`` (zero-weighted in the hot yaml; see SFOD.ELIDE_DEAD_BRANCHES)
and Detectron2's GeneralizedRCNN (preprocess_image: normalise + pad, A.1).

Each sub-module is one autograd node with a hand-written backward (backbone, RPN losses, ROI
losses), so ``sum(losses.values()).backward()`` in a trainer works exactly as in the reference.
"""
import torch
import torch.nn as nn

from .. import native
from ..registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY
from ..structures import Boxes, ImageList, Instances
from .batched import BatchedGT, gather_gt
from .dann import DAInsHead, FCDiscriminator_img, dc_img_loss, dc_ins_loss


_NO_PREFETCH_RPN = __import__("os").environ.get("SFOD_NO_PREFETCH_RPN", "0") == "1"


def build_model(cfg):
    """d2 build_model: META_ARCH_REGISTRY.get(name)(cfg).to(cfg.MODEL.DEVICE)."""
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    if not str(cfg.MODEL.DEVICE).startswith("cuda"):
        raise RuntimeError(
            "MODEL.DEVICE must be a GPU: the hot path runs on HIP kernels only (no CPU fallback). "
            "The CPU restatement lives in oracle/ and is test infrastructure.")
    return model.to(torch.device(cfg.MODEL.DEVICE))


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, None)
        shape = self.backbone.output_shape()
        self.proposal_generator = PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(cfg, shape)
        self.roi_heads = ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, shape)
        if hasattr(self.backbone, "needed_features"):   # stage outputs somebody consumes (see backbone_vgg._VGGFn)
            self.backbone.needed_features = set(self.proposal_generator.in_features) | set(self.roi_heads.in_features)
        self.input_format = cfg.INPUT.FORMAT
        self.vis_period = cfg.VIS_PERIOD
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), False)
        self.compute_dtype = native.mode_dtype(cfg.SFOD.COMPUTE_DTYPE)
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs, key="image"):
        """uint8 CHW BGR images -> ImageList of the normalised, zero-padded batch.  The tensor is
        an NCHW view of NHWC memory (channels padded to one 16-byte chunk)."""
        imgs = [x[key].to(self.device, non_blocking=True) for x in batched_inputs]
        for im in imgs:
            if im.dtype != torch.uint8:
                raise TypeError("images must be uint8 CHW tensors (the mapper's output format)")
        hm = max(int(im.shape[1]) for im in imgs)
        wm = max(int(im.shape[2]) for im in imgs)
        dt = native.dt_of_dtype(self.compute_dtype)
        x, _ = native.preprocess(imgs, hm, wm, native.chunk_elems(dt), self._mean, self._std, dt)
        sizes = [(int(im.shape[1]), int(im.shape[2])) for im in imgs]
        return ImageList(x.permute(0, 3, 1, 2), sizes)

    def _features(self, images):
        return self.backbone.forward_nhwc(images.tensor.permute(0, 2, 3, 1))

    def prefetch_features(self, batched_inputs):
        """Run preprocessing + the backbone for ``batched_inputs`` NOW; the next ``forward`` on the same list
        object picks the result up instead of recomputing it.  Lets a trainer start the student's backbone
        before the pseudo-labels (which only the heads' losses need) exist."""
        self.drop_prefetched()
        images = self.preprocess_image(batched_inputs)
        features = self._features(images)
        self._prefetched = (batched_inputs, images, features)
        # ... and what else of the training pass needs no labels: the RPN's head convolutions and proposals (only the anchor
        # matching and the losses wait for the pseudo labels) and the box head's weight packing (SFOD_NO_PREFETCH_RPN=1: A/B hook)
        if self.training and not _NO_PREFETCH_RPN:
            if hasattr(self.proposal_generator, "prefetch"):
                self.proposal_generator.prefetch(images, features)
            if hasattr(self.roi_heads, "prefetch_weights"):
                self.roi_heads.prefetch_weights()

    def drop_prefetched(self):
        """forget whatever an earlier ``prefetch_features`` left behind (a step that raised before using it)"""
        self.__dict__.pop("_prefetched", None)
        # (a GeneralizedRCNN may be built without a proposal generator or ROI heads: d2 allows precomputed proposals)
        if self.proposal_generator is not None:
            self.proposal_generator.__dict__.pop("_prefetched", None)
        if self.roi_heads is not None:
            self.roi_heads.__dict__.pop("_packed", None)

    def _images_and_features(self, batched_inputs):
        pf = self.__dict__.pop("_prefetched", None)
        if pf is not None and pf[0] is batched_inputs:
            return pf[1], pf[2]
        if self.proposal_generator is not None:
            self.proposal_generator.__dict__.pop("_prefetched", None)  # whatever was prefetched is not for this batch
        images = self.preprocess_image(batched_inputs)
        return images, self._features(images)

    def forward(self, batched_inputs):
        if not self.training:
            return self.inference(batched_inputs)
        images = self.preprocess_image(batched_inputs)
        gt = gather_gt(batched_inputs, self.device)
        features = self._features(images)
        proposals, proposal_losses = self.proposal_generator(images, features, gt, as_instances=False)
        _, detector_losses, _, _ = self.roi_heads(images, features, proposals, gt)
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses

    @torch.no_grad()
    def inference(self, batched_inputs, do_postprocess=True):
        assert not self.training
        images = self.preprocess_image(batched_inputs)
        features = self._features(images)
        proposals, _ = self.proposal_generator(images, features, None, as_instances=False)
        results, _ = self.roi_heads(images, features, proposals, None)
        if not do_postprocess:
            return results
        out = []
        for res, inp, size in zip(results, batched_inputs, images.image_sizes):
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            out.append({"instances": detector_postprocess(res, h, w)})
        return out


def detector_postprocess(results, output_height, output_width):
    """d2 detector_postprocess for boxes: rescale to the requested resolution, clip, drop empty."""
    sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
    out = Instances((output_height, output_width), **results.get_fields())
    boxes = out.pred_boxes.clone()
    boxes.tensor = boxes.tensor * torch.tensor([sx, sy, sx, sy], device=boxes.tensor.device)
    boxes.clip(out.image_size)
    out.pred_boxes = boxes
    return out[boxes.nonempty()]


@META_ARCH_REGISTRY.register()
class SourceFreeAdaptiveTeacherGeneralizedRCNN(GeneralizedRCNN):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.dis_type = cfg.SEMISUPNET.DIS_TYPE
        if hasattr(self.backbone, "needed_features"):
            self.backbone.needed_features.add(self.dis_type)
        self.DC_img = FCDiscriminator_img(self.backbone._out_feature_channels[self.dis_type],
                                          compute_dtype=self.compute_dtype)
        self.ins_dc = cfg.SEMISUPNET.INS_DC
        if self.ins_dc:
            self.DC_ins = DAInsHead(self.roi_heads.box_predictor.cls_score.in_features, [self.dis_type],
                                    compute_dtype=self.compute_dtype)
        self.elide = cfg.SFOD.ELIDE_DEAD_BRANCHES

    def forward(self, batched_inputs, branch="supervised", given_proposals=None, val_mode=False, batched=False):
        """``batched=True`` returns the fixed-capacity device containers instead of
        ``list[Instances]`` (no host synchronisation); everything else follows the reference."""
        if (not self.training) and (not val_mode):
            return self.inference(batched_inputs)
        if branch == "domain_classifier":
            return self._forward_domain_classifier(batched_inputs)
        images, features = self._images_and_features(batched_inputs)
        gt = gather_gt(batched_inputs, self.device)

        if branch in ("supervised", "supervised_target"):
            proposals_rpn, proposal_losses = self.proposal_generator(images, features, gt, as_instances=False)
            _, detector_losses, _, proposal_instances = self.roi_heads(images, features, proposals_rpn,
                                                                       compute_loss=True, targets=gt, branch=branch)
            losses = {}
            losses.update(detector_losses)
            losses.update(proposal_losses)
            if branch == "supervised":
                # loss_DC_img_s * 0.001 (rcnn.py:256) belongs to the with-source trainer (out of scope)
                return losses, [], []
            proposals_roih = []
            if not self.elide:
                with torch.no_grad():
                    proposals_roih, _ = self.roi_heads(images, features, proposals_rpn, targets=None,
                                                       compute_loss=False, branch=branch, as_instances=not batched)
            # BPC (rcnn.py:293, bpc_loss.py): calibration of the training pass's per-class predictions against
            # the pseudo labels; weighted by 0 and logged (trainer :549-550,566).  One fused launch, no grad.
            if proposal_instances is not None:
                with torch.no_grad():
                    losses["loss_bpc"] = proposal_instances.bpc_loss(gt)
            else:
                losses["loss_bpc"] = torch.zeros((), device=self.device)
            return losses, proposals_roih, [], []

        if branch == "unsup_data_weak":
            proposals_rpn, _ = self.proposal_generator(images, features, None, compute_loss=False,
                                                       as_instances=False)
            proposals_roih, _ = self.roi_heads(images, features, proposals_rpn, targets=None, compute_loss=False,
                                               branch=branch, as_instances=False)
            if batched:
                return {}, proposals_rpn, proposals_roih
            return {}, proposals_rpn.to_instances(), proposals_roih.to_instances()
        raise ValueError(f"unknown branch {branch}")

    def _forward_domain_classifier(self, batched_inputs):
        """rcnn.py:137-210.  ``image`` is the source-side sample (label 0), ``image_unlabeled`` the target
        (label 1); both pass the backbone with gradients, the discriminator sees them through the
        gradient-reversal layer, and so do the box head's features for the instance-level losses (INS_DC)."""
        losses = {}
        feats = {}
        for key, label, tag in (("image", 0, "s"), ("image_unlabeled", 1, "t")):
            images = self.preprocess_image(batched_inputs, key=key)
            features = self._features(images)
            feats[tag] = (images, features)
            losses["loss_DC_img_" + tag] = dc_img_loss(self.DC_img, features[self.dis_type], label)
        if self.ins_dc:
            # rcnn.py:157-201: RPN proposals (no loss) -> label & sample against the (pseudo-)GT when there is
            # one -> box features with gradients -> instance-level discriminator behind the GRL
            rh = self.roi_heads
            for tag, label, gkey in (("s", 0, "instances"), ("t", 1, "instances_unlabeled")):
                images, features = feats[tag]
                with torch.no_grad():
                    gt = gather_gt(batched_inputs, self.device, key=gkey)
                    props, _ = self.proposal_generator(images, features, None, compute_loss=False, as_instances=False)
                    if gt is not None:
                        rois = rh.label_and_sample_proposals(props, gt, branch="domain_classifier")["rois"]
                    else:
                        rois = native.make_rois(props.boxes, props.count)
                losses["loss_DC_ins_" + tag] = dc_ins_loss(rh, self.DC_ins, features[rh.in_features[0]], rois, label,
                                                           training=self.training)
        return losses, [], []


@META_ARCH_REGISTRY.register()
class AdaptiveTeacherGeneralizedRCNN(SourceFreeAdaptiveTeacherGeneralizedRCNN):
    """The with-source meta-architecture (``daod/modeling/meta_arch/adaptive_teacher_rcnn.py:24-350``, reached by
    ``TRAINER: "adaptive_teacher"``).  Same sub-modules as the source-free class; what differs per branch:

      ``supervised``         :210-257  the image-level discriminator sees the source features through the gradient-reversal
                                       layer (label 0) and ``loss_DC_img_s * 0.001`` joins the losses;
      ``supervised_target``  :259-288  RPN + ROI losses only: no second (loss-free) ROI pass, no BPC; three values;
      ``unsup_data_weak`` / ``domain_classifier``: the source-free class's (three values).
    """

    def forward(self, batched_inputs, branch="supervised", given_proposals=None, val_mode=False, batched=False):
        if (not self.training) and (not val_mode):
            return self.inference(batched_inputs)
        if branch in ("domain_classifier", "unsup_data_weak"):
            return super().forward(batched_inputs, branch=branch, given_proposals=given_proposals, val_mode=val_mode,
                                   batched=batched)
        if branch not in ("supervised", "supervised_target"):
            raise ValueError(f"unknown branch {branch}")
        images, features = self._images_and_features(batched_inputs)
        gt = gather_gt(batched_inputs, self.device)
        losses = {}
        loss_dc = None
        if branch == "supervised":          # :211-213, before the proposal generator like the reference
            loss_dc = dc_img_loss(self.DC_img, features[self.dis_type], 0)
        proposals_rpn, proposal_losses = self.proposal_generator(images, features, gt, as_instances=False)
        _, detector_losses, _, _ = self.roi_heads(images, features, proposals_rpn, compute_loss=True, targets=gt,
                                                  branch=branch)
        losses.update(detector_losses)
        losses.update(proposal_losses)
        if loss_dc is not None:
            losses["loss_DC_img_s"] = loss_dc * 0.001      # :256
        return losses, [], []
