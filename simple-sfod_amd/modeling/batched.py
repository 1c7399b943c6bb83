"""Fixed-capacity batched device containers.

The reference moves per-image ``Instances`` lists between RPN, ROI heads and the trainer and
reads their lengths on the host (``.item()`` syncs at source_free_adaptive_teacher.py:419,
roi_heads.py:203-204).  Here every variable-length per-image result is a fixed-capacity device
array plus an int32 count, so the whole step can be enqueued without a host round trip; the
``Instances`` view is materialised only when a caller asks for it.
"""
import torch

from ..structures import Boxes, Instances

GT_CAP = 100  # == TEST.DETECTIONS_PER_IMAGE: the teacher can emit at most 100 pseudo boxes / image


class BatchedGT:
    """Ground-truth (or pseudo ground-truth) boxes: boxes [B,G,4] fp32, classes [B,G] int32, count [B]."""

    def __init__(self, boxes, classes, count, image_sizes, scores=None):
        self.boxes, self.classes, self.count, self.image_sizes, self.scores = boxes, classes, count, image_sizes, scores

    @staticmethod
    def from_instances(instances, device, cap=GT_CAP):
        B = len(instances)
        boxes = torch.zeros(B, cap, 4, dtype=torch.float32)
        classes = torch.zeros(B, cap, dtype=torch.int32)
        count = torch.zeros(B, dtype=torch.int32)
        sizes = []
        for i, inst in enumerate(instances):
            n = len(inst) if inst.has("gt_boxes") else 0
            if n > cap:
                raise ValueError(f"more than {cap} ground-truth boxes in one image ({n})")
            if n:
                boxes[i, :n] = inst.gt_boxes.tensor.detach().float().cpu()
                classes[i, :n] = inst.gt_classes.detach().cpu().to(torch.int32)
            count[i] = n
            sizes.append(tuple(inst.image_size))
        return BatchedGT(boxes.to(device), classes.to(device), count.to(device), sizes)

    def __len__(self):
        return self.boxes.shape[0]

    def view(self, i):
        return InstancesRef(self, i)

    def to_instances(self):
        cnt = self.count.tolist()  # host sync
        out = []
        for i, n in enumerate(cnt):
            inst = Instances(self.image_sizes[i])
            inst.gt_boxes = Boxes(self.boxes[i, :n])
            inst.gt_classes = self.classes[i, :n].long()
            if self.scores is not None:
                inst.scores = self.scores[i, :n]
            out.append(inst)
        return out


class InstancesRef:
    """Stand-in for ``batched_inputs[i]["instances"]`` that points into a BatchedGT (no host copy)."""

    def __init__(self, batch, index):
        self.batch, self.index = batch, index

    def to(self, *a, **k):
        return self

    def materialize(self):
        return self.batch.to_instances()[self.index]


def gather_gt(batched_inputs, device, key="instances"):
    """``[x["instances"] for x in batched_inputs]`` -> BatchedGT (or None)."""
    if key not in batched_inputs[0]:
        return None
    insts = [x[key] for x in batched_inputs]
    if all(isinstance(i, InstancesRef) for i in insts):
        b = insts[0].batch
        if all(i.batch is b for i in insts) and [i.index for i in insts] == list(range(len(b))):
            return b
        insts = [i.materialize() for i in insts]
    return BatchedGT.from_instances(insts, device)


class BatchedProposals:
    """RPN output: boxes [B,P,4], objectness logits [B,P], count [B]."""

    def __init__(self, boxes, logits, count, image_sizes):
        self.boxes, self.logits, self.count, self.image_sizes = boxes, logits, count, image_sizes

    def __len__(self):
        return self.boxes.shape[0]

    def to_instances(self):
        cnt = self.count.tolist()
        out = []
        for i, n in enumerate(cnt):
            inst = Instances(self.image_sizes[i])
            inst.proposal_boxes = Boxes(self.boxes[i, :n])
            inst.objectness_logits = self.logits[i, :n]
            out.append(inst)
        return out


class BatchedDetections:
    """Teacher ROI-head inference output (<= 100 / image) and the thresholded pseudo labels."""

    def __init__(self, d, image_sizes):
        self.d, self.image_sizes = d, image_sizes

    def __len__(self):
        return self.d["det_boxes"].shape[0]

    def to_instances(self):
        cnt = self.d["det_count"].tolist()
        out = []
        for i, n in enumerate(cnt):
            inst = Instances(self.image_sizes[i])
            inst.pred_boxes = Boxes(self.d["det_boxes"][i, :n])
            inst.scores = self.d["det_scores"][i, :n]
            inst.pred_classes = self.d["det_classes"][i, :n].long()
            out.append(inst)
        return out

    def pseudo_gt(self):
        """The ``scores > BBOX_THRESHOLD`` subset (threshold_bbox 'roih') as a BatchedGT."""
        # fixed threshold: the labels are a prefix of the score-ordered detections; after the class-wise
        # adaptive selection (native.adaptive_pseudo_labels_) they are a subset with their own score array
        scores = self.d["gt_scores"] if self.d.get("gt_adaptive") else self.d["det_scores"]
        return BatchedGT(self.d["gt_boxes"], self.d["gt_classes"], self.d["gt_count"], self.image_sizes,
                         scores=scores)
