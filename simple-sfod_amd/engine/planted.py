"""Planted-label mode (BASELINE.md section 3 / SURVEY 8d: "teacher box-predictor bias is set so ~10-30 boxes/image pass").

Seeded random weights give no detection above ``SEMISUPNET.BBOX_THRESHOLD`` (0.8, reference
``source_free_adaptive_teacher.py:167-181``), so the synthetic workload plants labels: the class logits of the random
box predictor are spread by a moderate scale on ``cls_score.weight`` and the BACKGROUND bias is bisected over untimed
train-mode teacher passes until the mean number of pseudo labels per image is ``TARGET``.  This is the head ``bench.py``
times AND the head ``tests/test_gpu_fullsize.py`` / ``tests/test_gpu_pseudo_labels.py`` gate against the oracle: one
definition, here, so that the two cannot drift apart.
"""
import torch

# smaller scales leave fewer than 20 boxes above 0.8 at ANY bias (profiles/round5/r5_planted_label_calibration.txt)
SCALE = {"vgg": 16.0, "r101": 4.0}
TARGET = 20.0
RANGE = (10.0, 30.0)
BISECTION_STEPS = 24
BIAS_BRACKET = (-40.0, 40.0)


def spread_class_logits(box_predictor, scale):
    with torch.no_grad():
        box_predictor.cls_score.weight.mul_(scale)


def bisect_background_bias(counts, target=TARGET, bracket=BIAS_BRACKET, steps=BISECTION_STEPS, bias=None, tol=0.5):
    """``counts(bias) -> tensor`` of pseudo labels per image with that background bias in place.  More background bias ->
    fewer foreground detections: the count is monotone in it.  The bisection stops at the first midpoint whose mean count
    is within ``tol`` of the target: run to its last step it would converge ONTO a threshold crossing -- a bias at which one
    detection's score equals BBOX_THRESHOLD to the last bit -- which is the one score distribution nobody meets in practice
    (and the worst case for any comparison of two arithmetics).  ``bias``: use this one, no bisection.
    -> (bias, counts at it)"""
    lo, hi = bracket
    if bias is not None:
        return float(bias), counts(float(bias)).float()
    for _ in range(steps):
        mid = 0.5 * (lo + hi)
        c = counts(mid).float()
        m = c.mean().item()
        if abs(m - target) <= tol:
            return mid, c
        if m > target:
            lo = mid
        else:
            hi = mid
    b = 0.5 * (lo + hi)
    return b, counts(b).float()


def _summary(scale, b, c):
    return {"cls_score_weight_scale": scale, "background_bias": round(b, 4),
            "calibration_pseudo_labels_per_image": {"mean": round(c.mean().item(), 2), "min": int(c.min().item()),
                                                    "max": int(c.max().item()), "images": int(c.numel())}}


def plant_labels(trainer, scale, target=TARGET, note=lambda m: None, bias=None):
    """A teacher-student trainer: ``cls_score.weight *= scale`` on the student, bias calibrated on the weak views of two
    batches of the trainer's own loader; student and teacher get the same values (teacher <- student copy, as at
    construction), and the copy is repeated at the end so that the calibration passes leave nothing behind but the head.
    -> dict(scale, background bias, pseudo labels per image on the calibration frames: mean / min / max)."""
    bp = trainer.model.roi_heads.box_predictor
    K = bp.cls_score.bias.numel() - 1
    batches = [next(trainer._data_loader_iter)[1] for _ in range(2)]      # weak views of 2 batches

    def counts(delta):
        with torch.no_grad():
            bp.cls_score.bias[K] = delta
            trainer._copy_main_model()
            c = [trainer._teacher_pass([dict(d) for d in b]).count.float() for b in batches]
        trainer.storage._pending.clear()
        return torch.cat(c)

    spread_class_logits(bp, scale)
    b, c = bisect_background_bias(counts, target, bias=bias)
    with torch.no_grad():
        trainer._copy_main_model()
    out = _summary(scale, b, c)
    note(f"planted labels: {out}")
    return out


def plant_model(model, inputs, scale, target=TARGET, bias=None):
    """One model labelling ``inputs`` itself (the tests' form: no trainer, no loader).  The model's buffers (BatchNorm
    running statistics, counters: the train-mode passes move them) are restored afterwards."""
    bp = model.roi_heads.box_predictor
    K = bp.cls_score.bias.numel() - 1
    saved = {k: v.clone() for k, v in model.named_buffers()}

    def counts(delta):
        with torch.no_grad():
            bp.cls_score.bias[K] = delta
            _, _, dets = model([dict(d) for d in inputs], branch="unsup_data_weak", batched=True)
        return dets.d["gt_count"].float()

    spread_class_logits(bp, scale)
    b, c = bisect_background_bias(counts, target, bias=bias)
    with torch.no_grad():
        for k, v in model.named_buffers():
            v.copy_(saved[k])
    return _summary(scale, b, c)
