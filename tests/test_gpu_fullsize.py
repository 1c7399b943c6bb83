"""Parity gate of the benchmarked mode at the benchmark's own frame size.

Hot yaml (faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml), B = 2 frames of 600x1200 (what
ResizeShortestEdge makes of the 1024x2048 synthetic frames), teacher pass then student pass, in the default
``SFOD.COMPUTE_DTYPE bf16x3`` (what bench.py times) and in ``fp32``:

  * floating point (RPN logits / deltas, box-head scores / deltas, the four losses) within 1e-4 of the CPU oracle
    (BASELINE.json north_star: "losses/boxes within 1e-4 fp32");
  * every discrete decision bit-exact ON THE CAPTURED TENSORS: the device's proposal set equals the oracle's
    decode -> top-k -> NMS(0.7) run on the device's own logits / deltas (scores compared with ==, i.e. identical keep
    indices), the device's detections / pseudo-labels equal the oracle's softmax -> decode -> class-wise NMS(0.5) ->
    top-100 -> score > 0.8 on the device's own predictions (classes and ROI indices with ==), and the anchor labels
    after sampling equal the oracle's (==).

Feeding the oracle the device's tensors at the discrete steps is what makes "bit-exact" checkable: a 1e-7 difference in a
logit legitimately flips a rank / NMS decision in ANY two fp32 implementations (another summation order does it).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import model as om

pytestmark = pytest.mark.gpu
DEV = "cuda"
HOT_YAML = os.path.join(os.path.dirname(GOLDEN), "..", "configs",
                        "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml")


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _frames(B, H, W, seed):
    """smooth + noise frames: content-dependent scores (pure uniform noise makes all proposals near-ties)"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(B):
        yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        base = torch.stack([(yy * 3 + xx * 5).sin(), (yy * 7 - xx * 2).cos(), (yy * xx * 9).sin()]) * 70 + 120
        for _ in range(12):      # rectangles of constant colour
            y0, x0 = int(torch.randint(0, H - 64, (1,), generator=g)), int(torch.randint(0, W - 64, (1,), generator=g))
            h, w = int(torch.randint(32, 200, (1,), generator=g)), int(torch.randint(32, 300, (1,), generator=g))
            base[:, y0:y0 + h, x0:x0 + w] = torch.randint(0, 256, (3, 1, 1), generator=g).float()
        img = (base + torch.randn(3, H, W, generator=g) * 12).clamp(0, 255).to(torch.uint8)
        out.append({"image": img, "height": H, "width": W})
    return out


R101_YAML = os.path.join(os.path.dirname(GOLDEN), "..", "configs", "r101_c4_cs_foggy_adaptive_teacher_source_free.yaml")

# What "within 1e-4" is asserted on (BASELINE.json north_star: "losses/boxes within 1e-4 fp32; bit-exact anchor/label
# assignment and NMS indices"), per quantity, and what the intermediates are held to:
#   losses                      |dev / ref - 1| < 1e-4
#   decoded anchor boxes        relative L2 over all B x A x 4 coordinates < 1e-4, and the worst single coordinate
#                               |d| < 1e-4 x max(frame width, the box's own extent)  [= 0.12 px for boxes inside the frame]
#   decoded detection boxes     the same over all R x K x 4 coordinates
#   intermediates               RPN logits / deltas, box-head scores / deltas: relative L2 < GATE_INTERMEDIATE[dtype], or
#                               3 x the reference arithmetic's OWN error on this network when that is larger (below)
# bf16x3 carries 16 significand bits per operand (4.4e-6 rms per dot product, profiles/round2/r2_mfma_split_precision.txt);
# over 14 VGG convolutions the intermediates reach ~1e-4 relative L2, which is why they get 2e-4 and NOT the 1e-4 of the
# north-star quantities (bench.py's dtype_note says the same).
#
# Noise floor.  A randomly initialised ResNet-101-C4 is ill-conditioned: the fp32 ORACLE itself is 7e-5 away from the
# same network evaluated in fp64 at the RPN logits (VGG16: 3e-6), growing linearly over the 33 bottleneck blocks
# (2.8e-6 per res4 block); two correct fp32 implementations cannot agree better than that.  For a ResNet config the
# test therefore evaluates the oracle in fp64 as well and gates the intermediates at max(GATE_INTERMEDIATE, 3 x floor),
# floor = err(fp32 oracle, fp64 oracle), printed with the result, and the worst single box coordinate (a maximum over
# 2.7e5 values, ~5 sigma) at max(1e-4, 6 x floor).  Losses and the relative-L2 box gates keep their fixed 1e-4.
GATE_NORTH_STAR = 1e-4
GATE_INTERMEDIATE = {"fp32": 2e-5, "bf16x3": 2e-4, "f16x3": 2e-5}
# bf16x3 on the 101-layer config is NOT a parity mode (16-bit operands x the network's 20x worse conditioning:
# measured 1.7e-3 at the RPN logits, loss_box_reg 3.7e-4): bench.py --model r101 therefore reports f16x3, whose forward
# operands carry 22 bits (half pairs, weights under a per-tensor power-of-two scale: tests/test_gpu_f16x3.py) and which
# lands where fp32 does on both networks (VGG16: intermediates 7e-6; R101: 1.07e-4 against fp32's 1.13e-4 and a
# reference floor of 7.2e-5).  bf16x3 is still checked to TRACK the oracle at these looser, labelled gates, with every
# discrete step bit-exact as usual.
GATE_TRACKING = {"north_star": 2e-3, "intermediate": 5e-3, "px": 1e-2}


def _report(tag, errs, gates, info=None):
    rows = []
    for k, v in errs.items():
        g = gates[k]
        rows.append(f"{k} {v:.2e} (gate {g:.0e}, margin {'>1e6' if v < 1e-30 else format(g / v, '.1f')}x)")
    rows += [f"{k} {v:.2e} (not gated)" for k, v in (info or {}).items()]
    print(f"\n[fullsize parity {tag}] vs the CPU oracle: " + "; ".join(rows))
    bad = {k: (v, gates[k]) for k, v in errs.items() if not v < gates[k]}
    assert not bad, f"{tag}: outside the gate: {bad}"


# The benchmarked head (``bench_head``): engine/planted.py's moderate scale + bisected background bias, ~20 pseudo labels per
# image -- the score distribution bench.py times.  On it the class PROBABILITIES themselves (what the 0.05 and 0.8
# thresholds read) are gated, not only the logits: worst |p_dev - p_ref| over all R x (K + 1) entries.  The gate is the
# logit error pushed through the softmax at the planted scale (|dp| <= |dlogit| / 4 per unit error, measured values printed).
# Measured (round 6, profiles/r6_bench_head_parity.txt): VGG16 bf16x3 2.3e-4 (B = 8), fp32 2.5e-5; ResNet-101-C4 f16x3 3.6e-4 --
# on that network the reference arithmetic's OWN fp32-vs-fp64 error is 7e-5 relative on the logits (header), the class
# logits' error is the same 7e-5 in f16x3 and in fp32, and the planted x4 carries it into the probabilities.
GATE_PROBS_BENCH_HEAD = {"vgg": {"bf16x3": 6e-4, "f16x3": 1e-4, "fp32": 1e-4}, "r101": {"f16x3": 8e-4, "fp32": 8e-4}}


def _teacher_student_parity(sfod, yaml, ocfg, dtype, B, plant, tag, seed=7, tracking_only=False, last_gamma=None,
                            bench_head=None):
    """Teacher pass (captured intermediates) and student pass of ``yaml`` at 600x1200 against the oracle ``ocfg``.
    ``bench_head`` ("vgg" / "r101"): instead of the fixed ``plant`` scales, the head bench.py times -- cls_score x
    ``planted.SCALE``, background bias bisected on THESE frames to ``planted.TARGET`` pseudo labels per image.
    ``tracking_only``: the looser GATE_TRACKING set (a mode that is not the config's parity mode).
    ``last_gamma`` (ResNet): the BatchNorm weight of every bottleneck's LAST norm is set to this value (the default
    initialisation is 1; torchvision / Detectron2 checkpoints of trained networks sit well below, ``zero_init_residual``
    starts at 0): the block's branch then re-injects that fraction of the accumulated rounding error into the residual
    stream instead of all of it, the reference arithmetic's own fp32-vs-fp64 floor drops below 1e-5, and the FIXED gates
    (1e-4 on every box coordinate, 2e-5 on the intermediates) apply without any floor scaling."""
    NS = GATE_TRACKING["north_star"] if tracking_only else GATE_NORTH_STAR
    PX = GATE_TRACKING["px"] if tracking_only else GATE_NORTH_STAR
    S = sfod.structures
    H, W = 600, 1200
    cfg = sfod.config.setup_cfg(yaml, ["OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", dtype])
    torch.manual_seed(seed)
    model = sfod.modeling.build_model(cfg).train()
    inputs = _frames(B, H, W, seed=21)
    planted = None
    if bench_head is not None:
        PL = sfod.engine.planted
        planted = PL.plant_model(model, inputs, PL.SCALE[bench_head])
        plant = (PL.SCALE[bench_head], 1.0)
        print(f"\n[fullsize parity {tag} {dtype} B={B}] bench head: {planted}")
        c = planted["calibration_pseudo_labels_per_image"]
        assert PL.RANGE[0] <= c["mean"] <= PL.RANGE[1], planted
    with torch.no_grad():       # planted labels: some detections clear the 0.8 pseudo-label threshold
        if bench_head is None:
            model.roi_heads.box_predictor.cls_score.weight.mul_(plant[0])
            model.roi_heads.box_predictor.bbox_pred.weight.mul_(plant[1])
        if last_gamma is not None:
            n_set = 0
            for name, p_ in model.named_parameters():
                if name.startswith("backbone.res") and name.endswith("conv3.norm.weight") and p_.requires_grad:
                    p_.fill_(last_gamma)
                    n_set += 1
            assert n_set >= 20, n_set
    sd = om.clone_state({k: v.detach().float().cpu() if v.dtype != torch.int64 else v.detach().cpu()
                         for k, v in model.state_dict().items()})
    sd0 = om.clone_state(sd)        # before the train-mode passes move the running statistics
    images = [d["image"] for d in inputs]
    stride, A, K = ocfg.stride, ocfg.num_anchors, ocfg.num_classes
    Hf, Wf = -(-H // stride) if ocfg.backbone == "resnet" else H // stride, -(-W // stride) if ocfg.backbone == "resnet" else W // stride
    gates, errs, info = {}, {}, {}

    def put(name, value, gate):
        errs[name], gates[name] = value, gate

    # ---------------- teacher pass on the device, intermediates captured ----------------------------------------
    cap = {}
    rpn, rh = model.proposal_generator, model.roi_heads
    orig_props, orig_inf = rpn._proposals, rh._inference

    def cap_props(st, *a, **k):
        cap["rpn_out"] = st["rpn_out"].clone()
        cap["props"] = orig_props(st, *a, **k)
        return cap["props"]

    def cap_inf(*a, **k):
        dets, pred = orig_inf(*a, **k)
        cap["pred"] = pred.clone()
        return dets, pred
    rpn._proposals, rh._inference = cap_props, cap_inf
    with torch.no_grad():
        _, props, dets = model(inputs, branch="unsup_data_weak", batched=True)
    rpn._proposals, rh._inference = orig_props, orig_inf
    torch.cuda.synchronize()

    # ---------------- oracle: the same pass on the CPU ---------------------------------------------------------------
    with torch.no_grad():
        x, sizes = om.preprocess(images)
        feat = om.backbone_forward(sd, x, ocfg, training=True)
        logits_ref, deltas_ref = om.rpn_head(sd, feat)
    assert tuple(feat.shape[-2:]) == (Hf, Wf)
    anchors = om.anchors_for((Hf, Wf), ocfg)
    assert anchors.shape[0] == Hf * Wf * A
    out = cap["rpn_out"].cpu().view(B, Hf * Wf, -1)
    logits_dev = out[:, :, :A].reshape(B, Hf * Wf * A)
    deltas_dev = out[:, :, A:5 * A].reshape(B, Hf * Wf * A, 4)
    TI = GATE_TRACKING["intermediate"] if tracking_only else GATE_INTERMEDIATE[dtype]
    if ocfg.backbone == "resnet":       # the reference arithmetic's own error on this network (see the header)
        sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        with torch.no_grad():
            l64, d64 = om.rpn_head(sd64, om.backbone_forward(sd64, x.double(), ocfg, training=True))
        floor = max(rel_err(logits_ref, l64), rel_err(deltas_ref, d64))
        del sd64, l64, d64
        if last_gamma is not None:
            assert floor < 1e-5, f"the well-conditioned weight set should have a floor below 1e-5, got {floor:.2e}"
        elif not tracking_only:
            TI = max(TI, 3.0 * floor)
            # the worst of 2.7e5 coordinates sits ~5 sigma out: the reference arithmetic's own worst coordinate on this
            # network is ~5 x its relative L2 (the relative-L2 box gates keep the fixed 1e-4)
            PX = max(PX, 6.0 * floor)
        put("reference_fp32_vs_fp64_floor", floor, 2e-4)
    put("rpn_logits", rel_err(logits_dev, logits_ref), TI)
    put("rpn_deltas", rel_err(deltas_dev, deltas_ref), TI)
    # boxes within 1e-4: every anchor's decoded box (all B x A, before any ranking) from the device's deltas against
    # the oracle's -- relative L2 of the coordinates, and the largest deviation relative to the frame's 1200 px
    from oracle import box_ops as OB
    bx_dev = torch.stack([OB.apply_deltas(deltas_dev[b], anchors, ocfg.rpn_bbox_weights) for b in range(B)])
    bx_ref = torch.stack([OB.apply_deltas(deltas_ref[b], anchors, ocfg.rpn_bbox_weights) for b in range(B)])
    put("anchor_boxes_relL2", rel_err(bx_dev, bx_ref), NS)

    def worst_coord(dev, ref):
        """largest coordinate deviation of any box, relative to the frame width or -- for a box larger than the frame
        (random-init deltas reach exp(4.1) x a 724 px anchor) -- to the box's own extent"""
        dev, ref = dev.reshape(-1, 4).double(), ref.reshape(-1, 4).double()
        extent = torch.maximum(ref[:, 2] - ref[:, 0], ref[:, 3] - ref[:, 1]).clamp(min=float(W))
        return ((dev - ref).abs().max(dim=1).values / extent).max().item()
    put("anchor_boxes_worst_coord", worst_coord(bx_dev, bx_ref), PX)

    # proposals: the oracle's decode / top-k / NMS on the DEVICE's logits and deltas == the device's proposal set
    pr_ref = om.rpn_proposals(anchors, logits_dev, deltas_dev, sizes, ocfg, training=True)
    given = []
    for b in range(B):
        n = props.count[b].item()
        assert n == len(pr_ref[b][0])
        assert torch.equal(props.logits[b, :n].cpu(), pr_ref[b][1]), "NMS keep set / order differs"
        torch.testing.assert_close(props.boxes[b, :n].cpu(), pr_ref[b][0], rtol=1e-5, atol=1e-3)
        given.append(props.boxes[b, :n].cpu())
    # box head on the device's proposals
    with torch.no_grad():
        scores_ref, bdeltas_ref, _ = om.box_head(sd, feat, given, ocfg)
    P = props.boxes.shape[1]
    pred = cap["pred"].cpu().view(B, P, -1)
    scores_dev = torch.cat([pred[b, : len(given[b]), :K + 1] for b in range(B)])
    bdeltas_dev = torch.cat([pred[b, : len(given[b]), K + 1:5 * K + 1] for b in range(B)])
    put("box_scores", rel_err(scores_dev, scores_ref), TI)
    put("box_deltas", rel_err(bdeltas_dev, bdeltas_ref), TI)
    # detection boxes within 1e-4: per-class decode of every proposal from the device's deltas against the oracle's
    pb = torch.cat(given)
    db_dev = OB.apply_deltas(bdeltas_dev, pb, ocfg.roi_bbox_weights)
    db_ref = OB.apply_deltas(bdeltas_ref, pb, ocfg.roi_bbox_weights)
    put("det_boxes_relL2", rel_err(db_dev, db_ref), NS)
    put("det_boxes_worst_coord", worst_coord(db_dev, db_ref), PX)
    # class probabilities (what the 0.05 / 0.8 thresholds read).  Gated where the arithmetic is: on the class LOGITS -- their
    # worst single element against the logits' rms at 5 x the relative-L2 gate (the worst of R x (K + 1) ~ 1e5 values sits
    # ~4.5 sigma out) -- and on the probabilities of the head AS THE YAML INITIALISES IT, i.e. with the test's planted scale
    # on cls_score (x60 / x3, there so that detections clear the 0.8 threshold) divided out again.  The planted
    # probabilities' own difference is printed as well (ungated: it is the logit error times the planted scale pushed
    # through the softmax -- bf16x3 9.5e-4 at x60, i.e. 1.6e-5 per unit scale).
    rms = scores_ref.double().pow(2).mean().sqrt().item()
    put("box_logits_worst_over_rms", (scores_dev - scores_ref).abs().max().item() / rms, 5.0 * TI)
    put("det_probs_max_abs_unplanted",
        (torch.softmax(scores_dev / plant[0], -1) - torch.softmax(scores_ref / plant[0], -1)).abs().max().item(),
        GATE_TRACKING["north_star"] if tracking_only else GATE_NORTH_STAR)
    dp = (torch.softmax(scores_dev, -1) - torch.softmax(scores_ref, -1)).abs().max().item()
    if bench_head is not None and not tracking_only:
        put("det_probs_max_abs_bench_head", dp, GATE_PROBS_BENCH_HEAD[bench_head][dtype])
    else:
        info["det_probs_max_abs_planted"] = dp

    # detections + pseudo-labels: the oracle's post-processing of the DEVICE's predictions == the device's
    det_ref = om.fast_rcnn_inference(scores_dev, bdeltas_dev, given, sizes, ocfg)
    n_pseudo = 0
    for b in range(B):
        nd = dets.d["det_count"][b].item()
        assert nd == len(det_ref[b]["scores"])
        assert torch.equal(dets.d["det_classes"][b, :nd].cpu().long(), det_ref[b]["classes"])
        torch.testing.assert_close(dets.d["det_scores"][b, :nd].cpu(), det_ref[b]["scores"], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(dets.d["det_boxes"][b, :nd].cpu(), det_ref[b]["boxes"], rtol=1e-5, atol=2e-3)
        pl = om.threshold_bbox(det_ref[b], 0.8)
        ng = dets.d["gt_count"][b].item()
        # a score within 1e-6 of the threshold may land on either side of the strict '>'
        near = ((det_ref[b]["scores"] - 0.8).abs() < 1e-6).sum().item()
        assert abs(ng - len(pl["scores"])) <= near
        if near == 0:
            assert torch.equal(dets.d["gt_classes"][b, :ng].cpu().long(), pl["gt_classes"])
            torch.testing.assert_close(dets.d["gt_boxes"][b, :ng].cpu(), pl["gt_boxes"], rtol=1e-5, atol=2e-3)
        n_pseudo += ng
    assert n_pseudo >= 4, "planted labels should yield pseudo ground truth"
    if bench_head is not None:
        assert sfod.engine.planted.RANGE[0] <= n_pseudo / B <= sfod.engine.planted.RANGE[1], n_pseudo / B
    # BatchNorm running statistics refreshed by the train-mode teacher (AdaBN); frozen statistics untouched
    worst_rs = 0.0
    for name, buf in model.state_dict().items():
        if "running" in name:
            worst_rs = max(worst_rs, rel_err(buf, sd[name]))     # per tensor (a mean close to 0 has no relative error of its own)
        elif name.endswith("num_batches_tracked"):
            assert int(buf) == int(sd[name]) == 1, name
    put("running_stats_relL2", worst_rs, NS)

    # ---------------- student pass on the pseudo labels -------------------------------------------------------------
    for b, d in enumerate(inputs):
        ng = dets.d["gt_count"][b].item()
        inst = S.Instances((H, W))
        inst.gt_boxes = S.Boxes(dets.d["gt_boxes"][b, :ng].cpu().clone())
        inst.gt_classes = dets.d["gt_classes"][b, :ng].cpu().long().clone()
        d["instances"] = inst
    g = torch.Generator().manual_seed(5)
    rpn_keys = torch.randint(0, 2 ** 31 - 1, (B, Hf * Wf * A), generator=g, dtype=torch.int64)
    roi_keys = torch.randint(0, 2 ** 31 - 1, (B, 2100), generator=g, dtype=torch.int64)
    rpn._forced_keys = rpn_keys.to(torch.int32).to(DEV)
    rh._forced_keys = roi_keys.to(torch.int32).to(DEV)
    orig_lf = rpn._loss_forward

    def cap_lf(*a, **k):
        loss, lab_state = orig_lf(*a, **k)
        cap["labels"] = lab_state[0].clone()
        return loss, lab_state
    rpn._loss_forward, rpn._proposals = cap_lf, cap_props
    losses, _, _, _ = model(inputs, branch="supervised_target", batched=True)
    rpn._loss_forward, rpn._proposals = orig_lf, orig_props
    sum(v for k, v in losses.items() if k != "loss_bpc").backward()
    torch.cuda.synchronize()
    pr = cap["props"]
    given2 = [(pr.boxes[b, : pr.count[b].item()].cpu(), pr.logits[b, : pr.count[b].item()].cpu()) for b in range(B)]
    with torch.no_grad():
        ref, aux = om.student_losses(sd, images, [d["instances"].gt_boxes.tensor for d in inputs],
                                     [d["instances"].gt_classes for d in inputs], list(rpn_keys), list(roi_keys), ocfg,
                                     return_aux=True, proposals=given2)
    assert torch.equal(cap["labels"].cpu(), aux["labels"]), "anchor labels after sampling must be bit-exact"
    for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg"):
        assert np.isfinite(losses[k].item())
        put(k, abs(losses[k].item() / ref[k].item() - 1.0), NS)
    for name, p in model.named_parameters():
        if not name.startswith("DC_") and p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
    _report(f"{tag} {dtype} B={B}", errs, gates, info)
    return errs


@pytest.mark.parametrize("dtype,B", [("bf16x3", 2), ("fp32", 2), ("bf16x3", 8), ("f16x3", 2)])
def test_hot_yaml_teacher_and_student_at_600x1200(sfod, native, dtype, B):
    """BASELINE config #3; B = 8 is the batch bench.py times (the float comparison at that batch, once)."""
    _teacher_student_parity(sfod, HOT_YAML, om.Cfg(), dtype, B, plant=(60.0, 20.0), tag="VGG16 hot yaml")


@pytest.mark.parametrize("dtype,B", [("bf16x3", 8), ("fp32", 2)])
def test_hot_yaml_on_the_benchmarked_head_at_600x1200(sfod, native, dtype, B):
    """The same gate on the head bench.py times (engine/planted.py: cls_score x16, background bias bisected to ~20 pseudo
    labels per image), at the benchmark's batch in the benchmark's mode -- incl. the class probabilities themselves."""
    _teacher_student_parity(sfod, HOT_YAML, om.Cfg(), dtype, B, plant=None, tag="VGG16 hot yaml, bench head", bench_head="vgg")


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
def test_r101_yaml_on_the_benchmarked_head_at_600x1200(sfod, native, dtype):
    """bench.py --model r101's head (cls_score x4, bisected bias) in the mode it reports (and in fp32: behind ``sweep``)."""
    _teacher_student_parity(sfod, R101_YAML, om.Cfg.r101_c4(), dtype, 2, plant=None, tag="R101-C4 yaml, bench head",
                            bench_head="r101")


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3", "f16x3"])
def test_r101_yaml_teacher_and_student_at_600x1200(sfod, native, dtype):
    """BASELINE config #5 (r101_c4_cs_foggy_adaptive_teacher_source_free.yaml:1-28): ResNet-101-C4 trunk (frozen stem /
    res2, live BatchNorm res3 / res4), RPN on res4 at stride 16 with 4 sizes x 3 ratios = 12 anchors per location
    (38 x 75 x 12 = 34 200 anchors: the multi-chunk sort), ROIAlign at 1/16 on 1024 channels, FC_DIM 2048, 256 sampled
    ROIs per image -- same structure as the VGG test.  ``f16x3`` (what bench.py --model r101 reports) and ``fp32`` are
    this config's parity modes: losses / boxes at 1e-4.  ``bf16x3`` is held to the labelled tracking gates only (header)."""
    _teacher_student_parity(sfod, R101_YAML, om.Cfg.r101_c4(), dtype, 2, plant=(3.0, 4.0), tag="R101-C4 yaml",
                            tracking_only=(dtype == "bf16x3"))


@pytest.mark.parametrize("dtype", ["fp32", "f16x3"])
def test_r101_yaml_on_a_well_conditioned_weight_set_at_600x1200(sfod, native, dtype):
    """BASELINE config #5 once more, on weights whose conditioning is that of a trained residual network rather than of the
    default random initialisation: gamma of every bottleneck's last BatchNorm = 0.1.  The reference arithmetic's own error on
    this network (fp32 oracle vs fp64 oracle) is then < 1e-5 -- asserted -- so nothing is scaled by it: every box coordinate
    within 1e-4 of max(frame width, box extent), intermediates within 2e-5 relative L2, losses and boxes within 1e-4, discrete
    steps bit-exact -- the gates the VGG16 yaml is held to, in both of this config's parity modes."""
    _teacher_student_parity(sfod, R101_YAML, om.Cfg.r101_c4(), dtype, 2, plant=(12.0, 4.0), tag="R101-C4 yaml, last gamma 0.1",
                            last_gamma=0.1)


def _check_discrete_steps_on_captured_tensors(model, inputs, B, H, W, with_gt, native, ocfg=None):
    """One training-mode pass with the RPN's tensors captured; the oracle redoes every discrete step on them:
    proposals (decode -> top-k -> NMS keep, scores ==) and, with ground truth, the sampled anchor labels (==)."""
    ocfg = ocfg or om.Cfg()
    A = ocfg.num_anchors
    Hf, Wf = (-(-H // ocfg.stride), -(-W // ocfg.stride)) if ocfg.backbone == "resnet" else (H // ocfg.stride, W // ocfg.stride)
    rpn = model.proposal_generator
    cap = {}
    orig_props, orig_lf = rpn._proposals, rpn._loss_forward

    def cap_props(st, *a, **k):
        cap["rpn_out"] = st["rpn_out"].clone()
        cap["props"] = orig_props(st, *a, **k)
        return cap["props"]

    def cap_lf(*a, **k):
        loss, lab_state = orig_lf(*a, **k)
        cap["labels"] = lab_state[0].clone()
        return loss, lab_state
    rpn._proposals, rpn._loss_forward = cap_props, cap_lf
    g = torch.Generator().manual_seed(9)
    rpn_keys = torch.randint(0, 2 ** 31 - 1, (B, Hf * Wf * A), generator=g, dtype=torch.int64)
    rpn._forced_keys = rpn_keys.to(torch.int32).to(DEV)
    try:
        if with_gt:
            out = model(inputs, branch="supervised_target", batched=True) if hasattr(model, "elide") else model(inputs)
            losses = out[0] if isinstance(out, tuple) else out
            sum(v for k, v in losses.items() if k != "loss_bpc").backward()
        else:
            with torch.no_grad():
                model(inputs, branch="unsup_data_weak", batched=True)
            losses = {}
    finally:
        rpn._proposals, rpn._loss_forward = orig_props, orig_lf
        del rpn._forced_keys
    torch.cuda.synchronize()
    for k, v in losses.items():
        assert np.isfinite(v.item()), k
    anchors = om.anchors_for((Hf, Wf), ocfg)
    out = cap["rpn_out"].cpu().view(B, Hf * Wf, -1)
    logits = out[:, :, :A].reshape(B, Hf * Wf * A)
    deltas = out[:, :, A:5 * A].reshape(B, Hf * Wf * A, 4)
    sizes = [(H, W)] * B
    ref = om.rpn_proposals(anchors, logits, deltas, sizes, ocfg, training=True)
    props = cap["props"]
    for b in range(B):
        n = props.count[b].item()
        assert n == len(ref[b][0]) and 0 < n <= ocfg.rpn_post_topk_train
        assert torch.equal(props.logits[b, :n].cpu(), ref[b][1]), "NMS keep set / order differs"
        torch.testing.assert_close(props.boxes[b, :n].cpu(), ref[b][0], rtol=1e-5, atol=1e-3)
    if with_gt:
        labels, _ = om.rpn_label_anchors(anchors, [d["instances"].gt_boxes.tensor.cpu() for d in inputs], list(rpn_keys), ocfg)
        assert torch.equal(cap["labels"].cpu(), labels), "anchor labels after sampling must be bit-exact"
        for name, p in model.named_parameters():
            if not name.startswith("DC_") and p.requires_grad:
                assert p.grad is not None and torch.isfinite(p.grad).all(), name
    return losses


def _with_gt(inputs, S, n, seed):
    g = torch.Generator().manual_seed(seed)
    for d in inputs:
        H, W = d["height"], d["width"]
        xy = torch.rand(n, 2, generator=g) * torch.tensor([W * 0.7, H * 0.7])
        wh = torch.rand(n, 2, generator=g) * torch.tensor([W * 0.25, H * 0.25]) + 24
        inst = S.Instances((H, W))
        inst.gt_boxes = S.Boxes(torch.cat([xy, xy + wh], 1))
        inst.gt_classes = torch.randint(0, 8, (n,), generator=g)
        d["instances"] = inst
    return inputs


SRC_YAML = os.path.join(os.path.dirname(GOLDEN), "..", "configs", "faster_rcnn_VGG_cityscapes_source_new.yaml")


@pytest.mark.parametrize("which", ["source_b8_600x1200", "hot_b8_600x1200", "source_b2_1024x2048", "hot_teacher_b2_1024x2048"])
def test_configs_at_their_real_sizes(sfod, native, which):
    """BASELINE configs #2 / #3 at the sizes bench.py runs them (B = 8 frames of 600x1200; 1024x2048 tensors with
    30 720 anchors per image = the > 16 384-key path of the segmented sort), default bf16x3 mode: finite losses and
    gradients, proposal counts, and the discrete steps bit-exact against the oracle on the captured tensors."""
    S = sfod.structures
    yaml = SRC_YAML if which.startswith("source") else HOT_YAML
    B = 8 if "b8" in which else 2
    H, W = (600, 1200) if "600x1200" in which else (1024, 2048)
    cfg = sfod.config.setup_cfg(yaml, ["OUTPUT_DIR", ""])
    assert cfg.SFOD.COMPUTE_DTYPE == "bf16x3"
    torch.manual_seed(3)
    model = sfod.modeling.build_model(cfg).train()
    inputs = _frames(B, H, W, seed=33)
    teacher_only = "teacher" in which
    if not teacher_only:
        inputs = _with_gt(inputs, S, 12, seed=4)
    losses = _check_discrete_steps_on_captured_tensors(model, inputs, B, H, W, not teacher_only, native)
    if not teacher_only:
        assert {"loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"} <= set(losses)
        assert 0.1 < losses["loss_rpn_cls"].item() < 2.0 and 1.0 < losses["loss_cls"].item() < 4.0     # ln 2, ln 9 at init


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
@pytest.mark.parametrize("which", ["r101_b8_600x1200", "r101_teacher_b2_1024x2048"])
def test_r101_config_at_its_real_sizes(sfod, native, which, dtype):
    """BASELINE config #5 at the batch bench.py --model r101 runs (B = 8 frames of 600x1200: 34 200 anchors per image)
    and on 1024x2048 tensors (64 x 128 x 12 = 98 304 anchors per image: six 16 384-key chunks + merge passes of the
    segmented sort), in its parity modes (f16x3: what bench.py reports; fp32): finite losses and gradients, frozen stages
    without gradients, and the discrete steps bit-exact against the oracle on the captured tensors."""
    S = sfod.structures
    B = 8 if "b8" in which else 2
    H, W = (600, 1200) if "600x1200" in which else (1024, 2048)
    cfg = sfod.config.setup_cfg(R101_YAML, ["OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", dtype])
    torch.manual_seed(3)
    model = sfod.modeling.build_model(cfg).train()
    inputs = _frames(B, H, W, seed=33)
    teacher_only = "teacher" in which
    if not teacher_only:
        inputs = _with_gt(inputs, S, 12, seed=4)
    losses = _check_discrete_steps_on_captured_tensors(model, inputs, B, H, W, not teacher_only, native,
                                                       ocfg=om.Cfg.r101_c4())
    if not teacher_only:
        assert {"loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"} <= set(losses)
        for n, p in model.named_parameters():
            if n.startswith(("backbone.stem", "backbone.res2")):
                assert p.grad is None and not p.requires_grad, n
