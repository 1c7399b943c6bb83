"""Reference-OWNED glue of the hot path, pinned by fixtures the reference's own functions produced.

``tests/golden/glue_ref.npz`` / ``config_ref.json`` were recorded by ``oracle/gen_golden.py::gen_glue``, which loads the
reference's files by path behind ``oracle/ref_stub/hook.py`` and RUNS (not restates)

  a6   SourceFreeFastRCNNOutputLayers.fast_rcnn_inference_new        daod/modeling/roi_heads/source_free_fast_rcnn.py:38-147
  a7   threshold_bbox / process_pseudo_label                         daod/engine/trainers/source_free_adaptive_teacher.py:150-183,256-280
  a9   _update_teacher_model                                         ... :583-603
  a13  AspectRatioGroupedSemiSupDatasetTwoCropSourceFree.__iter__    daod/data/common.py:199-228
  a4   PseudoLabRPN.forward (layout + second loss weight)            daod/modeling/proposal_generator/rpn.py:16-58
  a11  reset_bn_stats / recursive_traversal                          daod/engine/trainers/base.py:318-328
  b    add_config                                                    daod/config.py:8-142
  a1   vgg_backbone backward: parameter gradients                    daod/modeling/meta_arch/vgg.py (vgg_ref.npz ``g/*``)

Here (no GPU): the oracle's restatements and the product's host-side twins (pure torch / Python) against those vectors.
The HIP kernels meet the same vectors in tests/test_gpu_glue.py.  ``Boxes.clip`` inside a6 is d2's (restated in the
stub): the fixture pins the reference's use of it, not d2's arithmetic.
"""
import json
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from util_weights import reference_vgg_state

from oracle import box_ops as B
from oracle import model as om


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "glue_ref.npz"), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


# ---- a6 ----------------------------------------------------------------------------------------------------------------
def test_fast_rcnn_inference_new_restatement_and_product_twin_equal_the_reference(fx, sfod):
    layers = object.__new__(sfod.modeling.roi_heads.SourceFreeFastRCNNOutputLayers)
    for i in range(2):
        bx, sc, size = T(fx[f"frcnn_boxes_in_{i}"]), T(fx[f"frcnn_scores_in_{i}"]), tuple(int(v) for v in fx[f"frcnn_size_{i}"])
        assert not torch.isfinite(bx).all() and not torch.isfinite(sc).all()      # the case has non-finite rows
        assert (sc[:, :-1] == 0).any()                                           # ... and exact zeros
        ref = {k: T(fx[f"frcnn_{k}_{i}"]) for k in ("pred_boxes", "scores", "pred_classes", "row")}
        o = om.frcnn_inference_new_single(bx.clone(), sc.clone(), size)
        assert torch.equal(o["boxes"], ref["pred_boxes"]) and torch.equal(o["scores"], ref["scores"])
        assert torch.equal(o["classes"], ref["pred_classes"]) and torch.equal(o["roi_idx"], ref["row"])
        inst, row = layers.fast_rcnn_inference_single_image_new(bx.clone(), sc.clone(), size, 0.05, 0.5, 100, None)
        assert torch.equal(inst.pred_boxes.tensor, ref["pred_boxes"]) and torch.equal(inst.scores, ref["scores"])
        assert torch.equal(inst.pred_classes, ref["pred_classes"]) and torch.equal(row, ref["row"])
        assert inst.image_size == size
        # what the fixture says about the function: rows index the finite survivors, so they stay below their number
        n_finite = int((torch.isfinite(bx).all(1) & torch.isfinite(sc).all(1)).sum())
        assert int(ref["row"].max()) == n_finite - 1 < len(bx) - 1
        assert (ref["pred_boxes"][:, 0::2] <= size[1]).all() and (ref["pred_boxes"][:, 1::2] <= size[0]).all()
        assert (ref["pred_boxes"] >= 0).all() and (ref["scores"] > 0).all()
    # class-agnostic regression: one box per row, shared by its classes
    bx, sc, size = T(fx["frcnn_agn_boxes_in"]), T(fx["frcnn_scores_in_0"]), tuple(int(v) for v in fx["frcnn_size_0"])
    o = om.frcnn_inference_new_single(bx.clone(), sc.clone(), size)
    inst, row = layers.fast_rcnn_inference_single_image_new(bx.clone(), sc.clone(), size)
    for got_b, got_s, got_c, got_r in ((o["boxes"], o["scores"], o["classes"], o["roi_idx"]),
                                       (inst.pred_boxes.tensor, inst.scores, inst.pred_classes, row)):
        assert torch.equal(got_b, T(fx["frcnn_agn_pred_boxes"])) and torch.equal(got_s, T(fx["frcnn_agn_scores"]))
        assert torch.equal(got_c, T(fx["frcnn_agn_pred_classes"])) and torch.equal(got_r, T(fx["frcnn_agn_row"]))
    # the list-level entry point, with the reference's argument order (boxes, scores, image_shapes, ...)
    res, kept = layers.fast_rcnn_inference_new([T(fx[f"frcnn_boxes_in_{i}"]) for i in range(2)],
                                               [T(fx[f"frcnn_scores_in_{i}"]) for i in range(2)],
                                               [tuple(int(v) for v in fx[f"frcnn_size_{i}"]) for i in range(2)],
                                               0.05, 0.5, 100, [None, None])
    assert [len(r) for r in res] == [len(fx["frcnn_row_0"]), len(fx["frcnn_row_1"])]
    assert torch.equal(kept[1], T(fx["frcnn_row_1"]))


# ---- a7 ----------------------------------------------------------------------------------------------------------------
def test_threshold_bbox_and_process_pseudo_label_equal_the_reference(fx, sfod):
    thr = float(fx["thr"])
    tr_mod = sfod.engine.trainer
    S = sfod.structures
    stub = object.__new__(tr_mod.SourceFreeAdaptiveTeacherTrainer)
    insts = []
    for i in range(3):
        p = S.Instances((600, 1200))
        p.pred_boxes = S.Boxes(T(fx[f"pl_in_boxes_{i}"]))
        p.scores = T(fx[f"pl_in_scores_{i}"])
        p.pred_classes = T(fx[f"pl_in_classes_{i}"])
        insts.append(p)
        # oracle
        o = om.threshold_bbox({"boxes": p.pred_boxes.tensor, "scores": p.scores, "classes": p.pred_classes}, thr)
        assert torch.equal(o["gt_boxes"], T(fx[f"pl_gt_boxes_{i}"])) and torch.equal(o["gt_classes"], T(fx[f"pl_gt_classes_{i}"]))
        assert torch.equal(o["scores"], T(fx[f"pl_scores_{i}"]))
    out, mean_n = tr_mod.SourceFreeAdaptiveTeacherTrainer.process_pseudo_label(stub, insts, thr, "roih", "thresholding")
    assert mean_n == float(fx["pl_mean_count"])
    for i, q in enumerate(out):
        assert sorted(q.get_fields().keys()) == list(fx[f"pl_fields_{i}"])
        assert torch.equal(q.gt_boxes.tensor, T(fx[f"pl_gt_boxes_{i}"])) and torch.equal(q.gt_classes, T(fx[f"pl_gt_classes_{i}"]))
        assert torch.equal(q.scores, T(fx[f"pl_scores_{i}"]))
    # strict '>' in float32: a score of exactly float32(0.8) is NOT a pseudo label, one ulp above is
    t32 = np.float32(thr)
    s0, kept0 = fx["pl_in_scores_0"], fx["pl_scores_0"]
    assert t32 in s0 and t32 not in kept0 and np.nextafter(t32, np.float32(1)) in kept0
    assert np.nextafter(t32, np.float32(0)) in s0 and kept0.min() > t32
    assert len(fx["pl_scores_1"]) == 0 and len(fx["pl_in_scores_1"]) == 0        # an image without detections
    # RPN flavour
    rp = S.Instances((600, 1200))
    rp.proposal_boxes = S.Boxes(T(fx["rpn_in_boxes"]))
    rp.objectness_logits = T(fx["rpn_in_logits"])
    q = tr_mod.threshold_bbox(rp, thres=thr, proposal_type="rpn")
    assert sorted(q.get_fields().keys()) == list(fx["rpn_fields"])
    assert torch.equal(q.gt_boxes.tensor, T(fx["rpn_gt_boxes"])) and torch.equal(q.objectness_logits, T(fx["rpn_logits"]))
    with pytest.raises(ValueError) as e:
        tr_mod.SourceFreeAdaptiveTeacherTrainer.process_pseudo_label(stub, insts, thr, "roih", "no_such_method")
    assert str(e.value) == str(fx["pl_error"])


# ---- a9 ----------------------------------------------------------------------------------------------------------------
def _ema_states(fx, prefix):
    return {str(k): T(fx[prefix + str(k)]).clone() for k in fx["ema_keys"]}


def test_ema_update_restatement_equals_the_reference_including_the_int64_counters(fx):
    student, teacher = _ema_states(fx, "ema_s/"), _ema_states(fx, "ema_t0/")
    int_keys = [k for k, v in teacher.items() if v.dtype == torch.int64]
    assert int_keys == ["1.num_batches_tracked", "3.num_batches_tracked"]
    for step, ((cs, ct), keep) in enumerate(zip(fx["ema_counters"].tolist(), fx["ema_keep"].tolist())):
        student["1.num_batches_tracked"].fill_(cs)
        teacher["1.num_batches_tracked"].fill_(ct)
        student["3.num_batches_tracked"].fill_(cs + 1)
        teacher["3.num_batches_tracked"].fill_(ct + 2)
        om.ema_update(teacher, student, keep)
        for k, v in teacher.items():
            assert torch.equal(v, T(fx[f"ema_t{step + 1}/{k}"])), (step, k)
    # what the reference's arithmetic does to the counter: int64 * python float -> float32, truncated by the copy back
    assert int(fx["ema_t1/1.num_batches_tracked"]) == 6            # 3 * 0.0004 + 7 * 0.9996 = 6.9984 -> 6
    assert int(fx["ema_t5/1.num_batches_tracked"]) == int(np.float32(np.float32(5) * np.float32(1 - 0.9996))
                                                          + np.float32(np.float32(123456789) * np.float32(0.9996)))
    # DDP: the student's keys carry "module."; same numbers
    s2 = {k[len("ema_ddp_s/"):]: T(fx[k]).clone() for k in fx.files if k.startswith("ema_ddp_s/")}
    t2 = {k[len("ema_ddp_t0/"):]: T(fx[k]).clone() for k in fx.files if k.startswith("ema_ddp_t0/")}
    om.ema_update(t2, s2, 0.9996)
    for k, v in t2.items():
        assert torch.equal(v, T(fx["ema_ddp_t1/" + k])), k
    with pytest.raises(Exception) as e:
        om.ema_update(dict(teacher), {k: v for k, v in student.items() if not k.startswith("4.")}, 0.9996)
    assert str(e.value) == str(fx["ema_error"]) == "4.weight is not found in student model"


# ---- a13 ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bs", [1, 2, 3, 4])
def test_two_crop_loader_batches_in_the_reference_bucket_order(fx, sfod, bs):
    """The product's loader on the recorded (width, height) stream: the same images in the same batches in the same
    order as ``AspectRatioGroupedSemiSupDatasetTwoCropSourceFree`` yielded them (strong ids even, weak ids odd)."""
    wh = fx["bucket_wh"]
    cfg = sfod.config.setup_cfg(os.path.join(ROOT, "configs", "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml"),
                                ["SOLVER.IMS_PER_BATCH_TARGET", str(bs), "INPUT.RANDOM_FLIP", "none", "MODEL.DEVICE", "cpu"])
    items = [{"image": torch.zeros(3, 4, 4, dtype=torch.uint8), "boxes": torch.zeros(0, 4), "classes": torch.zeros(0, dtype=torch.int64),
              "height": int(h), "width": int(w), "image_id": i, "file_name": str(i), "size": (4, 4)} for i, (w, h) in enumerate(wh)]
    ds = types.SimpleNamespace(items=items, size=(4, 4), __len__=lambda: len(items))

    class DS(list):
        items, size = ds.items, ds.size
    loader = sfod.data.TwoCropLoader(cfg, torch.device("cpu"), dataset=DS(items))
    loader.sampler = iter(range(len(items)))
    ref_w = fx[f"bucket_weak_ids_b{bs}"]
    assert np.array_equal(fx[f"bucket_strong_ids_b{bs}"] + 1, ref_w)      # the two crops of one image travel together
    got = []
    for _ in range(len(ref_w)):
        strong, weak = next(loader)
        assert [d["image_id"] for d in strong] == [d["image_id"] for d in weak]
        got.append([2 * d["image_id"] + 1 for d in weak])
    assert np.array_equal(np.array(got).reshape(-1, bs), ref_w)
    # one aspect class per batch (w > h | otherwise)
    for row in ref_w:
        cls = {bool(wh[(i - 1) // 2][0] > wh[(i - 1) // 2][1]) for i in row}
        assert len(cls) == 1


# ---- a4 ----------------------------------------------------------------------------------------------------------------
def test_rpn_flatten_layout_and_second_loss_weight_equal_the_reference(fx):
    lg, dl = om.rpn_flatten(T(fx["rpn_glue_logits_in"]), T(fx["rpn_glue_deltas_in"]))
    assert torch.equal(lg, T(fx["rpn_glue_logits_flat"])) and torch.equal(dl, T(fx["rpn_glue_deltas_flat"]))
    # element (n, a, y, x) lands at (n, (y * W + x) * A + a); delta channel 4a + c at [.., c]
    N, A, H, W = fx["rpn_glue_logits_in"].shape
    n, a, y, x, c = 1, 7, 2, 3, 2
    assert fx["rpn_glue_logits_flat"][n, (y * W + x) * A + a] == fx["rpn_glue_logits_in"][n, a, y, x]
    assert fx["rpn_glue_deltas_flat"][n, (y * W + x) * A + a, c] == fx["rpn_glue_deltas_in"][n, 4 * a + c, y, x]
    # PseudoLabRPN.forward multiplies what RPN.losses() returned by loss_weight AGAIN (rpn.py:49), unknown keys by 1:
    # the stub losses() returned cls 2, loc 3, other 7 with loss_weight {cls 1.5, loc 0.5}
    assert list(fx["rpn_glue_loss_keys"]) == ["loss_rpn_cls", "loss_rpn_loc", "other"]
    assert fx["rpn_glue_loss_vals"].tolist() == [3.0, 1.5, 7.0]
    # losses: (training and compute_loss) or compute_val_loss -- [train, eval, eval+val_loss, train+compute_loss False]
    assert fx["rpn_glue_branches"].tolist() == [3, 0, 3, 0]


# ---- a11 ---------------------------------------------------------------------------------------------------------------
def test_adabn_reset_equals_the_reference(fx, sfod):
    keys = [str(k) for k in fx["adabn_keys_after"]]
    assert keys == [str(k) for k in fx["adabn_keys_before"]]          # the state-dict surface survives buffer -> Parameter
    assert bool(fx["adabn_weight_untouched"])
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4),
                              torch.nn.Sequential(torch.nn.Conv2d(4, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.ReLU()))
    sd = {k[len("adabn_w/"):]: T(fx[k]) for k in fx.files if k.startswith("adabn_w/")}
    net.load_state_dict(sd, strict=False)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_()
                m.running_var.uniform_(0.5, 2.0)
                m.num_batches_tracked.fill_(41)
    sfod.engine.trainer.reset_bn_stats(net)
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert torch.equal(v, T(fx["adabn_after/" + k])), k      # mean 0, var 1, the counter is NOT reset (41)
    net.train()
    with torch.no_grad():
        net(T(fx["adabn_x"]))
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            torch.testing.assert_close(v, T(fx["adabn_fwd/" + k]), rtol=1e-6, atol=1e-7)
    assert int(net[1].num_batches_tracked) == 42


# ---- b -----------------------------------------------------------------------------------------------------------------
def test_config_defaults_equal_what_add_config_assigns(sfod):
    with open(os.path.join(GOLDEN, "config_ref.json")) as f:
        ref = json.load(f)["assigned"]
    cfg = sfod.config.get_cfg()          # d2's defaults ...
    sfod.config.add_config(cfg)          # ... + the product's twin of the reference's function
    n = [0]

    def walk(node, r, path):
        for k, v in r.items():
            assert k in node, "missing config key " + ".".join(path + [k])
            if isinstance(v, dict):
                walk(node[k], v, path + [k])
            else:
                got = node[k]
                got = list(got) if isinstance(got, (tuple, list)) else got
                assert got == v, (".".join(path + [k]), got, v)
                n[0] += 1
    walk(cfg, ref, [])
    assert n[0] == 43


# ---- a1 backward -------------------------------------------------------------------------------------------------------
def test_oracle_vgg_parameter_gradients_equal_the_reference_backward():
    fx = np.load(os.path.join(GOLDEN, "vgg_ref.npz"), allow_pickle=False)
    sd = om.clone_state(reference_vgg_state(int(fx["seed"])), requires_grad=True)
    x = T(fx["input"]).clone().requires_grad_(True)
    feats = om.vgg_forward(sd, x, om.Cfg(), training=True, return_all=True)
    gr = torch.Generator().manual_seed(int(fx["bwd_seed"]))
    loss = 0
    for i in fx["bwd_stages"].tolist():
        loss = loss + (feats[f"vgg{i}"] * torch.randn(feats[f"vgg{i}"].shape, generator=gr)).sum()
    loss.backward()
    torch.testing.assert_close(x.grad, T(fx["bwd_input_grad"]), rtol=1e-3, atol=1e-5)
    n = 0
    for k in fx.files:
        if not k.startswith("g/"):
            continue
        name = k[2:]
        g = sd["backbone." + name].grad.flatten()
        ref, stride = T(fx[k]), int(fx["gstride/" + name])
        if name.endswith("bias") and name.split(".")[1] in ("0", "3", "6"):
            # a conv bias in front of train-mode BatchNorm: analytically zero, the reference's autograd leaves rounding noise
            assert float(fx["gnorm/" + name]) < 1e-2 and g.norm() < 1e-2
            continue
        err = ((g[::stride] - ref).double().norm() / ref.double().norm()).item()
        assert err < 2e-4, (name, err)
        np.testing.assert_allclose(g.double().norm().item(), float(fx["gnorm/" + name]), rtol=1e-4)
        n += 1
    assert n == 39


# ---- third-party cross-check of the unpinned half ----------------------------------------------------------------------
def test_pairwise_iou_against_the_independent_box_iou_in_transformers():
    """detectron2 / torchvision are absent, so ``oracle.box_ops.pairwise_iou`` cannot be pinned to them here;
    ``transformers`` (installed) carries its own port of torchvision's ``box_iou`` -- an implementation written by
    somebody else.  d2's differs from it only in the ``where(inter > 0)`` guard (0 instead of 0/0 for two empty boxes)."""
    loss_mod = pytest.importorskip("transformers.loss.loss_for_object_detection")
    g = torch.Generator().manual_seed(3)
    xy = torch.rand(300, 2, generator=g) * 900
    wh = torch.rand(300, 2, generator=g) * 200 + 1
    a = torch.cat([xy, xy + wh], 1)
    xy = torch.rand(120, 2, generator=g) * 900
    wh = torch.rand(120, 2, generator=g) * 300 + 1
    b = torch.cat([xy, xy + wh], 1)
    b[:10] = a[:10]                       # IoU exactly 1
    b[10:20, :2] = a[10:20, 2:]           # touching corners: intersection 0
    b[10:20, 2:] = a[10:20, 2:] + 50
    ref, _ = loss_mod.box_iou(a, b)
    got = B.pairwise_iou(a, b)
    assert torch.equal(got, ref)
    assert (got[torch.arange(10), torch.arange(10)] == 1).all() and (got[torch.arange(10, 20), torch.arange(10, 20)] == 0).all()
    # box_area agrees too
    assert torch.equal(B.box_area(a), loss_mod.box_area(a))


# ---- a5 ----------------------------------------------------------------------------------------------------------------
def test_roi_label_and_sample_equals_the_reference_method(fx):
    """``label_and_sample_proposals`` (source_free_adaptive_teacher_roi_heads.py:165-215) run by the generator on the
    oracle's Detectron2 primitives: the oracle's one-piece restatement returns the same rows in the same order -- proposals
    (ground truth appended), their classes (background = K), the matched ground-truth boxes (zeros for the image without
    ground truth); only ``gt_*`` fields travel to the proposals; the logged means and their key names."""
    K, batch, frac = int(fx["roi_K"]), int(fx["roi_batch"]), float(fx["roi_frac"])
    cfg = om.Cfg(roi_batch=batch, roi_pos_frac=frac, num_classes=K)
    props = [(T(fx[f"roi_in_boxes_{i}"]), T(fx[f"roi_in_logits_{i}"])) for i in range(3)]
    gtb = [T(fx[f"roi_gt_boxes_{i}"]).reshape(-1, 4) for i in range(3)]
    gtc = [T(fx[f"roi_gt_classes_{i}"]) for i in range(3)]
    keys = [T(fx[f"roi_keys_{i}"]) for i in range(3)]
    res = om.roi_label_and_sample(props, gtb, gtc, keys, cfg)
    nfg, nbg = [], []
    for i, r in enumerate(res):
        assert torch.equal(r["boxes"], T(fx[f"roi_out_boxes_{i}"])), i
        assert torch.equal(r["gt_classes"], T(fx[f"roi_out_gt_classes_{i}"])), i
        assert torch.equal(r["gt_boxes"], T(fx[f"roi_out_gt_boxes_{i}"])), i
        assert list(fx[f"roi_out_fields_{i}"]) == ["gt_boxes", "gt_classes", "objectness_logits", "proposal_boxes"]
        nbg.append(int((r["gt_classes"] == K).sum()))
        nfg.append(len(r["gt_classes"]) - nbg[-1])
    assert len(gtb[1]) == 0 and (fx["roi_out_gt_boxes_1"] == 0).all() and (fx["roi_out_gt_classes_1"] == K).all()
    assert list(fx["roi_scalar_keys"]) == ["roi_head/num_target_bg_samples_supervised_target",
                                           "roi_head/num_target_fg_samples_supervised_target"]
    np.testing.assert_allclose(fx["roi_scalar_vals"], [np.mean(nbg), np.mean(nfg)])


def test_roi_heads_forward_control_flow_of_the_reference(fx, sfod):
    """``forward`` / ``_forward_box`` (:68-163) on a stub: return arity per (training, compute_loss, compute_val_loss),
    the calls of the box branch in order, ``proposal_append_gt`` switched off for the val-loss pass only, and the sampled
    proposals' boxes overwritten with the gt-class predictions AFTER the losses and BEFORE ``convert_bbox_scores``.  The
    product's ROI heads implement exactly these rules (simple-sfod_amd/modeling/roi_heads.py::forward; the 4-tuple /
    2-tuple shapes are asserted on the device in tests/test_gpu_model.py)."""
    flags = [tuple(bool(v) for v in r) for r in fx["roi_fwd_flags"]]
    train_path = "box_pooler|box_head|box_predictor|losses|predict_boxes_for_gt_classes|convert_bbox_scores"
    infer_path = "box_pooler|box_head|box_predictor|inference"
    for (training, cl, cvl), arity, path, seen in zip(flags, fx["roi_fwd_arity"], fx["roi_fwd_paths"], fx["roi_fwd_append_gt_seen"]):
        losses = (training and cl) or cvl
        assert int(arity) == (4 if losses else 2) and str(path) == (train_path if losses else infer_path)
        # GT is appended only in the training-loss pass; the val-loss pass samples from the proposals alone
        assert int(seen) == (1 if (training and cl) else (0 if cvl else -1))
    np.testing.assert_allclose(fx["roi_fwd_box_sum_at_convert"] - fx["roi_fwd_box_sum_at_losses"], 4.0 * fx["roi_fwd_rows"])
    assert np.array_equal(fx["roi_fwd_box_sum_returned"], fx["roi_fwd_box_sum_at_convert"])
    import inspect
    src = inspect.getsource(sfod.modeling.roi_heads.StandardROIHeads.forward)
    assert "compute_val_loss and not (self.training and compute_loss)" in src      # the same append-gt rule


# ---- a8 ----------------------------------------------------------------------------------------------------------------
def test_run_step_orchestration_and_loss_weights_equal_the_reference(fx, sfod):
    """``run_step`` (source_free_adaptive_teacher.py:335-581) was RUN on a stub ``self`` whose teacher / student are recorders
    (oracle/gen_golden.py::gen_run_step), for five combinations of the domain-classifier switches and loss weights.  The
    product's ``loss_weight`` gives every key the weight the reference's ``losses.backward()`` left on its loss leaf; the
    recorded orchestration is what ``engine/trainer.py::run_step`` does: teacher first (on data without labels, branch
    ``unsup_data_weak``), then the student on ``supervised_target`` with the thresholded detections attached, the domain pass
    only when the classifier is enabled, one ``zero_grad`` / ``step``, the four logged scalars."""
    from types import SimpleNamespace as NS
    tr = sfod.engine.trainer
    thr = 0.8
    n_roih = [int((fx[f"rs_det_scores_{i}"] > np.float32(thr)).sum()) for i in range(2)]
    n_rpn = [int((fx[f"rs_rpn_logits_{i}"] > np.float32(thr)).sum()) for i in range(2)]
    mean_conf = np.mean([fx[f"rs_det_scores_{i}"].mean() for i in range(2)])
    for ci, (dc_on, dc_img, dc_ins, unsup_w, dis_w, wsa) in enumerate(fx["rs_combos"].tolist()):
        pre = f"rs{ci}_"
        cfg = NS(SEMISUPNET=NS(UNSUP_LOSS_WEIGHT=unsup_w, DIS_LOSS_WEIGHT=dis_w),
                 DOMAIN_CLASSIFIER=NS(ENABLED=bool(dc_on), IMAGE=bool(dc_img), INSTANCE=bool(dc_ins)))
        for key, w in zip(fx[pre + "weight_keys"], fx[pre + "weights"]):
            assert np.float32(tr.loss_weight(str(key), cfg)) == np.float32(w), (ci, key, w)
        order = [str(c) for c in fx[pre + "call_order"]]
        assert order == ["teacher:unsup_data_weak", "student:supervised_target"] + (["student:domain_classifier"] if dc_on else [])
        assert not fx[pre + "teacher_saw_instances"].any()                       # labels removed before the teacher pass
        assert fx[pre + "student_label_counts"].tolist() == n_roih              # score > 0.8, per image
        assert list(fx[pre + "student_label_fields"]) == ["gt_boxes", "gt_classes", "scores"]
        # without WEAK_STRONG_AUGMENT the student's list is a deep copy of the WEAK list (tags k*), else the strong one
        assert list(fx[pre + "student_tags"]) == (["q0", "q1"] if wsa else ["k0", "k1"])
        if dc_on:
            assert "image_unlabeled" in fx[pre + "domain_keys"] and "instances_unlabeled" in fx[pre + "domain_keys"]
        sc = dict(zip([str(k) for k in fx[pre + "scalar_keys"]], fx[pre + "scalar_vals"]))
        assert sorted(sc) == ["calibration/bpc_loss", "roi_head/mean_confidence", "roi_head/num_pseudo_proposals", "rpn/num_pseudo_proposals"]
        np.testing.assert_allclose(sc["roi_head/mean_confidence"], mean_conf, rtol=1e-6)
        assert sc["roi_head/num_pseudo_proposals"] == np.mean(n_roih) and sc["rpn/num_pseudo_proposals"] == np.mean(n_rpn)
        assert sc["calibration/bpc_loss"] == 0.0                                  # the weighted (x 0) value is what is logged
        # what reaches _write_metrics: the WEIGHTED losses + data_time
        mk = [str(k) for k in fx[pre + "metrics_keys"]]
        assert mk == sorted(["data_time"] + [str(k) for k in fx[pre + "weight_keys"]])
        assert fx[pre + "opt_calls"].tolist() == [1, 1] and int(fx[pre + "trainer_iter"]) == 7
    # the product's run_step builds its key list and weights the same way (one source of truth)
    import inspect
    src = inspect.getsource(tr.SourceFreeAdaptiveTeacherTrainer.run_step)
    assert "loss_weight(key, cfg)" in src and 'branch="supervised_target"' in src and 'branch="domain_classifier"' in src


# ---- a3 ----------------------------------------------------------------------------------------------------------------
def test_meta_arch_branches_call_their_submodules_like_the_reference(fx, sfod, monkeypatch):
    """``SourceFreeAdaptiveTeacherGeneralizedRCNN.forward`` of the REFERENCE was run on a stub whose sub-modules record their
    calls (oracle/gen_golden.py::gen_meta_arch); here the PRODUCT's forward runs the same way (unbound, on a stub, CPU) and
    must produce the same call sequence with the same flags, the same tuple arity and the same loss keys per branch.  Known,
    documented differences: the product computes BPC through the ROI heads' ``InstanceProposals.bpc_loss`` (one fused
    launch) instead of a module-level function; ``supervised`` (the with-source trainer's branch, out of scope) does not add
    ``loss_DC_img_s * 0.001``; with SFOD.ELIDE_DEAD_BRANCHES the second, loss-free ROI pass of ``supervised_target`` is skipped."""
    import types
    ma = sfod.modeling.meta_arch
    cls = ma.SourceFreeAdaptiveTeacherGeneralizedRCNN
    S = sfod.structures
    trace = []

    def rpn(images, features, gt=None, compute_loss=True, compute_val_loss=False, as_instances=True):
        trace.append("rpn(images=%s,gt=%d,compute_loss=%d)" % (images.tag, gt is not None, compute_loss))
        return types.SimpleNamespace(boxes=None, count=None), {"loss_rpn_cls": torch.tensor(1.0), "loss_rpn_loc": torch.tensor(2.0)}

    class Roi:
        in_features = ["vgg4"]

        def __call__(self, images, features, proposals, targets=None, compute_loss=True, branch="", compute_val_loss=False, as_instances=True):
            trace.append("roi(images=%s,targets=%d,compute_loss=%d,branch=%s)" % (images.tag, targets is not None, compute_loss, branch))
            if compute_loss:
                ip = types.SimpleNamespace(bpc_loss=lambda gt: trace.append("bpc_loss(K=8,gt=%d,props=instance_proposals)" % (gt is not None)) or torch.tensor(0.25))
                return "samples", {"loss_cls": torch.tensor(3.0), "loss_box_reg": torch.tensor(4.0)}, None, ip
            return "pred_instances", "predictions"

        def label_and_sample_proposals(self, props, gt, branch=""):
            trace.append("roi(images=%s,targets=1,compute_loss=1,branch=%s)" % (self._tag, branch))
            return {"rois": "rois"}
    roi = Roi()

    def make(elide, ins_dc):
        stub = types.SimpleNamespace(training=True, device=torch.device("cpu"), elide=elide, ins_dc=ins_dc, dis_type="vgg4",
                                     proposal_generator=rpn, roi_heads=roi, DC_img="DC_img", DC_ins="DC_ins")
        stub._images_and_features = lambda b: (trace.append("backbone(x)") or types.SimpleNamespace(tag="k"), {"vgg4": "feat"})
        tags = iter(["s", "t"])

        def pre(b, key="image"):
            return types.SimpleNamespace(tag=next(tags), key=key)
        stub.preprocess_image = pre
        stub._features = lambda images: trace.append("backbone(x%s)" % images.tag) or {"vgg4": "feat" + images.tag}
        stub._forward_domain_classifier = lambda b: cls._forward_domain_classifier(stub, b)
        return stub

    def dc_img_loss(dc, feat, label):
        return torch.tensor(float(label))

    def dc_ins_loss(rh, dc, feat, rois, label, training=True):
        trace.append("instance_dc_loss(box_features,levels,label=%d)" % label)
        return torch.tensor(0.5 + label)
    monkeypatch.setattr(ma, "dc_img_loss", dc_img_loss)
    monkeypatch.setattr(ma, "dc_ins_loss", dc_ins_loss)

    def inst():
        i = S.Instances((64, 64))
        i.gt_boxes = S.Boxes(torch.tensor([[1.0, 2.0, 30.0, 40.0]]))
        i.gt_classes = torch.tensor([3])
        return i
    with_gt = [{"image": 0, "instances": inst(), "instances_unlabeled": inst(), "image_unlabeled": 0}]
    without = [{"image": 0, "image_unlabeled": 0}]
    cases = {str(n): i for i, n in enumerate(fx["ma_cases"])}
    # ---- supervised_target, every branch of the reference executed (ELIDE False) -------------------------------------------
    ci = cases["supervised_target|gt=1|ins_dc=0"]
    del trace[:]
    r = cls.forward(make(False, False), with_gt, branch="supervised_target")
    assert trace == [str(t) for t in fx[f"ma{ci}_trace"]]
    assert len(r) == int(fx[f"ma{ci}_arity"]) == 4 and sorted(r[0]) == [str(k) for k in fx[f"ma{ci}_loss_keys"]]
    assert [float(r[0][k]) for k in sorted(r[0])] == fx[f"ma{ci}_loss_vals"].tolist()
    assert r[1] == "pred_instances" and r[2] == [] and r[3] == []
    # ... and with the dead second ROI pass elided: the same trace minus that one call, proposals_roih empty
    del trace[:]
    r = cls.forward(make(True, False), with_gt, branch="supervised_target")
    ref = [str(t) for t in fx[f"ma{ci}_trace"]]
    assert trace == [t for t in ref if t != "roi(images=k,targets=0,compute_loss=0,branch=supervised_target)"] and r[1] == []
    # ---- unsup_data_weak ------------------------------------------------------------------------------------------------------
    ci = cases["unsup_data_weak|gt=0|ins_dc=0"]
    del trace[:]
    r = cls.forward(make(True, False), without, branch="unsup_data_weak", batched=True)
    assert trace == [str(t) for t in fx[f"ma{ci}_trace"]] and len(r) == int(fx[f"ma{ci}_arity"]) == 3 and r[0] == {}
    # ---- supervised: same calls; loss keys = the reference's minus loss_DC_img_s (documented) -----------------------------------
    ci = cases["supervised|gt=1|ins_dc=0"]
    del trace[:]
    r = cls.forward(make(True, False), with_gt, branch="supervised")
    assert trace == [str(t) for t in fx[f"ma{ci}_trace"]] and len(r) == 3
    assert sorted(r[0]) == [str(k) for k in fx[f"ma{ci}_loss_keys"] if str(k) != "loss_DC_img_s"]
    lg = T(fx[f"ma{ci}_dc_logits_0"])
    ref_l = torch.nn.functional.binary_cross_entropy_with_logits(lg, torch.zeros_like(lg)) * 0.001      # source label 0, x 0.001 (rcnn.py:256)
    np.testing.assert_allclose(dict(zip(fx[f"ma{ci}_loss_keys"], fx[f"ma{ci}_loss_vals"]))["loss_DC_img_s"], ref_l.item(), rtol=1e-6)
    # ---- domain_classifier ----------------------------------------------------------------------------------------------------
    for name, data, ins_dc in (("domain_classifier|gt=1|ins_dc=1", with_gt, True), ("domain_classifier|gt=1|ins_dc=0", with_gt, False)):
        ci = cases[name]
        del trace[:]
        roi._tag = None
        stub = make(True, ins_dc)
        seq = iter(["s", "t"])
        orig_lasp = roi.label_and_sample_proposals

        def lasp(props, gt, branch="", _seq=seq):
            roi._tag = next(_seq)
            return orig_lasp(props, gt, branch=branch)
        stub.roi_heads = types.SimpleNamespace(in_features=["vgg4"], label_and_sample_proposals=lasp)

        def rpn2(images, features, gt=None, compute_loss=True, as_instances=True):
            return rpn(images, features, gt, compute_loss)
        stub.proposal_generator = rpn2
        r = cls.forward(stub, data, branch="domain_classifier")
        ref = [str(t) for t in fx[f"ma{ci}_trace"]]
        # the reference computes both instance-level losses after both ROI passes; the product computes each right after its
        # pass: the same calls with the same flags, compared as a sorted list, the backbone passes first in both
        assert sorted(trace) == sorted(t.replace("backbone(xs)", "backbone(xs)").replace("backbone(xt)", "backbone(xt)") for t in ref)
        assert trace[:2] == ref[:2] == ["backbone(xs)", "backbone(xt)"]
        assert len(r) == int(fx[f"ma{ci}_arity"]) == 3 and sorted(r[0]) == [str(k) for k in fx[f"ma{ci}_loss_keys"]]
        assert r[1] == [] and r[2] == []
        # label constants: source 0, target 1 (the reference's BCE values on its recorded logits say the same)
        assert float(r[0]["loss_DC_img_s"]) == 0.0 and float(r[0]["loss_DC_img_t"]) == 1.0
        l0, l1 = T(fx[f"ma{ci}_dc_logits_0"]), T(fx[f"ma{ci}_dc_logits_1"])
        bce = torch.nn.functional.binary_cross_entropy_with_logits
        vals = dict(zip([str(k) for k in fx[f"ma{ci}_loss_keys"]], fx[f"ma{ci}_loss_vals"]))
        np.testing.assert_allclose(vals["loss_DC_img_s"], bce(l0, torch.zeros_like(l0)).item(), rtol=1e-6)
        np.testing.assert_allclose(vals["loss_DC_img_t"], bce(l1, torch.ones_like(l1)).item(), rtol=1e-6)
    assert str(fx["ma_eval_returns"]) == "'inference-result'"      # eval mode and not val_mode: inference()


# ---- a10 ---------------------------------------------------------------------------------------------------------------
def test_base_trainer_run_step_and_metrics_equal_the_reference(fx, sfod):
    """``BaseTrainer.run_step`` (base.py:93-123) run on a recorder model: keys with prefix ``loss`` that do not end in ``val``
    are summed into the loss (gradient 1 on their leaves, none on the others), the whole record + ``data_time`` reaches
    ``_write_metrics``, one zero_grad / step.  The product's ``BaseTrainer.run_step``, run the same way on a stub, does the same.
    ``_write_metrics`` with two ranks (base.py:186-220): data_time = max, every other key = mean, total_loss = sum of the
    averaged ``loss*`` keys -- the numbers the 2-rank gloo test (tests/test_distributed_cpu.py) reproduces with the product's
    EventStorage."""
    import types
    tr = sfod.engine.trainer
    keys = [str(k) for k in fx["bt_keys"]]
    leaves = {k: torch.tensor(float(i + 1), requires_grad=True) for i, k in enumerate(keys)}
    written = {}
    opt = types.SimpleNamespace(n_zero=0, n_step=0, flat=None, grad_scale=1.0)
    opt.zero_grad = lambda: setattr(opt, "n_zero", opt.n_zero + 1)
    opt.step = lambda: setattr(opt, "n_step", opt.n_step + 1)
    model = lambda data: dict(leaves)
    model.training = True
    stub = types.SimpleNamespace(model=model, optimizer=opt, _data_loader_iter=iter(["batch"]),
                                 _write_metrics=lambda d, total=None: written.update(d), _reduce_gradients=lambda: None)
    tr.BaseTrainer.run_step(stub)
    got = [float(v.grad) if v.grad is not None else float("nan") for v in leaves.values()]
    assert np.array_equal(np.array(got), fx["bt_grads"], equal_nan=True)
    assert sorted(written) == [str(k) for k in fx["bt_written_keys"]]
    assert [opt.n_zero, opt.n_step] == fx["bt_opt_calls"].tolist()
    # the two-rank logging semantics recorded from the reference
    mk = [str(k) for k in fx["bt_metric_keys"]]
    r0, r1 = dict(zip(mk, fx["bt_rank0"])), dict(zip(mk, fx["bt_rank1"]))
    logged = dict(zip([str(k) for k in fx["bt_logged_keys"]], fx["bt_logged_vals"]))
    assert logged["data_time"] == max(r0["data_time"], r1["data_time"])
    for k in mk:
        if k != "data_time":
            assert logged[k] == 0.5 * (r0[k] + r1[k])
    assert logged["total_loss"] == sum(logged[k] for k in mk if k[:4] == "loss")       # '*_val' keys included: prefix test only
    # one rank, the product's storage: the same keys, total_loss by the same prefix rule
    st = tr.EventStorage(0)
    holder = types.SimpleNamespace(storage=st)
    tr.BaseTrainer._write_metrics(holder, {k: torch.tensor(float(v)) if k != "data_time" else float(v) for k, v in r0.items()})
    rec = st.flush()
    assert sorted(k for k in rec if k != "iteration") == sorted(logged)
    assert rec["total_loss"] == sum(r0[k] for k in mk if k[:4] == "loss") and rec["data_time"] == r0["data_time"]


def test_loader_builder_per_rank_batch_and_errors_equal_the_reference(fx, sfod):
    """``build_semisup_batch_data_loader_two_crop_source_free`` (daod/data/build.py:312-367) run with ``get_world_size``
    patched: per-rank batch = IMS_PER_BATCH_TARGET // world, the divisibility assertion's text, and the
    ``ASPECT_RATIO_GROUPING = False`` error.  The product's ``TwoCropLoader`` does the same on the same (world, total) rows."""
    cfg = sfod.config.get_cfg()
    sfod.config.add_config(cfg)
    ds = [dict(image=torch.zeros(3, 8, 8, dtype=torch.uint8), boxes=torch.zeros(0, 4), classes=torch.zeros(0, dtype=torch.long))]
    ds = type("DS", (list,), {"size": (8, 8)})(ds * 8)
    for world, total, per_rank, group in fx["lb_world_total_batch"].tolist():
        cfg.SOLVER.IMS_PER_BATCH_TARGET = int(total)
        for rank in range(world):
            ld = sfod.data.synthetic.TwoCropLoader(cfg, "cpu", rank=rank, world=world, dataset=ds)
            assert ld.batch == per_rank == group
    cfg.SOLVER.IMS_PER_BATCH_TARGET = 4
    with pytest.raises(AssertionError) as e:
        sfod.data.synthetic.TwoCropLoader(cfg, "cpu", rank=0, world=3, dataset=ds)
    assert str(e.value) == str(fx["lb_assert_msg"])
    cfg.DATALOADER.ASPECT_RATIO_GROUPING = False
    with pytest.raises(NotImplementedError) as e:
        sfod.data.synthetic.TwoCropLoader(cfg, "cpu", rank=0, world=1, dataset=ds)
    assert str(e.value) == str(fx["lb_nogroup_error"])
