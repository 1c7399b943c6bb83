"""Generate tests/golden/*.npz by RUNNING THE REFERENCE where it can be imported.

Run in the build container only (``python -m oracle.gen_golden``); it reads
``/root/reference`` which does not exist on the GPU box.  Only data (inputs, expected
outputs, checksums) is written -- no reference source travels.

Pinned against the real reference here:
  * ``daod/modeling/meta_arch/vgg.py``  (loaded by file path behind ``oracle/ref_stub``):
    ``vgg_backbone`` forward in train mode, BN running stats after one forward, input
    gradient of ``vgg4.sum()``.
  * ``daod/modeling/dann/dann.py`` (pure torch, imported directly by file path):
    ``FCDiscriminator_img`` forward + gradient through ``gradient_scalar(x, -1.0)``;
    ``DAInsHead`` eval-mode forward.
  * ``daod/modeling/adaptive_thresh/adaptive_confidence.py`` (pure torch; ``Tensor.cuda`` neutralised while it
    runs): the class-wise confidence mask and ``update``.
  * ``daod/loss/bpc_loss.py`` (loaded by file path behind ``oracle/ref_stub``, ``Tensor.cuda`` neutralised): the
    BPC calibration scalar for recorded ground truth / detections.
Weights are not stored for the big modules: they are reproduced from the recorded seed by
constructing the same torch.nn containers in the same order (checked via checksums).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def checksum(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * torch.arange(1, t.numel() + 1,
                    dtype=torch.float64) % 7).sum().item()], dtype=np.float64)


def gen_vgg():
    sys.path.insert(0, os.path.join(HERE, "ref_stub"))
    vgg = _load_by_path("ref_vgg", os.path.join(REF, "daod/modeling/meta_arch/vgg.py"))
    cfg = types.SimpleNamespace(VGG=types.SimpleNamespace(BN=True))
    seed = 0
    torch.manual_seed(seed)
    model = vgg.build_vgg_backbone(cfg, None)
    model.train()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 3, 64, 128, generator=g)
    x.requires_grad_(True)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    feats = model(x)
    feats["vgg4"].sum().backward(retain_graph=True)
    input_grad = x.grad.clone()
    # parameter gradients of the reference's backward (conv weights, BatchNorm weight / bias; the conv biases in front of a
    # train-mode BatchNorm have an analytically zero gradient) for a non-degenerate loss: sum over stages of <vgg_i, r_i>
    model.zero_grad()
    x.grad = None
    gr = torch.Generator().manual_seed(4321)
    rs = {i: torch.randn(feats[f"vgg{i}"].shape, generator=gr) for i in (2, 3, 4)}
    sum((feats[f"vgg{i}"] * rs[i]).sum() for i in rs).backward()
    pgrads = {k: p.grad.clone() for k, p in model.named_parameters()}
    input_grad2 = x.grad.clone()
    x.grad = input_grad
    out = {
        "seed": np.int64(seed),
        "input": x.detach().numpy(),
        "input_grad": x.grad.numpy(),
        "torch_version": np.array(torch.__version__),
        "keys": np.array(list(sd0.keys())),
        "out_feature_channels": np.array([model._out_feature_channels[f"vgg{i}"] for i in range(5)]),
        "out_feature_strides": np.array([model._out_feature_strides[f"vgg{i}"] for i in range(5)]),
    }
    for i in range(5):
        f = feats[f"vgg{i}"].detach()
        out[f"vgg{i}_shape"] = np.array(f.shape)
        out[f"vgg{i}_checksum"] = checksum(f)
        out[f"vgg{i}"] = f.numpy() if i >= 2 else f[:, ::8, ::4, ::4].contiguous().numpy()
    sd1 = model.state_dict()
    for k, v in sd0.items():
        if k.endswith("weight") and v.dim() == 4:
            out["wsum/" + k] = checksum(v)
        if "running" in k or "num_batches" in k:
            out["after/" + k] = sd1[k].numpy()
    # the backward's fixture: per parameter the norm, a checksum and <= 4096 strided samples of the gradient
    out["bwd_seed"] = np.int64(4321)
    out["bwd_stages"] = np.array([2, 3, 4])
    out["bwd_input_grad"] = input_grad2.numpy()
    for k, gk in pgrads.items():
        flat = gk.flatten()
        stride = max(1, flat.numel() // 4096)
        out["gnorm/" + k] = np.float64(flat.double().norm().item())
        out["gsum/" + k] = checksum(gk)
        out["gstride/" + k] = np.int64(stride)
        out["g/" + k] = flat[::stride].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "vgg_ref.npz"), **out)
    print("vgg_ref.npz:", {k: feats[k].shape for k in feats})


def gen_dann():
    dann = _load_by_path("ref_dann", os.path.join(REF, "daod/modeling/dann/dann.py"))
    torch.manual_seed(7)
    dc = dann.FCDiscriminator_img(64, ndf1=32, ndf2=16)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(2, 64, 9, 11, generator=g, requires_grad=True)
    y = dc(dann.gradient_scalar(x, -1.0))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(y, torch.zeros_like(y))
    loss.backward()
    out = {"input": x.detach().numpy(), "output": y.detach().numpy(), "loss": loss.detach().numpy(),
           "input_grad": x.grad.numpy(), "torch_version": np.array(torch.__version__)}
    for k, v in dc.state_dict().items():
        out["w/" + k] = v.numpy()
    for k, p in dc.named_parameters():
        out["g/" + k] = p.grad.numpy()
    # instance head, eval mode (dropout off), weights reproduced from the seed
    torch.manual_seed(11)
    ins = dann.DAInsHead(64, ["vgg4"])
    ins.eval()
    xi = torch.randn(5, 64, generator=g)
    yi = ins(xi, levels=torch.zeros(5, dtype=torch.int64))
    out["ins_seed"] = np.int64(11)
    out["ins_input"] = xi.numpy()
    out["ins_output"] = yi.detach().numpy()
    for k, v in ins.state_dict().items():
        out["ins_wsum/" + k] = checksum(v)
    np.savez_compressed(os.path.join(OUT, "dann_ref.npz"), **out)
    print("dann_ref.npz: loss", float(loss))


def gen_adaptive():
    """``AdaptiveConfidenceBasedSelfTrainingLoss`` (adaptive_thresh/adaptive_confidence.py:6-34) is pure torch
    but its constructor calls ``.cuda()`` (:13); with ``Tensor.cuda`` neutralised for the duration of this
    function the class runs on the CPU unchanged.  Recorded: the mask for several per-class accuracy vectors
    (the trainer assigns ``classwise_acc`` directly, source_free_adaptive_teacher.py:306-309) and ``update``."""
    mod_path = os.path.join(REF, "daod/modeling/adaptive_thresh/adaptive_confidence.py")
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        ada = _load_by_path("ref_adaptive", mod_path)
        K = 8
        crit = ada.AdaptiveConfidenceBasedSelfTrainingLoss(threshold=0.8, num_classes=K)
        g = torch.Generator().manual_seed(5)
        conf = torch.rand(400, generator=g) * 0.95 + 0.05
        labels = torch.randint(0, K, (400,), generator=g)
        out = {"threshold": np.float64(0.8), "confidence": conf.numpy(), "labels": labels.numpy(),
               "mask_init": crit(conf, labels).numpy(), "torch_version": np.array(torch.__version__)}
        accs = []
        for i in range(6):
            counter = torch.randint(0, 40, (K,), generator=g).float()
            if i == 0:
                counter.zero_()
            counter[0] = 0
            counter[2] = 0
            acc = counter / max(counter.max(), 1)          # trainer :306-309
            acc[0] = 1
            acc[2] = 1
            crit.classwise_acc = acc
            # confidences exactly on a class threshold exercise the '>='
            c2 = conf.clone()
            c2[:K] = 0.8 * (acc / (2. - acc))
            l2 = labels.clone()
            l2[:K] = torch.arange(K)
            accs.append(acc.numpy())
            out[f"conf_{i}"] = c2.numpy()
            out[f"labels_{i}"] = l2.numpy()
            out[f"mask_{i}"] = crit(c2, l2).numpy()
        out["accs"] = np.stack(accs)
        sel = torch.randint(0, K, (57,), generator=g)
        crit.update(sel)
        out["update_labels"] = sel.numpy()
        out["update_acc"] = crit.classwise_acc.numpy()
    finally:
        torch.Tensor.cuda = orig
    np.savez_compressed(os.path.join(OUT, "adaptive_ref.npz"), **out)
    print("adaptive_ref.npz: kept", [int(out[f"mask_{i}"].sum()) for i in range(6)])


def gen_bpc():
    """``bpc_loss`` (daod/loss/bpc_loss.py:10-262) run on the CPU: loaded by file path behind the import stub
    (it only needs a box container with ``.tensor``), ``Tensor.cuda`` neutralised.  Recorded: per-image ground
    truth (boxes, classes), detections (boxes, scores, classes) and the scalar, for cases with classes without
    ground truth, exact IoU ties between two ground-truth boxes (the detection's score counts twice), an image
    without ground truth, an image whose detections are all below 0.5."""
    if os.path.join(HERE, "ref_stub") not in sys.path:
        sys.path.insert(0, os.path.join(HERE, "ref_stub"))
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        bpc = _load_by_path("ref_bpc", os.path.join(REF, "daod/loss/bpc_loss.py"))
        from detectron2.structures import Boxes
        K = 8
        g = torch.Generator().manual_seed(23)
        out = {"num_classes": np.int64(K), "torch_version": np.array(torch.__version__)}

        def rand_boxes(n, span=300.0, size=120.0):
            xy = torch.rand(n, 2, generator=g) * span
            wh = torch.rand(n, 2, generator=g) * size + 4
            return torch.cat([xy, xy + wh], 1)

        cases = []
        for ci in range(5):
            gts, dts = [], []
            B = 3
            for b in range(B):
                ng = 0 if (ci == 2 and b == 1) else int(torch.randint(1, 9, (1,), generator=g))
                gb = rand_boxes(ng)
                gc = torch.randint(0, K - 2, (ng,), generator=g)       # classes 6, 7 never have ground truth
                if ci == 1 and ng >= 2:                                 # two identical boxes of one class: IoU tie
                    gb[1] = gb[0]
                    gc[1] = gc[0]
                nd = 200
                db = rand_boxes(nd)
                jit = torch.randint(0, max(ng, 1), (nd,), generator=g)
                near = torch.rand(nd, generator=g) < 0.5
                if ng > 0:
                    db[near] = gb[jit[near]] + torch.randn(int(near.sum()), 4, generator=g) * 4
                dc = torch.randint(0, K, (nd,), generator=g)
                if ng > 0:
                    dc[near] = gc[jit[near]]
                ds = torch.rand(nd, generator=g)
                if ci == 3:
                    ds = ds * 0.45
                gts.append((gb, gc))
                dts.append((db, ds, dc))
            ins = [types.SimpleNamespace(gt_boxes=Boxes(gb), gt_classes=gc) for gb, gc in gts]
            outs = [types.SimpleNamespace(pred_boxes=Boxes(db), scores=ds, pred_classes=dc) for db, ds, dc in dts]
            val = bpc.bpc_loss(K, ins, outs)
            for b in range(B):
                out[f"c{ci}_gt_boxes_{b}"], out[f"c{ci}_gt_classes_{b}"] = gts[b][0].numpy(), gts[b][1].numpy()
                out[f"c{ci}_dt_boxes_{b}"], out[f"c{ci}_dt_scores_{b}"] = dts[b][0].numpy(), dts[b][1].numpy()
                out[f"c{ci}_dt_classes_{b}"] = dts[b][2].numpy()
            out[f"c{ci}_loss"] = np.float64(float(val))
            cases.append(float(val))
        out["num_cases"] = np.int64(5)
    finally:
        torch.Tensor.cuda = orig
    np.savez_compressed(os.path.join(OUT, "bpc_ref.npz"), **out)
    print("bpc_ref.npz:", cases)


def _ref_modules():
    """the reference's glue files, loaded by file path behind ``oracle/ref_stub/hook.py``"""
    sys.path.insert(0, os.path.join(HERE, "ref_stub"))
    import hook
    hook.install()
    mods = types.SimpleNamespace(hook=hook)
    mods.trainer = _load_by_path("ref_sfat", os.path.join(REF, "daod/engine/trainers/source_free_adaptive_teacher.py"))
    mods.frcnn = _load_by_path("ref_frcnn", os.path.join(REF, "daod/modeling/roi_heads/source_free_fast_rcnn.py"))
    mods.common = _load_by_path("ref_common", os.path.join(REF, "daod/data/common.py"))
    mods.rpn = _load_by_path("ref_rpn", os.path.join(REF, "daod/modeling/proposal_generator/rpn.py"))
    mods.base = _load_by_path("ref_base", os.path.join(REF, "daod/engine/trainers/base.py"))
    mods.config = _load_by_path("ref_config", os.path.join(REF, "daod/config.py"))
    mods.build = _load_by_path("ref_build", os.path.join(REF, "daod/data/build.py"))
    import importlib
    importlib.import_module("daod.modeling.roi_heads")       # (made up by the hook) parent of the file's relative import
    mods.roi_heads = _load_by_path("daod.modeling.roi_heads.ref_roi_heads", os.path.join(
        REF, "daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py"))
    return mods


def gen_roi_heads_glue(m, out, g):
    """a5: ``SourceFreeAdaptiveTeacherStandardROIHeads.{label_and_sample_proposals, forward, _forward_box}``
    (daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py:68-215), the reference's own code objects on a stub
    ``self``.  The Detectron2 primitives they call (``add_ground_truth_to_proposals``, ``pairwise_iou``, the Matcher,
    ``subsample_labels`` behind ``_sample_proposals``) are absent: the oracle's restatements stand in for them (with the
    sampling keys as an input, oracle/box_ops.py A.9) -- so what these vectors pin is the reference-OWNED part: which
    proposals receive which ground-truth fields, the zero boxes of an image without ground truth, the logged foreground /
    background counts and their key names, the return arity of every flag combination, the order of the calls in the box
    branch and the overwrite of the sampled proposals' boxes (:136-143)."""
    import math
    from detectron2.structures import Boxes, Instances
    from oracle import box_ops as OB
    mod = m.roi_heads
    cls = mod.SourceFreeAdaptiveTeacherStandardROIHeads
    K, BATCH, FRAC = 8, 64, 0.25

    def add_gt(gt_boxes, proposals):
        logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
        res = []
        for gb, p in zip(gt_boxes, proposals):
            q = Instances(p.image_size)
            q.proposal_boxes = Boxes(torch.cat([p.proposal_boxes.tensor, gb.tensor]))
            q.objectness_logits = torch.cat([p.objectness_logits, logit * torch.ones(len(gb))])
            res.append(q)
        return res
    mod.add_ground_truth_to_proposals = add_gt
    mod.pairwise_iou = lambda a, b: OB.pairwise_iou(a.tensor, b.tensor)
    scalars = {}
    mod.get_event_storage = lambda: types.SimpleNamespace(put_scalar=lambda k, v: scalars.__setitem__(k, float(v)))

    stub = object.__new__(cls)
    stub.proposal_append_gt = True
    stub.num_classes = K
    stub.proposal_matcher = lambda M: OB.matcher(M, [0.5], [0, 1], False)
    keys_iter = []

    def sample(matched_idxs, matched_labels, gt_classes):
        if gt_classes.numel() > 0:
            gc = gt_classes[matched_idxs]
            gc[matched_labels == 0] = K
            gc[matched_labels == -1] = -1
        else:
            gc = torch.zeros_like(matched_idxs) + K
        keys = keys_iter.pop(0)
        fg, bg = OB.subsample_labels(gc, BATCH, FRAC, K, keys[: len(gc)])
        idx = torch.cat([fg, bg], dim=0)
        return idx, gc[idx]
    stub._sample_proposals = sample
    size = (300, 400)
    props, targets, keys = [], [], []
    for i, ng in enumerate((5, 0, 3)):
        gb = torch.rand(ng, 2, generator=g) * torch.tensor([250.0, 180.0])
        gb = torch.cat([gb, gb + torch.rand(ng, 2, generator=g) * 120 + 20], 1)
        n = 150
        pb = torch.rand(n, 2, generator=g) * torch.tensor([300.0, 220.0])
        pb = torch.cat([pb, pb + torch.rand(n, 2, generator=g) * 150 + 8], 1)
        if ng:
            near = torch.randint(0, ng, (60,), generator=g)
            pb[:60] = gb[near] + torch.randn(60, 4, generator=g) * 6
        p = Instances(size)
        p.proposal_boxes = Boxes(pb)
        p.objectness_logits = torch.randn(n, generator=g)
        t = Instances(size)
        t.gt_boxes = Boxes(gb)
        t.gt_classes = torch.randint(0, K, (ng,), generator=g)
        t.scores = torch.rand(ng, generator=g)              # a non-"gt_" field: must NOT be copied to the proposals
        props.append(p)
        targets.append(t)
        keys.append(torch.randint(0, 2 ** 31 - 1, (n + ng,), generator=g))
    keys_iter.extend(k.clone() for k in keys)
    res = cls.label_and_sample_proposals(stub, props, targets, branch="supervised_target")
    out["roi_K"], out["roi_batch"], out["roi_frac"] = np.int64(K), np.int64(BATCH), np.float64(FRAC)
    out["roi_size"] = np.array(size)
    for i in range(3):
        out[f"roi_in_boxes_{i}"], out[f"roi_in_logits_{i}"] = props[i].proposal_boxes.tensor.numpy(), props[i].objectness_logits.numpy()
        out[f"roi_gt_boxes_{i}"], out[f"roi_gt_classes_{i}"] = targets[i].gt_boxes.tensor.numpy(), targets[i].gt_classes.numpy()
        out[f"roi_keys_{i}"] = keys[i].numpy()
        r = res[i]
        out[f"roi_out_fields_{i}"] = np.array(sorted(r.get_fields().keys()))
        out[f"roi_out_boxes_{i}"], out[f"roi_out_logits_{i}"] = r.proposal_boxes.tensor.numpy(), r.objectness_logits.numpy()
        out[f"roi_out_gt_classes_{i}"], out[f"roi_out_gt_boxes_{i}"] = r.gt_classes.numpy(), r.gt_boxes.tensor.numpy()
    out["roi_scalar_keys"] = np.array(sorted(scalars))
    out["roi_scalar_vals"] = np.array([scalars[k] for k in sorted(scalars)])

    # ---- forward / _forward_box: which path runs, what comes back ------------------------------------------------------
    trace = []
    stub.box_in_features = ["vgg4"]
    stub.box_pooler = lambda feats, boxes: trace.append(("box_pooler", len(boxes))) or "pooled"
    stub.box_head = lambda x: trace.append(("box_head", x)) or "box_features"

    class Predictor:
        def __call__(self, x):
            trace.append(("box_predictor", x))
            return "predictions"

        def losses(self, predictions, proposals):
            trace.append(("losses", [float(p.proposal_boxes.tensor.sum()) for p in proposals]))
            return {"loss_cls": 1.0, "loss_box_reg": 2.0}

        def predict_boxes_for_gt_classes(self, predictions, proposals):
            trace.append(("predict_boxes_for_gt_classes",))
            return [p.proposal_boxes.tensor + 1.0 for p in proposals]

        def convert_bbox_scores(self, predictions, proposals):
            trace.append(("convert_bbox_scores", [float(p.proposal_boxes.tensor.sum()) for p in proposals]))
            return "instance_proposals", "rows"

        def inference(self, predictions, proposals):
            trace.append(("inference",))
            return "pred_instances", "rows"
    stub.box_predictor = Predictor()
    feats = {"vgg4": "feature"}
    combos = [(True, True, False), (True, False, False), (False, True, False), (False, False, True), (True, False, True)]
    arity, paths, appended = [], [], []
    for training, compute_loss, compute_val_loss in combos:
        stub.training = training
        del trace[:]
        keys_iter.extend(k.clone() for k in keys)
        seen = []
        orig = stub._sample_proposals

        def spy(mi, ml, gc, _o=orig):
            seen.append(bool(stub.proposal_append_gt))
            return _o(mi, ml, gc)
        stub._sample_proposals = spy
        fresh = []
        for p in props:
            q = Instances(size)
            q.proposal_boxes = Boxes(p.proposal_boxes.tensor.clone())
            q.objectness_logits = p.objectness_logits.clone()
            fresh.append(q)
        r = cls.forward(stub, None, feats, fresh, targets, compute_loss=compute_loss, branch="b",
                        compute_val_loss=compute_val_loss)
        stub._sample_proposals = orig
        del keys_iter[:]
        arity.append(len(r))
        paths.append("|".join(t[0] for t in trace))
        appended.append(int(seen[0]) if seen else -1)
        if training and compute_loss:
            # the sampled proposals' boxes were overwritten AFTER the losses and BEFORE convert_bbox_scores (:136-143)
            l = [t for t in trace if t[0] == "losses"][0][1]
            c = [t for t in trace if t[0] == "convert_bbox_scores"][0][1]
            out["roi_fwd_box_sum_at_losses"], out["roi_fwd_box_sum_at_convert"] = np.array(l), np.array(c)
            out["roi_fwd_box_sum_returned"] = np.array([float(p.proposal_boxes.tensor.sum()) for p in r[0]])
            out["roi_fwd_rows"] = np.array([len(p) for p in r[0]])
    out["roi_fwd_flags"] = np.array(combos)
    out["roi_fwd_arity"], out["roi_fwd_paths"], out["roi_fwd_append_gt_seen"] = np.array(arity), np.array(paths), np.array(appended)
    assert stub.proposal_append_gt is True


def gen_run_step(m, out, g):
    """a8: ``SourceFreeAdaptiveTeacherTrainer.run_step`` (source_free_adaptive_teacher.py:335-581), the reference's own code
    object on a stub ``self`` whose teacher / student are recorders: the order of the calls and their ``branch`` arguments,
    q = deep copy of k without WEAK_STRONG_AUGMENT, labels removed before the teacher sees the data, the thresholded
    detections attached as ``instances`` to BOTH lists, the ``*_unlabeled`` keys of the domain pass, the scalars put into the
    storage, which keys reach ``_write_metrics`` -- and the weight of every loss key, read off as the gradient that
    ``losses.backward()`` leaves on each (leaf) loss tensor, for every combination of the domain-classifier switches."""
    from detectron2.structures import Boxes, Instances
    T = m.trainer.SourceFreeAdaptiveTeacherTrainer
    K = 8

    def cfgns(dc_enabled, dc_img, dc_ins, unsup_w, dis_w, wsa):
        ns = types.SimpleNamespace
        return ns(STYLE=ns(ENABLED=False), WEAK_STRONG_AUGMENT=wsa, ADAPTIVE_THRESHOLD=ns(ENABLED=False, WARM_UP=100, RESERVE=500),
                  SEMISUPNET=ns(BBOX_THRESHOLD=0.8, UNSUP_LOSS_WEIGHT=unsup_w, DIS_LOSS_WEIGHT=dis_w),
                  DOMAIN_CLASSIFIER=ns(ENABLED=dc_enabled, IMAGE=dc_img, INSTANCE=dc_ins), MODEL=ns(ROI_HEADS=ns(NUM_CLASSES=K)))

    def detections(n):
        p = Instances((600, 1200))
        p.pred_boxes = Boxes(torch.rand(n, 4, generator=g) * 500)
        p.scores = torch.rand(n, generator=g).sort(descending=True).values
        p.pred_classes = torch.randint(0, K, (n,), generator=g)
        return p

    def rpn_props(n):
        p = Instances((600, 1200))
        p.proposal_boxes = Boxes(torch.rand(n, 4, generator=g) * 500)
        p.objectness_logits = torch.randn(n, generator=g) * 2
        return p
    dets = [detections(40), detections(25)]
    props = [rpn_props(60), rpn_props(60)]
    loss_keys = ["loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "loss_bpc"]
    dc_keys = ["loss_DC_img_s", "loss_DC_img_t", "loss_DC_ins_s", "loss_DC_ins_t"]
    combos = [(True, False, False, 1.0, 0.1, False), (True, True, False, 1.0, 0.1, False), (True, False, True, 4.0, 0.25, False),
              (True, True, True, 1.0, 0.1, True), (False, False, False, 1.0, 0.1, True)]
    out["rs_combos"] = np.array([[float(v) for v in c] for c in combos])
    out["rs_loss_keys"], out["rs_dc_keys"] = np.array(loss_keys), np.array(dc_keys)
    for ci, (dc_on, dc_img, dc_ins, unsup_w, dis_w, wsa) in enumerate(combos):
        calls, scalars, written = [], {}, {}
        leaves = {}

        def teacher(data, branch=""):
            calls.append(("teacher", branch, [("instances" in d) for d in data], [id(d) for d in data]))
            return {}, props, dets

        class Student:
            training = True

            def __call__(self, data, branch=""):
                if branch == "supervised_target":
                    calls.append(("student", branch, [len(d["instances"]) for d in data],
                                  [sorted(d["instances"].get_fields().keys()) for d in data], [d["tag"] for d in data]))
                    rec = {}
                    for k in loss_keys:
                        leaves[k + "_pseudo"] = torch.tensor(float(len(leaves) + 1), requires_grad=True)
                        rec[k] = leaves[k + "_pseudo"]
                    return rec, "predictions", None, None
                calls.append(("student", branch, sorted(k for k in data[0].keys()), [d.get("tag_unlabeled") for d in data]))
                rec = {}
                for k in dc_keys:
                    leaves[k] = torch.tensor(float(len(leaves) + 1), requires_grad=True)
                    rec[k] = leaves[k]
                return rec, None, None
        q = [{"image": torch.zeros(3, 4, 4), "instances": "gt_q0", "tag": "q0"}, {"image": torch.zeros(3, 4, 4), "instances": "gt_q1", "tag": "q1"}]
        k_ = [{"image": torch.ones(3, 4, 4), "instances": "gt_k0", "tag": "k0"}, {"image": torch.ones(3, 4, 4), "instances": "gt_k1", "tag": "k1"}]
        opt = types.SimpleNamespace(n_zero=0, n_step=0)
        opt.zero_grad = lambda: setattr(opt, "n_zero", opt.n_zero + 1)
        opt.step = lambda: setattr(opt, "n_step", opt.n_step + 1)
        stub = object.__new__(T)
        stub.__dict__.update(dict(
            iter=7, cfg=cfgns(dc_on, dc_img, dc_ins, unsup_w, dis_w, wsa), model=Student(), model_teacher=teacher, optimizer=opt,
            _trainer=types.SimpleNamespace(iter=None, _data_loader_iter=iter([(q, k_)])),
            storage=types.SimpleNamespace(put_scalar=lambda n, v: scalars.__setitem__(n, float(v))),
            _write_metrics=lambda d: written.update({kk: (float(v) if not isinstance(v, float) else v) for kk, v in d.items()})))
        T.run_step(stub)
        pre = f"rs{ci}_"
        out[pre + "call_order"] = np.array([c[0] + ":" + c[1] for c in calls])
        out[pre + "teacher_saw_instances"] = np.array(calls[0][2])
        st = [c for c in calls if c[0] == "student" and c[1] == "supervised_target"][0]
        out[pre + "student_label_counts"] = np.array(st[2])
        out[pre + "student_label_fields"] = np.array(st[3][0])
        out[pre + "student_tags"] = np.array(st[4])          # k-tags: q was replaced by a deep copy of k
        dcall = [c for c in calls if c[1] == "domain_classifier"]
        out[pre + "domain_keys"] = np.array(dcall[0][2] if dcall else [])
        out[pre + "domain_unlabeled_tags"] = np.array([str(t) for t in dcall[0][3]] if dcall else [])
        out[pre + "weight_keys"] = np.array(sorted(leaves))
        out[pre + "weights"] = np.array([float(leaves[kk].grad) if leaves[kk].grad is not None else np.nan for kk in sorted(leaves)])
        out[pre + "scalar_keys"] = np.array(sorted(scalars))
        out[pre + "scalar_vals"] = np.array([scalars[kk] for kk in sorted(scalars)])
        out[pre + "metrics_keys"] = np.array(sorted(written))
        out[pre + "metrics_vals"] = np.array([written[kk] if kk != "data_time" else -1.0 for kk in sorted(written)])
        out[pre + "opt_calls"] = np.array([opt.n_zero, opt.n_step])
        out[pre + "trainer_iter"] = np.int64(stub._trainer.iter)
    for i in range(2):
        out[f"rs_det_scores_{i}"], out[f"rs_det_boxes_{i}"] = dets[i].scores.numpy(), dets[i].pred_boxes.tensor.numpy()
        out[f"rs_det_classes_{i}"] = dets[i].pred_classes.numpy()
        out[f"rs_rpn_logits_{i}"] = props[i].objectness_logits.numpy()


def gen_meta_arch(m, out, g):
    """a3: ``SourceFreeAdaptiveTeacherGeneralizedRCNN.forward`` (source_free_adaptive_teacher_rcnn.py:106-339) on a stub
    ``self`` whose sub-modules are recorders: per branch the sequence of sub-module calls with their flags (ground truth
    passed or not, ``compute_loss``, ``branch``), the arity of the returned tuple, the loss keys -- incl. the second, loss-free
    ROI pass and the BPC call of ``supervised_target``, the two backbone passes and label constants of ``domain_classifier``
    (source 0 / target 1 through the gradient-reversal layer: the recorded losses are BCE-with-logits of the recorded logits
    against those constants) and the x 0.001 of ``supervised``."""
    import importlib
    importlib.import_module("daod.modeling.meta_arch")
    mod = _load_by_path("daod.modeling.meta_arch.ref_rcnn", os.path.join(
        REF, "daod/modeling/meta_arch/source_free_adaptive_teacher_rcnn.py"))
    dann = _load_by_path("ref_dann_for_rcnn", os.path.join(REF, "daod/modeling/dann/dann.py"))
    mod.gradient_scalar = dann.gradient_scalar
    mod.assign_boxes_to_levels = lambda boxes, *a: "levels"
    cls = mod.SourceFreeAdaptiveTeacherGeneralizedRCNN
    trace = []
    mod.bpc_loss = lambda K, gt, props: trace.append("bpc_loss(K=%d,gt=%d,props=%s)" % (K, gt is not None, props)) or torch.tensor(0.25)
    feat = torch.randn(2, 4, 3, 5, generator=g)
    logits_by_call = []

    def dc_img(x):
        y = (x * torch.linspace(-1, 1, x.numel()).view_as(x)).sum(1, keepdim=True)
        logits_by_call.append(y.detach().clone())
        return y

    def rpn(images, features, gt=None, compute_loss=True, compute_val_loss=False):
        trace.append("rpn(images=%s,gt=%d,compute_loss=%d)" % (images.tag, gt is not None, compute_loss))
        return "proposals_rpn", {"loss_rpn_cls": torch.tensor(1.0), "loss_rpn_loc": torch.tensor(2.0)}

    def roi(images, features, proposals, targets=None, compute_loss=True, branch="", compute_val_loss=False):
        trace.append("roi(images=%s,targets=%d,compute_loss=%d,branch=%s)" % (images.tag, targets is not None, compute_loss, branch))
        if compute_loss:
            return [types.SimpleNamespace(proposal_boxes="b")], {"loss_cls": torch.tensor(3.0), "loss_box_reg": torch.tensor(4.0)}, "box_features", "instance_proposals"
        return "pred_instances", "predictions"
    roi.box_pooler = types.SimpleNamespace(min_level=4, max_level=4, canonical_box_size=224, canonical_level=4)

    def make(ins_dc):
        stub = object.__new__(cls)
        stub.__dict__.update(dict(
            training=True, device=torch.device("cpu"), vis_period=0, dis_type="vgg4", ins_dc=ins_dc,
            cfg=types.SimpleNamespace(MODEL=types.SimpleNamespace(ROI_HEADS=types.SimpleNamespace(NUM_CLASSES=8))),
            preprocess_image=lambda b: types.SimpleNamespace(tensor="x", tag="k"),
            preprocess_image_train=lambda b: (types.SimpleNamespace(tensor="xs", tag="s"), types.SimpleNamespace(tensor="xt", tag="t")),
            backbone=lambda t: trace.append("backbone(%s)" % t) or {"vgg4": feat},
            proposal_generator=rpn, roi_heads=roi, DC_img=dc_img,
            instance_dc_loss=lambda bf, lv, label: trace.append("instance_dc_loss(%s,%s,label=%d)" % (bf, lv, label)) or torch.tensor(0.5 + label)))
        return stub
    inst = types.SimpleNamespace(to=lambda dev: "gt")
    with_gt = [{"image": 0, "instances": inst, "instances_unlabeled": inst, "image_unlabeled": 0}]
    without = [{"image": 0, "image_unlabeled": 0}]
    cases = [("supervised_target", with_gt, False), ("unsup_data_weak", without, False), ("supervised", with_gt, False),
             ("domain_classifier", with_gt, True), ("domain_classifier", without, True), ("domain_classifier", with_gt, False)]
    names = []
    for ci, (branch, data, ins_dc) in enumerate(cases):
        del trace[:]
        del logits_by_call[:]
        r = cls.forward(make(ins_dc), data, branch=branch)
        pre = f"ma{ci}_"
        names.append("%s|gt=%d|ins_dc=%d" % (branch, "instances" in data[0], ins_dc))
        out[pre + "trace"] = np.array(list(trace))
        out[pre + "arity"] = np.int64(len(r))
        out[pre + "loss_keys"] = np.array(sorted(r[0].keys()))
        out[pre + "loss_vals"] = np.array([float(r[0][k]) for k in sorted(r[0].keys())])
        out[pre + "rest"] = np.array([repr(x) for x in r[1:]])
        for j, lg in enumerate(logits_by_call):
            out[pre + f"dc_logits_{j}"] = lg.numpy()
    out["ma_cases"] = np.array(names)
    # eval mode and not val_mode -> inference()
    stub = make(False)
    stub.training = False
    stub.inference = lambda b: "inference-result"
    out["ma_eval_returns"] = np.array(repr(cls.forward(stub, without)))


def gen_base_trainer(m, out, g):
    """a10: ``BaseTrainer.run_step`` / ``_write_metrics`` (daod/engine/trainers/base.py:93-123,186-220) on a stub ``self``:
    which keys of the model's record are summed into the loss (prefix ``loss``, not ending in ``val``: read off as the
    gradients on the leaves), what reaches ``_write_metrics``, and -- with ``comm.gather`` returning two ranks' dicts --
    what is logged: ``data_time`` = max over ranks, every other key = mean over ranks (float64), ``total_loss`` = sum of the
    averaged ``loss*`` keys (here the ``*_val`` keys are part of that sum: the prefix test only)."""
    B = m.base.BaseTrainer
    leaves = {k: torch.tensor(float(i + 1), requires_grad=True) for i, k in enumerate(
        ["loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "loss_cls_val", "bbox_num/gt_bboxes", "lossy_extra"])}
    written = {}
    opt = types.SimpleNamespace(n_zero=0, n_step=0)
    opt.zero_grad = lambda: setattr(opt, "n_zero", opt.n_zero + 1)
    opt.step = lambda: setattr(opt, "n_step", opt.n_step + 1)
    model = lambda data: dict(leaves)
    model.training = True
    stub = object.__new__(B)
    stub.__dict__.update(dict(iter=3, model=model, optimizer=opt, _trainer=types.SimpleNamespace(iter=None, _data_loader_iter=iter(["batch"])),
                              _write_metrics=lambda d: written.update({k: float(v) for k, v in d.items()})))
    B.run_step(stub)
    out["bt_keys"] = np.array(list(leaves))
    out["bt_grads"] = np.array([float(v.grad) if v.grad is not None else np.nan for v in leaves.values()])
    out["bt_written_keys"] = np.array(sorted(written))
    out["bt_opt_calls"] = np.array([opt.n_zero, opt.n_step])
    # _write_metrics with two ranks
    scal = {}
    storage = types.SimpleNamespace(put_scalar=lambda k, v: scal.__setitem__(k, float(v)),
                                    put_scalars=lambda **kw: scal.update({k: float(v) for k, v in kw.items()}))
    r0 = {"loss_cls": torch.tensor(1.25), "loss_box_reg": 0.5, "loss_cls_val": torch.tensor(2.0), "bbox_num/gt_bboxes": 7.0, "data_time": 0.25}
    r1 = {"loss_cls": 3.0, "loss_box_reg": 1.0, "loss_cls_val": 4.0, "bbox_num/gt_bboxes": 9.0, "data_time": 0.75}
    m.base.comm.gather = lambda d: [dict(d), dict(r1)]
    stub2 = object.__new__(B)
    stub2.__dict__.update(dict(storage=storage))
    B._write_metrics(stub2, dict(r0))
    out["bt_rank0"] = np.array([float(v) for v in r0.values()])
    out["bt_rank1"] = np.array([float(v) for v in r1.values()])
    out["bt_metric_keys"] = np.array(list(r0))
    out["bt_logged_keys"] = np.array(sorted(scal))
    out["bt_logged_vals"] = np.array([scal[k] for k in sorted(scal)])


def gen_loader_builder(m, out, g):
    """a13 / e: ``build_semisup_batch_data_loader_two_crop_source_free`` (daod/data/build.py:312-353) with ``get_world_size``
    set to 1 / 2 / 4: the per-rank batch = IMS_PER_BATCH_TARGET // world through the reference's own DataLoader + bucketing
    chain, the divisibility assertion's message, the NotImplementedError without aspect grouping."""
    mod = m.build
    mod.AspectRatioGroupedSemiSupDatasetTwoCropSourceFree = m.common.AspectRatioGroupedSemiSupDatasetTwoCropSourceFree
    mod.worker_init_reset_seed = None
    items = [({"width": 1200, "height": 600, "image_id": 2 * i}, {"width": 1200, "height": 600, "image_id": 2 * i + 1}) for i in range(16)]

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            return items[i]            # the two-crop mapper's (strong, weak) pair; collate_fn = itemgetter(0) unwraps the batch of one
    rows = []
    for world, total in ((1, 4), (2, 4), (4, 4), (2, 8)):
        mod.get_world_size = lambda w=world: w
        loader = mod.build_semisup_batch_data_loader_two_crop_source_free(DS(), list(range(16)), total, aspect_ratio_grouping=True,
                                                                          num_workers=0)
        strong, weak = next(iter(loader))
        rows.append([world, total, len(strong), len(weak)])
    out["lb_world_total_batch"] = np.array(rows)
    mod.get_world_size = lambda: 3
    try:
        mod.build_semisup_batch_data_loader_two_crop_source_free(DS(), list(range(16)), 4, aspect_ratio_grouping=True)
        out["lb_assert_msg"] = np.array("")
    except AssertionError as e:
        out["lb_assert_msg"] = np.array(str(e))
    mod.get_world_size = lambda: 1
    try:
        mod.build_semisup_batch_data_loader_two_crop_source_free(DS(), list(range(16)), 4, aspect_ratio_grouping=False)
        out["lb_nogroup_error"] = np.array("")
    except NotImplementedError as e:
        out["lb_nogroup_error"] = np.array(str(e))


def gen_glue():
    """Reference-OWNED glue of the hot path, run here (not restated) and recorded -> ``tests/golden/glue_ref.npz`` +
    ``config_ref.json``.  Every function is the reference's own code object, loaded from its file under
    /root/reference behind the import hook and called unbound on a stub ``self``:

      a6  ``SourceFreeFastRCNNOutputLayers.fast_rcnn_inference_new`` / ``..._single_image_new``
          (daod/modeling/roi_heads/source_free_fast_rcnn.py:38-147): finite mask, background column dropped, clip,
          ``scores > 0``, row index; class-specific and class-agnostic boxes
      a7  ``threshold_bbox`` (roih + rpn) and ``process_pseudo_label`` (source_free_adaptive_teacher.py:150-183,256-280)
          with scores exactly at float32(0.8) and one ulp either side
      a9  ``_update_teacher_model`` (:583-603) on a conv + BatchNorm model: parameters, float buffers and the int64
          ``num_batches_tracked`` (int64 * float -> float32 -> truncating copy), single process and under the DDP
          ``module.`` prefix, the missing-key exception
      a13 ``AspectRatioGroupedSemiSupDatasetTwoCropSourceFree.__iter__`` (daod/data/common.py:199-228): batch order
      a4  ``PseudoLabRPN.forward`` (daod/modeling/proposal_generator/rpn.py:16-58): the (N,A,H,W)->(N,HWA) and
          (N,4A,H,W)->(N,HWA,4) layout and the second multiplication by ``loss_weight``
      a11 ``reset_bn_stats`` / ``recursive_traversal`` (daod/engine/trainers/base.py:318-328)
      b   ``add_config`` (daod/config.py:8-142): every key and default it assigns
    """
    import json
    m = _ref_modules()
    from detectron2.structures import Boxes, Instances
    g = torch.Generator().manual_seed(2024)
    out = {"torch_version": np.array(torch.__version__)}

    # ---- a6 ------------------------------------------------------------------------------------------------------
    K = 8
    layers = object.__new__(m.frcnn.SourceFreeFastRCNNOutputLayers)
    sizes = [(60, 100), (75, 50)]
    boxes_l, scores_l = [], []
    for i, (h, w) in enumerate(sizes):
        R = 37 + 5 * i
        xy = torch.rand(R, K, 2, generator=g) * torch.tensor([w * 1.3, h * 1.3]) - torch.tensor([w * 0.15, h * 0.15])
        wh = torch.rand(R, K, 2, generator=g) * 40
        bx = torch.cat([xy, xy + wh], dim=2).reshape(R, 4 * K)
        logits = torch.randn(R, K + 1, generator=g) * 4
        logits[3, 2] = -200.0          # softmax underflows to exactly 0: dropped by ``scores > 0``
        logits[5, :K] = -300.0         # a row whose foreground probabilities are all exactly 0
        sc = torch.softmax(logits, dim=-1)
        bx[7, 5] = float("nan")        # non-finite rows are removed BEFORE indexing: the row index skips them
        bx[11, 0] = float("inf")
        sc[20, 1] = float("nan")
        boxes_l.append(bx)
        scores_l.append(sc)
    props = [Instances(s) for s in sizes]
    res, kept = layers.fast_rcnn_inference_new([b.clone() for b in boxes_l], [s.clone() for s in scores_l], sizes,
                                               0.05, 0.5, 100, props)
    for i in range(2):
        out[f"frcnn_boxes_in_{i}"], out[f"frcnn_scores_in_{i}"] = boxes_l[i].numpy(), scores_l[i].numpy()
        out[f"frcnn_size_{i}"] = np.array(sizes[i])
        out[f"frcnn_pred_boxes_{i}"] = res[i].pred_boxes.tensor.numpy()
        out[f"frcnn_scores_{i}"] = res[i].scores.numpy()
        out[f"frcnn_pred_classes_{i}"] = res[i].pred_classes.numpy()
        out[f"frcnn_row_{i}"] = kept[i].numpy()
    # class-agnostic regression (boxes R x 4): every class of a row shares the row's box
    bx1 = boxes_l[0][:, :4].clone()
    r1, k1 = layers.fast_rcnn_inference_single_image_new(bx1.clone(), scores_l[0].clone(), sizes[0], 0.05, 0.5, 100, props[0])
    out["frcnn_agn_boxes_in"] = bx1.numpy()
    out["frcnn_agn_pred_boxes"], out["frcnn_agn_scores"] = r1.pred_boxes.tensor.numpy(), r1.scores.numpy()
    out["frcnn_agn_pred_classes"], out["frcnn_agn_row"] = r1.pred_classes.numpy(), k1.numpy()

    # ---- a7 ------------------------------------------------------------------------------------------------------
    tr = object.__new__(m.trainer.SourceFreeAdaptiveTeacherTrainer)
    thr = 0.8
    t32 = np.float32(thr)
    insts = []
    for i, n in enumerate((100, 0, 17)):
        sc = torch.rand(n, generator=g)
        if n >= 6:
            sc[:6] = torch.tensor([t32, np.nextafter(t32, np.float32(1)), np.nextafter(t32, np.float32(0)), 1.0, 0.0, 0.9])
        sc = sc.sort(descending=True).values      # what the teacher's fast_rcnn_inference hands over: NMS order = by score
        p = Instances((600, 1200))
        p.pred_boxes = Boxes(torch.rand(n, 4, generator=g) * 500)
        p.scores = sc
        p.pred_classes = torch.randint(0, K, (n,), generator=g)
        insts.append(p)
    lst, mean_n = m.trainer.SourceFreeAdaptiveTeacherTrainer.process_pseudo_label(tr, insts, thr, "roih", "thresholding")
    out["thr"] = np.float64(thr)
    out["pl_mean_count"] = np.float64(mean_n)
    for i, (p, q) in enumerate(zip(insts, lst)):
        out[f"pl_in_boxes_{i}"], out[f"pl_in_scores_{i}"] = p.pred_boxes.tensor.numpy(), p.scores.numpy()
        out[f"pl_in_classes_{i}"] = p.pred_classes.numpy()
        out[f"pl_gt_boxes_{i}"], out[f"pl_gt_classes_{i}"] = q.gt_boxes.tensor.numpy(), q.gt_classes.numpy()
        out[f"pl_scores_{i}"] = q.scores.numpy()
        out[f"pl_fields_{i}"] = np.array(sorted(q.get_fields().keys()))
    rp = Instances((600, 1200))
    rp.proposal_boxes = Boxes(torch.rand(50, 4, generator=g) * 500)
    lg = torch.randn(50, generator=g) * 3
    lg[:3] = torch.tensor([t32, np.nextafter(t32, np.float32(1)), np.nextafter(t32, np.float32(0))])
    rp.objectness_logits = lg
    q = m.trainer.SourceFreeAdaptiveTeacherTrainer.threshold_bbox(tr, rp, thres=thr, proposal_type="rpn")
    out["rpn_in_boxes"], out["rpn_in_logits"] = rp.proposal_boxes.tensor.numpy(), lg.numpy()
    out["rpn_gt_boxes"], out["rpn_logits"] = q.gt_boxes.tensor.numpy(), q.objectness_logits.numpy()
    out["rpn_fields"] = np.array(sorted(q.get_fields().keys()))
    try:
        m.trainer.SourceFreeAdaptiveTeacherTrainer.process_pseudo_label(tr, insts, thr, "roih", "no_such_method")
        out["pl_error"] = np.array("")
    except ValueError as e:
        out["pl_error"] = np.array(str(e))

    # ---- a9 ------------------------------------------------------------------------------------------------------
    def small_model(seed):
        torch.manual_seed(seed)
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 6, 3), torch.nn.BatchNorm2d(6), torch.nn.Conv2d(6, 4, 1),
                                  torch.nn.BatchNorm2d(4), torch.nn.Linear(5, 3))
        with torch.no_grad():
            for mod in net:
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.normal_()
                    mod.running_var.uniform_(0.5, 2.0)
        return net
    student, teacher = small_model(1), small_model(2)
    counters = [(3, 7), (1000, 4), (12, 12), (0, 1), (5, 123456789)]      # (student, teacher) num_batches_tracked
    out["ema_keys"] = np.array(list(teacher.state_dict().keys()))
    for k, v in student.state_dict().items():
        out["ema_s/" + k] = v.numpy().copy()
    for k, v in teacher.state_dict().items():
        out["ema_t0/" + k] = v.numpy().copy()
    stub = types.SimpleNamespace(model=student, model_teacher=teacher)
    out["ema_counters"] = np.array(counters)
    out["ema_keep"] = np.array([0.9996, 0.9996, 0.9, 0.5, 0.9996])
    for step, ((cs, ct), keep) in enumerate(zip(counters, out["ema_keep"].tolist())):
        student[1].num_batches_tracked.fill_(cs)
        teacher[1].num_batches_tracked.fill_(ct)
        student[3].num_batches_tracked.fill_(cs + 1)
        teacher[3].num_batches_tracked.fill_(ct + 2)
        m.trainer.SourceFreeAdaptiveTeacherTrainer._update_teacher_model(stub, keep_rate=keep)
        for k, v in teacher.state_dict().items():
            out[f"ema_t{step + 1}/" + k] = v.numpy().copy()
    # DDP: the student's keys carry ``module.`` (key[7:] strips it)
    m.hook.set_world_size(2)
    teacher2 = small_model(2)
    wrapped = types.SimpleNamespace(state_dict=lambda: {"module." + k: v for k, v in student.state_dict().items()})
    stub2 = types.SimpleNamespace(model=wrapped, model_teacher=teacher2)
    teacher2[1].num_batches_tracked.fill_(7)
    student[1].num_batches_tracked.fill_(3)
    t2_before = {k: v.clone() for k, v in teacher2.state_dict().items()}
    m.trainer.SourceFreeAdaptiveTeacherTrainer._update_teacher_model(stub2, keep_rate=0.9996)
    for k, v in teacher2.state_dict().items():
        out["ema_ddp_t0/" + k] = t2_before[k].numpy()
        out["ema_ddp_t1/" + k] = v.numpy().copy()
    for k, v in student.state_dict().items():
        out["ema_ddp_s/" + k] = v.numpy().copy()
    m.hook.set_world_size(1)
    short = types.SimpleNamespace(state_dict=lambda: {k: v for k, v in student.state_dict().items() if not k.startswith("4.")})
    try:
        m.trainer.SourceFreeAdaptiveTeacherTrainer._update_teacher_model(
            types.SimpleNamespace(model=short, model_teacher=teacher), keep_rate=0.9996)
        out["ema_error"] = np.array("")
    except Exception as e:
        out["ema_error"] = np.array(str(e))

    # ---- a13 -----------------------------------------------------------------------------------------------------
    wh = [(1200, 600), (600, 1200), (800, 800), (1333, 750), (500, 900)]
    choice = torch.randint(0, len(wh), (60,), generator=g).tolist()
    stream = [({"width": wh[c][0], "height": wh[c][1], "image_id": 2 * i}, {"width": wh[c][0], "height": wh[c][1],
               "image_id": 2 * i + 1}) for i, c in enumerate(choice)]
    out["bucket_wh"] = np.array([wh[c] for c in choice])
    for bs in (1, 2, 3, 4):
        ds = m.common.AspectRatioGroupedSemiSupDatasetTwoCropSourceFree(iter(stream), bs)
        batches = list(iter(ds))
        out[f"bucket_strong_ids_b{bs}"] = np.array([[d["image_id"] for d in s] for s, w in batches]).reshape(-1, bs)
        out[f"bucket_weak_ids_b{bs}"] = np.array([[d["image_id"] for d in w] for s, w in batches]).reshape(-1, bs)

    # ---- a4 ------------------------------------------------------------------------------------------------------
    N, A, Hf, Wf = 2, 15, 4, 5
    logits = torch.randn(N, A, Hf, Wf, generator=g)
    deltas = torch.randn(N, 4 * A, Hf, Wf, generator=g)
    cap = {}
    rpn = object.__new__(m.rpn.PseudoLabRPN)
    rpn.in_features = ["vgg4"]
    rpn.training = True
    rpn.anchor_generator = lambda feats: ["anchors"]
    rpn.anchor_generator.box_dim = 4
    rpn.rpn_head = lambda feats: ([logits], [deltas])
    rpn.label_and_sample_anchors = lambda anchors, gt: ("labels", "boxes")
    rpn.loss_weight = {"loss_rpn_cls": 1.5, "loss_rpn_loc": 0.5}

    def losses(anchors, lg_, labels, dl_, boxes):
        cap["loss_args"] = (lg_, dl_)
        return {"loss_rpn_cls": torch.tensor(2.0), "loss_rpn_loc": torch.tensor(3.0), "other": torch.tensor(7.0)}

    def predict(anchors, lg_, dl_, image_sizes):
        cap["pred_args"] = (lg_, dl_, image_sizes)
        return "proposals"
    rpn.losses, rpn.predict_proposals = losses, predict
    images = types.SimpleNamespace(image_sizes=[(64, 80), (60, 80)])
    p_, l_ = m.rpn.PseudoLabRPN.forward(rpn, images, {"vgg4": torch.zeros(N, 1, Hf, Wf)}, gt_instances=["gt"])
    out["rpn_glue_logits_in"], out["rpn_glue_deltas_in"] = logits.numpy(), deltas.numpy()
    out["rpn_glue_logits_flat"] = cap["pred_args"][0][0].numpy()
    out["rpn_glue_deltas_flat"] = cap["pred_args"][1][0].numpy()
    out["rpn_glue_loss_keys"] = np.array(sorted(l_.keys()))
    out["rpn_glue_loss_vals"] = np.array([float(l_[k]) for k in sorted(l_.keys())])
    rpn.training = False            # eval and not compute_val_loss: no losses, proposals still produced
    p2, l2 = m.rpn.PseudoLabRPN.forward(rpn, images, {"vgg4": torch.zeros(N, 1, Hf, Wf)})
    _, l3 = m.rpn.PseudoLabRPN.forward(rpn, images, {"vgg4": torch.zeros(N, 1, Hf, Wf)}, compute_val_loss=True)
    rpn.training = True
    _, l4 = m.rpn.PseudoLabRPN.forward(rpn, images, {"vgg4": torch.zeros(N, 1, Hf, Wf)}, compute_loss=False)
    out["rpn_glue_branches"] = np.array([len(l_), len(l2), len(l3), len(l4)])

    # ---- a11 -----------------------------------------------------------------------------------------------------
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4),
                              torch.nn.Sequential(torch.nn.Conv2d(4, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.ReLU()))
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_()
                mod.running_var.uniform_(0.5, 2.0)
                mod.num_batches_tracked.fill_(41)
                mod.weight.normal_()
    keys_before = list(net.state_dict().keys())
    w_before = net[1].weight.detach().clone()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m.base.recursive_traversal(net)
        m.base.recursive_traversal(net)        # the reference calls it twice (base.py:331-332)
    sd = net.state_dict()
    out["adabn_keys_before"], out["adabn_keys_after"] = np.array(keys_before), np.array(list(sd.keys()))
    for k, v in sd.items():
        if "running" in k or "num_batches" in k:
            out["adabn_after/" + k] = v.numpy().copy()
    out["adabn_weight_untouched"] = np.array(bool(torch.equal(net[1].weight.detach(), w_before)))
    out["adabn_param_names"] = np.array([n for n, _ in net.named_parameters()])
    out["adabn_requires_grad"] = np.array([bool(p.requires_grad) for _, p in net.named_parameters()])
    # a train-mode forward still refreshes the (now Parameter) statistics with momentum 0.1
    net.train()
    x = torch.randn(2, 3, 12, 12, generator=g)
    with torch.no_grad():
        net(x)
    out["adabn_x"] = x.numpy()
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["adabn_fwd/" + k] = v.numpy().copy()
    for k, v in net.state_dict().items():
        if k.endswith("weight") or k.endswith("bias"):
            out["adabn_w/" + k] = v.numpy().copy()

    gen_roi_heads_glue(m, out, g)
    gen_run_step(m, out, g)
    gen_meta_arch(m, out, g)
    gen_base_trainer(m, out, g)
    gen_loader_builder(m, out, g)
    np.savez_compressed(os.path.join(OUT, "glue_ref.npz"), **out)

    # ---- b: add_config -------------------------------------------------------------------------------------------
    node = m.hook.RecordingNode()
    m.config.add_config(node)

    def plain(n):
        return {k: (plain(v) if isinstance(v, dict) else (list(v) if isinstance(v, tuple) else v)) for k, v in n.items()}
    with open(os.path.join(OUT, "config_ref.json"), "w") as f:
        json.dump({"source": "daod/config.py::add_config run on a recording node", "assigned": plain(node)}, f, indent=1,
                  sort_keys=True)
    m.hook.uninstall()
    print("glue_ref.npz: frcnn rows", [len(k) for k in kept], "pseudo mean", mean_n, "ema err:", str(out["ema_error"]),
          "buckets b2:", out["bucket_weak_ids_b2"].shape, "cfg keys:", sorted(node.keys()))


def gen_adaptive_teacher():
    """The with-source path (``TRAINER: "adaptive_teacher"``), the reference's own code objects on recorder stubs ->
    ``tests/golden/adaptive_teacher_ref.npz``:

      at*  ``AdaptiveTeacherTrainer.run_step`` (daod/engine/trainers/adaptive_teacher.py:191-336) at five iterations around
           BURN_UP_STEP = 5 with TEACHER_UPDATE_ITER = 2: which branches run on which lists (tags), when
           ``_update_teacher_model`` is called and with which keep_rate, the labels the teacher / student see, the scalars,
           the weight of every loss key (gradients ``losses.backward()`` leaves on the loss leaves), which value of the
           doubly-defined ``loss_DC_img_s`` survives, what reaches ``_write_metrics`` (unweighted);
      atm* ``AdaptiveTeacherGeneralizedRCNN.forward`` (daod/modeling/meta_arch/adaptive_teacher_rcnn.py:102-292): per branch the
           sub-module calls with their flags, tuple arity, loss keys / values (``loss_DC_img_s * 0.001`` of ``supervised``);
      at4* ``AspectRatioGroupedSemiSupDatasetTwoCrop.__iter__`` (daod/data/common.py:119-160): the four lists of every batch
           for streams of mixed aspect ratios, incl. the elements it drops while one side waits for the other."""
    import contextlib
    import io
    m = _ref_modules()
    from detectron2.structures import Boxes, Instances
    at = _load_by_path("ref_at", os.path.join(REF, "daod/engine/trainers/adaptive_teacher.py"))
    T = at.AdaptiveTeacherTrainer
    g = torch.Generator().manual_seed(77)
    out = {"torch_version": np.array(torch.__version__)}
    K = 8

    def detections(n):
        p = Instances((600, 1200))
        p.pred_boxes = Boxes(torch.rand(n, 4, generator=g) * 500)
        p.scores = torch.rand(n, generator=g).sort(descending=True).values
        p.pred_classes = torch.randint(0, K, (n,), generator=g)
        return p

    def rpn_props(n):
        p = Instances((600, 1200))
        p.proposal_boxes = Boxes(torch.rand(n, 4, generator=g) * 500)
        p.objectness_logits = torch.randn(n, generator=g) * 2
        return p
    dets, props = [detections(30)], [rpn_props(50)]
    sup_keys = ["loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "loss_DC_img_s"]
    tgt_keys = ["loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"]
    dc_keys = ["loss_DC_img_s", "loss_DC_img_t", "loss_DC_ins_s", "loss_DC_ins_t"]
    ns = types.SimpleNamespace
    cfg = ns(SEMISUPNET=ns(BURN_UP_STEP=5, TEACHER_UPDATE_ITER=2, EMA_KEEP_RATE=0.75, BBOX_THRESHOLD=0.8, UNSUP_LOSS_WEIGHT=3.0,
                           DIS_LOSS_WEIGHT=0.25))
    iters = [2, 5, 6, 7, 9]
    out["at_iters"], out["at_burn_up"], out["at_update_iter"] = np.array(iters), np.int64(5), np.int64(2)
    out["at_weights_cfg"] = np.array([3.0, 0.25, 0.75])           # UNSUP_LOSS_WEIGHT, DIS_LOSS_WEIGHT, EMA_KEEP_RATE
    for it in iters:
        calls, scalars, written, ema_calls, leaves = [], {}, {}, [], {}

        def teacher(data, branch=""):
            calls.append("teacher:%s:%s:inst=%s" % (branch, ",".join(d["tag"] for d in data), ",".join(str(int("instances" in d)) for d in data)))
            return {}, props, dets

        class Student:
            training = True

            def __call__(self, data, branch=""):
                if branch == "domain_classifier":
                    calls.append("student:%s:%s:unl=%s" % (branch, ",".join(d["tag"] for d in data),
                                                          ",".join(str(d.get("tag_unlabeled")) for d in data)))
                    keys, suffix = dc_keys, "@dc"
                else:
                    lab = ",".join(("n%d" % len(d["instances"])) if isinstance(d["instances"], Instances) else str(d["instances"])
                                   for d in data)
                    calls.append("student:%s:%s:labels=%s" % (branch, ",".join(d["tag"] for d in data), lab))
                    keys, suffix = (sup_keys, "@sup") if branch == "supervised" else (tgt_keys, "@tgt")
                rec = {}
                for k in keys:
                    leaves[k + suffix] = torch.tensor(float(len(leaves) + 1), requires_grad=True)
                    rec[k] = leaves[k + suffix]
                return rec, [], []
        lq = [{"image": 0, "instances": "gt_lq0", "tag": "lq0"}]
        lk = [{"image": 0, "instances": "gt_lk0", "tag": "lk0"}]
        uq = [{"image": 0, "instances": "gt_uq0", "tag": "uq0"}]
        uk = [{"image": 0, "instances": "gt_uk0", "tag": "uk0"}]
        opt = ns(n_zero=0, n_step=0)
        opt.zero_grad = lambda: setattr(opt, "n_zero", opt.n_zero + 1)
        opt.step = lambda: setattr(opt, "n_step", opt.n_step + 1)
        stub = object.__new__(T)
        stub.__dict__.update(dict(
            iter=it, cfg=cfg, model=Student(), model_teacher=teacher, optimizer=opt,
            _trainer=ns(iter=None, _data_loader_iter=iter([(lq, lk, uq, uk)])),
            _update_teacher_model=lambda keep_rate=0.9996: ema_calls.append(float(keep_rate)) or calls.append("ema:%g" % keep_rate),
            storage=ns(put_scalar=lambda n, v: scalars.__setitem__(n, float(v))),
            _write_metrics=lambda d: written.update({kk: (float(v) if not isinstance(v, float) else v) for kk, v in d.items()})))
        T.run_step(stub)
        pre = f"at{it}_"
        out[pre + "calls"] = np.array(calls)
        out[pre + "ema_calls"] = np.array(ema_calls)
        out[pre + "leaf_keys"] = np.array(list(leaves))
        out[pre + "leaf_vals"] = np.array([float(v) for v in leaves.values()])
        out[pre + "leaf_grads"] = np.array([float(v.grad) if v.grad is not None else np.nan for v in leaves.values()])
        out[pre + "scalar_keys"] = np.array(sorted(scalars))
        out[pre + "scalar_vals"] = np.array([scalars[kk] for kk in sorted(scalars)])
        out[pre + "metrics_keys"] = np.array(sorted(written))
        out[pre + "metrics_vals"] = np.array([written[kk] if kk != "data_time" else -1.0 for kk in sorted(written)])
        out[pre + "opt_calls"] = np.array([opt.n_zero, opt.n_step])
    out["at_det_scores"], out["at_rpn_logits"] = dets[0].scores.numpy(), props[0].objectness_logits.numpy()

    # ---- _update_teacher_model: keep_rate 0 is a copy, 0.75 the EMA (conv + BatchNorm model, int64 counter included)
    def net():
        return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    torch.manual_seed(5)
    student, teacher_m = net(), net()
    with torch.no_grad():
        student[1].num_batches_tracked.fill_(7)
        teacher_m[1].num_batches_tracked.fill_(3)
        student[1].running_mean.uniform_(-1, 1)
    for keep in (0.0, 0.75):
        t2 = net()
        t2.load_state_dict(teacher_m.state_dict())
        stub = object.__new__(T)
        stub.__dict__.update(dict(model=student, model_teacher=t2))
        T._update_teacher_model(stub, keep_rate=keep)
        for k, v in t2.state_dict().items():
            out["atema%g/%s" % (keep, k)] = v.numpy().copy()
    for k, v in student.state_dict().items():
        out["atema_student/" + k] = v.numpy().copy()
    for k, v in teacher_m.state_dict().items():
        out["atema_teacher/" + k] = v.numpy().copy()

    # ---- meta-architecture ----------------------------------------------------------------------------------------
    import importlib
    importlib.import_module("daod.modeling.meta_arch")
    mod = _load_by_path("daod.modeling.meta_arch.ref_at_rcnn", os.path.join(REF, "daod/modeling/meta_arch/adaptive_teacher_rcnn.py"))
    dann = _load_by_path("ref_dann_for_at_rcnn", os.path.join(REF, "daod/modeling/dann/dann.py"))
    mod.gradient_scalar = dann.gradient_scalar
    mod.assign_boxes_to_levels = lambda boxes, *a: "levels"
    cls = mod.AdaptiveTeacherGeneralizedRCNN
    trace = []
    feat = torch.randn(2, 4, 3, 5, generator=g)
    logits_by_call = []

    def dc_img(x):
        y = (x * torch.linspace(-1, 1, x.numel()).view_as(x)).sum(1, keepdim=True)
        logits_by_call.append(y.detach().clone())
        trace.append("DC_img")
        return y

    def rpn(images, features, gt=None, compute_loss=True, compute_val_loss=False):
        trace.append("rpn(images=%s,gt=%d,compute_loss=%d)" % (images.tag, gt is not None, compute_loss))
        return "proposals_rpn", {"loss_rpn_cls": torch.tensor(1.0), "loss_rpn_loc": torch.tensor(2.0)}

    def roi(images, features, proposals, targets=None, compute_loss=True, branch="", compute_val_loss=False):
        trace.append("roi(images=%s,targets=%d,compute_loss=%d,branch=%s)" % (images.tag, targets is not None, compute_loss, branch))
        if compute_loss:
            return [ns(proposal_boxes="b")], {"loss_cls": torch.tensor(3.0), "loss_box_reg": torch.tensor(4.0)}, "box_features"
        return "pred_instances", "predictions"
    roi.box_pooler = ns(min_level=4, max_level=4, canonical_box_size=224, canonical_level=4)

    def make(ins_dc):
        stub = object.__new__(cls)
        stub.__dict__.update(dict(
            training=True, device=torch.device("cpu"), vis_period=0, dis_type="vgg4", ins_dc=ins_dc,
            preprocess_image=lambda b: ns(tensor="x", tag="k"),
            preprocess_image_train=lambda b: (ns(tensor="xs", tag="s"), ns(tensor="xt", tag="t")),
            backbone=lambda t: trace.append("backbone(%s)" % t) or {"vgg4": feat},
            proposal_generator=rpn, roi_heads=roi, DC_img=dc_img,
            instance_dc_loss=lambda bf, lv, label: trace.append("instance_dc_loss(%s,%s,label=%d)" % (bf, lv, label)) or torch.tensor(0.5 + label)))
        return stub
    inst = ns(to=lambda dev: "gt")
    with_gt = [{"image": 0, "instances": inst, "instances_unlabeled": inst, "image_unlabeled": 0}]
    without = [{"image": 0, "image_unlabeled": 0}]
    cases = [("supervised", with_gt, False), ("supervised_target", with_gt, False), ("unsup_data_weak", without, False),
             ("domain_classifier", with_gt, False)]
    names = []
    for ci, (branch, data, ins_dc) in enumerate(cases):
        del trace[:]
        del logits_by_call[:]
        with contextlib.redirect_stdout(io.StringIO()):
            r = cls.forward(make(ins_dc), data, branch=branch)
        pre = f"atm{ci}_"
        names.append(branch)
        out[pre + "trace"] = np.array(list(trace))
        out[pre + "arity"] = np.int64(len(r))
        out[pre + "loss_keys"] = np.array(sorted(r[0].keys()))
        out[pre + "loss_vals"] = np.array([float(r[0][k]) for k in sorted(r[0].keys())])
        out[pre + "rest"] = np.array([repr(x) for x in r[1:]])
        for j, lg in enumerate(logits_by_call):
            out[pre + f"dc_logits_{j}"] = lg.numpy()
    out["atm_cases"] = np.array(names)

    # ---- four-way batches -----------------------------------------------------------------------------------------
    gen = torch.Generator().manual_seed(3)

    def stream(prefix, n, p_wide):
        for i in range(n):
            wide = bool(torch.rand(1, generator=gen).item() < p_wide)
            w, h = (1200, 600) if wide else (600, 1200)
            yield ({"width": w, "height": h, "id": f"{prefix}{i}s"}, {"width": w, "height": h, "id": f"{prefix}{i}w"})
    for ci, (bl, bu, pl, pu) in enumerate(((1, 1, 0.7, 0.4), (2, 2, 0.6, 0.5), (2, 3, 0.5, 0.8))):
        lab, unl = list(stream("L", 60, pl)), list(stream("U", 60, pu))
        ds = m.common.AspectRatioGroupedSemiSupDatasetTwoCrop((iter(lab), iter(unl)), (bl, bu))
        batches = list(ds)
        pre = f"at4_{ci}_"
        out[pre + "sizes"] = np.array([bl, bu])
        out[pre + "label_wide"] = np.array([d[0]["width"] > d[0]["height"] for d in lab])
        out[pre + "unlabel_wide"] = np.array([d[0]["width"] > d[0]["height"] for d in unl])
        out[pre + "n_batches"] = np.int64(len(batches))
        for j, part in enumerate(("ls", "lw", "us", "uw")):
            out[pre + part] = np.array([[d["id"] for d in b[j]] for b in batches])
    m.hook.uninstall()
    np.savez_compressed(os.path.join(OUT, "adaptive_teacher_ref.npz"), **out)
    print("adaptive_teacher_ref.npz:", {it: list(out[f"at{it}_calls"]) for it in iters})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "--upstream":
        # on a machine with detectron2 + torchvision: record THEIR outputs on oracle/upstream_cases.py
        # (tests/golden/upstream_ref.npz; checked by tests/test_oracle_upstream.py) -- pins the unpinned half
        try:
            from oracle import gen_upstream
        except ImportError:
            import gen_upstream
        try:
            sys.exit(gen_upstream.main())
        except ImportError as e:
            sys.exit("--upstream needs detectron2 and torchvision: %r (neither is in the build image; run this on a "
                     "machine that has them and commit tests/golden/upstream_ref.npz)" % (e,))
    if len(sys.argv) > 1 and sys.argv[1] == "adaptive":
        gen_adaptive()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bpc":
        gen_bpc()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "glue":
        gen_glue()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "adaptive_teacher":
        gen_adaptive_teacher()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vgg":
        gen_vgg()
        sys.exit(0)
    gen_vgg()
    gen_dann()
    gen_adaptive()
    gen_bpc()
    gen_glue()
