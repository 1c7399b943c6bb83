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
    feats["vgg4"].sum().backward()
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
    gen_vgg()
    gen_dann()
    gen_adaptive()
    gen_bpc()
