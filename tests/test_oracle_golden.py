"""Pin the oracle against vectors recorded from the REFERENCE itself (oracle/gen_golden.py)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from util_weights import checksum, reference_vgg_state

from oracle import model as om


def _vgg_fixture():
    return np.load(os.path.join(GOLDEN, "vgg_ref.npz"), allow_pickle=False)


def test_reference_vgg_weights_reproduce_from_seed():
    fx = _vgg_fixture()
    sd = reference_vgg_state(int(fx["seed"]))
    assert [k[len("backbone."):] for k in sd.keys()] == list(fx["keys"])
    for k, v in sd.items():
        key = "wsum/" + k[len("backbone."):]
        if key in fx:
            np.testing.assert_allclose(checksum(v), fx[key], rtol=1e-12)


def test_oracle_vgg_matches_reference_forward_stats_and_grad():
    fx = _vgg_fixture()
    sd = reference_vgg_state(int(fx["seed"]))
    sd = om.clone_state(sd, requires_grad=True)
    cfg = om.Cfg()
    x = torch.from_numpy(fx["input"]).clone().requires_grad_(True)
    feats = om.vgg_forward(sd, x, cfg, training=True, return_all=True)
    assert list(fx["out_feature_channels"]) == [64, 128, 256, 512, 512]
    assert list(fx["out_feature_strides"]) == [2, 4, 8, 16, 32]
    for i in range(5):
        f = feats[f"vgg{i}"].detach()
        assert list(f.shape) == list(fx[f"vgg{i}_shape"])
        ref = torch.from_numpy(fx[f"vgg{i}"])
        got = f if i >= 2 else f[:, ::8, ::4, ::4]
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(checksum(f)[:2], fx[f"vgg{i}_checksum"][:2], rtol=1e-4)
    feats["vgg4"].sum().backward()
    torch.testing.assert_close(x.grad, torch.from_numpy(fx["input_grad"]), rtol=1e-3, atol=1e-6)
    for k in fx.files:
        if k.startswith("after/"):
            name = "backbone." + k[len("after/"):]
            torch.testing.assert_close(sd[name].detach(), torch.from_numpy(fx[k]), rtol=1e-4, atol=1e-6)


def test_reference_dann_discriminator_fixture_is_a_plain_conv_stack():
    """The DC_img golden (dann.py:10-29) is reproduced by plain conv3x3 + LeakyReLU(0.2) and the
    gradient-reversal sign (dann.py:33-51): this is the oracle for the HIP conv on that module."""
    fx = np.load(os.path.join(GOLDEN, "dann_ref.npz"), allow_pickle=False)
    x = torch.from_numpy(fx["input"]).clone().requires_grad_(True)
    w = {k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w/")}
    h = x
    for name in ("conv1", "conv2", "conv3"):
        h = F.leaky_relu(F.conv2d(h, w[name + ".weight"], w[name + ".bias"], padding=1), 0.2)
    y = F.conv2d(h, w["classifier.weight"], w["classifier.bias"], padding=1)
    torch.testing.assert_close(y.detach(), torch.from_numpy(fx["output"]), rtol=1e-5, atol=1e-6)
    loss = F.binary_cross_entropy_with_logits(y, torch.zeros_like(y))
    torch.testing.assert_close(loss.detach(), torch.from_numpy(fx["loss"]), rtol=1e-6, atol=0)
    loss.backward()
    # gradient_scalar(x, -1.0) flips the sign of the gradient that reaches the features
    torch.testing.assert_close(-x.grad, torch.from_numpy(fx["input_grad"]), rtol=1e-4, atol=1e-8)


def test_adaptive_threshold_restatement_matches_reference_class():
    """adaptive_confidence.py:6-34 run on the CPU (oracle/gen_golden.py: gen_adaptive): the oracle's mask is
    bit-identical for every recorded per-class accuracy vector, including confidences exactly on a threshold
    ('>=') and classes with accuracy 0 (threshold 0: everything passes)."""
    from oracle import model as om
    fx = np.load(os.path.join(GOLDEN, "adaptive_ref.npz"), allow_pickle=False)
    at = om.AdaptiveThreshold(float(fx["threshold"]), 8, reserve=4)
    conf, labels = torch.from_numpy(fx["confidence"]), torch.from_numpy(fx["labels"])
    assert np.array_equal(at.mask(conf, labels).float().numpy(), fx["mask_init"])
    assert np.array_equal(fx["mask_init"], (fx["confidence"] >= np.float32(0.8)).astype(np.float32))
    for i in range(6):
        at.classwise_acc = torch.from_numpy(fx["accs"][i])
        m = at.mask(torch.from_numpy(fx[f"conf_{i}"]), torch.from_numpy(fx[f"labels_{i}"]))
        assert np.array_equal(m.float().numpy(), fx[f"mask_{i}"]), i
        assert fx[f"mask_{i}"][:8].all()            # the planted on-threshold confidences pass
    # bookkeeping (:282-309): counts of score > thr into the ring, classes 0 and 2 pinned to accuracy 1
    at = om.AdaptiveThreshold(0.8, 8, reserve=2)
    det = {"scores": torch.tensor([0.95, 0.9, 0.85, 0.81, 0.8, 0.3]), "classes": torch.tensor([1, 1, 3, 0, 1, 5]),
           "boxes": torch.arange(24.).view(6, 4)}
    at.update([det], 0, 0.8)
    assert at.reserve_matrix[0].tolist() == [1, 2, 0, 1, 0, 0, 0, 0]        # 0.8 itself is not > 0.8
    assert at.classwise_acc.tolist() == [1, 1, 1, 0.5, 0, 0, 0, 0]
    sel = at.select(det)
    # class 1: thr .8; class 3: .8*(.5/1.5); class 0: .8; class 5: 0
    assert sel["gt_classes"].tolist() == [1, 1, 3, 0, 1, 5] and len(sel["scores"]) == 6
    at.update([det], 1, 0.8)
    at.update([{"scores": torch.zeros(0), "classes": torch.zeros(0, dtype=torch.int64), "boxes": torch.zeros(0, 4)}], 2, 0.8)
    assert at.reserve_matrix[0].sum() == 0 and at.reserve_matrix[1].tolist() == [1, 2, 0, 1, 0, 0, 0, 0]


def test_bpc_restatement_matches_reference_function():
    """daod/loss/bpc_loss.py run on the CPU (oracle/gen_golden.py: gen_bpc): classes without ground truth,
    duplicated ground-truth boxes (IoU tie: the detection counts twice), an image without ground truth, a batch
    whose scores are all below 0.5."""
    from oracle import model as om
    fx = np.load(os.path.join(GOLDEN, "bpc_ref.npz"), allow_pickle=False)
    K = int(fx["num_classes"])
    for ci in range(int(fx["num_cases"])):
        gts, dts = [], []
        for b in range(3):
            gts.append((torch.from_numpy(fx[f"c{ci}_gt_boxes_{b}"]), torch.from_numpy(fx[f"c{ci}_gt_classes_{b}"])))
            dts.append({"boxes": torch.from_numpy(fx[f"c{ci}_dt_boxes_{b}"]),
                        "scores": torch.from_numpy(fx[f"c{ci}_dt_scores_{b}"]),
                        "classes": torch.from_numpy(fx[f"c{ci}_dt_classes_{b}"])})
        got = float(om.bpc_loss(K, gts, dts))
        assert abs(got - float(fx[f"c{ci}_loss"])) < 2e-6, (ci, got, float(fx[f"c{ci}_loss"]))
    assert float(om.bpc_loss(K, [], [])) == 0.0
