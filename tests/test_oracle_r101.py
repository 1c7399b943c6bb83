"""CPU run of the oracle's ResNet-101-C4 detector (BASELINE config #5) end to end at a small frame size: the generic
oracle (oracle/model.py with Cfg.r101_c4(), oracle/resnet.py) is what tests/test_gpu_fullsize.py and
tests/test_gpu_trajectory.py hold the device to on that config; this keeps it exercised without a GPU.  Parity unpinned
(Detectron2 is not vendored): structure, shapes, gradient flow and the state side effects are what is checked here."""
import pytest
import torch

from oracle import model as om


def test_r101_c4_oracle_teacher_student_step():
    cfg = om.Cfg.r101_c4()
    assert cfg.num_anchors == 12 and cfg.stride == 16 and cfg.feat_channels == 1024 and cfg.fc_dim == 2048
    sd = om.init_state(cfg, seed=3)
    keys = list(sd)
    assert "backbone.res4.22.conv3.norm.running_mean" in keys and "backbone.stem.conv1.norm.running_var" in keys
    assert "backbone.res3.0.conv1.norm.num_batches_tracked" in keys            # live BatchNorm carries the counter ...
    assert "backbone.res2.0.conv1.norm.num_batches_tracked" not in keys        # ... FrozenBatchNorm2d does not
    assert tuple(sd["proposal_generator.rpn_head.objectness_logits.weight"].shape) == (12, 1024, 1, 1)
    assert tuple(sd["proposal_generator.rpn_head.anchor_deltas.weight"].shape) == (48, 1024, 1, 1)
    assert tuple(sd["roi_heads.box_head.fc1.weight"].shape) == (2048, 1024 * 49)
    frozen = [k for k in keys if k.startswith(om.FROZEN_PREFIXES)]
    assert frozen and not any(om.is_param(k) for k in frozen)
    g = torch.Generator().manual_seed(1)
    H, W, B = 96, 160, 2
    images = [torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g) for _ in range(B)]
    # ---- teacher: train-mode BatchNorm under no_grad refreshes the live statistics (AdaBN), frozen ones stay ----------
    t = om.clone_state(sd)
    before = {k: t[k].clone() for k in ("backbone.res3.0.conv1.norm.running_mean", "backbone.res2.0.conv1.norm.running_mean",
                                        "backbone.res4.5.conv2.norm.num_batches_tracked")}
    props, dets = om.teacher_forward(t, images, cfg)
    assert len(props) == B and all(p[0].shape[1] == 4 and len(p[0]) <= cfg.rpn_post_topk_train for p in props)
    assert all(len(p[0]) > 0 for p in props)
    assert not torch.equal(t["backbone.res3.0.conv1.norm.running_mean"], before["backbone.res3.0.conv1.norm.running_mean"])
    assert torch.equal(t["backbone.res2.0.conv1.norm.running_mean"], before["backbone.res2.0.conv1.norm.running_mean"])
    assert int(t["backbone.res4.5.conv2.norm.num_batches_tracked"]) == int(before["backbone.res4.5.conv2.norm.num_batches_tracked"]) + 1
    assert len(dets) == B
    # ---- student: losses on given boxes, gradients reach every live parameter and no frozen one -------------------------
    s = om.clone_state(sd, requires_grad=True)
    gtb = [torch.tensor([[10.0, 12.0, 70.0, 80.0], [80.0, 20.0, 150.0, 90.0]]), torch.tensor([[30.0, 30.0, 120.0, 88.0]])]
    gtc = [torch.tensor([1, 5]), torch.tensor([3])]
    Hf, Wf = (H + 15) // 16, (W + 15) // 16
    rpn_keys = [torch.randint(0, 2 ** 31 - 1, (Hf * Wf * cfg.num_anchors,), generator=g) for _ in range(B)]
    roi_keys = [torch.randint(0, 2 ** 31 - 1, (cfg.rpn_post_topk_train + 16,), generator=g) for _ in range(B)]
    losses, aux = om.student_losses(s, images, gtb, gtc, rpn_keys, roi_keys, cfg, return_aux=True)
    assert set(losses) >= {"loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_bpc"}
    assert all(torch.isfinite(v) for v in losses.values())
    assert tuple(aux["feat"].shape) == (B, 1024, Hf, Wf) and aux["logits"].shape[-1] == Hf * Wf * 12      # 12 anchors per location
    assert all(len(x["boxes"]) <= cfg.roi_batch for x in aux["samp"])
    sum(v for k, v in losses.items() if k != "loss_bpc").backward()
    live = [k for k in keys if om.is_param(k)]
    assert live and all(s[k].grad is not None and torch.isfinite(s[k].grad).all() for k in live)
    assert all(getattr(s[k], "grad", None) is None for k in frozen)
    assert float(s["backbone.res3.0.conv1.weight"].grad.abs().sum()) > 0 and float(s["roi_heads.box_head.fc1.weight"].grad.abs().sum()) > 0


def _hf_resnet_c4(depth):
    """An independent port of the same architecture: HuggingFace ``transformers``' ResNet (installed here) configured as
    Detectron2's MSRA-style C4 trunk -- bottleneck blocks, stride in the FIRST 1x1 (``downsample_in_bottleneck``), three
    stages.  -> the model and a name map from Detectron2's state-dict keys to its parameters / buffers."""
    from transformers import ResNetConfig, ResNetModel
    from oracle import resnet as R
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024], depths=R.BLOCKS[depth],
                       layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False,
                       downsample_in_bottleneck=True)
    m = ResNetModel(cfg)
    names = {}

    def conv_bn(d2, hf):            # a ResNetConvLayer (convolution + normalization) <- a Detectron2 Conv2d with .norm
        names[d2 + ".weight"] = hf + ".convolution.weight"
        for k in ("weight", "bias", "running_mean", "running_var"):
            names[f"{d2}.norm.{k}"] = f"{hf}.normalization.{k}"

    conv_bn("backbone.stem.conv1", "embedder.embedder")
    for stage, bi, cin, cout, _, _ in R.block_specs(depth):
        hf = f"encoder.stages.{int(stage[3]) - 2}.layers.{bi}"
        if cin != cout:
            conv_bn(f"backbone.{stage}.{bi}.shortcut", hf + ".shortcut")
        for j in range(3):
            conv_bn(f"backbone.{stage}.{bi}.conv{j + 1}", f"{hf}.layer.{j}")
    return m, names


@pytest.mark.parametrize("depth", [50, 101])
def test_resnet_trunk_equals_the_transformers_port(depth):
    """Third-party cross-check of ``oracle/resnet.py`` (Detectron2 absent): the same weights and statistics through
    HuggingFace's ResNet.  Eval mode = FrozenBN everywhere (freeze_at 5 in the oracle); train mode = batch statistics in
    every BatchNorm (freeze_at 0) incl. the running-statistics update with momentum 0.1 and the unbiased variance."""
    from oracle import resnet as R
    m, names = _hf_resnet_c4(depth)
    g = torch.Generator().manual_seed(depth)
    sd = R.init_state(depth, seed=3, freeze_at=0)
    for k in list(sd):
        if k.endswith("norm.weight"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        elif k.endswith("norm.bias") or k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    hf = dict(m.named_parameters())
    hf.update(dict(m.named_buffers()))
    used = set()
    with torch.no_grad():
        for d2, name in names.items():
            assert hf[name].shape == sd[d2].shape, (d2, name)
            hf[name].copy_(sd[d2])
            used.add(name)
    assert {n for n in hf if not n.endswith("num_batches_tracked")} == used       # the two trunks have the same tensors
    x = torch.randn(2, 3, 67, 93, generator=g)
    # eval: every norm frozen
    m.eval()
    with torch.no_grad():
        ref = m(x).last_hidden_state
        got = R.forward({k: v.clone() for k, v in sd.items()}, x, depth=depth, training=False, freeze_at=5)
    assert got.shape == ref.shape == (2, 1024, 5, 6)
    assert ((got - ref).norm() / ref.norm()).item() < 2e-6
    # train: batch statistics, running statistics refreshed
    m.train()
    sd2 = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        ref = m(x).last_hidden_state
        got = R.forward(sd2, x, depth=depth, training=True, freeze_at=0)
    assert ((got - ref).norm() / ref.norm()).item() < 2e-5
    hf2 = dict(m.named_buffers())
    for d2, name in names.items():
        if d2.endswith("running_mean") or d2.endswith("running_var"):
            torch.testing.assert_close(sd2[d2], hf2[name], rtol=1e-4, atol=1e-6)
    assert int(sd2["backbone.res4.0.conv1.norm.num_batches_tracked"]) == 1
