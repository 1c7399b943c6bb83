"""Oracle (test infrastructure): CPU fp32 restatement of Detectron2's ResNet-C4 backbone
(``build_resnet_backbone`` -- BasicStem + BottleneckBlock, STRIDE_IN_1X1, FREEZE_AT=2, NORM=BN), the
backbone the reference selects in ``configs/r101_c4_cs_foggy_adaptive_teacher_source_free.yaml:1-28``.

Detectron2 is not vendored under /root/reference and is not installed here, so this restates its published
structure over plain torch CPU ops; the state-dict keys are Detectron2's.
No Detectron2 vector pins it; a third-party cross-check does: HuggingFace ``transformers``' ResNet (installed)
configured as the same MSRA-style C4 trunk gives the same output from the same weights in eval (FrozenBN) and
train (batch statistics + running-statistics update) mode, depth 50 and 101
(tests/test_oracle_r101.py::test_resnet_trunk_equals_the_transformers_port).  Only tests/ and __graft_entry__.smoke() may import this module.
"""
import torch
import torch.nn.functional as F

BLOCKS = {50: [3, 4, 6], 101: [3, 4, 23]}


def block_specs(depth):
    """[(stage, idx, cin, cout, bottleneck, stride)] for res2..res4."""
    out = []
    cin, cout, bott = 64, 256, 64
    for si, n in enumerate(BLOCKS[depth]):
        for bi in range(n):
            out.append((f"res{si + 2}", bi, cin, cout, bott, (1 if si == 0 else 2) if bi == 0 else 1))
            cin = cout
        cout, bott = cout * 2, bott * 2
    return out


def frozen_bn(x, sd, prefix, eps=1e-5):
    scale = sd[prefix + ".weight"] * (sd[prefix + ".running_var"] + eps).rsqrt()
    shift = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def live_bn(x, sd, prefix, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d; in training mode the running statistics in `sd` are refreshed in place and
    ``num_batches_tracked`` (when the state carries it) advances by one, also under no_grad."""
    if training and prefix + ".num_batches_tracked" in sd:
        sd[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training, momentum, eps)


def forward(sd, x, depth=101, training=True, freeze_at=2, prefix="backbone."):
    """x: [N,3,H,W] normalised image batch -> res4 features.  Frozen stages use FrozenBN semantics."""
    p = prefix

    def norm(t, name, frozen):
        return frozen_bn(t, sd, name) if frozen else live_bn(t, sd, name, training)

    fz = freeze_at >= 1
    x = F.conv2d(x, sd[p + "stem.conv1.weight"], None, stride=2, padding=3)
    x = F.relu(norm(x, p + "stem.conv1.norm", fz))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for stage, bi, cin, cout, bott, stride in block_specs(depth):
        fz = freeze_at >= int(stage[3])
        b = f"{p}{stage}.{bi}."
        if cin != cout:
            sc = norm(F.conv2d(x, sd[b + "shortcut.weight"], None, stride=stride), b + "shortcut.norm", fz)
        else:
            sc = x
        o = F.relu(norm(F.conv2d(x, sd[b + "conv1.weight"], None, stride=stride), b + "conv1.norm", fz))
        o = F.relu(norm(F.conv2d(o, sd[b + "conv2.weight"], None, padding=1), b + "conv2.norm", fz))
        o = norm(F.conv2d(o, sd[b + "conv3.weight"], None), b + "conv3.norm", fz)
        x = F.relu(o + sc)
    return x


def init_state(depth=101, seed=0, freeze_at=2, prefix="backbone."):
    """Detectron2's initialisation of the C4 trunk under its state-dict keys: ``c2_msra_fill`` =
    ``kaiming_normal_(mode="fan_out", nonlinearity="relu")`` for every (bias-free) conv, norm weight 1 / bias 0,
    running mean 0 / var 1; frozen stages (FREEZE_AT: stem, res2) carry FrozenBatchNorm2d's four buffers
    (``running_var = 1 - eps`` as in its constructor, no counter), live stages nn.BatchNorm2d's five entries."""
    import math
    from collections import OrderedDict
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def conv(name, cout, cin, k, frozen, eps=1e-5):
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / (cout * k * k))
        sd[name + ".norm.weight"] = torch.ones(cout)
        sd[name + ".norm.bias"] = torch.zeros(cout)
        sd[name + ".norm.running_mean"] = torch.zeros(cout)
        sd[name + ".norm.running_var"] = torch.ones(cout) - (eps if frozen else 0.0)
        if not frozen:
            sd[name + ".norm.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    conv(prefix + "stem.conv1", 64, 3, 7, freeze_at >= 1)
    for stage, bi, cin, cout, bott, stride in block_specs(depth):
        fz = freeze_at >= int(stage[3])
        b = f"{prefix}{stage}.{bi}."
        if cin != cout:
            conv(b + "shortcut", cout, cin, 1, fz)
        conv(b + "conv1", bott, cin, 1, fz)
        conv(b + "conv2", bott, bott, 3, fz)
        conv(b + "conv3", cout, bott, 1, fz)
    return sd
