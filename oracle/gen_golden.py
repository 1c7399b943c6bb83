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


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_vgg()
    gen_dann()
