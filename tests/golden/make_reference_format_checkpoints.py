"""Writes checkpoints in the two on-disk formats of the REFERENCE stack, by hand from the reference's key lists
(test infrastructure for SURVEY 8f rank 2: ``simple-sfod_amd/checkpoint.py`` must read files it did not write).

(i)  ``<dir>/vgg16_bn.pkl`` -- the output format of ``convert_pretrained_model/convert_vgg_bn.py``:
     ``pickle.dump({"model": {name: ndarray}, "__author__": "torchvision", "matching_heuristics": True})``
     (``:142-157``), names ``backbone.vgg{b}.{i}.{weight,bias,running_mean,running_var}`` -- the list the script's own
     trailing comment prints (``:173-196``: ``vgg0.{0,1,3,4}``, ``vgg1.{0,1,3,4}``, ``vgg2..4.{0,1,3,4,6,7}``).  The
     input it converts is torchvision's ``vgg16_bn`` state dict as published (``features.N.*``, 6 tensors per
     conv + BatchNorm pair -- that checkpoint predates ``num_batches_tracked``; the script's ``78 = 13 * 6`` table
     ``:100-127`` relies on it), ``classifier.*`` is dropped (``:133-134``).
(ii) ``<dir>/model_0001999.pth`` + ``last_checkpoint`` -- what fvcore's ``Checkpointer.save`` writes for the
     reference trainer's ``DetectionTSCheckpointer(EnsembleTSModel(teacher, student), optimizer=, scheduler=)``
     (``daod/engine/trainers/source_free_adaptive_teacher.py:81-89``, ``daod/modeling/meta_arch/ts_ensemble.py:6-15``):
     ``torch.save({"model": {"modelTeacher.*", "modelStudent.*"}, "optimizer": torch.optim.SGD.state_dict(),
     "scheduler": {"base_lrs", "last_epoch"}, "iteration": it})``.  The optimizer entry is produced by a REAL
     ``torch.optim.SGD`` whose parameter groups are built like Detectron2's ``get_default_optimizer_params`` +
     ``reduce_param_groups`` (norm parameters: weight decay 0.0, everything else 1e-4; groups in first-seen order).

Values are seeded random numbers: the tests check names, containers, dtypes and that every tensor lands where it
belongs.  ``python tests/golden/make_reference_format_checkpoints.py --manifest`` rewrites
``tests/golden/ref_ckpt_manifest.json`` (names + shapes of both files; committed, the files themselves are generated
into a temporary directory by the tests: 60 MB / 200 MB).
"""
import json
import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
BLOCK_OF_CONV = [0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4]        # convert_vgg_bn.py:104-117 (``i // 6`` -> block)


def torchvision_vgg16_bn_state(seed=0):
    """torchvision ``vgg16_bn().state_dict()`` as in the published weights file: ``features.{n}`` indices of
    ``make_layers`` (conv, bn, relu per layer, one index per max-pool), no ``num_batches_tracked``."""
    g = torch.Generator().manual_seed(seed)
    sd, n, cin = {}, 0, 3
    for v in VGG16:
        if v == "M":
            n += 1
            continue
        sd[f"features.{n}.weight"] = torch.randn(v, cin, 3, 3, generator=g) * 0.05
        sd[f"features.{n}.bias"] = torch.randn(v, generator=g) * 0.01
        sd[f"features.{n + 1}.weight"] = torch.rand(v, generator=g) + 0.5
        sd[f"features.{n + 1}.bias"] = torch.randn(v, generator=g) * 0.1
        sd[f"features.{n + 1}.running_mean"] = torch.randn(v, generator=g) * 0.2
        sd[f"features.{n + 1}.running_var"] = torch.rand(v, generator=g) + 0.25
        n += 3
        cin = v
    for i, (o, k) in zip((0, 3, 6), ((8, 16), (8, 8), (4, 8))):     # stand-in classifier (dropped by the conversion)
        sd[f"classifier.{i}.weight"] = torch.zeros(o, k)
        sd[f"classifier.{i}.bias"] = torch.zeros(o)
    return sd


def reference_pkl_names():
    """The 78 target names in conversion order: per conv of a block the slots (3j, 3j+1) -- conv at 0/3/6, its
    BatchNorm at 1/4/7 (convert_vgg_bn.py:100 ``index_array`` + ``:140`` name format)."""
    names, seen = [], {}
    for c, b in enumerate(BLOCK_OF_CONV):
        j = seen.get(b, 0)
        seen[b] = j + 1
        names += [f"backbone.vgg{b}.{3 * j}.weight", f"backbone.vgg{b}.{3 * j}.bias"]
        names += [f"backbone.vgg{b}.{3 * j + 1}.{s}" for s in ("weight", "bias", "running_mean", "running_var")]
    return names


def write_pkl(path, seed=0):
    tv = torchvision_vgg16_bn_state(seed)
    feats = [(k, v) for k, v in tv.items() if not k.startswith("classifier")]
    names = reference_pkl_names()
    assert len(feats) == len(names) == 78
    model = {}
    for (k, v), new in zip(feats, names):
        assert k.rsplit(".", 1)[1] == new.rsplit(".", 1)[1]
        model[new] = v.detach().numpy()
    with open(path, "wb") as f:
        pickle.dump({"model": model, "__author__": "torchvision", "matching_heuristics": True}, f)
    return tv, model


def d2_param_groups(model, base_lr, weight_decay, weight_decay_norm):
    """Detectron2 ``get_default_optimizer_params`` + ``reduce_param_groups``: one entry per trainable parameter in
    module-walk order, merged by (lr, weight_decay) in first-seen order."""
    norm_types = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm,
                  torch.nn.GroupNorm, torch.nn.InstanceNorm1d, torch.nn.InstanceNorm2d, torch.nn.InstanceNorm3d,
                  torch.nn.LayerNorm, torch.nn.LocalResponseNorm)
    groups, index, memo = [], {}, set()
    for module in model.modules():
        for _, p in module.named_parameters(recurse=False):
            if not p.requires_grad or p in memo:
                continue
            memo.add(p)
            wd = weight_decay_norm if isinstance(module, norm_types) else weight_decay
            key = (base_lr, wd)
            if key not in index:
                index[key] = len(groups)
                groups.append({"params": [], "lr": base_lr, "weight_decay": wd})
            groups[index[key]]["params"].append(p)
    return groups


def write_ensemble_pth(save_dir, student, teacher, iteration=1999, base_lr=0.0025, seed=1):
    """-> (path, {student parameter name: momentum buffer}).  ``student`` / ``teacher``: torch modules whose
    ``state_dict()`` keys are the reference's (any model with the reference's names works: the product model built
    on the CPU, or a Detectron2 model on a machine that has it)."""
    g = torch.Generator().manual_seed(seed)
    opt = torch.optim.SGD(d2_param_groups(student, base_lr, 1e-4, 0.0), lr=base_lr, momentum=0.9)
    for p in student.parameters():
        if p.requires_grad:
            p.grad = torch.randn(p.shape, generator=g) * 1e-3
    opt.step()                                   # creates the momentum buffers (first step: buf = grad + wd * p)
    for p in student.parameters():
        p.grad = None
    sd = {}
    for k, v in teacher.state_dict().items():
        sd["modelTeacher." + k] = v
    for k, v in student.state_dict().items():
        sd["modelStudent." + k] = v
    data = {"model": sd, "optimizer": opt.state_dict(),
            "scheduler": {"base_lrs": [base_lr] * len(opt.param_groups), "last_epoch": iteration + 1},
            "iteration": iteration}
    os.makedirs(save_dir, exist_ok=True)
    name = "model_{:07d}.pth".format(iteration)
    torch.save(data, os.path.join(save_dir, name))
    with open(os.path.join(save_dir, "last_checkpoint"), "w") as f:
        f.write(name)
    ids = {id(p): n for n, p in student.named_parameters()}
    moms = {}
    for grp in opt.param_groups:
        for p in grp["params"]:
            st = opt.state.get(p, {})
            if "momentum_buffer" in st:
                moms[ids[id(p)]] = st["momentum_buffer"].clone()
    return os.path.join(save_dir, name), moms


def manifest():
    names = reference_pkl_names()
    tv = torchvision_vgg16_bn_state(0)
    feats = [v for k, v in tv.items() if not k.startswith("classifier")]
    return {
        "pkl": {"top_level": ["model", "__author__", "matching_heuristics"], "__author__": "torchvision",
                "matching_heuristics": True, "value_type": "numpy.ndarray float32",
                "model": {n: list(v.shape) for n, v in zip(names, feats)}},
        "pth": {"top_level": ["model", "optimizer", "scheduler", "iteration"],
                "model_prefixes": ["modelTeacher.", "modelStudent."],
                "optimizer": {"top_level": ["state", "param_groups"], "state_entry": ["momentum_buffer"],
                              "param_group_keys_read": ["lr", "weight_decay", "params"]},
                "scheduler": ["base_lrs", "last_epoch"]},
    }


if __name__ == "__main__":
    if "--manifest" in sys.argv:
        with open(os.path.join(HERE, "ref_ckpt_manifest.json"), "w") as f:
            json.dump(manifest(), f, indent=1, sort_keys=True)
        print("wrote", os.path.join(HERE, "ref_ckpt_manifest.json"))
    else:
        out = sys.argv[1] if len(sys.argv) > 1 else "."
        write_pkl(os.path.join(out, "vgg16_bn.pkl"))
        print("wrote", os.path.join(out, "vgg16_bn.pkl"))
