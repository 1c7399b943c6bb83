"""Checkpoint I/O in the reference's formats.

Mirrors what the reference reaches through Detectron2 / fvcore:
  * ``DetectionCheckpointer(model).resume_or_load(cfg.MODEL.WEIGHTS, resume=False)`` for the student AND
    the teacher at trainer construction (``daod/engine/trainers/source_free_adaptive_teacher.py:47-64``):
    ``.pth`` (``torch.save`` dict with a ``"model"`` state dict) or ``.pkl`` (pickle:
    ``{"model": {name: ndarray}, "__author__": ..., "matching_heuristics": True}``, the output of
    ``convert_pretrained_model/convert_vgg_bn.py:156``);
  * ``DetectionTSCheckpointer`` (``daod/checkpoint/detection_ts_checkpointer.py:10-89``) over
    ``EnsembleTSModel`` (``daod/modeling/meta_arch/ts_ensemble.py:6-15``): one dict with
    ``modelTeacher.*`` / ``modelStudent.*`` keys + optimizer / scheduler / iteration, ``last_checkpoint``
    file next to it, ``resume_or_load(resume=True)`` continues from it.
fvcore semantics kept: a ``"module."`` prefix common to all keys is stripped, tensors whose shape differs
from the model's are skipped (reported), missing / unexpected keys are reported, loading is in place
(``copy_`` into the existing parameters: they are views of the flat buffers here).  With
``matching_heuristics`` the checkpoint keys are matched to model keys by longest common suffix like
Detectron2's ``align_and_update_state_dicts`` (restated from its documented behaviour: D2 is not vendored).
"""
import logging
import os
import pickle
from collections import namedtuple

import numpy as np
import torch

IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys", "incorrect_shapes"])
logger = logging.getLogger("sfod.checkpoint")


def load_file(path):
    """-> checkpoint dict with at least a ``"model"`` entry."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        if "model" in data and "__author__" in data:
            return data
        # Detectron (Caffe2) model-zoo pickles ("blobs", or a bare name -> array dict) need d2's c2_model_loading
        # name conversion (conv1_w -> stem.conv1.weight, res2_0_branch2a_bn_s -> ...norm.weight, ...), which is not
        # restated here: suffix matching would find nothing and leave the model at its random initialisation
        raise ValueError("{}: Caffe2 / Detectron model-zoo pickles are not supported (no c2 name conversion); convert "
                         "the weights to a Detectron2-style checkpoint ({{'model': ..., '__author__': ...}}, e.g. with "
                         "the reference's convert_pretrained_model/convert_vgg_bn.py) first".format(path))
    data = torch.load(path, map_location="cpu", weights_only=False)
    if "model" not in data:
        data = {"model": data}
    return data


def _to_tensors(sd):
    out = {}
    for k, v in sd.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v.copy())
        if not isinstance(v, torch.Tensor):
            raise ValueError("Unsupported type found in checkpoint! {}: {}".format(k, type(v)))
        out[k] = v
    return out


def strip_prefix_if_present(sd, prefix):
    """fvcore ``_strip_prefix_if_present``: only when EVERY key carries the prefix."""
    keys = sorted(sd.keys())
    if not keys or not all(k.startswith(prefix) for k in keys):
        return sd
    return {k[len(prefix):]: v for k, v in sd.items()}


def align_by_suffix(model_keys, ckpt_sd):
    """Detectron2's matching heuristic: a checkpoint key is assigned to the model key it is the longest
    suffix of (a key that is a suffix of several model keys goes to the one with the longest match only)."""
    ckpt_keys = sorted(ckpt_sd.keys())
    out, used = {}, set()
    for mk in model_keys:
        best = None
        for ck in ckpt_keys:
            if mk == ck or mk.endswith("." + ck):
                if best is None or len(ck) > len(best):
                    best = ck
        if best is not None:
            out[mk] = ckpt_sd[best]
            used.add(best)
    for ck in ckpt_keys:                 # unmatched keys stay under their own name (-> unexpected)
        if ck not in used:
            out.setdefault(ck, ckpt_sd[ck])
    return out


def load_state_into(module, state_dict, heuristics=False):
    """In-place ``load_state_dict(strict=False)`` with fvcore's shape filter.  -> IncompatibleKeys."""
    sd = strip_prefix_if_present(_to_tensors(dict(state_dict)), "module.")
    model_sd = module.state_dict()
    if heuristics:
        sd = align_by_suffix(list(model_sd.keys()), sd)
    incorrect = []
    for k in list(sd.keys()):
        if k in model_sd and tuple(model_sd[k].shape) != tuple(sd[k].shape):
            incorrect.append((k, tuple(sd[k].shape), tuple(model_sd[k].shape)))
            sd.pop(k)
    inc = module.load_state_dict(sd, strict=False)
    missing = [k for k in inc.missing_keys if k not in ("pixel_mean", "pixel_std")]
    return IncompatibleKeys(missing, list(inc.unexpected_keys), incorrect)


def report_incompatible(inc, n_model_keys, path, who="model", require_backbone=True, model_keys=()):
    """What fvcore's Checkpointer logs after a load (missing / unexpected / shape-mismatched keys) -- plus a hard
    error when NOTHING (or no backbone tensor) matched: a checkpoint whose names do not fit would otherwise leave
    the model at its random initialisation without a word, and in source-free training the pseudo-labels then come
    from a random teacher."""
    missing, unexpected, incorrect = inc
    # shape-mismatched keys were popped before load_state_dict, so they are in ``missing`` as well: count them once
    matched = n_model_keys - len(set(missing) | {k for k, _, _ in incorrect})
    for k, cs, ms in incorrect:
        logger.warning("[%s] skip loading '%s' from %s: checkpoint shape %s, model shape %s", who, k, path, cs, ms)
    if missing:
        logger.warning("[%s] keys of the model missing in %s (left at their initialisation): %s", who, path,
                       ", ".join(sorted(missing)))
    if unexpected:
        logger.warning("[%s] keys of %s not used by the model: %s", who, path, ", ".join(sorted(unexpected)))
    logger.info("[%s] %s: %d of %d model tensors loaded", who, path, matched, n_model_keys)
    if matched <= 0:
        raise RuntimeError("{}: none of the checkpoint's {} keys matches a tensor of the {} "
                           "(first keys: {})".format(path, len(unexpected), who, sorted(unexpected)[:5]))
    if require_backbone:
        bb = [k for k in model_keys if k.startswith("backbone.")]
        if bb and all(k in set(missing) | {k2 for k2, _, _ in incorrect} for k in bb):
            raise RuntimeError("{}: no backbone tensor of the {} was matched by the checkpoint".format(path, who))


def load_model_weights(model, path, who="model"):
    """``DetectionCheckpointer(model).resume_or_load(path, resume=False)``; ``""`` is a no-op.  Incompatible keys are
    logged like fvcore does; a checkpoint that matches nothing raises."""
    if not path:
        return None
    if not os.path.isfile(path):
        raise FileNotFoundError("Checkpoint {} not found!".format(path))
    ckpt = load_file(path)
    sd = ckpt["model"]
    # an ensemble checkpoint given as MODEL.WEIGHTS: take the student's part (what a plain D2 checkpointer
    # would fail on; the reference's eval scripts strip the prefix the same way)
    if any(k.startswith("modelStudent.") for k in sd):
        sd = {k[len("modelStudent."):]: v for k, v in sd.items() if k.startswith("modelStudent.")}
    inc = load_state_into(model, sd, heuristics=bool(ckpt.get("matching_heuristics", False)))
    keys = [k for k in model.state_dict().keys() if k not in ("pixel_mean", "pixel_std")]
    report_incompatible(inc, len(keys), path, who=who, model_keys=keys)
    return inc


class DetectionTSCheckpointer:
    """Teacher + student ensemble checkpoints of a trainer (``save`` / ``resume_or_load``)."""

    def __init__(self, trainer, save_dir):
        self.trainer, self.save_dir = trainer, save_dir

    # ---- fvcore Checkpointer conventions ------------------------------------------------------------
    def _last_file(self):
        return os.path.join(self.save_dir, "last_checkpoint")

    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(self._last_file())

    def get_checkpoint_file(self):
        with open(self._last_file()) as f:
            return os.path.join(self.save_dir, f.read().strip())

    def save(self, name, **extra):
        os.makedirs(self.save_dir, exist_ok=True)
        data = self.trainer.state_dict_for_checkpoint()
        data.update(extra)
        fn = name + ".pth"
        torch.save(data, os.path.join(self.save_dir, fn))
        with open(self._last_file(), "w") as f:
            f.write(fn)
        return os.path.join(self.save_dir, fn)

    def load(self, path, with_training_state=True):
        tr = self.trainer
        ckpt = load_file(path)
        sd = strip_prefix_if_present(_to_tensors(dict(ckpt["model"])), "module.")
        if any(k.startswith("modelTeacher.") or k.startswith("modelStudent.") for k in sd):
            teacher = {k[len("modelTeacher."):]: v for k, v in sd.items() if k.startswith("modelTeacher.")}
            student = {k[len("modelStudent."):]: v for k, v in sd.items() if k.startswith("modelStudent.")}
            inc_s = load_state_into(tr.model, student)
            inc_t = load_state_into(tr.model_teacher, teacher) if getattr(tr, "model_teacher", None) is not None else None
        else:   # a plain (source-trained) model: "pretrained model weight: only update student model" (:13-24)
            inc_s = load_state_into(tr.model, sd, heuristics=bool(ckpt.get("matching_heuristics", False)))
            inc_t = None
        if with_training_state:
            if "optimizer" in ckpt and hasattr(tr.optimizer, "load_state_dict"):
                tr.optimizer.load_state_dict(ckpt["optimizer"])
            if "scheduler" in ckpt and hasattr(tr.scheduler, "load_state_dict"):
                tr.scheduler.load_state_dict(ckpt["scheduler"])
        return ckpt, inc_s, inc_t

    def resume_or_load(self, path, resume=True):
        """fvcore: with ``resume`` and a ``last_checkpoint`` file continue from it (model + optimizer +
        scheduler, ``start_iter = iteration + 1``); otherwise load ``path`` (model weights only)."""
        tr = self.trainer
        if resume and self.has_checkpoint():
            ckpt, _, _ = self.load(self.get_checkpoint_file(), with_training_state=True)
            tr.start_iter = int(ckpt.get("iteration", -1)) + 1
            tr.iter = tr.start_iter
            return ckpt
        if path:
            ckpt, _, _ = self.load(path, with_training_state=False)
            return ckpt
        return {}
