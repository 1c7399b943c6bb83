"""Optimiser, LR schedule and teacher EMA of the hot path, MI355X-first.

Restates Detectron2's ``build_optimizer`` (SGD momentum 0.9, weight decay 1e-4, 0.0 for norm
layers) + ``WarmupMultiStepLR`` and the reference's
``_update_teacher_model`` (``daod/engine/trainers/source_free_adaptive_teacher.py:583-603``).

Design: all parameters of a model live in ONE flat fp32 device buffer (the nn.Parameters are
views into it), gradients and momentum likewise.  One fused kernel per weight-decay group
applies  g += wd*p ; m = mu*m + g ; p -= lr*m  and -- when a teacher is attached -- the EMA
t = (1-k)*p + k*t  in the same pass (each value read once).  The flat gradient buffer is also the
single RCCL all-reduce payload of the data-parallel step.  The learning rate lives in a device
scalar, so the schedule never synchronises the stream.
"""
import bisect
from collections import OrderedDict

import torch

from .. import native


def _is_norm_module(m):
    return isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.GroupNorm, torch.nn.LayerNorm))


class FlatModelState:
    """Re-homes a model's parameters / buffers into flat device buffers (views keep their names)."""

    def __init__(self, model, frozen_prefixes=(), with_grad=True):
        norm_ids = set()
        self._gen_holders = [m for m in model.modules() if hasattr(m, "_frozen_gen")]
        for m in model.modules():
            if _is_norm_module(m):
                for p in m.parameters(recurse=False):
                    norm_ids.add(id(p))
        named = [(n, p) for n, p in model.named_parameters()]
        pref = tuple(frozen_prefixes)

        def is_frozen(n, p):  # not optimised: named prefixes, or requires_grad False (d2 freeze())
            return (bool(pref) and n.startswith(pref)) or not p.requires_grad

        decay = [(n, p) for n, p in named if id(p) not in norm_ids and not is_frozen(n, p)]
        norm = [(n, p) for n, p in named if id(p) in norm_ids and not is_frozen(n, p)]
        frozen = [(n, p) for n, p in named if is_frozen(n, p)]
        self.model = model
        self.order = decay + norm + frozen
        self.named_order = [n for n, _ in named]       # model.named_parameters() order (torch / Detectron2 indexing)
        dev = named[0][1].device

        def pad4(n):
            return (n + 3) // 4 * 4

        self.offsets = OrderedDict()
        off = 0
        bounds = []
        for group in (decay, norm, frozen):
            for n, p in group:
                self.offsets[n] = (off, p.numel(), tuple(p.shape))
                off += pad4(p.numel())
            bounds.append(off)
        self.n_decay, self.n_norm_end, self.n_total = bounds
        self.param = torch.zeros(max(self.n_total, 4), dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.param) if with_grad else None
        with torch.no_grad():
            for n, p in self.order:
                o, k, shp = self.offsets[n]
                view = self.param[o:o + k].view(shp)
                view.copy_(p.data)
                p.data = view
        self.params = OrderedDict(self.order)
        if with_grad:
            self.attach_grads()
        # fp32 buffers (BN running stats) and int64 buffers (num_batches_tracked)
        fb, ib = [], []
        for mname, m in model.named_modules():
            for bname, b in m._buffers.items():
                if b is None or bname in m._non_persistent_buffers_set:
                    continue
                (fb if b.dtype == torch.float32 else ib).append((m, bname, b))
        self.fbuf = torch.zeros(max(sum(pad4(b.numel()) for _, _, b in fb), 4), dtype=torch.float32, device=dev)
        self.ibuf = torch.zeros(max(len(ib), 1), dtype=torch.int64, device=dev)
        off = 0
        with torch.no_grad():
            for m, bname, b in fb:
                view = self.fbuf[off:off + b.numel()].view(b.shape)
                view.copy_(b)
                m._buffers[bname] = view
                off += pad4(b.numel())
            for i, (m, bname, b) in enumerate(ib):
                assert b.numel() == 1
                view = self.ibuf[i:i + 1].view(b.shape)
                view.copy_(b)
                m._buffers[bname] = view

    def values_rewritten(self):
        """Call after writing parameters / buffers through the flat buffers (EMA kernels, teacher <- student copies):
        modules that cache derived tensors of never-trained parameters (ResNet: FrozenBN folded into packed weights)
        drop them -- in-place writes through the flat buffer do not bump the views' version counters."""
        for m in self._gen_holders:
            m._frozen_gen += 1

    def attach_grads(self):
        for n, p in self.order:
            o, k, shp = self.offsets[n]
            p.grad = self.grad[o:o + k].view(shp)


class FusedSGD:
    """torch.optim-like facade (zero_grad / step / param_groups / state_dict) over FlatModelState."""

    def __init__(self, flat, lr, momentum=0.9, weight_decay=1e-4, weight_decay_norm=0.0):
        self.flat = flat
        self.momentum, self.weight_decay, self.weight_decay_norm = momentum, weight_decay, weight_decay_norm
        self.mom = torch.zeros_like(flat.param)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=flat.param.device)
        self.param_groups = [{"lr": float(lr), "initial_lr": float(lr)}]
        self._steps = 0
        self.teacher = None
        self.ema_keep = 0.0
        self.grad_scale = 1.0

    def attach_teacher(self, teacher_flat, keep_rate):
        """Fuse the teacher EMA (``_update_teacher_model``) into the parameter update.  A teacher key the student
        lacks is the reference's error (source_free_adaptive_teacher.py:600-601), raised once here instead of at
        every update."""
        for key in teacher_flat.offsets:
            if key not in self.flat.offsets:
                raise Exception("{} is not found in student model".format(key))
        assert list(teacher_flat.offsets.items()) == list(self.flat.offsets.items()), \
            "teacher and student must have identical parameter layouts"
        self.teacher, self.ema_keep = teacher_flat, float(keep_rate)

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = float(lr)
        self.lr_dev.fill_(float(lr))

    def zero_grad(self, set_to_none=False):
        self.flat.grad.zero_()
        for n, p in self.flat.order:
            if p.grad is None or p.grad.data_ptr() != self.flat.grad.data_ptr() + 4 * self.flat.offsets[n][0]:
                self.flat.attach_grads()
                break

    @torch.no_grad()
    def step(self, ema=True):
        f = self.flat
        for n, p in f.order:  # a trainer that replaced .grad (set_to_none) still works
            o, k, shp = f.offsets[n]
            if p.grad is not None and p.grad.data_ptr() != f.grad.data_ptr() + 4 * o:
                f.grad[o:o + k].view(shp).copy_(p.grad)
                p.grad = f.grad[o:o + k].view(shp)
        first = self._steps == 0
        t = self.teacher.param if (self.teacher is not None and ema) else None
        segs = [(0, f.n_decay, self.weight_decay), (f.n_decay, f.n_norm_end, self.weight_decay_norm)]
        for a, b, wd in segs:
            if b > a:
                native.sgd_ema_(f.param[a:b], f.grad[a:b], self.mom[a:b], None if t is None else t[a:b],
                                self.lr_dev, self.momentum, wd, self.grad_scale, self.ema_keep, first)
        if t is not None:
            if f.n_total > f.n_norm_end:  # frozen (never updated) parameters still take part in the EMA
                native.ema_(t[f.n_norm_end:f.n_total], f.param[f.n_norm_end:f.n_total], self.ema_keep)
            native.ema_(self.teacher.fbuf, f.fbuf, self.ema_keep)
            # int64 buffers: float32 arithmetic, truncated on the copy back (SURVEY A.17 iv)
            native.ema_i64_(self.teacher.ibuf, f.ibuf, self.ema_keep)
            self.teacher.values_rewritten()
        self._steps += 1

    def state_dict(self):
        return {"momentum_buffer": self.mom, "steps": self._steps, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        if "param_groups" in sd and "state" in sd:     # written by torch.optim.SGD (the reference stack's checkpoints)
            return self._load_torch_sgd_state(sd)
        self.mom.copy_(sd["momentum_buffer"])
        self._steps = sd["steps"]
        self.set_lr(sd["lr"])

    def torch_param_order(self, rule="norm"):
        """Names of the reference stack's optimiser parameters in ``torch.optim.SGD`` index order: Detectron2's
        ``get_default_optimizer_params`` walks the modules' own parameters (= ``named_parameters()`` order), skips
        ``requires_grad == False``, assigns per-parameter hyper-parameters, ``reduce_param_groups`` merges the
        per-parameter groups by hyper-parameters in first-seen order, and torch numbers the parameters group by group.
        Which parameters share hyper-parameters depends on the Detectron2 version that wrote the checkpoint -- ``rule``:
          "norm"         norm-layer weights AND biases take WEIGHT_DECAY_NORM, everything else WEIGHT_DECAY (current d2;
                         WEIGHT_DECAY_BIAS None / equal, BIAS_LR_FACTOR 1 as in the named yamls)
          "norm_weight"  a parameter called ``bias`` takes WEIGHT_DECAY_BIAS (= WEIGHT_DECAY by default) before the norm
                         test: only the norm WEIGHTS form the no-decay group (sizes [N + #norms, #norms])
          "single"       WEIGHT_DECAY_NORM == WEIGHT_DECAY: one group
        -> [[names of group 0], [names of group 1], ...]."""
        f = self.flat
        norm = {n for n, (o, _, _) in f.offsets.items() if f.n_decay <= o < f.n_norm_end}
        if rule == "single" or (rule == "norm" and self.weight_decay_norm == self.weight_decay):
            norm = set()
        elif rule == "norm_weight":
            norm = {n for n in norm if not n.endswith("bias")}
        groups, index = [], {}
        for n in f.named_order:
            if not f.params[n].requires_grad:
                continue
            key = n in norm
            if key not in index:
                index[key] = len(groups)
                groups.append([])
            groups[index[key]].append(n)
        return groups

    @torch.no_grad()
    def _load_torch_sgd_state(self, sd):
        """Momentum buffers + learning rate from a ``torch.optim.SGD.state_dict()`` (fvcore ``Checkpointer.save``
        stores it under "optimizer", daod/engine/trainers/source_free_adaptive_teacher.py:84-89).  Accepts the merged
        groups of current Detectron2, the merged groups of versions that put the norm biases with the decayed
        parameters (``torch_param_order``'s rules, chosen by group sizes and buffer shapes) and the one-group-per-parameter
        layout of older versions.  Parameters this
        optimiser does not update (zero-gradient domain-classifier heads with DOMAIN_CLASSIFIER off) are skipped."""
        f = self.flat
        theirs = [list(g["params"]) for g in sd["param_groups"]]

        def shapes_fit(names):
            for i, st in sd["state"].items():
                buf, n = st.get("momentum_buffer"), names.get(int(i))
                if buf is not None and n is not None and tuple(buf.shape) != tuple(f.offsets[n][2]):
                    return False
            return True

        # the grouping is read off the checkpoint: the rule whose group sizes AND momentum-buffer shapes fit it
        names, tried = None, []
        for rule in ("norm", "norm_weight", "single"):
            ours = self.torch_param_order(rule)
            tried.append([len(g) for g in ours])
            if [len(g) for g in theirs] == tried[-1]:
                cand = {i: n for g_t, g_o in zip(theirs, ours) for i, n in zip(g_t, g_o)}
                if shapes_fit(cand):
                    names = cand
                    break
        if names is None and all(len(g) == 1 for g in theirs) and len(theirs) == sum(tried[0]):
            order = [n for n in f.named_order if f.params[n].requires_grad]   # one group per parameter, in walk order
            names = {g[0]: n for g, n in zip(theirs, order)}
        if names is None:
            raise ValueError("optimizer state does not fit this model: checkpoint groups {} vs {} here".format(
                [len(g) for g in theirs], tried))
        self.mom.zero_()
        loaded = 0
        for i, st in sd["state"].items():
            buf = st.get("momentum_buffer")
            n = names.get(int(i))
            if buf is None or n is None:
                continue
            o, k, shp = f.offsets[n]
            if tuple(buf.shape) != tuple(shp):
                raise ValueError("momentum buffer {} of the checkpoint is {} but '{}' is {}".format(
                    i, tuple(buf.shape), n, tuple(shp)))
            if o >= f.n_norm_end:         # not updated here (see build_optimizer: never-run domain-classifier heads)
                continue
            self.mom[o:o + k].view(shp).copy_(buf)
            loaded += 1
        self._steps = 1 if loaded else 0      # torch initialises a missing buffer with the gradient: same as mu * 0 + g
        self.set_lr(sd["param_groups"][0]["lr"])
        return loaded


def build_optimizer(cfg, model, frozen=None):
    """d2 build_optimizer for SOLVER.{BASE_LR,MOMENTUM,WEIGHT_DECAY,WEIGHT_DECAY_NORM}.  ``frozen``: parameter prefixes that
    never receive a gradient in this trainer (default: the discriminators when DOMAIN_CLASSIFIER is off)."""
    if frozen is None:
        frozen = ()
        if "DOMAIN_CLASSIFIER" in cfg and not cfg.DOMAIN_CLASSIFIER.ENABLED:
            # the domain branch never runs: the reference leaves these grads None and SGD skips them
            frozen = ("DC_img.", "DC_ins.")
    flat = FlatModelState(model, frozen_prefixes=frozen)
    assert cfg.SOLVER.BIAS_LR_FACTOR == 1.0 and cfg.SOLVER.WEIGHT_DECAY_BIAS in (None, cfg.SOLVER.WEIGHT_DECAY)
    assert not cfg.SOLVER.CLIP_GRADIENTS.ENABLED and not cfg.SOLVER.NESTEROV
    return FusedSGD(flat, cfg.SOLVER.BASE_LR, cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY,
                    cfg.SOLVER.WEIGHT_DECAY_NORM)


class WarmupMultiStepLR:
    """d2 WarmupMultiStepLR; steps beyond MAX_ITER are dropped (the yamls list 360000 > 100000)."""

    def __init__(self, optimizer, cfg):
        self.optimizer = optimizer
        self.base_lr = cfg.SOLVER.BASE_LR
        self.milestones = sorted(s for s in cfg.SOLVER.STEPS if s <= cfg.SOLVER.MAX_ITER)
        self.gamma = cfg.SOLVER.GAMMA
        self.warmup_factor = cfg.SOLVER.WARMUP_FACTOR
        self.warmup_iters = cfg.SOLVER.WARMUP_ITERS
        assert cfg.SOLVER.WARMUP_METHOD == "linear" and cfg.SOLVER.LR_SCHEDULER_NAME == "WarmupMultiStepLR"
        self.last_epoch = 0
        self.optimizer.set_lr(self.get_lr(0))

    def get_lr(self, it):
        f = 1.0
        if it < self.warmup_iters:
            alpha = it / self.warmup_iters
            f = self.warmup_factor * (1 - alpha) + alpha
        return self.base_lr * f * self.gamma ** bisect.bisect_right(self.milestones, it)

    def step(self):
        self.last_epoch += 1
        self.optimizer.set_lr(self.get_lr(self.last_epoch))

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"]
        self.optimizer.set_lr(self.get_lr(self.last_epoch))
