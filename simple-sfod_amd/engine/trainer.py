"""Step drivers of the hot path: ``BaseTrainer`` (source-only) and
``SourceFreeAdaptiveTeacherTrainer`` (teacher -> pseudo-labels -> student -> SGD -> EMA).

Mirrors (reference file:line)
  daod/engine/trainers/base.py:30-123,186-220     BaseTrainer.__init__/run_step/_write_metrics
  daod/engine/trainers/base.py:270-337            AdaBN refinement (reset + <=1400 no-grad train-mode forwards)
  daod/engine/trainers/source_free_adaptive_teacher.py
      :39-94    __init__  (student + teacher from the same weights, DDP around the student)
      :150-183  threshold_bbox      :256-280 process_pseudo_label
      :335-581  run_step            :583-603 _update_teacher_model (EMA, keep rate 0.9996)
  train_net_mt.py:45-87                           trainer dispatch on cfg.TRAINER

MI355X-first differences (documented in DESIGN.md):
  * no host synchronisation inside a step: pseudo-labels stay in fixed-capacity device arrays,
    metrics are accumulated on the device and read every ``WRITER_PERIOD`` iterations
    (the reference syncs on every ``.item()`` and gathers a pickled dict over gloo each step);
  * DDP is replaced by one RCCL all-reduce of the flat gradient buffer (no per-parameter hooks,
    parameters without gradient in a step are zero-filled identically on all ranks);
  * the EMA is fused into the SGD kernel; ``SFOD.EMA.ENABLED`` exposes quirk q1 (the dispatched
    reference trainer has the call commented out, its ``_single`` twin has it on);
  * zero-weighted dead branches (2nd ROI pass, BPC, domain classifier) are elided
    (``SFOD.ELIDE_DEAD_BRANCHES``).
"""
import collections
import contextlib
import json
import os
import time

import torch
import torch.distributed as dist

from .. import native
from ..data.synthetic import TwoCropLoader
from ..modeling import build_model
from ..modeling import offchain
from ..modeling.batched import BatchedGT
from ..structures import Boxes, Instances
from .solver import FlatModelState, WarmupMultiStepLR, build_optimizer

WRITER_PERIOD = 20  # hooks.PeriodicWriter(period=20), source_free_adaptive_teacher.py:679


# Stage markers for rocprofv3 --marker-trace (the reference has no profiler hooks; d2's would be
# torch.profiler).  SFOD_ROCTX=1 (or SFOD.PROFILE_RANGES True): roctx ranges "sfod/teacher", "sfod/student_forward",
# "sfod/student_backward", "sfod/exchange", "sfod/update" around the stages of run_step, so that a kernel trace can be cut
# by stage.  The ranges are HOST-side: pushed and popped when the stage is ENQUEUED, and the host runs up to
# SFOD.MAX_STEPS_IN_FLIGHT steps ahead of the GPU, so their timestamps do not bracket the GPU work -- attribute kernels to a
# stage through the correlation ids of the launches made inside a range (rocprofv3 --marker-trace --kernel-trace writes both;
# tools/step_timeline.py reads the kernel trace per stream), not by time.  Off: a no-op context manager, nothing is pushed.
_ROCTX_ENV = os.environ.get("SFOD_ROCTX", "0") == "1"
_ROCTX = [_ROCTX_ENV]


@contextlib.contextmanager
def stage(name):
    if not _ROCTX[0]:
        yield
        return
    torch.cuda.nvtx.range_push("sfod/" + name)      # = roctxRangePush on ROCm builds of torch
    try:
        yield
    finally:
        torch.cuda.nvtx.range_pop()


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class EventStorage:
    """Scalars are kept as device tensors until ``flush`` (one host sync per writer period).

    Two kinds of scalars, as in the reference: the step's ``metrics_dict`` (``_write_metrics``, base.py:198-209:
    gathered from every rank, logged as the mean over ranks, ``data_time`` as the max) is put with ``rank_mean=True``;
    everything else a trainer or a hook puts (``roi_head/mean_confidence``, ``acc_thres/*``, ``calibration/bpc_loss``,
    the ValLossHook's ``*_val`` keys, evaluation results) is rank-local there -- each rank has its own storage and only
    the main process writes -- so the record carries this rank's value and rank 0's is what reaches ``metrics.json``."""

    SMOOTH_WINDOW = 20       # d2 JSONWriter(window_size=20) / EventStorage.latest_with_smoothing_hint

    def __init__(self, start_iter=0):
        self.iter = start_iter
        self._pending = {}
        self._rank_mean = set()
        self._window = {}        # name -> the last SMOOTH_WINDOW values put with a smoothing hint (d2 HistoryBuffer)
        self.history = []

    def put_scalar(self, name, value, rank_mean=False, smoothing_hint=True):
        """d2 ``EventStorage.put_scalar``: ``smoothing_hint`` (default True, as there) says the writers may report the
        median of the last values instead of the last one -- the reference puts every loss / statistic with the default,
        d2's EvalHook and LR hook put theirs with ``False``."""
        if isinstance(value, torch.Tensor) and value.requires_grad:
            value = value.detach()          # a logged scalar must not keep its autograd graph (and everything it saved) alive
        self._pending[name] = value
        (self._rank_mean.add if rank_mean else self._rank_mean.discard)(name)
        if smoothing_hint:
            w = self._window.get(name)
            if w is None:
                w = self._window[name] = collections.deque(maxlen=self.SMOOTH_WINDOW)
            w.append(value)
        else:
            self._window.pop(name, None)

    def put_scalars(self, _rank_mean=False, smoothing_hint=True, **kw):
        """``_rank_mean`` is positional / underscored so that a metric may itself be called ``rank_mean``."""
        for k, v in kw.items():
            self.put_scalar(k, v, rank_mean=_rank_mean, smoothing_hint=smoothing_hint)

    def _smoothed(self, name):
        """median of the window (numpy's: the mean of the two middle values for an even count), on the device for device
        scalars -- no host synchronisation here"""
        w = self._window.get(name)
        if w is None or len(w) < 2:
            return self._pending[name]
        n = len(w)
        if all(isinstance(v, torch.Tensor) for v in w):
            srt = torch.sort(torch.stack([v.detach().float().reshape(()) for v in w])).values
            return (srt[(n - 1) // 2] + srt[n // 2]) * 0.5
        srt = sorted(float(v) for v in w)
        return 0.5 * (srt[(n - 1) // 2] + srt[n // 2])

    def flush(self, reduce_over_ranks=False, smooth=False):
        """-> the record of this writer period.  ``reduce_over_ranks``: the ``rank_mean`` keys of the period are stacked
        and averaged with ONE small all-reduce (+ one MAX all-reduce for ``data_time``); every rank must call it with
        the same ``rank_mean`` keys (they follow from the config, not from the data).  If the ranks disagree on their
        number anyway, nothing is exchanged (a mismatched collective would hang), every rank keeps its own values and
        says so once on stderr."""
        if not self._pending:
            return {}
        if smooth:      # the periodic writer (d2 JSONWriter): hinted scalars as the median of their last 20 values.  With
            # N > 1 ranks this is the rank mean of per-rank medians, where d2 takes the median of per-step rank means
            # (its _write_metrics gathers every step; here nothing is exchanged between writer periods): equal at N = 1
            self._pending = {n_: self._smoothed(n_) for n_ in self._pending}
        names = list(self._pending)
        vals = [v.detach().float().reshape(()) if isinstance(v, torch.Tensor) else None for v in self._pending.values()]
        world = get_world_size() if reduce_over_ranks else 1
        if world > 1:
            # host floats among the rank-mean keys are averaged like the device scalars (the reference's _write_metrics
            # averages every key of the gathered dicts); data_time keeps its own MAX reduction below
            dev0 = next((v.device for v in vals if v is not None), torch.device("cuda", torch.cuda.current_device())
                        if torch.cuda.is_available() else torch.device("cpu"))
            for i, n_ in enumerate(names):
                if vals[i] is None and n_ in self._rank_mean and n_ != "data_time":
                    vals[i] = torch.tensor(float(self._pending[n_]), dtype=torch.float32, device=dev0)
        dev_idx = [i for i, v in enumerate(vals) if v is not None]
        shared = [i for i in dev_idx if names[i] in self._rank_mean] if world > 1 else []
        has_dt = world > 1 and "data_time" in self._rank_mean and "data_time" in self._pending
        if world > 1:
            dev = vals[dev_idx[0]].device if dev_idx else (torch.device("cuda", torch.cuda.current_device())
                                                        if torch.cuda.is_available() else torch.device("cpu"))
            n = torch.tensor([len(shared), -len(shared), int(has_dt), -int(has_dt)], dtype=torch.int64, device=dev)
            dist.all_reduce(n, op=dist.ReduceOp.MAX)
            if int(n[0]) != -int(n[1]) or int(n[2]) != -int(n[3]):
                if not getattr(self, "_warned_mismatch", False):
                    import sys
                    print("[EventStorage] rank {}: the ranks bring different numbers of rank-mean scalars ({} here, up to "
                          "{} elsewhere); logging rank-local values".format(get_rank(), len(shared), int(n[0])),
                          file=sys.stderr)
                    self._warned_mismatch = True
                shared, has_dt = [], False
        host = {}
        if dev_idx:
            stacked = torch.stack([vals[i] for i in dev_idx])
            if shared:
                pos = {i: j for j, i in enumerate(dev_idx)}
                sel = torch.tensor([pos[i] for i in shared], dtype=torch.int64, device=stacked.device)
                part = stacked[sel]
                dist.all_reduce(part)
                stacked = stacked.index_copy(0, sel, part / world)
            host = dict(zip(dev_idx, stacked.cpu().tolist()))
        out = {"iteration": self.iter}
        for i, n_ in enumerate(names):
            out[n_] = host[i] if i in host else float(self._pending[n_])
        if has_dt:
            t = torch.tensor([out["data_time"]], dtype=torch.float32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out["data_time"] = t.item()
        self._pending = {}
        self._rank_mean = set()
        self.history.append(out)
        return out


class GradientReducer:
    """Sum all-reduce of the flat gradient buffer in phases so that communication overlaps the backbone
    backward.  ``launch_early()`` (called when the backbone's backward starts, i.e. when every gradient of the
    heads is final) starts an asynchronous all-reduce of the contiguous heads slice on the collective's own
    stream; ``launch_mid()`` (called by the backbone's backward once the slice it named in ``reduce_schedule`` is
    written -- VGG16: stages vgg2..vgg4 = 96 % of the trunk's parameters, with the two most expensive stages of the
    backward still to run; ResNet-C4: res4 = 95 % of the live trunk, res3 still to run) does the same for that slice;
    ``finish()`` waits for both and reduces what is left (VGG: vgg0/vgg1, < 1 MB; R101: res3, 4.9 MB).  xGMI is
    point-to-point, so few large messages (not 25 MB DDP buckets) keep every link busy."""

    def __init__(self, flat, prefixes=("proposal_generator.", "roi_heads."),
                 mid_prefixes=("backbone.vgg2.", "backbone.vgg3.", "backbone.vgg4."), skip_prefixes=()):
        """``skip_prefixes``: parameters whose gradient is identically zero on every rank in this run (a zero-weighted,
        elided domain classifier: 3.1 M parameters of the hot yaml) -- never exchanged."""
        self.flat = flat
        self.work, self.work_mid = None, None
        self.lo, self.hi = self._pure_run(prefixes)
        self.mlo, self.mhi = self._pure_run(mid_prefixes) if mid_prefixes else (0, 0)
        if self.mhi > self.mlo and self.hi > self.lo and not (self.mhi <= self.lo or self.hi <= self.mlo):
            self.mlo = self.mhi = 0          # overlapping slices: keep the heads phase only
        self.skip = self._runs(tuple(skip_prefixes)) if skip_prefixes else []

    def _runs(self, prefixes):
        """every maximal run of consecutive parameters (buffer order) matching ``prefixes`` -> [(lo, hi)] in elements"""
        items = sorted(((o, n) for n, (o, k, _) in self.flat.offsets.items()), key=lambda t: t[0])
        out, cur = [], None
        for i, (o, n) in enumerate(items):
            nxt = items[i + 1][0] if i + 1 < len(items) else self.flat.grad.numel()
            if n.startswith(prefixes):
                cur = (cur[0], nxt) if cur is not None else (o, nxt)
            elif cur is not None:
                out.append(cur)
                cur = None
        if cur is not None:
            out.append(cur)
        return out

    def final_elements(self):
        """elements the blocking phase exchanges after the backward (diagnostics / tests)"""
        end = int(getattr(self.flat, "n_norm_end", self.flat.grad.numel()))
        covered = sorted([(self.lo, self.hi), (self.mlo, self.mhi)] + list(self.skip))
        pos = tot = 0
        for lo, hi in covered + [(end, end)]:
            lo, hi = min(lo, end), min(hi, end)
            if lo > pos:
                tot += lo - pos
            pos = max(pos, hi)
        return tot

    def _pure_run(self, prefixes):
        """Longest run of consecutive parameters (in buffer order) that all match ``prefixes`` -> [lo, hi) in
        elements, hi rounded up to the 16-byte padding of the last parameter; (0, 0) if there is none."""
        prefixes = tuple(prefixes)
        items = sorted(((o, o + k, n) for n, (o, k, _) in self.flat.offsets.items()), key=lambda t: t[0])
        best, cur = (0, 0), None
        for i, (o, e, n) in enumerate(items):
            if n.startswith(prefixes):
                nxt = items[i + 1][0] if i + 1 < len(items) else self.flat.grad.numel()
                cur = (cur[0], nxt) if cur is not None else (o, nxt)      # up to the next parameter (padding)
                if cur[1] - cur[0] > best[1] - best[0]:
                    best = cur
            else:
                cur = None
        return best

    # measurement only (bench.py's ``exchange`` block): run a subset of the phases to see what each one exposes
    phases = ("early", "mid", "final")

    def launch_early(self):
        if "early" in self.phases and self.work is None and self.hi > self.lo and get_world_size() > 1:
            self.work = dist.all_reduce(self.flat.grad[self.lo:self.hi], async_op=True)

    def launch_mid(self):
        if "mid" in self.phases and self.work_mid is None and self.mhi > self.mlo and get_world_size() > 1:
            self.work_mid = dist.all_reduce(self.flat.grad[self.mlo:self.mhi], async_op=True)

    def finish(self):
        # parameters that are not optimised (frozen stages, a disabled domain classifier) sit behind ``n_norm_end`` in the
        # flat buffer with zero gradients on every rank: nothing to exchange there
        end = int(getattr(self.flat, "n_norm_end", self.flat.grad.numel()))
        g = self.flat.grad[:end] if 0 < end < self.flat.grad.numel() else self.flat.grad
        if "final" not in self.phases:       # timing variant: only what was launched asynchronously
            for w in (self.work, self.work_mid):
                if w is not None:
                    w.wait()
            self.work = self.work_mid = None
            return
        done = sorted([(lo, hi) for lo, hi, w in ((self.lo, self.hi, self.work), (self.mlo, self.mhi, self.work_mid))
                       if w is not None] + [(min(lo, g.numel()), min(hi, g.numel())) for lo, hi in self.skip])
        pos = 0
        for lo, hi in done + [(g.numel(), g.numel())]:
            if lo > pos:
                dist.all_reduce(g[pos:lo])
            pos = max(pos, hi)
        for w in (self.work, self.work_mid):
            if w is not None:
                w.wait()
        self.work = self.work_mid = None


def _apply_process_knobs(cfg):
    """Config keys that are process-wide library settings: applied when a trainer is constructed."""
    if "DETERMINISTIC" in cfg.SFOD and torch.device(cfg.MODEL.DEVICE).type == "cuda":
        # the mode is on when EITHER the config key or the SFOD_DETERMINISTIC=1 environment setting asks for it
        # (include/sfod_hip.h documents both): a trainer built from a default config no longer switches a user's
        # environment setting off.  Process-wide: the last trainer constructed decides.
        want = bool(cfg.SFOD.DETERMINISTIC) or os.environ.get("SFOD_DETERMINISTIC", "0") == "1"
        native.set_deterministic(want)
    if "PROFILE_RANGES" in cfg.SFOD:
        # like DETERMINISTIC: on when EITHER the config key or the environment asks; the last trainer constructed decides
        _ROCTX[0] = bool(cfg.SFOD.PROFILE_RANGES) or _ROCTX_ENV


def _throttle(trainer):
    """Keep the host at most ``SFOD.MAX_STEPS_IN_FLIGHT`` steps ahead of the GPU.  A step has no host synchronisation and
    the host enqueues it ~10x faster than the GPU runs it, so an unbounded host runs as far ahead as the HIP queues let it;
    every tensor that crossed streams (teacher stream, weight-gradient side stream, loader stream: ``record_stream``) stays
    reserved until the GPU gets there, and torch's caching allocator grew to 137 GB for 14 GB of live tensors on the hot
    yaml at B = 8 (246 of 288 GB on 1024 x 2048 tensors, where the next process then met a device still being emptied).
    Waiting on the event recorded after step i - depth costs nothing while the GPU is the slower side."""
    depth = _in_flight_depth(trainer)
    if depth <= 0:
        return
    ev = trainer.__dict__.setdefault("_step_events", collections.deque())
    while len(ev) >= depth:
        ev.popleft().synchronize()


def _in_flight_depth(trainer):
    """``trainer._max_in_flight`` (settable: bench.py lifts the bound for its host-enqueue measurement), initialised from the
    config the first time a step runs; 0 for anything that is not a trainer on a GPU (the tests' recorder stubs)."""
    d = trainer.__dict__ if hasattr(trainer, "__dict__") else {}
    if "_max_in_flight" not in d:
        cfg, dev = d.get("cfg"), d.get("device")
        on_gpu = dev is not None and torch.device(dev).type == "cuda"
        depth = 0
        if on_gpu and cfg is not None and "SFOD" in cfg:
            depth = int(cfg.SFOD.MAX_STEPS_IN_FLIGHT) if "MAX_STEPS_IN_FLIGHT" in cfg.SFOD else 2
        try:
            trainer._max_in_flight = depth
        except AttributeError:
            return depth
    return trainer._max_in_flight


@contextlib.contextmanager
def _on_step_stream(trainer):
    """Run a LOOP of training steps on the trainer's own HIGH-priority stream (``SFOD.STEP_STREAM_PRIORITY``, default -1; 0:
    the caller's stream).  The student's chain -- forward, losses, data-gradient chain, update -- is the step's critical
    path; the weight gradients it feeds run on a normal-priority side stream, and as workgroup slots free up the dispatcher
    hands them to the chain first (+1.0 % on the hot yaml at B = 8, +0.65 % on the R101 yaml).  Around the whole loop, not
    around each step: switching streams per step -- a wait on the caller's (null) stream at every step boundary -- cost
    4-7 % on the R101 yaml / at one frame per GPU (profiles/round5/r5_step_stream_priority.txt).  Ordered after everything the
    caller's stream holds at entry; the caller's stream waits for the loop at exit.  ``run_step`` called on its own (tests)
    runs on the caller's stream as before."""
    d = trainer.__dict__ if hasattr(trainer, "__dict__") else {}
    cfg, dev = d.get("cfg"), d.get("device")
    prio = 0
    if dev is not None and torch.device(dev).type == "cuda" and cfg is not None and "SFOD" in cfg:
        prio = int(cfg.SFOD.STEP_STREAM_PRIORITY) if "STEP_STREAM_PRIORITY" in cfg.SFOD else -1
    if prio >= 0:
        yield
        return
    st = d.get("_step_stream")
    if st is None:
        st = trainer._step_stream = torch.cuda.Stream(device=dev, priority=prio)
    caller = torch.cuda.current_stream()
    if caller == st:                     # nested (the single-model trainer calls its parent's run_step)
        yield
        return
    st.wait_stream(caller)
    try:
        with torch.cuda.stream(st):
            yield
    finally:
        # also when a step raised (check_finite's FloatingPointError is a designed path): a handler that checkpoints or
        # evaluates on the caller's stream must not read parameters while the SGD + EMA update is still in flight here
        caller.wait_stream(st)
        torch.cuda.current_stream().wait_stream(st)


def _step_enqueued(trainer):
    if _in_flight_depth(trainer) > 0:
        e = torch.cuda.Event()
        e.record()                      # main stream: the update at the end of a step has waited for every side stream
        trainer.__dict__.setdefault("_step_events", collections.deque()).append(e)


class BaseTrainer:
    """Source-only training (``TRAINER: "base"``)."""

    def __init__(self, cfg, data_loader=None):
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        _apply_process_knobs(cfg)
        self.model = self.build_model(cfg)
        self.optimizer = self.build_optimizer(cfg, self.model)
        if cfg.MODEL.WEIGHTS:     # DetectionCheckpointer(model).resume_or_load(MODEL.WEIGHTS, resume=False), train_net_mt.py:75
            from ..checkpoint import load_model_weights
            load_model_weights(self.model, cfg.MODEL.WEIGHTS, who="model")
        self._broadcast_initial_state()
        self._attach_reducer()
        self.scheduler = WarmupMultiStepLR(self.optimizer, cfg)
        self.data_loader = data_loader or self.build_train_loader(cfg)
        self._data_loader_iter = iter(self.data_loader)
        self.start_iter, self.max_iter = 0, cfg.SOLVER.MAX_ITER
        self.iter = 0
        self.storage = EventStorage(0)
        self.model.train()

    def _broadcast_initial_state(self):
        """DDP's constructor broadcast: every rank starts from rank 0's parameters and buffers -- also after loading
        MODEL.WEIGHTS, because keys the checkpoint lacks (DC_img / DC_ins when a source-only GeneralizedRCNN
        checkpoint goes into the adaptive-teacher architecture, shape-mismatched heads) keep their per-rank random
        initialisation (seed = SEED + rank) and only gradients are exchanged afterwards."""
        if get_world_size() > 1:
            f = self.optimizer.flat
            for t in (f.param, f.fbuf, f.ibuf):
                if t.numel():
                    dist.broadcast(t, 0)
            f.values_rewritten()

    def _attach_reducer(self):
        """N > 1: overlap the heads' gradient all-reduce with the backbone backward.  The backbone's autograd
        node calls ``_pre_backward`` when its backward starts -- every gradient of the RPN / ROI heads is final
        by then, unless a live domain-classifier branch adds a second pass through the backbone."""
        cfg = self.cfg
        dc_live = ("DOMAIN_CLASSIFIER" in cfg and cfg.DOMAIN_CLASSIFIER.ENABLED and
                   (cfg.DOMAIN_CLASSIFIER.IMAGE or cfg.DOMAIN_CLASSIFIER.INSTANCE or not cfg.SFOD.ELIDE_DEAD_BRANCHES))
        self._reducer = None
        if get_world_size() > 1 and not dc_live:
            # the mid phase: the backbone says which slice of its gradients is final while its backward still runs and calls
            # ``_mid_backward`` there (VGG: the deep stages once vgg<s> is done; ResNet-C4: res4 once its first block is done)
            total = cfg.SOLVER.IMS_PER_BATCH if type(self) is BaseTrainer else cfg.SOLVER.IMS_PER_BATCH_TARGET
            b_local = max(1, int(total) // get_world_size())
            short = min(cfg.INPUT.MIN_SIZE_TRAIN) if len(cfg.INPUT.MIN_SIZE_TRAIN) else 600
            bb = self.model.backbone
            mids = tuple(bb.reduce_schedule(b_local * short * short * 2)) if hasattr(bb, "reduce_schedule") else ()
            # (this branch runs only when the domain classifier contributes no gradient: its slots stay zero everywhere)
            self._reducer = GradientReducer(self.optimizer.flat, mid_prefixes=mids, skip_prefixes=("DC_img.", "DC_ins."))
            bb._pre_backward = self._reducer.launch_early
            bb._mid_backward = self._reducer.launch_mid if mids else None

    @classmethod
    def build_model(cls, cfg):
        return build_model(cfg)

    @classmethod
    def build_optimizer(cls, cfg, model):
        return build_optimizer(cfg, model)

    def build_train_loader(self, cfg):
        return TwoCropLoader(cfg, self.device, get_rank(), get_world_size(), labeled=True)

    # ---- evaluation (d2 DefaultTrainer.test; base.py:125-171) ------------------------------------------
    @classmethod
    def build_test_loader(cls, cfg, dataset_name):
        from ..data import TestLoader
        return TestLoader(cfg, torch.device(cfg.MODEL.DEVICE), get_rank(), get_world_size(), dataset_name=dataset_name)

    @classmethod
    def build_evaluator(cls, cfg, dataset_name, output_folder=None, data_loader=None):
        """``NewCOCOEvaluator`` for the "coco" evaluator type (base.py:133-142); the synthetic target set
        stands in for the registered datasets (no dataset files in the build environment)."""
        from ..data import CITYSCAPES_CLASSES, TestLoader
        from ..evaluation import NewCOCOEvaluator
        loader = data_loader or TestLoader(cfg, torch.device("cpu"), dataset_name=dataset_name)
        names = getattr(loader.dataset, "class_names", None) or CITYSCAPES_CLASSES[: cfg.MODEL.ROI_HEADS.NUM_CLASSES]
        if len(names) < cfg.MODEL.ROI_HEADS.NUM_CLASSES:
            names = [str(i) for i in range(cfg.MODEL.ROI_HEADS.NUM_CLASSES)]
        return NewCOCOEvaluator(dataset_name, loader.dataset.dataset_dicts(cfg), names, output_dir=output_folder)

    @classmethod
    def test(cls, cfg, model, evaluators=None):
        """``DefaultTrainer.test``: one ``inference_on_dataset`` per ``DATASETS.TEST`` entry ->
        ``OrderedDict{dataset: results}``, flattened when there is a single dataset."""
        from collections import OrderedDict
        from ..evaluation import inference_on_dataset, print_csv_format
        if evaluators is not None and not isinstance(evaluators, (list, tuple)):
            evaluators = [evaluators]
        names = list(cfg.DATASETS.TEST) or ["synthetic_test"]
        if evaluators is not None:
            assert len(names) == len(evaluators), "{} != {}".format(len(names), len(evaluators))
        results = OrderedDict()
        for idx, dataset_name in enumerate(names):
            data_loader = cls.build_test_loader(cfg, dataset_name)
            evaluator = evaluators[idx] if evaluators is not None else \
                cls.build_evaluator(cfg, dataset_name, data_loader=data_loader)
            results_i = inference_on_dataset(model, data_loader, evaluator)
            results[dataset_name] = results_i
            if get_rank() == 0:
                assert isinstance(results_i, dict), \
                    "Evaluator must return a dict on the main process. Got {} instead.".format(results_i)
                print_csv_format(results_i)
        if len(results) == 1:
            results = list(results.values())[0]
        return results

    # ---- step ----------------------------------------------------------------------------------------
    def step_stream(self):
        """``with trainer.step_stream(): <loop of run_step()>`` -- see ``_on_step_stream``; ``train()`` and bench.py use it"""
        return _on_step_stream(self)

    def run_step(self):
        assert self.model.training, "[BaseTrainer] model was changed to eval mode!"
        _throttle(self)
        start = time.perf_counter()
        data = next(self._data_loader_iter)
        data_time = time.perf_counter() - start
        with stage("student_forward"):
            record_dict = self.model(data)
        loss_dict = {k: v for k, v in record_dict.items() if k[:4] == "loss" and k[-3:] != "val"}
        losses = sum(loss_dict.values())
        metrics_dict = dict(record_dict)
        metrics_dict["data_time"] = data_time
        self._write_metrics(metrics_dict)
        self.optimizer.zero_grad()
        with stage("student_backward"):
            try:
                losses.backward()
            finally:
                offchain.clear_loss_grads_mark()      # a mark no head consumed must not meet the next backward
        with stage("exchange"):
            self._reduce_gradients()
        with stage("update"):
            self.optimizer.step()
        _step_enqueued(self)

    def _reduce_gradients(self):
        """The one exchange step: sum the flat gradient buffer over ranks (RCCL); the 1/world averaging
        (DDP semantics) is folded into the fused SGD kernel.  With a GradientReducer attached the heads'
        slice (RPN + ROI heads, ~117 MB of the 190 MB) is already in flight: it was launched when the
        backbone backward started and overlaps it."""
        w = get_world_size()
        if w > 1:
            red = getattr(self, "_reducer", None)
            if red is not None:
                if red.skip and not self.__dict__.get("_skip_checked"):
                    # the skipped slots (a zero-weighted, elided domain classifier) are exchanged by nobody: that is only
                    # right while their gradients are exactly zero on this rank -- checked once, on the first step
                    self._skip_checked = True
                    g = self.optimizer.flat.grad
                    nz = sum(float(g[lo:min(hi, g.numel())].abs().max()) for lo, hi in red.skip if hi > lo)
                    assert nz == 0.0, "GradientReducer skips parameters with non-zero gradients (DC_img / DC_ins)"
                red.finish()
            else:
                dist.all_reduce(self.optimizer.flat.grad)
            self.optimizer.grad_scale = 1.0 / w

    def _write_metrics(self, metrics_dict, total=None):
        loss_keys = [k for k in metrics_dict if k[:4] == "loss"]
        if loss_keys:
            self.storage.put_scalar("total_loss", total if total is not None
                                    else sum(metrics_dict[k].detach() for k in loss_keys), rank_mean=True)
        self.storage.put_scalars(True, **{k: (v.detach() if isinstance(v, torch.Tensor) else v)
                                          for k, v in metrics_dict.items()})

    def after_step(self):
        """d2 hook order (source_free_adaptive_teacher.py:622-679 ``build_hooks``): LRScheduler, PeriodicCheckpointer,
        EvalHooks, ValLossHooks, and the PeriodicWriter LAST, so that what the evaluation of this iteration put into
        the storage is written with this iteration's number."""
        self.scheduler.step()
        nxt = self.iter + 1
        # d2 stamps a scalar with the iteration it was put in (TrainerBase.before_step: ``storage.iter = self.iter``; JSONWriter
        # writes that number): the last record of a 4-iteration run says ``"iteration": 3``
        self.storage.iter = self.iter
        # PeriodicCheckpointer: every CHECKPOINT_PERIOD iterations, and ``model_final`` after the last one (what
        # the reference's eval / AdaBN configs point MODEL.WEIGHTS at)
        p = self.cfg.SOLVER.CHECKPOINT_PERIOD
        if get_rank() == 0 and self.cfg.OUTPUT_DIR and p > 0:
            if nxt % p == 0:          # fvcore PeriodicCheckpointer: the numbered file on the last iteration too
                self.save_checkpoint("model_{:07d}".format(self.iter))
            if nxt >= self.max_iter:
                self.save_checkpoint("model_final")
        # d2 hooks.EvalHook(cfg.TEST.EVAL_PERIOD, ...): every EVAL_PERIOD iterations, and once after the last one
        ep = self.cfg.TEST.EVAL_PERIOD
        evaluated = False
        if self.cfg.SFOD.EVAL_HOOK and ((ep > 0 and nxt % ep == 0 and nxt != self.max_iter) or nxt >= self.max_iter):
            self._do_eval()
            evaluated = True
        # ValLossHook(cfg.TEST.EVAL_PERIOD, ...) when TEST.VAL_LOSS (val_loss.py:89-93: final iteration or period)
        if self.cfg.SFOD.EVAL_HOOK and self.cfg.TEST.VAL_LOSS and (nxt == self.max_iter or (ep > 0 and nxt % ep == 0)):
            self._do_val_loss()
            evaluated = True
        # PeriodicWriter(period 20) -- also whenever an evaluation ran (its scalars would otherwise be overwritten by
        # the next evaluation before a flush, or be stamped with a later iteration) and after the last iteration
        if nxt % WRITER_PERIOD == 0 or nxt >= self.max_iter or evaluated:
            self._flush_metrics()

    def _eval_targets(self):
        """[(attribute suffix, result-key suffix, model)]: what the trainer's EvalHooks evaluate (base.py:254-258)."""
        return [("", "", self.model)]

    def _val_loss_targets(self):
        """[(model_name, model)] of the ValLossHooks (base.py:259-265)."""
        return [("", self.model)]

    @torch.no_grad()
    def _do_val_loss(self):
        """``ValLossHook._do_loss_eval`` (daod/engine/hooks/val_loss.py:15-78): the model AS IT IS (training mode, so
        BatchNorm statistics move, as in the reference) on ``DATASETS.TEST[0]`` with ground truth, one image per
        batch, under no_grad; per-key mean over the batches -> ``<key><model_name>_val`` and
        ``total_loss<model_name>_val`` (sum of the ``loss*`` keys)."""
        from ..data import TestLoader
        for name, model in self._val_loss_targets():
            loader = TestLoader(self.cfg, self.device, get_rank(), get_world_size())
            loader.batch = 1
            acc, nb = {}, 0
            for inputs in loader:
                rec = model(inputs)
                if isinstance(rec, tuple):
                    rec = rec[0]
                if isinstance(rec, list):
                    rec = {}
                for k, v in rec.items():
                    v = v.detach().float() if isinstance(v, torch.Tensor) else torch.tensor(float(v), device=self.device)
                    acc[k] = acc[k] + v if k in acc else v
                nb += 1
            losses = {k: v / max(nb, 1) for k, v in acc.items() if k[:4] == "loss"}
            if losses:     # rank-local like the reference's (val_loss.py:64-66 writes on the main process only): the
                           # record of rank 0 -- its share of the test set -- is what reaches metrics.json
                self.storage.put_scalar("total_loss" + name + "_val", sum(losses.values()))
                if len(losses) > 1:
                    self.storage.put_scalars(**{k + name + "_val": v for k, v in losses.items()})

    def _do_eval(self):
        for attr, suffix, model in self._eval_targets():
            results = self.test(self.cfg, model)
            setattr(self, "_last_eval_results" + attr, results)
            if isinstance(results, dict):          # EvalHook._do_eval: flattened scalars into the storage
                flat = {}

                def walk(prefix, d):
                    for k, v in d.items():
                        if isinstance(v, dict):
                            walk(prefix + k + "/", v)
                        else:
                            flat[prefix + k] = float(v)
                walk("", {k + suffix: v for k, v in results.items()} if suffix else results)
                self.storage.put_scalars(smoothing_hint=False, **flat)       # d2 EvalHook: smoothing_hint=False

    def _flush_metrics(self):
        rpn = getattr(self.model, "proposal_generator", None)
        if rpn is not None:
            rpn.check_finite()
        if str(self.cfg.SFOD.COMPUTE_DTYPE).lower() == "f16x3":     # half pairs: clamped values are reported, not silent
            native.check_f16x3_range(torch.device(self.device))
        rec = self.storage.flush(reduce_over_ranks=True, smooth=True)
        rec["lr"] = self.optimizer.param_groups[0]["lr"]
        # d2 hooks.IterationTimer puts ``time`` (seconds per iteration) every step; a step here has no host synchronisation, so
        # the figure is the wall time of the writer period over its iterations (the flush above is the period's one host sync)
        now, it_now = time.perf_counter(), int(getattr(self, "iter", 0))
        last = self.__dict__.get("_writer_mark")
        if last is not None and it_now > last[1] and rec:
            rec["time"] = (now - last[0]) / (it_now - last[1])
        self._writer_mark = (now, it_now)
        if get_rank() == 0 and self.cfg.OUTPUT_DIR:
            os.makedirs(self.cfg.OUTPUT_DIR, exist_ok=True)
            with open(os.path.join(self.cfg.OUTPUT_DIR, "metrics.json"), "a") as f:
                f.write(json.dumps(rec) + "\n")
        return rec

    def train(self):
        with self.step_stream():
            for self.iter in range(self.start_iter, self.max_iter):
                self.run_step()
                self.after_step()
            if self.storage._pending:        # nothing a hook logged after the last flush is lost
                self._flush_metrics()

    def state_dict_for_checkpoint(self):
        return {"model": self.model.state_dict(), "iteration": self.iter, "optimizer": self.optimizer.state_dict(),
                "scheduler": self.scheduler.state_dict()}

    def save_checkpoint(self, name):
        """fvcore Checkpointer.save: ``<OUTPUT_DIR>/<name>.pth`` + the ``last_checkpoint`` pointer file."""
        from ..checkpoint import DetectionTSCheckpointer
        return DetectionTSCheckpointer(self, self.cfg.OUTPUT_DIR).save(name)

    def resume_or_load(self, resume=True):
        """DefaultTrainer.resume_or_load: continue from ``last_checkpoint`` in OUTPUT_DIR (model, optimizer,
        scheduler, iteration) when ``resume`` and it exists, else (re)load ``MODEL.WEIGHTS``."""
        from ..checkpoint import DetectionTSCheckpointer
        return DetectionTSCheckpointer(self, self.cfg.OUTPUT_DIR).resume_or_load(self.cfg.MODEL.WEIGHTS, resume=resume)


class _WeightedLossSum(torch.autograd.Function):
    """(w, l_0 .. l_{n-1}) -> (w * l as one [n] tensor, sum(w * l)); backward: d l_i = w_i * (g_sum + g_i)."""

    @staticmethod
    def forward(ctx, w, *losses):
        stacked = torch.stack([l.detach().reshape(()).float() for l in losses])
        weighted = stacked * w
        ctx.save_for_backward(w)
        ctx.n = len(losses)
        return weighted, weighted.sum()

    @staticmethod
    def backward(ctx, g_weighted, g_total):
        (w,) = ctx.saved_tensors
        g = w * g_total if g_weighted is None else w * (g_total + g_weighted)
        from ..modeling.offchain import mark_loss_grads_ready
        mark_loss_grads_ready(g)         # every loss gradient exists from here on (modeling/offchain.py)
        return (None,) + tuple(g[i] for i in range(ctx.n))


def weighted_loss_sum(w, losses):
    """``w``: device float32 [n]; ``losses``: n scalar tensors (autograd leaves of the step).  -> (weighted [n], total)."""
    return _WeightedLossSum.apply(w, *losses)


def loss_weight(key, cfg):
    """The weight ``run_step`` gives a loss key (source_free_adaptive_teacher.py:540-564, the same branch order): box
    regression and every other ``*_pseudo`` loss UNSUP_LOSS_WEIGHT, ``loss_bpc_pseudo`` 0, the domain-classifier losses
    DIS_LOSS_WEIGHT when their switch is on, everything else 0.  Pinned by running the reference's ``run_step`` on stubs
    (tests/golden/glue_ref.npz ``rs*_weights``: the gradients its ``losses.backward()`` leaves on the loss leaves)."""
    if key == "loss_rpn_loc_pseudo" or key == "loss_box_reg_pseudo":
        return float(cfg.SEMISUPNET.UNSUP_LOSS_WEIGHT)
    if key == "loss_bpc_pseudo":
        return 0.0
    if key[-6:] == "pseudo":
        return float(cfg.SEMISUPNET.UNSUP_LOSS_WEIGHT)
    if key in ("loss_DC_img_s", "loss_DC_img_t") and cfg.DOMAIN_CLASSIFIER.IMAGE:
        return float(cfg.SEMISUPNET.DIS_LOSS_WEIGHT)
    if key in ("loss_DC_ins_s", "loss_DC_ins_t") and cfg.DOMAIN_CLASSIFIER.INSTANCE:
        return float(cfg.SEMISUPNET.DIS_LOSS_WEIGHT)
    return 0.0


def threshold_bbox(proposal_bbox_inst, thres=0.7, proposal_type="roih"):
    """Instances-level API twin of source_free_adaptive_teacher.py:150-183 (strict '>')."""
    new = Instances(proposal_bbox_inst.image_size)
    if proposal_type == "rpn":
        valid = proposal_bbox_inst.objectness_logits > thres
        new.gt_boxes = Boxes(proposal_bbox_inst.proposal_boxes.tensor[valid, :])
        new.objectness_logits = proposal_bbox_inst.objectness_logits[valid]
    elif proposal_type == "roih":
        valid = proposal_bbox_inst.scores > thres
        new.gt_boxes = Boxes(proposal_bbox_inst.pred_boxes.tensor[valid, :])
        new.gt_classes = proposal_bbox_inst.pred_classes[valid]
        new.scores = proposal_bbox_inst.scores[valid]
    return new


class SourceFreeAdaptiveTeacherTrainer(BaseTrainer):
    def __init__(self, cfg, data_loader=None):
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        _apply_process_knobs(cfg)
        self.data_loader = data_loader or self.build_train_loader(cfg)
        self._data_loader_iter = iter(self.data_loader)
        # student, then teacher: both start from the same weights (:51-64).  With no checkpoint
        # to load, the teacher is initialised as a copy of the student.
        self.model = self.build_model(cfg)
        self.optimizer = self.build_optimizer(cfg, self.model)
        self._attach_reducer()
        self.model_teacher = self.build_model(cfg)
        self.teacher_flat = FlatModelState(self.model_teacher, frozen_prefixes=self._frozen(cfg), with_grad=False)
        # DetectionCheckpointer(model).resume_or_load(cfg.MODEL.WEIGHTS, resume=False) for the student and for the
        # teacher (:51-64); loading is in place, so the flat buffers see it.  Without weights: teacher <- student.
        if cfg.MODEL.WEIGHTS:
            from ..checkpoint import load_model_weights
            load_model_weights(self.model, cfg.MODEL.WEIGHTS, who="student")
            load_model_weights(self.model_teacher, cfg.MODEL.WEIGHTS, who="teacher")
            self._broadcast_initial_state()
            if get_world_size() > 1:         # the teacher's unmatched keys: rank 0's draw as well
                for t in (self.teacher_flat.param, self.teacher_flat.fbuf, self.teacher_flat.ibuf):
                    if t.numel():
                        dist.broadcast(t, 0)
                self.teacher_flat.values_rewritten()
        else:
            self._broadcast_initial_state()
            self._copy_main_model()
        self.ema_enabled = bool(cfg.SFOD.EMA.ENABLED)
        if self.ema_enabled:
            self.optimizer.attach_teacher(self.teacher_flat, cfg.SFOD.EMA.KEEP_RATE)
        for p in self.model_teacher.parameters():
            p.requires_grad_(False)
        self.scheduler = WarmupMultiStepLR(self.optimizer, cfg)
        self.start_iter, self.max_iter = 0, cfg.SOLVER.MAX_ITER
        self.iter = 0
        self.storage = EventStorage(0)
        self.model.train()
        self.model_teacher.train()  # quirk q2: the teacher is never put in eval mode (:385-390)
        self.elide = bool(cfg.SFOD.ELIDE_DEAD_BRANCHES)
        # Eliding the zero-weighted domain branch (:527-537) removes two of the student's three backbone passes per
        # step.  They see the same images as the first (WEAK_STRONG_AUGMENT False: q is a copy of k) with the same
        # weights, i.e. the same batch statistics, so their only effect -- two more momentum updates of every
        # BatchNorm's running statistics and num_batches_tracked += 2 -- is applied in closed form by the one pass that
        # runs.  With WEAK_STRONG_AUGMENT the two extra passes would see differently augmented frames: not
        # reproducible without running them (documented deviation; ELIDE_DEAD_BRANCHES False runs everything).
        dc_live = cfg.DOMAIN_CLASSIFIER.ENABLED and (cfg.DOMAIN_CLASSIFIER.IMAGE or cfg.DOMAIN_CLASSIFIER.INSTANCE)
        # Scoped to the student pass inside run_step: every other training-mode forward of the student (ValLossHook,
        # AdaBN passes, user code) is one backbone pass in the reference as well and must count as one.
        self._elided_bn_updates = 1
        if self.elide and cfg.DOMAIN_CLASSIFIER.ENABLED and not dc_live and not cfg.WEAK_STRONG_AUGMENT \
                and hasattr(self.model.backbone, "bn_updates_per_forward"):
            self._elided_bn_updates = 3

    @staticmethod
    def _frozen(cfg):
        return () if cfg.DOMAIN_CLASSIFIER.ENABLED else ("DC_img.", "DC_ins.")

    def build_train_loader(self, cfg):
        return TwoCropLoader(cfg, self.device, get_rank(), get_world_size(), labeled=False)

    @torch.no_grad()
    def _copy_main_model(self):
        """:605-617 -- teacher <- student (all parameters and buffers)."""
        s = self.optimizer.flat
        self.teacher_flat.param.copy_(s.param)
        self.teacher_flat.fbuf.copy_(s.fbuf)
        self.teacher_flat.ibuf.copy_(s.ibuf)
        self.teacher_flat.values_rewritten()

    def _val_loss_targets(self):
        """source_free_adaptive_teacher.py:663-675: the student (``_student``), then the teacher."""
        out = [("_student", self.model)]
        if self.model_teacher is not None and self.model_teacher is not self.model:
            out.append(("", self.model_teacher))
        return out

    def _eval_targets(self):
        """source_free_adaptive_teacher.py:648-662: two EvalHooks -- the student (result keys suffixed
        ``_student``), then the teacher."""
        out = [("_student", "_student", self.model)]
        if self.model_teacher is not None and self.model_teacher is not self.model:
            out.append(("_teacher", "", self.model_teacher))
        return out

    # ---- pseudo-labelling -------------------------------------------------------------------------
    def process_pseudo_label(self, proposals, cur_threshold, proposal_type, pseudo_label_method=""):
        """Instances-level API twin of :256-280 (host side; the step itself uses the fused kernel)."""
        if pseudo_label_method == "thresholding":
            out = [threshold_bbox(p, thres=cur_threshold, proposal_type=proposal_type) for p in proposals]
        elif pseudo_label_method in ("adaptive_thresholding", "prediction_thresholding"):
            # :185-254 -- both use the class-wise criterion (``cur_threshold`` is ignored by the reference);
            # the former returns gt_* fields, the latter keeps pred_*
            out = [self._adaptive_threshold_bbox(p, proposal_type, pseudo_label_method == "adaptive_thresholding")
                   for p in proposals]
        else:
            raise ValueError("Unkown pseudo label boxes methods")
        return out, sum(len(p) for p in out) / max(len(out), 1)

    def _adaptive_threshold_bbox(self, inst, proposal_type, as_gt):
        if proposal_type != "roih":
            return threshold_bbox(inst, thres=self.cfg.SEMISUPNET.BBOX_THRESHOLD, proposal_type=proposal_type)
        acc = self.__dict__.get("classwise_acc")
        if acc is None:
            acc = torch.ones(self.cfg.MODEL.ROI_HEADS.NUM_CLASSES, device=inst.scores.device)
        a = acc.to(inst.scores.device)[inst.pred_classes.long()]
        idx = torch.nonzero(inst.scores >= self.cfg.SEMISUPNET.BBOX_THRESHOLD * (a / (2. - a))).flatten()
        new = Instances(inst.image_size)
        if as_gt:
            new.gt_boxes = Boxes(inst.pred_boxes.tensor[idx, :])
            new.gt_classes = inst.pred_classes[idx]
        else:
            new.pred_boxes = Boxes(inst.pred_boxes.tensor[idx, :])
            new.pred_classes = inst.pred_classes[idx]
        new.scores = inst.scores[idx]
        return new

    @staticmethod
    def remove_label(label_data):
        for d in label_data:
            d.pop("instances", None)
        return label_data

    @staticmethod
    def add_label(unlabeled_data, label):
        for i, d in enumerate(unlabeled_data):
            d["instances"] = label.view(i) if isinstance(label, BatchedGT) else label[i]
        return unlabeled_data

    @contextlib.contextmanager
    def _student_pass_bn_updates(self):
        """The closed-form BatchNorm side effect of the elided passes (see __init__) applies to the student's
        backbone pass of ``run_step`` only; restored afterwards so that ValLossHook / AdaBN / user forwards count
        one momentum update per pass like the reference's."""
        bb = self.model.backbone
        k = self.__dict__.get("_elided_bn_updates", 1)
        if k == 1 or not hasattr(bb, "bn_updates_per_forward"):
            yield
            return
        prev = bb.bn_updates_per_forward
        bb.bn_updates_per_forward = k
        try:
            yield
        finally:
            bb.bn_updates_per_forward = prev

    def _teacher_pass(self, unlabel_data_k):
        """Steps 1-2 of run_step: train-mode teacher under no_grad, score threshold -> pseudo ground truth."""
        cfg = self.cfg
        with torch.no_grad():
            _, proposals_rpn_k, proposals_roih_k = self.model_teacher(unlabel_data_k, branch="unsup_data_weak",
                                                                      batched=True)
        d = proposals_roih_k.d
        # 2. thresholding (fused into the teacher post-processing kernel: score > BBOX_THRESHOLD)
        cur_threshold = cfg.SEMISUPNET.BBOX_THRESHOLD
        if "ADAPTIVE_THRESHOLD" in cfg and cfg.ADAPTIVE_THRESHOLD.ENABLED:
            # :393-404 ring of per-class counts -> class-wise accuracy; :461-466 after WARM_UP the pseudo labels
            # are score >= thr * acc/(2-acc) per class instead of the fixed threshold (one launch, no sync)
            K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
            if self.__dict__.get("reserve_matrix") is None:
                self.reserve_matrix = torch.zeros(cfg.ADAPTIVE_THRESHOLD.RESERVE, K, device=self.device)
                self.classwise_acc = torch.ones(K, device=self.device)
            native.adaptive_pseudo_labels_(d, cur_threshold, self.reserve_matrix,
                                           self.iter % cfg.ADAPTIVE_THRESHOLD.RESERVE, self.classwise_acc,
                                           select=self.iter >= cfg.ADAPTIVE_THRESHOLD.WARM_UP)
            acc = self.classwise_acc.clone()
            for i in range(min(8, K)):
                self.storage.put_scalar("acc_thres/class_" + str(i), acc[i])
        pseudo = proposals_roih_k.pseudo_gt()
        # logged scalars of the pass (:411-423, :445-452) in one launch: mean detection confidence (detection arrays are
        # not touched by the adaptive selection), RPN proposals above the threshold per image, mean pseudo-label count
        m = native.teacher_metrics(d["det_scores"], d["det_count"], proposals_rpn_k.logits, proposals_rpn_k.count,
                                   pseudo.count, cur_threshold)
        self.storage.put_scalar("roi_head/mean_confidence", m[0])
        self.storage.put_scalar("rpn/num_pseudo_proposals", m[1])
        self.storage.put_scalar("roi_head/num_pseudo_proposals", m[2])
        return pseudo

    # ---- step (:335-581) ----------------------------------------------------------------------------
    def run_step(self):
        cfg = self.cfg
        assert self.model.training, "[AdaptiveTeacherTrainer] model was changed to eval mode!"
        _throttle(self)
        if hasattr(self.model, "drop_prefetched"):
            self.model.drop_prefetched()
        start = time.perf_counter()
        unlabel_data_q, unlabel_data_k = next(self._data_loader_iter)
        if not cfg.WEAK_STRONG_AUGMENT:
            unlabel_data_q = [dict(d) for d in unlabel_data_k]
        data_time = time.perf_counter() - start
        record_dict = {}
        # 0. remove the labels of the target data (source-free)
        unlabel_data_q = self.remove_label(unlabel_data_q)
        unlabel_data_k = self.remove_label(unlabel_data_k)
        # 1. pseudo-labels from the (train-mode) teacher.  The student's backbone forward does not depend on
        # them (only its heads' losses do), so with SFOD.OVERLAP_TEACHER the teacher pass runs on a second
        # stream beside it: its low-occupancy tail (sort, NMS, ROIAlign, small GEMMs) fills the conv kernels' gaps.
        ov = self.__dict__.get("overlap_teacher")      # None -> config; bench.py's roofline segment forces False
        overlap = ((cfg.SFOD.OVERLAP_TEACHER if ov is None else ov) and self.model_teacher is not self.model
                   and hasattr(self.model, "prefetch_features"))
        if overlap:
            main = torch.cuda.current_stream()
            side = self.__dict__.get("_side_stream")
            if side is None:
                side = self._side_stream = torch.cuda.Stream(device=self.device, priority=-1)
            side.wait_stream(main)       # EMA'd teacher weights, input frames, last step's readers of side buffers
            with torch.cuda.stream(side), stage("teacher"):
                pseudo = self._teacher_pass(unlabel_data_k)
            with self._student_pass_bn_updates(), stage("student_forward"):
                self.model.prefetch_features(unlabel_data_q)
            main.wait_stream(side)
        else:
            with stage("teacher"):
                pseudo = self._teacher_pass(unlabel_data_k)
        # 3. attach the pseudo-labels
        unlabel_data_q = self.add_label(unlabel_data_q, pseudo)
        unlabel_data_k = self.add_label(unlabel_data_k, pseudo)
        # 5. student on the pseudo-labelled target data
        with self._student_pass_bn_updates(), stage("student_forward"):
            record_all_unlabel_data, _, _, _ = self.model(unlabel_data_q, branch="supervised_target", batched=True)
        for key, v in record_all_unlabel_data.items():
            record_dict[key + "_pseudo"] = v
        # 6. domain-classifier branch (:527-537): zero-weighted unless DOMAIN_CLASSIFIER.IMAGE/INSTANCE
        dc_live = cfg.DOMAIN_CLASSIFIER.ENABLED and (cfg.DOMAIN_CLASSIFIER.IMAGE or cfg.DOMAIN_CLASSIFIER.INSTANCE)
        if cfg.DOMAIN_CLASSIFIER.ENABLED and (dc_live or not self.elide):
            for i in range(len(unlabel_data_q)):
                for k, v in unlabel_data_q[i].items():
                    unlabel_data_k[i][k + "_unlabeled"] = v
            record_all_domain_data, _, _ = self.model(unlabel_data_k, branch="domain_classifier")
            record_dict.update(record_all_domain_data)
        # loss weighting (:540-564): the same weights key by key; the products and their sum are ONE autograd node over
        # the stacked scalars (3 small kernels forward, 1 backward) instead of a multiply and an add per key and
        # direction -- ~30 single-element launches per step otherwise
        keys = [key for key in record_dict.keys() if key.startswith("loss") and key[-3:] != "val"]
        weights = [loss_weight(key, cfg) for key in keys]
        wdev = native.dev_const(tuple(weights), torch.float32, self.device)
        weighted, losses = weighted_loss_sum(wdev, [record_dict[k] for k in keys])
        loss_dict = {k: weighted[i] for i, k in enumerate(keys)}
        self.storage.put_scalar("calibration/bpc_loss", loss_dict["loss_bpc_pseudo"])
        metrics_dict = dict(loss_dict)
        metrics_dict["data_time"] = data_time
        self._write_metrics(metrics_dict, total=losses.detach())
        self.optimizer.zero_grad()
        with stage("student_backward"):
            try:
                losses.backward()
            finally:
                offchain.clear_loss_grads_mark()      # a mark no head consumed must not meet the next backward
        with stage("exchange"):
            self._reduce_gradients()
        with stage("update"):
            self.optimizer.step(ema=self.ema_enabled)  # EMA fused: _update_teacher_model (:583-603)
        _step_enqueued(self)

    def _flush_metrics(self):
        self.model_teacher.proposal_generator.check_finite()
        return super()._flush_metrics()

    def state_dict_for_checkpoint(self):
        """detection_ts_checkpointer.py / ts_ensemble.py: one dict, modelTeacher.* + modelStudent.*"""
        sd = {}
        for k, v in self.model_teacher.state_dict().items():
            sd["modelTeacher." + k] = v
        for k, v in self.model.state_dict().items():
            sd["modelStudent." + k] = v
        return {"model": sd, "iteration": self.iter, "optimizer": self.optimizer.state_dict(),
                "scheduler": self.scheduler.state_dict()}


class SourceFreeAdaptiveTeacherSingleTrainer(SourceFreeAdaptiveTeacherTrainer):
    """``source_free_adaptive_teacher_single.py``: the student labels itself (:390), EMA on (:581)."""

    def run_step(self):
        teacher = self.model_teacher
        self.model_teacher = self.model
        try:
            super().run_step()
        finally:
            self.model_teacher = teacher


class AdaptiveTeacherTrainer(SourceFreeAdaptiveTeacherTrainer):
    """``TRAINER: "adaptive_teacher"`` (daod/engine/trainers/adaptive_teacher.py:31-420): the with-source teacher-student
    loop the source-free trainer was derived from.  Differences that matter here:

      * four lists per step -- labelled source frames (strong, weak) and unlabelled target frames (strong, weak), :195-199;
      * burn-in (:202-213): until ``SEMISUPNET.BURN_UP_STEP`` only ``branch="supervised"`` on the 2 x IMS_PER_BATCH
        labelled crops, every loss x 1;
      * the teacher is updated at the START of a step on its own schedule (:215-223): a copy of the student
        (``keep_rate = 0``) at ``iter == BURN_UP_STEP``, afterwards the EMA every ``TEACHER_UPDATE_ITER`` steps -- not
        fused into the optimizer step (the update of step i therefore sees the student as step i - 1 left it, and a
        checkpoint / evaluation between two steps sees the teacher the reference's hooks see);
      * after burn-in (:225-329): teacher on the weak target crops -> pseudo labels -> student on ALL labelled crops
        (``supervised``), on the strong target crops (``supervised_target``) and on (weak source, weak target) pairs
        (``domain_classifier``, always: no DOMAIN_CLASSIFIER switch on this path); the domain pass's ``loss_DC_img_s``
        REPLACES the ``supervised`` branch's (same dict key, :314); weights :316-327 -- pseudo box regression 0, other
        ``*_pseudo`` UNSUP_LOSS_WEIGHT, ``loss_DC_img_{s,t}`` DIS_LOSS_WEIGHT, everything else (incl. the instance-level
        discriminator) 1; the UNWEIGHTED record goes to ``_write_metrics`` (:331-333).
    Pinned by running the reference's ``run_step`` on recorder stubs (tests/golden/adaptive_teacher_ref.npz)."""

    def __init__(self, cfg, data_loader=None):
        super().__init__(cfg, data_loader)
        self.ema_enabled = False                 # never fused: see above
        self.elide = False
        self._elided_bn_updates = 1              # every pass of the reference runs
        self.burn_up_step = int(cfg.SEMISUPNET.BURN_UP_STEP)
        self.teacher_update_iter = int(cfg.SEMISUPNET.TEACHER_UPDATE_ITER)
        self.ema_keep_rate = float(cfg.SEMISUPNET.EMA_KEEP_RATE)

    @staticmethod
    def _frozen(cfg):
        return ()                                # the discriminators train on this path

    @classmethod
    def build_optimizer(cls, cfg, model):
        return build_optimizer(cfg, model, frozen=cls._frozen(cfg))

    def _attach_reducer(self):
        self._reducer = None                     # several backbone passes per backward: one blocking exchange afterwards

    def build_train_loader(self, cfg):
        from ..data.synthetic import FourWayLoader
        return FourWayLoader(cfg, self.device, get_rank(), get_world_size())

    @torch.no_grad()
    def _update_teacher_model(self, keep_rate=0.9996):
        """:338-357 -- teacher <- student * (1 - keep_rate) + teacher * keep_rate over parameters AND buffers (the int64
        counters in float32, truncated: SURVEY A.17 iv); keep_rate 0 is the burn-in hand-over."""
        s, t = self.optimizer.flat, self.teacher_flat
        native.ema_(t.param, s.param, keep_rate)
        native.ema_(t.fbuf, s.fbuf, keep_rate)
        native.ema_i64_(t.ibuf, s.ibuf, keep_rate)
        t.values_rewritten()

    @staticmethod
    def loss_weight(key, cfg):
        """:316-327, the same branch order"""
        if key == "loss_rpn_loc_pseudo" or key == "loss_box_reg_pseudo":
            return 0.0
        if key[-6:] == "pseudo":
            return float(cfg.SEMISUPNET.UNSUP_LOSS_WEIGHT)
        if key == "loss_DC_img_s" or key == "loss_DC_img_t":
            return float(cfg.SEMISUPNET.DIS_LOSS_WEIGHT)
        return 1.0

    def run_step(self):
        cfg = self.cfg
        assert self.model.training, "[AdaptiveTeacherTrainer] model was changed to eval mode!"
        _throttle(self)
        if hasattr(self.model, "drop_prefetched"):
            self.model.drop_prefetched()
        start = time.perf_counter()
        label_data_q, label_data_k, unlabel_data_q, unlabel_data_k = next(self._data_loader_iter)
        data_time = time.perf_counter() - start
        if self.iter < self.burn_up_step:
            label_data_q = list(label_data_q)
            label_data_q.extend(label_data_k)
            with stage("student_forward"):
                record_dict, _, _ = self.model(label_data_q, branch="supervised")
            keys = [k for k in record_dict if k[:4] == "loss" and k[-3:] != "val"]
            weights = [1.0] * len(keys)
        else:
            if self.iter == self.burn_up_step:
                self._update_teacher_model(keep_rate=0.00)
            elif (self.iter - self.burn_up_step) % self.teacher_update_iter == 0:
                self._update_teacher_model(keep_rate=self.ema_keep_rate)
            record_dict = {}
            unlabel_data_q = self.remove_label(unlabel_data_q)
            unlabel_data_k = self.remove_label(unlabel_data_k)
            with stage("teacher"):
                pseudo = self._teacher_pass(unlabel_data_k)
            self.storage._pending.pop("roi_head/mean_confidence", None)      # (the source-free trainer's scalar, not logged here)
            unlabel_data_q = self.add_label(unlabel_data_q, pseudo)
            unlabel_data_k = self.add_label(unlabel_data_k, pseudo)
            all_label_data = list(label_data_q) + list(label_data_k)
            with stage("student_forward"):
                record_all_label_data, _, _ = self.model(all_label_data, branch="supervised")
                record_dict.update(record_all_label_data)
                record_all_unlabel_data, _, _ = self.model(unlabel_data_q, branch="supervised_target", batched=True)
                for key, v in record_all_unlabel_data.items():
                    record_dict[key + "_pseudo"] = v
                for i in range(len(unlabel_data_k)):
                    for k, v in unlabel_data_k[i].items():
                        label_data_k[i][k + "_unlabeled"] = v
                record_all_domain_data, _, _ = self.model(label_data_k, branch="domain_classifier")
                record_dict.update(record_all_domain_data)       # its loss_DC_img_s replaces the supervised branch's
            keys = [k for k in record_dict if k.startswith("loss") and k[-3:] != "val"]
            weights = [self.loss_weight(k, cfg) for k in keys]
        wdev = native.dev_const(tuple(weights), torch.float32, self.device)
        _, losses = weighted_loss_sum(wdev, [record_dict[k] for k in keys])
        metrics_dict = dict(record_dict)
        metrics_dict["data_time"] = data_time
        self._write_metrics(metrics_dict)
        self.optimizer.zero_grad()
        with stage("student_backward"):
            try:
                losses.backward()
            finally:
                offchain.clear_loss_grads_mark()
        with stage("exchange"):
            self._reduce_gradients()
        with stage("update"):
            self.optimizer.step(ema=False)
        _step_enqueued(self)


# ---- AdaBN refinement (base.py:270-337) -----------------------------------------------------------
def reset_bn_stats(model):
    """reset every BatchNorm2d running_mean / running_var to 0 / 1 (:318-328)."""
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.zero_()
            m.running_var.fill_(1.0)


@torch.no_grad()
def adabn_refinement(cfg, model, data_loader, max_iters=1400):
    """forward-only passes in train mode under no_grad: BN momentum-0.1 running-stat refresh."""
    reset_bn_stats(model)
    model.train()
    it = iter(data_loader)
    for i in range(max_iters + 1):
        try:
            data = next(it)
        except StopIteration:
            break
        if isinstance(data, tuple):
            data = data[1]
        images = model.preprocess_image(data)
        model._features(images)
    return model


def test_refinement(cfg, model, data_loader, max_iters=1400, trainer_cls=None):
    """``base.py:270-315`` as called from ``adabn_refinement`` (:330-337): the statistics passes, then
    ``trainer.test(cfg, model)`` and a ``adabn.pth`` checkpoint in ``OUTPUT_DIR``.  -> test results."""
    adabn_refinement(cfg, model, data_loader, max_iters)
    results = (trainer_cls or BaseTrainer).test(cfg, model)
    if get_rank() == 0 and cfg.OUTPUT_DIR:
        os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
        torch.save({"model": model.state_dict()}, os.path.join(cfg.OUTPUT_DIR, "adabn.pth"))
    return results


TRAINERS = {
    "base": BaseTrainer,
    "source_free_adaptive_teacher": SourceFreeAdaptiveTeacherTrainer,
    "source_free_adaptive_teacher_single": SourceFreeAdaptiveTeacherSingleTrainer,
    "adaptive_teacher": AdaptiveTeacherTrainer,
}


def get_trainer_class(cfg):
    """train_net_mt.py:48-69."""
    if cfg.TRAINER not in TRAINERS:
        raise ValueError(f"Trainer {cfg.TRAINER} not found.")
    return TRAINERS[cfg.TRAINER]
