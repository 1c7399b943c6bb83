"""Parameter-gradient launches of a head's backward beside its data-gradient path (roi_heads.py, rpn.py)."""
import os

import torch

_SHARED = os.environ.get("SFOD_SHARED_SIDE_STREAM", "1") != "0"
_HEAD_WGRAD_STREAM = os.environ.get("SFOD_HEAD_WGRAD_STREAM", "1") != "0"      # 0: everything on one stream (A/B hook)


_shared = {}


def shared_stream():
    """ONE side stream per device for everything that runs beside a data-gradient path (the backbones' weight gradients, the
    heads' parameter gradients, the RPN's backward): HIP multiplexes streams onto 4 hardware queues, and a stream per module
    (six streams in a step) cost the 1024 x 2048 configuration 4.5 % (profiles/round5/r5_prefetch_label_free_work.txt)."""
    dev = torch.cuda.current_device()
    st = _shared.get(dev)
    if st is None:
        st = _shared[dev] = torch.cuda.Stream()
    return st


class OffChain:
    """Parameter-gradient launches of a backward beside its data-gradient path: ``run(fn, *reads)`` executes ``fn`` on the
    module's side stream once everything enqueued on the current stream so far is done (the tensors in ``reads`` live in the
    current stream's pool: the allocator is told the side stream uses them); ``join(*outs)`` makes the current stream wait
    and hands the results over.  Disabled (SFOD_HEAD_WGRAD_STREAM=0, CPU tensors): plain calls."""

    def __init__(self, owner, on):
        self.side = None
        if on and _HEAD_WGRAD_STREAM:
            self.side = shared_stream() if _SHARED else owner.__dict__.get("_wgrad_stream")
            if self.side is None:
                self.side = owner.__dict__["_wgrad_stream"] = torch.cuda.Stream()

    def run(self, fn, *reads):
        # already ON the side stream (the RPN's whole backward runs there, and its own parameter gradients come through here):
        # plain calls -- an event recorded on a stream and awaited by the same stream is a no-op eagerly (and a self-dependency
        # inside a graph capture: what made hipStreamEndCapture crash in round 6's graph experiment)
        if self.side is None or torch.cuda.current_stream() == self.side:
            return fn()
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            out = fn()
        for t_ in reads:
            t_.record_stream(self.side)
        return out

    def join(self, *outs):
        if self.side is None:
            return
        main = torch.cuda.current_stream()
        if main == self.side:
            return
        main.wait_stream(self.side)
        for t_ in outs:
            if t_ is not None:
                t_.record_stream(main)


# ---- a whole head's backward beside the other head's --------------------------------------------------------------------
# The RPN's and the ROI heads' backward passes are independent until their feature-map gradients are added; autograd runs
# them one after the other on one stream.  The trainer's weighted-loss node marks the point where every loss gradient
# exists (``mark_loss_grads_ready``); a head that finds the mark runs its whole backward on its side stream from THERE --
# not from wherever the main stream has got to -- and the main stream waits for it at the end.
_loss_grads_ready = None


def mark_loss_grads_ready(g):
    """``g``: the tensor the loss gradients are slices of.  The mark keeps ``g`` itself alive (so its storage cannot be
    handed to another tensor while the mark exists) and is valid for ONE backward: ``clear_loss_grads_mark`` (the trainers
    call it when ``backward()`` returns) or the first ``take`` drops it."""
    global _loss_grads_ready
    if _HEAD_WGRAD_STREAM and g.is_cuda:
        ev = torch.cuda.Event()
        ev.record()
        _loss_grads_ready = (ev, g, g.device)


def take_loss_grads_ready(grad):
    """the mark, if ``grad`` is a view of the very tensor it was made for (a mark of another backward or another device is
    dropped): the marked tensor is held by reference, so equal storage addresses mean the same live storage"""
    global _loss_grads_ready
    m, _loss_grads_ready = _loss_grads_ready, None
    if m is None or not grad.is_cuda or grad.device != m[2]:
        return None
    if grad.untyped_storage().data_ptr() != m[1].untyped_storage().data_ptr():
        return None
    return m[0]


def clear_loss_grads_mark():
    """after ``backward()``: a mark nobody consumed (a step without an RPN loss) must not survive into the next backward"""
    global _loss_grads_ready
    _loss_grads_ready = None
