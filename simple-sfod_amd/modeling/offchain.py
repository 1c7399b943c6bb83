"""Parameter-gradient launches of a head's backward beside its data-gradient path (roi_heads.py, rpn.py)."""
import os

import torch

_HEAD_WGRAD_STREAM = os.environ.get("SFOD_HEAD_WGRAD_STREAM", "1") != "0"      # 0: everything on one stream (A/B hook)


class OffChain:
    """Parameter-gradient launches of a backward beside its data-gradient path: ``run(fn, *reads)`` executes ``fn`` on the
    module's side stream once everything enqueued on the current stream so far is done (the tensors in ``reads`` live in the
    current stream's pool: the allocator is told the side stream uses them); ``join(*outs)`` makes the current stream wait
    and hands the results over.  Disabled (SFOD_HEAD_WGRAD_STREAM=0, CPU tensors): plain calls."""

    def __init__(self, owner, on):
        self.side = None
        if on and _HEAD_WGRAD_STREAM:
            self.side = owner.__dict__.get("_wgrad_stream")
            if self.side is None:
                self.side = owner.__dict__["_wgrad_stream"] = torch.cuda.Stream()

    def run(self, fn, *reads):
        if self.side is None:
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
        main.wait_stream(self.side)
        for t_ in outs:
            if t_ is not None:
                t_.record_stream(main)
