"""Region proposal network on the HIP kernels behind the names ``RPN`` / ``PseudoLabRPN``.

Mirrors ``/root/reference/daod/modeling/proposal_generator/rpn.py:10-58`` (PseudoLabRPN: losses
only when ``(training and compute_loss) or compute_val_loss``, proposals always, ``loss_weight``
applied a second time at ``:49`` -- quirk q3) on top of the Detectron2 RPN it subclasses
(SURVEY.md Appendix A.3-A.10): DefaultAnchorGenerator, StandardRPNHead, Matcher [0.3, 0.7] with
low-quality matches, subsample 256 @ 0.5, BCE + L1 losses, find_top_rpn_proposals.

State-dict keys: ``proposal_generator.rpn_head.{conv,objectness_logits,anchor_deltas}.{weight,bias}``
and ``proposal_generator.anchor_generator.cell_anchors.0``.
"""
import math

import torch
import torch.nn as nn

from .. import native
from ..registry import PROPOSAL_GENERATOR_REGISTRY
from .batched import BatchedGT, BatchedProposals
from .offchain import OffChain, take_loss_grads_ready


class BufferList(nn.Module):
    def __init__(self, buffers):
        super().__init__()
        for i, b in enumerate(buffers):
            self.register_buffer(str(i), b, persistent=False)

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


class DefaultAnchorGenerator(nn.Module):
    """d2 DefaultAnchorGenerator, offset 0.0 (A.3).  Only the cell anchors are materialised: the
    grid shift of an anchor index is recomputed in-register by the kernels."""
    box_dim = 4

    def __init__(self, cfg, input_shape):
        super().__init__()
        sizes = cfg.MODEL.ANCHOR_GENERATOR.SIZES
        ratios = cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS
        assert len(input_shape) == 1, "single-level RPN only (C4-style heads)"
        self.strides = [s.stride for s in input_shape]
        self.cell_anchors = BufferList([self.generate_cell_anchors(sizes[0], ratios[0])])

    @staticmethod
    def generate_cell_anchors(sizes, aspect_ratios):
        anchors = []
        for size in sizes:
            area = size ** 2.0
            for ar in aspect_ratios:
                w = math.sqrt(area / ar)
                h = ar * w
                anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(anchors, dtype=torch.float32)

    @property
    def num_anchors(self):
        return [len(c) for c in self.cell_anchors]

    def forward(self, features):
        """Materialised anchors [(Hf*Wf*A, 4)] in (y, x, a) order -- API parity / tests only."""
        out = []
        for f, stride, cell in zip(features, self.strides, self.cell_anchors):
            hf, wf = f.shape[-2:]
            sx = torch.arange(0, wf * stride, step=stride, dtype=torch.float32, device=cell.device)
            sy = torch.arange(0, hf * stride, step=stride, dtype=torch.float32, device=cell.device)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
            out.append((shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4))
        return out


class StandardRPNHead(nn.Module):
    """conv3x3 + ReLU -> objectness 1x1 (A) and anchor deltas 1x1 (4A); init normal(0.01) (A.4)."""

    def __init__(self, in_channels, num_anchors, box_dim=4):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.objectness_logits = nn.Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.anchor_deltas = nn.Conv2d(in_channels, num_anchors * box_dim, kernel_size=1, stride=1)
        for layer in [self.conv, self.objectness_logits, self.anchor_deltas]:
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)

    def params(self):
        return [self.conv.weight, self.conv.bias, self.objectness_logits.weight, self.objectness_logits.bias,
                self.anchor_deltas.weight, self.anchor_deltas.bias]


class _RPNLossFn(torch.autograd.Function):
    """features -> (loss_rpn_cls, loss_rpn_loc); backward: loss grads -> 1x1 heads -> 3x3 conv."""

    @staticmethod
    def forward(ctx, rpn, feat_nchw, gt, keys, weights, *params):
        """``weights``: the two loss weights (python floats) -- applied here, as fp32 scalar multiplications like the
        ``loss * weight`` they replace, so that the node's incoming gradients are the trainer's loss gradients themselves"""
        pf = rpn._take_prefetched(feat_nchw)        # the head's forward may have run already (``prefetch``): same state
        st = pf[0] if pf is not None else rpn._head_forward(feat_nchw, need_grad=True)
        rpn._prefetched_proposals = pf[1] if pf is not None else None
        loss, lab_state = rpn._loss_forward(st, gt, keys)
        ctx.rpn, ctx.st, ctx.lab_state, ctx.gt, ctx.weights = rpn, st, lab_state, gt, weights
        rpn._last_head_state = st
        return loss[0] * weights[0], loss[1] * weights[1]

    @staticmethod
    def backward(ctx, g_cls, g_loc):
        rpn, st = ctx.rpn, ctx.st

        def run():
            gs = torch.stack([(g_cls * ctx.weights[0]).reshape(()), (g_loc * ctx.weights[1]).reshape(())]).float().contiguous()
            return rpn._loss_backward(st, ctx.lab_state, ctx.gt, gs)

        # beside the ROI heads' backward when the trainer marked where the loss gradients became available (offchain.py)
        ev = take_loss_grads_ready(g_cls)
        side = OffChain(rpn, g_cls.is_cuda).side if ev is not None else None
        if side is not None:
            main = torch.cuda.current_stream()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dfeat, pgrads = run()
            for t_ in (g_cls, g_loc):
                t_.record_stream(side)
            main.wait_stream(side)
            for t_ in [dfeat] + list(pgrads):
                if t_ is not None:
                    t_.record_stream(main)
        else:
            dfeat, pgrads = run()
        ctx.st = ctx.lab_state = None
        return (None, dfeat, None, None, None) + tuple(pgrads)


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPN(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.in_features = cfg.MODEL.RPN.IN_FEATURES
        shapes = [input_shape[f] for f in self.in_features]
        assert len(shapes) == 1, "single feature level"
        self.anchor_generator = DefaultAnchorGenerator(cfg, shapes)
        A = self.anchor_generator.num_anchors[0]
        self.rpn_head = StandardRPNHead(shapes[0].channels, A)
        self.stride = shapes[0].stride
        self.num_anchors = A
        self.channels = shapes[0].channels
        r = cfg.MODEL.RPN
        self.batch_size_per_image = r.BATCH_SIZE_PER_IMAGE
        self.positive_fraction = r.POSITIVE_FRACTION
        self.iou_thresholds = list(r.IOU_THRESHOLDS)
        assert list(r.IOU_LABELS) == [0, -1, 1] and r.BBOX_REG_LOSS_TYPE == "smooth_l1" and r.SMOOTH_L1_BETA == 0.0
        assert tuple(r.BBOX_REG_WEIGHTS) == (1.0, 1.0, 1.0, 1.0)
        self.pre_nms_topk = {True: r.PRE_NMS_TOPK_TRAIN, False: r.PRE_NMS_TOPK_TEST}
        self.post_nms_topk = {True: r.POST_NMS_TOPK_TRAIN, False: r.POST_NMS_TOPK_TEST}
        self.nms_thresh = r.NMS_THRESH
        self.loss_weight = {"loss_rpn_cls": r.LOSS_WEIGHT, "loss_rpn_loc": r.BBOX_REG_LOSS_WEIGHT * r.LOSS_WEIGHT}
        self.compute_dtype = native.mode_dtype(cfg.SFOD.COMPUTE_DTYPE)
        self.ld = (5 * A + 7) // 8 * 8  # fused head output row stride (fp32), whole 16-byte chunks
        self._flags = None
        self._last_head_state = None

    # ---- the label-free part of a training pass, ahead of time -----------------------------------------
    @torch.no_grad()
    def prefetch(self, images, features):
        """Run what a training-mode ``forward`` does WITHOUT labels -- the head's convolutions and the proposals (decode,
        sort, top-k, NMS) -- now; the next ``forward`` on the same feature tensor picks both up.  The trainer calls it (through
        ``model.prefetch_features``) while the teacher is still producing the pseudo labels: only the anchor matching and
        the losses wait for them.  Same kernels on the same inputs: the results are the ones ``forward`` would compute."""
        feat = features[self.in_features[0]]
        st = self._head_forward(feat, need_grad=True)
        # ... and the weights in the forms the backward multiplies with (no weight changes before this step's update)
        gdt = native.dt_of_dtype(native.grad_dtype_of(self.compute_dtype))
        st["w1t"] = native.pack_fc_weight(st["w1"], gdt, transpose=True, ld=self.ld)
        st["wr"] = native.pack_conv_weight(self.rpn_head.conv.weight.detach(), self.channels, gdt, rot180=True)
        self._prefetched = (feat, st, self._proposals(st, images.image_sizes))

    def _take_prefetched(self, feat):
        pf = self.__dict__.pop("_prefetched", None)
        return (pf[1], pf[2]) if pf is not None and pf[0] is feat else None

    # ---- head ----------------------------------------------------------------------------------------
    def _cell(self):
        return self.anchor_generator.cell_anchors._buffers["0"]

    def _head_forward(self, feat_nchw, need_grad=False):
        """feat (NCHW view of NHWC memory) -> dict(feat, t, rpn_out).  ``need_grad``: a backward follows -- the input is also
        kept as the operand the 3x3 convolution's weight gradient reads (f16x3: bf16 pairs, from the same pass)"""
        dtype = self.compute_dtype
        dt = native.dt_of_dtype(dtype)
        nhwc = feat_nchw.permute(0, 2, 3, 1)
        if nhwc.dtype == torch.float32 and native.is_pairs(dtype):
            feat, feat_g = native.operands_for(nhwc.contiguous(), dtype, need_grad)
        else:
            feat = native.nhwc_operand(feat_nchw, dtype)
            feat_g = feat if need_grad else None
        h = self.rpn_head
        C, A = self.channels, self.num_anchors
        wp = native.pack_conv_weight(h.conv.weight.detach(), C, dt)
        t = native.conv_fwd(feat, wp, h.conv.bias.detach(), C, 3, act=1)
        w1 = torch.cat([h.objectness_logits.weight.detach().view(A, C), h.anchor_deltas.weight.detach().view(4 * A, C)])
        b1 = torch.cat([h.objectness_logits.bias.detach(), h.anchor_deltas.bias.detach()])
        w1p = native.pack_fc_weight(w1, dt)
        B, Hf, Wf, _ = feat.shape
        rpn_out = native.conv_fwd(t.view(B * Hf * Wf, C), w1p, b1, 5 * A, 1, out_dtype=torch.float32, ldy=self.ld)
        return {"feat": feat, "feat_g": feat_g, "t": t, "rpn_out": rpn_out, "w1": w1, "shape": (B, Hf, Wf)}

    def _sizes_dev(self, image_sizes, device):
        return native.dev_const(tuple((int(s[0]), int(s[1])) for s in image_sizes), torch.int32, device)

    def _proposals(self, st, image_sizes, sizes_dev=None):
        B, Hf, Wf = st["shape"]
        dev = st["rpn_out"].device
        if self._flags is None or self._flags.device != dev:
            self._flags = torch.zeros(1, dtype=torch.int32, device=dev)
        if sizes_dev is None:
            sizes_dev = self._sizes_dev(image_sizes, dev)
        props, scores = native.rpn_decode(st["rpn_out"], self._cell(), B, Hf, Wf, self.stride, sizes_dev, self._flags)
        ss, si = native.segmented_sort_desc(scores)
        NA = scores.shape[1]
        k = min(self.pre_nms_topk[self.training], NA)
        cb, cs, cv = native.rpn_gather_topk(props, ss, si, k)
        post = self.post_nms_topk[self.training]
        keep_idx, keep_cnt = native.nms(cb, self.nms_thresh, post, valid=cv)
        pb, ps = native.gather_kept(cb, cs, keep_idx, keep_cnt)
        return BatchedProposals(pb, ps, keep_cnt, list(image_sizes))

    def check_finite(self):
        """d2 raises FloatingPointError inside predict_proposals; here the flag is raised on the
        device and checked by the trainer at its logging period (one host sync per period)."""
        if self._flags is not None and self._flags.item() != 0:
            raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")

    # ---- losses --------------------------------------------------------------------------------------
    def _loss_forward(self, st, gt, keys):
        B, Hf, Wf = st["shape"]
        cell = self._cell()
        lo, hi = self.iou_thresholds
        matched, labels = native.anchor_match(cell, B, Hf, Wf, self.stride, gt.boxes, gt.count, lo, hi)
        native.subsample_rpn_(labels, keys, self.batch_size_per_image, self.positive_fraction)
        loss, _ = native.rpn_loss(st["rpn_out"], cell, B, Hf, Wf, self.stride, labels, matched, gt.boxes,
                                  gt.count, self.batch_size_per_image)
        return loss, (labels, matched)

    def _loss_backward(self, st, lab_state, gt, grad_scale):
        labels, matched = lab_state
        B, Hf, Wf = st["shape"]
        cell = self._cell()
        C, A = self.channels, self.num_anchors
        dtype = native.grad_dtype_of(self.compute_dtype)      # operands of the backward products ("f16x3": bf16 pairs)
        dt = native.dt_of_dtype(dtype)
        _, d_out = native.rpn_loss(st["rpn_out"], cell, B, Hf, Wf, self.stride, labels, matched, gt.boxes,
                                   gt.count, self.batch_size_per_image, grad_scale=grad_scale)
        M = B * Hf * Wf
        t2 = st["t"].view(M, C)
        d_out_c = native.cast(d_out, dtype)
        # 1x1 heads: weight / bias gradients (beside the data-gradient path: ``OffChain``), then data gradient into the hidden map
        off = OffChain(self, d_out.is_cuda)
        dw1, db1 = off.run(lambda: (native.conv_wgrad(t2, d_out_c, 5 * A, 1, operand=dtype).view(5 * A, C),
                                    native.bias_grad(d_out, 5 * A)), st["t"], d_out_c, d_out)
        w1t = st.get("w1t")
        if w1t is None:
            w1t = native.pack_fc_weight(st["w1"], dt, transpose=True, ld=self.ld)
        dt_ = native.conv_fwd(d_out_c, w1t, None, C, 1)
        native.act_bwd_(dt_, t2, 1)
        # 3x3 conv
        h = self.rpn_head
        dt4 = dt_.view(B, Hf, Wf, C)
        dw0, db0 = off.run(lambda: (native.conv_weight_grad(st["feat_g"], dt4, h.conv.weight, operand=dtype),
                                    native.bias_grad(dt_, C)), st["feat_g"], dt_)
        wr = st.get("wr")
        if wr is None:
            wr = native.pack_conv_weight(h.conv.weight.detach(), C, dt, rot180=True)
        dfeat = native.conv_fwd(dt4, wr, None, C, 3)
        off.join(dw0, db0, dw1, db1)
        pgrads = [dw0, db0, dw1[:A].reshape(A, C, 1, 1), db1[:A], dw1[A:].reshape(4 * A, C, 1, 1), db1[A:]]
        return dfeat.permute(0, 3, 1, 2), pgrads

    # ---- module surface (rpn.py:16-58) ------------------------------------------------------------
    def forward(self, images, features, gt_instances=None, compute_loss=True, compute_val_loss=False,
                as_instances=True, keys=None):
        feat = features[self.in_features[0]]
        want_loss = (self.training and compute_loss) or compute_val_loss
        if want_loss:
            gt = gt_instances if isinstance(gt_instances, BatchedGT) else BatchedGT.from_instances(
                gt_instances, feat.device)
            B, _, Hf, Wf = feat.shape
            keys = keys if keys is not None else getattr(self, "_forced_keys", None)
            if keys is None:
                keys = torch.randint(0, 2 ** 31 - 1, (B, Hf * Wf * self.num_anchors), dtype=torch.int32,
                                     device=feat.device)
            # losses() applies loss_weight, PseudoLabRPN.forward applies it again (rpn.py:49)
            twice = isinstance(self, PseudoLabRPN)
            wts = (self.loss_weight["loss_rpn_cls"] ** (2 if twice else 1), self.loss_weight["loss_rpn_loc"] ** (2 if twice else 1))
            l_cls, l_loc = _RPNLossFn.apply(self, feat, gt, keys, wts, *self.rpn_head.params())
            st = self._last_head_state
            losses = {"loss_rpn_cls": l_cls, "loss_rpn_loc": l_loc}
        else:
            # a head state prefetched for a LOSS pass is not consumed here (it was made under the training pass's settings):
            # dropped, so that it neither lives to the next step nor gets mistaken for this pass's
            self.__dict__.pop("_prefetched", None)
            self.__dict__.pop("_prefetched_proposals", None)
            with torch.no_grad():
                st = self._head_forward(feat)
            losses = {}
        proposals = self.__dict__.pop("_prefetched_proposals", None) if want_loss else None
        if proposals is None:
            with torch.no_grad():
                proposals = self._proposals(st, images.image_sizes)
        self._last_head_state = None
        return (proposals.to_instances() if as_instances else proposals), losses


@PROPOSAL_GENERATOR_REGISTRY.register()
class PseudoLabRPN(RPN):
    """Same compute; differs from RPN only in the forward flags and the squared loss weight."""
