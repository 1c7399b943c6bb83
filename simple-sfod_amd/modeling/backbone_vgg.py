"""VGG16(-BN) backbone on the HIP kernels, behind the reference's ``build_vgg_backbone`` name.

Mirrors ``/root/reference/daod/modeling/meta_arch/vgg.py``: layers ``:10-24``, stage split
``:70-74`` (``vgg0..vgg4``, strides 2..32 -- the 5th max-pool is part of ``vgg4``), init
``:102-113``, builder ``:116-118``.  The torch.nn containers exist only to own the parameters
under the reference's state-dict keys (``backbone.vgg{s}.{i}.*``) and to reproduce its
initialisation draw for draw; none of their forward methods is ever called.  Compute:

  conv3x3+bias (MFMA implicit GEMM, BN partial statistics in the epilogue)
  -> BN statistics finalize (+ running-stat refresh = the AdaBN update, also under no_grad)
  -> BN apply + ReLU (+ 2x2 max-pool) in one pass

Activations are NHWC internally; the NCHW tensors this module accepts/returns are zero-copy
channels-last views, so the Detectron2 ``Backbone`` surface is kept without layout traffic.
"""
import os

import torch
import torch.nn as nn

from .. import native
from . import offchain as _offchain
from ..registry import BACKBONE_REGISTRY
from ..structures import ShapeSpec

VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
STAGE_SLICES = [(0, 7), (7, 14), (14, 24), (24, 34), (34, 44)]


def make_layers(cfg_list, batch_norm=False):
    layers = []
    in_channels = 3
    for v in cfg_list:
        if v == "M":
            layers += [nn.MaxPool2d(kernel_size=2, stride=2)]
        else:
            conv2d = nn.Conv2d(in_channels, v, kernel_size=3, padding=1)
            if batch_norm:
                layers += [conv2d, nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            else:
                layers += [conv2d, nn.ReLU(inplace=True)]
            in_channels = v
    return nn.Sequential(*layers)


def nhwc_from_nchw_view(x, dtype, cpad=None):
    """NCHW-shaped tensor (any strides) -> contiguous NHWC tensor of `dtype` (channel-padded)."""
    n, c, h, w = x.shape
    p = x.permute(0, 2, 3, 1)
    if cpad is not None and cpad != c:
        out = torch.zeros(n, h, w, cpad, dtype=dtype, device=x.device)
        out[..., :c] = p
        return out
    return p.to(dtype).contiguous()


class _VGGFn(torch.autograd.Function):
    """The whole 13-layer trunk as one autograd node (hand-written backward)."""

    @staticmethod
    def forward(ctx, module, save, x_nhwc, *params):
        saved, outs = module._forward_impl(x_nhwc, save=save)
        ctx.set_materialize_grads(False)   # unused stage outputs arrive as None, not as zero tensors to copy / add
        ctx.module = module
        ctx.saved = saved
        ctx.need_dx = False
        if native.is_pairs(module.compute_dtype):
            # bf16x3: between the layers activations are (hi, lo) operand pairs; what leaves the trunk (and meets
            # autograd, whose gradients are fp32) is fp32 -- converted for the stages somebody consumes only
            # (``needed_features``; vgg4 is 1/1024 of vgg0's pixels), the others are not returned
            return tuple(native.cast(o, torch.float32).permute(0, 3, 1, 2) if n in module.needed_features else None
                         for n, o in zip(module._stage_names, outs))
        return tuple(o.permute(0, 3, 1, 2) for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        module = ctx.module
        pgrads = module._backward_impl(ctx.saved, grads)
        ctx.saved = None
        return (None, None, None) + tuple(pgrads)


class vgg_backbone(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if not cfg.VGG.BN:
            raise NotImplementedError(
                "VGG.BN=False is not on the hot path (the reference mis-slices its stages, quirk q10)")
        self.vgg = make_layers(VGG16, batch_norm=True)
        self._initialize_weights()
        chans, strides = [64, 128, 256, 512, 512], [2, 4, 8, 16, 32]
        mods = list(self.vgg._modules.values())
        self.stages = [nn.Sequential(*mods[a:b]) for a, b in STAGE_SLICES]
        self._out_feature_channels, self._out_feature_strides, self._stage_names = {}, {}, []
        for i, stage in enumerate(self.stages):
            name = "vgg{}".format(i)
            self.add_module(name, stage)
            self._stage_names.append(name)
            self._out_feature_channels[name] = chans[i]
            self._out_feature_strides[name] = strides[i]
        self._out_features = self._stage_names
        # stages whose output is materialised for consumers in bf16x3 mode (the meta-architecture narrows this to
        # the heads' IN_FEATURES); the other modes return zero-copy views of all five
        self.needed_features = set(self._stage_names)
        del self.vgg
        self.compute_dtype = native.mode_dtype(cfg.SFOD.COMPUTE_DTYPE)      # operands of the forward products
        self.grad_dtype = native.grad_dtype_of(self.compute_dtype)         # operands of dgrad / wgrad ("f16x3": bf16 pairs)
        self.bn_momentum, self.bn_eps = 0.1, 1e-5
        # momentum updates of the running statistics per training-mode forward.  The reference's student passes the same
        # batch through its backbone three times per step when the (zero-weighted) domain branch is on; with
        # SFOD.ELIDE_DEAD_BRANCHES only one pass runs and the trainer sets this to 3, which reproduces the side effect
        # exactly: r <- r (1-m)^3 + s (1 - (1-m)^3), num_batches_tracked += 3.
        self.bn_updates_per_forward = 1
        self.fuse_first = bool(cfg.SFOD.FUSE_FIRST_LAYER) if "SFOD" in cfg and "FUSE_FIRST_LAYER" in cfg.SFOD else True
        self.fuse_bn_input = bool(cfg.SFOD.FUSE_BN_INPUT) if "SFOD" in cfg and "FUSE_BN_INPUT" in cfg.SFOD else True
        self.fuse_bn_input_min_bytes = int(os.environ.get("SFOD_BNIN_MIN_BYTES", str(256 << 20)))
        self.wgrad_stream = os.environ.get("SFOD_VGG_WGRAD_STREAM", "1") != "0"
        self.fuse_bn_reduce = os.environ.get("SFOD_NO_FUSE_BN_REDUCE", "0") != "1"   # A/B hook
        # execution plan: (conv, bn, pool_after, stage_end)
        self._plan = []
        for s, stage in enumerate(self.stages):
            ms = list(stage)
            i = 0
            while i < len(ms):
                if isinstance(ms[i], nn.Conv2d):
                    pool = i + 3 < len(ms) and isinstance(ms[i + 3], nn.MaxPool2d)
                    self._plan.append([ms[i], ms[i + 1], pool, False])
                    i += 4 if pool else 3
                else:
                    i += 1
            self._plan[-1][3] = True

    # ---- Detectron2 Backbone surface ------------------------------------------------------------
    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def _param_list(self):
        ps = []
        for conv, bn, _, _ in self._plan:
            ps += [conv.weight, conv.bias, bn.weight, bn.bias]
        return ps

    def forward(self, x):
        """x: [N,3,H,W] normalised image batch (NCHW logical).  Returns {"vgg0".."vgg4"} as NCHW (channels-last)
        views -- the Detectron2 Backbone surface (fp32 tensors in fp32 and bf16x3 mode)."""
        dt = native.dt_of_dtype(self.compute_dtype)
        if native.is_pairs(self.compute_dtype):     # fp32 NHWC padded to one 8-channel group, then converted to (hi, lo) pairs
            xn = native.cast(nhwc_from_nchw_view(x, torch.float32, native.chunk_elems(dt)), self.compute_dtype)
            return self.forward_nhwc(xn)
        xn = nhwc_from_nchw_view(x, self.compute_dtype, native.chunk_elems(dt))
        return self.forward_nhwc(xn)

    def forward_nhwc(self, x_nhwc):
        params = self._param_list()
        save = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        outs = _VGGFn.apply(self, save, x_nhwc, *params)
        return {n: o for n, o in zip(self._stage_names, outs) if o is not None}

    # ---- engine -------------------------------------------------------------------------------------
    def reduce_schedule(self, pixels_per_rank):
        """Which part of this backbone's gradients may be all-reduced while its backward is still running
        (engine/trainer.py::GradientReducer): -> parameter-name prefixes of the slice that is final once stage
        ``vgg<s>``'s gradients are written; ``_backward_impl`` calls ``_mid_backward`` right there.  With enough pixels per
        rank the backward of vgg0 alone (its two most expensive layers) hides the 57 MB message, so vgg1's weights ride in it
        too and the blocking rest shrinks to vgg0 + nothing else; one 600x1200 frame per rank keeps the earlier point."""
        stage = 1 if pixels_per_rank >= 2 * 600 * 1200 else 2
        self._mid_stage = stage
        return tuple("backbone.vgg{}.".format(i) for i in range(stage, 5))

    def _first_layer_of_stage(self, stage):
        """index into ``_plan`` of the first conv of ``vgg<stage>``"""
        s, first = 0, 0
        for i, (_, _, _, stage_end) in enumerate(self._plan):
            if s == stage:
                return first
            if stage_end:
                s += 1
                first = i + 1
        return 0

    def _packed_weights(self, dt, cin0_pad, with_dgrad):
        """Packed forward weights of all layers (+ the rotated dgrad weights of layers 1.. when a backward
        follows), refreshed from the fp32 master weights by ONE launch per call (one per operand format: "f16x3" packs
        the forward weights as half pairs and the rotated ones as bf16 pairs)."""
        gdt = native.dt_of_dtype(native.grad_dtype_of(native.torch_dtype(dt)))
        if os.environ.get("SFOD_NO_MULTIPACK"):   # A/B hook: one launch per layer
            f = [native.pack_conv_weight(c.weight.detach(), cin0_pad if i == 0 else c.in_channels, dt)
                 for i, (c, _, _, _) in enumerate(self._plan)]
            r = ([None] + [native.pack_conv_weight(c.weight.detach(), c.out_channels, gdt, rot180=True)
                           for c, _, _, _ in self._plan[1:]]) if with_dgrad else None
            return f, r
        key = (dt, cin0_pad, bool(with_dgrad))
        packers = self.__dict__.setdefault("_packers", {})
        pk = packers.get(key)
        if pk is None:
            fwd = [(conv.weight, cin0_pad if li == 0 else conv.in_channels, False)
                   for li, (conv, _, _, _) in enumerate(self._plan)]
            rot = [(conv.weight, conv.out_channels, True) for conv, _, _, _ in self._plan[1:]] if with_dgrad else []
            if gdt == dt or not rot:
                pk = (native.ConvWeightPacker(fwd + rot, dt), None)
            else:
                pk = (native.ConvWeightPacker(fwd, dt), native.ConvWeightPacker(rot, gdt))
            packers[key] = pk
        views = list(pk[0].pack()) + (list(pk[1].pack()) if pk[1] is not None else [])
        n = len(self._plan)
        return views[:n], ([None] + list(views[n:]) if with_dgrad else None)

    def _defer_bn(self, li, y, fwd_w, save, training):
        """May layer ``li``'s BatchNorm + ReLU be left to the next convolution's operand path (sfod_conv_fwd_bnin)?  Only in a
        train-mode forward-only pass (the teacher: nobody needs the activated tensor -- no backward, no pooling, not a stage
        output) and where the consumer's shape is served."""
        if save or not training or not self.fuse_bn_input or li + 1 >= len(self._plan):
            return False
        _, _, pool, stage_end = self._plan[li]
        if pool or stage_end:
            return False
        # the in-LDS transform costs the convolution 12-15 % (it is not hidden behind the MFMAs); the apply pass it removes
        # costs its HBM bytes, and a tensor that fits the 256 MB Infinity Cache is cheap to re-read: the fold pays for
        # conv2_2 / conv3_2 / conv3_3 of a teacher batch at 600 x 1200 and loses on conv4_x (profiles/round4/r4_bnin_layers.txt)
        if y.numel() * 4 < self.fuse_bn_input_min_bytes:
            return False
        nxt = self._plan[li + 1][0]
        return native.conv_fwd_bnin_supported(y, fwd_w[li + 1], nxt.out_channels)

    def _forward_impl(self, x, save=True):
        dt = native.dt_of(x)
        training = self.training
        saved, outs = [], []
        fwd_w, rot_w = self._packed_weights(dt, x.shape[-1], save)
        self._rot_w = rot_w
        x_g = native.as_operand(x, self.grad_dtype) if save else None      # conv1_1's weight-gradient operand (8 channels)
        pending = None      # (y, mean, invstd, bn) of the layer below when its BatchNorm + ReLU is left to this layer's conv
        for li, (conv, bn, pool, stage_end) in enumerate(self._plan):
            cout = conv.out_channels
            wp = fwd_w[li]
            if pending is not None:
                # forward-only pass: the layer below left its BatchNorm + ReLU to this convolution's operand path
                yb, mb, ib, bnb = pending
                pending = None
                B, H, W, _ = yb.shape
                y, stats = native.conv_fwd_bnin(yb, mb, ib, bnb.weight.detach(), bnb.bias.detach(), wp, conv.bias.detach(),
                                                cout, want_stats=True)
                mean, invstd = native.bn_finalize(stats, B * H * W, cout, bn.running_mean, bn.running_var,
                                                  self.bn_momentum, self.bn_eps, self.bn_updates_per_forward,
                                                  num_batches_tracked=bn.num_batches_tracked)
                if self._defer_bn(li, y, fwd_w, save, training):
                    pending = (y, mean, invstd, bn)
                    continue
                x = native.bn_relu_pool_fwd(y, mean, invstd, bn.weight.detach(), bn.bias.detach(), pool, out_dtype=x.dtype)
                if stage_end:
                    outs.append(x)
                continue
            B, H, W, _ = x.shape
            if (li == 0 and training and not save and not pool and not stage_end and self.fuse_first
                    and native.conv_first_supported(x, cout)):
                # forward-only (teacher): the first layer's K is 27 and its cost is its 64-channel output, so
                # BatchNorm + ReLU are folded in by recomputation -- a store-free statistics pass, then a second
                # pass that writes z directly; y (only a backward would need it) is never materialised
                stats = native.conv_first_stats(x, wp, conv.bias.detach())
                mean, invstd = native.bn_finalize(stats, B * H * W, cout, bn.running_mean, bn.running_var,
                                                  self.bn_momentum, self.bn_eps, self.bn_updates_per_forward,
                                                  num_batches_tracked=bn.num_batches_tracked)
                scale = bn.weight.detach() * invstd
                shift = bn.bias.detach() - mean * scale
                x = native.conv_first_apply(x, wp, conv.bias.detach(), scale, shift, relu=True)
                continue
            if training:
                y, stats = native.conv_fwd(x, wp, conv.bias.detach(), cout, 3, want_stats=True)
                mean, invstd = native.bn_finalize(stats, B * H * W, cout, bn.running_mean, bn.running_var,
                                                  self.bn_momentum, self.bn_eps, self.bn_updates_per_forward,
                                                  num_batches_tracked=bn.num_batches_tracked)
            else:
                y = native.conv_fwd(x, wp, conv.bias.detach(), cout, 3)
                mean = bn.running_mean
                invstd = torch.rsqrt(bn.running_var + self.bn_eps)
            if self._defer_bn(li, y, fwd_w, save, training):
                pending = (y, mean, invstd, bn)
                continue
            # bf16x3 / f16x3: y is fp32, z is written as (hi, lo) pairs; a pass that will be differentiated also gets
            # the operand of the next layer's weight gradient from the same launch (f16x3: bf16 pairs; else z itself)
            zz = native.bn_relu_pool_fwd(y, mean, invstd, bn.weight.detach(), bn.bias.detach(), pool,
                                         out_dtype=x.dtype, with_grad_operand=save)
            z, z_g = zz if save else (zz, None)
            if save:
                saved.append((x_g, y, mean, invstd))
            x, x_g = z, z_g
            if stage_end:
                outs.append(z)
        return saved, outs

    def _backward_impl(self, saved, out_grads):
        """out_grads: grads of the 5 stage outputs (NCHW views or None) -> flat list of param grads."""
        hook = getattr(self, "_pre_backward", None)
        if hook is not None:
            hook()   # e.g. GradientReducer.launch_early: the heads' gradients are final now
        dtype = self.grad_dtype
        pgrads = [None] * (4 * len(self._plan))
        stage_of = []
        s = 0
        for i, (_, _, _, stage_end) in enumerate(self._plan):
            stage_of.append(s if stage_end else None)
            if stage_end:
                s += 1
        dz = dz_red = None
        # weight gradients on a second HIP stream beside the data-gradient chain (the two MFMA kernels of a layer's backward
        # are independent): fills the chip where one kernel's grid does not -- one frame per GPU, the ragged last round of
        # the deep layers (SFOD_VGG_WGRAD_STREAM=0: everything on one stream)
        side, side_keep = None, []
        if self.wgrad_stream and dz_dev_is_cuda(out_grads):
            side = self.__dict__.get("_side_stream")
            if side is None:
                side = self.__dict__["_side_stream"] = _offchain.shared_stream() if _offchain._SHARED else torch.cuda.Stream()
        mid_hook = getattr(self, "_mid_backward", None)
        mid_layer = self._first_layer_of_stage(getattr(self, "_mid_stage", 2)) if mid_hook is not None else -1
        for li in range(len(self._plan) - 1, -1, -1):
            conv, bn, pool, stage_end = self._plan[li]
            x, y, mean, invstd = saved[li]
            if stage_end and out_grads[stage_of[li]] is not None:
                g = out_grads[stage_of[li]].permute(0, 2, 3, 1).to(native.out_dtype_of(dtype)).contiguous()
                dz = g if dz is None else native.add_(dz, g)
            if dz is None:
                continue
            # parameter gradients go straight into the flat gradient buffer when the parameter has one
            # (grad_sink): no temporaries, no per-parameter accumulate kernels in autograd
            gsink, bsink = native.grad_sink(bn.weight), native.grad_sink(bn.bias)
            direct_bn = gsink is not None and bsink is not None
            dy, dgamma, dbeta = native.bn_relu_pool_bwd(dz, y, mean, invstd, bn.weight.detach(),
                                                        bn.bias.detach(), pool,
                                                        dgamma_acc=gsink if direct_bn else None,
                                                        dbeta_acc=bsink if direct_bn else None,
                                                        out_dtype=dtype,    # bf16x3: dz / y fp32 -> dy pairs
                                                        reduced=dz_red)
            dz_red = None
            if direct_bn:
                dgamma = dbeta = None
            cout, cin = conv.out_channels, conv.in_channels
            if side is not None:
                # dy is complete on the main stream; the weight gradient reads (x, dy) on the side stream while the main one
                # goes on with the data gradient.  Their memory belongs to the main stream's pool: record_stream tells the
                # allocator not to hand it out again before the side kernel has run, so the references can be dropped
                # layer by layer (holding all 13 (x, dy) pairs until the join cost ~180 MB per image and layer at conv1_x)
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    dw = native.conv_weight_grad(x, dy, conv.weight)
                for t_ in (x, dy):
                    if t_ is not None:
                        t_.record_stream(side)
                side_keep.append(dw)
            else:
                dw = native.conv_weight_grad(x, dy, conv.weight)
            # a conv bias followed by train-mode BN has an analytically zero gradient
            # (sum_rows dy == 0); the reference's autograd produces rounding noise around 0.
            db = None if native.grad_sink(conv.bias) is not None else torch.zeros_like(conv.bias)
            pgrads[4 * li:4 * li + 4] = [dw, db, dgamma, dbeta]
            if li > 0:
                # rotated weights were packed together with the forward ones (same step, same values).  When the layer
                # below has no pooling and takes no extra stage gradient, its BatchNorm-backward reduction rides in
                # this data-gradient kernel's epilogue (one pass over dz and y less)
                fused = None
                _, bn_b, pool_b, stage_end_b = self._plan[li - 1]
                if self.fuse_bn_reduce and not pool_b and not (stage_end_b and out_grads[stage_of[li - 1]] is not None):
                    yb, mb, ib = saved[li - 1][1], saved[li - 1][2], saved[li - 1][3]
                    fused = native.conv_dgrad_bnred(dy, self._rot_w[li], cin, yb, mb, ib, bn_b.weight.detach(),
                                                    bn_b.bias.detach())
                if fused is not None:
                    dz, dz_red = fused
                else:
                    dz = native.conv_fwd(dy, self._rot_w[li], None, cin, 3)
            del saved[li]
            if li == mid_layer and mid_hook is not None:
                if side is not None:     # the collective is ordered behind the main stream: the side stream's gradients first
                    torch.cuda.current_stream().wait_stream(side)
                mid_hook()   # gradients of vgg2..vgg4 are in the flat buffer: GradientReducer.launch_mid
        if side is not None:             # every weight gradient is in its buffer before anyone (all-reduce, SGD, autograd) reads it
            main = torch.cuda.current_stream()
            main.wait_stream(side)
            for dw in side_keep:
                if dw is not None:
                    dw.record_stream(main)
        return pgrads


def dz_dev_is_cuda(grads):
    return any(g is not None and g.is_cuda for g in grads)


@BACKBONE_REGISTRY.register()
def build_vgg_backbone(cfg, _=None):
    return vgg_backbone(cfg)
