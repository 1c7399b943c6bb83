"""ResNet-C4 backbone (ResNet-50/101, ``res4`` output) on the HIP kernels, behind Detectron2's
``build_resnet_backbone`` name.

The reference selects this backbone by NOT naming one
(``/root/reference/configs/r101_c4_cs_foggy_adaptive_teacher_source_free.yaml:1-28``: ``RESNETS.DEPTH
101``, ``NORM BN``, default ``BACKBONE.NAME = build_resnet_backbone``, ``FREEZE_AT 2``); the module
itself lives in Detectron2 (not vendored), restated here from its published structure (SURVEY.md
8a row a2, Appendix A.14):

  BasicStem        conv 7x7 s2 p3 (3->64, no bias) + norm + ReLU + max_pool 3x3 s2 p1
  BottleneckBlock  conv1 1x1 (stride here: STRIDE_IN_1X1) -> norm -> ReLU -> conv2 3x3 -> norm -> ReLU
                   -> conv3 1x1 -> norm ; (+ shortcut 1x1 conv + norm when channels change) ; ReLU
  stages           res2 (3 blocks, 256 ch, stride 4), res3 (4, 512, /8), res4 (6 | 23, 1024, /16)
  freeze           FREEZE_AT=2: stem + res2 frozen, their norms become FrozenBatchNorm2d (affine);
                   res3 / res4 keep live BatchNorm2d (train mode: batch statistics + running-stat
                   refresh = the AdaBN update, also under no_grad for the teacher)

State-dict keys follow Detectron2 (``backbone.stem.conv1.{weight,norm.*}``,
``backbone.res{2,3,4}.{i}.{conv1,conv2,conv3,shortcut}.{weight,norm.*}``).  The torch modules only own
parameters; compute:

  frozen stem/res2  FrozenBN folded into the packed weights (scale) and the bias (shift): one conv
                    kernel per layer with fused ReLU; the 7x7 stem runs as im2col + MFMA GEMM
  live blocks       conv (+ BN partial statistics in the epilogue) -> finalize -> BN apply (+ReLU);
                    residual join relu(a + b); hand-written backward (BN, wgrad, dgrad per conv)
  stride-2 1x1      even-pixel subsampling shared by conv1 and the shortcut, then plain GEMMs
"""
import os

import torch
import torch.nn as nn

from .. import native
from . import offchain as _offchain
from ..registry import BACKBONE_REGISTRY
from ..structures import ShapeSpec


class FrozenBatchNorm2d(nn.Module):
    """d2 FrozenBatchNorm2d: buffers only; y = x * (w * rsqrt(var + eps)) + (b - mean * w * rsqrt(var + eps))."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale


class Conv2d(nn.Conv2d):
    """d2 layers.Conv2d: a bias-free conv that owns its norm as ``.norm``."""

    def __init__(self, cin, cout, k, stride=1, padding=0, norm=None):
        super().__init__(cin, cout, kernel_size=k, stride=stride, padding=padding, bias=False)
        self.norm = norm
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")  # c2_msra_fill


def _norm(kind, ch):
    if kind == "BN":
        return nn.BatchNorm2d(ch)
    if kind == "FrozenBN":
        return FrozenBatchNorm2d(ch)
    raise NotImplementedError(f"RESNETS.NORM={kind}")


class BasicStem(nn.Module):
    def __init__(self, cin, cout, norm):
        super().__init__()
        self.conv1 = Conv2d(cin, cout, 7, stride=2, padding=3, norm=_norm(norm, cout))
        self.in_channels, self.out_channels, self.stride = cin, cout, 4


class BottleneckBlock(nn.Module):
    def __init__(self, cin, cout, bottleneck, stride, norm, stride_in_1x1=True):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = cin, cout, stride
        self.shortcut = Conv2d(cin, cout, 1, stride=stride, norm=_norm(norm, cout)) if cin != cout else None
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        if s3 != 1:
            raise NotImplementedError("stride in the 3x3 conv (STRIDE_IN_1X1=False) is not on the hot path")
        self.conv1 = Conv2d(cin, bottleneck, 1, stride=s1, norm=_norm(norm, bottleneck))
        self.conv2 = Conv2d(bottleneck, bottleneck, 3, stride=s3, padding=1, norm=_norm(norm, bottleneck))
        self.conv3 = Conv2d(bottleneck, cout, 1, norm=_norm(norm, cout))

    def convs(self):
        cs = [self.conv1, self.conv2, self.conv3]
        return cs + ([self.shortcut] if self.shortcut is not None else [])


def _freeze(module):
    """d2 CNNBlockBase.freeze: requires_grad False + BatchNorm -> FrozenBatchNorm2d (same statistics)."""
    for p in module.parameters():
        p.requires_grad = False
    for m in module.modules():
        if isinstance(m, Conv2d) and isinstance(m.norm, nn.BatchNorm2d):
            f = FrozenBatchNorm2d(m.norm.num_features, m.norm.eps)
            f.weight.copy_(m.norm.weight.data)
            f.bias.copy_(m.norm.bias.data)
            f.running_mean.copy_(m.norm.running_mean)
            f.running_var.copy_(m.norm.running_var)
            m.norm = f


class _ResNetFn(torch.autograd.Function):
    """The whole trunk as one autograd node (hand-written backward through the live stages)."""

    @staticmethod
    def forward(ctx, module, save, x_nhwc, *params):
        saved, outs = module._forward_impl(x_nhwc, save=save)
        ctx.set_materialize_grads(False)   # unused stage outputs arrive as None, not as zero tensors
        ctx.module, ctx.saved = module, saved
        return tuple(o.permute(0, 3, 1, 2) for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        pgrads = ctx.module._backward_impl(ctx.saved, grads)
        ctx.saved = None
        return (None, None, None) + tuple(pgrads)


class ResNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        r = cfg.MODEL.RESNETS
        depth = r.DEPTH
        blocks_per_stage = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
        assert r.NUM_GROUPS == 1 and r.RES5_DILATION == 1, "plain bottleneck ResNets only"
        self.norm_kind = r.NORM
        out_features = list(r.OUT_FEATURES)
        names = ["res2", "res3", "res4", "res5"]
        last = max(names.index(f) for f in out_features)
        self.stem = BasicStem(3, r.STEM_OUT_CHANNELS, r.NORM)
        cin, cout, bott = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS, r.NUM_GROUPS * r.WIDTH_PER_GROUP
        self.stage_names = []
        self._out_feature_channels, self._out_feature_strides = {}, {}
        stride_total = 4
        for si in range(last + 1):
            first_stride = 1 if si == 0 else 2
            blocks = []
            for bi in range(blocks_per_stage[si]):
                blocks.append(BottleneckBlock(cin, cout, bott, first_stride if bi == 0 else 1, r.NORM,
                                              r.STRIDE_IN_1X1))
                cin = cout
            stride_total *= first_stride
            self.add_module(names[si], nn.Sequential(*blocks))
            self.stage_names.append(names[si])
            self._out_feature_channels[names[si]] = cout
            self._out_feature_strides[names[si]] = stride_total
            cout, bott = cout * 2, bott * 2
        self._out_features = out_features
        # d2 build_resnet_backbone: freeze_at >= 1 freezes the stem, >= k+1 freezes res{k+1}
        self.freeze_at = cfg.MODEL.BACKBONE.FREEZE_AT
        self.frozen = set()
        if self.freeze_at >= 1:
            _freeze(self.stem)
            self.frozen.add("stem")
        for i, n in enumerate(self.stage_names, start=2):
            if self.freeze_at >= i:
                _freeze(getattr(self, n))
                self.frozen.add(n)
        for n in self.stage_names:
            if n not in self.frozen and self.norm_kind != "BN":
                raise NotImplementedError("trainable stages need RESNETS.NORM=BN (the named r101 config)")
        # MFMA operand dtype (packed weights) and the dtype activations are stored in between the kernels.  bf16x3:
        # activations stay fp32 here (residual joins, stride-2 subsampling and the elementwise backward all work on
        # them) and are converted to (hi, lo) operand pairs at each convolution's input (native.as_operand); the VGG
        # trunk instead has its BatchNorm kernels write the pairs directly.
        self.compute_dtype = native.mode_dtype(cfg.SFOD.COMPUTE_DTYPE)       # operands of the forward products
        self.grad_dtype = native.grad_dtype_of(self.compute_dtype)          # operands of dgrad / wgrad ("f16x3": bf16 pairs)
        self.act_dtype = native.out_dtype_of(self.compute_dtype)
        self.bn_momentum = 0.1
        self._frozen_gen = 0                 # generation of the frozen stages' values (see _frozen_packed)
        self.bn_updates_per_forward = 1      # see backbone_vgg: momentum updates folded into one forward
        self.fuse_residual = os.environ.get("SFOD_NO_FUSE_RESIDUAL", "0") != "1"   # A/B hook: bn3 + shortcut + ReLU in one pass
        self.dual_join = os.environ.get("SFOD_NO_DUAL_JOIN", "0") != "1"           # A/B hook: join kernels also emit the operand pairs
        self.fuse_stem = os.environ.get("SFOD_NO_FUSE_STEM", "0") != "1"                   # A/B hook: 7x7 stem without the im2col matrix
        # weight gradients of the live bottleneck convolutions on a second HIP stream beside the data-gradient chain: at this
        # config's sizes (res4: 22 800 rows) one 1x1 kernel fills 70 % of the CUs for 60 us, so the two independent
        # GEMMs of a layer's backward share the chip (SFOD_RESNET_WGRAD_STREAM=0: everything on one stream)
        self.wgrad_stream = os.environ.get("SFOD_RESNET_WGRAD_STREAM", "1") != "0"

    # ---- Detectron2 Backbone surface -----------------------------------------------------------------
    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    def _live_blocks(self):
        return [b for n in self.stage_names if n not in self.frozen for b in getattr(self, n)]

    def reduce_schedule(self, pixels_per_rank):
        """(engine/trainer.py::GradientReducer) -> parameter-name prefixes of the gradient slice that is final while this
        backbone's backward is still running: the live stage the backward finishes FIRST (the last one, res4 on the C4 path:
        23 of R101's 27 live blocks, 26 M of its 27.3 M live parameters).  ``_backward_impl`` joins the weight-gradient side
        stream and calls ``_mid_backward`` when that stage's first block is done; what is left for the blocking phase is the
        earlier live stages (res3: 1.2 M parameters, 4.9 MB)."""
        live = [n for n in self.stage_names if n not in self.frozen]
        if len(live) < 2:
            self._mid_stage_name = None     # a single live stage: its end is the end of the backward
            return ()
        self._mid_stage_name = live[-1]
        return ("backbone.{}.".format(live[-1]),)

    def _param_list(self):
        ps = []
        for blk in self._live_blocks():
            for c in blk.convs():
                ps += [c.weight, c.norm.weight, c.norm.bias]
        return ps

    def forward(self, x):
        dt = native.dt_of_dtype(self.compute_dtype)
        n, c, h, w = x.shape
        xn = torch.zeros(n, h, w, native.chunk_elems(dt), dtype=self.act_dtype, device=x.device)
        xn[..., :c] = x.permute(0, 2, 3, 1)
        return self.forward_nhwc(xn)

    def forward_nhwc(self, x_nhwc):
        params = self._param_list()
        save = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        outs = _ResNetFn.apply(self, save, x_nhwc, *params)
        return dict(zip(self._out_features, outs))

    # ---- engine ------------------------------------------------------------------------------------------
    def _frozen_packed(self, conv, cin_pad, dt):
        """(packed weights with the FrozenBN scale folded in, shift) of a frozen conv, cached: frozen tensors
        only change through load_state_dict / EMA copies, which bump their version counters."""
        # _frozen_gen: bumped by whoever rewrites parameters through the flat buffers (the fused EMA step updates every
        # key of the teacher, frozen ones included -- source_free_adaptive_teacher.py:583-603 -- without touching the
        # views' version counters)
        key = (conv.weight._version, conv.norm.weight._version, conv.norm.bias._version,
               conv.norm.running_mean._version, conv.norm.running_var._version, conv.weight.data_ptr(), cin_pad, dt,
               self._frozen_gen)
        hit = getattr(conv, "_packed_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        scale, shift = conv.norm.scale_shift()
        w = conv.weight.detach() * scale.view(-1, 1, 1, 1)
        wp = native.pack_conv_weight(w, cin_pad, dt)
        shift = shift.contiguous()
        conv._packed_cache = (key, wp, shift)
        return wp, shift

    def _frozen_conv(self, x, conv, act, dt):
        """conv + FrozenBN (+ReLU) as one kernel: scale folded into the weights, shift as the bias."""
        wp, shift = self._frozen_packed(conv, x.shape[-1], dt)
        return native.conv_fwd(x, wp, shift, conv.out_channels, conv.kernel_size[0], act=act)

    def _stem_forward(self, x, dt):
        conv = self.stem.conv1
        assert isinstance(conv.norm, FrozenBatchNorm2d), "the stem is frozen on the hot path (FREEZE_AT >= 1)"
        scale, shift = conv.norm.scale_shift()
        w = (conv.weight.detach() * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(conv.out_channels, 147)
        # 147 columns padded to whole K tiles of the GEMM's uniform-tap path (8 x 16-byte chunks): 160 (5 tiles) for fp32 / pairs, 192 (3) for bf16
        kpad = 192 if self.compute_dtype == torch.bfloat16 else 160
        wk = torch.zeros(conv.out_channels, kpad, dtype=torch.float32, device=x.device)
        wk[:, :147] = w
        wp = native.pack_fc_weight(wk, dt)
        if self.fuse_stem and kpad == 160 and conv.out_channels == 64 and native.stem7x7_supported(x, dt):
            # operand pairs: the im2col matrix (921 MB for eight 600x1200 frames) is never written -- gathered from an LDS
            # patch inside the kernel (csrc/stem7x7.hip; bit-identical to the two launches below)
            y = native.stem7x7(x, wp, shift.contiguous(), act=1)
            return native.maxpool3s2(y)
        cols = native.im2col_stem(x, kpad, out_dtype=self.compute_dtype if native.is_pairs(self.compute_dtype) else None)
        B, Ho, Wo, _ = cols.shape
        y = native.conv_fwd(cols.view(B * Ho * Wo, kpad), wp, shift.contiguous(), conv.out_channels, 1, act=1)
        return native.maxpool3s2(y.view(B, Ho, Wo, conv.out_channels))

    def _pack_live_weights(self, dt, with_dgrad):
        """Forward (and, for a backward, rotated) packed weights of every live conv in ONE launch per forward and operand
        format (native.ConvWeightPacker) instead of two tiny launches per conv."""
        key = (dt, bool(with_dgrad))
        gdt = native.dt_of_dtype(self.grad_dtype)
        packers = self.__dict__.setdefault("_packers", {})
        ent = packers.get(key)
        if ent is None:
            convs = []
            for name in self.stage_names:
                if name in self.frozen:
                    continue
                for blk in getattr(self, name):
                    for c in (blk.conv1, blk.conv2, blk.conv3, blk.shortcut):
                        if c is not None:
                            convs.append(c)
            fwd = [(c.weight, c.in_channels, False) for c in convs]
            rot = [(c.weight, c.out_channels, True) for c in convs] if with_dgrad else []
            if gdt == dt or not rot:
                pks = (native.ConvWeightPacker(fwd + rot, dt), None)
            else:
                pks = (native.ConvWeightPacker(fwd, dt), native.ConvWeightPacker(rot, gdt))
            ent = packers[key] = (pks, convs)
        pks, convs = ent
        views = list(pks[0].pack()) + (list(pks[1].pack()) if pks[1] is not None else [])
        n = len(convs)
        self._wp = {id(c): views[i] for i, c in enumerate(convs)}
        self._wr = {id(c): views[n + i] for i, c in enumerate(convs)} if with_dgrad else {}

    def _live_conv_bn(self, x, conv, relu, dt, residual=None, z_operand=False, dual=False, z_grad=False, dual_g=False):
        """conv + train-mode BatchNorm (+ ReLU / residual join).  ``x``: an MFMA operand tensor (bf16x3: pairs) or an
        activation-dtype tensor (converted by conv_fwd).  ``z_operand``: write the output directly as the next
        convolution's operand (bf16x3: the BatchNorm kernel emits the pairs; no separate conversion pass)."""
        bn = conv.norm
        k = conv.kernel_size[0]
        wp = self.__dict__.get("_wp", {}).get(id(conv)) if x.shape[-1] == conv.in_channels else None
        if wp is None:
            wp = native.pack_conv_weight(conv.weight.detach(), x.shape[-1], dt)
        B, H, W, _ = x.shape
        if self.training:
            y, stats = native.conv_fwd(x, wp, None, conv.out_channels, k, want_stats=True)
            mean, invstd = native.bn_finalize(stats, B * H * W, conv.out_channels, bn.running_mean, bn.running_var,
                                              self.bn_momentum, bn.eps, self.bn_updates_per_forward,
                                              num_batches_tracked=bn.num_batches_tracked)
        else:
            y = native.conv_fwd(x, wp, None, conv.out_channels, k)
            mean, invstd = bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
        if residual is not None:      # bottleneck tail: relu(bn(y) + shortcut) without materialising bn(y)
            # dual: z = (fp32 block output, the same as operand pairs for the next block's conv1 / shortcut)
            z = native.bn_add_relu_fwd(y, mean, invstd, bn.weight.detach(), bn.bias.detach(), residual,
                                       with_operand=self.compute_dtype if dual else None, with_grad_operand=dual and dual_g)
        else:
            # z_grad (a pass that will be differentiated): z = (forward operand, the operand of the consuming
            # convolution's weight gradient) from one launch -- the same tensor unless the mode splits them (f16x3)
            z = native.bn_relu_pool_fwd(y, mean, invstd, bn.weight.detach(), bn.bias.detach(), False, relu=relu,
                                        out_dtype=self.compute_dtype if z_operand else None,
                                        with_grad_operand=z_grad and z_operand)
        return y, mean, invstd, z

    def _block_forward(self, blk, x, live, dt, x_op=None, want_op=False, save=False, want_g=False):
        """-> (block output, its operand-pair copy or None, saved activations or None).  ``x_op``: the block input as
        MFMA operand when the producer already wrote it (bf16x3: the previous block's join kernel emits the fp32
        residual stream AND the pairs in one pass), or the tuple (forward operand, weight-gradient operand) when it wrote
        both (f16x3, differentiated pass); ``want_op`` / ``want_g``: do the same for the next block."""
        dual = want_op and self.dual_join and native.is_pairs(self.compute_dtype) and x.shape[-1] % 8 == 0
        want_g = want_g and dual and self.grad_dtype != self.compute_dtype and os.environ.get("SFOD_NO_JOIN_G", "0") != "1"   # A/B hook
        x_g_in = None
        if isinstance(x_op, tuple):
            x_op, x_g_in = x_op
        if blk.stride == 2:
            xs, x_op, x_g_in = native.subsample2(x), None, None
        else:
            xs = x
        # the block input becomes an operand ONCE, shared by conv1, the shortcut conv and (live blocks) both weight
        # gradients
        split_g = live and save and self.grad_dtype != self.compute_dtype
        xs_g = x_g_in
        if x_op is not None:
            xs_op = x_op
        else:       # (f16x3, differentiated pass: half pairs AND the weight gradient's bf16 pairs from one pass over xs)
            xs_op, xs_g = native.operands_for(xs, self.compute_dtype, need_grad=split_g)
        if not live:
            sc = x if blk.shortcut is None else self._frozen_conv(xs_op, blk.shortcut, 0, dt)
            o = self._frozen_conv(xs_op, blk.conv1, 1, dt)
            o = self._frozen_conv(o, blk.conv2, 1, dt)
            o = self._frozen_conv(o, blk.conv3, 0, dt)
            if dual and want_g:
                out, out_op, out_g = native.add_act(o, sc, 1, with_operand=self.compute_dtype, with_grad_operand=True)
                return out, (out_op, out_g), None
            if dual:
                out, out_op = native.add_act(o, sc, 1, with_operand=self.compute_dtype)
                return out, out_op, None
            return native.add_act(o, sc, 1), None, None
        # a1 / a2 only feed convolutions, so their BatchNorm kernels write pairs directly (a differentiated pass: also the
        # weight-gradient operands a1g / a2g -- the same tensors except in f16x3 mode)
        xs_act, xs = xs, xs_op
        if not split_g:
            xs_g = xs
        elif xs_g is None:
            xs_g = native.as_operand(xs_act, self.grad_dtype)
        y1, m1, i1, a1 = self._live_conv_bn(xs, blk.conv1, True, dt, z_operand=True, z_grad=split_g)
        a1, a1g = a1 if split_g else (a1, a1)
        y2, m2, i2, a2 = self._live_conv_bn(a1, blk.conv2, True, dt, z_operand=True, z_grad=split_g)
        a2, a2g = a2 if split_g else (a2, a2)
        if blk.shortcut is not None:
            ys, ms, is_, ts = self._live_conv_bn(xs, blk.shortcut, False, dt)
        else:
            ys = ms = is_ = None
            ts = x
        del xs_act
        out_op = None
        if self.fuse_residual:
            y3, m3, i3, out = self._live_conv_bn(a2, blk.conv3, False, dt, residual=ts, dual=dual, dual_g=want_g)
            if dual and want_g:
                out, out_op = out[0], (out[1], out[2])
            elif dual:
                out, out_op = out
        else:
            y3, m3, i3, t3 = self._live_conv_bn(a2, blk.conv3, False, dt)
            out = native.add_act(t3, ts, 1, with_operand=self.compute_dtype if dual else None, with_grad_operand=want_g)
            if dual and want_g:
                out, out_op = out[0], (out[1], out[2])
            elif dual:
                out, out_op = out
        return out, out_op, (x.shape, xs_g, y1, m1, i1, a1g, y2, m2, i2, a2g, y3, m3, i3, ys, ms, is_, out)

    def _forward_impl(self, x, save=True):
        dt = native.dt_of_dtype(self.compute_dtype)
        if x.dtype != self.act_dtype:        # bf16x3: the preprocess kernel's pair tensor (3 channels: tiny) -> fp32
            x = native.cast(x, self.act_dtype)
        saved, outs = [], {}
        self._pack_live_weights(dt, with_dgrad=save)
        x = self._stem_forward(x, dt)
        x_op = None
        blocks = [(name, blk) for name in self.stage_names for blk in getattr(self, name)]
        for bi, (name, blk) in enumerate(blocks):
            live = name not in self.frozen
            # the next block reads this output as a convolution operand unless it subsamples first (stride 2)
            want_op = bi + 1 < len(blocks) and blocks[bi + 1][1].stride != 2
            # ... and, in a differentiated f16x3 pass, a live next block also wants the bf16 pairs for its weight gradients
            want_g = want_op and save and blocks[bi + 1][0] not in self.frozen
            x, x_op, sv = self._block_forward(blk, x, live, dt, x_op=x_op, want_op=want_op, save=save, want_g=want_g)
            if live and save:
                saved.append(sv)
            if name in self._out_features and (bi + 1 == len(blocks) or blocks[bi + 1][0] != name):
                outs[name] = x
        return saved, [outs[n] for n in self._out_features]

    def _conv_bwd(self, g, x_in, y, mean, invstd, conv, relu, need_dx):
        """grad wrt the norm output -> (dx or None, [dw, dgamma, dbeta])."""
        bn = conv.norm
        k = conv.kernel_size[0]
        gsink, bsink = native.grad_sink(bn.weight), native.grad_sink(bn.bias)
        direct_bn = gsink is not None and bsink is not None
        dy, dgamma, dbeta = native.bn_relu_pool_bwd(g, y, mean, invstd, bn.weight.detach(), bn.bias.detach(), False,
                                                    relu=relu, dgamma_acc=gsink if direct_bn else None,
                                                    dbeta_acc=bsink if direct_bn else None,
                                                    out_dtype=self.grad_dtype)   # dy only feeds wgrad / dgrad MFMAs
        if direct_bn:
            dgamma = dbeta = None
        side = self.__dict__.get("_side")
        if side is not None:
            # dy is complete on the main stream; the weight gradient reads (x_in, dy) on the side stream while the main
            # stream goes on with the data gradient.  record_stream keeps the allocator from handing their memory out again
            # before the side kernel has run, so they need not stay referenced until the join at the end of the backward.
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dw = native.conv_weight_grad(x_in, dy, conv.weight, operand=self.grad_dtype)
            for t_ in (x_in, dy):
                if t_ is not None:
                    t_.record_stream(side)
            self._side_keep.append(dw)
        else:
            dw = native.conv_weight_grad(x_in, dy, conv.weight, operand=self.grad_dtype)
        dx = None
        if need_dx:
            wr = self.__dict__.get("_wr", {}).get(id(conv))      # packed with the forward weights (same step, same values)
            if wr is None:
                wr = native.pack_conv_weight(conv.weight.detach(), conv.out_channels,
                                             native.dt_of_dtype(self.grad_dtype), rot180=True)
            dx = native.conv_fwd(dy, wr, None, conv.in_channels, k)
        return dx, [dw, dgamma, dbeta]

    def _block_backward(self, blk, sv, dout, need_dx=True, masked=False, below_out=None):
        """dout (grad of the block output, consumed) -> (dx or None, [dw, dgamma, dbeta] per conv, dx already taken
        through the ReLU of the block below).  ``masked``: dout already went through this block's joining ReLU (the
        block above fused it into its own last pass); ``below_out``: output of the block below -- with an identity
        shortcut the sum of the two input-gradient branches and that block's ReLU mask are one pass."""
        (xshape, xs, y1, m1, i1, a1, y2, m2, i2, a2, y3, m3, i3, ys, ms, is_, out) = sv
        g = dout if masked else native.act_bwd_(dout, out, 1)  # through the joining ReLU
        da2, p3 = self._conv_bwd(g, a2, y3, m3, i3, blk.conv3, False, True)
        da1, p2 = self._conv_bwd(da2, a1, y2, m2, i2, blk.conv2, True, True)
        dxs, p1 = self._conv_bwd(da1, xs, y1, m1, i1, blk.conv1, True, need_dx)
        ps = []
        if blk.shortcut is not None:
            dsc, ps = self._conv_bwd(g, xs, ys, ms, is_, blk.shortcut, False, need_dx)
            if need_dx:
                dxs = native.add_(dxs, dsc)
        elif need_dx:
            if below_out is not None and blk.stride != 2:
                return native.add_act_bwd_(dxs, g, below_out), p1 + p2 + p3 + ps, True
            dxs = native.add_(dxs, g)
        dx = None
        if need_dx:
            dx = native.subsample2_bwd(dxs, xshape) if blk.stride == 2 else dxs
        return dx, p1 + p2 + p3 + ps, False

    def _backward_impl(self, saved, out_grads):
        hook = getattr(self, "_pre_backward", None)
        if hook is not None:
            hook()   # e.g. GradientReducer.launch_early: the heads' gradients are final now
        blocks = self._live_blocks()
        assert len(saved) == len(blocks)
        self._side, self._side_keep = None, []
        if self.wgrad_stream and torch.cuda.is_available():
            st = self.__dict__.get("_side_stream")
            if st is None:
                st = self.__dict__["_side_stream"] = _offchain.shared_stream() if _offchain._SHARED else torch.cuda.Stream()
            self._side = st
        # only the last requested feature feeds the heads on the C4 path; earlier ones would add here
        live_names = [n for n in self.stage_names if n not in self.frozen]
        block_stage = [n for n in live_names for _ in getattr(self, n)]
        stage_last = {n: max(i for i, s in enumerate(block_stage) if s == n) for n in live_names}
        gmap = {n: g for n, g in zip(self._out_features, out_grads) if g is not None}
        mid_hook = getattr(self, "_mid_backward", None)
        mid_name = getattr(self, "_mid_stage_name", None)
        mid_first = min(i for i, s in enumerate(block_stage) if s == mid_name) if (mid_hook is not None and mid_name in block_stage) else -1
        pg = {}
        dx, masked = None, False
        for bi in range(len(blocks) - 1, -1, -1):
            blk = blocks[bi]
            name = block_stage[bi]
            if stage_last[name] == bi and name in gmap:
                assert not masked, "a stage output gradient must be added before the ReLU mask"
                g_in = gmap[name].permute(0, 2, 3, 1).to(self.act_dtype).contiguous()
                dx = g_in if dx is None else native.add_(dx, g_in)
            if dx is None:
                saved[bi] = None
                continue
            # the block below's output (its joining ReLU's mask) -- unless a stage gradient still has to be added there
            below = None
            if bi > 0 and saved[bi - 1] is not None and not (stage_last[block_stage[bi - 1]] == bi - 1
                                                              and block_stage[bi - 1] in gmap):
                below = saved[bi - 1][-1]
            dx, pg[bi], masked = self._block_backward(blk, saved[bi], dx, need_dx=bi > 0,   # first live block: frozen input
                                                      masked=masked, below_out=below)
            saved[bi] = None
            if mid_hook is not None and mid_first == bi:
                # every gradient of the last live stage is written (weights through the side stream: join it first, so that
                # the collective, ordered behind the main stream, sees them): GradientReducer.launch_mid
                if self._side is not None:
                    torch.cuda.current_stream().wait_stream(self._side)
                mid_hook()
        if self._side is not None:      # every weight gradient is in its buffer before anyone (all-reduce, SGD, autograd) reads it
            main = torch.cuda.current_stream()
            main.wait_stream(self._side)
            for dw in self._side_keep:      # a gradient tensor made on the side stream is consumed on the main one
                if dw is not None:
                    dw.record_stream(main)
            self._side, self._side_keep = None, []
        out = []
        for bi, blk in enumerate(blocks):
            if bi in pg:
                out += pg[bi]
            else:
                for c in blk.convs():
                    out += [None, None, None]
        return out


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, _=None):
    return ResNet(cfg)
