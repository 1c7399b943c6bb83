"""Domain discriminators of the reference (``/root/reference/daod/modeling/dann/dann.py``).

``FCDiscriminator_img`` ``:10-29`` (4 x conv3x3, LeakyReLU 0.2), ``GradientScalarLayer`` /
``gradient_scalar`` ``:33-51`` and ``DAInsHead`` ``:97-155`` are constructed by the hot config's
meta-architecture (``source_free_adaptive_teacher_rcnn.py:68-71``), so their parameters are part
of the state dict (checkpoints, EMA key matching, weight decay).  In the hot yaml their losses are
weighted by zero (``DOMAIN_CLASSIFIER.IMAGE/INSTANCE: False``); with SFOD.ELIDE_DEAD_BRANCHES the
domain branch is not executed.  Forward passes run on the HIP conv / GEMM kernels; the image-level
discriminator also has a hand-written backward (``dc_img_loss``: BCE-with-logits against a constant
domain label, ``source_free_adaptive_teacher_rcnn.py:145-155``), so ``DOMAIN_CLASSIFIER.IMAGE: True``
trains; the instance-level head has one too (``dc_ins_loss``: ROIAlign -> box head -> GRL -> DAInsHead with
dropout -> BCE, ``source_free_adaptive_teacher_rcnn.py:157-201,341-349``), gradients reach the box head and,
through ROIAlign, the backbone.
"""
import torch
import torch.nn as nn

from .. import native


class GradientScalarLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha=1.0):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.clone() * ctx.alpha, None


def gradient_scalar(x, alpha=1.0):
    return GradientScalarLayer.apply(x, alpha)


class FCDiscriminator_img(nn.Module):
    def __init__(self, in_channels, ndf1=256, ndf2=128, compute_dtype=torch.float32):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, ndf1, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(ndf1, ndf2, kernel_size=3, padding=1)
        self.conv3 = nn.Conv2d(ndf2, ndf2, kernel_size=3, padding=1)
        self.classifier = nn.Conv2d(ndf2, 1, kernel_size=3, padding=1)
        self.compute_dtype = compute_dtype

    @torch.no_grad()
    def forward(self, x):
        """x: [N,C,H,W] (NCHW logical) -> [N,1,H,W] fp32 logits.  Forward only (see module doc)."""
        dtype = self.compute_dtype
        dt = native.dt_of_dtype(dtype)
        h = native.nhwc_operand(x, dtype)
        for conv, act in ((self.conv1, 2), (self.conv2, 2), (self.conv3, 2)):
            wp = native.pack_conv_weight(conv.weight, h.shape[-1], dt)
            h = native.conv_fwd(h, wp, conv.bias, conv.out_channels, 3, act=act)
        wp = native.pack_conv_weight(self.classifier.weight, h.shape[-1], dt)
        y = native.conv_fwd(h, wp, self.classifier.bias, 1, 3, out_dtype=torch.float32, ldy=8)
        return y[..., :1].permute(0, 3, 1, 2)


class _DCImgLossFn(torch.autograd.Function):
    """features -> mean BCE-with-logits of the 4-conv discriminator against a constant label.
    backward: d logits -> classifier / conv3 / conv2 / conv1 (wgrad + bias grad + dgrad, LeakyReLU)."""

    @staticmethod
    def forward(ctx, dc, feat_nchw, label, *params):
        dtype = dc.compute_dtype
        dt = native.dt_of_dtype(dtype)
        h = native.nhwc_operand(feat_nchw, dtype)
        acts = [h]
        for conv in (dc.conv1, dc.conv2, dc.conv3):
            wp = native.pack_conv_weight(conv.weight.detach(), h.shape[-1], dt)
            h = native.conv_fwd(h, wp, conv.bias.detach(), conv.out_channels, 3, act=2)
            acts.append(h)
        wp = native.pack_conv_weight(dc.classifier.weight.detach(), h.shape[-1], dt)
        z = native.conv_fwd(h, wp, dc.classifier.bias.detach(), 1, 3, out_dtype=torch.float32, ldy=8)
        logits = z[..., 0]
        # scalar glue on N*H*W logits: F.binary_cross_entropy_with_logits(z, label) (mean)
        loss = (torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-logits.abs()))).mean()
        ctx.dc, ctx.acts, ctx.logits, ctx.label = dc, acts, logits, float(label)
        ctx.feat_dtype = feat_nchw.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        dc, acts, logits = ctx.dc, ctx.acts, ctx.logits
        dtype = native.grad_dtype_of(dc.compute_dtype)      # operands of the backward products ("f16x3": bf16 pairs)
        dt = native.dt_of_dtype(dtype)
        E = native.chunk_elems(dt)
        adt = native.out_dtype_of(dtype)       # dtype gradients flow in (fp32 in bf16x3 mode: converted at the MFMA)
        B, H, W = logits.shape
        dz = torch.zeros(B, H, W, E, dtype=adt, device=logits.device)     # 1 channel padded to a chunk
        dz[..., 0] = ((torch.sigmoid(logits) - ctx.label) * (g / logits.numel())).to(adt)
        pgrads = []
        dy = dz
        convs = [dc.conv1, dc.conv2, dc.conv3, dc.classifier]
        for li in range(3, -1, -1):
            conv, x_in = convs[li], acts[li]
            cout = conv.out_channels
            if li < 3:
                dy = native.act_bwd_(dy, acts[li + 1], 2)              # LeakyReLU(0.2) of this layer's output
            dwp = native.conv_wgrad(x_in, dy, cout, 3, operand=dtype)
            dw = torch.empty_like(conv.weight)
            native.unpack_conv_wgrad(dwp, dw)
            db = native.bias_grad(dy, cout)
            pgrads = [dw, db] + pgrads
            wr = native.pack_conv_weight(conv.weight.detach(), dy.shape[-1], dt, rot180=True)
            dy = native.conv_fwd(dy, wr, None, conv.in_channels, 3)
        dfeat = dy.permute(0, 3, 1, 2).to(native.out_dtype_of(ctx.feat_dtype))
        ctx.acts = None
        return (None, dfeat, None) + tuple(pgrads)


def dc_img_loss(dc, features_nchw, domain_label):
    """loss_DC_img_{s,t} of rcnn.py:145-155: GRL(-1) -> FCDiscriminator_img -> BCE-with-logits(mean)."""
    rev = gradient_scalar(features_nchw, -1.0)
    params = [dc.conv1.weight, dc.conv1.bias, dc.conv2.weight, dc.conv2.bias, dc.conv3.weight, dc.conv3.bias,
              dc.classifier.weight, dc.classifier.bias]
    return _DCImgLossFn.apply(dc, rev, float(domain_label), *params)


class DAInsHead(nn.Module):
    def __init__(self, in_channels, levels, compute_dtype=torch.float32):
        super().__init__()
        self.da_ins_fc1_layers, self.da_ins_fc2_layers, self.da_ins_fc3_layers = [], [], []
        for level in levels:
            names = ["da_ins_fc{}_level_{}".format(k, level) for k in (1, 2, 3)]
            mods = [nn.Linear(in_channels, 1024), nn.Linear(1024, 1024), nn.Linear(1024, 1)]
            for m in mods:
                nn.init.normal_(m.weight, std=0.01)
                nn.init.constant_(m.bias, 0)
            for n, m in zip(names, mods):
                self.add_module(n, m)
            self.da_ins_fc1_layers.append(names[0])
            self.da_ins_fc2_layers.append(names[1])
            self.da_ins_fc3_layers.append(names[2])
        self.compute_dtype = compute_dtype

    @torch.no_grad()
    def forward(self, x, levels=None):
        """Eval-mode forward (dropout off) on the GEMM kernel; single level."""
        assert len(self.da_ins_fc1_layers) == 1
        dtype = self.compute_dtype
        dt = native.dt_of_dtype(dtype)
        fc1 = getattr(self, self.da_ins_fc1_layers[0])
        fc2 = getattr(self, self.da_ins_fc2_layers[0])
        fc3 = getattr(self, self.da_ins_fc3_layers[0])
        h = native.as_operand(x.contiguous(), dtype)
        h = native.conv_fwd(h, native.pack_fc_weight(fc1.weight, dt), fc1.bias, 1024, 1, act=1)
        h = native.conv_fwd(h, native.pack_fc_weight(fc2.weight, dt), fc2.bias, 1024, 1, act=1)
        y = native.conv_fwd(h, native.pack_fc_weight(fc3.weight, dt), fc3.bias, 1, 1, out_dtype=torch.float32, ldy=8)
        return y[:, :1]


class _DCInsLossFn(torch.autograd.Function):
    """feature map (+ sampled rois) -> mean BCE-with-logits of the instance-level discriminator on the box
    head's features behind a gradient-reversal layer (``instance_dc_loss``, rcnn.py:341-349).  The mean runs
    over the live rois (padding rows of the fixed-capacity roi array carry batch index -1).  ``masks``: the two
    dropout masks (uint8, [R,1024]) or None in eval mode."""

    @staticmethod
    def forward(ctx, heads, head, feat_nchw, rois, label, masks, *params):
        dtype = heads.compute_dtype
        dt = native.dt_of_dtype(dtype)
        st = heads._box_forward(feat_nchw, rois)
        fc1 = getattr(head, head.da_ins_fc1_layers[0])
        fc2 = getattr(head, head.da_ins_fc2_layers[0])
        fc3 = getattr(head, head.da_ins_fc3_layers[0])
        r1 = native.conv_fwd(st["h2"], native.pack_fc_weight(fc1.weight.detach(), dt), fc1.bias.detach(), 1024, 1, act=1)
        z1 = r1
        if masks is not None:
            z1 = native.mul_mask_(r1.clone(), masks[0], 2.0)
        r2 = native.conv_fwd(z1, native.pack_fc_weight(fc2.weight.detach(), dt), fc2.bias.detach(), 1024, 1, act=1)
        z2 = r2
        if masks is not None:
            z2 = native.mul_mask_(r2.clone(), masks[1], 2.0)
        y = native.conv_fwd(z2, native.pack_fc_weight(fc3.weight.detach(), dt), fc3.bias.detach(), 1, 1,
                            out_dtype=torch.float32, ldy=8)
        z = y[:, 0]
        live = (rois[:, 0] >= 0).float()
        n_live = live.sum().clamp(min=1)
        # scalar glue on R logits: F.binary_cross_entropy_with_logits(z, label) over the live rows
        bce = torch.clamp(z, min=0) - z * label + torch.log1p(torch.exp(-z.abs()))
        loss = (bce * live).sum() / n_live
        ctx.heads, ctx.head, ctx.st, ctx.rois, ctx.masks = heads, head, st, rois, masks
        ctx.acts = (r1, z1, r2, z2, z, live, n_live)
        ctx.label = float(label)
        return loss

    @staticmethod
    def backward(ctx, g):
        heads, head, st, rois, masks = ctx.heads, ctx.head, ctx.st, ctx.rois, ctx.masks
        r1, z1, r2, z2, z, live, n_live = ctx.acts
        dtype = native.grad_dtype_of(heads.compute_dtype)
        dt = native.dt_of_dtype(dtype)
        E = native.chunk_elems(dt)
        fc1 = getattr(head, head.da_ins_fc1_layers[0])
        fc2 = getattr(head, head.da_ins_fc2_layers[0])
        fc3 = getattr(head, head.da_ins_fc3_layers[0])
        R = z.shape[0]
        adt = native.out_dtype_of(dtype)
        dz = torch.zeros(R, E, dtype=adt, device=z.device)              # 1 logit padded to a chunk
        dz[:, 0] = ((torch.sigmoid(z) - ctx.label) * live * (g / n_live)).to(adt)
        # fc3
        dw3 = native.conv_wgrad(z2, dz, 1, 1, operand=dtype).view(1, -1)
        db3 = native.bias_grad(dz, 1)
        d2 = native.conv_fwd(dz, native.pack_fc_weight(fc3.weight.detach(), dt, transpose=True, ld=E), None, 1024, 1)
        if masks is not None:
            native.mul_mask_(d2, masks[1], 2.0)
        native.act_bwd_(d2, r2, 1)
        # fc2
        dw2 = native.conv_wgrad(z1, d2, 1024, 1, operand=dtype).view(1024, -1)
        db2 = native.bias_grad(d2, 1024)
        d1 = native.conv_fwd(d2, native.pack_fc_weight(fc2.weight.detach(), dt, transpose=True), None, 1024, 1)
        if masks is not None:
            native.mul_mask_(d1, masks[0], 2.0)
        native.act_bwd_(d1, r1, 1)
        # fc1
        dw1 = native.conv_wgrad(st["h2"], d1, 1024, 1, operand=dtype).view(1024, -1)
        db1 = native.bias_grad(d1, 1024)
        dh2 = native.conv_fwd(d1, native.pack_fc_weight(fc1.weight.detach(), dt, transpose=True), None,
                              fc1.in_features, 1)
        # gradient reversal (gradient_scalar(box_features, -1.0)), then the box head and ROIAlign
        dh2 = dh2 * -1.0
        dfeat, (bw1, bb1, bw2, bb2) = heads._box_head_backward(st, rois, dh2)
        ctx.st = ctx.acts = None
        return (None, None, dfeat, None, None, None, bw1, bb1, bw2, bb2, dw1, db1, dw2, db2, dw3, db3)


def dc_ins_loss(roi_heads, head, feat_nchw, rois, domain_label, training=True):
    """loss_DC_ins_{s,t}: box_features = box_head(ROIAlign(features, sampled proposals)) -> GRL(-1) -> DAInsHead
    (dropout p = 0.5 in training mode, masks drawn with torch's generator) -> BCE-with-logits (mean)."""
    assert len(head.da_ins_fc1_layers) == 1
    bh = roi_heads.box_head
    fc = [getattr(head, n[0]) for n in (head.da_ins_fc1_layers, head.da_ins_fc2_layers, head.da_ins_fc3_layers)]
    masks = None
    if training:
        R = rois.shape[0]
        masks = [(torch.rand(R, 1024, device=rois.device) >= 0.5).to(torch.uint8) for _ in range(2)]
    params = [bh.fc1.weight, bh.fc1.bias, bh.fc2.weight, bh.fc2.bias,
              fc[0].weight, fc[0].bias, fc[1].weight, fc[1].bias, fc[2].weight, fc[2].bias]
    return _DCInsLossFn.apply(roi_heads, head, feat_nchw, rois, float(domain_label), masks, *params)
