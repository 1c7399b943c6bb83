"""Domain discriminators of the reference (``/root/reference/daod/modeling/dann/dann.py``).

``FCDiscriminator_img`` ``:10-29`` (4 x conv3x3, LeakyReLU 0.2), ``GradientScalarLayer`` /
``gradient_scalar`` ``:33-51`` and ``DAInsHead`` ``:97-155`` are constructed by the hot config's
meta-architecture (``source_free_adaptive_teacher_rcnn.py:68-71``), so their parameters are part
of the state dict (checkpoints, EMA key matching, weight decay).  In the hot yaml their losses are
weighted by zero (``DOMAIN_CLASSIFIER.IMAGE/INSTANCE: False``); with SFOD.ELIDE_DEAD_BRANCHES the
domain branch is not executed.  Forward passes run on the HIP conv / GEMM kernels; the image-level
discriminator also has a hand-written backward (``dc_img_loss``: BCE-with-logits against a constant
domain label, ``source_free_adaptive_teacher_rcnn.py:145-155``), so ``DOMAIN_CLASSIFIER.IMAGE: True``
trains.  The instance-level head is forward-only (its loss is reported, not differentiated).
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
        dt = native.F32 if dtype == torch.float32 else native.BF16
        h = x.permute(0, 2, 3, 1).to(dtype).contiguous()
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
        dt = native.F32 if dtype == torch.float32 else native.BF16
        h = feat_nchw.permute(0, 2, 3, 1).to(dtype).contiguous()
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
        dtype = dc.compute_dtype
        dt = native.F32 if dtype == torch.float32 else native.BF16
        E = native.chunk_elems(dt)
        B, H, W = logits.shape
        dz = torch.zeros(B, H, W, E, dtype=dtype, device=logits.device)     # 1 channel padded to a chunk
        dz[..., 0] = ((torch.sigmoid(logits) - ctx.label) * (g / logits.numel())).to(dtype)
        pgrads = []
        dy = dz
        convs = [dc.conv1, dc.conv2, dc.conv3, dc.classifier]
        for li in range(3, -1, -1):
            conv, x_in = convs[li], acts[li]
            cout = conv.out_channels
            if li < 3:
                dy = native.act_bwd_(dy, acts[li + 1], 2)              # LeakyReLU(0.2) of this layer's output
            dwp = native.conv_wgrad(x_in, dy, cout, 3)
            dw = torch.empty_like(conv.weight)
            native.unpack_conv_wgrad(dwp, dw)
            db = native.bias_grad(dy, cout)
            pgrads = [dw, db] + pgrads
            wr = native.pack_conv_weight(conv.weight.detach(), dy.shape[-1], dt, rot180=True)
            dy = native.conv_fwd(dy, wr, None, conv.in_channels, 3)
        dfeat = dy.permute(0, 3, 1, 2).to(ctx.feat_dtype)
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
        dt = native.F32 if dtype == torch.float32 else native.BF16
        fc1 = getattr(self, self.da_ins_fc1_layers[0])
        fc2 = getattr(self, self.da_ins_fc2_layers[0])
        fc3 = getattr(self, self.da_ins_fc3_layers[0])
        h = x.to(dtype).contiguous()
        h = native.conv_fwd(h, native.pack_fc_weight(fc1.weight, dt), fc1.bias, 1024, 1, act=1)
        h = native.conv_fwd(h, native.pack_fc_weight(fc2.weight, dt), fc2.bias, 1024, 1, act=1)
        y = native.conv_fwd(h, native.pack_fc_weight(fc3.weight, dt), fc3.bias, 1, 1, out_dtype=torch.float32, ldy=8)
        return y[:, :1]
