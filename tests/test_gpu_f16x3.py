"""GPU parity tests of the SFOD_F16X3 operand format (the forward products of ``SFOD.COMPUTE_DTYPE = "f16x3"``).

The same split-precision product as SFOD_BF16X3 -- hi*hi + hi*lo + lo*hi, three MFMAs, fp32 accumulation -- on IEEE
half pairs: hi = f16(v), lo = f16(v - hi).  22 significand bits per operand for |v| in [2^-3, 65504], an absolute 2^-25
below (half's subnormals, which v_mfma_f32_32x32x16_f16 keeps: tools/experiments/mfma_f16_vs_bf16.hip), saturation at
+-65504 above.  Activations and weights live inside that window, so every forward kernel must agree with an
fp64-accumulated reference on the UNROUNDED fp32 operands about as well as the fp32 MFMA path does: the gate is 4e-6
relative L2 (fp32 MFMA on the same data: ~1e-6; SFOD_BF16X3's gate: 3e-5).  Gradients do not fit the window: the
weight-gradient entry points reject the format (checked below) and the mode's backward products use bf16 pairs.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-6


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def splith_ref(x):
    """fp32 [..., C] -> fp16 [..., 2C] in the documented layout (per 8 channels: 8 hi then 8 lo), saturating."""
    c = x.clamp(-65504.0, 65504.0)             # saturation: exactly +-65504 (lo = 0) beyond half's range, infinities included
    hi = torch.where(torch.isnan(x), x.half(), c.half())
    r = c - hi.float()
    lo = r.half()                              # NaN stays NaN
    s = x.shape
    hi = hi.reshape(*s[:-1], s[-1] // 8, 1, 8)
    lo = lo.reshape(*s[:-1], s[-1] // 8, 1, 8)
    return torch.cat([hi, lo], dim=-2).reshape(*s[:-1], 2 * s[-1])


def to_pairs(native, x_dev):
    return native.cast(x_dev.contiguous(), native.SPLITH_DTYPE)


def conv_ref64(x, w, bias=None, padding=0):
    return F.conv2d(x.double(), w.double(), None if bias is None else bias.double(), padding=padding)


def test_cast_layout_range_and_round_trip(native):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 5, 64, generator=g) * torch.logspace(-6, 3, 64)
    x[0, 0, :8] = torch.tensor([0.0, -0.0, 1.0, 1e-30, -3e38, 65504.0, 2.0 ** -20, 1 + 2.0 ** -12])
    x[0, 1, :4] = torch.tensor([float("nan"), 70000.0, -131000.0, 65519.9])
    s = to_pairs(native, x.to(DEV))
    assert s.dtype == native.SPLITH_DTYPE and s.shape == x.shape
    got = s.view(torch.float16).cpu()
    ref = splith_ref(x)
    nan = torch.isnan(ref.float())
    assert torch.equal(torch.isnan(got.float()), nan)
    assert torch.equal(got.view(torch.int16)[~nan], ref.view(torch.int16)[~nan])
    back = native.cast(s, torch.float32).cpu()
    ok = ~torch.isnan(x)
    inr = ok & (x.abs() <= 65504.0)
    # inside the window: 2^-22 relative or 2^-25 absolute (half's subnormal spacing is 2^-24)
    assert ((back - x).abs()[inr] <= torch.maximum(x.abs()[inr] * 2.0 ** -22, torch.tensor(2.0 ** -25))).all()
    expect = ref.float().reshape(37, 5, 8, 2, 8).sum(-2).reshape(37, 5, 64)
    assert torch.equal(back[ok], expect[ok])
    assert back[0, 0, 4] == -65504.0 and back[0, 1, 1] == 65504.0 and back[0, 1, 2] == -65504.0    # beyond half's range: saturated
    assert torch.isnan(back[0, 1, 0])
    assert torch.equal(native.cast(native.cast(back.to(DEV), native.SPLITH_DTYPE), torch.float32).cpu()[ok], back[ok])
    # pairs -> pairs: the bf16 split of hi + lo, as if converted through fp32
    xs = x.clone()
    xs[0, 1, 0] = 0.0
    a = native.cast(to_pairs(native, xs.to(DEV)), native.SPLIT_DTYPE)
    b = native.cast(native.cast(to_pairs(native, xs.to(DEV)), torch.float32), native.SPLIT_DTYPE)
    assert a.dtype == native.SPLIT_DTYPE and torch.equal(a.view(torch.bfloat16), b.view(torch.bfloat16))
    # one pass, both formats
    h, bb = native.operands_for(xs.to(DEV), native.SPLITH_DTYPE, need_grad=True)
    assert h.dtype == native.SPLITH_DTYPE and bb.dtype == native.SPLIT_DTYPE
    assert torch.equal(h.view(torch.float16), to_pairs(native, xs.to(DEV)).view(torch.float16))
    assert torch.equal(bb.view(torch.bfloat16), native.cast(xs.to(DEV), native.SPLIT_DTYPE).view(torch.bfloat16))
    only, none = native.operands_for(xs.to(DEV), native.SPLITH_DTYPE, need_grad=False)
    assert none is None and only.dtype == native.SPLITH_DTYPE


@pytest.mark.parametrize("shape", [
    # B, H, W, Cin, Cout, ks
    (2, 9, 13, 64, 64, 3),
    (1, 18, 37, 64, 128, 3),
    (2, 7, 5, 128, 256, 3),
    (1, 5, 6, 512, 512, 3),
    (3, 16, 16, 3, 64, 3),      # first layer: 3 channels in one 8-channel group
    (1, 18, 37, 512, 75, 1),    # RPN 1x1 heads fused
    (2, 1, 1, 256, 1024, 1),
    (1, 10, 12, 24, 40, 3),
    (1, 38, 75, 1024, 256, 1),  # ResNet-101 res4 conv1
])
@pytest.mark.parametrize("act", [0, 1])
def test_conv_fwd_generic_kernel(native, shape, act):
    B, H, W, Cin, Cout, ks = shape
    if Cin == 24:
        ks = 1
    g = torch.Generator().manual_seed(hash(shape) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    bias = torch.randn(Cout, generator=g)
    cin_pad = (Cin + 7) // 8 * 8
    ref = conv_ref64(x, w, bias, padding=ks // 2)
    ref = F.relu(ref) if act else ref
    xd = torch.zeros(B, H, W, cin_pad, device=DEV)
    xd[..., :Cin] = nhwc(x).to(DEV)
    wp = native.pack_conv_weight(w.to(DEV), cin_pad, native.F16X3)
    assert wp.dtype == native.SPLITH_DTYPE
    try:
        native.set_conv_algo(1)
        y = native.conv_fwd(to_pairs(native, xd), wp, bias.to(DEV), Cout, ks, act=act)
        y2 = native.conv_fwd(xd, wp, bias.to(DEV), Cout, ks, act=act)     # fp32 input: converted by the wrapper
        wpb = native.pack_conv_weight(w.to(DEV), cin_pad, native.BF16X3)
        yb = native.conv_fwd(xd, wpb, bias.to(DEV), Cout, ks, act=act)
    finally:
        native.set_conv_algo(0)
    assert y.dtype == torch.float32
    e, eb = rel_err(nchw(y.cpu()), ref), rel_err(nchw(yb.cpu()), ref)
    print(f"[f16x3 conv {shape} act={act}] relative L2 vs fp64: f16x3 {e:.2e}, bf16x3 {eb:.2e}")
    assert e < TOL, (e, eb)
    assert torch.equal(y, y2)
    if Cin >= 64:
        assert e < 0.5 * eb, (e, eb)      # the point of the format: well below the bf16-pair error on the same data


@pytest.mark.parametrize("shape", [(12900, 264, 1000), (17000, 1032, 520), (4096, 25088, 1024)])
def test_linear_kernels_wide_and_tall(native, shape):
    """nn.Linear shapes that select the 256 x 256 tile (64-byte K stages) and the 256 x 64 long-K tile."""
    M, K, N = shape
    g = torch.Generator().manual_seed(M)
    x = torch.relu(torch.randn(M, K, generator=g))
    w = torch.randn(N, K, generator=g) * 0.01
    b = torch.randn(N, generator=g)
    rows = torch.randint(0, M, (64,), generator=g)
    ref = x[rows].double() @ w.double().t() + b.double()
    y = native.conv_fwd(x.to(DEV), native.pack_fc_weight(w.to(DEV), native.F16X3), b.to(DEV), N, 1)
    # long K: the fp32 accumulation's own error grows with sqrt(K) (fp32 MFMA at K = 25 088: 2e-6)
    assert rel_err(y[rows.to(DEV)].cpu(), ref) < TOL * max(1.0, math.sqrt(K / 4608.0))


@pytest.mark.parametrize("shape", [
    # B, H, W, Cin, Cout  -- halo-patch kernel on physical channels 2 * Cin
    (2, 37, 75, 64, 128),
    (1, 20, 50, 64, 64),
    (1, 33, 40, 128, 64),
    (2, 9, 13, 16, 200),
    (1, 70, 150, 48, 136),
    (3, 5, 6, 256, 256),
])
@pytest.mark.parametrize("variant", ["plain", "relu_stats", "ldy"])
@pytest.mark.parametrize("wg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])   # 7 / 8 / 9: the 64-channel tile shapes on 16x16x32 (round 6)
def test_conv3x3_patch_kernel(native, shape, variant, wg):
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g) + 0.3
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    ref = conv_ref64(x, w, bias, padding=1)
    xd = to_pairs(native, nhwc(x).to(DEV))
    wp = native.pack_conv_weight(w.to(DEV), Cin, native.F16X3)
    try:
        native.set_conv_algo(2)
        native.set_conv3x3_variant(wg)
        assert native.query("sfod_conv_fwd_algo", B, H, W, Cin, Cout, 3, native.F16X3) == 2
        if variant == "plain":
            y = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3)
            assert rel_err(nchw(y.cpu()), ref) < TOL
        elif variant == "relu_stats":
            y, stats = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3, act=1, want_stats=True)
            assert rel_err(nchw(y.cpu()), F.relu(ref)) < TOL
            rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
            mean, invstd = native.bn_finalize(stats, B * H * W, Cout, rm, rv, 0.1, 1e-5)
            torch.testing.assert_close(mean.cpu().double(), ref.mean(dim=(0, 2, 3)), rtol=1e-4, atol=2e-5)
            torch.testing.assert_close(invstd.cpu().double(), torch.rsqrt(ref.var(dim=(0, 2, 3), unbiased=False) + 1e-5),
                                       rtol=1e-4, atol=1e-5)
        else:
            y = native.conv_fwd(xd, wp, None, Cout, 3, ldy=Cout + 8)
            assert rel_err(nchw(y[..., :Cout].cpu()), ref - bias.double().view(1, -1, 1, 1)) < TOL
            assert (y[..., Cout:] == 0).all()
    finally:
        native.set_conv_algo(0)
        native.set_conv3x3_variant(0)


def test_backward_entry_points_reject_half_pairs(native):
    """Gradients do not fit half's exponent range: weight gradient and the fused data gradient take bf16 pairs only."""
    x = to_pairs(native, torch.randn(1, 8, 8, 64, device=DEV))
    dy = to_pairs(native, torch.randn(1, 8, 8, 64, device=DEV))
    with pytest.raises(native.NativeLibraryError):
        native.conv_wgrad(x, dy, 64, 3)
    assert native.query("sfod_conv_dgrad_bnred_blocks", 2, 37, 75, 64, 64, native.F16X3) == 0
    assert native.query("sfod_conv_wgrad_oihw_supported", 2, 37, 75, 64, 64, 3, 64, native.F16X3) == 0
    assert native.grad_dtype_of(native.SPLITH_DTYPE) == native.SPLIT_DTYPE
    assert native.grad_dtype_of(native.SPLIT_DTYPE) == native.SPLIT_DTYPE
    assert native.grad_dtype_of(torch.float32) == torch.float32


@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("hw", [(8, 12), (7, 9), (37, 75)])
def test_bn_apply_pair_outputs(native, pool, relu, hw):
    """BatchNorm apply writing half pairs == the fp32 kernel's output converted afterwards; the two-output form (half
    pairs + bf16 pairs from one pass) == the two single-output launches, bit for bit."""
    H, W = hw
    B, C = 2, 64
    g = torch.Generator().manual_seed(5)
    y = torch.randn(B, H, W, C, generator=g).to(DEV)
    mean, var = y.mean(dim=(0, 1, 2)), y.var(dim=(0, 1, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    gamma, beta = torch.rand(C, generator=g).to(DEV) + 0.5, torch.randn(C, generator=g).to(DEV) * 0.1
    z32 = native.bn_relu_pool_fwd(y, mean, invstd, gamma, beta, pool, relu=relu)
    zh = native.bn_relu_pool_fwd(y, mean, invstd, gamma, beta, pool, relu=relu, out_dtype=native.SPLITH_DTYPE)
    assert zh.dtype == native.SPLITH_DTYPE and zh.shape == z32.shape
    zf = native.cast(zh, torch.float32)
    assert ((zf - z32).abs() <= z32.abs() * 2.0 ** -21 + 1e-6).all()
    if relu:
        assert ((zf == 0) == (z32 == 0)).all()
    zb = native.bn_relu_pool_fwd(y, mean, invstd, gamma, beta, pool, relu=relu, out_dtype=native.SPLIT_DTYPE)
    z1, z2 = native.bn_relu_pool_fwd(y, mean, invstd, gamma, beta, pool, relu=relu, out_dtype=native.SPLITH_DTYPE,
                                     with_grad_operand=True)
    assert z1.dtype == native.SPLITH_DTYPE and z2.dtype == native.SPLIT_DTYPE
    assert torch.equal(z1.view(torch.float16), zh.view(torch.float16))
    assert torch.equal(z2.view(torch.bfloat16), zb.view(torch.bfloat16))
    same, same2 = native.bn_relu_pool_fwd(y, mean, invstd, gamma, beta, pool, relu=relu, out_dtype=native.SPLIT_DTYPE,
                                          with_grad_operand=True)
    assert same is same2 and torch.equal(same.view(torch.bfloat16), zb.view(torch.bfloat16))


def test_resnet_join_kernels_emit_half_pairs(native):
    g = torch.Generator().manual_seed(11)
    rows, C = 300, 256
    y, res = torch.randn(rows, C, generator=g).to(DEV), torch.randn(rows, C, generator=g).to(DEV)
    mean, invstd = torch.randn(C, generator=g).to(DEV) * 0.1, torch.rand(C, generator=g).to(DEV) + 0.5
    gamma, beta = torch.rand(C, generator=g).to(DEV) + 0.5, torch.randn(C, generator=g).to(DEV) * 0.1
    z = native.bn_add_relu_fwd(y, mean, invstd, gamma, beta, res)
    for dtype, view in ((native.SPLITH_DTYPE, torch.float16), (native.SPLIT_DTYPE, torch.bfloat16)):
        z2, zp = native.bn_add_relu_fwd(y, mean, invstd, gamma, beta, res, with_operand=dtype)
        assert zp.dtype == dtype
        torch.testing.assert_close(z2, z, rtol=2e-7, atol=1e-7)       # (the two instantiations may contract a*b+c differently)
        assert torch.equal(zp.view(view), native.cast(z2, dtype).view(view))      # the pairs: the exact split of that fp32 half
        o = native.add_act(y, res, 1)
        o2, op = native.add_act(y, res, 1, with_operand=dtype)
        assert torch.equal(o, o2) and torch.equal(op.view(view), native.cast(o, dtype).view(view))
        # three outputs from one pass: fp32 stream, forward operand, weight-gradient operand (bf16 pairs; the same tensor
        # when the forward operand already is one)
        z3, zp3, zg3 = native.bn_add_relu_fwd(y, mean, invstd, gamma, beta, res, with_operand=dtype, with_grad_operand=True)
        assert torch.equal(z3, z2) and torch.equal(zp3.view(view), zp.view(view)) and zg3.dtype == native.SPLIT_DTYPE
        assert torch.equal(zg3.view(torch.bfloat16), native.cast(z2, native.SPLIT_DTYPE).view(torch.bfloat16))
        assert (zg3 is zp3) == (dtype == native.SPLIT_DTYPE)
        o3, op3, og3 = native.add_act(y, res, 1, with_operand=dtype, with_grad_operand=True)
        assert torch.equal(o3, o) and torch.equal(op3.view(view), op.view(view))
        assert torch.equal(og3.view(torch.bfloat16), native.cast(o, native.SPLIT_DTYPE).view(torch.bfloat16))


def test_roi_align_preprocess_packers_im2col(native):
    g = torch.Generator().manual_seed(3)
    B, H, W, C, R = 2, 19, 38, 64, 200
    feat = torch.randn(B, H, W, C, generator=g).to(DEV)
    xy = torch.rand(R, 2, generator=g) * torch.tensor([W * 32.0, H * 32.0])
    wh = torch.rand(R, 2, generator=g) * 300 + 4
    rois = torch.cat([torch.randint(0, B, (R, 1), generator=g).float(), xy, xy + wh], dim=1)
    rois[7, 0] = -1
    rois = rois.to(DEV)
    fs = to_pairs(native, feat)
    ref = native.roi_align_fwd(native.cast(fs, torch.float32), rois, 7, 1.0 / 32)
    out = native.roi_align_fwd(fs, rois, 7, 1.0 / 32)
    assert out.dtype == native.SPLITH_DTYPE and out.shape == ref.shape
    got = native.cast(out, torch.float32)
    assert (got[7] == 0).all()
    assert rel_err(got.cpu(), ref.cpu()) < 2.0 ** -21
    # preprocess
    imgs = [torch.randint(0, 256, (3, 20, 31), dtype=torch.uint8, generator=g).to(DEV),
            torch.randint(0, 256, (3, 17, 40), dtype=torch.uint8, generator=g).to(DEV)]
    mean, std = [103.53, 116.28, 123.675], [1.0, 57.0, 2.5]
    x32, _ = native.preprocess(imgs, 20, 40, 8, mean, std, native.F32)
    xs, _ = native.preprocess(imgs, 20, 40, 8, mean, std, native.F16X3)
    assert xs.dtype == native.SPLITH_DTYPE and xs.shape == (2, 20, 40, 8)
    assert torch.equal(xs.view(torch.float16), to_pairs(native, x32).view(torch.float16))
    # weight packers: w * s, s = the power of two that puts max|w| * s into [2^13, 2^14), max|w| published beside it
    def scaled_pairs(p32, src):
        amax = src.abs().max()
        s = 2.0 ** (13 - math.floor(math.log2(amax.item())))
        assert 2.0 ** 13 <= amax.item() * s < 2.0 ** 14
        return to_pairs(native, (p32 * s).contiguous()).view(torch.float16), amax
    w = torch.randn(40, 24, 3, 3, generator=g).to(DEV) * 0.03
    for rot in (False, True):
        inner = 40 if rot else 24
        p32 = native.pack_conv_weight(w, inner, native.F32, rot180=rot)
        ps = native.pack_conv_weight(w, inner, native.F16X3, rot180=rot)
        ref, amax = scaled_pairs(p32, w)
        assert torch.equal(ps.view(torch.float16), ref)
        assert ps.wscale.dtype == torch.int32 and ps.wscale.view(torch.float32).item() == amax.item()
    w2 = torch.randn(64, 40, 1, 1, generator=g).to(DEV) * 3.0
    specs = [(w, 24, False), (w2, 40, False), (w, 40, True)]
    pk32 = native.ConvWeightPacker(specs, native.F32).pack()
    pks = native.ConvWeightPacker(specs, native.F16X3).pack()
    for (src, _, _), a, b in zip(specs, pk32, pks):
        ref, amax = scaled_pairs(a.contiguous(), src)
        assert torch.equal(b.view(torch.float16), ref) and b.wscale.view(torch.float32).item() == amax.item()
    fc = torch.randn(1024, 512 * 49, generator=g).to(DEV) * 0.01
    for tr in (False, True):
        a = native.pack_fc_weight(fc, native.F32, chw_c=512, transpose=tr)
        b = native.pack_fc_weight(fc, native.F16X3, chw_c=512, transpose=tr)
        assert torch.equal(b.view(torch.float16), scaled_pairs(a, fc)[0])
    small = torch.randn(41, 1024, generator=g).to(DEV)
    a = native.pack_fc_weight(small, native.F32, transpose=True, ld=48)
    b = native.pack_fc_weight(small, native.F16X3, transpose=True, ld=48)
    assert torch.equal(b.view(torch.float16), scaled_pairs(a, small)[0])
    zero = native.pack_fc_weight(torch.zeros(16, 64, device=DEV), native.F16X3)         # all-zero weights: unscaled
    assert zero.wscale.item() == 0 and (zero.view(torch.float16) == 0).all()
    x = torch.randn(33, 64, device=DEV)
    assert (native.conv_fwd(x, zero, None, 16, 1) == 0).all()
    with pytest.raises(native.NativeLibraryError):       # a view drops the scale word: refused, not silently mis-scaled
        native.conv_fwd(x, b.view(b.shape), None, 16, 1)
    # stem im2col
    x = torch.zeros(2, 33, 47, 4, device=DEV)
    x[..., :3] = torch.randn(2, 33, 47, 3, generator=g).to(DEV)
    c32 = native.im2col_stem(x, 192)
    ch = native.im2col_stem(x, 192, out_dtype=native.SPLITH_DTYPE)
    assert torch.equal(ch.view(torch.float16), to_pairs(native, c32).view(torch.float16))


@pytest.mark.parametrize("hw", [(50, 70), (64, 96), (9, 500)])
def test_conv_first_layer_kernel(native, hw):
    H, W = hw
    B, Cout = 2, 64
    g = torch.Generator().manual_seed(H * W)
    x = torch.zeros(B, H, W, 8)
    x[..., :3] = torch.randn(B, H, W, 3, generator=g) * 50
    w = torch.randn(Cout, 3, 3, 3, generator=g) / 5
    bias = torch.randn(Cout, generator=g)
    ref = conv_ref64(nchw(x)[:, :3], w, bias, padding=1)
    xd = to_pairs(native, x.to(DEV))
    wp = native.pack_conv_weight(w.to(DEV), 8, native.F16X3)
    assert native.conv_first_supported(xd, 64)
    assert native.query("sfod_conv_fwd_algo", B, H, W, 8, Cout, 3, native.F16X3) == 3
    y, st = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3, want_stats=True)
    assert y.dtype == torch.float32
    assert rel_err(nchw(y.cpu()), ref) < TOL
    st2 = native.conv_first_stats(xd, wp, bias.to(DEV))
    assert st.nblk == st2.nblk and torch.equal(st, st2)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    mean, invstd = native.bn_finalize(st, B * H * W, Cout, rm, rv, 0.1, 1e-5)
    try:
        native.set_conv_algo(1)
        y_gen = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3)
    finally:
        native.set_conv_algo(0)
    assert rel_err(y.cpu(), y_gen.cpu()) < TOL
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    scale = gamma.to(DEV) * invstd
    shift = beta.to(DEV) - mean * scale
    z = native.conv_first_apply(xd, wp, bias.to(DEV), scale, shift, relu=True)
    assert z.dtype == native.SPLITH_DTYPE
    zref = F.relu(F.batch_norm(ref, None, None, gamma.double(), beta.double(), True, 0.1, 1e-5))
    zf = native.cast(z, torch.float32).cpu()
    assert rel_err(nchw(zf), zref) < TOL
    z2 = native.bn_relu_pool_fwd(y, mean, invstd, gamma.to(DEV), beta.to(DEV), False, out_dtype=native.SPLITH_DTYPE)
    assert rel_err(zf, native.cast(z2, torch.float32).cpu()) < 1e-6


def test_values_beyond_half_range_are_reported(native):
    """A value that had to be clamped at +-65504 (finite or infinite) raises the library's flag; the trainer polls it at its
    metrics period (native.check_f16x3_range) -- saturation is loud, not silent.  NaN stays NaN (the finite checks' business)."""
    dev = torch.device(DEV)
    try:
        native.check_f16x3_range(dev)          # whatever earlier tests left behind
    except FloatingPointError:
        pass
    x = torch.randn(64, 64, device=DEV) * 100.0
    to_pairs(native, x)
    native.pack_fc_weight(x, native.F16X3)
    y = torch.randn(2, 8, 8, 64, device=DEV)
    m, i = y.mean(dim=(0, 1, 2)), torch.rsqrt(y.var(dim=(0, 1, 2)) + 1e-5)
    native.bn_relu_pool_fwd(y, m, i, torch.ones(64, device=DEV), torch.zeros(64, device=DEV), False, out_dtype=native.SPLITH_DTYPE)
    native.check_f16x3_range(dev)              # in range: silent
    x[4, 5] = float("nan")
    to_pairs(native, x)
    native.check_f16x3_range(dev)              # NaN is not a range error (it stays NaN in the pairs)
    x[3, 5] = float("inf")
    to_pairs(native, x)
    with pytest.raises(FloatingPointError):    # an infinity is clamped like any other value beyond the range: reported
        native.check_f16x3_range(dev)
    x[3, 5], x[4, 5] = 70000.0, 0.0
    to_pairs(native, x)
    with pytest.raises(FloatingPointError):
        native.check_f16x3_range(dev)
    native.check_f16x3_range(dev)              # reported once, then clear
    feat = torch.full((1, 8, 8, 64), 1.0e5, device=DEV)        # the ROIAlign producer (its own translation unit)
    rois = torch.tensor([[0.0, 0.0, 0.0, 100.0, 100.0]], device=DEV)
    fp = native.cast(feat, native.SPLITH_DTYPE)
    with pytest.raises(FloatingPointError):    # the conversion itself saturates 1e5
        native.check_f16x3_range(dev)
    native.roi_align_fwd(fp, rois, 7, 1.0 / 32)                 # pooled values are convex combinations of saturated (in-range)
    native.check_f16x3_range(dev)                               # inputs: nothing left to clamp in that kernel, silent


@pytest.mark.parametrize("fmt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(700, 256, 96), (513, 1024, 264), (300, 4608, 64)])
def test_kernels_compute_the_product_they_are_defined_to_compute(native, fmt, shape):
    """The device result against oracle/split_precision.py -- the definition of the mode (hi*hi + hi*lo + lo*hi on the
    rounded pairs, weights under their power-of-two scale), accumulated in fp64: what is left is the fp32 accumulation
    order (1.1e-8 * sqrt(K)) -- for bf16 pairs an order of magnitude below the mode's own distance from the exact product (4.5e-6),
    for half pairs the whole of it (the definition itself sits 8e-8 from exact)."""
    from oracle import split_precision as sp
    M, K, N = shape
    g = torch.Generator().manual_seed(M + K)
    x = torch.relu(torch.randn(M, K, generator=g)) + 0.01 * torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * (2.0 / K) ** 0.5
    dt = native.F16X3 if fmt == "f16" else native.BF16X3
    y = native.conv_fwd(x.to(DEV), native.pack_fc_weight(w.to(DEV), dt), None, N, 1).cpu()
    defined = sp.linear(x, w, fmt)
    exact = x.double() @ w.double().t()
    e_def, e_exact = rel_err(y, defined), rel_err(y, exact)
    print(f"[{fmt} pairs, M={M} K={K} N={N}] device vs its definition {e_def:.2e}, vs the exact product {e_exact:.2e}, "
          f"definition vs exact {rel_err(defined, exact):.2e}")
    assert e_def < 2.5e-7 * math.sqrt(K / 256.0), (e_def, e_exact)      # fp32 accumulation: ~1.1e-8 * sqrt(K) (seen 1.8e-7 / 3.5e-7 / 7.7e-7)
    if fmt == "bf16":
        assert e_def < 0.2 * e_exact        # the bf16-pair mode's error IS its definition's, not the kernel's
