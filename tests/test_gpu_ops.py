"""GPU parity tests: every HIP entry point (called through the C ABI) against the CPU oracle.

Integer / index outputs are compared bit-exactly; floating point within the stated tolerance
(fp32 parity mode: 1e-4 relative as in BASELINE.json north_star; bf16 mode against an oracle fed
bf16-rounded operands).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import box_ops as OB
from oracle import model as om
from oracle.roi_align import roi_align as oracle_roi_align

pytestmark = pytest.mark.gpu
DEV = "cuda"


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _dtypes():
    return [torch.float32, torch.bfloat16]


# -------------------------------------------------------------------------------------------------
# conv / linear
# -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", _dtypes())
@pytest.mark.parametrize("shape", [
    # B, H, W, Cin, Cout, ks
    (2, 9, 13, 64, 64, 3),
    (1, 18, 37, 64, 128, 3),
    (2, 7, 5, 128, 256, 3),
    (1, 5, 6, 512, 512, 3),
    (3, 16, 16, 3, 64, 3),      # first layer: Cin padded to one 16-byte chunk
    (1, 18, 37, 512, 75, 1),    # RPN 1x1 heads fused (15 logits + 60 deltas)
    (2, 1, 1, 256, 1024, 1),
])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv_fwd(native, dtype, shape, act):
    B, H, W, Cin, Cout, ks = shape
    if act == 2 and Cout != 64:
        pytest.skip("leaky relu only needs one shape")
    g = torch.Generator().manual_seed(hash(shape) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    bias = torch.randn(Cout, generator=g)
    dt = native.dt_of(torch.empty(0, dtype=dtype))
    E = native.chunk_elems(dt)
    cin_pad = (Cin + E - 1) // E * E
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x, w, bias, padding=ks // 2)
    ref = {0: ref, 1: F.relu(ref), 2: F.leaky_relu(ref, 0.2)}[act]
    xd = torch.zeros(B, H, W, cin_pad, dtype=dtype, device=DEV)
    xd[..., :Cin] = nhwc(x).to(DEV).to(dtype)
    wp = native.pack_conv_weight(w.to(DEV), cin_pad, dt)
    y = native.conv_fwd(xd, wp, bias.to(DEV), Cout, ks, act=act)
    torch.cuda.synchronize()
    got = nchw(y.float().cpu())
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert rel_err(got, ref) < tol


@pytest.mark.parametrize("shape", [
    # B, H, W, Cin, Cout  -- halo-patch kernel: tile overhang, N tails, both workgroup shapes
    (2, 37, 75, 64, 128),     # G=1, 2 body iterations
    (1, 20, 50, 64, 64),      # G=2 (64 output channels / workgroup), 1 body
    (1, 33, 40, 128, 64),     # G=2, 2 bodies
    (2, 9, 13, 32, 200),      # G=1, single 32-channel slice, Cout tail (200 = 128 + 72)
    (1, 70, 150, 96, 136),    # several tiles per image in x and y, 3 slices
    (3, 5, 6, 256, 256),      # tiny maps, 8 slices
])
@pytest.mark.parametrize("variant", ["plain", "relu_stats", "f32out"])
@pytest.mark.parametrize("wg", [0, 1, 2, 3, 4, 5, 6])   # workgroup shape: auto, 512x128, 256x128, 256x64, 512x64, 256x128 on 16x16x32 (8 / 4 waves; pairs only)
def test_conv3x3_patch_kernel(native, shape, variant, wg):
    """k_conv3x3_patch (forced) against F.conv2d on bf16-rounded operands and against the generic
    implicit-GEMM kernel; BatchNorm partial statistics through sfod_bn_finalize."""
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(B, Cin, H, W, generator=g) + 0.3).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).bfloat16().float()
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, bias, padding=1)
    xd = nhwc(x).to(DEV).bfloat16()
    wp = native.pack_conv_weight(w.to(DEV), Cin, native.BF16)
    try:
        native.set_conv_algo(2)
        native.set_conv3x3_variant(wg)
        assert native.query("sfod_conv_stats_blocks", B, H, W, Cin, Cout, 3, native.BF16) < (B * H * W + 127) // 128 + B * 64
        if variant == "plain":
            y = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3)
            assert rel_err(nchw(y.float().cpu()), ref) < 6e-3
            if Cin & (Cin - 1) == 0:  # the generic kernel needs a power-of-two channel count
                native.set_conv_algo(1)
                y_gen = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3)
                # same bf16 products, fp32 accumulation in a different order
                assert rel_err(y.float().cpu(), y_gen.float().cpu()) < 4e-3
        elif variant == "relu_stats":
            y, stats = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3, act=1, want_stats=True)
            assert rel_err(nchw(y.float().cpu()), F.relu(ref)) < 6e-3
            rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
            mean, invstd = native.bn_finalize(stats, B * H * W, Cout, rm, rv, 0.1, 1e-5)
            m_ref = ref.mean(dim=(0, 2, 3))
            v_ref = ref.var(dim=(0, 2, 3), unbiased=False)
            torch.testing.assert_close(mean.cpu(), m_ref, rtol=2e-3, atol=2e-3)
            torch.testing.assert_close(invstd.cpu(), torch.rsqrt(v_ref + 1e-5), rtol=3e-3, atol=1e-4)
        else:
            y = native.conv_fwd(xd, wp, None, Cout, 3, out_dtype=torch.float32, ldy=Cout + 8)
            assert y.shape[-1] == Cout + 8
            assert rel_err(nchw(y[..., :Cout].cpu()), ref - bias.view(1, -1, 1, 1)) < 2e-3
            assert (y[..., Cout:] == 0).all()
    finally:
        native.set_conv_algo(0)
        native.set_conv3x3_variant(0)


@pytest.mark.parametrize("wg", [1, 2, 3, 4])
def test_conv3x3_patch_under_load_is_deterministic(native, wg):
    """Full-chip launch (thousands of workgroups, two per CU for the small shapes): every workgroup shape must
    be run-to-run bit-identical and agree with the generic kernel.  Guards the DMA-vs-pending-ds_read hazard of
    the counted-wait pipeline (a stage's last fragment reads must have RETURNED before the barrier that lets
    other waves' DMA overwrite the ring slot), which only shows under load."""
    B, H, W, Cin, Cout = 8, 150, 300, 256, 256
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, H, W, Cin, device=DEV, generator=g).bfloat16()
    w = (torch.randn(Cout, 9, Cin, device=DEV, generator=g) / (3 * Cin ** 0.5)).bfloat16()
    bias = torch.randn(Cout, device=DEV, generator=g)
    try:
        native.set_conv_algo(1)
        ref = native.conv_fwd(x, w, bias, Cout, 3).float()
        native.set_conv_algo(2)
        native.set_conv3x3_variant(wg)
        ys = [native.conv_fwd(x, w, bias, Cout, 3, want_stats=True)[0] for _ in range(6)]
        torch.cuda.synchronize()
    finally:
        native.set_conv_algo(0)
        native.set_conv3x3_variant(0)
    for y in ys[1:]:
        assert torch.equal(ys[0], y), "patch conv is not run-to-run deterministic"
    # same bf16 products, fp32 accumulation in another order: a few outputs differ by one bf16 ulp
    assert rel_err(ys[0].float(), ref) < 2e-4


@pytest.mark.parametrize("hw", [(50, 70), (64, 96), (9, 500)])
@pytest.mark.parametrize("stats", [False, True])
def test_conv_first_layer_kernel(native, hw, stats):
    """k_conv_first (3 real channels in one 8-wide chunk -> 64): image edges, tiles hanging over the
    right / bottom border, BatchNorm statistics; against F.conv2d and the generic kernel."""
    H, W = hw
    B, Cin, Cout = 2, 3, 64
    g = torch.Generator().manual_seed(H + W)
    x = (torch.randn(B, Cin, H, W, generator=g) * 50).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 5).bfloat16().float()
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, bias, padding=1)
    xd = torch.zeros(B, H, W, 8, dtype=torch.bfloat16, device=DEV)
    xd[..., :3] = nhwc(x).to(DEV).bfloat16()
    wp = native.pack_conv_weight(w.to(DEV), 8, native.BF16)
    assert native.query("sfod_conv_fwd_algo", B, H, W, 8, Cout, 3, native.BF16) == 3
    if stats:
        y, st = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3, want_stats=True)
        rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
        mean, invstd = native.bn_finalize(st, B * H * W, Cout, rm, rv, 0.1, 1e-5)
        torch.testing.assert_close(mean.cpu(), ref.mean(dim=(0, 2, 3)), rtol=2e-3, atol=2e-2)
        torch.testing.assert_close(invstd.cpu(), torch.rsqrt(ref.var(dim=(0, 2, 3), unbiased=False) + 1e-5), rtol=3e-3,
                                   atol=1e-5)
    else:
        y = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3)
    assert rel_err(nchw(y.float().cpu()), ref) < 6e-3
    try:
        native.set_conv_algo(1)
        y_gen = native.conv_fwd(xd, wp, bias.to(DEV), Cout, 3)
    finally:
        native.set_conv_algo(0)
    assert rel_err(y.float().cpu(), y_gen.float().cpu()) < 4e-3


@pytest.mark.parametrize("dtype", _dtypes())
def test_linear_big_k_and_ld_padding(native, dtype):
    """fc1-shaped GEMM (K = 25088, permuted (c,p)->(p,c)) and the 41-wide predictor with ld 48."""
    g = torch.Generator().manual_seed(5)
    R, C, PP, N = 70, 512, 49, 256
    K = C * PP
    pooled = torch.randn(R, C, 7, 7, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    if dtype == torch.bfloat16:
        pooled, w = pooled.bfloat16().float(), w.bfloat16().float()
    ref = F.relu(F.linear(pooled.flatten(1), w, b))
    dt = native.dt_of(torch.empty(0, dtype=dtype))
    xp = pooled.permute(0, 2, 3, 1).reshape(R, K).contiguous().to(DEV).to(dtype)  # (p, c) order
    wp = native.pack_fc_weight(w.to(DEV), dt, chw_c=C)
    y = native.conv_fwd(xp, wp, b.to(DEV), N, 1, act=1)
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert rel_err(y.float().cpu(), ref) < tol
    # predictor: N = 41 -> fp32 output with ld 48
    w2 = torch.randn(41, N, generator=g) / math.sqrt(N)
    b2 = torch.randn(41, generator=g)
    h = ref.to(DEV).to(dtype)
    if dtype == torch.bfloat16:
        w2 = w2.bfloat16().float()
    ref2 = F.linear(h.float().cpu(), w2, b2)
    wp2 = native.pack_fc_weight(w2.to(DEV), dt)
    y2 = native.conv_fwd(h, wp2, b2.to(DEV), 41, 1, out_dtype=torch.float32, ldy=48)
    assert y2.shape == (R, 48)
    assert rel_err(y2[:, :41].cpu(), ref2) < tol
    assert (y2[:, 41:] == 0).all()


@pytest.mark.parametrize("dtype", _dtypes())
@pytest.mark.parametrize("shape", [(2, 9, 13, 64, 64, 3), (1, 11, 7, 128, 256, 3), (1, 6, 5, 512, 75, 1),
                                   (3, 8, 8, 3, 64, 3)])
def test_conv_dgrad_and_wgrad(native, dtype, shape):
    B, H, W, Cin, Cout, ks = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    dy = torch.randn(B, Cout, H, W, generator=g)
    if dtype == torch.bfloat16:
        x, w, dy = x.bfloat16().float(), w.bfloat16().float(), dy.bfloat16().float()
    x.requires_grad_(True)
    w.requires_grad_(True)
    F.conv2d(x, w, None, padding=ks // 2).backward(dy)
    dt = native.dt_of(torch.empty(0, dtype=dtype))
    E = native.chunk_elems(dt)
    cin_pad = (Cin + E - 1) // E * E
    cout_pad = (Cout + E - 1) // E * E
    xd = torch.zeros(B, H, W, cin_pad, dtype=dtype, device=DEV)
    xd[..., :Cin] = nhwc(x.detach()).to(DEV).to(dtype)
    dyd = torch.zeros(B, H, W, cout_pad, dtype=dtype, device=DEV)
    dyd[..., :Cout] = nhwc(dy).to(DEV).to(dtype)
    tol = 3e-5 if dtype == torch.float32 else 8e-3
    # weight gradient
    dwp = native.conv_wgrad(xd, dyd, Cout, ks)
    dw = torch.empty(Cout, Cin, ks, ks, dtype=torch.float32, device=DEV)
    native.unpack_conv_wgrad(dwp, dw)
    assert rel_err(dw.cpu(), w.grad) < tol
    # data gradient = conv with the rotated / transposed weight
    if Cin >= E:
        wr = native.pack_conv_weight(w.detach().to(DEV), cout_pad, dt, rot180=True)
        dx = native.conv_fwd(dyd, wr, None, Cin, ks)
        assert rel_err(nchw(dx.float().cpu()), x.grad) < tol


@pytest.mark.parametrize("shape", [
    # B, H, W, Cin, Cout -- halo-patch weight gradient: tile overhang, channel tails, both variants
    (2, 37, 75, 64, 128),
    (1, 20, 50, 64, 64),      # CO=2 variant (Cout <= 64): two k-step groups, two slabs per split
    (1, 33, 40, 128, 64),
    (2, 9, 13, 32, 96),       # single 32-channel plane of x, Cout tail inside a 128 tile
    (1, 70, 150, 96, 160),    # several tiles per image, 2 co tiles, 2 ci tiles (second half empty)
    (3, 5, 6, 256, 256),
])
def test_conv3x3_patch_wgrad(native, shape):
    """k_wgrad3x3_patch + slab reduction (forced) against autograd on bf16-rounded operands."""
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    dy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float()
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    xd = nhwc(x).to(DEV).bfloat16()
    dyd = nhwc(dy).to(DEV).bfloat16()
    try:
        native.set_conv_algo(2)
        assert native.query("sfod_conv_wgrad_ws_bytes", B, H, W, Cin, Cout, 3, Cout, native.BF16) > 0
        dwp = native.conv_wgrad(xd, dyd, Cout, 3)
        dwp2 = native.conv_wgrad(xd, dyd, Cout, 3)
    finally:
        native.set_conv_algo(0)
    dw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=DEV)
    native.unpack_conv_wgrad(dwp, dw)
    assert rel_err(dw.cpu(), w.grad) < 2e-5 * math.sqrt(B * H * W) / 10 + 1e-4
    assert torch.equal(dwp, dwp2), "slab reduction must be deterministic"
    # the same gradient reduced straight into the state-dict layout (overwrite, then accumulate)
    try:
        native.set_conv_algo(2)
        assert native.conv_wgrad_oihw_supported(xd, dyd, Cout, 3)
        direct = torch.full((Cout, Cin, 3, 3), float("nan"), dtype=torch.float32, device=DEV)
        native.conv_wgrad_oihw(xd, dyd, direct, accumulate=False)
        acc = torch.ones(Cout, Cin, 3, 3, dtype=torch.float32, device=DEV)
        native.conv_wgrad_oihw(xd, dyd, acc, accumulate=True)
    finally:
        native.set_conv_algo(0)
    assert torch.equal(direct, dw), "OIHW reduction must equal packed reduction + unpack bit for bit"
    assert torch.allclose(acc, dw + 1.0, rtol=0, atol=1e-6 * float(dw.abs().max()) + 1e-6)


def test_bn_fused_accumulators(native):
    """sfod_bn_finalize bumps num_batches_tracked; sfod_bn_relu_pool_bwd adds dgamma / dbeta into the
    gradient accumulators it is given (what autograd's per-parameter accumulate would have done)."""
    B, H, W, C = 2, 9, 11, 64
    g = torch.Generator().manual_seed(5)
    y = torch.randn(B, H, W, C, generator=g).to(DEV)
    dz = torch.randn(B, H, W, C, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    wp = native.pack_conv_weight(torch.randn(C, C, 3, 3, generator=g).to(DEV) * 0.05, C, native.F32)
    yc, stats = native.conv_fwd(y, wp, None, C, 3, want_stats=True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.tensor(41, dtype=torch.int64, device=DEV)
    mean, invstd = native.bn_finalize(stats, B * H * W, C, rm, rv, 0.1, 1e-5, True, num_batches_tracked=nbt)
    assert int(nbt) == 42
    native.bn_finalize(stats, B * H * W, C, rm, rv, 0.1, 1e-5, False, num_batches_tracked=nbt)
    assert int(nbt) == 42, "no running-stat update -> no batch counted"
    _, dgam, dbet = native.bn_relu_pool_bwd(dz, yc, mean, invstd, gamma, beta, False)
    ga, ba = torch.full((C,), 2.0, device=DEV), torch.full((C,), -1.0, device=DEV)
    _, dgam2, dbet2 = native.bn_relu_pool_bwd(dz, yc, mean, invstd, gamma, beta, False, dgamma_acc=ga, dbeta_acc=ba)
    assert torch.equal(dgam, dgam2) and torch.equal(dbet, dbet2)
    assert torch.allclose(ga, dgam + 2.0, rtol=0, atol=1e-5) and torch.allclose(ba, dbet - 1.0, rtol=0, atol=1e-5)


@pytest.mark.parametrize("shape", [(1, 38, 75, 256), (2, 19, 37, 1024), (1, 75, 150, 128), (1, 9, 11, 64), (1, 120, 128, 72)])
def test_one_launch_batchnorm_finalize_is_the_two_launch_one_bit_for_bit(native, shape):
    """Layers of <= 64 statistics blocks are finalised in one launch (k_bn_finalize_one): mean, invstd, the running statistics
    after k momentum updates and the batch counter equal the two-launch form's bit for bit (the last shape has > 64 blocks:
    both settings take the two launches there)."""
    B, H, W, C = shape
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, H, W, 64, generator=g) * 3.0 + 0.7).to(DEV)
    wp = native.pack_conv_weight(torch.randn(C, 64, 1, 1, generator=g).to(DEV) * 0.2, 64, native.F32)
    _, stats = native.conv_fwd(x, wp, None, C, 1, want_stats=True)
    outs = []
    try:
        for fused in (True, False):
            native.set_bn_finalize_fused(fused)
            rm, rv = torch.linspace(-1, 1, C).to(DEV), (torch.linspace(0.5, 2, C)).to(DEV)
            nbt = torch.tensor(7, dtype=torch.int64, device=DEV)
            mean, invstd = native.bn_finalize(stats, B * H * W, C, rm, rv, 0.1, 1e-5, 3, num_batches_tracked=nbt)
            outs.append((mean, invstd, rm, rv, int(nbt)))
    finally:
        native.set_bn_finalize_fused(True)
    (m1, i1, rm1, rv1, n1), (m0, i0, rm0, rv0, n0) = outs
    assert torch.equal(m1, m0) and torch.equal(i1, i0) and torch.equal(rm1, rm0) and torch.equal(rv1, rv0) and n1 == n0 == 10
    assert torch.isfinite(m1).all() and (i1 > 0).all()
    if shape in ((1, 38, 75, 256), (2, 19, 37, 1024), (1, 9, 11, 64)):          # these are one-launch shapes; the others have > 64 blocks
        assert stats.nblk <= 64, stats.nblk


# -------------------------------------------------------------------------------------------------
# BatchNorm + ReLU + pool
# -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", _dtypes())
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("hw", [(8, 12), (7, 9), (37, 75)])
def test_conv_bn_relu_pool_block_fwd_bwd(native, dtype, pool, hw):
    H, W = hw
    B, Cin, C = 2, 64, 128
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(C, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(C, generator=g) * 0.1
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    gamma[3] = -0.7  # negative scale: max-pool must come after the affine + relu
    rm, rv = torch.zeros(C), torch.ones(C)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    with torch.no_grad():
        yc = F.conv2d(x, w, bias, padding=1)
        rm_ref, rv_ref = rm.clone(), rv.clone()
        z = F.relu(F.batch_norm(yc, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
        if pool:
            z = F.max_pool2d(z, 2, 2)
    dz = torch.randn(z.shape, generator=g)
    if dtype == torch.bfloat16:
        dz = dz.bfloat16().float()
    dt = native.dt_of(torch.empty(0, dtype=dtype))
    xd = nhwc(x).to(DEV).to(dtype)
    wp = native.pack_conv_weight(w.to(DEV), Cin, dt)
    y, stats = native.conv_fwd(xd, wp, bias.to(DEV), C, 3, want_stats=True)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    mean, invstd = native.bn_finalize(stats, B * H * W, C, rmd, rvd, 0.1, 1e-5)
    zd = native.bn_relu_pool_fwd(y, mean, invstd, gamma.to(DEV), beta.to(DEV), pool)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(nchw(zd.float().cpu()), z.detach()) < tol
    torch.testing.assert_close(rmd.cpu(), rm_ref, rtol=1e-4 if dtype == torch.float32 else 2e-2, atol=1e-5)
    torch.testing.assert_close(rvd.cpu(), rv_ref, rtol=1e-4 if dtype == torch.float32 else 2e-2, atol=1e-5)
    # backward of the BN+ReLU(+pool) block: dy w.r.t. the conv output, dgamma, dbeta
    # the reference backward starts from the conv output as STORED on the device (bf16-rounded in
    # throughput mode), so relu / arg-max decisions are taken on identical values
    yc2 = nchw(y.float().cpu()).requires_grad_(True)
    g2, b2 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z2 = F.relu(F.batch_norm(yc2, None, None, g2, b2, True, 0.1, 1e-5))
    if pool:
        z2 = F.max_pool2d(z2, 2, 2)
    z2.backward(dz)
    dzd = nhwc(dz).to(DEV).to(dtype)
    dy, dgamma, dbeta = native.bn_relu_pool_bwd(dzd, y, mean, invstd, gamma.to(DEV), beta.to(DEV), pool)
    btol = 1e-4 if dtype == torch.float32 else 4e-2
    assert rel_err(dgamma.cpu(), g2.grad) < btol
    assert rel_err(dbeta.cpu(), b2.grad) < btol
    assert rel_err(nchw(dy.float().cpu()), yc2.grad) < btol


# -------------------------------------------------------------------------------------------------
# ROIAlign
# -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", _dtypes())
def test_roi_align_fwd_bwd(native, dtype):
    g = torch.Generator().manual_seed(2)
    B, C, H, W = 2, 64, 18, 37
    feat = torch.randn(B, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        feat = feat.bfloat16().float()
    n = 40
    xy = torch.rand(n, 2, generator=g) * torch.tensor([1100.0, 500.0])
    wh = torch.rand(n, 2, generator=g) * torch.tensor([500.0, 300.0]) + 1
    rois = torch.cat([torch.randint(0, B, (n, 1), generator=g).float(), xy, xy + wh], 1)
    rois[5] = torch.tensor([1, -40.0, -30.0, 1300.0, 700.0])     # larger than the image
    rois[6] = torch.tensor([0, 100.0, 100.0, 100.5, 100.25])     # tiny
    rois[7, 0] = -1                                              # padding row
    fr = feat.clone().requires_grad_(True)
    live = rois[:, 0] >= 0
    ref = torch.zeros(n, C, 7, 7)
    ref[live] = oracle_roi_align(fr, rois[live], 7, 1 / 32.0, 0, True)
    dout = torch.randn(n, C, 7, 7, generator=g)
    if dtype == torch.bfloat16:
        dout = dout.bfloat16().float()
    ref.backward(dout)
    fd = nhwc(feat).to(DEV).to(dtype)
    out = native.roi_align_fwd(fd, rois.to(DEV), 7, 1 / 32.0)          # [R, 49, C]
    got = out.float().cpu().view(n, 7, 7, C).permute(0, 3, 1, 2)
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert rel_err(got, ref.detach()) < tol
    dd = dout.permute(0, 2, 3, 1).reshape(n, 49, C).contiguous().to(DEV).to(dtype)
    dfeat = native.roi_align_bwd(dd, rois.to(DEV), (B, H, W, C), 7, 1 / 32.0)
    assert rel_err(nchw(dfeat.cpu()), fr.grad) < (1e-5 if dtype == torch.float32 else 1e-4)


def test_roi_align_published_known_answer(native):
    """Detectron2's own ROIAlign vector (tests/layers/test_roi_align.py, aligned=True; quoted in tests/test_oracle_ops.py with
    its derivation) through ``sfod_roi_align_fwd`` with no oracle in between: 5 x 5 ramp, roi (1, 1, 3, 3), 4 x 4 bins, scale 1;
    channel c carries (c + 1) x the ramp."""
    C = 8
    ramp = torch.arange(25, dtype=torch.float32).view(1, 5, 5, 1) * torch.arange(1, C + 1, dtype=torch.float32)
    rois = torch.tensor([[0.0, 1.0, 1.0, 3.0, 3.0]])
    out = native.roi_align_fwd(ramp.contiguous().to(DEV), rois.to(DEV), 4, 1.0).cpu().view(4, 4, C)
    expected = torch.tensor([[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]])
    for c in range(C):
        torch.testing.assert_close(out[:, :, c], expected * (c + 1), rtol=0, atol=1e-5 * (c + 1))
    # the configs' 7 x 7 form (the separable kernel) on the same ramp: bin centres at 1 + (p + .5) * 2/7 - .5
    out7 = native.roi_align_fwd(ramp.contiguous().to(DEV), rois.to(DEV), 7, 1.0).cpu().view(7, 7, C)
    ctr = 0.5 + (torch.arange(7, dtype=torch.float32) + 0.5) * (2.0 / 7.0)
    torch.testing.assert_close(out7[:, :, 0], 5 * ctr.view(7, 1) + ctr.view(1, 7), rtol=0, atol=1e-5)


def test_roi_align_many_rois_unordered_images(native):
    """C = 512 (one 16-byte vector per lane), 2500 ROIs in random image order (the backward compacts
    its image's ROIs in segments of 1024) and padding rows sprinkled in between."""
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 3, 512, 18, 37
    feat = torch.randn(B, C, H, W, generator=g).bfloat16().float()
    n = 2500
    xy = torch.rand(n, 2, generator=g) * torch.tensor([1100.0, 500.0])
    wh = torch.rand(n, 2, generator=g) * torch.tensor([400.0, 250.0]) + 1
    rois = torch.cat([torch.randint(0, B, (n, 1), generator=g).float(), xy, xy + wh], 1)
    rois[::97, 0] = -1
    live = rois[:, 0] >= 0
    fr = feat.clone().requires_grad_(True)
    ref = torch.zeros(n, C, 7, 7)
    ref[live] = oracle_roi_align(fr, rois[live], 7, 1 / 32.0, 0, True)
    dout = torch.randn(n, C, 7, 7, generator=g).bfloat16().float()
    ref.backward(dout)
    fd = nhwc(feat).to(DEV).bfloat16()
    out = native.roi_align_fwd(fd, rois.to(DEV), 7, 1 / 32.0)
    got = out.float().cpu().view(n, 7, 7, C).permute(0, 3, 1, 2)
    assert rel_err(got, ref.detach()) < 8e-3
    dd = dout.permute(0, 2, 3, 1).reshape(n, 49, C).contiguous().to(DEV).bfloat16()
    dfeat = native.roi_align_bwd(dd, rois.to(DEV), (B, H, W, C), 7, 1 / 32.0)
    assert rel_err(nchw(dfeat.cpu()), fr.grad) < 1e-4
    # accumulate semantics: a second call adds onto the first result
    dfeat2 = native.roi_align_bwd(dd, rois.to(DEV), (B, H, W, C), 7, 1 / 32.0, dfeat=dfeat.clone())
    assert rel_err(dfeat2.cpu(), 2 * dfeat.cpu()) < 1e-5


def test_roi_align_bwd_tiled_gather_reproducible_and_chunked(native):
    """The pooled-7 backward is a gather (one owner per gradient element): two runs are bit-equal, more
    ROIs than one list pass holds (4096) are processed in chunks, maps whose sides are not multiples of the
    8-pixel tile and ROIs hanging over every border are handled; checked against the oracle's autograd."""
    g = torch.Generator().manual_seed(33)
    B, C, H, W = 2, 40, 21, 30
    n = 4500
    xy = torch.rand(n, 2, generator=g) * torch.tensor([W * 16.0 + 60, H * 16.0 + 60]) - 60
    wh = torch.rand(n, 2, generator=g) * torch.tensor([260.0, 200.0]) + 0.5
    rois = torch.cat([torch.randint(0, B, (n, 1), generator=g).float(), xy, xy + wh], 1)
    rois[::211, 0] = -1
    rois[3] = torch.tensor([1, -500.0, -500.0, -400.0, -450.0])      # entirely outside: no samples
    rois[4] = torch.tensor([0, 10.0, 10.0, 10.0, 10.0])              # zero-sized
    live = rois[:, 0] >= 0
    fr = torch.zeros(B, C, H, W, requires_grad=True)
    ref = oracle_roi_align(fr, rois[live], 7, 1 / 16.0, 0, True)
    dout = torch.randn(n, C, 7, 7, generator=g)
    ref.backward(dout[live])
    dd = dout.permute(0, 2, 3, 1).reshape(n, 49, C).contiguous().to(DEV)
    d1 = native.roi_align_bwd(dd, rois.to(DEV), (B, H, W, C), 7, 1 / 16.0)
    d2 = native.roi_align_bwd(dd, rois.to(DEV), (B, H, W, C), 7, 1 / 16.0)
    assert torch.equal(d1, d2)
    assert rel_err(nchw(d1.cpu()), fr.grad) < 1e-5


# -------------------------------------------------------------------------------------------------
# NMS / sort / matcher / sampling: bit-exact
# -------------------------------------------------------------------------------------------------
def _rand_boxes(n, g, span=600.0, size=200.0):
    xy = torch.rand(n, 2, generator=g) * span
    wh = torch.rand(n, 2, generator=g) * size + 1
    return torch.cat([xy, xy + wh], 1)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 129, 1000, 4097])
def test_nms_indices_bit_exact(native, n):
    g = torch.Generator().manual_seed(n)
    B = 2
    keeps = []
    boxes = torch.stack([_rand_boxes(n, g) for _ in range(B)])
    scores = torch.rand(B, n, generator=g)
    scores[:, n // 2] = scores[:, 0]                 # a tie
    valid = torch.ones(B, n, dtype=torch.uint8)
    if n > 10:
        boxes[0, 3, 2] = boxes[0, 3, 0]              # empty box: removed before NMS
    ss, si = native.segmented_sort_desc(scores.to(DEV))
    ref_order = torch.sort(scores, dim=1, descending=True, stable=True)[1]
    assert torch.equal(si.cpu().long(), ref_order)
    sb = torch.gather(boxes, 1, ref_order[..., None].expand(-1, -1, 4))
    nonempty = ((sb[..., 2] - sb[..., 0]) > 0) & ((sb[..., 3] - sb[..., 1]) > 0)
    max_keep = 300
    keep_idx, keep_cnt = native.nms(sb.to(DEV), 0.7, max_keep, valid=nonempty.to(torch.uint8).to(DEV))
    for b in range(B):
        live = torch.nonzero(nonempty[b]).squeeze(1)
        ref = live[OB.nms(sb[b][live], torch.arange(len(live), 0, -1).float(), 0.7)][:max_keep]
        c = keep_cnt[b].item()
        assert c == len(ref)
        assert keep_idx[b, :c].cpu().long().tolist() == ref.tolist()


@pytest.mark.parametrize("n,max_keep,dense", [(9990, 2000, False), (9990, 2000, True), (16000, 100, True),
                                              (16384, 16384, False), (20000, 2000, True)])
def test_nms_large_bit_exact(native, n, max_keep, dense):
    """The hot-path sizes: 9990 / 16000 candidates (16-wave reduce with prefetched mask rows, early
    exit at max_keep) and n > 16384 (single-wave-resolve fallback kernel)."""
    g = torch.Generator().manual_seed(n + max_keep)
    B = 2
    span = 250.0 if dense else 1100.0
    boxes = torch.stack([_rand_boxes(n, g, span=span, size=220.0) for _ in range(B)])
    npi = torch.tensor([n, n - 1234], dtype=torch.int32)
    keep_idx, keep_cnt = native.nms(boxes.to(DEV), 0.7, max_keep, n_per_image=npi.to(DEV))
    for b in range(B):
        m = int(npi[b])
        ref = OB.nms(boxes[b, :m], torch.arange(m, 0, -1).float(), 0.7)[:max_keep]
        c = keep_cnt[b].item()
        assert c == len(ref)
        assert keep_idx[b, :c].cpu().long().tolist() == ref.tolist()


def test_nms_progressive_phases_mixed_images(native):
    """Progressive NMS: image 0 (sparse boxes) reaches max_keep inside the first phase's 4096 boxes and is skipped by
    the later launches, image 1 (dense boxes) does not and is redone on more boxes, image 2 has a short live prefix.
    All three must equal the one-shot greedy result."""
    g = torch.Generator().manual_seed(77)
    n, max_keep = 9990, 2000
    base = _rand_boxes(300, g, span=1100.0, size=220.0)       # image 1: 300 clusters of near-duplicates
    dup = base[torch.randint(0, 300, (n,), generator=g)] + torch.rand(n, 4, generator=g)
    boxes = torch.stack([_rand_boxes(n, g, span=1100.0, size=220.0), dup, _rand_boxes(n, g, span=1100.0, size=220.0)])
    npi = torch.tensor([n, n, 700], dtype=torch.int32)
    keep_idx, keep_cnt = native.nms(boxes.to(DEV), 0.7, max_keep, n_per_image=npi.to(DEV))
    counts = []
    for b in range(3):
        m = int(npi[b])
        ref = OB.nms(boxes[b, :m], torch.arange(m, 0, -1).float(), 0.7)[:max_keep]
        c = keep_cnt[b].item()
        counts.append(c)
        assert c == len(ref)
        assert keep_idx[b, :c].cpu().long().tolist() == ref.tolist()
    assert counts[0] == max_keep and counts[1] < max_keep      # the two regimes the phases distinguish


@pytest.mark.parametrize("n", [2, 100, 9990, 16000, 16384, 16385, 20000, 30720, 34200, 98304, 131072])
def test_segmented_sort_stable_descending_with_ties(native, n):
    """LDS bitonic sort (n <= 16384) and the chunk-sort + merge-path passes above it: order identical to
    torch.sort(descending=True, stable=True), incl. heavy ties, +-0.0, -inf / +inf."""
    g = torch.Generator().manual_seed(n)
    B = 3
    scores = torch.randn(B, n, generator=g)
    scores[0] = torch.randint(0, 7, (n,), generator=g).float()          # heavy ties
    scores[1, ::5] = -1.0                                               # the "filtered candidate" marker
    if n > 10:
        scores[2, 3], scores[2, 7], scores[2, 5], scores[2, 9] = 0.0, -0.0, float("inf"), float("-inf")
    ss, si = native.segmented_sort_desc(scores.to(DEV))
    ref_v, ref_i = torch.sort(scores, dim=1, descending=True, stable=True)
    assert torch.equal(si.cpu().long(), ref_i)
    assert torch.equal(ss.cpu(), ref_v)


def test_nms_exact_tie_iou_and_threshold_strictness(native):
    boxes = torch.tensor([[[0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 10.0, 5.0], [0.0, 0.0, 10.0, 10.0]]])
    k, c = native.nms(boxes.to(DEV), 0.5, 8)
    assert k[0, :c[0].item()].tolist() == [0, 1]     # IoU == 0.5 is kept (strict >), duplicate removed


@pytest.mark.parametrize("G", [0, 1, 7, 100])
def test_anchor_match_labels_bit_exact(native, G):
    g = torch.Generator().manual_seed(G)
    B, Hf, Wf, stride = 2, 18, 37, 32
    cell = OB.cell_anchors((32, 64, 128, 256, 512), (0.5, 1.0, 2.0))
    anchors = OB.grid_anchors(Hf, Wf, stride, cell)
    gcap = 100
    gt = torch.zeros(B, gcap, 4)
    cnt = torch.tensor([G, max(G - 1, 0)], dtype=torch.int32)
    for b in range(B):
        gt[b, : cnt[b]] = _rand_boxes(int(cnt[b]), g, span=900.0, size=300.0)
    if G >= 7:
        gt[0, 2] = anchors[4000]                     # exact overlap -> IoU 1
        gt[0, 3] = torch.tensor([5000.0, 5000.0, 5100.0, 5100.0])   # overlaps nothing (the all-zero quirk)
    matched, labels = native.anchor_match(cell.to(DEV), B, Hf, Wf, stride, gt.to(DEV), cnt.to(DEV), 0.3, 0.7)
    for b in range(B):
        M = OB.pairwise_iou(gt[b, : cnt[b]], anchors)
        ridx, rlab = OB.matcher(M, [0.3, 0.7], [0, -1, 1], True)
        assert torch.equal(labels[b].cpu(), rlab)
        assert torch.equal(matched[b].cpu().long(), ridx)


@pytest.mark.parametrize("n_equal", [2000, 7000])   # 7000 equal keys overflow the histogram bin's list: bitwise fallback
def test_subsample_rpn_and_roi_exact(native, n_equal):
    g = torch.Generator().manual_seed(9)
    B, n = 3, 9990
    labels = torch.randint(-1, 2, (B, n), generator=g).to(torch.int8)
    labels[1] = 0
    labels[1, :40] = 1                                # fewer positives than 128
    labels[2] = -1
    labels[2, 5] = 0                                  # almost nothing to sample
    keys = torch.randint(0, 2 ** 31 - 1, (B, n), generator=g, dtype=torch.int64)
    keys[0, :n_equal] = 7                             # many equal keys: index breaks the tie
    ld = labels.clone().to(DEV)
    native.subsample_rpn_(ld, keys.to(torch.int32).to(DEV), 256, 0.5)
    for b in range(B):
        pos, neg = OB.subsample_labels(labels[b], 256, 0.5, 0, keys[b])
        ref = torch.full((n,), -1, dtype=torch.int8)
        ref[pos] = 1
        ref[neg] = 0
        assert torch.equal(ld[b].cpu(), ref)
    # ROI mode
    K = 8
    cls = torch.randint(0, K + 1, (B, 2100), generator=g).to(torch.int32)
    cls[:, 2050:] = -2                                # beyond the live prefix
    cls[0, :1500] = K
    keys = torch.randint(0, 2 ** 31 - 1, (B, 2100), generator=g, dtype=torch.int64)
    idx, cnt = native.subsample_roi(cls.to(DEV), keys.to(torch.int32).to(DEV), 512, 0.25, K)
    for b in range(B):
        c = cls[b].long().clone()
        live = c != -2
        fg, bg = OB.subsample_labels(torch.where(live, c, torch.full_like(c, -1)), 512, 0.25, K, keys[b])
        ref = torch.cat([fg, bg])
        assert cnt[b].item() == len(ref)
        assert idx[b, : len(ref)].cpu().long().tolist() == ref.tolist()


def test_roi_match_and_sample_building(native):
    g = torch.Generator().manual_seed(21)
    B, P, gcap, K = 2, 300, 100, 8
    props = torch.stack([_rand_boxes(P, g) for _ in range(B)])
    pc = torch.tensor([300, 250], dtype=torch.int32)
    gt = torch.zeros(B, gcap, 4)
    gcl = torch.zeros(B, gcap, dtype=torch.int32)
    gc = torch.tensor([6, 0], dtype=torch.int32)
    gt[0, :6] = props[0, 10:16] + 3.0
    gcl[0, :6] = torch.randint(0, K, (6,), generator=g).int()
    boxes, cnt = native.append_gt(props.to(DEV), pc.to(DEV), gt.to(DEV), gc.to(DEV))
    assert cnt.tolist() == [306, 250]
    torch.testing.assert_close(boxes[0, 300:306].cpu(), gt[0, :6])
    matched, cls = native.roi_match(boxes, cnt, gt.to(DEV), gcl.to(DEV), gc.to(DEV), 0.5, K)
    M = OB.pairwise_iou(gt[0, :6], boxes[0, :306].cpu())
    ridx, rlab = OB.matcher(M, [0.5], [0, 1], False)
    rc = gcl[0, :6].long()[ridx]
    rc[rlab == 0] = K
    assert torch.equal(cls[0, :306].cpu().long(), rc)
    assert (cls[0, 306:] == -2).all() and (cls[1, :250] == K).all() and (cls[1, 250:] == -2).all()
    keys = torch.randint(0, 2 ** 31 - 1, (B, P + gcap), generator=g, dtype=torch.int32)
    sidx, scnt = native.subsample_roi(cls.clone(), keys.to(DEV), 64, 0.25, K)
    rois, gt_cls, gt_box, n_valid = native.roi_build_samples(boxes, cls, matched, sidx, scnt, gt.to(DEV), gc.to(DEV))
    assert n_valid.item() == scnt.sum().item()
    s0 = scnt[0].item()
    torch.testing.assert_close(rois[:s0, 1:].cpu(), boxes[0].cpu()[sidx[0, :s0].cpu().long()])
    assert (rois[:s0, 0] == 0).all() and (rois[64:64 + scnt[1].item(), 0] == 1).all()
    assert torch.equal(gt_cls[:s0].cpu(), cls[0].cpu()[sidx[0, :s0].cpu().long()])
    torch.testing.assert_close(gt_box[:s0].cpu(), gt[0][matched[0].cpu().long()[sidx[0, :s0].cpu().long()]])
    assert (gt_box[64:] == 0).all()


# -------------------------------------------------------------------------------------------------
# RPN proposals, losses, teacher post-processing
# -------------------------------------------------------------------------------------------------
def _rpn_out(B, Hf, Wf, A, g, ld=80):
    out = torch.zeros(B * Hf * Wf, ld)
    out[:, :A] = torch.randn(B * Hf * Wf, A, generator=g) * 2
    out[:, A:5 * A] = torch.randn(B * Hf * Wf, 4 * A, generator=g) * 0.5
    return out


def _split_rpn_out(out, B, Hf, Wf, A):
    logits = out[:, :A].reshape(B, Hf * Wf * A)
    deltas = out[:, A:5 * A].reshape(B, Hf * Wf * A, 4)
    return logits, deltas


def test_rpn_proposal_pipeline_matches_oracle(native):
    g = torch.Generator().manual_seed(4)
    B, Hf, Wf, stride, A = 2, 18, 37, 32, 15
    cfg = om.Cfg(rpn_pre_topk_train=3000, rpn_post_topk_train=500)
    cell = OB.cell_anchors(cfg.anchor_sizes, cfg.anchor_ratios)
    anchors = OB.grid_anchors(Hf, Wf, stride, cell)
    out = _rpn_out(B, Hf, Wf, A, g)
    logits, deltas = _split_rpn_out(out, B, Hf, Wf, A)
    sizes = [(600, 1200), (576, 1184)]
    ref = om.rpn_proposals(anchors, logits, deltas, sizes, cfg, training=True)
    szd = torch.tensor(sizes, dtype=torch.int32, device=DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    props, scores = native.rpn_decode(out.to(DEV), cell.to(DEV), B, Hf, Wf, stride, szd, flags)
    ss, si = native.segmented_sort_desc(scores)
    cb, cs, cv = native.rpn_gather_topk(props, ss, si, 3000)
    keep_idx, keep_cnt = native.nms(cb, 0.7, 500, valid=cv)
    pb, ps = native.gather_kept(cb, cs, keep_idx, keep_cnt)
    assert flags.item() == 0
    for b in range(B):
        n = keep_cnt[b].item()
        assert n == len(ref[b][0])
        torch.testing.assert_close(pb[b, :n].cpu(), ref[b][0], rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(ps[b, :n].cpu(), ref[b][1], rtol=0, atol=0)
        assert (pb[b, n:] == 0).all()


def test_rpn_loss_and_grad(native):
    g = torch.Generator().manual_seed(6)
    B, Hf, Wf, stride, A = 2, 9, 11, 32, 15
    cfg = om.Cfg()
    cell = OB.cell_anchors(cfg.anchor_sizes, cfg.anchor_ratios)
    anchors = OB.grid_anchors(Hf, Wf, stride, cell)
    NA = Hf * Wf * A
    out = _rpn_out(B, Hf, Wf, A, g)
    gcap = 100
    gt = torch.zeros(B, gcap, 4)
    gc = torch.tensor([5, 0], dtype=torch.int32)
    gt[0, :5] = _rand_boxes(5, g, span=250.0, size=120.0)
    keys = torch.randint(0, 2 ** 31 - 1, (B, NA), generator=g, dtype=torch.int64)
    labels, matched_gt = om.rpn_label_anchors(anchors, [gt[b, : gc[b]] for b in range(B)], list(keys), cfg)
    outr = out.clone().requires_grad_(True)
    logits, deltas = _split_rpn_out(outr, B, Hf, Wf, A)
    ref = om.rpn_losses(anchors, logits, deltas, labels, matched_gt, cfg)
    (ref["loss_rpn_cls"] * 0.7 + ref["loss_rpn_loc"] * 1.3).backward()
    matched, lab = native.anchor_match(cell.to(DEV), B, Hf, Wf, stride, gt.to(DEV), gc.to(DEV), 0.3, 0.7)
    native.subsample_rpn_(lab, keys.to(torch.int32).to(DEV), 256, 0.5)
    assert torch.equal(lab.cpu(), labels)
    gs = torch.tensor([0.7, 1.3], device=DEV)
    loss, d_out = native.rpn_loss(out.to(DEV), cell.to(DEV), B, Hf, Wf, stride, lab, matched, gt.to(DEV),
                                  gc.to(DEV), 256, grad_scale=gs)
    np.testing.assert_allclose(loss[0].item(), ref["loss_rpn_cls"].item(), rtol=1e-5)
    np.testing.assert_allclose(loss[1].item(), ref["loss_rpn_loc"].item(), rtol=1e-5)
    torch.testing.assert_close(d_out.cpu(), outr.grad, rtol=1e-4, atol=1e-9)


def test_frcnn_loss_and_grad(native):
    g = torch.Generator().manual_seed(8)
    R, K, ld = 300, 8, 48
    cfg = om.Cfg()
    pred = torch.zeros(R, ld)
    pred[:, :41] = torch.randn(R, 41, generator=g)
    boxes = _rand_boxes(R, g)
    gtb = boxes + torch.randn(R, 4, generator=g) * 4
    gtb[:, 2:] = torch.max(gtb[:, 2:], gtb[:, :2] + 1)
    cls = torch.randint(0, K + 1, (R,), generator=g)
    cls[250:] = -1                                      # padding rows
    live = cls >= 0
    pr = pred.clone().requires_grad_(True)
    ref = om.fast_rcnn_losses(pr[live, :9], pr[live, 9:41], boxes[live], cls[live], gtb[live], cfg)
    (ref["loss_cls"] * 1.5 + ref["loss_box_reg"] * 0.5).backward()
    rois = torch.cat([torch.zeros(R, 1), boxes], 1)
    nv = torch.tensor([int(live.sum())], dtype=torch.int32)
    gs = torch.tensor([1.5, 0.5], device=DEV)
    loss, d_pred = native.frcnn_loss(pred.to(DEV), K, rois.to(DEV), cls.int().to(DEV), gtb.to(DEV), nv.to(DEV), gs)
    np.testing.assert_allclose(loss[0].item(), ref["loss_cls"].item(), rtol=1e-5)
    np.testing.assert_allclose(loss[1].item(), ref["loss_box_reg"].item(), rtol=1e-5)
    torch.testing.assert_close(d_pred.cpu(), pr.grad, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("limit", [20000, 400])
def test_teacher_postprocessing_matches_oracle(native, limit):
    """softmax + decode + clip + score>0.05 + class-wise NMS(0.5) + top-100 + score>0.8, for both
    torchvision batched_nms strategies (coordinate trick / per-class loop)."""
    g = torch.Generator().manual_seed(12)
    B, P, K, ld = 2, 400, 8, 48
    cfg = om.Cfg(nms_numel_limit=limit)
    pc = torch.tensor([400, 333], dtype=torch.int32)
    props = torch.stack([_rand_boxes(P, g, span=900.0, size=250.0) for _ in range(B)])
    pred = torch.zeros(B * P, ld)
    pred[:, :9] = torch.randn(B * P, 9, generator=g) * 3
    pred[:, 9:41] = torch.randn(B * P, 32, generator=g) * 0.7
    sizes = [(600, 1200), (590, 1100)]
    scores = torch.cat([pred[b * P: b * P + pc[b], :9] for b in range(B)])
    deltas = torch.cat([pred[b * P: b * P + pc[b], 9:41] for b in range(B)])
    ref = om.fast_rcnn_inference(scores, deltas, [props[b, : pc[b]] for b in range(B)], sizes, cfg)
    out = native.frcnn_inference(pred.to(DEV), K, props.to(DEV), pc.to(DEV),
                                 torch.tensor(sizes, dtype=torch.int32, device=DEV), 0.05, 0.5, 100, 0.8,
                                 numel_limit=limit)
    for b in range(B):
        n = out["det_count"][b].item()
        assert n == len(ref[b]["scores"])
        assert out["det_classes"][b, :n].cpu().long().tolist() == ref[b]["classes"].tolist()
        torch.testing.assert_close(out["det_scores"][b, :n].cpu(), ref[b]["scores"], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(out["det_boxes"][b, :n].cpu(), ref[b]["boxes"], rtol=1e-5, atol=2e-3)
        pl = om.threshold_bbox(ref[b], 0.8)
        m = out["gt_count"][b].item()
        assert m == len(pl["gt_classes"])
        assert out["gt_classes"][b, :m].cpu().long().tolist() == pl["gt_classes"].tolist()
        torch.testing.assert_close(out["gt_boxes"][b, :m].cpu(), pl["gt_boxes"], rtol=1e-5, atol=2e-3)


# -------------------------------------------------------------------------------------------------
# preprocess, optimiser
# -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", _dtypes())
def test_preprocess(native, dtype):
    g = torch.Generator().manual_seed(1)
    imgs = [torch.randint(0, 256, (3, 20, 31), generator=g, dtype=torch.uint8),
            torch.randint(0, 256, (3, 17, 33), generator=g, dtype=torch.uint8)]
    ref, sizes = om.preprocess(imgs)
    dt = native.dt_of(torch.empty(0, dtype=dtype))
    cpad = native.chunk_elems(dt)
    x, sz = native.preprocess([im.to(DEV) for im in imgs], 20, 33, cpad, om.PIXEL_MEAN, om.PIXEL_STD, dt)
    assert sz.tolist() == [list(s) for s in sizes]
    got = nchw(x.float().cpu())
    if dtype == torch.float32:
        torch.testing.assert_close(got[:, :3], ref, rtol=0, atol=0)
    else:
        torch.testing.assert_close(got[:, :3], ref.bfloat16().float(), rtol=0, atol=0)
    assert (got[:, 3:] == 0).all()


def test_sgd_ema_fused(native):
    g = torch.Generator().manual_seed(3)
    n = 100003
    p = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g)
    t = torch.randn(n, generator=g)
    sd = {"w": p.clone()}
    bufs = {}
    pd, td = p.clone().to(DEV), t.clone().to(DEV)
    md = torch.zeros(n, device=DEV)
    lr = torch.tensor([0.02], device=DEV)
    tref = {"w": t.clone()}
    for step in range(3):
        om.sgd_step(sd, {"w": gr * (step + 1)}, bufs, lr=0.02, momentum=0.9, weight_decay=1e-4)
        om.ema_update(tref, sd, 0.9996)
        native.sgd_ema_(pd, (gr * (step + 1)).to(DEV), md, td, lr, 0.9, 1e-4, 1.0, 0.9996, step == 0)
    torch.testing.assert_close(pd.cpu(), sd["w"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(md.cpu(), bufs["w"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(td.cpu(), tref["w"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("shape", [(64, 128, 38, 75), (100, 50, 37, 19), (33, 77, 66, 154), (512, 1024, 300, 600),
                                   (1024, 2048, 600, 1200), (97, 131, 41, 53), (75, 64, 30, 64)])
@pytest.mark.parametrize("flip", [False, True])
def test_resize_bilinear_u8_equals_pillow(native, shape, flip):
    """sfod_resize_bilinear_u8 (the mapper's ResizeShortestEdge [+ RandomFlip] on device) is bit-exact with
    Pillow's Image.resize(BILINEAR) -- and with oracle/resize.py, its restatement."""
    from PIL import Image
    from oracle.resize import resize_bilinear_u8
    H, W, h, w = shape
    g = torch.Generator().manual_seed(H + W)
    img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8)
    ref = np.asarray(Image.fromarray(img.permute(1, 2, 0).numpy()).resize((w, h), Image.BILINEAR)).transpose(2, 0, 1)
    if H * W <= 64 * 128:
        assert np.array_equal(resize_bilinear_u8(img.numpy(), h, w), ref)
    if flip:
        ref = ref[:, :, ::-1]
    got = native.resize_bilinear_u8(img.to(DEV), h, w, flip=flip).cpu().numpy()
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("dtype", _dtypes())
def test_conv_weight_packer_equals_per_layer_packing(native, dtype):
    """sfod_pack_conv_weights_multi (all layers in one launch, LDS-tiled transpose) == sfod_pack_conv_weight per
    layer, bit for bit: forward and rotated layouts, channel padding (3 -> 8), ragged tiles, 1x1 kernels."""
    dt = native.dt_of(torch.empty(0, dtype=dtype))
    E = native.chunk_elems(dt)
    g = torch.Generator().manual_seed(4)
    shapes = [(64, 3, 3), (72, 40, 3), (128, 64, 3), (40, 96, 1), (257, 33, 3)]
    ws = [torch.randn(co, ci, k, k, generator=g).to(DEV) for co, ci, k in shapes]
    pad = lambda c: (c + E - 1) // E * E
    specs = [(w, pad(w.shape[1]), False) for w in ws] + [(w, pad(w.shape[0]), True) for w in ws]
    views = native.ConvWeightPacker(specs, dt).pack()
    for (w, p_, rot), v in zip(specs, views):
        ref = native.pack_conv_weight(w, p_, dt, rot180=rot)
        assert v.shape == ref.shape and torch.equal(v, ref), (tuple(w.shape), rot)


def test_conv_first_recompute_fused_batchnorm(native):
    """sfod_conv_first_fused: the store-free pass yields exactly the statistics of the storing pass, and the
    second pass writes relu(bn(conv)) directly (checked against torch on bf16-rounded operands; it is slightly
    MORE accurate than conv -> bf16 -> BatchNorm because the affine sees the fp32 accumulator)."""
    B, H, W = 2, 70, 130
    g = torch.Generator().manual_seed(8)
    x = torch.zeros(B, H, W, 8)
    x[..., :3] = torch.randn(B, H, W, 3, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) / math.sqrt(27)
    bias = torch.randn(64, generator=g) * 0.1
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    xd = x.to(DEV).bfloat16()
    wp = native.pack_conv_weight(w.to(DEV), 8, native.BF16)
    assert native.conv_first_supported(xd, 64)
    y, st_ref = native.conv_fwd(xd, wp, bias.to(DEV), 64, 3, want_stats=True)
    st = native.conv_first_stats(xd, wp, bias.to(DEV))
    assert st.nblk == st_ref.nblk and torch.equal(st, st_ref)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    mean, invstd = native.bn_finalize(st, B * H * W, 64, rm, rv, 0.1, 1e-5, False)
    scale = gamma.to(DEV) * invstd
    shift = beta.to(DEV) - mean * scale
    z = native.conv_first_apply(xd, wp, bias.to(DEV), scale, shift, relu=True)
    xr, wr = nchw(xd.float().cpu())[:, :3], w.bfloat16().float()
    yc = F.conv2d(xr, wr, bias, padding=1)
    ref = F.relu(F.batch_norm(yc, None, None, gamma, beta, True, 0.1, 1e-5))
    assert rel_err(nchw(z.float().cpu()), ref) < 6e-3
    # and it agrees with the unfused product path (conv -> bf16 -> BN kernel) to bf16 rounding
    z2 = native.bn_relu_pool_fwd(y, mean, invstd, gamma.to(DEV), beta.to(DEV), False)
    assert rel_err(z.float().cpu(), z2.float().cpu()) < 8e-3


# -------------------------------------------------------------------------------------------------
# strong augmentation: bit-exact against the oracle (pinned to Pillow in tests/test_augment.py)
# -------------------------------------------------------------------------------------------------
def _aug_frame(seed, h=97, w=131):
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)
    img[:, :8, :8] = 0                      # black, white, gray and saturated corners: HSV special cases
    img[:, :8, 8:16] = 255
    img[:, 8:16, :8] = 128
    img[0, 8:16, 8:16], img[1, 8:16, 8:16], img[2, 8:16, 8:16] = 255, 0, 0
    return img


def _hwc(t):
    return t.permute(1, 2, 0).contiguous().numpy()


def test_aug_color_ops_bit_exact(native):
    from oracle import augment as A
    import itertools
    img = _aug_frame(3)
    ref_in = _hwc(img)
    d = img.to(DEV)
    rng = np.random.default_rng(7)
    singles = [[(A.BRIGHTNESS, f)] for f in (0.6, 0.93, 1.0, 1.4)] + [[(A.CONTRAST, f)] for f in (0.6, 1.0, 1.37)] + \
              [[(A.SATURATION, f)] for f in (0.6, 1.21, 1.4)] + [[(A.HUE, f)] for f in (-0.1, -0.037, 0.0, 0.02, 0.1)] + \
              [[(A.GRAYSCALE, 0.0)]]
    perms = []
    for p in list(itertools.permutations(range(4)))[::3]:
        f = {0: rng.uniform(0.6, 1.4), 1: rng.uniform(0.6, 1.4), 2: rng.uniform(0.6, 1.4), 3: rng.uniform(-0.1, 0.1)}
        ops = [(c, float(f[c])) for c in p]
        if rng.random() < 0.5:
            ops.append((A.GRAYSCALE, 0.0))
        perms.append(ops)
    for ops in singles + perms:
        got = native.aug_color(d, ops)
        ref = A.apply_ops(ref_in, ops)
        assert np.array_equal(_hwc(got.cpu()), ref), ops
    assert torch.equal(native.aug_color(d, []), d)
    # full-size frame: the contrast mean is a 720 000-pixel integer sum
    big = torch.randint(0, 256, (3, 600, 1200), generator=torch.Generator().manual_seed(1), dtype=torch.uint8)
    ops = [(A.BRIGHTNESS, 1.13), (A.CONTRAST, 0.71), (A.HUE, 0.06), (A.SATURATION, 1.33)]
    assert np.array_equal(_hwc(native.aug_color(big.to(DEV), ops).cpu()), A.apply_ops(_hwc(big), ops))


def test_aug_gaussian_blur_bit_exact(native):
    from oracle import augment as A
    for (h, w) in [(97, 131), (5, 7), (1, 40), (40, 1), (600, 1200)]:
        img = torch.randint(0, 256, (3, h, w), generator=torch.Generator().manual_seed(h), dtype=torch.uint8)
        for sigma in ([0.1, 0.3, 0.77, 1.0, 1.3, 1.7, 2.0, 3.5] if h < 600 else [1.234]):
            got = native.aug_gaussian_blur(img.to(DEV), sigma)
            assert np.array_equal(_hwc(got.cpu()), A.gaussian_blur(_hwc(img), sigma)), (h, w, sigma)


def test_aug_erase_bit_exact(native):
    from oracle import augment as A
    img = _aug_frame(5)
    g = torch.Generator().manual_seed(9)
    d = img.to(DEV).clone()
    ref = _hwc(img)
    for (i, j, h, w) in [(0, 0, 10, 20), (50, 60, 47, 71), (96, 130, 1, 1), (3, 4, 0, 5)]:
        noise = torch.randn(3, h, w, generator=g) * 1.7
        native.aug_erase_(d, i, j, h, w, noise.to(DEV))
        ref = A.erase(ref, i, j, h, w, noise.numpy())
        assert np.array_equal(_hwc(d.cpu()), ref), (i, j, h, w)
    with pytest.raises(native.NativeLibraryError):
        native.aug_erase_(d, 90, 0, 10, 10, torch.zeros(3, 10, 10, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_add_relu_fwd_equals_the_two_pass_form(native, dtype):
    """Bottleneck tail relu(bn(y) + shortcut) in one pass: identical to BN-apply followed by add + ReLU in fp32;
    in bf16 it skips the rounding of the intermediate, so it is compared against the fp32 result."""
    g = torch.Generator().manual_seed(4)
    B, H, W, C = 2, 19, 23, 64
    y = torch.randn(B, H, W, C, generator=g)
    res = torch.randn(B, H, W, C, generator=g)
    mean, invstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    ref = torch.relu((y - mean) * (invstd * gamma) + beta + res)
    args = [t.to(DEV) for t in (mean, invstd, gamma, beta)]
    yd, rd = y.to(DEV).to(dtype), res.to(DEV).to(dtype)
    got = native.bn_add_relu_fwd(yd, *args, rd)
    if dtype == torch.float32:
        two = native.add_act(native.bn_relu_pool_fwd(yd, *args, False, relu=False), rd, 1)
        assert torch.equal(got, two)
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-6, atol=1e-6)
        # dual form: the fp32 result plus the same values as bf16x3 operand pairs, one pass (vector width 8 instead of
        # 4: the fp32 half may differ by FMA contraction, the pairs are the exact split of that half)
        z, zp = native.bn_add_relu_fwd(yd, *args, rd, with_operand=True)
        torch.testing.assert_close(z, got, rtol=2e-7, atol=1e-7)
        assert zp.dtype == native.SPLIT_DTYPE and torch.equal(zp.view(torch.bfloat16), native.cast(z, native.SPLIT_DTYPE).view(torch.bfloat16))
        # backward-side fusion of the same join: (a + b) * [y > 0] == add_ followed by act_bwd_
        a1, b1 = torch.randn_like(yd), torch.randn_like(yd)
        ref_g = native.act_bwd_(native.add_(a1.clone(), b1), got, 1)
        assert torch.equal(native.add_act_bwd_(a1.clone(), b1, got), ref_g)
        o, op = native.add_act(yd, rd, 1, with_operand=True)
        assert torch.equal(o, native.add_act(yd, rd, 1))
        assert torch.equal(op.view(torch.bfloat16), native.cast(o, native.SPLIT_DTYPE).view(torch.bfloat16))
    else:
        ref16 = torch.relu((yd.float().cpu() - mean) * (invstd * gamma) + beta + rd.float().cpu())
        torch.testing.assert_close(got.float().cpu(), ref16, rtol=1e-2, atol=1e-2)
        assert (got.float().cpu() - ref16).abs().max() <= (ref16.abs().max() / 128)


@pytest.mark.parametrize("dtype", ["bf16x3", "f16x3", "fp32"])
@pytest.mark.parametrize("shape", [(512, 25088, 1024), (1000, 25088, 1024), (300, 16384, 260), (64, 50176, 2048)])
def test_linear_split_k_for_few_rows(native, dtype, shape):
    """sfod_conv_fwd_scratch: a linear layer whose grid would fill less than half the chip (the ROI head's fc1 at one frame
    per GPU) runs split along K into slabs + a fixed-order slab sum with bias and activation.  Same values as the unsplit
    kernel up to fp32 summation order, run-to-run identical, and shapes that fill the chip ask for no scratch."""
    M, K, N = shape
    g = torch.Generator(device=DEV).manual_seed(M + N)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    if dtype == "bf16x3":
        xo, wo, dt = native.cast(x, native.SPLIT_DTYPE), native.cast(w, native.SPLIT_DTYPE), native.BF16X3
    elif dtype == "f16x3":
        xo, wo, dt = native.cast(x, native.SPLITH_DTYPE), native.pack_fc_weight(w, native.F16X3), native.F16X3
    else:
        xo, wo, dt = x, w, native.F32
    nbytes = native.load().sfod_conv_fwd_scratch_bytes(M, 1, 1, K, N, 1, dt, native.F32, 0)
    assert native.load().sfod_conv_fwd_scratch_bytes(16000, 1, 1, K, N, 1, dt, native.F32, 0) == 0
    assert native.load().sfod_conv_fwd_scratch_bytes(M, 1, 1, K, N, 1, dt, native.F32, 1) == 0     # statistics: unsplit
    assert native.load().sfod_conv_fwd_scratch_bytes(M, 1, 1, 1024, N, 1, dt, native.F32, 0) == 0  # short K: unsplit
    assert nbytes > 0 and nbytes % (M * N * 4) == 0
    ref = torch.empty(M, N, device=DEV)
    native.call("sfod_conv_fwd_ws", xo, wo, native.wscale_of(wo), bias, ref, M, 1, 1, K, N, 1, N, 1, None, dt, native.F32)
    outs = [native.conv_fwd(xo, wo, bias, N, 1, act=1) for _ in range(3)]
    wide = native.conv_fwd(xo, wo, bias, N, 1, act=1, ldy=N + 8)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(wide[:, :N], outs[0]) and float(wide[:, N:].abs().max()) == 0.0
    exact = torch.relu(x.double() @ w.double().t() + bias.double())
    err_ref = ((ref.double() - exact).norm() / exact.norm()).item()
    err_split = ((outs[0].double() - exact).norm() / exact.norm()).item()
    assert err_split <= max(1.5 * err_ref, 2e-6), (err_split, err_ref)
    # (true-fp32 products over K = 25088 carry ~3e-6 of summation-order noise themselves)
    assert ((outs[0] - ref).double().norm() / ref.double().norm()).item() < max(2e-6, 3 * err_ref)


@pytest.mark.parametrize("n,thr,span", [(600, 0.7, 300.0), (2000, 0.5, 500.0)])
def test_nms_kernel_equals_the_independent_huggingface_implementation(native, n, thr, span):
    """The ballot / readlane NMS kernels against an NMS nobody here wrote: HuggingFace transformers' OwlViT post-processing
    (tests/helpers/hf_nms.py; greedy, fp32 inter / union, strict '>').  Same corner boxes, same scores -> the same keep list."""
    from helpers import hf_nms as H
    g = torch.Generator().manual_seed(n)
    B = 2
    keep_ref, sorted_boxes, orders = [], [], []
    for b in range(B):
        centers, scores = H.random_centers(n, g, span=span), H.distinct_scores(n, g)
        corners = H.hf_corners(centers)
        order = torch.argsort(scores, descending=True)
        keep_ref.append(H.hf_greedy_nms(centers, scores, thr))
        sorted_boxes.append(corners[order])
        orders.append(order)
    keep_idx, keep_cnt = native.nms(torch.stack(sorted_boxes).to(DEV), thr, n)
    for b in range(B):
        c = keep_cnt[b].item()
        got = orders[b][keep_idx[b, :c].cpu().long()]
        assert 0 < len(keep_ref[b]) < n and got.tolist() == keep_ref[b].tolist()
