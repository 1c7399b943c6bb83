"""GPU box: per-layer time of the BatchNorm-input fold (sfod_conv_fwd_bnin) against the two-launch form
(sfod_bn_relu_pool_fwd writing pairs + sfod_conv_fwd), teacher batch of 8 frames at 600 x 1200.
usage: python3 tools/experiments/bnin_layers.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("simple-sfod_amd.native")
DEV = "cuda"
LAYERS = [("conv2_2", 8, 300, 600, 128, 128), ("conv3_2", 8, 150, 300, 256, 256), ("conv4_1", 8, 75, 150, 256, 512),
          ("conv4_2", 8, 75, 150, 512, 512)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, B, H, W, Cin, Cout in LAYERS:
    g = torch.Generator(device=DEV).manual_seed(1)
    y_pre = torch.randn(B, H, W, Cin, device=DEV, generator=g)
    gamma = torch.rand(Cin, device=DEV, generator=g) + 0.5
    beta = torch.rand(Cin, device=DEV, generator=g)
    mean = y_pre.mean(dim=(0, 1, 2))
    invstd = torch.rsqrt(y_pre.var(dim=(0, 1, 2), unbiased=False) + 1e-5)
    w = native.cast(torch.randn(Cout, 9, Cin, device=DEV, generator=g) / (3 * Cin ** 0.5), native.SPLIT_DTYPE)
    bias = torch.randn(Cout, device=DEV, generator=g)
    z = native.bn_relu_pool_fwd(y_pre, mean, invstd, gamma, beta, False, out_dtype=native.SPLIT_DTYPE)
    t_bn = timeit(lambda: native.bn_relu_pool_fwd(y_pre, mean, invstd, gamma, beta, False, out_dtype=native.SPLIT_DTYPE))
    res = {}
    for m in (1, 2):
        native.set_conv3x3_m16(m)
        res[m] = timeit(lambda: native.conv_fwd(z, w, bias, Cout, 3, want_stats=True))
    native.set_conv3x3_m16(1)
    t_f = timeit(lambda: native.conv_fwd_bnin(y_pre, mean, invstd, gamma, beta, w, bias, Cout, want_stats=True))
    print("{:8s} bn {:.3f}  conv(8w) {:.3f}  conv(4w) {:.3f}  two-launch {:.3f}  fold {:.3f} ms".format(
        name, t_bn, res[1], res[2], t_bn + res[1], t_f), flush=True)
