"""Micro-benchmark of the frozen-stage pieces of the ResNet-C4 trunk (config #5) at B = 8, 600x1200:
the 7x7 stem, sfod_stem7x7 vs sfod_im2col_stem + GEMM."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

sfod = importlib.import_module("simple-sfod_amd")
n = sfod.native
n.load()


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1000.0


for dtype in ("f16x3", "bf16x3"):
    tdt = n.mode_dtype(dtype)
    dt = n.dt_of_dtype(tdt)
    B, H, W = 8, 600, 1200
    x = torch.zeros(B, H, W, 8, device="cuda")
    x[..., :3] = torch.randn(B, H, W, 3, device="cuda") * 60
    wk = torch.zeros(64, 160, device="cuda")
    wk[:, :147] = torch.randn(64, 147, device="cuda") * 0.05
    bias = torch.randn(64, device="cuda")
    wp = n.pack_fc_weight(wk, dt)

    def old():
        cols = n.im2col_stem(x, 160, out_dtype=tdt)
        Bc, Ho, Wo, _ = cols.shape
        return n.conv_fwd(cols.view(Bc * Ho * Wo, 160), wp, bias, 64, 1, act=1)

    t_im2col = timeit(lambda: n.im2col_stem(x, 160, out_dtype=tdt))
    t_old = timeit(old)
    t_new = timeit(lambda: n.stem7x7(x, wp, bias, act=1))
    y = n.stem7x7(x, wp, bias, act=1)
    t_pool = timeit(lambda: n.maxpool3s2(y))
    print(f"[{dtype}] stem: im2col {t_im2col:8.1f} us, im2col + GEMM {t_old:8.1f} us, fused {t_new:8.1f} us, maxpool {t_pool:8.1f} us")
