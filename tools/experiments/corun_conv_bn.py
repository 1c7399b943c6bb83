"""Experiment: does a BatchNorm apply pass (HBM-bound, no LDS, ~64 VGPRs) run BESIDE a halo-patch convolution launched on
another stream, or only in its tails?  Times 12 convolutions (conv3_2 shape, operand pairs) on stream A, 24 BatchNorm apply
launches on stream B, and both together, per convolution variant (2: 32x32x16, 128 VGPRs x 4 waves per SIMD = the whole
register file; 5: 16x16x32, 8 waves, 128 x 4; 6: 16x16x32, 4 waves per workgroup, 208 x 2 = room for one 72-register wave per
SIMD)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
B, H, W, Cin, Cout = 8, 150, 300, 256, 256
x = native.cast(torch.relu(torch.randn(B, H, W, Cin, device=dev, generator=g)), native.SPLIT_DTYPE)
w = native.cast(torch.randn(Cout, 9, Cin, device=dev, generator=g) / (3 * Cin ** 0.5), native.SPLIT_DTYPE)
bias = torch.randn(Cout, device=dev, generator=g)
yb = torch.randn(8, 300, 600, 128, device=dev, generator=g)
mean, invstd = yb.mean(dim=(0, 1, 2)), torch.rsqrt(yb.var(dim=(0, 1, 2)) + 1e-5)
gamma, beta = torch.ones(128, device=dev), torch.zeros(128, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
NA, NB = 12, 24

def run(do_a, do_b):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    if do_a:
        with torch.cuda.stream(sa):
            for _ in range(NA):
                native.conv_fwd(x, w, bias, Cout, 3, want_stats=True)
    if do_b:
        with torch.cuda.stream(sb):
            for _ in range(NB):
                native.bn_relu_pool_fwd(yb, mean, invstd, gamma, beta, False, out_dtype=native.SPLIT_DTYPE)
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)

native.set_conv_algo(2)
for v in (2, 5, 6):
    native.set_conv3x3_variant(v)
    for _ in range(2):
        run(True, True)
    ta = min(run(True, False) for _ in range(3)); tb = min(run(False, True) for _ in range(3)); tab = min(run(True, True) for _ in range(3))
    print(f"variant {v}: conv alone {ta:7.3f} ms | bn alone {tb:7.3f} ms | together {tab:7.3f} ms | sum {ta + tb:7.3f} | hidden {(ta + tb - tab) / tb * 100:5.1f} % of the bn time", flush=True)
