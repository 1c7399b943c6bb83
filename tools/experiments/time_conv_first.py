"""GPU box: the three forms of the first VGG layer at a teacher / student batch of 8 frames (600 x 1200), ms per launch:
student forward (y fp32 + statistics), teacher pass 1 (statistics only), teacher pass 2 (BatchNorm + ReLU applied, z pairs)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("simple-sfod_amd.native"); native.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
x = native.cast(torch.randn(8, 600, 1200, 8, device=dev, generator=g), native.SPLIT_DTYPE)
w = native.pack_conv_weight(torch.randn(64, 3, 3, 3, device=dev, generator=g) * 0.2, 8, native.BF16X3)
bias = torch.randn(64, device=dev, generator=g)
scale, shift = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("student forward   {:.3f} ms".format(timeit(lambda: native.conv_fwd(x, w, bias, 64, 3, want_stats=True))))
print("teacher stats     {:.3f} ms".format(timeit(lambda: native.conv_first_stats(x, w, bias))))
print("teacher apply     {:.3f} ms".format(timeit(lambda: native.conv_first_apply(x, w, bias, scale, shift, relu=True))))
