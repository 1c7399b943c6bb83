"""Times the bf16x3 halo-patch convolution in its 16x16x32 form (variant 5; forward, with BatchNorm statistics) of three VGG
layers with whatever library SFOD_HIP_LIB names (no checks).  argv[1]: randn | zeros (zeros: the same instruction stream at
the clock the chip holds without switching power)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "randn"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for name, H, W, Cin, Cout in (("conv2_2", 300, 600, 128, 128), ("conv3_2", 150, 300, 256, 256), ("conv4_2", 75, 150, 512, 512)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = native.cast(torch.randn(8, H, W, Cin, device="cuda", generator=g), native.SPLIT_DTYPE)
    w = native.cast(torch.randn(Cout, 9, Cin, device="cuda", generator=g) / (3 * Cin ** 0.5), native.SPLIT_DTYPE)
    if mode == "zeros":
        x.view(torch.uint8).zero_(); w.view(torch.uint8).zero_()
    bias = torch.randn(Cout, device="cuda", generator=g)
    native.set_conv_algo(2)
    native.set_conv3x3_variant(variant)
    ts = []
    for r in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); native.conv_fwd(x, w, bias, Cout, 3, want_stats=True); e1.record(); torch.cuda.synchronize()
        if r > 1: ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    print(f"{name:8s} {t:7.3f} ms  {2.0 * 8 * H * W * Cout * 9 * Cin / t / 1e9:6.0f} TF/s-equivalent", flush=True)
