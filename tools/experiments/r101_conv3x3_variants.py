"""ResNet-101-C4 3x3 bottleneck convolutions (res3: 8 x 75 x 150, 128 -> 128; res4: 8 x 38 x 75, 256 -> 256; f16x3, with
BatchNorm statistics) on every workgroup shape of the halo-patch kernel and on the generic implicit GEMM, interleaved rounds."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
names = {0: "auto", 1: "512x128", 2: "256x128", 3: "256x64", 4: "512x64", 5: "256x128 m16", 7: "256x64 m16 4w", 9: "256x64 m16 8w", -1: "generic GEMM"}
for lname, H, W, C in (("res3 3x3", 75, 150, 128), ("res4 3x3", 38, 75, 256)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = native.cast(torch.randn(8, H, W, C, device="cuda", generator=g), native.SPLITH_DTYPE)
    w = native.pack_conv_weight(torch.randn(C, C, 3, 3, device="cuda", generator=g) / (3 * C ** 0.5), C, native.F16X3)
    ts = {v: [] for v in names}
    for r in range(8):
        for v in names:
            native.set_conv_algo(1 if v == -1 else 2)
            native.set_conv3x3_variant(max(v, 0))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); native.conv_fwd(x, w, None, C, 3, want_stats=True); e1.record(); torch.cuda.synchronize()
            if r > 0: ts[v].append(e0.elapsed_time(e1))
    native.set_conv_algo(0); native.set_conv3x3_variant(0)
    fl = 2.0 * 8 * H * W * C * 9 * C
    print(lname, " | ".join(f"{names[v]} {sorted(t)[len(t)//2]*1e3:6.1f} us {fl/sorted(t)[len(t)//2]/1e9:4.0f}" for v, t in ts.items()), flush=True)
