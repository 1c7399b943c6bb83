"""Times the 1x1 convolutions of the ResNet-101-C4 trunk at B = 8, 600x1200 (with BatchNorm statistics, as the model calls
them) with whatever library SFOD_HIP_LIB names (no checks).  argv[1]: f16x3 | bf16x3"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
for name, M, K, N in (("res4 256->1024", 22800, 256, 1024), ("res4 1024->256", 22800, 1024, 256),
                      ("res4 512->1024 sc", 22800, 512, 1024), ("res3 128->512", 90000, 128, 512),
                      ("res3 512->128", 90000, 512, 128), ("res3 256->512 sc", 90000, 256, 512)):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(8, M // 8, 1, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    if mode == "bf16x3":
        a, w = native.cast(a, native.SPLIT_DTYPE), native.cast(w, native.SPLIT_DTYPE)
    else:
        a, w = native.cast(a, native.SPLITH_DTYPE), native.pack_fc_weight(w, native.F16X3)
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); native.conv_fwd(a, w, None, N, 1, want_stats=True); e1.record(); torch.cuda.synchronize()
        if r > 2: ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    print(f"{name:18s} M={M:6d} K={K:5d} N={N:5d}  {t * 1e3:7.1f} us  {2.0 * M * K * N / t / 1e9:6.0f} TF/s-equivalent", flush=True)
