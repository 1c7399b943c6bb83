"""Time the generic (non-patch) bf16x3 weight-gradient kernels at the benchmark's shapes: wide-tile kernel (algo 4) vs the
64 x 64 kernel (algo 3), same box.   python tools/bench_wgrad.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
dev = "cuda"
SHAPES = [("vgg fc1", 4096, 25088, 1024), ("vgg fc2", 4096, 1024, 1024), ("r101 fc1", 2048, 50176, 2048),
          ("r101 res4 1x1 a", 8 * 38 * 75, 1024, 256), ("r101 res4 1x1 b", 8 * 38 * 75, 256, 1024),
          ("r101 res3 1x1 a", 8 * 75 * 150, 512, 128), ("r101 res3 1x1 b", 8 * 75 * 150, 128, 512)]
for (name, M, K, N) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    x = native.cast(torch.randn(M, K, device=dev, generator=g), native.SPLIT_DTYPE).view(M, 1, 1, K)
    dy = native.cast(torch.randn(M, N, device=dev, generator=g) * 1e-3, native.SPLIT_DTYPE).view(M, 1, 1, N)
    row = f"{name:16s} M={M:6d} Cin={K:6d} Cout={N:5d}"
    for algo in (4, 3):
        native.set_conv_algo(algo)
        ts = []
        for r in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dw = torch.zeros(N, 1, K, device=dev)
            e0.record(); native.conv_wgrad(x, dy, N, 1, dw_packed=dw); e1.record(); torch.cuda.synchronize()
            if r > 1: ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2]
        row += f"   algo {algo}: {t:7.3f} ms {2.0 * M * K * N / t / 1e9:6.1f} TF/s"
    native.set_conv_algo(0)
    print(row, flush=True)
