"""Time the FC-head GEMMs (generic implicit-GEMM kernel, ksize 1) at the benchmark's shapes."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
dev = "cuda"
for (name, M, K, N) in [("fc1 teacher", 16000, 25088, 1024), ("fc1 student", 4096, 25088, 1024), ("fc1 dgrad", 4096, 1024, 25088),
                        ("fc2 teacher", 16000, 1024, 1024), ("fc2 student", 4096, 1024, 1024), ("rpn 1x1", 5328, 512, 75)]:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    ts = []
    for r in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); native.conv_fwd(a, w, None, N, 1, act=1); e1.record(); torch.cuda.synchronize()
        if r > 1: ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    print(f"{name:12s} M={M:6d} K={K:6d} N={N:6d}  {t:7.3f} ms  {2.0 * M * K * N / t / 1e9:7.1f} TF/s", flush=True)
