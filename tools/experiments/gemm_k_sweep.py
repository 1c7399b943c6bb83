"""Fixed cost vs slope of the pair-mode GEMM at ResNet-101's res4 shapes: time(K) for M = 22 800 rows, N = 256 / 1024
(python tools/experiments/gemm_k_sweep.py [f16x3|bf16x3])."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
dtn = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
dev = "cuda"
for M, N in ((22800, 256), (22800, 1024), (90000, 128), (90000, 512)):
    for K in (128, 256, 512, 1024, 2048, 4096):
        g = torch.Generator(device=dev).manual_seed(1)
        a = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        if dtn == "f16x3":
            a, w = native.cast(a, native.SPLITH_DTYPE), native.pack_fc_weight(w, native.F16X3)
        else:
            a, w = native.cast(a, native.SPLIT_DTYPE), native.cast(w, native.SPLIT_DTYPE)
        ts = []
        for r in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = native.conv_fwd(a, w, None, N, 1); e1.record(); torch.cuda.synchronize()
            if r > 3: ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2]
        print(f"M={M:6d} N={N:5d} K={K:5d}  {1000 * t:8.1f} us  {2.0 * M * K * N / t / 1e9:7.1f} TF/s   bytes in+out {4e-6 * (M * K + M * N):7.1f} MB -> {4e-9 * (M * K + M * N) / t:6.2f} TB/s", flush=True)
