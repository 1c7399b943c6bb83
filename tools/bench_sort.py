"""Time sfod_segmented_sort_desc at the step's shapes.  python tools/bench_sort.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
for (B, n) in [(8, 9990), (8, 12000), (8, 16000), (1, 9990), (8, 4096), (8, 30720), (8, 34200), (8, 65536), (8, 98304), (2, 200000)]:
    g = torch.Generator(device="cuda").manual_seed(n)
    keys = torch.randn(B, n, generator=g, device="cuda")
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); native.segmented_sort_desc(keys); e1.record(); torch.cuda.synchronize()
        if r > 1: ts.append(e0.elapsed_time(e1))
    print(f"B={B} n={n:6d}: {sorted(ts)[len(ts)//2]*1e3:7.1f} us")
