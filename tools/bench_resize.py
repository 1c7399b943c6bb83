"""Time sfod_resize_bilinear_u8 on the benchmark's frame (1024x2048 -> 600x1200, uint8 RGB).  python tools/bench_resize.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
g = torch.Generator(device="cuda").manual_seed(0)
img = torch.randint(0, 256, (3, 1024, 2048), generator=g, device="cuda", dtype=torch.uint8)
for (h, w) in [(600, 1200), (512, 1024), (1024, 2048)]:
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = native.resize_bilinear_u8(img, h, w, flip=bool(r & 1)); e1.record(); torch.cuda.synchronize()
        if r > 1: ts.append(e0.elapsed_time(e1))
    print(f"1024x2048 -> {h}x{w}: {sorted(ts)[len(ts)//2]*1e3:7.1f} us")
