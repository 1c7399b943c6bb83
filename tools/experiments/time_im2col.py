import importlib, sys, torch
sys.path.insert(0, "/root/repo")
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
x = torch.zeros(8, 600, 1200, 8, device="cuda"); x[..., :3] = torch.randn(8, 600, 1200, 3, device="cuda")
for kp in (160, 192):
    ts = []
    for r in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c = native.im2col_stem(x, kp, out_dtype=native.SPLITH_DTYPE); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(kp, sorted(ts)[len(ts) // 2], "ms")
