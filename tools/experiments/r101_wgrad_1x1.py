"""ResNet-101-C4 bottleneck 1x1 weight gradients (B = 8, 600 x 1200: res3 on 75 x 150, res4 on 38 x 75; bf16-pair backward
products) on the two generic weight-gradient kernels: algo 3 = always the 64 x 64 tile (k_conv_wgrad), algo 4 = always the
128 x 256 tile (k_conv_wgrad_x3w), 0 = the planner's choice.  Interleaved rounds; the results of 3 and 4 are compared
(summation order differs: relative L2)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
shapes = (("res3 256->128 (first)", 75, 150, 256, 128), ("res3 512->128", 75, 150, 512, 128), ("res3 128->512", 75, 150, 128, 512),
          ("res3 256->512 shortcut", 75, 150, 256, 512),
          ("res4 512->256 (first)", 38, 75, 512, 256), ("res4 1024->256", 38, 75, 1024, 256), ("res4 256->1024", 38, 75, 256, 1024),
          ("res4 512->1024 shortcut", 38, 75, 512, 1024), ("rpn/head 1024->1024", 38, 75, 1024, 1024))
algos = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 3, 4]
for lname, H, W, Cin, Cout in shapes:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = native.cast(torch.randn(8, H, W, Cin, device="cuda", generator=g), native.SPLIT_DTYPE)
    dy = native.cast(torch.randn(8, H, W, Cout, device="cuda", generator=g), native.SPLIT_DTYPE)
    ts = {a: [] for a in algos}
    outs = {}
    for r in range(9):
        for a in algos:
            native.set_conv_algo(a)
            dw = torch.zeros(Cout, 1, Cin, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); native.conv_wgrad(x, dy, Cout, 1, dw_packed=dw); e1.record(); torch.cuda.synchronize()
            if r > 1: ts[a].append(e0.elapsed_time(e1))
            outs[a] = dw
    native.set_conv_algo(0)
    fl = 2.0 * 8 * H * W * Cout * Cin
    ref = outs[algos[0]].double()
    rel = {a: ((outs[a].double() - ref).norm() / ref.norm()).item() for a in algos}
    print(f"{lname:26s} {fl/1e9:6.2f} GF | " + " | ".join(f"algo {a}: {sorted(t)[len(t)//2]*1e3:6.1f} us {fl/sorted(t)[len(t)//2]/1e9:4.0f} TF/s (rel {rel[a]:.1e})" for a, t in ts.items()), flush=True)
