"""Run-to-run determinism and cross-variant agreement of the halo-patch conv (debug helper)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sfod = importlib.import_module("simple-sfod_amd")
native = sfod.native
native.load()
dev = "cuda"
for (B, H, W, Cin, Cout) in [(8, 300, 600, 128, 128), (8, 150, 300, 256, 256), (8, 75, 150, 512, 512), (8, 300, 600, 64, 128)]:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).bfloat16()
    w = (torch.randn(Cout, 9, Cin, device=dev, generator=g) / (3 * Cin ** 0.5)).bfloat16()
    bias = torch.randn(Cout, device=dev, generator=g)
    native.set_conv_algo(1)
    ref = native.conv_fwd(x, w, bias, Cout, 3).float()
    native.set_conv_algo(2)
    outs = {}
    for v in (1, 2, 3, 4):
        native.set_conv3x3_variant(v)
        ys = [native.conv_fwd(x, w, bias, Cout, 3, want_stats=True)[0] for _ in range(6)]
        torch.cuda.synchronize()
        same = all(torch.equal(ys[0], y) for y in ys[1:])
        nbad = max(int((ys[0] != y).sum()) for y in ys[1:])
        d = (ys[0].float() - ref)
        rel = (d.norm() / ref.norm()).item()
        big = int((d.abs() > 0.05 * ref.abs().max()).sum())
        print(f"{B}x{H}x{W} {Cin}->{Cout} variant {v}: deterministic={same} (max differing elems {nbad}) rel diff vs generic {rel:.2e}, gross errors {big}", flush=True)
    native.set_conv3x3_variant(0)
    native.set_conv_algo(0)

# ---- weight gradient (halo-patch slabs) and the generic 3-stage GEMM (fc1-sized) under load ---------------
for (B, H, W, Cin, Cout) in [(8, 150, 300, 256, 256), (8, 300, 600, 64, 128), (8, 600, 1200, 64, 64)]:
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).bfloat16()
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g).bfloat16()
    native.set_conv_algo(1)
    ref = native.conv_wgrad(x, dy, Cout, 3)
    native.set_conv_algo(2)
    ds = [native.conv_wgrad(x, dy, Cout, 3) for _ in range(5)]
    native.set_conv_algo(0)
    torch.cuda.synchronize()
    same = all(torch.equal(ds[0], d) for d in ds[1:])
    print(f"wgrad {B}x{H}x{W} {Cin}->{Cout}: deterministic={same} rel diff vs generic {((ds[0] - ref).norm() / ref.norm()).item():.2e}", flush=True)
M, K, N = 16000, 25088, 1024
g = torch.Generator(device=dev).manual_seed(3)
a = torch.randn(M, K, device=dev, generator=g).bfloat16()
wt = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
ys = [native.conv_fwd(a, wt, None, N, 1, act=1) for _ in range(5)]
torch.cuda.synchronize()
ref = torch.relu(a[:2048].float() @ wt.float().t())
print(f"gemm {M}x{K}x{N}: deterministic={all(torch.equal(ys[0], y) for y in ys[1:])} "
      f"rel diff vs torch (first 2048 rows) {((ys[0][:2048].float() - ref).norm() / ref.norm()).item():.2e}", flush=True)
