"""Time the generic implicit-GEMM kernel (ksize 1: FC head, 1x1 convolutions) at the benchmark's shapes.
  python tools/bench_gemm.py [--dtype bf16x3]      SFOD_GEMM_WIDE=0|1|2 selects never / auto / always the 256 x 256 tile"""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", choices=["bf16", "bf16x3", "f16x3", "fp32"], default="bf16x3")
ap.add_argument("--only", default="", help="substring filter on the shape names")
args = ap.parse_args()
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
dev = "cuda"
SHAPES = [("vgg fc1 teacher", 16000, 25088, 1024), ("vgg fc1 student", 4096, 25088, 1024), ("vgg fc1 dgrad", 4096, 1024, 25088),
          ("vgg fc2 teacher", 16000, 1024, 1024), ("vgg fc2 student", 4096, 1024, 1024), ("rpn 1x1", 5328, 512, 75),
          ("r101 fc1 teacher", 16000, 50176, 2048), ("r101 fc1 student", 2048, 50176, 2048),
          ("r101 res4 256->1024", 22800, 256, 1024), ("r101 res4 1024->256", 22800, 1024, 256),
          ("r101 res3 128->512", 90000, 128, 512), ("r101 res3 512->128", 90000, 512, 128),
          ("r101 res2 64->256", 360000, 64, 256), ("r101 res2 256->64", 360000, 256, 64),
          ("r101 res4 1024->1024 (shortcut-like)", 22800, 1024, 1024), ("r101 rpn 1x1 1024->60", 22800, 1024, 60)]
split = args.dtype == "bf16x3"
SHAPES += [("r101 res3 256->512 (shortcut)", 90000, 256, 512), ("r101 res4 512->1024 (shortcut)", 22800, 512, 1024),
           ("r101 res2 64->64", 360000, 64, 64), ("r101 stem im2col 160->64", 1440000, 160, 64), ("r101 fc2 teacher", 16000, 2048, 2048), ("r101 predictor", 16000, 2048, 48)]
for (name, M, K, N) in SHAPES:
    if args.only and args.only not in name:
        continue
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    if split:
        a, w = native.cast(a, native.SPLIT_DTYPE), native.cast(w, native.SPLIT_DTYPE)
    elif args.dtype == "f16x3":       # half pairs; the weights under their per-tensor scale
        a, w = native.cast(a, native.SPLITH_DTYPE), native.pack_fc_weight(w, native.F16X3)
    elif args.dtype == "bf16":
        a, w = a.bfloat16(), w.bfloat16()
    ts = []
    for r in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = native.conv_fwd(a, w, None, N, 1, act=1); e1.record(); torch.cuda.synchronize()
        if r > 1: ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    print(f"{name:20s} M={M:6d} K={K:6d} N={N:6d}  {t:7.3f} ms  {2.0 * M * K * N / t / 1e9:7.1f} TF/s", flush=True)
    del a, w, y
