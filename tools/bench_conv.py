"""Per-layer A/B of the 3x3 convolution kernels on the VGG16 shapes of the benchmark config
(B images of 600x1200, or --res full): generic implicit GEMM (algo 1) vs halo-patch (algo 2),
interleaved rounds in one process, random bf16 operands.  Prints TFLOP/s per layer and kernel.
  python tools/bench_conv.py [--batch 8] [--res r600|full] [--rounds 5] [--wgrad]
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--res", default="r600")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--layers", default="")
    ap.add_argument("--dtype", choices=["bf16", "bf16x3"], default="bf16x3")
    ap.add_argument("--variants", action="store_true",
                    help="time the halo-patch kernel's workgroup shapes 1..4 (and auto = 0) instead of algo 1 vs 2")
    ap.add_argument("--data", choices=["randn", "zeros", "relu"], default="randn",
                    help="operand values: zeros shows how much of a kernel's time is the clock the chip holds under load "
                         "(same instruction stream, less switching power); relu = half of x's entries zero like a ReLU output")
    ap.add_argument("--w3pipe", action="store_true",
                    help="halo-patch weight gradient: round-2 chunk loop vs the pipelined one, interleaved (sfod_set_wgrad3x3_pipe)")
    args = ap.parse_args()
    sfod = importlib.import_module("simple-sfod_amd")
    native = sfod.native
    native.load()
    H0, W0 = (600, 1200) if args.res == "r600" else (1024, 2048)
    B = args.batch
    layers = [  # name, H, W, Cin, Cout
        ("conv1_1", H0, W0, 8, 64), ("conv1_2", H0, W0, 64, 64), ("conv2_1", H0 // 2, W0 // 2, 64, 128), ("conv2_2", H0 // 2, W0 // 2, 128, 128),
        ("conv3_1", H0 // 4, W0 // 4, 128, 256), ("conv3_2", H0 // 4, W0 // 4, 256, 256),
        ("conv4_1", H0 // 8, W0 // 8, 256, 512), ("conv4_2", H0 // 8, W0 // 8, 512, 512),
        ("conv5_1", H0 // 16, W0 // 16, 512, 512), ("rpn", H0 // 32, W0 // 32, 512, 512),
        ("dgrad2_1", H0 // 2, W0 // 2, 128, 64), ("dgrad3_1", H0 // 4, W0 // 4, 256, 128), ("dgrad1_2", H0, W0, 64, 64),
    ]
    if args.layers:
        keep = set(args.layers.split(","))
        layers = [l for l in layers if l[0] in keep]
    dev = "cuda"
    for name, H, W, Cin, Cout in layers:
        g = torch.Generator(device=dev).manual_seed(1)
        odt = torch.bfloat16 if args.dtype == "bf16" else native.SPLIT_DTYPE
        x = native.cast(torch.randn(B, H, W, Cin, device=dev, generator=g), odt)
        w = native.cast(torch.randn(Cout, 9, Cin, device=dev, generator=g) / (3 * Cin ** 0.5), odt)
        bias = torch.randn(Cout, device=dev, generator=g)
        if args.data == "zeros":
            x.view(torch.uint8).zero_()
            w.view(torch.uint8).zero_()
        elif args.data == "relu":
            x = native.cast(torch.relu(torch.randn(B, H, W, Cin, device=dev, generator=g)), odt)
        flops = 2.0 * B * H * W * Cout * 9 * Cin
        if args.w3pipe:
            if name == "conv1_1":
                continue
            dy = native.cast(torch.randn(B, H, W, Cout, device=dev, generator=g), odt)
            native.set_conv_algo(2)
            ts, dws = {0: [], 1: [], 2: []}, {}
            for r in range(args.rounds + 1):
                for v in ts:
                    native.set_wgrad3x3_pipe(v)
                    dw = torch.zeros(Cout, 9, Cin, dtype=torch.float32, device=dev)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    native.conv_wgrad(x, dy, Cout, 3, dw)
                    e1.record()
                    torch.cuda.synchronize()
                    if r > 0:
                        ts[v].append(e0.elapsed_time(e1))
                    dws[v] = dw
            native.set_wgrad3x3_pipe(2)
            native.set_conv_algo(0)
            line = f"{name:9s} {B}x{H}x{W} {Cin:4d}->{Cout:4d} {flops / 1e9:8.1f} GF wgrad"
            for v in ts:
                t = sorted(ts[v])[len(ts[v]) // 2]
                line += f" | {('round 2', 'pipelined', '64x64 block')[v]} {t:6.3f} ms {flops / t / 1e9:6.0f} TF/s"
            d2 = ((dws[2] - dws[0]).double().norm() / dws[0].double().norm()).item()
            print(line + f" | 0 == 1: {torch.equal(dws[0], dws[1])}, rel diff of 2: {d2:.1e}", flush=True)
            continue
        if args.variants:
            native.set_conv_algo(2)
            vt = {v: [] for v in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9)}
            for r in range(args.rounds + 1):
                for v in vt:
                    native.set_conv3x3_variant(v)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    native.conv_fwd(x, w, bias, Cout, 3, want_stats=True)
                    e1.record()
                    torch.cuda.synchronize()
                    if r > 0:
                        vt[v].append(e0.elapsed_time(e1))
            native.set_conv3x3_variant(0)
            native.set_conv_algo(0)
            names = {0: "auto", 1: "512x128", 2: "256x128", 3: "256x64", 4: "512x64", 5: "256x128 m16", 6: "256x128 m16 4w",
                     7: "256x64 m16 4w", 8: "512x64 m16", 9: "256x64 m16 8w"}
            line = f"{name:9s} {B}x{H}x{W} {Cin:4d}->{Cout:4d} {flops / 1e9:8.1f} GF"
            for v in vt:
                t = sorted(vt[v])[len(vt[v]) // 2]
                line += f" | {names[v]} {t:6.3f} ms {flops / t / 1e9:6.0f}"
            print(line, flush=True)
            continue
        res = {}
        algos = [1, 2]
        times = {a: [] for a in algos}
        outs = {}
        for r in range(args.rounds + 1):
            for a in algos:
                native.set_conv_algo(a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                y, st = native.conv_fwd(x, w, bias, Cout, 3, want_stats=True)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[a].append(e0.elapsed_time(e1))
                outs[a] = y
        native.set_conv_algo(0)
        err = ((outs[1].float() - outs[2].float()).norm() / outs[1].float().norm()).item()
        line = f"{name:9s} {B}x{H}x{W} {Cin:4d}->{Cout:4d} {flops / 1e9:8.1f} GF"
        for a in algos:
            t = sorted(times[a])[len(times[a]) // 2]
            line += f" | algo{a} {t:7.3f} ms {flops / t / 1e9:7.1f} TF/s"
        line += f" | rel diff {err:.2e}"
        print(line, flush=True)
        if args.wgrad:
            dy = native.cast(torch.randn(B, H, W, Cout, device=dev, generator=g), odt)
            ts = {1: [], 2: []}
            dws = {}
            for r in range(args.rounds + 1):
                for al in (1, 2):
                    native.set_conv_algo(al)
                    dw = torch.zeros(Cout, 9, Cin, dtype=torch.float32, device=dev)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    native.conv_wgrad(x, dy, Cout, 3, dw)
                    e1.record()
                    torch.cuda.synchronize()
                    if r > 0:
                        ts[al].append(e0.elapsed_time(e1))
                    dws[al] = dw
            native.set_conv_algo(0)
            err = ((dws[1] - dws[2]).norm() / dws[1].norm()).item()
            line = f"{'':9s} wgrad"
            for al in (1, 2):
                t = sorted(ts[al])[len(ts[al]) // 2]
                line += f" | algo{al} {t:7.3f} ms {flops / t / 1e9:7.1f} TF/s"
            print(line + f" | rel diff {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
