"""Diagnostic (library built with -DM16_STAMP, tools/experiments/m16_knockout.sh, variant STAMP): where wave 0 of every
workgroup of k_conv3x3_m16<4> spends its cycles, per stage.  argv[1]: randn | zeros."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; lib = native.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "randn"
buf = (ctypes.c_ulonglong * 8)()
for name, H, W, Cin, Cout in (("conv2_2", 300, 600, 128, 128), ("conv3_2", 150, 300, 256, 256), ("conv4_2", 75, 150, 512, 512)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = native.cast(torch.randn(8, H, W, Cin, device="cuda", generator=g), native.SPLIT_DTYPE)
    w = native.cast(torch.randn(Cout, 9, Cin, device="cuda", generator=g) / (3 * Cin ** 0.5), native.SPLIT_DTYPE)
    if mode == "zeros":
        x.view(torch.uint8).zero_(); w.view(torch.uint8).zero_()
    bias = torch.randn(Cout, device="cuda", generator=g)
    native.set_conv_algo(2); native.set_conv3x3_variant(5)
    for r in range(3):
        native.conv_fwd(x, w, bias, Cout, 3, want_stats=True)
    lib.sfod_debug_m16_stamps(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); native.conv_fwd(x, w, bias, Cout, 3, want_stats=True); e1.record(); torch.cuda.synchronize()
    lib.sfod_debug_m16_stamps(buf, 1)
    v = list(buf)
    st = max(v[4], 1)
    print(f"{name:8s} {e0.elapsed_time(e1):6.3f} ms | per stage (cycles): dma_wait {v[0]/st:7.0f}  barrier {v[1]/st:7.0f}  reads+dma_issue {v[2]/st:7.0f}"
          f"  mfma_phase {v[3]/st:7.0f}  sum {sum(v[:4])/st:7.0f} | per workgroup: kernel {v[5]/max(v[6],1):9.0f} cycles, loop {sum(v[:4])/max(v[6],1):9.0f},"
          f" stages {v[4]/max(v[6],1):5.1f}, workgroups {v[6]}", flush=True)
