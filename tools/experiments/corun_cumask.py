"""Experiment: partition the chip with CU-masked streams (hipExtStreamCreateWithCUMask) -- the matrix kernels of the step on
most of the CUs, the HBM-bound BatchNorm-class passes on the rest -- instead of letting two full-chip streams take turns.

The step is power-bound on its MFMA kernels (1288 W of 1400 at 1.93 GHz) and 15 % of it are HBM-bound passes that a matrix
kernel's workgroups leave no room for (corun_conv_bn.py: only 16-26 % of a BatchNorm pass hides beside a convolution: tails).
Question: (1) what does a convolution lose on 224 / 192 / 160 CUs (if power bounds it: little -- the clock rises);
(2) what HBM rate does a BatchNorm apply pass reach on 32 / 64 / 96 CUs; (3) both together on complementary masks.
Mask bit i = CU i; KFD spreads consecutive bits round-robin over the 8 XCDs, so a contiguous range is XCD-balanced."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
dev = "cuda"
torch.zeros(1, device=dev)
NCU = torch.cuda.get_device_properties(0).multi_processor_count
print("CUs:", NCU, flush=True)


def masked_stream(cus):
    """stream restricted to the CU indices in ``cus``"""
    words = (NCU + 31) // 32
    m = [0] * words
    for c in cus:
        m[c // 32] |= 1 << (c % 32)
    arr = (ctypes.c_uint32 * words)(*m)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    return torch.cuda.ExternalStream(s.value)


g = torch.Generator(device=dev).manual_seed(1)
B, H, W, Cin, Cout = 8, 150, 300, 256, 256
x = native.cast(torch.relu(torch.randn(B, H, W, Cin, device=dev, generator=g)), native.SPLIT_DTYPE)
w = native.cast(torch.randn(Cout, 9, Cin, device=dev, generator=g) / (3 * Cin ** 0.5), native.SPLIT_DTYPE)
bias = torch.randn(Cout, device=dev, generator=g)
yb = torch.randn(8, 300, 600, 128, device=dev, generator=g)
mean, invstd = yb.mean(dim=(0, 1, 2)), torch.rsqrt(yb.var(dim=(0, 1, 2)) + 1e-5)
gamma, beta = torch.ones(128, device=dev), torch.zeros(128, device=dev)
NA, NB = 12, 24
bn_bytes = NB * yb.numel() * 8            # read fp32, write pairs
conv_flop = NA * 2.0 * B * H * W * Cin * Cout * 9


def run(sa, sb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record()
    for s in (sa, sb):
        if s is not None:
            s.wait_stream(cur)
    if sa is not None:
        with torch.cuda.stream(sa):
            for _ in range(NA):
                native.conv_fwd(x, w, bias, Cout, 3, want_stats=True)
    if sb is not None:
        with torch.cuda.stream(sb):
            for _ in range(NB):
                native.bn_relu_pool_fwd(yb, mean, invstd, gamma, beta, False, out_dtype=native.SPLIT_DTYPE)
    for s in (sa, sb):
        if s is not None:
            cur.wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def best(sa, sb, n=3):
    run(sa, sb)
    return min(run(sa, sb) for _ in range(n))


full_a, full_b = torch.cuda.Stream(), torch.cuda.Stream()
ta, tb = best(full_a, None), best(None, full_b)
tab = best(full_a, full_b)
print(f"full chip : conv {ta:7.3f} ms ({conv_flop / ta / 1e9:6.1f} TF/s) | bn {tb:7.3f} ms ({bn_bytes / tb / 1e9:5.2f} TB/s) | "
      f"two full-chip streams together {tab:7.3f} ms (sum {ta + tb:7.3f})", flush=True)
for small in (32, 64, 96, 128):
    big = NCU - small
    s_small = masked_stream(range(0, small))
    s_big = masked_stream(range(small, NCU))
    ca, cb = best(s_big, None), best(None, s_small)
    cab = best(s_big, s_small)
    print(f"{big:3d} + {small:3d} : conv on {big} CUs {ca:7.3f} ms ({conv_flop / ca / 1e9:6.1f} TF/s, x{ca / ta:.3f}) | bn on {small} CUs {cb:7.3f} ms "
          f"({bn_bytes / cb / 1e9:5.2f} TB/s) | together {cab:7.3f} ms = max-model {max(ca, cb):7.3f}, vs serial full-chip {ta + tb:7.3f} "
          f"({100 * (1 - cab / (ta + tb)):+.1f} %), vs two full-chip streams {tab:7.3f}", flush=True)
# balance: how much bn work fits beside the conv work?  conv 12 launches ~ fixed; scale bn launches so both end together
