"""Oracle (test infrastructure): CPU restatement of the split-precision product the MFMA kernels compute
(include/sfod_hip.h, SFOD_BF16X3 / SFOD_F16X3) -- NOT of anything in the reference, whose arithmetic is plain fp32: this
pins what the device modes are DEFINED to compute, so that a kernel is checked against its own definition (agreement at
the level of the fp32 accumulation order, ~1e-7) as well as against the exact product (tests/test_gpu_bf16x3.py,
tests/test_gpu_f16x3.py).  Only tests/ may import this module.

    operand v  ->  hi = rnd16(v), lo = rnd16(v - hi)            rnd16 = bfloat16 or IEEE half (round to nearest even)
    product    ->  hi_a * hi_b + hi_a * lo_b + lo_a * hi_b      (lo_a * lo_b dropped), summed over K in high precision here
    f16x3 weights: stored as w * s, s = the power of two with max|w| * s in [2^13, 2^14); the result is multiplied by 1 / s
"""
import math

import torch


def split_pairs(v, fmt):
    """fp32 tensor -> (hi, lo) as fp64 tensors holding the exactly representable 16-bit values."""
    v = v.float()
    if fmt == "bf16":
        hi = v.bfloat16().float()
        lo = (v - hi).bfloat16().float()
    elif fmt == "f16":
        c = v.clamp(-65504.0, 65504.0)          # beyond half's range (infinities too): exactly +-65504, lo = 0; NaN stays NaN
        hi = c.half().float()
        lo = (c - hi).half().float()
    else:
        raise ValueError(fmt)
    return hi.double(), lo.double()


def weight_scale(w):
    """the per-tensor power of two of the SFOD_F16X3 weight packers (csrc/common.h wscale_from_absmax)."""
    amax = float(w.abs().max())
    if amax == 0.0 or not math.isfinite(amax):
        return 1.0
    return 2.0 ** (13 - math.floor(math.log2(amax)))


def linear(x, w, fmt):
    """x [M, K], w [N, K] fp32 -> x @ w.T as the split-precision modes define it (fp64 accumulation)."""
    s = weight_scale(w) if fmt == "f16" else 1.0
    xh, xl = split_pairs(x, fmt)
    wh, wl = split_pairs(w.float() * s, fmt)
    return (xh @ wh.t() + xh @ wl.t() + xl @ wh.t()) / s
