"""Oracle (test infrastructure): ROIAlign (torchvision semantics, aligned=True, adaptive grid).

``roi_align`` wraps the plain-C restatement in ``oracle/csrc/roi_align.c`` (built with gcc by
``build()``) in a torch.autograd.Function so the CPU model step can back-propagate through
it; ``roi_align_py`` is an independent slow pure-Python version used to pin the C one on tiny
hand-checkable cases.  Reference call site:
``daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py:117`` (box_pooler).
PARITY UNPINNED (torchvision absent) -- see oracle/__init__.py.
"""
import ctypes
import math
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_roi_align.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "csrc", "roi_align.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SO, src, "-lm"]
        )
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        args = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        _lib.oracle_roi_align_fwd.argtypes = args
        _lib.oracle_roi_align_bwd.argtypes = args
    return _lib


class _RoiAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rois, out_size, scale, sampling_ratio, aligned):
        x = x.contiguous().float()
        rois = rois.contiguous().float()
        R = rois.shape[0]
        _, C, H, W = x.shape
        out = torch.zeros(R, C, out_size, out_size, dtype=torch.float32)
        _load().oracle_roi_align_fwd(
            x.data_ptr(), rois.data_ptr(), out.data_ptr(), R, C, H, W, out_size, out_size,
            float(scale), int(sampling_ratio), int(aligned))
        ctx.save_for_backward(rois)
        ctx.meta = (x.shape, out_size, scale, sampling_ratio, aligned)
        return out

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        shape, out_size, scale, sampling_ratio, aligned = ctx.meta
        g = g.contiguous().float()
        gi = torch.zeros(shape, dtype=torch.float32)
        _, C, H, W = shape
        _load().oracle_roi_align_bwd(
            g.data_ptr(), rois.data_ptr(), gi.data_ptr(), rois.shape[0], C, H, W, out_size,
            out_size, float(scale), int(sampling_ratio), int(aligned))
        return gi, None, None, None, None, None


def roi_align(x, rois, out_size=7, scale=1.0 / 32, sampling_ratio=0, aligned=True):
    """x [B,C,H,W] fp32, rois [R,5] (batch_idx,x1,y1,x2,y2) -> [R,C,out,out]."""
    return _RoiAlign.apply(x, rois, out_size, scale, sampling_ratio, aligned)


def roi_align_py(x, rois, out_size=7, scale=1.0 / 32, sampling_ratio=0, aligned=True):
    """Slow independent reference (double loops, fp64 accumulate) for tiny pinning cases."""
    B, C, H, W = x.shape
    R = rois.shape[0]
    out = torch.zeros(R, C, out_size, out_size, dtype=torch.float64)
    xd = x.double()
    for n in range(R):
        b = int(rois[n, 0])
        off = 0.5 if aligned else 0.0
        sw = float(rois[n, 1]) * scale - off
        sh = float(rois[n, 2]) * scale - off
        ew = float(rois[n, 3]) * scale - off
        eh = float(rois[n, 4]) * scale - off
        rw, rh = ew - sw, eh - sh
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / out_size, rw / out_size
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / out_size))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / out_size))
        count = max(gh * gw, 1)
        for ph in range(out_size):
            for pw in range(out_size):
                acc = torch.zeros(C, dtype=torch.float64)
                for iy in range(gh):
                    y = sh + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        xx = sw + pw * bw + (ix + 0.5) * bw / gw
                        yy = y
                        if yy < -1.0 or yy > H or xx < -1.0 or xx > W:
                            continue
                        yy = max(yy, 0.0)
                        xx = max(xx, 0.0)
                        yl, xl = int(yy), int(xx)
                        if yl >= H - 1:
                            yh = yl = H - 1
                            yy = float(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xh = xl = W - 1
                            xx = float(xl)
                        else:
                            xh = xl + 1
                        ly, lx = yy - yl, xx - xl
                        hy, hx = 1 - ly, 1 - lx
                        acc += (hy * hx * xd[b, :, yl, xl] + hy * lx * xd[b, :, yl, xh]
                                + ly * hx * xd[b, :, yh, xl] + ly * lx * xd[b, :, yh, xh])
                out[n, :, ph, pw] = acc / count
    return out.float()
