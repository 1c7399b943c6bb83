"""ctypes binding of ``libsfod_hip.so`` (the C-ABI drop-in boundary, ``include/sfod_hip.h``).

The prototypes are parsed from the header itself, so Python argument types can never drift
from the declared C ABI.  There is NO CPU fallback: if the library is missing or fails to
load, importing any compute entry point raises -- the product path fails loudly.

PyTorch is used here only as the owner of device memory and streams (``tensor.data_ptr()``,
``torch.cuda.current_stream()``).
"""
import ctypes
import os
import re
import warnings

import torch

# SFOD_BF16X3 tensors are tagged with torch.complex32 (see SPLIT_DTYPE below); torch only ever allocates / views them
warnings.filterwarnings("ignore", message="ComplexHalf support is experimental")

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "sfod_hip.h")
SO_PATH = os.environ.get("SFOD_HIP_LIB", os.path.join(_HERE, "lib", "libsfod_hip.so"))   # override: kernel A/B builds

F32, BF16, BF16X3, F16X3 = 0, 1, 2, 3

# Storage tag of SFOD_BF16X3 ("split") tensors (include/sfod_hip.h): 4 bytes per logical element -- a (hi, lo) bf16
# pair, stored per 8 channels as 8 hi then 8 lo.  torch never does arithmetic on them (only empty / zeros / view /
# permute); torch.complex32 is used purely as a 4-byte dtype that cannot be confused with fp32 data, so tensor
# shapes stay the LOGICAL shapes and every wrapper below dispatches on ``dt_of(t)``.  Conversions go through
# ``cast`` (sfod_cast), never through ``Tensor.to``.
SPLIT_DTYPE = torch.complex32
# SFOD_F16X3 tensors -- the same storage with IEEE half pairs -- carry their own 4-byte tag, so that a tensor of one pair
# format can never reach a kernel instantiated for the other (the weight-gradient entry points reject SFOD_F16X3).
SPLITH_DTYPE = torch.uint32
PAIR_DTYPES = (SPLIT_DTYPE, SPLITH_DTYPE)

# cfg.SFOD.COMPUTE_DTYPE -> operand type of the FORWARD products.  "f16x3": forward products on half pairs (22-bit
# operands; activations and weights sit inside half's exponent range), backward products (data and weight gradients:
# tiny values, no loss scaling) on bf16 pairs -- ``grad_dtype_of``.
COMPUTE_MODES = {"fp32": F32, "bf16": BF16, "bf16x3": BF16X3, "f16x3": F16X3}


def mode_dt(name):
    """cfg.SFOD.COMPUTE_DTYPE -> dt code of the MFMA operands."""
    try:
        return COMPUTE_MODES[str(name).lower()]
    except KeyError:
        raise ValueError(f"SFOD.COMPUTE_DTYPE must be one of {sorted(COMPUTE_MODES)}, got {name!r}") from None


def mode_dtype(name):
    """cfg.SFOD.COMPUTE_DTYPE -> torch dtype of the MFMA operand tensors (activations fed to conv / GEMM, weights)."""
    return torch_dtype(mode_dt(name))


def out_dtype_of(dtype):
    """dtype convolutions / GEMMs write (and elementwise gradients flow in) for operands of ``dtype``."""
    return torch.float32 if dtype in PAIR_DTYPES else dtype


def grad_dtype_of(dtype):
    """operand dtype of the BACKWARD products (data gradient, weight gradient) of a model whose forward operands are
    ``dtype``: half pairs hold activations and weights, not gradients (range) -> bf16 pairs."""
    return SPLIT_DTYPE if dtype == SPLITH_DTYPE else dtype


def is_pairs(dtype):
    return dtype in PAIR_DTYPES


class NativeLibraryError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(sfod_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int)
        argtypes = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                elif a.startswith("int64_t"):
                    argtypes.append(ctypes.c_int64)
                elif a.startswith("float"):
                    argtypes.append(ctypes.c_float)
                elif a.startswith("int") or a.startswith("int32_t"):
                    argtypes.append(ctypes.c_int)
                else:
                    raise NativeLibraryError(f"cannot map C parameter '{a}' of {name}")
        protos[name] = (restype, argtypes)
    return protos


_lib = None
_protos = None


def load(path=SO_PATH):
    """Load the shared library and bind every symbol the header declares."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} not found: build it with `python {os.path.join(_HERE, 'csrc', 'build.py')}` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
    lib = ctypes.CDLL(path)
    protos = parse_header()
    for name, (restype, argtypes) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{path} does not export {name} declared in {HEADER}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib, _protos = lib, protos
    return lib


def exported_symbols():
    load()
    return sorted(_protos)


def _chk(rc, name):
    if rc != 0:
        msg = _lib.sfod_last_error().decode()
        raise NativeLibraryError(f"{name} failed with code {rc}: {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # torch.cuda.current_stream() costs ~8 us of Python per call; the raw getter is a single C call
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "native ops need device tensors"
    assert t.is_contiguous(), "native ops need contiguous tensors"
    return t.data_ptr()


def dt_of(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == SPLIT_DTYPE:
        return BF16X3
    if t.dtype == SPLITH_DTYPE:
        return F16X3
    raise TypeError(f"unsupported dtype {t.dtype}")


def dt_of_dtype(dtype):
    return {torch.float32: F32, torch.bfloat16: BF16, SPLIT_DTYPE: BF16X3, SPLITH_DTYPE: F16X3}[dtype]


def torch_dtype(dt):
    return {F32: torch.float32, BF16: torch.bfloat16, BF16X3: SPLIT_DTYPE, F16X3: SPLITH_DTYPE}[dt]


def chunk_elems(dt):
    """channel granularity of a tensor of storage type dt (one 16-byte chunk; bf16x3: one 32-byte (hi | lo) group)"""
    return 4 if dt == F32 else 8


def nhwc_operand(x_nchw, dtype):
    """NCHW-shaped view of NHWC memory (what the backbones return) -> contiguous NHWC MFMA operand of ``dtype``."""
    p = x_nchw.permute(0, 2, 3, 1)
    if p.dtype == dtype:
        return p.contiguous()
    if dtype in PAIR_DTYPES or p.dtype in PAIR_DTYPES:
        return cast(p.contiguous(), dtype)
    return p.to(dtype).contiguous()


def as_operand(t, dtype):
    """t as an MFMA operand of ``dtype``: unchanged if it already is, else converted by sfod_cast (fp32 -> bf16 /
    operand pairs; half pairs -> bf16 pairs for a weight gradient whose producer did not write both).  The trunk's
    producers write operand tensors directly; the heads' small fp32 tensors pass here."""
    return t if t.dtype == dtype else cast(t, dtype)


def operands_for(t, dtype, need_grad):
    """fp32 ``t`` -> (forward operand of ``dtype``, the operand the weight gradient will read or None).  One pass over
    ``t`` also when the two differ ("f16x3": half pairs + bf16 pairs)."""
    gdt = grad_dtype_of(dtype)
    if not need_grad:
        return as_operand(t, dtype), None
    if gdt == dtype:
        op = as_operand(t, dtype)
        return op, op
    if t.dtype == torch.float32 and dtype == SPLITH_DTYPE and gdt == SPLIT_DTYPE:
        t = t.contiguous()
        a = torch.empty(t.shape, dtype=SPLITH_DTYPE, device=t.device)
        b = torch.empty(t.shape, dtype=SPLIT_DTYPE, device=t.device)
        call("sfod_cast_pairs_both", t, a, b, t.numel())
        return a, b
    return as_operand(t, dtype), as_operand(t, gdt)


class _Everything:
    def __contains__(self, item):
        return True


class KernelTimer:
    """Optional per-entry-point timing with HIP events recorded on the launch stream (used by
    bench.py for the roofline figure of the dominant kernels).  ``flops`` is the algorithmic work
    of the launch (2 x MACs of the GEMM the entry point computes)."""

    def __init__(self, watch=("sfod_conv_fwd", "sfod_conv_wgrad", "sfod_conv_wgrad_oihw")):
        # entry points; records are keyed "<entry>[:<kernel tag>]".  watch=None: EVERY entry point (bench.py's
        # `native_kernel_time` figure: how much of a step's wall time the library's own launches cover)
        self.watch = _Everything() if watch is None else set(watch)
        self.records = []      # (name, flops, start_event, end_event)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, a, b in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0})
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
            d["flops"] += flops
        return out


_timer = None
_pending_flops = 0.0
_pending_tag = ""


def set_timer(timer):
    global _timer
    _timer = timer


def call(name, *args, timer_name=None):
    """Raw call: tensors are converted to device pointers, the current stream is appended.  ``timer_name``: the entry
    point a KernelTimer files this launch under (sfod_conv_dgrad_bnred runs the sfod_conv_fwd kernel family)."""
    global _pending_flops
    lib = load()
    conv = [(_p(a) if isinstance(a, torch.Tensor) else a) for a in args]
    if _timer is not None and (timer_name or name) in _timer.watch:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = getattr(lib, name)(*conv, _stream())
        b.record()
        _timer.records.append(((timer_name or name) + _pending_tag, _pending_flops, a, b))
        _pending_flops = 0.0
    else:
        rc = getattr(lib, name)(*conv, _stream())
    _chk(rc, name)


def query(name, *args):
    return getattr(load(), name)(*args)


def set_conv_algo(algo):
    """0 auto, 1 generic implicit GEMM, 2 halo-patch kernel whenever the shape allows (3x3 bf16)."""
    load().sfod_set_conv_algo(int(algo))


def set_wgrad3x3_pipe(on):
    """bf16x3 halo-patch weight gradient: 2 = 64 x 64-block kernel on 128-pixel tiles (default), 1 = pipelined 64 x 32-block
    loop, 0 = the round-2 loop (A/B knob, include/sfod_hip.h)."""
    load().sfod_set_wgrad3x3_pipe(int(on))


def set_gemm_tile(tile):
    """Tile shape of the generic implicit-GEMM kernel: 0 the planner's choice, 1..10 see include/sfod_hip.h (A/B runs, tests)."""
    load().sfod_set_gemm_tile(int(tile))


def set_conv3x3_variant(variant):
    """Workgroup shape of the halo-patch kernel: 0 auto, 1..5 see include/sfod_hip.h (A/B runs, tests)."""
    load().sfod_set_conv3x3_variant(int(variant))


def set_deterministic(on):
    """No float atomics in weight / bias gradients (include/sfod_hip.h: sfod_set_deterministic); -> previous setting."""
    lib = load()
    prev = bool(lib.sfod_get_deterministic())
    lib.sfod_set_deterministic(int(bool(on)))
    return prev


def set_bn_finalize_fused(on):
    """One-launch BatchNorm finalize for layers of <= 64 statistics blocks (default on; include/sfod_hip.h)."""
    load().sfod_set_bn_finalize_fused(int(bool(on)))


def set_conv3x3_m16(on):
    """Automatic shape choice: run the 256 x 128 shape on 16x16x32 MFMAs (default on; include/sfod_hip.h)."""
    load().sfod_set_conv3x3_m16(int(on))       # 0 off, 1 the 8-wave form, 2 the 4-wave form


# =================================================================================================
# tensor-level wrappers (allocate outputs with torch, call the C ABI)
# =================================================================================================
_const_cache = {}


def dev_const(values, dtype, device):
    """Small host-side constants (image sizes, pointer tables) as device tensors, cached by value: a
    pageable host-to-device copy is synchronous and would stall the launch queue every time it recurs."""
    key = (values, dtype, str(device))
    t = _const_cache.get(key)
    if t is None:
        if len(_const_cache) > 4096:
            _const_cache.clear()
        t = torch.tensor(values, dtype=dtype).to(device)
        _const_cache[key] = t
    return t


_PIN_SLOTS, _PIN_WIDTH = 32, 64
_pin_ring = {}


def _pinned_upload_i64(values, dev):
    """Small int64 host vector -> device tensor through a ring of pinned staging slots (asynchronous
    host-to-device copy on the current stream).  A slot is reused only after the copy that read it has
    completed (its event), so the host may run any number of steps ahead."""
    n = len(values)
    if n > _PIN_WIDTH:
        return torch.tensor(values, dtype=torch.int64).to(dev)
    st = _pin_ring.get(dev)
    if st is None:
        st = {"buf": torch.empty(_PIN_SLOTS, _PIN_WIDTH, dtype=torch.int64).pin_memory(), "ev": [None] * _PIN_SLOTS,
              "next": 0}
        _pin_ring[dev] = st
    i = st["next"]
    st["next"] = (i + 1) % _PIN_SLOTS
    if st["ev"][i] is not None:
        st["ev"][i].synchronize()
    slot = st["buf"][i, :n]
    slot.copy_(torch.tensor(values, dtype=torch.int64))
    out = slot.to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st["ev"][i] = ev
    return out


def preprocess(images_u8, Hp, Wp, cpad, mean, std, dt):
    """images_u8: list of uint8 device tensors [3,h,w] -> (x [B,Hp,Wp,cpad], sizes int32 [B,2])."""
    dev = images_u8[0].device
    B = len(images_u8)
    imgs = [im.contiguous() for im in images_u8]
    # pointer table: written into a pinned staging slot and copied asynchronously (a pageable copy blocks
    # the host until the stream drains, and the image addresses change whenever the allocator hands out
    # another block, so caching them by value still missed every other step)
    ptrs = _pinned_upload_i64([im.data_ptr() for im in imgs], dev)
    sizes = dev_const(tuple((int(im.shape[1]), int(im.shape[2])) for im in imgs), torch.int32, dev)
    out = torch.empty(B, Hp, Wp, cpad, dtype=torch_dtype(dt), device=dev)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    lib = load()
    rc = lib.sfod_preprocess(ptrs.data_ptr(), sizes.data_ptr(), B, Hp, Wp, cpad,
                             ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
                             out.data_ptr(), dt, _stream())
    _chk(rc, "sfod_preprocess")
    out._keepalive = (imgs, ptrs)
    return out, sizes


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs (BILINEAR, box = the whole axis) + normalize_coeffs_8bpc, same float64
    operation order: -> (bounds int32 [out,2] = (first index, count), coeffs int32 [out,ksize], ksize)."""
    import math
    import numpy as np
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C double -> int: truncation (operand is > -1 here)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            v = k[x] * (1 << 22)
            kk[xx, x] = int(v - 0.5) if k[x] < 0 else int(v + 0.5)
    return bounds, kk, ksize


_resize_tabs = {}


def resize_bilinear_u8(img, newh, neww, flip=False):
    """uint8 [C,H,W] device frame -> [C,newh,neww], bit-exact with PIL ``Image.resize((neww, newh), BILINEAR)``
    (optionally followed by a horizontal flip), in one launch."""
    assert img.dtype == torch.uint8 and img.dim() == 3
    img = img.contiguous()
    C, H, W = img.shape
    key = (img.device, H, W, newh, neww)
    tabs = _resize_tabs.get(key)
    if tabs is None:
        hb, hk, ksh = pil_bilinear_coeffs(W, neww)
        vb, vk, ksv = pil_bilinear_coeffs(H, newh)
        tabs = tuple(torch.from_numpy(a).to(img.device) for a in (hb, hk, vb, vk)) + (ksh, ksv)
        _resize_tabs[key] = tabs
    hb, hk, vb, vk, ksh, ksv = tabs
    out = torch.empty(C, newh, neww, dtype=torch.uint8, device=img.device)
    call("sfod_resize_bilinear_u8", img, out, C, H, W, newh, neww, hb, hk, ksh, vb, vk, ksv, int(flip))
    return out


AUG_BRIGHTNESS, AUG_CONTRAST, AUG_SATURATION, AUG_HUE, AUG_GRAYSCALE = 0, 1, 2, 3, 4
_aug_ws = {}


def aug_color(img, ops):
    """Point operations of the strong augmentation on a uint8 [3,H,W] device frame, in order: ``ops`` =
    [(code, factor)] with the factors torchvision draws (brightness / contrast / saturation factor, hue_factor in
    [-0.5, 0.5]; grayscale ignores it).  Bit-exact with the torchvision-on-Pillow path.  -> new tensor."""
    assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[0] == 3
    img = img.contiguous()
    if not ops:
        return img.clone()
    codes = torch.tensor([int(c) for c, _ in ops], dtype=torch.int32)
    # hue: np.uint8(hue_factor * 255) -- double product, truncation toward zero, wrap-around
    factors = torch.tensor([float(int(f * 255) & 255) if c == AUG_HUE else float(f) for c, f in ops],
                           dtype=torch.float32)
    key = (img.device, torch.cuda.current_stream(img.device).cuda_stream)
    ws = _aug_ws.get(key)
    if ws is None:
        ws = _aug_ws[key] = torch.zeros(1, dtype=torch.int64, device=img.device)
    out = torch.empty_like(img)
    # codes / factors are HOST arrays (copied into the launch arguments before the call returns)
    call("sfod_aug_color", img, out, img.shape[1], img.shape[2], len(ops), codes.data_ptr(), factors.data_ptr(), ws)
    return out


def aug_gaussian_blur(img, sigma):
    """Pillow ``ImageFilter.GaussianBlur(radius=sigma)`` on a uint8 [C,H,W] device frame.  -> new tensor."""
    assert img.dtype == torch.uint8 and img.dim() == 3
    img = img.contiguous()
    out, tmp = torch.empty_like(img), torch.empty_like(img)
    call("sfod_aug_gaussian_blur", img, out, tmp, img.shape[0], img.shape[1], img.shape[2], float(sigma))
    return out


def aug_erase_(img, i, j, h, w, noise):
    """In place: RandomErasing(value="random") + ToPILImage -- rectangle <- (uint8)(noise * 255), noise [C,h,w]."""
    assert img.dtype == torch.uint8 and img.dim() == 3 and img.is_contiguous()
    assert noise.dtype == torch.float32 and tuple(noise.shape) == (img.shape[0], h, w) and noise.is_contiguous()
    call("sfod_aug_erase", img, img.shape[0], img.shape[1], img.shape[2], int(i), int(j), int(h), int(w), noise)
    return img


def hflip_u8(img):
    """uint8 [C,H,W] device image -> horizontally flipped copy."""
    assert img.dtype == torch.uint8 and img.dim() == 3
    img = img.contiguous()
    out = torch.empty_like(img)
    call("sfod_hflip_u8", img, out, img.shape[0], img.shape[1], img.shape[2])
    return out


def _wscale_slots(n, dev):
    """device words the `_ws` packers publish max|w| in (SFOD_F16X3 weights carry a per-tensor power-of-two scale,
    include/sfod_hip.h); a packed tensor keeps its word as ``.wscale`` and the forward wrappers hand it to the kernels"""
    return torch.empty(n, dtype=torch.int32, device=dev)


def wscale_of(w_packed):
    ws = getattr(w_packed, "wscale", None)
    if w_packed.dtype == SPLITH_DTYPE and ws is None:
        raise NativeLibraryError("an SFOD_F16X3 weight tensor lost its scale word (.wscale): pass the packer's own tensor, "
                                 "not a view of it")
    return ws


_f16_words = {}


def check_f16x3_range(device):
    """Raise if any producer of half pairs on ``device`` had to clamp a value beyond +-65504 (infinities included) since the last call
    (sfod_f16x3_poll).  One host synchronisation: call it where the step synchronises anyway (the metrics flush)."""
    w = _f16_words.get(device)
    if w is None:
        w = _f16_words[device] = torch.zeros(1, dtype=torch.int32, device=device)
    call("sfod_f16x3_poll", w)
    if w.item() != 0:
        w.zero_()
        raise FloatingPointError(
            "SFOD.COMPUTE_DTYPE f16x3: an activation or scaled weight beyond half's range (|v| > 65504) was clamped "
            "since the last check -- the model leaves the exponent window this mode assumes; use \"fp32\" (or \"bf16x3\").")


def pack_conv_weight(w_oihw, cin_pad, dt, rot180=False):
    cout, cin, ks, _ = w_oihw.shape
    rows = cin if rot180 else cout
    out = torch.empty(rows, ks * ks, cin_pad, dtype=torch_dtype(dt), device=w_oihw.device)
    ws = _wscale_slots(1, w_oihw.device) if dt == F16X3 else None
    call("sfod_pack_conv_weight_ws", w_oihw.contiguous(), out, ws, cout, cin, ks, cin_pad, int(rot180), dt)
    if ws is not None:
        out.wscale = ws
    return out


def grad_sink(param):
    """The tensor a hand-written backward may ACCUMULATE a parameter gradient into directly (the
    parameter's view of the flat gradient buffer, zeroed by zero_grad), or None -> return the gradient to
    autograd instead.  Saves the temporary and autograd's separate accumulate kernel per parameter."""
    g = getattr(param, "grad", None)
    if g is not None and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape == param.shape:
        return g
    return None


class ConvWeightPacker:
    """All conv weights of a module packed by ONE launch per call (the per-layer kernels are launch-bound:
    ~8 us each for <= 9 MB).  ``specs``: list of (weight parameter OIHW, inner_pad, rot180).  The descriptor
    table lives on the device and is rebuilt only when a parameter's storage moved."""

    def __init__(self, specs, dt):
        self.specs, self.dt = list(specs), dt
        self._ptrs = None
        self.views = None

    def _build(self):
        dev = self.specs[0][0].device
        tdt = torch_dtype(self.dt)
        sizes, blocks = [], []
        for w, pad, rot in self.specs:
            cout, cin, ks, _ = w.shape
            rows = cin if rot else cout
            sizes.append((rows, ks * ks, pad))
            blocks.append(query("sfod_pack_conv_weights_blocks", cout, cin, ks, pad, int(rot)))
        offs, tot = [], 0
        for r, t, p_ in sizes:
            offs.append(tot)
            tot += (r * t * p_ + 127) // 128 * 128          # keep every packed tensor 256-byte aligned
        self._buf = torch.empty(tot, dtype=tdt, device=dev)
        self.views = [self._buf[o:o + r * t * p_].view(r, t, p_) for o, (r, t, p_) in zip(offs, sizes)]
        self._amax = _wscale_slots(len(self.specs), dev) if self.dt == F16X3 else None
        if self._amax is not None:
            for i, v in enumerate(self.views):
                v.wscale = self._amax[i:i + 1]
        rows, first = [], 0
        for (w, pad, rot), v, nb in zip(self.specs, self.views, blocks):
            cout, cin, ks, _ = w.shape
            rows.append([w.data_ptr(), v.data_ptr(), cout, cin, ks, pad, int(rot), first])
            first += nb
        rows.append([0, 0, 0, 0, 0, 0, 0, first])
        self._total = first
        self._desc = torch.tensor(rows, dtype=torch.int64).to(dev)
        self._ptrs = tuple(w.data_ptr() for w, _, _ in self.specs)

    def pack(self):
        ptrs = tuple(w.data_ptr() for w, _, _ in self.specs)
        if ptrs != self._ptrs:
            for w, _, _ in self.specs:
                assert w.is_contiguous() and w.dtype == torch.float32
            self._build()
        call("sfod_pack_conv_weights_multi_ws", self._desc, len(self.specs), self._total, self.dt, self._amax)
        return self.views


def unpack_conv_wgrad(dw_packed, dw_oihw, accumulate=False):
    cout, cin, ks, _ = dw_oihw.shape
    call("sfod_unpack_conv_wgrad", dw_packed, dw_oihw, cout, cin, ks, dw_packed.shape[-1], int(accumulate))


def pack_fc_weight(w, dt, chw_c=0, transpose=False, ld=None):
    n, k = w.shape
    inner = n if transpose else k
    ld = ld or inner
    out = torch.empty(k if transpose else n, ld, dtype=torch_dtype(dt), device=w.device)
    ws = _wscale_slots(1, w.device) if dt == F16X3 else None
    call("sfod_pack_fc_weight_ld_ws", w.contiguous(), out, ws, n, k, chw_c, int(transpose), ld, dt)
    if ws is not None:
        out.wscale = ws
    return out


def unpack_fc_wgrad(dw_packed, dw, chw_c=0, accumulate=False):
    n, k = dw.shape
    call("sfod_unpack_fc_wgrad_ld", dw_packed, dw, n, k, chw_c, dw_packed.shape[-1], int(accumulate))


def conv_fwd(x, w_packed, bias, cout, ksize, act=0, out_dtype=None, ldy=None, want_stats=False):
    """x [B,H,W,Cin] NHWC (or [R,K] with ksize=1) -> y [B,H,W,ldy]; optional BN partial stats."""
    x = as_operand(x, w_packed.dtype)
    dt = dt_of(x)
    if x.dim() == 2:
        B, H, W, cin = x.shape[0], 1, 1, x.shape[1]
        oshape = lambda ld: (x.shape[0], ld)
    else:
        B, H, W, cin = x.shape
        oshape = lambda ld: (B, H, W, ld)
    out_dtype = out_dtype or out_dtype_of(x.dtype)
    ldy = ldy or cout
    alloc = torch.zeros if ldy != cout else torch.empty
    y = alloc(oshape(ldy), dtype=out_dtype, device=x.device)
    stats = None
    if want_stats:
        nb = query("sfod_conv_stats_blocks", B, H, W, cin, cout, ksize, dt)
        stats = torch.empty(nb * (2 * cout + 1), dtype=torch.float32, device=x.device)
        stats.nblk = nb
    global _pending_flops, _pending_tag
    _pending_flops = 2.0 * B * H * W * cout * ksize * ksize * cin
    if _timer is not None:
        _pending_tag = ":patch3x3" if query("sfod_conv_fwd_algo", B, H, W, cin, cout, ksize, dt) == 2 else ":gemm"
    odt = dt_of_dtype(out_dtype)
    # linear layers with few rows (one frame per GPU): the library asks for slab scratch and splits along K
    nscratch = load().sfod_conv_fwd_scratch_bytes(B, H, W, cin, cout, ksize, dt, odt, int(want_stats)) if ksize == 1 else 0
    if nscratch > 0:
        scratch = torch.empty(nscratch, dtype=torch.uint8, device=x.device)
        call("sfod_conv_fwd_scratch", x, w_packed, wscale_of(w_packed), bias, y, B, H, W, cin, cout, ksize, ldy, act, stats,
             dt, odt, scratch, nscratch, timer_name="sfod_conv_fwd")
    else:
        call("sfod_conv_fwd_ws", x, w_packed, wscale_of(w_packed), bias, y, B, H, W, cin, cout, ksize, ldy, act, stats, dt,
             odt, timer_name="sfod_conv_fwd")
    _pending_tag = ""
    return (y, stats) if want_stats else y


def conv_fwd_bnin_supported(y_pre, w_packed, cout):
    """Can the 3x3 convolution that consumes relu(bn(y_pre)) take y_pre itself (BatchNorm + ReLU + pair split in the kernel's
    LDS patch, include/sfod_hip.h: sfod_conv_fwd_bnin)?"""
    if y_pre.dim() != 4 or y_pre.dtype != torch.float32 or w_packed.dtype != SPLIT_DTYPE:
        return False
    B, H, W, cin = y_pre.shape
    return bool(query("sfod_conv_fwd_bnin_supported", B, H, W, cin, cout, BF16X3))


def conv_fwd_bnin(y_pre, mean, invstd, gamma, beta, w_packed, bias, cout, act=0, want_stats=False):
    """conv3x3(relu(bn(y_pre))) without materialising the activated tensor (forward-only passes).  -> y fp32 [B,H,W,cout]
    (+ BatchNorm partial statistics), bit-identical to bn_relu_pool_fwd(pairs) followed by conv_fwd."""
    B, H, W, cin = y_pre.shape
    y = torch.empty((B, H, W, cout), dtype=torch.float32, device=y_pre.device)
    stats = None
    if want_stats:
        nb = query("sfod_conv_stats_blocks", B, H, W, cin, cout, 3, BF16X3)
        stats = torch.empty(nb * (2 * cout + 1), dtype=torch.float32, device=y_pre.device)
        stats.nblk = nb
    global _pending_flops, _pending_tag
    _pending_flops = 2.0 * B * H * W * cout * 9 * cin
    _pending_tag = ":patch3x3"
    try:
        call("sfod_conv_fwd_bnin", y_pre, mean, invstd, gamma, beta, w_packed, bias, y, B, H, W, cin, cout, cout, act, stats,
             BF16X3, timer_name="sfod_conv_fwd")
    finally:
        _pending_tag = ""
    return (y, stats) if want_stats else y


def conv_first_supported(x, cout):
    if x.dim() != 4 or x.dtype not in (torch.bfloat16,) + PAIR_DTYPES:
        return False
    B, H, W, cin = x.shape
    return bool(query("sfod_conv_first_supported", B, H, W, cin, cout, dt_of(x), cout))


def conv_first_stats(x, w_packed, bias):
    """First VGG layer, pass 1 of the recompute-fused form: BatchNorm partial statistics only, nothing stored."""
    B, H, W, cin = x.shape
    nb = query("sfod_conv_stats_blocks", B, H, W, cin, 64, 3, dt_of(x))
    stats = torch.empty(nb * (2 * 64 + 1), dtype=torch.float32, device=x.device)
    stats.nblk = nb
    call("sfod_conv_first_fused_ws", x, w_packed, wscale_of(w_packed), bias, None, None, None, stats, B, H, W, 64, 0,
         dt_of(x))
    return stats


def conv_first_apply(x, w_packed, bias, scale, shift, relu=True):
    """Pass 2: z = act(scale * (conv + bias) + shift) [B,H,W,64] in x's operand dtype (bf16 / bf16x3 pairs); the
    pre-BatchNorm output is never stored."""
    B, H, W, cin = x.shape
    z = torch.empty(B, H, W, 64, dtype=x.dtype, device=x.device)
    global _pending_flops
    call("sfod_conv_first_fused_ws", x, w_packed, wscale_of(w_packed), bias, scale, shift, z, None, B, H, W, 64,
         1 if relu else 0, dt_of(x))
    return z


_ws_cache = {}


def _workspace(dev, nbytes):
    """Grow-only scratch buffer per (device, stream): reuse is stream-ordered, two streams never share one."""
    key = (dev, _stream())
    cur = _ws_cache.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _ws_cache[key] = cur
    return cur


def conv_wgrad(x, dy, cout, ksize, dw_packed=None, operand=None):
    """-> fp32 packed grad [Cout, taps, Cin] (accumulated into dw_packed if given).  ``operand``: MFMA operand dtype
    (default: x's); fp32 inputs of a bf16x3 model are converted to pairs here."""
    operand = operand or x.dtype
    x, dy = as_operand(x, operand), as_operand(dy, operand)
    dt = dt_of(x)
    if x.dim() == 2:
        B, H, W, cin = x.shape[0], 1, 1, x.shape[1]
    else:
        B, H, W, cin = x.shape
    lddy = dy.shape[-1]
    # bf16x3 rows come in whole 8-channel groups: compute ceil8(cout) rows (dy's padding columns are zero)
    cout_k = (cout + 7) // 8 * 8 if dt in (BF16X3, F16X3) else cout
    assert cout_k <= lddy, "dy narrower than the padded output-channel count"
    if dw_packed is None or cout_k != cout:
        assert dw_packed is None, "accumulating bf16x3 weight gradients needs Cout % 8 == 0"
        dw_packed = torch.zeros(cout_k, ksize * ksize, cin, dtype=torch.float32, device=x.device)
    nbytes = query("sfod_conv_wgrad_ws_bytes", B, H, W, cin, cout_k, ksize, lddy, dt)
    ws = _workspace(x.device, nbytes) if nbytes > 0 else None
    global _pending_flops
    _pending_flops = 2.0 * B * H * W * cout * ksize * ksize * cin
    call("sfod_conv_wgrad", x, dy, dw_packed, B, H, W, cin, cout_k, ksize, lddy, dt, ws, nbytes)
    return dw_packed if cout_k == cout else dw_packed[:cout]


def conv_wgrad_oihw_supported(x, dy, cout, ksize):
    if x.dim() != 4:
        return False
    B, H, W, cin = x.shape
    if dt_of(x) == BF16X3 and (cout % 8 or dy.shape[-1] % 8):
        return False
    return bool(query("sfod_conv_wgrad_oihw_supported", B, H, W, cin, cout, ksize, dy.shape[-1], dt_of(x)))


def conv_wgrad_oihw(x, dy, dw_oihw, accumulate=False):
    """3x3 weight gradient written straight into the state-dict layout (e.g. ``grad_sink(conv.weight)``)."""
    cout, cin, ksize, _ = dw_oihw.shape
    B, H, W, cinp = x.shape
    assert cinp == cin and dw_oihw.is_contiguous() and dw_oihw.dtype == torch.float32
    dy = as_operand(dy, x.dtype)
    lddy = dy.shape[-1]
    dt = dt_of(x)
    nbytes = query("sfod_conv_wgrad_ws_bytes", B, H, W, cin, cout, ksize, lddy, dt)
    ws = _workspace(x.device, nbytes)
    global _pending_flops
    _pending_flops = 2.0 * B * H * W * cout * ksize * ksize * cin
    call("sfod_conv_wgrad_oihw", x, dy, dw_oihw, B, H, W, cin, cout, ksize, lddy, dt, int(accumulate), ws, nbytes)
    return dw_oihw


def conv_weight_grad(x, dy, weight, operand=None):
    """dL/dweight of a conv whose state-dict weight is ``weight`` (OIHW).  Accumulated straight into
    ``grad_sink(weight)`` when the parameter has one (returns None: nothing for autograd to add), else
    returned as a new tensor."""
    operand = operand or x.dtype
    x, dy = as_operand(x, operand), as_operand(dy, operand)
    cout, cin, k, _ = weight.shape
    sink = grad_sink(weight)
    if sink is not None and k == 3 and x.shape[-1] == cin and conv_wgrad_oihw_supported(x, dy, cout, 3):
        conv_wgrad_oihw(x, dy, sink, accumulate=True)     # the slab reduction writes OIHW directly
        return None
    if sink is not None and k == 1 and x.shape[-1] == cin and (dt_of(x) != BF16X3 or cout % 8 == 0):
        conv_wgrad(x, dy, cout, 1, dw_packed=sink.view(cout, 1, cin))    # packed [Cout,1,Cin] IS the OIHW layout: the
        return None                                                     # kernel's atomics accumulate in place
    dwp = conv_wgrad(x, dy, cout, k)
    if sink is not None:
        unpack_conv_wgrad(dwp, sink, accumulate=True)
        return None
    dw = torch.empty_like(weight)
    unpack_conv_wgrad(dwp, dw)
    return dw


def bias_grad(dy, n, db=None, accumulate=False):
    m = dy.numel() // dy.shape[-1]
    if db is None:
        db = torch.empty(n, dtype=torch.float32, device=dy.device)
    call("sfod_bias_grad", dy, db, m, n, dy.shape[-1], int(accumulate), dt_of(dy))
    return db


def bn_finalize(stats, M, C, running_mean, running_var, momentum=0.1, eps=1e-5, update_running=True,
                num_batches_tracked=None):
    """``update_running``: False / 0 = no update, True / k = k momentum updates with this batch's statistics;
    ``num_batches_tracked`` (int64 scalar buffer) is incremented by k in the same launch when given."""
    mean = torch.empty(C, dtype=torch.float32, device=stats.device)
    invstd = torch.empty_like(mean)
    ws = torch.empty(query("sfod_bn_finalize_ws_floats", C), dtype=torch.float32, device=stats.device)
    call("sfod_bn_finalize", stats, stats.nblk, M, C, mean, invstd, running_mean, running_var,
         float(momentum), float(eps), int(update_running), num_batches_tracked, ws)
    return mean, invstd


def bn_relu_pool_fwd(y, mean, invstd, gamma, beta, pool, relu=True, out_dtype=None, with_grad_operand=False):
    """``out_dtype``: y.dtype, or a pair dtype from an fp32 y (the next convolution's operand, written directly).
    ``with_grad_operand`` (out_dtype SPLITH_DTYPE): -> (z, the same values as bf16 pairs: the operand of that
    convolution's weight gradient), both written by the one pass."""
    B, H, W, C = y.shape
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    z = torch.empty(B, Ho, Wo, C, dtype=out_dtype or y.dtype, device=y.device)
    if with_grad_operand and z.dtype == SPLITH_DTYPE:
        z2 = torch.empty(B, Ho, Wo, C, dtype=SPLIT_DTYPE, device=y.device)
        call("sfod_bn_relu_pool_fwd2", y, mean, invstd, gamma, beta, z, z2, B, H, W, C,
             int(pool) | (0 if relu else 2), dt_of(y), dt_of(z))
        return z, z2
    call("sfod_bn_relu_pool_fwd", y, mean, invstd, gamma, beta, z, B, H, W, C, int(pool) | (0 if relu else 2),
         dt_of(y), dt_of(z))
    return (z, z) if with_grad_operand else z


def bn_add_relu_fwd(y, mean, invstd, gamma, beta, residual, with_operand=None, with_grad_operand=False):
    """relu(bn(y) + residual) in one pass (bottleneck tail).  ``with_operand`` (a pair dtype; fp32 data of a bf16x3 /
    f16x3 model): also return the same values as (hi, lo) operand pairs of that type -> (z, z_pairs);
    ``with_grad_operand`` (half pairs only): -> (z, z_pairs, the same as bf16 pairs), all from the one pass."""
    assert residual.shape == y.shape and residual.dtype == y.dtype
    z = torch.empty_like(y)
    C = y.shape[-1]
    with_operand = SPLIT_DTYPE if with_operand is True else with_operand
    zp = torch.empty(y.shape, dtype=with_operand, device=y.device) if with_operand else None
    zg = torch.empty(y.shape, dtype=SPLIT_DTYPE, device=y.device) if (with_grad_operand and with_operand == SPLITH_DTYPE) else None
    call("sfod_bn_add_relu_fwd", y, mean, invstd, gamma, beta, residual.contiguous(), z, zp, zg, y.numel() // C, C, dt_of(y),
         dt_of(zp) if zp is not None else BF16X3)
    if with_grad_operand:
        return z, zp, (zg if zg is not None else zp)
    return (z, zp) if with_operand else z


def bn_relu_pool_bwd(dz, y, mean, invstd, gamma, beta, pool, dgamma=None, dbeta=None, dy=None, relu=True,
                     dgamma_acc=None, dbeta_acc=None, out_dtype=None, reduced=None):
    """``dgamma_acc`` / ``dbeta_acc``: gradient accumulators (see ``grad_sink``) updated in the same launch.
    ``out_dtype``: dtype of dy (y.dtype, or SPLIT_DTYPE from fp32 inputs: dy only feeds the wgrad / dgrad MFMAs).
    ``reduced``: the partial-sum workspace ``conv_dgrad_bnred`` returned with dz -- the reduction pass is skipped."""
    B, H, W, C = y.shape
    if dy is None:
        dy = torch.empty(y.shape, dtype=out_dtype or y.dtype, device=y.device)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=y.device)
    if dbeta is None:
        dbeta = torch.empty(C, dtype=torch.float32, device=y.device)
    if reduced is not None:
        ws, nred = reduced, reduced.shape[0] - BN_BWD_SCRATCH_ROWS
        assert reduced.shape[1] == 2 * C and nred > 0 and not pool and relu
    else:
        ws, nred = torch.empty(query("sfod_bn_bwd_ws_floats", B * H * W, C), dtype=torch.float32, device=y.device), 0
    call("sfod_bn_relu_pool_bwd", dz, y, mean, invstd, gamma, beta, dy, dgamma, dbeta, dgamma_acc, dbeta_acc, ws,
         B, H, W, C,
         int(pool) | (0 if relu else 2), dt_of(y), dt_of(dy), nred)
    return dy, dgamma, dbeta


BN_BWD_SCRATCH_ROWS = 32     # SFOD_BN_BWD_SCRATCH_ROWS of include/sfod_hip.h


def conv_dgrad_bnred(dy, w_rot, cout, y_below, mean, invstd, gamma, beta):
    """3x3 data gradient dz = conv(dy, rotated weights) [B,H,W,cout] fp32, with the BatchNorm-backward reduction of the
    layer below (whose saved pre-BatchNorm output is ``y_below``, same shape as dz) folded into the epilogue.
    -> (dz, partial-sum workspace for ``bn_relu_pool_bwd(reduced=...)``), or None when the shape / dtype is not served
    (caller: plain ``conv_fwd`` + unfused BatchNorm backward)."""
    B, H, W, cin = dy.shape
    if y_below.dtype != torch.float32 or tuple(y_below.shape) != (B, H, W, cout) or not y_below.is_contiguous():
        return None
    nb = query("sfod_conv_dgrad_bnred_blocks", B, H, W, cin, cout, dt_of(dy))
    if nb <= 0:
        return None
    dz = torch.empty(B, H, W, cout, dtype=torch.float32, device=dy.device)
    ws = torch.empty(nb + BN_BWD_SCRATCH_ROWS, 2 * cout, dtype=torch.float32, device=dy.device)   # partial rows + scratch
    global _pending_flops, _pending_tag
    _pending_flops = 2.0 * B * H * W * cout * 9 * cin
    _pending_tag = ":patch3x3"       # the halo-patch kernel with one more epilogue: same roofline line as sfod_conv_fwd's
    try:
        call("sfod_conv_dgrad_bnred", dy, w_rot, dz, B, H, W, cin, cout, dt_of(dy), y_below, mean, invstd, gamma, beta,
             ws, timer_name="sfod_conv_fwd")
    finally:
        _pending_tag = ""
    return dz, ws


def act_bwd_(dy, y, act):
    call("sfod_act_bwd", dy, y, dy.numel(), act, dt_of(dy))
    return dy


def add_(a, b):
    call("sfod_add_inplace", a, b, a.numel(), dt_of(a))
    return a


def add_act_bwd_(a, b, y):
    """a = (a + b) * (y > 0) in place (add_ followed by act_bwd_(.., y, 1))."""
    call("sfod_add_act_bwd", a, b, y, a.numel(), dt_of(a))
    return a


def mul_mask_(a, mask_u8, scale):
    call("sfod_mul_mask", a, mask_u8, a.numel(), float(scale), dt_of(a))
    return a


def add_act(a, b, act=1, with_operand=None, with_grad_operand=False):
    """act(a + b); ``with_operand`` (a pair dtype; fp32 data of a bf16x3 / f16x3 model): also the (hi, lo) operand pairs
    of that type -> (out, out_pairs); ``with_grad_operand``: -> (out, out_pairs, the same as bf16 pairs)."""
    out = torch.empty_like(a)
    with_operand = SPLIT_DTYPE if with_operand is True else with_operand
    op = torch.empty(a.shape, dtype=with_operand, device=a.device) if with_operand else None
    og = torch.empty(a.shape, dtype=SPLIT_DTYPE, device=a.device) if (with_grad_operand and with_operand == SPLITH_DTYPE) else None
    call("sfod_add_act", a, b, out, op, og, a.numel(), int(act), dt_of(a), dt_of(op) if op is not None else BF16X3)
    if with_grad_operand:
        return out, op, (og if og is not None else op)
    return (out, op) if with_operand else out


def subsample2(x):
    B, H, W, C = x.shape
    y = torch.empty(B, (H + 1) // 2, (W + 1) // 2, C, dtype=x.dtype, device=x.device)
    call("sfod_subsample2", x, y, B, H, W, C, 0, dt_of(x))
    return y


def subsample2_bwd(dy, full_shape):
    B, H, W, C = full_shape
    dx = torch.empty(B, H, W, C, dtype=dy.dtype, device=dy.device)
    call("sfod_subsample2", dy, dx, B, H, W, C, 1, dt_of(dy))
    return dx


def im2col_stem(x, kpad=192, out_dtype=None):
    """``out_dtype``: x.dtype, or SPLIT_DTYPE from fp32 (the stem GEMM's operand, no separate conversion pass)."""
    B, H, W, Cp = x.shape
    out = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, kpad, dtype=out_dtype or x.dtype, device=x.device)
    call("sfod_im2col_stem", x, out, B, H, W, Cp, kpad, dt_of(x), dt_of(out))
    return out


def stem7x7_supported(x, dt):
    B, H, W, Cp = x.shape
    return x.dtype == torch.float32 and bool(query("sfod_stem7x7_supported", B, H, W, Cp, dt))


def stem7x7(x, w_packed, bias, act=1):
    """fp32 NHWC input [B,H,W,Cp] -> relu(conv7x7 s2 p3 (+ bias)) fp32 [B,Ho,Wo,64]; w_packed = pack_fc_weight([64,160])."""
    B, H, W, Cp = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, 64, dtype=torch.float32, device=x.device)
    global _pending_flops, _pending_tag
    _pending_flops = 2.0 * B * Ho * Wo * 64 * 147
    if _timer is not None:
        _pending_tag = ":stem"
    call("sfod_stem7x7", x.contiguous(), w_packed, wscale_of(w_packed), bias, y, B, H, W, Cp, int(act), dt_of_dtype(w_packed.dtype),
         timer_name="sfod_conv_fwd")
    _pending_tag = ""
    return y


def maxpool3s2(x):
    B, H, W, C = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, dtype=x.dtype, device=x.device)
    call("sfod_maxpool3s2", x, y, B, H, W, C, dt_of(x))
    return y


def cast(src, dtype):
    if src.dtype == dtype:
        return src
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    call("sfod_cast", src.contiguous(), dst, src.numel(), dt_of(src), dt_of_dtype(dtype))
    return dst


# ---- detection ops -----------------------------------------------------------------------------
def rpn_decode(rpn_out, cell, B, Hf, Wf, stride, sizes, flags):
    A = cell.shape[0]
    NA = Hf * Wf * A
    props = torch.empty(B, NA, 4, dtype=torch.float32, device=rpn_out.device)
    scores = torch.empty(B, NA, dtype=torch.float32, device=rpn_out.device)
    call("sfod_rpn_decode", rpn_out, rpn_out.shape[-1], cell, A, B, Hf, Wf, stride, sizes, props, scores, flags)
    return props, scores


_sort_ws = {}


def segmented_sort_desc(keys):
    """keys fp32 [B,n] -> (sorted_keys [B,n], idx int32 [B,n]); stable."""
    B, n = keys.shape
    dev = keys.device
    nbytes = query("sfod_sort_ws_bytes", B, n)
    k = (dev, _stream(), nbytes)      # per stream: the teacher's side stream and the main stream may both sort
    if k not in _sort_ws:
        _sort_ws[k] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ws = _sort_ws[k]
    out_keys = torch.empty_like(keys)
    out_idx = torch.empty(B, n, dtype=torch.int32, device=dev)
    call("sfod_segmented_sort_desc", keys, B, n, out_keys, out_idx, ws, nbytes)
    return out_keys, out_idx


def rpn_gather_topk(props, sorted_scores, sorted_idx, k):
    B, NA, _ = props.shape
    dev = props.device
    cb = torch.empty(B, k, 4, dtype=torch.float32, device=dev)
    cs = torch.empty(B, k, dtype=torch.float32, device=dev)
    cv = torch.empty(B, k, dtype=torch.uint8, device=dev)
    call("sfod_rpn_gather_topk", props, sorted_scores, sorted_idx, B, NA, k, cb, cs, cv)
    return cb, cs, cv


_mask_ws = {}


def nms(boxes, thr, max_keep, valid=None, n_per_image=None, alt_boxes=None, classes=None, mode=None):
    """boxes fp32 [B,n,4] sorted by descending score -> (keep_idx int32 [B,max_keep], count [B])."""
    B, n, _ = boxes.shape
    dev = boxes.device
    nbytes = max(8, query("sfod_nms_mask_bytes", B, n))
    k = (dev, _stream(), nbytes)      # per stream: two streams may run NMS at once
    if k not in _mask_ws:
        _mask_ws[k] = torch.empty(nbytes // 8, dtype=torch.int64, device=dev)
    mask = _mask_ws[k]
    keep_idx = torch.zeros(B, max_keep, dtype=torch.int32, device=dev)
    keep_count = torch.zeros(B, dtype=torch.int32, device=dev)
    call("sfod_nms", boxes, alt_boxes, classes, mode, valid, n_per_image, B, n, float(thr), max_keep, mask,
         keep_idx, keep_count)
    return keep_idx, keep_count


def gather_kept(cand_boxes, cand_scores, keep_idx, keep_count):
    B, n, _ = cand_boxes.shape
    max_keep = keep_idx.shape[1]
    ob = torch.empty(B, max_keep, 4, dtype=torch.float32, device=cand_boxes.device)
    os_ = torch.empty(B, max_keep, dtype=torch.float32, device=cand_boxes.device)
    call("sfod_gather_kept", cand_boxes, cand_scores, keep_idx, keep_count, B, n, max_keep, ob, os_)
    return ob, os_


def anchor_match(cell, B, Hf, Wf, stride, gt_boxes, gt_count, lo, hi):
    A = cell.shape[0]
    NA = Hf * Wf * A
    dev = gt_boxes.device
    gcap = gt_boxes.shape[1]
    matched = torch.empty(B, NA, dtype=torch.int32, device=dev)
    labels = torch.empty(B, NA, dtype=torch.int8, device=dev)
    scratch = torch.empty(B * gcap + B * NA, dtype=torch.float32, device=dev)
    call("sfod_anchor_match", cell, A, B, Hf, Wf, stride, gt_boxes, gt_count, gcap, float(lo), float(hi),
         matched, labels, scratch)
    return matched, labels


def roi_match(boxes, box_count, gt_boxes, gt_classes, gt_count, thr, num_classes):
    B, P, _ = boxes.shape
    matched = torch.empty(B, P, dtype=torch.int32, device=boxes.device)
    cls = torch.empty(B, P, dtype=torch.int32, device=boxes.device)
    call("sfod_roi_match", boxes, box_count, B, P, gt_boxes, gt_classes, gt_count, gt_boxes.shape[1],
         float(thr), num_classes, matched, cls)
    return matched, cls


def subsample_rpn_(labels, keys, num, pos_frac):
    B, n = labels.shape
    counts = torch.empty(B, 2, dtype=torch.int32, device=labels.device)
    call("sfod_subsample", labels, keys, B, n, num, float(pos_frac), 0, 0, None, counts)
    return labels, counts


def subsample_roi(cls, keys, num, pos_frac, bg_label):
    B, n = cls.shape
    idx = torch.zeros(B, num, dtype=torch.int32, device=cls.device)
    cnt = torch.empty(B, dtype=torch.int32, device=cls.device)
    call("sfod_subsample", cls, keys, B, n, num, float(pos_frac), bg_label, 1, idx, cnt)
    return idx, cnt


def rpn_loss(rpn_out, cell, B, Hf, Wf, stride, labels, matched, gt_boxes, gt_count, batch_per_image,
             grad_scale=None):
    A = cell.shape[0]
    NA = Hf * Wf * A
    dev = rpn_out.device
    loss = torch.empty(2, dtype=torch.float32, device=dev)
    nblk = (NA + 255) // 256 * B
    ws = torch.empty(nblk * 2, dtype=torch.float32, device=dev)
    d_out = torch.empty_like(rpn_out) if grad_scale is not None else None
    call("sfod_rpn_loss", rpn_out, rpn_out.shape[-1], cell, A, B, Hf, Wf, stride, labels, matched, gt_boxes,
         gt_count, gt_boxes.shape[1], batch_per_image, loss, grad_scale, d_out, ws)
    return loss, d_out


def append_gt(props, prop_count, gt_boxes, gt_count):
    B, P, _ = props.shape
    gcap = gt_boxes.shape[1]
    out = torch.empty(B, P + gcap, 4, dtype=torch.float32, device=props.device)
    cnt = torch.empty(B, dtype=torch.int32, device=props.device)
    call("sfod_append_gt", props, prop_count, B, P, gt_boxes, gt_count, gcap, out, cnt)
    return out, cnt


def roi_build_samples(boxes, cls, matched, samp_idx, samp_count, gt_boxes, gt_count):
    B, P, _ = boxes.shape
    S = samp_idx.shape[1]
    dev = boxes.device
    rois = torch.empty(B * S, 5, dtype=torch.float32, device=dev)
    gt_cls = torch.empty(B * S, dtype=torch.int32, device=dev)
    gt_box = torch.empty(B * S, 4, dtype=torch.float32, device=dev)
    n_valid = torch.empty(1, dtype=torch.int32, device=dev)
    call("sfod_roi_build_samples", boxes, cls, matched, samp_idx, samp_count, B, P, S, gt_boxes, gt_count,
         gt_boxes.shape[1], rois, gt_cls, gt_box, n_valid)
    return rois, gt_cls, gt_box, n_valid


def make_rois(props, prop_count):
    B, P, _ = props.shape
    rois = torch.empty(B * P, 5, dtype=torch.float32, device=props.device)
    call("sfod_make_rois", props, prop_count, B, P, rois)
    return rois


def roi_align_fwd(feat, rois, pooled, scale):
    B, H, W, C = feat.shape
    R = rois.shape[0]
    out = torch.empty(R, pooled * pooled, C, dtype=feat.dtype, device=feat.device)
    call("sfod_roi_align_fwd", feat, B, H, W, C, rois, R, pooled, float(scale), out, dt_of(feat))
    return out


def roi_align_bwd(dout, rois, feat_shape, pooled, scale, dfeat=None):
    B, H, W, C = feat_shape
    if dfeat is None:
        dfeat = torch.zeros(B, H, W, C, dtype=torch.float32, device=dout.device)
    call("sfod_roi_align_bwd", dout, B, H, W, C, rois, rois.shape[0], pooled, float(scale), dfeat, dt_of(dout))
    return dfeat


def frcnn_loss(pred, K, rois, gt_cls, gt_box, n_valid, grad_scale=None):
    R, ld = pred.shape
    dev = pred.device
    loss = torch.empty(2, dtype=torch.float32, device=dev)
    ws = torch.empty(((R + 255) // 256) * 2, dtype=torch.float32, device=dev)
    d_pred = torch.empty_like(pred) if grad_scale is not None else None
    call("sfod_frcnn_loss", pred, ld, R, K, rois, gt_cls, gt_box, n_valid, loss, grad_scale, d_pred, ws)
    return loss, d_pred


def frcnn_inference(pred, K, props, prop_count, sizes, score_thresh, nms_thresh, max_det, pseudo_thr,
                    numel_limit=20000):
    """Teacher post-processing -> dict of fixed-capacity per-image arrays + counts."""
    B, P, _ = props.shape
    dev = pred.device
    n = P * K
    cb = torch.empty(B, n, 4, dtype=torch.float32, device=dev)
    cs = torch.empty(B, n, dtype=torch.float32, device=dev)
    cc = torch.empty(B, dtype=torch.int32, device=dev)
    call("sfod_frcnn_candidates", pred, pred.shape[-1], B, P, K, props, prop_count, sizes, float(score_thresh),
         cb, cs, cc)
    ss, si = segmented_sort_desc(cs)
    sb = torch.empty(B, n, 4, dtype=torch.float32, device=dev)
    sa = torch.empty(B, n, 4, dtype=torch.float32, device=dev)
    sc = torch.empty(B, n, dtype=torch.int32, device=dev)
    mode = torch.empty(B, dtype=torch.int32, device=dev)
    mx = torch.empty(B, dtype=torch.float32, device=dev)
    call("sfod_frcnn_prepare_nms", cb, ss, si, cc, B, n, K, numel_limit, sb, sa, sc, mode, mx)
    keep_idx, keep_count = nms(sb, nms_thresh, max_det, n_per_image=cc, alt_boxes=sa, classes=sc, mode=mode)
    out = {
        "det_boxes": torch.empty(B, max_det, 4, dtype=torch.float32, device=dev),
        "det_scores": torch.empty(B, max_det, dtype=torch.float32, device=dev),
        "det_classes": torch.empty(B, max_det, dtype=torch.int32, device=dev),
        "det_count": torch.empty(B, dtype=torch.int32, device=dev),
        "gt_boxes": torch.empty(B, max_det, 4, dtype=torch.float32, device=dev),
        "gt_classes": torch.empty(B, max_det, dtype=torch.int32, device=dev),
        "gt_count": torch.empty(B, dtype=torch.int32, device=dev),
        "cand_count": cc, "mode": mode,
    }
    call("sfod_frcnn_finalize", sb, ss, sc, keep_idx, keep_count, B, n, max_det, float(pseudo_thr),
         out["det_boxes"], out["det_scores"], out["det_classes"], out["det_count"], out["gt_boxes"],
         out["gt_classes"], out["gt_count"])
    return out


def bpc_loss(pred, K, rois, roi_cls, sizes_dev, gt_boxes, gt_classes, gt_count, iou_thresh=0.5):
    """BPC calibration scalar of one training pass (fused convert_bbox_scores + bpc_loss) -> 0-dim tensor."""
    B, G = gt_classes.shape
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    ws = torch.empty(max(B, 1) * 4, dtype=torch.float64, device=pred.device)
    call("sfod_bpc_loss", pred, pred.shape[-1], pred.shape[0], K, rois, roi_cls, B, sizes_dev, gt_boxes, gt_classes,
         gt_count, G, float(iou_thresh), loss, ws)
    return loss[0]


def predict_probs(scores):
    """Row softmax of the class scores [R, K+1] (any row stride) in the teacher post-processing kernel's operation order."""
    assert scores.is_cuda, "native ops need device tensors"
    assert scores.dim() == 2 and scores.dtype == torch.float32 and scores.stride(1) == 1
    R, n = scores.shape
    out = torch.empty(R, n, dtype=torch.float32, device=scores.device)
    # (rows may be a strided view of the prediction matrix: the pointer goes in as is, the row stride as ld)
    call("sfod_predict_probs", scores.data_ptr(), scores.stride(0) if R > 1 else n, R, n - 1, out)
    return out


def adaptive_pseudo_labels_(d, thr, reserve, row, class_acc, select):
    """In place on the detection dict of ``frcnn_inference``: updates ``reserve[row]`` / ``class_acc`` and, with
    ``select``, replaces ``gt_boxes`` / ``gt_classes`` / ``gt_count`` (adds ``gt_scores``) by the class-wise
    adaptive selection."""
    B, max_det = d["det_scores"].shape
    R, K = reserve.shape
    if "gt_scores" not in d:
        d["gt_scores"] = torch.zeros(B, max_det, dtype=torch.float32, device=reserve.device)
    call("sfod_adaptive_pseudo_labels", d["det_boxes"], d["det_scores"], d["det_classes"], d["det_count"], B, max_det,
         K, float(thr), reserve, R, int(row), class_acc, int(bool(select)), d["gt_boxes"], d["gt_classes"],
         d["gt_scores"], d["gt_count"])
    if select:
        d["gt_adaptive"] = True
    return d


def sgd_ema_(param, grad, mom, teacher, lr, momentum, weight_decay, grad_scale, ema_keep, first_step):
    call("sfod_sgd_ema", param, grad, mom, teacher, param.numel(), lr, float(momentum), float(weight_decay),
         float(grad_scale), float(ema_keep), float(1.0 - float(ema_keep)), int(first_step))


def ema_(teacher, student, keep):
    call("sfod_ema", teacher, student, teacher.numel(), float(keep), float(1.0 - float(keep)))


def ema_i64_(teacher, student, keep):
    """int64 buffers: float32 arithmetic, truncated on the way back (the reference's EMA + load_state_dict copy)."""
    assert teacher.dtype == torch.int64 and student.dtype == torch.int64 and teacher.numel() == student.numel()
    call("sfod_ema_i64", teacher, student, teacher.numel(), float(keep), float(1.0 - keep))


def teacher_metrics(det_scores, det_count, rpn_logits, rpn_count, gt_count, thr):
    """-> float32 [3]: mean detection confidence, RPN proposals above ``thr`` per image, mean pseudo-label count."""
    B, D = det_scores.shape
    out = torch.empty(3, dtype=torch.float32, device=det_scores.device)
    call("sfod_teacher_metrics", det_scores, det_count, D, rpn_logits, rpn_count, rpn_logits.shape[1], gt_count, B,
         float(thr), out)
    return out
