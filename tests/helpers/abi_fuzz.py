"""Hostile-argument fuzz of the C ABI's HOST side (tests/test_abi.py runs this in a child process with the ASan
runtime preloaded, against lib/libsfod_hip_hostsan.so = the library's host code built with
-fsanitize=address,undefined and no device code).

Part 1 (every entry point, random hostile vectors): negative / zero / huge / misaligned sizes, NULL and valid pointers,
NaN / infinite floats.  Whatever the arguments, a call must RETURN: SFOD_EBADARG (-1000) with ``sfod_last_error()``
set, a HIP error code (a syntactically valid call reaches a launch, which cannot run here), 0, or -- for the
``*_supported`` / ``*_bytes`` / ``*_blocks`` / ``*_floats`` queries -- a non-negative answer.  The sanitizers report
integer overflow, out-of-bounds host reads and the like on the way (printed to stderr; the parent test fails on any).

Part 2 (the contract of the entry points the hot path calls): from a valid call of each, ONE argument is broken at a time
(size < 0, channel count not a multiple of the 8-channel group, ldy < Cout, a NULL operand, unknown dtype ...):
every such call returns -1000 with a message and therefore never reached a launch.

No torch, no GPU.  Prints one JSON line.
"""
import ctypes
import json
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HEADER = os.path.join(ROOT, "include", "sfod_hip.h")
EBADARG = -1000
BF16, F32, BF16X3, F16X3 = 0, 1, 2, 3


def parse(path=HEADER):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(sfod_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                pname = re.sub(r"[\*\s]", " ", a).split()[-1]
                if "*" in a:
                    params.append((pname, "ptr"))
                elif a.startswith("int64_t"):
                    params.append((pname, "i64"))
                elif a.startswith("float"):
                    params.append((pname, "f32"))
                else:
                    params.append((pname, "i32"))
        out[name] = ("str" if "char" in ret else ("i64" if ret == "int64_t" else "i32"), params)
    return out


CT = {"ptr": ctypes.c_void_p, "i64": ctypes.c_int64, "i32": ctypes.c_int, "f32": ctypes.c_float}
I32_POOL = [-2 ** 31, -2 ** 31 + 1, -65536, -9, -8, -1, 0, 1, 2, 3, 7, 8, 9, 15, 16, 31, 32, 33, 63, 64, 100, 127, 128, 255, 256,
            257, 511, 512, 1000, 1024, 4095, 4096, 65535, 65536, 10 ** 6, 2 ** 24, 2 ** 30, 2 ** 31 - 2, 2 ** 31 - 1]
I64_POOL = I32_POOL + [-2 ** 63, -2 ** 40, 2 ** 32, 2 ** 40, 2 ** 62, 2 ** 63 - 1]
F32_POOL = [float("nan"), float("inf"), float("-inf"), -1.0, -0.0, 0.0, 1e-30, 0.5, 1.0, 16.0, 1e30]


def main():
    so = sys.argv[1]
    n_random = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    protos = parse()
    lib = ctypes.CDLL(so)
    for name, (ret, params) in protos.items():
        fn = getattr(lib, name)
        fn.restype = {"str": ctypes.c_char_p, "i64": ctypes.c_int64, "i32": ctypes.c_int}[ret]
        fn.argtypes = [CT[t] for _, t in params]
    rng = random.Random(1234)
    buf = ctypes.create_string_buffer(1 << 16)          # a valid, 16-byte aligned host range for every pointer argument
    base = (ctypes.addressof(buf) + 63) // 64 * 64
    calls = bad = 0
    outcomes = {}
    setters = [n for n in protos if n.startswith("sfod_set_")]
    queries = [n for n in protos if n.endswith(("_supported", "_bytes", "_blocks", "_floats", "_algo")) and not n.startswith("sfod_set_")]

    def rand_args(params):
        a = []
        for pname, t in params:
            if t == "ptr":
                a.append(None if (pname != "stream" and rng.random() < 0.25) or pname == "stream" else base + 64 * rng.randrange(0, 8))
            elif t == "i64":
                a.append(rng.choice(I64_POOL) if rng.random() < 0.8 else rng.randrange(-2 ** 40, 2 ** 40))
            elif t == "i32":
                if pname in ("dt", "out_dt", "src_dt", "dst_dt", "pairs_dt") and rng.random() < 0.7:
                    a.append(rng.randrange(0, 4))
                else:
                    a.append(rng.choice(I32_POOL) if rng.random() < 0.8 else rng.randrange(-5000, 5000))
            else:
                a.append(rng.choice(F32_POOL))
        return a

    crashed = []
    for name, (ret, params) in sorted(protos.items()):
        if ret == "str" or name in setters or not params:
            continue
        fn = getattr(lib, name)
        # one child per entry point: a crash (or a fatal sanitizer report) in one does not hide the others
        vectors = [rand_args(params) for _ in range(n_random)]
        sys.stdout.flush()
        sys.stderr.flush()
        pid = os.fork()
        if pid == 0:
            code = 0
            for args in vectors:
                rc = fn(*args)
                if name in queries:
                    if rc < 0 and rc != EBADARG:
                        sys.stderr.write("FUZZ negative_query_answer %s %r -> %d\n" % (name, args, rc))
                        code = 3
                elif rc == EBADARG and not lib.sfod_last_error():
                    sys.stderr.write("FUZZ ebadarg_without_message %s %r\n" % (name, args))
                    code = 3
            sys.stderr.flush()
            os._exit(code)
        _, status = os.waitpid(pid, 0)
        calls += len(vectors)
        if status != 0:
            bad += 1
            crashed.append((name, status))
            outcomes.setdefault("crashed_or_violated", []).append((name, status))
    # the process-wide switches take any int
    for name in setters:
        for v in (-2 ** 31, -1, 0, 1, 2, 5, 6, 7, 99, 2 ** 31 - 1):
            getattr(lib, name)(v)
    for name, v in (("sfod_set_conv_algo", 0), ("sfod_set_conv3x3_variant", 0), ("sfod_set_deterministic", 0),
                    ("sfod_set_conv3x3_m16", 1), ("sfod_set_wgrad3x3_pipe", 1)):
        if name in protos:
            getattr(lib, name)(v)

    # ---- part 2: the contract ------------------------------------------------------------------------------------------
    P = base
    contract = []

    def expect_badarg(name, valid, mutations):
        fn = getattr(lib, name)
        names = [p for p, _ in protos[name][1]]
        for key, val in mutations:
            a = list(valid)
            a[names.index(key)] = val
            rc = fn(*a)
            msg = lib.sfod_last_error() or b""
            ok = rc == EBADARG and len(msg) > 0
            contract.append({"fn": name, "arg": key, "value": val if val is None or isinstance(val, (int, float)) else str(val),
                             "rc": rc, "ok": ok, "msg": msg.decode()[:80]})

    # (x, w, bias, y, B, H, W, Cin, Cout, ksize, ldy, act, stats, dt, out_dt, stream)
    conv = [P, P, P, P, 2, 32, 64, 64, 128, 3, 128, 1, None, BF16X3, F32, None]
    expect_badarg("sfod_conv_fwd", conv, [("B", -1), ("H", -4), ("W", -1), ("Cin", -8), ("Cout", -128), ("Cin", 60), ("Cout", 100),
                                          ("ksize", 2), ("ksize", -3), ("ldy", 64), ("ldy", -128), ("ldy", 129), ("x", None),
                                          ("w", None), ("y", None), ("dt", 9), ("dt", -1), ("out_dt", 7), ("act", 17)])
    # (x_pre, in_mean, in_invstd, in_gamma, in_beta, w, bias, y, B, H, W, Cin, Cout, ldy, act, stats, dt, stream)
    bnin = [P, P, P, P, P, P, P, P, 8, 150, 300, 256, 256, 256, 0, None, BF16X3, None]
    expect_badarg("sfod_conv_fwd_bnin", bnin, [("B", -8), ("B", 0), ("Cin", 250), ("ldy", 128), ("x_pre", None), ("in_mean", None),
                                               ("w", None), ("y", None), ("dt", F32), ("dt", 11)])
    # (x, dy, dw, B, H, W, Cin, Cout, ksize, lddy, dt, ws, ws_bytes, stream)
    wg = [P, P, P, 2, 32, 64, 64, 128, 3, 128, BF16X3, P, 1 << 30, None]
    expect_badarg("sfod_conv_wgrad", wg, [("B", -2), ("H", -1), ("Cin", 63), ("Cout", -1), ("ksize", 5), ("lddy", 64), ("x", None),
                                          ("dy", None), ("dw", None), ("dt", 12), ("ws_bytes", -1)])
    # (feat, B, H, W, C, rois, R, pooled, scale, out, dt, stream)
    ra = [P, 2, 19, 38, 512, P, 100, 7, 1.0 / 32, P, F32, None]
    expect_badarg("sfod_roi_align_fwd", ra, [("B", -1), ("H", 0), ("W", -3), ("C", -512), ("C", 500), ("R", -1), ("pooled", 0),
                                             ("pooled", -7), ("feat", None), ("rois", None), ("out", None), ("dt", 8)])
    # (boxes, alt_boxes, classes, mode, valid, n_per_image, B, n, thr, max_keep, mask, keep_idx, keep_count, stream)
    nms = [P, None, None, None, None, None, 2, 1000, 0.7, 100, P, P, P, None]
    expect_badarg("sfod_nms", nms, [("B", -1), ("n", -5), ("max_keep", -1), ("boxes", None), ("mask", None), ("keep_idx", None),
                                    ("keep_count", None)])
    # (keys, B, n, out_keys, out_idx, ws, ws_bytes, stream)
    srt = [P, 2, 1000, P, P, P, 1 << 24, None]
    expect_badarg("sfod_segmented_sort_desc", srt, [("B", -1), ("n", -1), ("keys", None), ("out_keys", None), ("out_idx", None),
                                                    ("ws", None), ("ws_bytes", 16), ("ws_bytes", -1)])
    # (param, grad, mom, teacher, n, lr, momentum, weight_decay, grad_scale, ema_keep, ema_one_minus_keep, first_step, stream)
    sgd = [P, P, P, None, 1000, P, 0.9, 1e-4, 1.0, 0.9996, 0.0004, 0, None]
    expect_badarg("sfod_sgd_ema", sgd, [("n", -4), ("param", None), ("grad", None), ("mom", None), ("lr", None)])
    n_contract_bad = sum(not c["ok"] for c in contract)
    print(json.dumps({"entry_points": len(protos), "random_calls": calls, "random_violations": bad,
                      "violations": {k: [str(x)[:300] for x in v[:5]] for k, v in outcomes.items()},
                      "contract_cases": len(contract), "contract_violations": [c for c in contract if not c["ok"]]}))
    return 0 if bad == 0 and n_contract_bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
