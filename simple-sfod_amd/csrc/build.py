"""Build libsfod_hip.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]

hipcc cross-compiles without a GPU; the built library is git-ignored but travels with the
source tree to the GPU box.  detect/roi_align are built with -ffp-contract=off so that box /
IoU arithmetic rounds exactly like the reference's fp32 torch ops.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "lib")
SO = os.path.join(OUT_DIR, "libsfod_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = {
    "detect.hip": ["-ffp-contract=off"],
    "roi_align.hip": ["-ffp-contract=off"],
    "elementwise.hip": [],
    "gemm_conv.hip": [],
    "conv3x3_patch.hip": [],
    "wgrad3x3_patch.hip": [],
    "conv_first.hip": [],
    "stem7x7.hip": [],
    "sort.hip": [],
    "augment.hip": ["-ffp-contract=off"],
    "runtime.cpp": [],
}
HEADERS = ["common.h", "conv_internal.h", os.path.join("..", "..", "include", "sfod_hip.h")]


def source_fingerprint():
    """sha256 over the kernel sources and headers (names + contents, sorted): what a measurement of the kernels (the PMC
    captures under profiles/) is stamped with, so that a stale capture is recognised after any source change.  (The GPU box
    has no .git: a content hash instead of ``git rev-parse HEAD:simple-sfod_amd/csrc``.)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(HERE) if f.endswith((".hip", ".h", ".cpp")))
    for f in files + [os.path.join("..", "..", "include", "sfod_hip.h")]:
        h.update(os.path.basename(f).encode())
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    obj_dir = os.path.join(OUT_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        sp = os.path.join(HERE, src)
        op = os.path.join(obj_dir, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + hdrs + [os.path.abspath(__file__)]):
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value",
                   "-c", sp, "-o", op] + extra
            if src.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(SO, objs):
        run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", SO] + objs)
    return SO


def build_host_sanitized(verbose=False):
    """HOST side of the library only (``--cuda-host-only``: argument validation, launch planners, the ``*_supported`` /
    ``*_bytes`` / ``*_blocks`` queries; kernels are host stubs that can never run) with AddressSanitizer +
    UndefinedBehaviorSanitizer -> lib/libsfod_hip_hostsan.so.  CPU test infrastructure (tests/test_abi.py fuzzes the C ABI
    through it under LD_PRELOAD of the ASan runtime); never loaded by the product.  GPU sanitizers are not available on
    this pool, and the device code is not what this checks."""
    os.makedirs(OUT_DIR, exist_ok=True)
    obj_dir = os.path.join(OUT_DIR, "obj_hostsan")
    os.makedirs(obj_dir, exist_ok=True)
    so = os.path.join(OUT_DIR, "libsfod_hip_hostsan.so")
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    san = ["-fsanitize=address,undefined", "-fsanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    jobs, objs = [], []
    for src, extra in SOURCES.items():
        sp = os.path.join(HERE, src)
        op = os.path.join(obj_dir, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        if _stale(op, [sp] + hdrs + [os.path.abspath(__file__)]):
            cmd = [HIPCC, f"--offload-arch={ARCH}", "--cuda-host-only", "-fPIC", "-std=c++17", "-Wno-unused-value"] + san + \
                  ["-c", sp, "-o", op] + extra
            if src.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or _stale(so, objs):
        # a host-only object still references its device blob (__hip_fatbin_<hash>): give each an EMPTY clang offload
        # bundle (magic + zero entries) -- the HIP runtime finds no code object in it, so a launch returns an error
        syms = set()
        for o in objs:
            for line in subprocess.check_output(["nm", "-u", o]).decode().splitlines():
                parts = line.split()
                if parts and parts[-1].startswith("__hip_fatbin_"):
                    syms.add(parts[-1])
        stub_c = os.path.join(obj_dir, "fatbin_stub.c")
        with open(stub_c, "w") as f:
            f.write("/* generated by build.py::build_host_sanitized */\n")
            for sname in sorted(syms):
                f.write('__attribute__((aligned(4096))) const char %s[32] = "__CLANG_OFFLOAD_BUNDLE__";\n' % sname)
        stub_o = os.path.join(obj_dir, "fatbin_stub.o")
        run(["gcc", "-c", "-fPIC", stub_c, "-o", stub_o])
        run([HIPCC, "-shared", "-fPIC", "-shared-libsan", "-o", so] + san + objs + [stub_o])
    return so


def asan_runtime():
    """path of the shared ASan runtime to LD_PRELOAD when a non-instrumented executable (python) loads the library"""
    out = subprocess.check_output([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


if __name__ == "__main__":
    if "--host-sanitized" in sys.argv:
        print(build_host_sanitized(verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv))
