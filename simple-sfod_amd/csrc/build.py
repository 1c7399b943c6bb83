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
    "sort.hip": [],
    "augment.hip": ["-ffp-contract=off"],
    "runtime.cpp": [],
}
HEADERS = ["common.h", "conv_internal.h", os.path.join("..", "..", "include", "sfod_hip.h")]


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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
