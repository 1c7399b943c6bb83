"""The C-ABI library loads and exports every symbol include/sfod_hip.h declares (no compute)."""
import ctypes
import os
import subprocess
import sys


def test_header_parses(sfod):
    protos = sfod.native.parse_header()
    assert len(protos) >= 40
    for name in ("sfod_conv_fwd", "sfod_conv_wgrad", "sfod_nms", "sfod_anchor_match", "sfod_roi_align_fwd",
                 "sfod_roi_align_bwd", "sfod_sgd_ema", "sfod_bn_relu_pool_bwd", "sfod_frcnn_loss"):
        assert name in protos
    assert protos["sfod_sort_ws_bytes"][0] is ctypes.c_int64


def test_library_exports_every_declared_symbol(sfod):
    so = sfod.native.SO_PATH
    if not os.path.exists(so):
        subprocess.check_call([sys.executable, os.path.join(os.path.dirname(so), "..", "csrc", "build.py")])
    lib = sfod.native.load()
    for name in sfod.native.parse_header():
        assert hasattr(lib, name), name
    assert lib.sfod_version() >= 100
    assert lib.sfod_last_error() == b""


def test_missing_library_fails_loudly(sfod, tmp_path):
    import pytest
    saved = sfod.native._lib
    sfod.native._lib = None
    try:
        with pytest.raises(sfod.native.NativeLibraryError):
            sfod.native.load(str(tmp_path / "nope.so"))
    finally:
        sfod.native._lib = saved


def test_launch_planners_answer_without_a_gpu(sfod):
    """The shape queries of the library are host logic: which layers the BatchNorm-input form serves and where the split-K
    form of a linear layer asks for scratch (include/sfod_hip.h: sfod_conv_fwd_bnin_supported, sfod_conv_fwd_scratch_bytes)."""
    n = sfod.native
    lib = n.load()
    # BatchNorm-input convolution: bf16x3 only, only where the planner runs the 256 x 128 shape (enough tiles to fill the chip)
    assert lib.sfod_conv_fwd_bnin_supported(8, 150, 300, 256, 256, n.BF16X3) == 1       # conv3_2 of a teacher batch
    assert lib.sfod_conv_fwd_bnin_supported(8, 300, 600, 128, 128, n.BF16X3) == 1       # conv2_2
    assert lib.sfod_conv_fwd_bnin_supported(1, 150, 300, 256, 256, n.BF16X3) == 0       # one frame: the 64-channel shape runs
    assert lib.sfod_conv_fwd_bnin_supported(8, 150, 300, 256, 256, n.F16X3) == 0
    assert lib.sfod_conv_fwd_bnin_supported(8, 150, 300, 256, 256, n.F32) == 0
    assert lib.sfod_conv_fwd_bnin_supported(8, 600, 1200, 8, 64, n.BF16X3) == 0         # first layer: its own kernel
    assert lib.sfod_conv_fwd_bnin_supported(0, 150, 300, 256, 256, n.BF16X3) == 0
    # split-K: linear layers (ksize 1, fp32 out, no statistics) with few rows and a long K
    for dt in (n.BF16X3, n.F16X3, n.F32):
        b = lib.sfod_conv_fwd_scratch_bytes(512, 1, 1, 25088, 1024, 1, dt, n.F32, 0)    # fc1 of the ROI head, one frame per GPU
        assert b > 0 and b % (512 * 1024 * 4) == 0 and 2 <= b // (512 * 1024 * 4) <= 8
        assert lib.sfod_conv_fwd_scratch_bytes(4096, 1, 1, 25088, 1024, 1, dt, n.F32, 0) == 0      # eight frames: fills the chip
        assert lib.sfod_conv_fwd_scratch_bytes(512, 1, 1, 1024, 1024, 1, dt, n.F32, 0) == 0        # short K
        assert lib.sfod_conv_fwd_scratch_bytes(512, 1, 1, 25088, 1024, 1, dt, n.F32, 1) == 0       # statistics wanted
        assert lib.sfod_conv_fwd_scratch_bytes(1, 37, 75, 512, 512, 3, dt, n.F32, 0) == 0          # 3x3: never
    assert lib.sfod_conv_fwd_scratch_bytes(512, 1, 1, 25088, 1024, 1, n.BF16, n.BF16, 0) == 0      # 2-byte output: unsplit
    assert lib.sfod_last_error() == b""


def test_c_abi_host_side_survives_hostile_arguments_under_asan_and_ubsan(sfod, tmp_path):
    """The library's HOST code (argument validation, launch planners, ``*_supported`` / ``*_bytes`` / ``*_blocks`` queries)
    built with -fsanitize=address,undefined and no device code (csrc/build.py::build_host_sanitized), fuzzed by
    tests/helpers/abi_fuzz.py in a child process under the ASan runtime: random hostile vectors into every entry point
    (negative / huge / misaligned sizes, NULL pointers, NaN floats) produce no sanitizer report and no crash, every
    SFOD_EBADARG comes with a message, no query answers with a negative number; and for the entry points the hot path
    calls, each single broken argument of an otherwise valid call (size < 0, channels not a multiple of the 8-group,
    ldy < Cout, NULL operand, unknown dtype) returns -1000 before any launch."""
    import importlib.util
    import json
    import pytest
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("sfod_csrc_build", os.path.join(here, "..", "simple-sfod_amd", "csrc", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    if not os.path.exists(b.HIPCC):
        pytest.skip("no hipcc")
    rt = b.asan_runtime()
    if rt is None:
        pytest.skip("no shared ASan runtime in this toolchain")
    so = b.build_host_sanitized()
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="halt_on_error=0:detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=0")
    r = subprocess.run([sys.executable, os.path.join(here, "helpers", "abi_fuzz.py"), so, "200"], env=env, cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    err = r.stderr.decode(errors="replace")
    rep = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert "runtime error" not in err and "AddressSanitizer" not in err, err[-3000:]
    assert r.returncode == 0, (rep, err[-2000:])
    assert rep["entry_points"] >= 85 and rep["random_calls"] >= 15000 and rep["random_violations"] == 0
    assert rep["contract_cases"] >= 70 and rep["contract_violations"] == []
