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
