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
