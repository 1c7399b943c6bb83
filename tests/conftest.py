import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "sweep: exhaustive tile / variant / shape sweeps of one kernel family -- not part of "
                            "the default selections; run with -m \"gpu and sweep\" (the builder does, through gpurun; "
                            "summaries under profiles/)")
    # SFOD_BF16X3 tensors are tagged torch.complex32 (native.SPLIT_DTYPE); torch only allocates / views them
    config.addinivalue_line("filterwarnings", "ignore:ComplexHalf support is experimental")


@pytest.fixture(scope="session")
def sfod():
    """The product package (directory name is not an identifier)."""
    return importlib.import_module("simple-sfod_amd")


@pytest.fixture(scope="session")
def native(sfod):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sfod.native.load()
    return sfod.native


GOLDEN = os.path.join(ROOT, "tests", "golden")


# Which cases are sweeps (one table instead of marks scattered over the parametrisations).  What stays in the default GPU
# selection: every parity gate at BASELINE sizes in the mode bench.py reports for that config, every golden-fixture test, the
# two-rank run of config #4, and ONE case per kernel family and variant axis (the planner's own choice); the exhaustive
# axes -- forced workgroup shapes, forced algorithms, the other arithmetic modes of the long oracle comparisons -- run
# with -m "gpu and sweep".  Durations that decided the cut: gpurun_out / profiles/round5/r5a_gpu_suite_durations.txt.
import re

SWEEP_PATTERNS = [re.compile(p) for p in (
    # forced workgroup shapes of the halo-patch convolution: keep auto (0) and the 16x16x32 defaults (5: 128-channel, 7: 64-channel tiles)
    r"test_conv3x3_patch_kernel\[(1|2|3|4|6|8|9)-",
    r"test_gpu_bf16x3\.py::test_dgrad_with_fused_batchnorm_backward_reduction\[(1|2|4|5|6|8|9)-",
    r"test_gpu_bf16x3\.py::test_conv_dgrad_and_wgrad\[(1|4)-",                  # forced algorithms; 0 = the planner
    r"test_gpu_bf16x3\.py::test_conv3x3_patch_wgrad\[(1|0)-",                   # the non-default loops of the weight gradient
    r"test_gpu_bf16x3\.py::test_conv3x3_with_batchnorm_folded_into_its_input\[(2|6)",
    # long oracle comparisons: the default keeps the mode bench.py reports for the config (+ one fp32 / non-elided case)
    r"test_three_steps_match_the_oracle_trajectory\[(vgg-fp32-True|r101-fp32-True|vgg-f16x3-True|vgg-f16x3-False)\]",
    # two-rank steps: config #4 and #5 each in the mode bench.py reports for it; config #4 in fp32 is the sweep's
    r"test_two_ranks_on_one_gpu_take_the_oracles_mean_gradient_step\[vgg-fp32\]",
    r"test_bench_two_ranks_on_one_gpu\[r101\]",
    # (the hot yaml at full size: the default keeps the cases on the head bench.py times, bf16x3 at B = 8 and fp32; the (60, 20)
    # head's cases -- same sizes, same gates, another score distribution -- run in the sweep)
    r"test_hot_yaml_teacher_and_student_at_600x1200\[(f16x3-2|bf16x3-2|bf16x3-8|fp32-2)\]",
    # (R101 at full size: the default keeps the case on the head bench.py times; the (60, 20) head's cases are the sweep's)
    r"test_r101_yaml_teacher_and_student_at_600x1200\[(fp32|bf16x3|f16x3)\]",
    r"test_r101_yaml_on_the_benchmarked_head_at_600x1200\[fp32\]",
    r"test_r101_yaml_on_a_well_conditioned_weight_set_at_600x1200\[(fp32|f16x3)\]",   # default: the yaml's own weights, f16x3
)]


def pytest_itemcollected(item):
    if any(p.search(item.nodeid) for p in SWEEP_PATTERNS):
        item.add_marker(pytest.mark.sweep)


def pytest_collection_modifyitems(config, items):
    """``sweep`` tests run only when the -m expression names them: ``-m gpu`` (the driver's selection: parity gates at
    BASELINE sizes, golden fixtures, one case per kernel family) stays well inside its time limit, ``-m "gpu and sweep"``
    runs the exhaustive tile / variant sweeps."""
    if "sweep" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("sweep") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
