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
