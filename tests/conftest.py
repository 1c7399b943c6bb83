import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
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
