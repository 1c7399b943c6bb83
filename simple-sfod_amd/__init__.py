"""simple-sfod_amd: MI355X-native teacher-student Faster R-CNN self-training hot path.

A from-scratch drop-in for the adaptation hot path of EPFL-IMOS/simple-SFOD (reference layer
map: SURVEY.md section 8): the reference's registry / config surface on the host side, the per-step
compute as hand-written HIP kernels for gfx950 behind the C ABI in ``include/sfod_hip.h``.

The directory name is not a Python identifier; import it with
``importlib.import_module("simple-sfod_amd")`` (``tests/conftest.py`` and ``bench.py`` do).
"""
__version__ = "0.1.0"

from . import config, structures, native, registry  # noqa: F401
from . import modeling, engine, data, checkpoint, evaluation  # noqa: F401  (registers the architectures by import)
