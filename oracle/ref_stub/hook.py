"""Import hook for ``oracle/gen_golden.py`` (TEST SCAFFOLDING, build container only).

The reference's own glue (``daod/engine/trainers/source_free_adaptive_teacher.py``,
``daod/modeling/roi_heads/source_free_fast_rcnn.py``, ``daod/data/common.py`` ...) is a few
dozen lines of plain torch per function, but every file starts with twenty ``from detectron2...`` /
``from fvcore...`` / ``from daod...`` imports and neither library is installed.  With this hook
installed such a file can be loaded BY FILE PATH and the functions the hot path owns can be RUN
(unbound, on a stub ``self``), so their outputs can be recorded as golden vectors.

What the hook provides
  * for every module name under ``FABRICATED_ROOTS`` that has no real stub on disk: an empty module
    whose attributes are made up on access -- permissive placeholder classes that can be subclassed,
    used as decorators (``@configurable``, ``@X_REGISTRY.register()``) and called, and that do nothing;
  * the three containers the recorded functions actually touch, in ``detectron2.structures``
    (``Boxes``, ``Instances``) and ``detectron2.utils.comm`` (``get_world_size`` -> the value set with
    ``set_world_size``): a few lines each, written from d2's documented behaviour.

It contains no Detectron2 / fvcore / reference code, is never imported by the product package, and does
not travel as anything but this scaffolding (only the recorded ``.npz`` data is used on the GPU box).
"""
import importlib.abc
import importlib.machinery
import inspect
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))

FABRICATED_ROOTS = ("detectron2", "fvcore", "daod", "pycocotools")
_WORLD = [1]


def set_world_size(n):
    _WORLD[0] = int(n)


class _FabMeta(type):
    """class-level behaviour of a placeholder: ``X.anything`` is another placeholder, ``X(fn)`` is ``fn``"""

    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return fabricate(cls.__name__ + "." + name)

    def __call__(cls, *a, **k):
        if "_fabricated" in cls.__dict__:
            if len(a) == 1 and not k and (inspect.isfunction(a[0]) or inspect.isclass(a[0])):
                return a[0]                      # used as a bare decorator
            return _FabInstance(cls.__name__)
        return super().__call__(*a, **k)         # a reference class deriving from a placeholder: normal construction


class _FabInstance:
    def __init__(self, name):
        self.__dict__["_name"] = name

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and (inspect.isfunction(a[0]) or inspect.isclass(a[0])):
            return a[0]                          # ``@REGISTRY.register()`` / ``@configurable(from_config=...)``
        return _FabInstance(self._name + "()")

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _FabInstance(self._name + "." + name)

    def __iter__(self):
        return iter(())

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def fabricate(name):
    return _FabMeta(name, (object,), {"_fabricated": True, "__init__": lambda self, *a, **k: None})


class _FabModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        v = fabricate(name)
        setattr(self, name, v)
        return v


# ---- the real (small) pieces: detectron2.structures is the on-disk stub next to this file -----------------------
class RecordingNode(dict):
    """stands in for ``detectron2.config.CfgNode`` while ``daod/config.py::add_config`` runs: attribute = key, a
    node that is read before it exists is created (the d2 defaults ``_C.TEST`` / ``_C.SOLVER`` ... are not here), so
    afterwards it holds exactly the keys and values the reference's function assigned"""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name not in self:
            self[name] = RecordingNode()
        return self[name]

    def __setattr__(self, name, value):
        self[name] = value


def _real_modules():
    comm = {"get_world_size": lambda: _WORLD[0], "get_rank": lambda: 0, "is_main_process": lambda: True}
    return {"detectron2.utils.comm": comm, "detectron2.config": {"CfgNode": RecordingNode}}


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self):
        self.real = _real_modules()

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in FABRICATED_ROOTS:
            spec = importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            rel = os.path.join(HERE, *fullname.split("."))
            for cand in (rel + ".py", os.path.join(rel, "__init__.py")):
                if os.path.isfile(cand):         # a written stub next to this file: executed into the permissive module,
                    spec.origin = cand           # so names it does not define are still made up on access
            return spec
        return None

    def create_module(self, spec):
        m = _FabModule(spec.name)
        m.__path__ = [os.path.dirname(spec.origin)] if spec.origin and spec.origin.endswith("__init__.py") else []
        for k, v in self.real.get(spec.name, {}).items():
            setattr(m, k, v)
        return m

    def exec_module(self, module):
        origin = module.__spec__.origin
        if origin:
            module.__file__ = origin
            with open(origin) as f:
                exec(compile(f.read(), origin, "exec"), module.__dict__)


_INSTALLED = []


def install():
    """idempotent; modules of the fabricated roots that were imported from the on-disk stub before are dropped"""
    if _INSTALLED:
        return _INSTALLED[0]
    for name in list(sys.modules):
        if name.split(".")[0] in FABRICATED_ROOTS:
            del sys.modules[name]
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    f = _Finder()
    sys.meta_path.insert(0, f)
    _INSTALLED.append(f)
    return f


def uninstall():
    while _INSTALLED:
        f = _INSTALLED.pop()
        if f in sys.meta_path:
            sys.meta_path.remove(f)
    for name in list(sys.modules):
        if name.split(".")[0] in FABRICATED_ROOTS:
            del sys.modules[name]
