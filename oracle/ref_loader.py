"""oracle/ref_loader.py -- TEST INFRASTRUCTURE (fixture generation only, build container only).

Imports the reference's numpy-only functions from /root/reference/FunscriptFlow.pyw with its
GUI / OpenCV imports replaced by inert stub modules.  Used ONLY by
oracle/gen_golden.py to produce tests/golden/*.npz; nothing in tests/, bench.py or the product
imports this at run time (the reference does not exist on the GPU box).
"""
import importlib.machinery
import importlib.util
import sys
import types

REF_PATH = "/root/reference/FunscriptFlow.pyw"


class _Stub(types.ModuleType):
    """Module whose every attribute is a fresh dummy class accepting any constructor args."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        class _Dummy:
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                return _Dummy()

            def __getattr__(self, n):
                if n.startswith("__"):
                    raise AttributeError(n)
                return _Dummy()

        _Dummy.__name__ = name
        return _Dummy


def load_reference(cv2_module=None):
    sys.dont_write_bytecode = True
    names = ["cv2", "PySide6", "PySide6.QtWidgets", "PySide6.QtCore", "PySide6.QtGui", "PySide6.QtMultimedia",
             "PySide6.QtMultimediaWidgets", "matplotlib.backends.backend_qt5agg"]
    for n in names:
        if n == "cv2" and cv2_module is not None:
            sys.modules[n] = cv2_module
        elif n not in sys.modules:
            sys.modules[n] = _Stub(n)
    loader = importlib.machinery.SourceFileLoader("funscriptflow_reference", REF_PATH)
    spec = importlib.util.spec_from_loader("funscriptflow_reference", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod
