"""Drop-in alias: `import proxy_inference_engine` resolves to the MI355X implementation, so code written
against the reference's package (InferenceEngine, pie_core.hello(), .cache, .samplers, .models ...) runs unchanged.

Every `proxy_inference_engine.X[.Y...]` import returns the SAME module object as `proxy_inference_engine_amd.X[.Y...]`
(a meta-path finder, not copies: sampler random state, loaded libraries and caches are shared between the two names)."""
import importlib
import importlib.abc
import importlib.util
import sys

import proxy_inference_engine_amd as _impl

_PREFIX, _REAL = __name__ + ".", _impl.__name__ + "."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        try:
            importlib.import_module(_REAL + fullname[len(_PREFIX):])
        except ImportError:
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        return sys.modules[_REAL + spec.name[len(_PREFIX):]]

    def exec_module(self, module):  # already executed under its real name
        return None


sys.meta_path.insert(0, _AliasFinder())

InferenceEngine = _impl.InferenceEngine
pie_core = _impl.pie_core
__all__ = ["InferenceEngine", "pie_core"]
