"""Drop-in alias: `import proxy_inference_engine` resolves to the MI355X implementation, so code written
against the reference's package (InferenceEngine, pie_core.hello(), .cache, .samplers, .models ...) runs unchanged."""
import importlib
import sys

import proxy_inference_engine_amd as _impl

for _name in ("engine", "cache", "cache.kv_cache", "cache.prompt_cache", "samplers", "logits_processors", "models",
              "models.base", "models.utils", "models.llama", "models.llama.language", "models.llama.utils", "pie_core"):
    sys.modules[f"{__name__}.{_name}"] = importlib.import_module(f"proxy_inference_engine_amd.{_name}")

InferenceEngine = _impl.InferenceEngine
pie_core = _impl.pie_core
__all__ = ["InferenceEngine", "pie_core"]
