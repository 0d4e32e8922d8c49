"""proxy_inference_engine_amd: MI355X-native decode hot path behind the `proxy_inference_engine` Python API.

Importing the package is cheap and GPU-free; anything that computes goes through libpie_hip.so
(hand-written HIP for gfx950) and raises if the library or the device is missing.
"""
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import pie_core  # noqa: E402
from .engine import InferenceEngine  # noqa: E402

__all__ = ["InferenceEngine", "pie_core"]
