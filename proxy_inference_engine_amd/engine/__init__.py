from .inference_engine import InferenceEngine

__all__ = ["InferenceEngine"]
