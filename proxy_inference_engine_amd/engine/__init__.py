from .batch_engine import BatchedEngine
from .inference_engine import InferenceEngine

__all__ = ["InferenceEngine", "BatchedEngine"]
