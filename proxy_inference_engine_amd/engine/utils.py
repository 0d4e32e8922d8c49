"""engine/utils.py of the reference (:4-48): `get_top_logprobs`, importable from the same path."""
from .inference_engine import get_top_logprobs

__all__ = ["get_top_logprobs"]
