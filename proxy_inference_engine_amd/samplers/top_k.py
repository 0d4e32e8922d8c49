"""top_k_sampling (mirror of samplers/top_k.py:7-29 of the reference)."""
from __future__ import annotations

import torch

from .. import hip_ops


def top_k_sampling(logprobs: torch.Tensor, top_k: int, temperature: float = 1.0) -> torch.Tensor:
    """Sample from only the top K tokens ranked by probability; everything else is masked out (top_k.py:24-29).  One HIP kernel
    (csrc/sampler.hip: radix select instead of argpartition); device tensors only."""
    vocab_size = logprobs.shape[-1]
    if not isinstance(top_k, int) or not (0 < top_k < vocab_size):
        raise ValueError(f"`top_k` has to be an integer in the (0, {vocab_size}] interval, but is {top_k}.")
    return hip_ops.sample(logprobs, "top_k", temperature, k=top_k)
