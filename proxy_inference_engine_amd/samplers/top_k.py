"""top_k_sampling (mirror of samplers/top_k.py:7-29 of the reference)."""
from __future__ import annotations

import torch

from .categorical import sample_from_logits


def top_k_sampling(logprobs: torch.Tensor, top_k: int, temperature: float = 1.0) -> torch.Tensor:
    """Sample from only the top K tokens ranked by probability; everything else is masked to -inf (top_k.py:24-29)."""
    vocab_size = logprobs.shape[-1]
    if not isinstance(top_k, int) or not (0 < top_k < vocab_size):
        raise ValueError(f"`top_k` has to be an integer in the (0, {vocab_size}] interval, but is {top_k}.")
    if logprobs.is_cuda:  # the product path: one HIP kernel (csrc/sampler.hip)
        from .. import hip_ops
        return hip_ops.sample(logprobs, "top_k", temperature, k=top_k)
    logprobs = logprobs.float() * (1 / temperature)
    keep = torch.topk(logprobs, top_k, dim=-1).indices                    # argpartition(-logprobs)[..., :top_k]
    masked = torch.full_like(logprobs, float("-inf")).scatter(-1, keep, logprobs.gather(-1, keep))
    return sample_from_logits(masked)
