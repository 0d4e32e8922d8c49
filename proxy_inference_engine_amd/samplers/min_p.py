"""min_p_sampling (mirror of samplers/min_p.py:8-60 of the reference)."""
from __future__ import annotations

import torch

from .. import hip_ops


def min_p_sampling(logprobs: torch.Tensor, min_p: float, min_tokens_to_keep: int = 1, temperature: float = 1.0) -> torch.Tensor:
    """Keeps the tokens whose probability is at least min_p times the top token's (and always the first `min_tokens_to_keep` of the
    descending order), then samples among them (min_p.py:37-60).  One sort-free HIP kernel (csrc/sampler.hip); device tensors only.
    min_p == 0 keeps every token: the categorical kernel."""
    if not (0 <= min_p <= 1.0):
        raise ValueError(f"`min_p` has to be a float in the [0, 1] interval, but is {min_p}")
    if not isinstance(min_tokens_to_keep, int) or (min_tokens_to_keep < 1):
        raise ValueError(f"`min_tokens_to_keep` has to be a positive integer, but is {min_tokens_to_keep}")
    if min_p == 0:
        return hip_ops.sample(logprobs, "categorical", temperature)
    return hip_ops.sample(logprobs, "min_p", temperature, p=min_p, k=min_tokens_to_keep)
