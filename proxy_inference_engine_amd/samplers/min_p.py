"""min_p_sampling (mirror of samplers/min_p.py:8-60 of the reference)."""
from __future__ import annotations

import math

import torch

from .categorical import sample_from_logits


def min_p_sampling(logprobs: torch.Tensor, min_p: float, min_tokens_to_keep: int = 1, temperature: float = 1.0) -> torch.Tensor:
    """Keeps the tokens whose probability is at least min_p times the top token's (and always the first
    `min_tokens_to_keep` of the descending order), then samples among them (min_p.py:37-60)."""
    if not (0 <= min_p <= 1.0):
        raise ValueError(f"`min_p` has to be a float in the [0, 1] interval, but is {min_p}")
    if not isinstance(min_tokens_to_keep, int) or (min_tokens_to_keep < 1):
        raise ValueError(f"`min_tokens_to_keep` has to be a positive integer, but is {min_tokens_to_keep}")
    if logprobs.is_cuda and min_p > 0:  # the product path: one HIP kernel (csrc/sampler.hip), no sort
        from .. import hip_ops
        return hip_ops.sample(logprobs, "min_p", temperature, p=min_p, k=min_tokens_to_keep)
    logprobs = logprobs.float() * (1 / temperature)
    sorted_logprobs, sorted_indices = torch.sort(logprobs, dim=-1, descending=True)
    scaled_min_p = sorted_logprobs[..., 0:1] + (math.log(min_p) if min_p > 0 else float("-inf"))
    remove = sorted_logprobs < scaled_min_p
    remove[..., :min_tokens_to_keep] = False
    selected = torch.where(remove, torch.full_like(sorted_logprobs, float("-inf")), sorted_logprobs)
    sorted_tokens = sample_from_logits(selected)[..., None]
    return sorted_indices.gather(-1, sorted_tokens.long()).squeeze(-1).to(torch.int32)
