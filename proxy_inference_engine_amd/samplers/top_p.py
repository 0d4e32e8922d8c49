"""top_p_sampling (mirror of samplers/top_p.py:7-34 of the reference)."""
from __future__ import annotations

import torch

from .categorical import sample_from_logits


def top_p_sampling(logits: torch.Tensor, top_p: float, temperature: float) -> torch.Tensor:
    """Nucleus sampling exactly as the reference writes it: ascending sort, cumulative sum, keep the tokens whose
    cumulative probability exceeds 1 - top_p (top_p.py:18-31), sample among them, map back to vocabulary ids."""
    if logits.is_cuda:  # the product path: one HIP kernel (csrc/sampler.hip), no sort
        from .. import hip_ops
        return hip_ops.sample(logits, "top_p", temperature, p=top_p)
    probs = torch.softmax(logits.float() * (1 / temperature), dim=-1)
    sorted_probs, sorted_indices = torch.sort(probs, dim=-1)              # ascending, like mx.argsort
    cumulative = torch.cumsum(sorted_probs, dim=-1)
    top_probs = torch.where(cumulative > 1 - top_p, sorted_probs, torch.zeros_like(sorted_probs))
    sorted_tokens = sample_from_logits(torch.log(top_probs))[..., None]
    return sorted_indices.gather(-1, sorted_tokens.long()).squeeze(-1).to(torch.int32)
