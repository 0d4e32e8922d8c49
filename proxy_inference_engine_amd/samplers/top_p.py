"""top_p_sampling (mirror of samplers/top_p.py:7-34 of the reference)."""
from __future__ import annotations

import torch

from .. import hip_ops


def top_p_sampling(logits: torch.Tensor, top_p: float, temperature: float) -> torch.Tensor:
    """Nucleus sampling as the reference defines it -- ascending order, cumulative sum, keep the tokens whose cumulative probability exceeds
    1 - top_p (top_p.py:18-31), sample among them -- as ONE sort-free HIP kernel (csrc/sampler.hip); device tensors only."""
    return hip_ops.sample(logits, "top_p", temperature, p=top_p)
