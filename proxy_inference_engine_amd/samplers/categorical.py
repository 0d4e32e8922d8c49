"""categorical_sampling (mirror of samplers/categorical.py:6-7 of the reference): mx.random.categorical(logits / temp)."""
from __future__ import annotations

import torch

from .. import hip_ops


def categorical_sampling(logits: torch.Tensor, temp: float) -> torch.Tensor:
    """One index ~ softmax(logits / temp) per row (Gumbel-max with a Philox stream, csrc/sampler.hip); device tensors only."""
    return hip_ops.sample(logits, "categorical", temp)
