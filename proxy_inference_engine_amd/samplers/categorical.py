"""categorical_sampling (mirror of samplers/categorical.py:6-7 of the reference): mx.random.categorical(logits / temp)."""
from __future__ import annotations

import torch

from . import _rng


def sample_from_logits(logits: torch.Tensor) -> torch.Tensor:
    """mx.random.categorical over the last axis: one index ~ softmax(logits) per row (the Gumbel-max form MLX uses:
    argmax(logits + G), G = -log(-log(U))); -inf entries are never drawn.  Runs on the tensor's device, no host sync."""
    x = logits.float()
    u = torch.rand(x.shape, generator=_rng.generator(x.device), device=x.device, dtype=torch.float32)
    g = -torch.log(-torch.log(u.clamp_(min=torch.finfo(torch.float32).tiny, max=1.0 - 2.0 ** -24)))
    return torch.argmax(x + g, dim=-1).to(torch.int32)


def categorical_sampling(logits: torch.Tensor, temp: float) -> torch.Tensor:
    if logits.is_cuda:  # the product path: one HIP kernel (csrc/sampler.hip); the torch ops below serve host tensors (CPU tests) only
        from .. import hip_ops
        return hip_ops.sample(logits, "categorical", temp)
    return sample_from_logits(logits * (1 / temp))
