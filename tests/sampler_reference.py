"""The reference's stochastic samplers restated with torch ops on host tensors -- TEST INFRASTRUCTURE, the comparator of the HIP
sampler kernels (csrc/sampler.hip); the product package holds no CPU path.

samplers/top_p.py:18-31, min_p.py:37-60, top_k.py:24-29, categorical.py:6-7 of the reference, op for op; the random stream is
torch's (a Gumbel-max draw like mx.random.categorical), so what is comparable with the product is the kept-token set and the
distribution, not the individual draw."""
from __future__ import annotations

import math

import torch

_gen = torch.Generator()


def seed(n: int) -> None:
    _gen.manual_seed(int(n))


def sample_from_logits(logits: torch.Tensor) -> torch.Tensor:
    """mx.random.categorical over the last axis: argmax(logits + G), G = -log(-log(U)); -inf entries are never drawn."""
    x = logits.float()
    u = torch.rand(x.shape, generator=_gen, dtype=torch.float32)
    g = -torch.log(-torch.log(u.clamp_(min=torch.finfo(torch.float32).tiny, max=1.0 - 2.0 ** -24)))
    return torch.argmax(x + g, dim=-1).to(torch.int32)


def categorical_sampling(logits: torch.Tensor, temp: float) -> torch.Tensor:
    return sample_from_logits(logits * (1 / temp))


def top_p_sampling(logits: torch.Tensor, top_p: float, temperature: float) -> torch.Tensor:
    probs = torch.softmax(logits.float() * (1 / temperature), dim=-1)
    sorted_probs, sorted_indices = torch.sort(probs, dim=-1)              # ascending, like mx.argsort
    cumulative = torch.cumsum(sorted_probs, dim=-1)
    top_probs = torch.where(cumulative > 1 - top_p, sorted_probs, torch.zeros_like(sorted_probs))
    sorted_tokens = sample_from_logits(torch.log(top_probs))[..., None]
    return sorted_indices.gather(-1, sorted_tokens.long()).squeeze(-1).to(torch.int32)


def top_k_sampling(logprobs: torch.Tensor, top_k: int, temperature: float = 1.0) -> torch.Tensor:
    logprobs = logprobs.float() * (1 / temperature)
    keep = torch.topk(logprobs, top_k, dim=-1).indices                    # argpartition(-logprobs)[..., :top_k]
    masked = torch.full_like(logprobs, float("-inf")).scatter(-1, keep, logprobs.gather(-1, keep))
    return sample_from_logits(masked)


def min_p_sampling(logprobs: torch.Tensor, min_p: float, min_tokens_to_keep: int = 1, temperature: float = 1.0) -> torch.Tensor:
    logprobs = logprobs.float() * (1 / temperature)
    sorted_logprobs, sorted_indices = torch.sort(logprobs, dim=-1, descending=True)
    scaled_min_p = sorted_logprobs[..., 0:1] + (math.log(min_p) if min_p > 0 else float("-inf"))
    remove = sorted_logprobs < scaled_min_p
    remove[..., :min_tokens_to_keep] = False
    selected = torch.where(remove, torch.full_like(sorted_logprobs, float("-inf")), sorted_logprobs)
    sorted_tokens = sample_from_logits(selected)[..., None]
    return sorted_indices.gather(-1, sorted_tokens.long()).squeeze(-1).to(torch.int32)


def make_sampler(temp: float, top_p: float = 0.0, min_p: float = 0.0, min_tokens_to_keep: int = 1, top_k: int = -1):
    """samplers/__init__.py:11-46's dispatch over the restatements above (temp > 0)."""
    if 0 < top_p < 1.0:
        return lambda x: top_p_sampling(x, top_p, temp)
    if min_p != 0.0:
        return lambda x: min_p_sampling(x, min_p, min_tokens_to_keep, temp)
    if top_k > 0:
        return lambda x: top_k_sampling(x, top_k, temp)
    return lambda x: categorical_sampling(x, temp)
