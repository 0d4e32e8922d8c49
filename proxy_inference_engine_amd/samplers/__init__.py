"""Samplers (mirror of samplers/__init__.py:11-46 of the reference).

`make_sampler(temp=0)` -- the greedy branch (samplers/__init__.py:37-38) -- is the parity target of the decode
path and runs in the fused HIP tail (argmax of the fp32 log-probabilities, first maximal index).
The stochastic branches (top-p, min-p, top-k, categorical: samplers/{top_p,min_p,top_k,categorical}.py, SURVEY.md
8f-4) are sort-free HIP kernels (csrc/sampler.hip) on the device that holds the log-probabilities (no host sync: the
drawn token stays a device tensor and feeds the next step; host tensors are refused -- the reference's definitions restated
with torch ops live in tests/sampler_reference.py as the comparator); their random stream is Philox, not MLX's, so parity
is the kept-token set (exact) and the distribution, not the individual draw.  `seed(n)` = mx.random.seed(n).
"""
from __future__ import annotations

from collections.abc import Callable

import torch

from .. import hip_ops
from ._rng import seed
from .categorical import categorical_sampling
from .min_p import min_p_sampling
from .top_k import top_k_sampling
from .top_p import top_p_sampling


def greedy(logprobs: torch.Tensor) -> torch.Tensor:
    """logprobs [B=1, V] (fp32 or 16-bit) -> token ids [1] int32: argmax over the last axis."""
    x = logprobs.reshape(-1)
    if x.dtype == torch.float32:
        # argmax of fp32 log-probabilities: monotone in the logits, done by the same tail kernel on a 16-bit
        # view is not possible, so use the fp32 path of the tail kernel's host mirror
        return _argmax_f32(x)
    tok, _ = hip_ops.logprobs_argmax(x)
    return tok


def _argmax_f32(x: torch.Tensor) -> torch.Tensor:
    # first maximal index (mx.argmax semantics); tiny host-glue reduction over V values
    m = x.max()
    idx = torch.nonzero(x == m)[0, 0]
    return idx.to(torch.int32).reshape(1)


def make_sampler(temp: float = 0.0, top_p: float = 0.0, min_p: float = 0.0, min_tokens_to_keep: int = 1,
                 top_k: int = -1) -> Callable[[torch.Tensor], torch.Tensor]:
    if temp == 0:
        return greedy
    elif top_p > 0 and top_p < 1.0:
        return lambda x: top_p_sampling(x, top_p, temp)
    elif min_p != 0.0:
        return lambda x: min_p_sampling(x, min_p, min_tokens_to_keep, temp)
    elif top_k > 0:
        return lambda x: top_k_sampling(x, top_k, temp)
    else:
        return lambda x: categorical_sampling(x, temp)


greedy.is_greedy = True  # lets the engine pick the fused tail
