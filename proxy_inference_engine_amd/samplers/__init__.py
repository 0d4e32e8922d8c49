"""Samplers (mirror of samplers/__init__.py:11-46 of the reference).

`make_sampler(temp=0)` -- the greedy branch (samplers/__init__.py:37-38) -- is the parity target of the decode
path and runs in the fused HIP tail (argmax of the fp32 log-probabilities, first maximal index).
The stochastic branches (top-p, min-p, top-k, categorical: samplers/{top_p,min_p,top_k,categorical}.py) are the
"next" row SURVEY.md 8f-4; asking for them raises NotImplementedError instead of silently sampling differently.
"""
from __future__ import annotations

from collections.abc import Callable

import torch

from .. import hip_ops


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
    raise NotImplementedError(
        "only the greedy sampler (temp=0) is on the MI355X decode path; top-p/min-p/top-k/categorical sampling is "
        "SURVEY.md 8f-4 (pass temp=0: the reference's default temp=1.0 samples, inference_engine.py:305)")


greedy.is_greedy = True  # lets the engine pick the fused tail
