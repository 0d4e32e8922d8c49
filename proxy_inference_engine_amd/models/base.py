"""Shared model plumbing (mirror of models/base.py of the reference).

create_causal_mask / create_attention_mask (base.py:18-53) have no device counterpart on this path: the decode
step has L == 1 (mask None, base.py:39-53) and the prompt is processed by iterated decode steps, which is the
causal mask by construction.  They are kept (torch, host-side) for API parity and for tests.
"""
from __future__ import annotations

import torch
from pydantic import BaseModel, ConfigDict


class BaseModelArgs(BaseModel):
    """config.json -> typed args; unknown keys are ignored (base.py:10-16)."""
    model_config = ConfigDict(extra="ignore", protected_namespaces=())


def create_causal_mask(N: int, offset: int = 0, window_size: int | None = None, lengths=None, device=None):
    rinds = torch.arange(offset + N, device=device)
    linds = torch.arange(offset, offset + N, device=device) if offset else rinds
    mask = linds[:, None] < rinds[None]
    if window_size is not None:
        mask = mask | (linds[:, None] > rinds[None] + window_size)
    if lengths is not None:
        mask = mask | (rinds[None] >= lengths[:, None, None, None])
    return mask * -1e9


def create_attention_mask(h: torch.Tensor, cache=None):
    T = h.shape[1]
    if T <= 1:
        return None
    offset = cache[0].offset if cache is not None and len(cache) > 0 else 0
    return create_causal_mask(T, offset, device=h.device).to(h.dtype)


def sanitize(weights: dict, tie_word_embeddings: bool) -> dict:
    """Model.sanitize (llama/language.py:212-219): drop precomputed rotary tables and a tied lm_head."""
    out = {k: v for k, v in weights.items() if "self_attn.rotary_emb.inv_freq" not in k}
    if tie_word_embeddings:
        for k in [k for k in out if k.startswith("lm_head.")]:
            out.pop(k)
    return out
