"""Repetition penalty (mirror of logits_processors/repetition.py:6-24 of the reference).

Only installed when repetition_penalty != 1.0 (engine/inference_engine.py:328-333), i.e. off the measured
greedy path; it edits <= context_size logits in place, host-orchestrated on device tensors."""
from __future__ import annotations

from collections.abc import Callable

import torch


def make_repetition_penalty(penalty: float = 1.0, context_size: int = 60) -> Callable:
    if penalty < 0 or context_size < 0:
        raise ValueError(f"Parameters must be non-negative, got penalty={penalty} and context_size={context_size}")

    def repetition_penalty_processor(tokens, logits: torch.Tensor) -> torch.Tensor:
        if len(tokens) > 0:
            idx = torch.as_tensor(list(tokens[-context_size:]), dtype=torch.long, device=logits.device)
            sel = logits[:, idx]
            sel = torch.where(sel < 0, sel * penalty, sel / penalty)
            logits[:, idx] = sel
        return logits

    return repetition_penalty_processor
