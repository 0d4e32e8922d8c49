"""Sampler random state: one torch.Generator per device, the role mx.random.state plays for the reference's compiled
samplers (samplers/*.py: `@partial(mx.compile, inputs=mx.random.state, outputs=mx.random.state)`)."""
from __future__ import annotations

import torch

_generators: dict[str, torch.Generator] = {}
_seed: int | None = None


def seed(n: int) -> None:
    """mx.random.seed(n): restarts every device's stream."""
    global _seed
    _seed = int(n)
    for g in _generators.values():
        g.manual_seed(_seed)


def generator(device: torch.device) -> torch.Generator:
    key = str(device)
    if key not in _generators:
        g = torch.Generator(device=device)
        g.manual_seed(_seed) if _seed is not None else g.seed()
        _generators[key] = g
    return _generators[key]
