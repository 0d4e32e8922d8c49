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


# ---- the HIP sampler's stream (csrc/sampler.hip): a 64-bit seed + a device-side call counter per device
_hip_counters: dict[str, torch.Tensor] = {}
_hip_seed: int | None = None


def hip_state(device: torch.device) -> tuple[int, torch.Tensor]:
    """(seed, counter) for pie_sample on `device`: counter is a device int64[2] the kernel advances per call."""
    global _hip_seed
    if _hip_seed is None:
        import os
        _hip_seed = (_seed if _seed is not None else int.from_bytes(os.urandom(8), "little")) & (2 ** 64 - 1)
    key = str(device)
    if key not in _hip_counters:
        _hip_counters[key] = torch.zeros(2, dtype=torch.int64, device=device)
    return _hip_seed, _hip_counters[key]


_torch_seed = seed


def seed(n: int) -> None:  # noqa: F811 -- extends the torch-stream seed() above to the HIP stream
    """mx.random.seed(n): restarts every device's stream (the torch generators of the CPU mirror and the HIP sampler's counters)."""
    global _hip_seed
    _torch_seed(n)
    _hip_seed = int(n) & (2 ** 64 - 1)
    for c in _hip_counters.values():
        c.zero_()
