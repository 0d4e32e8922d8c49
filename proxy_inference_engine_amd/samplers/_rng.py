"""Sampler random state: the role mx.random.state plays for the reference's compiled samplers
(samplers/*.py: `@partial(mx.compile, inputs=mx.random.state, outputs=mx.random.state)`): a 64-bit seed plus a device-side call counter
per device for the HIP sampler's Philox stream (csrc/sampler.hip)."""
from __future__ import annotations

import torch

_hip_counters: dict[str, torch.Tensor] = {}
_hip_seed: int | None = None


def hip_state(device: torch.device) -> tuple[int, torch.Tensor]:
    """(seed, counter) for pie_sample on `device`: counter is a device int64[2] the kernel advances per call."""
    global _hip_seed
    if _hip_seed is None:
        import os
        _hip_seed = int.from_bytes(os.urandom(8), "little") & (2 ** 64 - 1)
    key = str(device)
    if key not in _hip_counters:
        _hip_counters[key] = torch.zeros(2, dtype=torch.int64, device=device)
    return _hip_seed, _hip_counters[key]


def seed(n: int) -> None:
    """mx.random.seed(n): restarts every device's stream."""
    global _hip_seed
    _hip_seed = int(n) & (2 ** 64 - 1)
    for c in _hip_counters.values():
        c.zero_()
