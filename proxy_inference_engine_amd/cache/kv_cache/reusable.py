"""ReusableKVCache on HBM-resident torch buffers.

Behavioural mirror of cache/kv_cache/reusable.py:8-254 of the reference: per-layer K and V buffers
[B, n_kv_heads, capacity, head_dim], zero-initialised, capacity a multiple of `step` (256), first allocation
ceil(needed/step)*step, growth max(int(capacity*growth_factor), required) rounded up to `step`, writes at
[offset, offset+L), `offset += L`, reads are views of the first `offset` positions.

The decode kernels write new K/V rows into these buffers themselves (the QKV GEMV epilogue appends at the
device-side offset), so besides the reference's `update_and_fetch` the class splits that method into its two
halves: `reserve(n)` = the capacity logic (reusable.py:113-131), `advance(n)` = `offset += n` (:139).
"""
from __future__ import annotations

import torch

from . import BaseCache


class ReusableKVCache(BaseCache):
    def __init__(self, step: int = 256, growth_factor: float = 1.5, max_capacity: int | None = None):
        self.keys: torch.Tensor | None = None
        self.values: torch.Tensor | None = None
        self.offset = 0
        self.step = step
        self.growth_factor = growth_factor
        self.max_capacity = max_capacity

    # ------------------------------------------------------------------ capacity
    def _aligned(self, n: int) -> int:
        n = ((n + self.step - 1) // self.step) * self.step
        return n if self.max_capacity is None else min(n, self.max_capacity)

    @property
    def capacity(self) -> int:
        return 0 if self.keys is None else self.keys.shape[2]

    def _reallocate(self, new_capacity: int) -> None:
        B, n_kv, _, kd = self.keys.shape
        vd = self.values.shape[3]
        new_k = torch.zeros((B, n_kv, new_capacity, kd), dtype=self.keys.dtype, device=self.keys.device)
        new_v = torch.zeros((B, n_kv, new_capacity, vd), dtype=self.values.dtype, device=self.values.device)
        keep = min(self.offset, new_capacity)
        new_k[..., :keep, :] = self.keys[..., :keep, :]
        new_v[..., :keep, :] = self.values[..., :keep, :]
        self.keys, self.values = new_k, new_v

    def reuse(self, new_prompt_length: int, common_prefix_length: int) -> None:
        """Trim to the common prefix and make room for the whole new prompt (reusable.py:44-94)."""
        if self.keys is None or self.values is None:
            return
        self.offset = common_prefix_length
        current = self.keys.shape[2]
        if current < new_prompt_length:
            self._reallocate(self._aligned(max(int(current * self.growth_factor), new_prompt_length)))

    def reserve(self, needed: int, n_kv_heads: int, head_dim: int, dtype: torch.dtype, device, batch: int = 1) -> None:
        """Capacity half of update_and_fetch (reusable.py:113-131, 144-203): room for `needed` more positions."""
        if self.keys is None or self.values is None:
            cap = self._aligned(needed)
            self.keys = torch.zeros((batch, n_kv_heads, cap, head_dim), dtype=dtype, device=device)
            self.values = torch.zeros((batch, n_kv_heads, cap, head_dim), dtype=dtype, device=device)
            self.offset = 0
        elif self.offset + needed > self.keys.shape[2]:
            # reusable.py:125-129: at an offset that is not a multiple of `step` the reference first slices the buffers to
            # the offset, so the growth is computed from the OFFSET, not from the old capacity (520 then 300 tokens:
            # 1024, not 1280).  Only the number matters here: the rows beyond the offset are zeros either way.
            current = self.offset if self.offset % self.step != 0 else self.keys.shape[2]
            self._reallocate(self._aligned(max(int(current * self.growth_factor), self.offset + needed)))
        if self.offset + needed > self.keys.shape[2]:
            raise RuntimeError(f"KV cache max_capacity={self.max_capacity} exceeded")

    def advance(self, n: int) -> None:
        self.offset += n

    # ------------------------------------------------------------------ reference protocol
    def update_and_fetch(self, keys: torch.Tensor, values: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """keys/values [B, n_kv, L, D] -> views of the first offset+L positions (reusable.py:96-142)."""
        needed = keys.shape[2]
        prev = self.offset
        self.reserve(needed, keys.shape[1], keys.shape[3], keys.dtype, keys.device, keys.shape[0])
        self.keys[..., prev:prev + needed, :] = keys
        self.values[..., prev:prev + needed, :] = values
        self.offset += needed
        return self.keys[..., :self.offset, :], self.values[..., :self.offset, :]

    @property
    def state(self):
        return self.keys, self.values

    @state.setter
    def state(self, v):
        self.keys, self.values = v
        self.offset = self.keys.shape[2] if self.keys is not None else 0

    def is_trimmable(self) -> bool:
        return True

    def trim(self, n: int) -> int:
        n = min(self.offset, n)
        self.offset -= n
        return n

    def to_quantized(self, group_size: int = 64, bits: int = 4) -> BaseCache:
        return self  # the reference returns self too (reusable.py:250-254): KV stays 16-bit on this path
