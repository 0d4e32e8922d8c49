"""KV-cache protocol of the decode path (host-side mirror of the reference's
cache/kv_cache/__init__.py:10-161; buffers are torch ROCm tensors instead of mx.array)."""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch


class BaseCache(ABC):
    """Interface every per-layer cache implements (cache/kv_cache/__init__.py:10-161)."""

    offset: int
    step: int

    @staticmethod
    def make_kv_cache(model, max_kv_size: int | None = None, reusable: bool = True) -> list["BaseCache"]:
        if hasattr(model, "make_cache") and model.make_cache is not None:
            return model.make_cache()
        if max_kv_size is not None or not reusable:
            raise NotImplementedError("only ReusableKVCache is on the engine path (prompt_cache.py:34-41, :73)")
        return [ReusableKVCache() for _ in range(len(model.layers))]

    @property
    def state(self):
        return []

    @state.setter
    def state(self, v):
        if v is not None and v:
            raise ValueError("This cache has no state but a state was set.")

    @property
    def meta_state(self):
        return ""

    @meta_state.setter
    def meta_state(self, v):
        if v is not None and v:
            raise ValueError("This cache has no meta_state but a meta_state was set.")

    def is_trimmable(self) -> bool:
        return False

    @abstractmethod
    def trim(self, n: int) -> int: ...

    @abstractmethod
    def update_and_fetch(self, keys: torch.Tensor, values: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]: ...

    @abstractmethod
    def to_quantized(self, group_size: int = 64, bits: int = 4) -> "BaseCache": ...


from .reusable import ReusableKVCache  # noqa: E402

__all__ = ["BaseCache", "ReusableKVCache"]
