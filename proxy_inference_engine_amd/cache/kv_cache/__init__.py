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

    # -- persistence (cache/kv_cache/__init__.py:163-210): one .safetensors file, arrays named by their position in the
    # nested [layer][keys|values] list ("3.0" = layer 3 keys), metadata flattened the same way:
    # "0.<i>" = meta_state of layer i, "1.<key>" = the caller's metadata, "2.<i>" = cache class name of layer i.
    @staticmethod
    def save_cache(file_name: str, cache: list["BaseCache"], metadata: dict[str, str] | None = None) -> None:
        from safetensors.torch import save_file
        arrays: dict[str, torch.Tensor] = {}
        meta: dict[str, str] = {}
        for i, c in enumerate(cache):
            for j, t in enumerate(c.state):
                if t is not None:
                    arrays[f"{i}.{j}"] = t.detach().to("cpu").contiguous()
            meta[f"0.{i}"] = str(c.meta_state)
            # a paged cache is stored as its gathered rows, the layout a ReusableKVCache restores from
            meta[f"2.{i}"] = "ReusableKVCache" if type(c).__name__ == "PagedKVCache" else type(c).__name__
        for k, v in (metadata or {}).items():
            meta[f"1.{k}"] = str(v)
        save_file(arrays, file_name, metadata=meta)

    @staticmethod
    def load_cache(file_name: str, device=None) -> tuple[list["BaseCache"], dict[str, str]]:
        from safetensors import safe_open
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        classes = {"ReusableKVCache": ReusableKVCache}
        with safe_open(file_name, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            n = sum(1 for k in meta if k.startswith("2."))
            cache: list[BaseCache] = []
            for i in range(n):
                name = meta[f"2.{i}"]
                if name not in classes:
                    raise ValueError(f"{file_name}: cache class {name} is not on the MI355X path")
                c = classes[name]()
                state = tuple(f.get_tensor(f"{i}.{j}").to(device) if f"{i}.{j}" in f.keys() else None for j in range(2))
                c.state = state
                c.meta_state = meta.get(f"0.{i}", "")
                cache.append(c)
        return cache, {k[2:]: v for k, v in meta.items() if k.startswith("1.")}

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
from .paged import PageAllocator, PagedKVCache, PagedSequence  # noqa: E402

__all__ = ["BaseCache", "ReusableKVCache", "PagedKVCache", "PagedSequence", "PageAllocator"]
