"""PromptCache: per-layer KV caches + the token history they encode, with longest-common-prefix reuse
between consecutive requests (mirror of cache/prompt_cache.py:13-76 of the reference).

`computed_ids` is a host list that is resolved LAZILY: update() may be handed device tensors (the greedy tokens
the tail kernel wrote into the decoder's history buffer); they are read back in one copy the first time the
history is needed (the next request's prefix match, a logits processor) instead of one host sync per generated
token -- the role mx.async_eval plays in the reference's loop (engine/inference_engine.py:279,289).
Disk persistence (cache_prompt / load_cached_prompt, prompt_cache.py:78-125): one safetensors file per prompt hash in
the cache directory, written / read through BaseCache.save_cache / load_cache.  Like the reference, errors are logged,
never raised; unlike it the default directory is created when something is first saved, not at construction.
"""
from __future__ import annotations

import hashlib
import json
import logging
import pathlib

import torch

logger = logging.getLogger(__name__)

from .kv_cache import BaseCache, PagedKVCache, ReusableKVCache


def _as_list(ids) -> list[int]:
    if isinstance(ids, torch.Tensor):
        return [int(v) for v in ids.reshape(-1).tolist()]
    return [int(v) for v in ids]


class PromptCache:
    def __init__(self, directory=None, cache: list[BaseCache] | None = None, computed_ids=None):
        self.cache_directory = self._get_cache_directory(directory)
        self.cache: list[BaseCache] = cache or []
        self._ids: list[int] = _as_list(computed_ids) if computed_ids is not None else []
        self._pending: list[torch.Tensor] = []  # device tensors not yet read back

    @property
    def computed_ids(self) -> list[int]:
        if self._pending:
            flat = torch.cat([t.reshape(-1) for t in self._pending]) if len(self._pending) > 1 else self._pending[0].reshape(-1)
            self._ids.extend(int(v) for v in flat.tolist())  # one device->host copy for all pending tokens
            self._pending = []
        return self._ids

    @computed_ids.setter
    def computed_ids(self, ids) -> None:
        self._ids, self._pending = _as_list(ids), []

    def __call__(self, prompt_ids):
        return self.reuse_cache(prompt_ids)

    def create_kv_cache(self, model) -> None:
        if hasattr(model, "make_cache") and callable(model.make_cache):
            self.cache = model.make_cache()
            return
        assert hasattr(model, "layers") and isinstance(model.layers, list), "Model must have a layers attribute"
        self.cache = [ReusableKVCache() for _ in range(len(model.layers))]

    def update(self, prompt_ids) -> None:
        """Append processed ids (prompt_cache.py:43-50).  Device tensors are kept as they are until needed."""
        if isinstance(prompt_ids, torch.Tensor) and prompt_ids.is_cuda:
            self._pending.append(prompt_ids)
        else:
            self.computed_ids.extend(_as_list(prompt_ids))

    def reuse_cache(self, prompt_ids):
        """Returns the suffix of `prompt_ids` that still has to be processed; at least one token always is
        (the comparison stops at len(prompt)-1, prompt_cache.py:64-67)."""
        history = self.computed_ids
        if not self.cache or not history:
            return prompt_ids
        ids = _as_list(prompt_ids)
        common = 0
        for i, tok in enumerate(history):
            if i >= len(ids) - 1 or ids[i] != tok:
                break
            common += 1
        if common == 0:
            return prompt_ids
        for layer_cache in self.cache:
            assert isinstance(layer_cache, (ReusableKVCache, PagedKVCache))
            layer_cache.reuse(len(ids), common)
        # DEVIATION from prompt_cache.py:52-76, which leaves computed_ids untouched here: after a diverging request B the
        # reference's history reads A + B[k:] while the KV rows beyond k belong to B, so a later request that matches A beyond k
        # (chat pattern A -> B -> A) attends over B's rows as if they were A's.  The history is cut to what the caches now hold.
        del history[common:]
        return prompt_ids[common:]

    def cache_prompt(self) -> None:
        """Writes the KV caches and the token ids they encode to <dir>/<sha256(ids)>.safetensors (prompt_cache.py:78-100)."""
        try:
            self.cache_directory.mkdir(parents=True, exist_ok=True)
            ids = self.computed_ids
            path = self.cache_directory / f"{self._compute_prompt_hash(ids)}.safetensors"
            BaseCache.save_cache(str(path), self.cache, {"computed_ids": json.dumps(ids)})
            logger.debug("Cached system prompt to %s", path)
        except Exception as e:  # noqa: BLE001  (the reference logs and carries on, prompt_cache.py:99-100)
            logger.error("Failed to cache system prompt: %s", e)

    def load_cached_prompt(self, token_ids) -> None:
        """Restores caches + ids saved for exactly these token ids, if such a file exists (prompt_cache.py:102-125).
        NB (reference behaviour, kept): a loaded ReusableKVCache reports offset = its saved capacity
        (reusable.py:215-224); the reuse() call of the following request sets it to the matched prefix."""
        try:
            path = self.cache_directory / f"{self._compute_prompt_hash(_as_list(token_ids))}.safetensors"
            if not path.exists():
                logger.debug("No cache found for prompt %s", path.stem)
                return
            cache, metadata = BaseCache.load_cache(str(path))
            ids = json.loads(metadata["computed_ids"])
            assert isinstance(ids, list)
            self.computed_ids = ids
            self.cache = cache
        except Exception as e:  # noqa: BLE001
            logger.error("Failed to load cached system prompt: %s", e)

    @staticmethod
    def _get_cache_directory(directory=None) -> pathlib.Path:
        if isinstance(directory, str):
            directory = pathlib.Path(directory)
        if isinstance(directory, pathlib.Path):
            return directory
        return pathlib.Path(__file__).parent.absolute() / ".cache"  # prompt_cache.py:142-150

    @staticmethod
    def _compute_prompt_hash(token_ids) -> str:
        return hashlib.sha256(str(_as_list(token_ids)).encode()).hexdigest()  # prompt_cache.py:165-166
