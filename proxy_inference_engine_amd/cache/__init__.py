from .kv_cache import BaseCache, ReusableKVCache
from .prompt_cache import PromptCache

__all__ = ["BaseCache", "ReusableKVCache", "PromptCache"]
