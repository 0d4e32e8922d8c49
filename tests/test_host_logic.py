"""CPU: host-side mirrors of the reference's cache / prompt-cache / sampler logic behave like the oracle's
restatement (and therefore like cache/kv_cache/reusable.py, cache/prompt_cache.py, samplers/__init__.py)."""
import numpy as np
import pytest
import torch

from oracle import pie_oracle as po
from proxy_inference_engine_amd.cache import PromptCache, ReusableKVCache
from proxy_inference_engine_amd.logits_processors import make_repetition_penalty
from proxy_inference_engine_amd.models.base import create_attention_mask, create_causal_mask
from proxy_inference_engine_amd.models.llama import ModelArgs
from proxy_inference_engine_amd.samplers import make_sampler


def test_reusable_cache_capacity_sequence_matches_oracle():
    ref, mine = po.OracleKVCache(), ReusableKVCache()
    k = np.ones((1, 2, 128, 8), np.float32)
    ref.update_and_fetch(k, k)
    mine.update_and_fetch(torch.ones(1, 2, 128, 8), torch.ones(1, 2, 128, 8))
    one = torch.full((1, 2, 1, 8), 2.0)
    for _ in range(1100):
        ref.update_and_fetch(one.numpy(), one.numpy())
        kk, vv = mine.update_and_fetch(one, one)
        assert (mine.offset, mine.capacity) == (ref.offset, ref.keys.shape[2])
    assert kk.shape[2] == mine.offset and np.array_equal(mine.keys.numpy(), ref.keys)
    assert mine.trim(28) == 28 and mine.offset == 1200 and mine.is_trimmable() and mine.to_quantized() is mine
    # reserve()/advance() = the two halves of update_and_fetch used by the fused decoder
    c = ReusableKVCache()
    c.reserve(300, 2, 8, torch.float32, "cpu")
    assert c.capacity == 512 and c.offset == 0
    c.advance(300)
    c.reserve(300, 2, 8, torch.float32, "cpu")
    assert c.capacity == 768
    capped = ReusableKVCache(max_capacity=256)
    capped.reserve(200, 1, 8, torch.float32, "cpu")
    capped.advance(200)
    with pytest.raises(RuntimeError):
        capped.reserve(100, 1, 8, torch.float32, "cpu")


def test_prompt_cache_lcp_matches_oracle():
    ref, mine = po.OraclePromptCache(), PromptCache()
    ref.cache, mine.cache = [po.OracleKVCache()], [ReusableKVCache()]
    k = np.ones((1, 2, 128, 8), np.float32)
    ref.cache[0].update_and_fetch(k, k)
    mine.cache[0].update_and_fetch(torch.ones(1, 2, 128, 8), torch.ones(1, 2, 128, 8))
    ref.update(np.arange(128)); mine.update(torch.arange(128))
    for prompt in (np.arange(128), np.concatenate([np.arange(50), np.arange(900, 1200)]), np.array([999, 1, 2]), np.arange(3)):
        a, b = ref(prompt), mine(torch.from_numpy(prompt))
        assert list(a) == list(b.tolist())
        assert (mine.cache[0].offset, mine.cache[0].capacity) == (ref.cache[0].offset, ref.cache[0].keys.shape[2])
    assert PromptCache()(torch.arange(5)).tolist() == [0, 1, 2, 3, 4]     # nothing cached yet


def test_sampler_and_processor_conventions():
    with pytest.raises(NotImplementedError):
        make_sampler(temp=0.7, top_p=0.9)
    assert getattr(make_sampler(temp=0), "is_greedy", False)
    with pytest.raises(ValueError):
        make_repetition_penalty(-1.0)
    proc = make_repetition_penalty(2.0, context_size=3)
    logits = torch.tensor([[1.0, -1.0, 4.0, -4.0, 8.0]])
    out = proc([4, 0, 1, 2, 3], logits.clone())                         # only the last 3 tokens count
    assert out.tolist() == [[1.0, -2.0, 2.0, -8.0, 8.0]]


def test_masks_and_model_args():
    m = create_causal_mask(3, offset=2)
    assert m.shape == (3, 5) and m[0].tolist() == [0, 0, 0, -1e9, -1e9] and m[2].tolist() == [0, 0, 0, 0, 0]
    assert np.array_equal(m.numpy().astype(np.float32), po.causal_mask(3, 2, "float32"))
    assert create_attention_mask(torch.zeros(1, 1, 8)) is None          # L == 1: no mask (models/base.py:39-53)
    a = ModelArgs(model_type="llama", hidden_size=64, num_hidden_layers=1, intermediate_size=128, num_attention_heads=2,
                  rms_norm_eps=1e-5, vocab_size=10, some_unknown_key=1)
    assert a.tie_word_embeddings is True and a.rope_theta == 10000 and not hasattr(a, "some_unknown_key")
