"""-m gpu: the persistent one-launch decode step (csrc/step_engine.hip, pie_decoder_configure(PIE_OPT_ENGINE)) and the per-q-head
attention plan it shares with the launch sequence (csrc/attn_head.hpp).

Two bars.  (1) Against the CPU oracle, with the tolerances of tests/test_gpu_decode.py: the persistent step is a product path like
any other.  (2) Against the launch sequence under the same attention plan: logits, greedy tokens and the hidden state bit for bit
(every fp32 operation is the same and in the same order: RMSNorm tree, unit dot products, RoPE, attention streams, epilogue
roundings), log-probabilities to fp32 rounding (their log-sum-exp is summed over each path's own wave partition of the vocabulary).
A stale hand-off granule would show up as a bit difference, which is why the second bar is exact.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pie_oracle as po
from tests._util import assert_vec_close, to_bits
from tests.test_gpu_decode import build, margin_bound

pytestmark = pytest.mark.gpu


def _cfg(H, I, heads, kv, V, layers=2, theta=500000.0):
    return {"model_type": "llama", "hidden_size": H, "num_hidden_layers": layers, "intermediate_size": I, "num_attention_heads": heads,
            "num_key_value_heads": kv, "rms_norm_eps": 1e-5, "vocab_size": V, "rope_theta": theta, "max_position_embeddings": 8192,
            "tie_word_embeddings": False, "quantization": {"group_size": 64, "bits": 4}}


GEOMETRIES = {
    # name: (config, dtype) -- the engine kernel's four instantiations (head_dim 128 / 64 x one / two K slices of the hidden size)
    "8b-shaped": (_cfg(4096, 14336, 32, 8, 8192), "bfloat16"),           # D 128, K slices 2 (H) and 7 (I): the headline geometry
    "8b-shaped-f16": (_cfg(4096, 14336, 32, 8, 8192), "float16"),
    "h2048-d128": (_cfg(2048, 5632, 16, 4, 4096), "bfloat16"),            # one K slice; I = 2.75 slices (ragged last unit of a row)
    "tinyllama-shaped": (_cfg(2048, 5632, 32, 4, 32000, theta=10000.0), "float16"),  # D 64, 8 q-heads per kv-head
    "h3072-d128": (_cfg(3072, 8192, 24, 8, 4096, layers=3), "bfloat16"),  # 1.5 K slices: zero-padded lanes in every row's last unit; odd layer count
}


def _configure(model, engine, heads):
    from proxy_inference_engine_amd import _ffi
    lib = _ffi.load()
    _ffi.check(lib.pie_decoder_configure(model._dec, _ffi.PIE_OPT_ENGINE, engine))
    _ffi.check(lib.pie_decoder_configure(model._dec, _ffi.PIE_OPT_ATTN_HEADS, heads))
    return lib


def _status(model):
    from proxy_inference_engine_amd import _ffi
    err = C.c_uint(0)
    _ffi.check(_ffi.load().pie_decoder_status(model._dec, C.byref(err)))
    return err.value


@pytest.mark.parametrize("name", list(GEOMETRIES))
def test_persistent_step_vs_oracle_and_launch_sequence(name):
    from proxy_inference_engine_amd import _ffi
    cfg, dtype = GEOMETRIES[name]
    w = po.synth_checkpoint(cfg, seed=11, dtype=dtype, lm_head_gain=4.0)
    model = build(cfg, w, dtype)
    orc = po.OracleLlama(cfg, w, dtype)
    prompt = np.random.default_rng(5).integers(0, cfg["vocab_size"], 5)  # below the batched-prefill threshold (6 rows): four steps without logits + one with
    n_steps = 5

    def run(engine, graph):
        lib = _configure(model, engine, 1)
        cache = model.make_cache()
        tok, lp, logits = model.step(torch.from_numpy(prompt).cuda(), cache)
        # the capacity exists only after the first step: that is when the plan is chosen
        assert lib.pie_decoder_query(model._dec, _ffi.PIE_QUERY_ATTN_HEADS) == 1
        assert lib.pie_decoder_query(model._dec, _ffi.PIE_QUERY_ENGINE) == engine, "the persistent launch is not what ran"
        out = [(int(tok.item()), to_bits(logits).copy(), lp.float().cpu().numpy().copy(), to_bits(model.hidden).copy())]
        for _ in range(n_steps):
            tok, lp, logits = model.step(None, cache, graph=graph)
            out.append((int(tok.item()), to_bits(logits).copy(), lp.float().cpu().numpy().copy(), to_bits(model.hidden).copy()))
        assert _status(model) == 0, "a bounded wait of the persistent launch gave up"
        return out

    eng = run(1, True)
    # (1) the oracle, teacher-forced with the engine's own tokens
    ocache = [po.OracleKVCache() for _ in orc.layers]
    want = orc.forward(prompt, ocache)[-1]
    matched = 0
    for i, (tok, lbits, lp, _) in enumerate(eng):
        got = po.from_bits(lbits, dtype)
        assert_vec_close(got, want, dtype, what=f"{name}: logits of step {i}")
        otok, olp = po.logprobs_argmax(want)
        top2 = np.sort(olp)[-2:]
        if top2[1] - top2[0] > margin_bound(want, dtype):
            assert tok == otok, (name, i)
            matched += 1
        assert np.abs(lp - olp).max() <= 4 * margin_bound(want, dtype), (name, i)
        want = orc.forward(np.array([tok]), ocache)[0]
    assert matched >= 2, "the synthetic checkpoint's margins are too small to test a single greedy id"
    # (2) the launch sequence under the same attention plan, eager and graph-replayed; and the engine launched eagerly
    for key, other in {"launches, graph": run(0, True), "launches, eager": run(0, False), "engine, eager": run(1, False)}.items():
        for i, (a, b) in enumerate(zip(eng, other)):
            assert a[0] == b[0], (name, key, i)
            assert np.array_equal(a[1], b[1]), f"{name}: logits of step {i} differ from {key}"
            assert np.array_equal(a[3], b[3]), f"{name}: hidden state of step {i} differs from {key}"
            assert np.abs(a[2] - b[2]).max() <= 1e-5, (name, key, i)
    _configure(model, 0, -1)


def test_head_plan_attention_vs_oracle_at_longer_context():
    """The per-q-head attention plan through the launch sequence and the persistent step with a few hundred cached positions
    (more row blocks than a wave keeps in registers: the streaming part of attn_head_score), across a cache re-allocation
    (256 -> 512 positions) and up to the capacity where the plan hands over to split-KV attention."""
    from proxy_inference_engine_amd import _ffi
    cfg = _cfg(2048, 5632, 16, 4, 4096)
    dtype = "bfloat16"
    w = po.synth_checkpoint(cfg, seed=3, dtype=dtype, lm_head_gain=4.0)
    model = build(cfg, w, dtype)
    orc = po.OracleLlama(cfg, w, dtype)
    rng = np.random.default_rng(9)
    prompt = rng.integers(0, cfg["vocab_size"], 250)
    forced = rng.integers(0, cfg["vocab_size"], 12)
    for engine in (0, 1):
        lib = _configure(model, engine, -1 if engine else 1)
        ocache = [po.OracleKVCache() for _ in orc.layers]
        orc.forward(prompt, ocache)
        cache = model.make_cache()
        model.step(torch.from_numpy(prompt).cuda(), cache)
        for i, t in enumerate(forced):
            want = orc.forward(np.array([t]), ocache)[0]
            _, _, logits = model.step(torch.tensor([int(t)], dtype=torch.int32, device="cuda"), cache)
            assert lib.pie_decoder_query(model._dec, _ffi.PIE_QUERY_ENGINE) == engine
            assert_vec_close(logits.float().cpu().numpy(), want, dtype, what=f"engine {engine}, step {i}, offset {cache[0].offset}")
        assert cache[0].capacity == 512 and cache[0].offset == 262
        assert _status(model) == 0
    # beyond ENGINE_MAX_CAP positions of capacity the automatic plan is split-KV again and the persistent launch steps aside
    lib = _configure(model, 1, -1)
    cache = model.make_cache()
    model.step(torch.from_numpy(rng.integers(0, cfg["vocab_size"], 520)).cuda(), cache)
    model.step(None, cache)
    assert cache[0].capacity > 512
    assert lib.pie_decoder_query(model._dec, _ffi.PIE_QUERY_ENGINE) == 0 and lib.pie_decoder_query(model._dec, _ffi.PIE_QUERY_ATTN_HEADS) == 0
    _configure(model, 0, -1)
