"""CPU checks of the oracle itself (no GPU): closed-form known answers, an independent
implementation (HF transformers on torch-CPU) and explicit dequantise-then-matmul.

The reference pins nothing for this path ("parity unpinned"), so these checks are what
stands between the oracle and a typo; they mirror SURVEY.md 8c "What pins the build's results instead".
"""
import numpy as np
import pytest
import torch

from oracle import pie_oracle as po

RNG = np.random.default_rng(1234)


@pytest.mark.parametrize("dt,tt", [("bfloat16", torch.bfloat16), ("float16", torch.float16)])
def test_rounding_matches_torch_casts(dt, tt):
    x = (RNG.standard_normal(200_000) * 10.0 ** RNG.integers(-9, 6, 200_000)).astype(np.float32)
    x[:10] = [0, -0.0, 1e-40, 65504, 65520, 1e5, 6e-8, 5.96e-8, 3e-8, -2.98e-8]
    ref = torch.from_numpy(x).to(tt)
    assert np.array_equal(po.to_bits(x, dt), ref.view(torch.int16).numpy().view(np.uint16))
    assert np.array_equal(po.round_T(x, dt).view(np.uint32), ref.float().numpy().view(np.uint32))
    assert np.array_equal(po.from_bits(po.to_bits(x, dt), dt), po.round_T(x, dt))


def test_quantize_known_answers():
    # one group whose extremes must land exactly on codes 0 and 15: w = bias + scale*q
    q = np.arange(64) % 16
    w = (-1.125 + 0.125 * q).astype(np.float32)[None, :]         # |w_min| > |w_max| -> side: edge=w_min
    wq, s, b = po.quantize(w, 64, 4, "float32")
    codes = (wq.reshape(-1)[:, None] >> (4 * np.arange(8))[None, :]) & 0xF   # little-endian nibble order
    assert np.array_equal(codes.reshape(-1), q)
    assert np.allclose(po.dequantize(wq, s, b, 64, 4, "float32"), w, atol=1e-7)
    # mirrored group: |w_max| > |w_min| -> scale negative, bias = w_max (scales can be negative)
    wq2, s2, b2 = po.quantize(-w, 64, 4, "float32")
    assert s2[0, 0] < 0 and b2[0, 0] == pytest.approx(1.125)
    assert np.allclose(po.dequantize(wq2, s2, b2, 64, 4, "float32"), -w, atol=1e-7)


@pytest.mark.parametrize("dt", ["float32", "bfloat16", "float16"])
def test_quantize_roundtrip_bound(dt):
    w = po.round_T(RNG.standard_normal((48, 256)) * 0.02, dt)
    wq, s, b = po.quantize(w, 64, 4, dt)
    wh = po.dequantize(wq, s, b, 64, 4, dt)
    scale = np.repeat(np.abs(po.from_bits(s, dt)), 64, axis=1)
    # interior codes err <= scale/2; the far end of the range can clip at code 15 because scale is re-fitted
    # to make the edge exact (scale = edge/q0, |edge/scale0| >= 7.5) -> <= 1.0*scale; plus storage rounding.
    slack = {"float32": 1e-6, "bfloat16": 2 ** -8, "float16": 2 ** -11}[dt]
    bound = 1.0 * scale + slack * (16 * scale + np.abs(w) + np.repeat(np.abs(po.from_bits(b, dt)), 64, axis=1))
    assert np.all(np.abs(w - wh) <= bound + 1e-9)


@pytest.mark.parametrize("dt", ["float32", "bfloat16"])
def test_quantized_matmul_vs_dequantize_then_matmul(dt):
    N, K, M = 96, 512, 3
    w = po.round_T(RNG.standard_normal((N, K)) * 0.05, dt)
    wq, s, b = po.quantize(w, 64, 4, dt)
    x = po.round_T(RNG.standard_normal((M, K)), dt)
    y = po.quantized_matmul(x, wq, s, b, dtype=dt)
    sf, bf = po.from_bits(s, dt).astype(np.float64), po.from_bits(b, dt).astype(np.float64)
    codes = ((wq[:, :, None] >> (4 * np.arange(8))) & 0xF).reshape(N, K).astype(np.float64)
    wd = np.repeat(sf, 64, 1) * codes + np.repeat(bf, 64, 1)      # un-rounded affine dequant, fp64
    ref = x.astype(np.float64) @ wd.T
    tol = 1e-4 if dt == "float32" else 2 ** -8 * np.abs(ref).max()
    assert np.max(np.abs(y - ref)) <= tol
    assert np.array_equal(y, po.round_T(y, dt))


def test_rms_norm_rope_sdpa_closed_forms():
    H = 128
    w = po.to_bits(np.full(H, 2.0, np.float32), "float32")
    y = po.rms_norm(np.full((1, H), 3.0, np.float32), w, 0.0, "float32")
    assert np.allclose(y, 2.0)                                     # 3/sqrt(9) * 2
    freqs = po.llama3_rope_freqs(64, 10000.0)
    assert np.allclose(freqs, 10000.0 ** (np.arange(0, 64, 2) / 64), rtol=1e-6)
    x = RNG.standard_normal((4, 1, 64)).astype(np.float32)
    assert np.array_equal(po.rope(x, freqs, 0, "float32"), x)      # position 0 = identity
    r = po.rope(x, freqs, 7, "float32")                            # rotation preserves pair norms
    assert np.allclose(r[..., :32] ** 2 + r[..., 32:] ** 2, x[..., :32] ** 2 + x[..., 32:] ** 2, rtol=1e-5)
    th = 7.0 / freqs[3]
    assert r[0, 0, 3] == pytest.approx(x[0, 0, 3] * np.cos(th) - x[0, 0, 35] * np.sin(th), rel=1e-5)
    k = np.zeros((2, 16, 64), np.float32)
    v = RNG.standard_normal((2, 16, 64)).astype(np.float32)
    q = RNG.standard_normal((4, 1, 64)).astype(np.float32)
    o = po.sdpa(q, k, v, 0.125, None, "float32", T=10)             # equal scores -> mean of the first T rows of V
    assert np.allclose(o[0, 0], v[0, :10].mean(0), atol=1e-6)
    assert np.allclose(o[3, 0], v[1, :10].mean(0), atol=1e-6)      # GQA: q-head 3 -> kv-head 1


def test_logprobs_argmax_first_max():
    x = np.array([0.5, 2.0, 2.0, -1.0], np.float32)
    tok, lp = po.logprobs_argmax(x)
    assert tok == 1
    assert np.exp(lp.astype(np.float64)).sum() == pytest.approx(1.0, rel=1e-6)


def _hf_model(cfg, weights_f32):
    from transformers import LlamaConfig, LlamaForCausalLM

    hc = LlamaConfig(hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                     num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                     num_key_value_heads=cfg["num_key_value_heads"], vocab_size=cfg["vocab_size"],
                     rms_norm_eps=cfg["rms_norm_eps"], rope_theta=cfg["rope_theta"],
                     max_position_embeddings=cfg["max_position_embeddings"], tie_word_embeddings=False,
                     attention_bias=False, mlp_bias=False)
    hc._attn_implementation = "eager"
    m = LlamaForCausalLM(hc).eval()
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights_f32.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in k for k in missing), (missing, unexpected)
    return m


def test_llama_graph_vs_hf_transformers_fp32():
    """Independent implementation check: same random fp32 weights, prefill 12 tokens then 3 decode steps."""
    cfg = dict(po.TINY_CONFIG)
    cfg.pop("quantization")
    w = po.synth_checkpoint(cfg, seed=3, dtype="float32")
    hf = _hf_model(cfg, w)
    orc = po.OracleLlama(cfg, w, "float32")
    cache = [po.OracleKVCache() for _ in orc.layers]
    ids = RNG.integers(0, cfg["vocab_size"], 15)
    with torch.no_grad():
        ref_all = hf(torch.from_numpy(ids[None, :]).long()).logits[0].numpy()
    got = orc.forward(ids[:12], cache)
    assert np.max(np.abs(got - ref_all[:12])) <= 2e-5 * np.abs(ref_all).max() + 1e-5
    for t in range(12, 15):                                        # L=1 path: no mask, offset = t
        step = orc.forward(ids[t:t + 1], cache)
        assert np.max(np.abs(step[0] - ref_all[t])) <= 2e-5 * np.abs(ref_all).max() + 1e-5
    assert cache[0].offset == 15 and cache[0].keys.shape[2] == 256


def test_llama_graph_bf16_close_to_fp32():
    cfg = dict(po.TINY_CONFIG)
    cfg.pop("quantization")
    w32 = po.synth_checkpoint(cfg, seed=5, dtype="float32")
    wbf = {k: po.to_bits(v, "bfloat16") for k, v in w32.items()}
    w32r = {k: po.from_bits(v, "bfloat16") for k, v in wbf.items()}
    ids = RNG.integers(0, cfg["vocab_size"], 10)
    a = po.OracleLlama(cfg, w32r, "float32").forward(ids, [po.OracleKVCache() for _ in range(2)])
    b = po.OracleLlama(cfg, wbf, "bfloat16").forward(ids, [po.OracleKVCache() for _ in range(2)])
    assert np.max(np.abs(a - b)) <= 2e-2                           # bf16 activations, logits max-abs
    u = po.OracleLlama(cfg, wbf, "bfloat16").forward(ids, [po.OracleKVCache() for _ in range(2)], sdpa_fused=False)
    assert np.max(np.abs(u - b)) <= 2e-2                           # fused vs unfused-fallback attention contracts


def test_kv_cache_growth_and_prefix_reuse():
    c = po.OracleKVCache()
    k = np.ones((1, 2, 128, 8), np.float32)
    c.update_and_fetch(k, k)
    assert (c.offset, c.keys.shape[2]) == (128, 256)               # ceil(128/256)*256
    one = np.ones((1, 2, 1, 8), np.float32)
    caps = []
    for _ in range(1100):
        c.update_and_fetch(one * 2, one * 2)
        caps.append(c.keys.shape[2])
    assert sorted(set(caps)) == [256, 512, 768, 1280]              # max(int(cap*1.5), need) rounded up to 256
    assert c.offset == 1228 and np.all(c.keys[0, 0, :128] == 1) and np.all(c.keys[0, 0, 128:1228] == 2)
    assert c.trim(28) == 28 and c.offset == 1200
    pc = po.OraclePromptCache()
    pc.cache = [po.OracleKVCache()]
    pc.cache[0].update_and_fetch(k, k)
    pc.update(np.arange(128))
    rest = pc(np.arange(128))                                       # identical prompt: >= 1 token re-processed
    assert rest.tolist() == [127] and pc.cache[0].offset == 127
    rest = pc(np.concatenate([np.arange(50), np.arange(900, 1200)]))
    assert len(rest) == 300 and pc.cache[0].offset == 50 and pc.cache[0].keys.shape[2] == 512
    assert pc(np.array([999, 1, 2])).tolist() == [999, 1, 2]       # no common prefix: cache untouched


def test_qmm_regime_is_dequantize_then_matmul_and_close_to_qmv():
    """MLX's two quantized_matmul regimes (oracle/pie_oracle.c: lin): the qmm form equals mx.dequantize followed by a dense
    T x T -> fp32 matmul, and stays within a fraction of an output ulp (rms) of the exact qmv form."""
    rng = np.random.default_rng(5)
    N, K, M = 96, 256, 20
    wq, sc, bi = po.quantize(po.round_T(rng.standard_normal((N, K)) * 0.05, "bfloat16"), dtype="bfloat16")
    x = po.round_T(rng.standard_normal((M, K)), "bfloat16")
    exact = po.quantized_matmul(x, wq, sc, bi, dtype="bfloat16")
    dense = po.linear(x, po.to_bits(po.dequantize(wq, sc, bi, dtype="bfloat16"), "bfloat16"), "bfloat16")
    assert np.array_equal(po.quantized_matmul(x, wq, sc, bi, dtype="bfloat16", regime="qmm"), dense)
    cfg = dict(po.TINY_CONFIG)
    w = po.synth_checkpoint(cfg, seed=2, dtype="bfloat16")
    orc = po.OracleLlama(cfg, w, "bfloat16")
    ids = rng.integers(0, cfg["vocab_size"], 20)
    try:
        po.set_qmm_min_rows(0)
        a = orc.forward(ids, [po.OracleKVCache() for _ in orc.layers])
        po.set_qmm_min_rows(6)
        b = orc.forward(ids, [po.OracleKVCache() for _ in orc.layers])
        c = orc.forward(ids[:5], [po.OracleKVCache() for _ in orc.layers])    # below the threshold: the exact form
    finally:
        po.set_qmm_min_rows(6)
    assert po.get_qmm_min_rows() == 6
    assert np.array_equal(c, a[:5])                                   # causal model: first 15 rows do not see the rest
    assert not np.array_equal(a, b)                                    # the regimes differ ...
    err = np.abs(a - b)
    assert err.max() <= 4 * 2.0 ** -8 * np.abs(a).max()                # ... inside the end-to-end parity bound
    d = np.abs(exact - dense)
    assert d.max() <= 2.0 ** -7 * np.abs(exact).max() and (d > 0).mean() < 0.6


def test_vision_oracle_matches_hf_qwen2_5_vl_tower_fp32():
    """oracle/vision_oracle.py against an independent implementation of the same tower -- HF transformers'
    Qwen2_5_VisionTransformerPretrainedModel on torch-CPU with identical weights, fp32: patch embedding, 2-D rotary, window
    order and cu_seqlens, windowed and full-attention blocks, SwiGLU MLP, patch merger and the inverse permutation."""
    torch = pytest.importorskip("torch")
    modeling = pytest.importorskip("transformers.models.qwen2_5_vl.modeling_qwen2_5_vl")
    configuration = pytest.importorskip("transformers.models.qwen2_5_vl.configuration_qwen2_5_vl")
    from oracle import vision_oracle as vo
    cfg = dict(depth=3, hidden_size=64, intermediate_size=48, out_hidden_size=40, num_heads=2, patch_size=14, in_channels=3,
               spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1])
    hc = configuration.Qwen2_5_VLVisionConfig(**cfg, hidden_act="silu")
    hc._attn_implementation = "eager"
    torch.manual_seed(0)
    hf = modeling.Qwen2_5_VisionTransformerPretrainedModel(hc).float().eval()
    w = {"vision_tower." + k: v.detach().numpy().astype(np.float32) for k, v in hf.state_dict().items()}
    for grid in ([(1, 6, 10), (2, 4, 4)], [(1, 2, 2)], [(1, 8, 8)]):
        n = sum(t * h * ww for t, h, ww in grid)
        pix = np.random.default_rng(n).standard_normal((n, 3 * 2 * 14 * 14)).astype(np.float32)
        with torch.no_grad():
            out = hf(torch.from_numpy(pix), grid_thw=torch.tensor(grid))
        ref = out if isinstance(out, torch.Tensor) else (out.pooler_output if getattr(out, "pooler_output", None) is not None else out[0])
        got = vo.vision_forward(cfg, w, pix, grid, "float32")
        assert got.shape == tuple(ref.shape)
        assert np.max(np.abs(got - ref.numpy())) <= 2e-5 * max(1.0, float(ref.abs().max())), grid


def test_oracle_variants_vs_hf_transformers_fp32():
    """The configuration switches of the reference's Llama (language.py:19-53) against independent implementations, fp32:
    attention / MLP biases and llama3 rope scaling against HF LlamaForCausalLM, the Qwen2-VL text tower's form (biases on q, k, v
    only, theta 1e6) against HF Qwen2ForCausalLM."""
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM
    base = dict(po.TINY_CONFIG)
    base.pop("quantization")
    ids = RNG.integers(0, base["vocab_size"], 14)

    def check(cfg, hf, w):
        missing, unexpected = hf.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
        assert not unexpected and all("rotary" in k for k in missing), (missing, unexpected)
        with torch.no_grad():
            ref = hf(torch.from_numpy(ids[None, :]).long()).logits[0].numpy()
        orc = po.OracleLlama(cfg, w, "float32")
        cache = [po.OracleKVCache() for _ in orc.layers]
        got = np.concatenate([orc.forward(ids[:11], cache)] + [orc.forward(ids[t:t + 1], cache) for t in range(11, 14)])
        assert np.max(np.abs(got - ref)) <= 2e-5 * np.abs(ref).max() + 1e-5

    # Llama with every Linear biased and llama3 rope scaling
    cfg = dict(base, attention_bias=True, mlp_bias=True, max_position_embeddings=8192,
               rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192})   # the reference takes
    # max_position_embeddings for both lengths of Llama3RoPE (language.py:57-66), i.e. HF's original_max_position_embeddings
    hc = LlamaConfig(hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
                     num_attention_heads=cfg["num_attention_heads"], num_key_value_heads=cfg["num_key_value_heads"], vocab_size=cfg["vocab_size"],
                     rms_norm_eps=cfg["rms_norm_eps"], rope_theta=cfg["rope_theta"], max_position_embeddings=8192, tie_word_embeddings=False,
                     attention_bias=True, mlp_bias=True, rope_scaling=dict(cfg["rope_scaling"]))
    hc._attn_implementation = "eager"
    check(cfg, LlamaForCausalLM(hc).eval(), po.synth_checkpoint(cfg, seed=8, dtype="float32"))
    # Qwen2: q / k / v biases only
    cfg = dict(base, attention_bias=True, rope_theta=1000000.0)
    w = po.synth_checkpoint(cfg, seed=9, dtype="float32")
    for k in [k for k in w if k.endswith("o_proj.bias")]:
        del w[k]
    qc = Qwen2Config(hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
                     num_attention_heads=cfg["num_attention_heads"], num_key_value_heads=cfg["num_key_value_heads"], vocab_size=cfg["vocab_size"],
                     rms_norm_eps=cfg["rms_norm_eps"], rope_theta=1000000.0, max_position_embeddings=cfg["max_position_embeddings"],
                     tie_word_embeddings=False, use_sliding_window=False)
    qc._attn_implementation = "eager"
    check(cfg, Qwen2ForCausalLM(qc).eval(), w)


def test_byte_packed_3_and_6_bit_codes_known_answers():
    """MLX packs 3- and 6-bit codes as a little-endian bit stream over bytes (mlx >= 0.21: `bits == 3 || bits == 6`).  Known answers written
    out from its published unpacking formulas -- 3 bits: w0 & 7, (w0 >> 3) & 7, (w0 >> 6) | ((w1 & 1) << 2), (w1 >> 1) & 7, (w1 >> 4) & 7,
    (w1 >> 7) | ((w2 & 3) << 1), (w2 >> 2) & 7, w2 >> 5; 6 bits: w0 & 63, (w0 >> 6) | ((w1 & 15) << 2), (w1 >> 4) | ((w2 & 3) << 4), w2 >> 2."""
    # codes 0..7 (3 bits) -> bytes 136, 198, 250; codes 1, 2, 3, 60 (6 bits) -> bytes 129, 48, 240
    ones, zeros = po.to_bits(np.ones((1, 1), np.float32), "float32"), po.to_bits(np.zeros((1, 1), np.float32), "float32")
    row3 = np.zeros((1, 6), np.uint32)                     # 64 codes * 3 bits = 6 words
    row3.view(np.uint8)[0, :3] = (136, 198, 250)
    got = po.dequantize(row3, ones, zeros, group_size=64, bits=3, dtype="float32")
    assert got[0, :8].tolist() == [0, 1, 2, 3, 4, 5, 6, 7] and not got[0, 8:].any()
    row6 = np.zeros((1, 12), np.uint32)
    row6.view(np.uint8)[0, :3] = (129, 48, 240)
    got = po.dequantize(row6, ones, zeros, group_size=64, bits=6, dtype="float32")
    assert got[0, :4].tolist() == [1, 2, 3, 60] and not got[0, 4:].any()


@pytest.mark.parametrize("bits", [2, 3, 4, 6, 8])
def test_quantize_round_trip_every_code_width(bits):
    """mx.quantize / mx.dequantize at every width nn.quantize accepts: |w - w_hat| < one bin per group (the edge that defines the scale is exact, the far
    end may clip), codes use the full range."""
    rng = np.random.default_rng(bits)
    w = rng.standard_normal((8, 256)).astype(np.float32)
    wq, s, b = po.quantize(w, 64, bits, "float32")
    assert wq.shape == (8, 256 * bits // 32)
    back = po.dequantize(wq, s, b, 64, bits, "float32")
    scale = np.abs(po.from_bits(s, "float32")).repeat(64, axis=1)
    assert np.all(np.abs(back - w) <= scale * (1 + 1e-5) + 1e-6)
    codes = np.rint((back - po.from_bits(b, "float32").repeat(64, axis=1)) / po.from_bits(s, "float32").repeat(64, axis=1))
    assert codes.min() == 0 and codes.max() == (1 << bits) - 1


@pytest.mark.parametrize("bits,to_bits", [(2, 4), (3, 4), (6, 8)])
def test_loader_repacks_narrow_codes_without_touching_a_value(bits, to_bits):
    """Model.__init__ stores 2- / 3-bit codes as 4-bit codes and 6-bit codes as bytes (models/llama/language.py: _recode_mlx_codes): the re-packed
    words, read at the wider width, must dequantise to exactly what the checkpoint's own words give."""
    import torch
    from proxy_inference_engine_amd.models.llama.language import _recode_mlx_codes
    rng = np.random.default_rng(10 + bits)
    w = rng.standard_normal((5, 192)).astype(np.float32)
    wq, s, b = po.quantize(w, 64, bits, "bfloat16")
    wide = _recode_mlx_codes(torch.from_numpy(wq.view(np.int32)), bits, to_bits).numpy().view(np.uint32)
    assert wide.shape == (5, 192 * to_bits // 32)
    assert np.array_equal(po.dequantize(wide, s, b, 64, to_bits, "bfloat16"), po.dequantize(wq, s, b, 64, bits, "bfloat16"))
