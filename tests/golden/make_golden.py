"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference cannot run here (no MLX, Python 3.10 < 3.12) and holds no fixtures for this path, so these
vectors are ORACLE-CAPTURED (labelled `source="oracle"`), not reference-captured.  They pin the oracle
against drift (tests/test_golden.py, CPU) and give the GPU tests fixed inputs/outputs (-m gpu).
Inputs are stored explicitly (not re-derived from a seed) so the fixtures stay valid if the generator changes.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import pie_oracle as po  # noqa: E402

OUT = Path(__file__).resolve().parent
DT = "bfloat16"


def ops_fixture():
    rng = np.random.default_rng(20250509)
    d = {"source": "oracle", "dtype": DT}
    # K1: one Llama-3-8B-shaped GEMV slice: 48 rows of gate_proj (K=4096) and 16 rows of down_proj (K=14336)
    for tag, N, K in (("gemv4096", 48, 4096), ("gemv14336", 16, 14336)):
        w = po.round_T(rng.standard_normal((N, K)) * 0.02, DT)
        wq, s, b = po.quantize(w, 64, 4, DT)
        x = po.round_T(rng.standard_normal((1, K)), DT)
        d[f"{tag}_wq"], d[f"{tag}_scales"], d[f"{tag}_biases"] = wq, s, b
        d[f"{tag}_x"] = po.to_bits(x, DT)
        d[f"{tag}_y"] = po.to_bits(po.quantized_matmul(x, wq, s, b, dtype=DT), DT)
    # K3 rms_norm
    x = po.round_T(rng.standard_normal((2, 4096)) * 3, DT)
    w = po.to_bits(1 + 0.02 * rng.standard_normal(4096), DT)
    d["rms_x"], d["rms_w"], d["rms_eps"] = po.to_bits(x, DT), w, np.float32(1e-5)
    d["rms_y"] = po.to_bits(po.rms_norm(x, w, 1e-5, DT), DT)
    # K4 rope, Llama-3 theta, position 1234
    freqs = po.llama3_rope_freqs(128, 500000.0)
    x = po.round_T(rng.standard_normal((8, 1, 128)), DT)
    d["rope_freqs"], d["rope_x"], d["rope_offset"] = freqs, po.to_bits(x, DT), np.int32(1234)
    d["rope_y"] = po.to_bits(po.rope(x, freqs, 1234, DT), DT)
    # K2 sdpa decode: 8 q-heads / 2 kv-heads, T=333 of cap 512, D=128
    q = po.round_T(rng.standard_normal((8, 1, 128)), DT)
    k = po.round_T(rng.standard_normal((2, 512, 128)), DT)
    v = po.round_T(rng.standard_normal((2, 512, 128)), DT)
    d["sdpa_q"], d["sdpa_k"], d["sdpa_v"] = po.to_bits(q, DT), po.to_bits(k, DT), po.to_bits(v, DT)
    d["sdpa_T"], d["sdpa_scale"] = np.int32(333), np.float32(128 ** -0.5)
    d["sdpa_out"] = po.to_bits(po.sdpa(q, k, v, 128 ** -0.5, None, DT, True, T=333), DT)
    # K6 silu*mul, K7 add
    a = po.round_T(rng.standard_normal(14336) * 2, DT)
    b = po.round_T(rng.standard_normal(14336), DT)
    d["act_a"], d["act_b"] = po.to_bits(a, DT), po.to_bits(b, DT)
    d["act_silu_mul"] = po.to_bits(po.silu_mul(a, b, DT), DT)
    d["act_add"] = po.to_bits(po.add(a, b, DT), DT)
    # K9 logits tail
    logits = po.round_T(rng.standard_normal(128256) * 2, DT)
    tok, lp = po.logprobs_argmax(logits)
    d["tail_logits"], d["tail_token"], d["tail_logprobs"] = po.to_bits(logits, DT), np.int32(tok), lp
    np.savez_compressed(OUT / "ops_bf16.npz", **d)


def tiny_llama_fixture():
    """Tiny Llama (H=256, 2 layers, 4/2 heads, D=64, V=512), int4 g=64, bf16: 24-token prompt, 16 greedy tokens."""
    cfg = po.TINY_CONFIG
    w = po.synth_checkpoint(cfg, seed=7, dtype=DT, lm_head_gain=8.0)
    model = po.OracleLlama(cfg, w, DT)
    prompt = np.random.default_rng(11).integers(0, cfg["vocab_size"], 24)
    pc = po.OraclePromptCache()
    gen = po.generate_step(model, pc, prompt)
    toks, lps, margins = [], [], []
    for _ in range(16):
        t, lp = next(gen)
        toks.append(t)
        lps.append(lp.copy())
        top2 = np.sort(lp)[-2:]
        margins.append(top2[1] - top2[0])
    # layer-wise check vector: residual stream after the last block for the prompt's last position
    cache2 = [po.OracleKVCache() for _ in model.layers]
    logits, hidden = model.forward(prompt, cache2, want_hidden=True)
    d = {"source": "oracle", "dtype": DT, "config_json": np.array(__import__("json").dumps(cfg)),
         "prompt": prompt.astype(np.int32), "tokens": np.array(toks, np.int32), "logprobs": np.stack(lps),
         "margins": np.array(margins, np.float32), "prefill_last_logits": po.to_bits(logits[-1], DT),
         "prefill_hidden_last": po.to_bits(hidden[-1], DT)}
    for k_, v_ in w.items():
        d["w:" + k_] = v_
    np.savez_compressed(OUT / "tiny_llama_w4_bf16.npz", **d)
    print("tiny llama: tokens", toks, "min margin", float(min(margins)))


def variant_fixture(name, overrides, dtype, seed):
    """One-layer tiny Llama variants for the other checkpoint formats / ModelArgs flags: 24-token prompt (batched prefill
    regime), 8 greedy tokens with their log-probabilities."""
    cfg = dict(po.TINY_CONFIG, num_hidden_layers=1, vocab_size=256, **overrides)
    if cfg.get("quantization") is None:
        cfg.pop("quantization", None)
    w = po.synth_checkpoint(cfg, seed=seed, dtype=dtype, lm_head_gain=8.0)
    model = po.OracleLlama(cfg, w, dtype)
    prompt = np.random.default_rng(seed + 1).integers(0, cfg["vocab_size"], 24)
    gen = po.generate_step(model, po.OraclePromptCache(), prompt)
    toks, lps, margins = [], [], []
    for _ in range(8):
        t, lp = next(gen)
        toks.append(t), lps.append(lp.copy())
        top2 = np.sort(lp)[-2:]
        margins.append(top2[1] - top2[0])
    d = {"source": "oracle", "dtype": dtype, "config_json": np.array(__import__("json").dumps(cfg)), "prompt": prompt.astype(np.int32),
         "tokens": np.array(toks, np.int32), "logprobs": np.stack(lps), "margins": np.array(margins, np.float32)}
    for k_, v_ in w.items():
        d["w:" + k_] = v_
    np.savez_compressed(OUT / f"{name}.npz", **d)
    print(name, "tokens", toks, "min margin", float(min(margins)))


VARIANTS = [
    ("tiny_dense_f16_bias", {"quantization": None, "attention_bias": True, "mlp_bias": True, "tie_word_embeddings": True}, "float16", 122),
    ("tiny_w8_bf16_trad", {"quantization": {"group_size": 64, "bits": 8}, "rope_traditional": True, "tie_word_embeddings": False}, "bfloat16", 202),
]

def vision_fixture():
    """A small Qwen2.5-VL-shaped vision tower (head_dim 80 like the 7B tower, windowed + full-attention blocks, two images, one
    with two frames) from oracle/vision_oracle.py, and a few-row int4 product in the many-row regime (the W4M kernel's contract)."""
    from oracle import vision_oracle as vo
    cfg = dict(depth=3, hidden_size=160, intermediate_size=212, out_hidden_size=96, num_heads=2, patch_size=14, in_channels=3,
               spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1])
    grid = [(1, 6, 10), (2, 4, 4)]
    w = vo.synth_vision_checkpoint(cfg, seed=4, dtype=DT)
    n = sum(t * h * ww for t, h, ww in grid)
    pix = po.round_T(np.random.default_rng(7).standard_normal((n, 3 * 2 * 14 * 14)), DT)
    import json
    d = {"source": "oracle", "dtype": DT, "config_json": json.dumps(cfg), "grid": np.array(grid, np.int32), "pixels": po.to_bits(pix, DT),
         "features": po.to_bits(vo.vision_forward(cfg, w, pix, grid, DT), DT)}
    d.update({"w:" + k: v for k, v in w.items()})
    rng = np.random.default_rng(99)
    wq, sc, bi = po.quantize(po.round_T(rng.standard_normal((96, 1408)) * 0.05, DT), 64, 4, DT)
    x = po.round_T(rng.standard_normal((17, 1408)), DT)
    d["qmm_wq"], d["qmm_scales"], d["qmm_biases"], d["qmm_x"] = wq, sc, bi, po.to_bits(x, DT)
    d["qmm_y"] = po.to_bits(po.quantized_matmul(x, wq, sc, bi, group_size=64, bits=4, dtype=DT, regime="qmm"), DT)
    np.savez_compressed(OUT / "tiny_vision_bf16.npz", **d)


if __name__ == "__main__":
    ops_fixture()
    tiny_llama_fixture()
    for v in VARIANTS:
        variant_fixture(*v)
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size // 1024, "KiB")
    vision_fixture()
