"""CPU: the oracle must reproduce the committed golden vectors (tests/golden/*.npz, made by
tests/golden/make_golden.py).  Guards the oracle against drift; the same files drive the -m gpu parity tests."""
import json

import numpy as np
import pytest

from oracle import pie_oracle as po


def test_ops_fixture_reproduced(golden_dir):
    g = np.load(golden_dir / "ops_bf16.npz")
    dt = str(g["dtype"])
    assert str(g["source"]) == "oracle"
    for tag in ("gemv4096", "gemv14336"):
        x = po.from_bits(g[f"{tag}_x"], dt)
        y = po.quantized_matmul(x, g[f"{tag}_wq"], g[f"{tag}_scales"], g[f"{tag}_biases"], dtype=dt)
        assert np.array_equal(po.to_bits(y, dt), g[f"{tag}_y"])
    y = po.rms_norm(po.from_bits(g["rms_x"], dt), g["rms_w"], float(g["rms_eps"]), dt)
    assert np.array_equal(po.to_bits(y, dt), g["rms_y"])
    y = po.rope(po.from_bits(g["rope_x"], dt), g["rope_freqs"], int(g["rope_offset"]), dt)
    assert np.array_equal(po.to_bits(y, dt), g["rope_y"])
    o = po.sdpa(po.from_bits(g["sdpa_q"], dt), po.from_bits(g["sdpa_k"], dt), po.from_bits(g["sdpa_v"], dt),
                float(g["sdpa_scale"]), None, dt, True, T=int(g["sdpa_T"]))
    assert np.array_equal(po.to_bits(o, dt), g["sdpa_out"])
    a, b = po.from_bits(g["act_a"], dt), po.from_bits(g["act_b"], dt)
    assert np.array_equal(po.to_bits(po.silu_mul(a, b, dt), dt), g["act_silu_mul"])
    assert np.array_equal(po.to_bits(po.add(a, b, dt), dt), g["act_add"])
    tok, lp = po.logprobs_argmax(po.from_bits(g["tail_logits"], dt))
    assert tok == int(g["tail_token"]) and np.allclose(lp, g["tail_logprobs"], atol=1e-6)


def test_tiny_llama_fixture_reproduced(golden_dir):
    g = np.load(golden_dir / "tiny_llama_w4_bf16.npz")
    dt = str(g["dtype"])
    cfg = json.loads(str(g["config_json"]))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    model = po.OracleLlama(cfg, w, dt)
    pc = po.OraclePromptCache()
    gen = po.generate_step(model, pc, g["prompt"])
    for i, want in enumerate(g["tokens"]):
        tok, lp = next(gen)
        assert tok == int(want), f"step {i}"
        assert np.allclose(lp, g["logprobs"][i], atol=1e-6)
    # prefix reuse: the same prompt again re-processes exactly one token and lands on the same first token
    pc2_first = next(po.generate_step(model, pc, g["prompt"]))[0]
    assert pc2_first == int(g["tokens"][0]) and pc.cache[0].offset == len(g["prompt"])


@pytest.mark.parametrize("name", ["tiny_dense_f16_bias", "tiny_w8_bf16_trad"])
def test_variant_fixtures_reproduced(golden_dir, name):
    """Dense f16 + Linear biases, and int8 g=64 + traditional RoPE: the oracle reproduces its committed tokens / log-probs."""
    g = np.load(golden_dir / f"{name}.npz")
    dt = str(g["dtype"])
    cfg = json.loads(str(g["config_json"]))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    gen = po.generate_step(po.OracleLlama(cfg, w, dt), po.OraclePromptCache(), g["prompt"])
    for i, want in enumerate(g["tokens"]):
        tok, lp = next(gen)
        assert tok == int(want), f"{name} step {i}"
        assert np.allclose(lp, g["logprobs"][i], atol=1e-6)


def test_vision_and_few_row_fixture_reproduced(golden_dir):
    """The vision-tower oracle (oracle/vision_oracle.py) and the many-row regime of quantized_matmul reproduce their committed
    outputs bit for bit (pins both against drift; the GPU tests read the same file)."""
    from oracle import vision_oracle as vo
    g = np.load(golden_dir / "tiny_vision_bf16.npz")
    dt = str(g["dtype"])
    cfg = json.loads(str(g["config_json"]))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    grid = [tuple(int(v) for v in row) for row in g["grid"]]
    feats = vo.vision_forward(cfg, w, po.from_bits(g["pixels"], dt), grid, dt)
    assert np.array_equal(po.to_bits(feats, dt), g["features"])
    y = po.quantized_matmul(po.from_bits(g["qmm_x"], dt), g["qmm_wq"], g["qmm_scales"], g["qmm_biases"], group_size=64, bits=4, dtype=dt, regime="qmm")
    assert np.array_equal(po.to_bits(y, dt), g["qmm_y"])
