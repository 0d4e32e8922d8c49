"""Checkpoint loading (mirror of models/utils.py:27-125 of the reference) and the synthetic random-weight
checkpoints the benchmarks use (no network: SURVEY.md 8d).

On-disk format consumed: HF-layout `model*.safetensors` with MLX-quantised triplets
`{name}.weight` (uint32 packed codes), `{name}.scales`, `{name}.biases` and
config.json["quantization"] = {"group_size": 64, "bits": 4}  (models/utils.py:96-111).
"""
from __future__ import annotations

import glob
import json
from dataclasses import dataclass
from pathlib import Path
from typing import Any

import torch

from .. import _ffi, hip_ops
from .llama import Model, ModelArgs

_ARCH_ALIASES = {"llama": "llama", "mistral": "llama"}  # models/utils.py:139-155 (the Llama-shaped subset)


@dataclass
class LargeLanguageModel:
    model: Model
    hf_tokenizer: Any
    tokenizer_config: dict


def load_model(model_path: str, kv_splits: int = 0) -> tuple[Model, dict]:
    """Reads config.json + model*.safetensors from a local directory and builds the device model."""
    from safetensors import safe_open

    path = Path(model_path)
    files = sorted(glob.glob(str(path / "model*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no model*.safetensors under {path}")
    with open(path / "config.json") as f:
        config = json.load(f)
    if _ARCH_ALIASES.get(config.get("model_type")) != "llama":
        raise ValueError(f"model_type {config.get('model_type')!r} is not on the MI355X decode path (Llama-shaped only)")
    device = _ffi.require_gpu()
    weights: dict[str, torch.Tensor] = {}
    for wf in files:
        with safe_open(wf, framework="pt", device="cpu") as sf:
            for k in sf.keys():
                t = sf.get_tensor(k)
                if t.dtype in (torch.uint32,):
                    t = t.view(torch.int32)
                weights[k] = t.to(device)
    model = Model(ModelArgs(**config), weights, kv_splits=kv_splits)
    try:
        with open(path / "tokenizer_config.json") as f:
            tokenizer_config = json.load(f)
    except FileNotFoundError:
        tokenizer_config = {}
    return model, tokenizer_config


def load(path_or_hf_repo: str) -> LargeLanguageModel:
    """models/utils.py:27-48.  Local directories only (no network in this build's environment)."""
    model, tokenizer_config = load_model(path_or_hf_repo)
    try:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(path_or_hf_repo)
    except Exception:
        tok = None
    return LargeLanguageModel(model=model, hf_tokenizer=tok, tokenizer_config=tokenizer_config)


LLAMA3_8B = {
    "model_type": "llama", "hidden_size": 4096, "num_hidden_layers": 32, "intermediate_size": 14336,
    "num_attention_heads": 32, "num_key_value_heads": 8, "rms_norm_eps": 1e-5, "vocab_size": 128256,
    "rope_theta": 500000.0, "max_position_embeddings": 8192, "tie_word_embeddings": False,
    "quantization": {"group_size": 64, "bits": 4},
}


LLAMA3_70B = {
    "model_type": "llama", "hidden_size": 8192, "num_hidden_layers": 80, "intermediate_size": 28672,
    "num_attention_heads": 64, "num_key_value_heads": 8, "rms_norm_eps": 1e-5, "vocab_size": 128256,
    "rope_theta": 500000.0, "max_position_embeddings": 8192, "tie_word_embeddings": False,
    "quantization": {"group_size": 64, "bits": 4},
}


LLAMA32_3B = {
    "model_type": "llama", "hidden_size": 3072, "num_hidden_layers": 28, "intermediate_size": 8192,
    "num_attention_heads": 24, "num_key_value_heads": 8, "head_dim": 128, "rms_norm_eps": 1e-5, "vocab_size": 128256,
    "rope_theta": 500000.0, "max_position_embeddings": 8192, "tie_word_embeddings": True,
    "quantization": {"group_size": 64, "bits": 4},
}

# Qwen2-VL-7B's TEXT tower (BASELINE.json configs[3]) expressed in the Llama ModelArgs: q/k/v biases (attention_bias; the
# synthetic checkpoint also gets an o_proj bias, which the real model lacks -- 7 KB per layer), 7 q-heads per kv-head.
QWEN2VL_7B_TEXT = {
    "model_type": "llama", "hidden_size": 3584, "num_hidden_layers": 28, "intermediate_size": 18944,
    "num_attention_heads": 28, "num_key_value_heads": 4, "rms_norm_eps": 1e-6, "vocab_size": 152064,
    "rope_theta": 1000000.0, "max_position_embeddings": 32768, "tie_word_embeddings": False, "attention_bias": True,
    "quantization": {"group_size": 64, "bits": 4},
}


def synthetic_checkpoint(config: dict, seed: int = 0, dtype: torch.dtype = torch.bfloat16, device=None,
                         lm_head_gain: float = 1.0, lm_head_tail: float = 0.0) -> dict[str, torch.Tensor]:
    """Random-weight checkpoint in the reference's on-disk layout, generated and quantised ON the GPU
    (hip_ops.quantize = the HIP restatement of mx.quantize).  Linear/Embedding W ~ N(0, 0.02^2) cast to `dtype`
    then int4 g=64 quantised -- or left dense when the config has no "quantization" entry; norm weights 1 + N(0, 0.02^2).
    lm_head_tail = s > 0 multiplies row v of lm_head by exp(s * N(0, 1)): a heavy-tailed logit distribution, so that the greedy
    token is decided by a clear margin on most steps, as in a trained model.  (A uniform gain cannot do that: it scales the top-2
    gap and the 16-bit error of the logits alike.  With i.i.d. rows the gap between the two largest of 128k logits is ~0.2 sigma,
    the same size as a few bf16 ulps of the largest one: an id comparison would be decided by rounding noise.)"""
    device = device or _ffi.require_gpu()
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    H, I, V = config["hidden_size"], config["intermediate_size"], config["vocab_size"]
    nh = config["num_attention_heads"]
    nkv = config.get("num_key_value_heads") or nh
    D = config.get("head_dim") or H // nh
    out: dict[str, torch.Tensor] = {}

    def put_linear(name: str, N: int, K: int, gain: float = 1.0, bias: bool = False, row_tail: float = 0.0) -> None:
        if bias:
            out[f"{name}.bias"] = (torch.randn(N, generator=gen, device=device, dtype=torch.float32) * 0.1).to(dtype)
        w = torch.randn((N, K), generator=gen, device=device, dtype=torch.float32) * (0.02 * gain)
        if row_tail > 0.0:
            w *= torch.exp(row_tail * torch.randn((N, 1), generator=gen, device=device, dtype=torch.float32))
        w = w.to(dtype)
        if config.get("quantization"):
            out[f"{name}.weight"], out[f"{name}.scales"], out[f"{name}.biases"] = hip_ops.quantize(w, bits=int(config["quantization"]["bits"]))
        else:
            out[f"{name}.weight"] = w

    def norm_w() -> torch.Tensor:
        return (1.0 + 0.02 * torch.randn(H, generator=gen, device=device, dtype=torch.float32)).to(dtype)

    put_linear("model.embed_tokens", V, H)
    for i in range(config["num_hidden_layers"]):
        p = f"model.layers.{i}"
        out[f"{p}.input_layernorm.weight"] = norm_w()
        out[f"{p}.post_attention_layernorm.weight"] = norm_w()
        ab, mb = bool(config.get("attention_bias")), bool(config.get("mlp_bias"))
        put_linear(f"{p}.self_attn.q_proj", nh * D, H, bias=ab)
        put_linear(f"{p}.self_attn.k_proj", nkv * D, H, bias=ab)
        put_linear(f"{p}.self_attn.v_proj", nkv * D, H, bias=ab)
        put_linear(f"{p}.self_attn.o_proj", H, nh * D, bias=ab)
        put_linear(f"{p}.mlp.gate_proj", I, H, bias=mb)
        put_linear(f"{p}.mlp.up_proj", I, H, bias=mb)
        put_linear(f"{p}.mlp.down_proj", H, I, bias=mb)
    out["model.norm.weight"] = norm_w()
    if not config.get("tie_word_embeddings", True):
        put_linear("lm_head", V, H, gain=lm_head_gain, row_tail=lm_head_tail)
    return out
