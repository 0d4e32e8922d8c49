"""Diagnostic (GPU box): error statistics of the HIP decode path vs the oracle, used to set the test tolerances.
Prints ulp-distance histograms of the hidden state and logits after prefill and during teacher-forced decode."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import pie_oracle as po  # noqa: E402
from tests._util import codes_dev, to_bits, to_dev, ulp_key  # noqa: E402
from proxy_inference_engine_amd.models.llama import Model, ModelArgs  # noqa: E402


def hist(tag, got_bits, want_bits, dt):
    d = np.abs(ulp_key(got_bits.reshape(-1)) - ulp_key(want_bits.reshape(-1)))
    a = np.abs(po.from_bits(got_bits, dt) - po.from_bits(want_bits, dt))
    print(f"{tag:28s} n={d.size:6d} ulp0={np.mean(d == 0):.4f} ulp1={np.mean(d == 1):.4f} ulp2={np.mean(d == 2):.4f} "
          f"ulp>2={np.mean(d > 2):.5f} max_ulp={d.max()} max_abs={a.max():.5f} max|ref|={np.abs(po.from_bits(want_bits, dt)).max():.3f}")


def run(name, cfg, w, dt, prompt, n_decode):
    dev_w = {k: (codes_dev(v) if v.dtype == np.uint32 else to_dev(v, dt)) for k, v in w.items()}
    model = Model(ModelArgs(**cfg), dev_w)
    orc = po.OracleLlama(cfg, w, dt)
    oc = [po.OracleKVCache() for _ in orc.layers]
    want, hid = orc.forward(prompt, oc, want_hidden=True)
    cache = model.make_cache()
    tok, lp, logits = model.step(torch.from_numpy(prompt).cuda(), cache)
    print(f"== {name} ({dt})")
    hist("prefill hidden(last)", to_bits(model.hidden), po.to_bits(hid[-1], dt), dt)
    hist("prefill logits(last)", to_bits(logits), po.to_bits(want[-1], dt), dt)
    t = int(po.logprobs_argmax(want[-1])[0])
    for i in range(n_decode):
        (want1, hid1) = orc.forward(np.array([t]), oc, want_hidden=True)
        tok, lp, logits = model.step(torch.tensor([t], dtype=torch.int32, device="cuda"), cache)
        hist(f"decode {i} hidden", to_bits(model.hidden), po.to_bits(hid1[0], dt), dt)
        hist(f"decode {i} logits", to_bits(logits), po.to_bits(want1[0], dt), dt)
        otok, olp = po.logprobs_argmax(want1[0])
        top2 = np.sort(olp)[-2:]
        print(f"   token gpu={int(tok.item())} oracle={otok} margin={top2[1]-top2[0]:.4f} max|dlogprob|={np.abs(lp.cpu().numpy()-olp).max():.5f}")
        t = otok


if __name__ == "__main__":
    g = np.load(ROOT / "tests/golden/tiny_llama_w4_bf16.npz")
    cfg = json.loads(str(g["config_json"]))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    run("tiny", cfg, w, "bfloat16", g["prompt"], 4)
    big = {"model_type": "llama", "hidden_size": 4096, "num_hidden_layers": 2, "intermediate_size": 14336,
           "num_attention_heads": 32, "num_key_value_heads": 8, "rms_norm_eps": 1e-5, "vocab_size": 8192,
           "rope_theta": 500000.0, "max_position_embeddings": 8192, "tie_word_embeddings": False,
           "quantization": {"group_size": 64, "bits": 4}}
    for dt in ("bfloat16", "float16"):
        wb = po.synth_checkpoint(big, seed=1, dtype=dt, lm_head_gain=4.0)
        run("8B-shaped x2 layers", big, wb, dt, np.random.default_rng(2).integers(0, 8192, 6), 3)
