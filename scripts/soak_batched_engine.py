"""Soak of the continuous-batching loop on the tiny golden model: 3 rounds of 60 requests with random prompt lengths (1..200),
slot counts, stop tokens and lengths through a 30-page pool; every request must finish and every page return.

    python scripts/soak_batched_engine.py [--kv-int8] [--chunk]    (on an MI355X; --chunk: chunked prefill with a random chunk size per round)
"""
import json, sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_decode import build
from proxy_inference_engine_amd.engine import BatchedEngine
g = np.load('tests/golden/tiny_llama_w4_bf16.npz')
cfg = json.loads(str(g['config_json'])); w = {k[2:]: g[k] for k in g.files if k.startswith('w:')}
model = build(cfg, w)
rng = np.random.default_rng(0)
I8 = "--kv-int8" in sys.argv   # the same soak on int8 pages (per-head scales 1/16: coarse, the point is the plumbing)
kw = {}
if I8:
    L, Hkv = cfg["num_hidden_layers"], cfg["num_key_value_heads"]
    kw = dict(kv_dtype=torch.int8, kv_scales=(torch.full((L, Hkv), 1 / 16, dtype=torch.float16), torch.full((L, Hkv), 1 / 16, dtype=torch.float16)))
CHUNK = "--chunk" in sys.argv
for rnd in range(3):
    if CHUNK:
        kw["prefill_chunk"] = int(rng.integers(1, 80))
    prompts = [rng.integers(0, cfg['vocab_size'], int(n)).tolist() for n in rng.integers(1, 200, 60)]
    eng = BatchedEngine(model, num_pages=30, max_batch=int(rng.integers(2, 12)), stop_tokens=[int(rng.integers(0, cfg['vocab_size']))], **kw)
    out = eng.generate(prompts, int(rng.integers(3, 40)))
    assert all(len(o) >= 1 for o in out) and eng.pool.get_num_free_pages() == eng.pool.size()
    print('round', rnd, 'ok', sum(len(o) for o in out), 'tokens', eng.steps, 'steps', flush=True)
print('soak ok')
