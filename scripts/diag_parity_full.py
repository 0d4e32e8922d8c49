"""How chaotic is the benchmarked model?  Teacher-forces the full synthetic Llama-3-8B (32 layers, V = 128256) through two HIP
configurations that differ ONLY in the fp32 summation order of attention (4 merged splits vs 1 split) and prints their logit
distance in bf16 ulps of the largest logit -- the noise floor any comparison with the CPU oracle (a third summation order)
has to be read against.  Used to set bench.py's PARITY_TOL_EPS."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from proxy_inference_engine_amd.models.llama import Model, ModelArgs  # noqa: E402
from proxy_inference_engine_amd.models.utils import LLAMA3_8B, synthetic_checkpoint  # noqa: E402

cfg = dict(LLAMA3_8B)
if len(sys.argv) > 1:
    cfg["num_hidden_layers"] = int(sys.argv[1])
w = synthetic_checkpoint(cfg, seed=0, dtype=torch.bfloat16)
a, b = Model(ModelArgs(**cfg), w, kv_splits=0), Model(ModelArgs(**cfg), w, kv_splits=1)
rng = np.random.default_rng(1)
prompt = torch.from_numpy(rng.integers(0, cfg["vocab_size"], 4)).to(torch.int32).cuda()
ca, cb = a.make_cache(), b.make_cache()
a.step(prompt, ca), b.step(prompt, cb)
tok = torch.tensor([1], dtype=torch.int32, device="cuda")
eps = 2.0 ** -8
worst = worst_rms = 0.0
same = 0
for i in range(32):
    ta, _, la = a.step(tok, ca)
    tb, _, lb = b.step(tok, cb)
    la, lb = la.float().cpu().numpy(), lb.float().cpu().numpy()
    scale = np.abs(la).max()
    worst = max(worst, np.abs(la - lb).max() / (eps * scale))
    worst_rms = max(worst_rms, np.sqrt(np.mean((la - lb) ** 2)) / (eps * np.sqrt(np.mean(la ** 2))))
    same += int(ta.item()) == int(tb.item())
    tok = ta.reshape(1)
print(f"layers {cfg['num_hidden_layers']}: 4 merged splits vs 1 split over 32 teacher-forced steps: max err {worst:.2f} eps, rms err {worst_rms:.2f} eps, same greedy id on {same}/32 steps")
