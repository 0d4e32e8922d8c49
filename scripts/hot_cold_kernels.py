#!/usr/bin/env python
"""Per-kernel duration of the decode step's GEMVs with their weights cold (cycling through the layers: every launch streams from HBM)
vs hot (the same layer again and again: <= 66 MB, resident in the 256 MiB Infinity Cache).  Developer tool."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from proxy_inference_engine_amd import InferenceEngine  # noqa: E402
from proxy_inference_engine_amd.models.llama import Model, ModelArgs  # noqa: E402
from proxy_inference_engine_amd.models.utils import LLAMA3_8B, synthetic_checkpoint  # noqa: E402

cfg = dict(LLAMA3_8B)
model = Model(ModelArgs(**cfg), synthetic_checkpoint(cfg, seed=0, dtype=torch.bfloat16))
eng = InferenceEngine(model=model)
prompt = torch.randint(0, cfg["vocab_size"], (128,), generator=torch.Generator().manual_seed(1))
eng.prepare_engine(prompt, temp=0)
gen = eng.generate_step(prompt)
for _ in range(8):
    next(gen)
torch.cuda.synchronize()
L = cfg["num_hidden_layers"]


def timed(name, layers, reps=8):
    for li in layers:
        model.launch_kernel(name, li)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for li in layers:
            model.launch_kernel(name, li)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / len(layers) * 1e3)
    return float(np.median(ts))


for name in ("qkv", "o_proj", "gate_up", "down"):
    cold = timed(name, list(range(L)))
    hot = timed(name, [0] * L)
    print(f"{name:8s}: cold {cold:6.2f} us   hot {hot:6.2f} us   ({model.kernel_bytes(name, 150) / 1e6:.1f} MB)", flush=True)
