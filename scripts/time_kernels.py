#!/usr/bin/env python
"""Per-kernel time of the decode step's launches, each timed alone by event-bracketed sweeps over the layers
(the same method bench.py uses for gate/up).  Developer tool:  python scripts/time_kernels.py [--prompt 128]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=200)
    args = ap.parse_args()
    from proxy_inference_engine_amd.models.llama import Model, ModelArgs
    from proxy_inference_engine_amd.models.utils import LLAMA3_8B, synthetic_checkpoint
    cfg = dict(LLAMA3_8B)
    model = Model(ModelArgs(**cfg), synthetic_checkpoint(cfg, seed=0, dtype=torch.bfloat16))
    cache = model.make_cache()
    ids = torch.randint(0, cfg["vocab_size"], (args.prompt,), generator=torch.Generator().manual_seed(1)).cuda()
    model.step(ids, cache)
    for _ in range(4):
        model.step(None, cache)
    n_l, total = cfg["num_hidden_layers"], 0.0
    for name in ("qkv", "attn", "o_proj", "gate_up", "down"):
        for li in range(n_l):
            model.launch_kernel(name, li)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for li in range(n_l):
                model.launch_kernel(name, li)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / n_l * 1e3)
        t = float(np.median(ts))
        total += t
        print(f"{name:8s} {t:7.2f} us   {model.kernel_bytes(name, args.prompt + 5) / t / 1e6:7.2f} TB/s", flush=True)
    print(f"layer    {total:7.2f} us (launched back to back without the graph: includes host launch gaps)")


if __name__ == "__main__":
    main()
