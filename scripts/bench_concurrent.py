#!/usr/bin/env python
"""Aggregate decode throughput of S independent batch-1 sequences on ONE MI355X, each with its own decoder, KV cache and
HIP stream, driven round-robin from one host thread (the replayed step graphs of different streams overlap on the GPU).

    python scripts/bench_concurrent.py [--streams 1,2,3,4] [--steps 128]

Not the BASELINE.json metric (that is ONE sequence: bench.py); it quantifies the serving-level lever DESIGN.md 5 describes:
a single sequence is bound by per-launch latency, so a second sequence fills the gaps.
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,2,3,4")
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--prompt", type=int, default=128)
    args = ap.parse_args()
    from proxy_inference_engine_amd import InferenceEngine
    from proxy_inference_engine_amd.models.llama import Model, ModelArgs
    from proxy_inference_engine_amd.models.utils import LLAMA3_8B, synthetic_checkpoint

    cfg = dict(LLAMA3_8B)
    weights = synthetic_checkpoint(cfg, seed=0, dtype=torch.bfloat16)
    counts = [int(x) for x in args.streams.split(",")]
    n_max = max(counts)
    engines, streams, gens = [], [], []
    for i in range(n_max):  # every sequence owns a decoder (its own W4S copy: 4.2 GB each, nothing next to 288 GB)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            eng = InferenceEngine(model=Model(ModelArgs(**cfg), weights))
            prompt = torch.randint(0, cfg["vocab_size"], (args.prompt,), generator=torch.Generator().manual_seed(1 + i))
            eng.prepare_engine(prompt, temp=0)
            g = eng.generate_step(prompt)
            next(g)
            for _ in range(16):
                next(g)
        engines.append(eng), streams.append(s), gens.append(g)
    torch.cuda.synchronize()
    for n in counts:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for s, g in zip(streams[:n], gens[:n]):
                with torch.cuda.stream(s):
                    next(g)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{n} sequence(s): {n * args.steps / dt:8.1f} tok/s aggregate, {args.steps / dt:7.1f} tok/s per sequence, "
              f"{1e3 * dt / args.steps:6.3f} ms per round", flush=True)


if __name__ == "__main__":
    main()
